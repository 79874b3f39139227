#!/bin/bash
# Spot step: tape sweep vs sweep with its own trace, grid of one round of resident blocks vs three rounds.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_adjoint.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3e_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r3e_tests.log; [ $rc -ne 0 ] && { tail -60 gpurun_out/r3e_tests.log; exit $rc; }
for rep in 1 2; do
  for mode in adjoint retrace; do
    for rounds in 1 3; do
    RTT_BACKWARD=$mode timeout -k 10 120 python bench.py --workload spot --no-cpu-baseline --tune-adjoint-rounds $rounds 2>>gpurun_out/r3e_spot.err | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('spot $mode rounds $rounds', 'event_ms %.4f step_ms %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))" || exit 1
    done
  done
done
for mode in adjoint retrace; do
  RTT_BACKWARD=$mode timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3e_trace_$mode -- python3 bench.py --workload spot --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r3e_trace_$mode.log 2>&1 || exit 1
  python3 profiles/summarize.py gpurun_out/r3e_trace_$mode | grep -v "at::native\|rocclr\|Cijk" | head -8
done
