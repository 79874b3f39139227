#!/bin/bash
# Gradient tests with the tape-based reverse sweep, then the spot step swept from a tape against the sweep that traces
# again (three alternations) and a kernel trace of each.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_adjoint.py tests/test_gpu_autograd.py tests/test_zz_no_alloc.py tests/test_gpu_two_ranks.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3c_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r3c_tests.log; [ $rc -ne 0 ] && { tail -60 gpurun_out/r3c_tests.log; exit $rc; }
for rep in 1 2 3; do
  for mode in adjoint retrace; do
    RTT_BACKWARD=$mode timeout -k 10 120 python bench.py --workload spot --no-cpu-baseline 2>>gpurun_out/r3c_spot.err | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('spot $mode', 'event_ms %.4f step_ms %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))" || exit 1
  done
done
for mode in adjoint; do
  RTT_BACKWARD=$mode timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3c_trace_$mode -- python3 bench.py --workload spot --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r3c_trace_$mode.log 2>&1 || exit 1
  python3 profiles/summarize.py gpurun_out/r3c_trace_$mode | grep -v "at::native\|rocclr\|Cijk" | head -12
done
