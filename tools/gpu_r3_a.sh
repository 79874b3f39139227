#!/bin/bash
# gradient tests with the register-resident reverse sweep, then the spot step with states in
# registers against states in LDS (three alternations), with a kernel trace of each.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_adjoint.py tests/test_gpu_autograd.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3a_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r3a_tests.log; [ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for st in 0 1 2; do
    timeout -k 10 120 python bench.py --workload spot --no-cpu-baseline --tune-adjoint-state $st 2>>gpurun_out/r3a_spot.err | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('spot state=$st (0 compact LDS, 1 full LDS, 2 registers)', 'event_ms %.4f step_ms %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))" || exit 1
  done
done
for st in 0; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3a_trace_$st -- python3 bench.py --workload spot --no-cpu-baseline --steps 10 --warmup 2 --tune-adjoint-state $st > gpurun_out/r3a_trace_$st.log 2>&1 || exit 1
  python3 profiles/summarize.py gpurun_out/r3a_trace_$st | grep -v "at::native\|rocclr\|Cijk" | head -12
done
