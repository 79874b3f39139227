#!/bin/bash
# Reverse sweep: the instantiation that holds only the visits of a lens system (rtt_tune key 11), same box, alternating;
# then the gradient tests with it.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 1 0 1 0 1 0; do
  timeout -k 10 200 python bench.py --workload spot --no-cpu-baseline --steps 30 --tune-lens-set $v 2>gpurun_out/r3o.err | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('lens set $v: step event ms mean %.4f min %.4f median %.4f' % (r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median']))" || exit 1
done
timeout -k 10 800 python -u -m pytest tests/test_gpu_adjoint.py tests/test_gpu_autograd.py tests/test_zz_no_alloc.py tests/test_gpu_two_ranks.py tests/test_gpu_scale.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3o_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r3o_tests.log; [ $rc -ne 0 ] && { tail -60 gpurun_out/r3o_tests.log; exit $rc; }
exit 0
