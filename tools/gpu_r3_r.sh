#!/bin/bash
# Full GPU suite with the lean instantiations, then the one-bounce step with and without them (same box, alternating).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?; tail -2 gpurun_out/gpu_tests.log; [ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests.log; exit $rc; }
for v in 1 0 1 0 1 0; do
  timeout -k 10 200 python bench.py --workload naive32 --no-cpu-baseline --steps 100 --tune-lean-bounces $v 2>gpurun_out/r3r.err | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('lean step $v: step_ms %.4f event_ms mean %.4f min %.4f median %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median']))" || exit 1
done
