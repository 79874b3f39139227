#!/bin/bash
# Host cost per call after the prepared launches and the one-pass parameter snapshot; parity tests of the paths touched.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_abi_and_host.py tests/test_project.py tests/test_fuzzy.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3h_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r3h_tests.log; [ $rc -ne 0 ] && { tail -60 gpurun_out/r3h_tests.log; exit $rc; }
timeout -k 10 200 python tools/step_latency.py | tee gpurun_out/r3h_step_latency.txt || exit 1
timeout -k 10 200 python tools/small_n_latency.py | tee gpurun_out/r3h_small_n_latency.txt || exit 1
for w in "--workload naive32" "--rays 1048576"; do
  timeout -k 10 200 python bench.py $w --no-cpu-baseline 2>>gpurun_out/r3h.err | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$w', 'event_ms %.4f step_ms %.4f' % (r['kernel_ms'], d['ms_per_step']))" || exit 1
done
