#!/bin/bash
# Last pass of the round: the GPU suite as the round end runs it, smoke(), the default bench line.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?; tail -2 gpurun_out/gpu_tests.log; [ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests.log; exit $rc; }
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids || exit 1
timeout -k 10 300 python bench.py 2>gpurun_out/bench_last.err | tee gpurun_out/bench_last.json | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('value %.4g ms_per_step %.4f frac %.3f traffic %s (%s) cpu %s' % (d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['traffic_source'], d['cpu_baseline']['value']))"
