#!/bin/bash
# Staged bounce loop with late rays placed from records: bit-identity tests, time, HBM traffic.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_scale.py -m gpu -x -q --timeout 300 --timeout-method thread -k "staged or naive or simulate" > gpurun_out/r3j_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r3j_tests.log; [ $rc -ne 0 ] && { tail -60 gpurun_out/r3j_tests.log; exit $rc; }
timeout -k 10 200 python bench.py --workload naive_sim --no-cpu-baseline --steps 20 2>gpurun_out/r3j.err | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('naive_sim step_ms %.3f event_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" || exit 1
bash profiles/run_pmc.sh naive_sim --workload naive_sim > gpurun_out/pmc_naive_sim.txt 2>&1 || { tail -5 gpurun_out/pmc_naive_sim.txt; exit 1; }
python3 profiles/make_traffic.py gpurun_out/prof_naive_sim naive_sim 4194304 gpurun_out/traffic_naive_sim.json 7 || exit 1
python3 -c "import json; d=json.load(open('gpurun_out/traffic_naive_sim.json')); print(d['traffic_bytes_per_launch']/ (4194304*56.0)); print(json.dumps(d['kernels'], indent=1))"
