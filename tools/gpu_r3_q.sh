#!/bin/bash
# Bounce loops without the generic visit (rtt_tune key 12): bit-identity tests, time (alternating, same box), traffic.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_fuzzy.py tests/test_physics_extra.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3q_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r3q_tests.log; [ $rc -ne 0 ] && { tail -60 gpurun_out/r3q_tests.log; exit $rc; }
for v in 1 0 1 0 1 0; do
  timeout -k 10 200 python bench.py --workload naive_sim --no-cpu-baseline --steps 20 --tune-lean-bounces $v 2>gpurun_out/r3q.err | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('lean bounce loop $v: step_ms %.3f event_ms mean %.3f min %.3f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min']))" || exit 1
done
for v in 1 0; do
  rm -rf gpurun_out/prof_naive_sim_$v
  bash profiles/run_pmc.sh naive_sim_$v --workload naive_sim --tune-lean-bounces $v > gpurun_out/pmc_naive_sim_$v.txt 2>&1 || { tail -5 gpurun_out/pmc_naive_sim_$v.txt; exit 1; }
  python3 profiles/make_traffic.py gpurun_out/prof_naive_sim_$v naive_sim 4194304 gpurun_out/traffic_naive_sim_$v.json 7 > /dev/null || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/traffic_naive_sim_$v.json')); print('lean bounce loop $v: HBM MB per simulate %.1f' % (d['traffic_bytes_per_launch']/1e6), {k[:16]: (round(x['FETCH_SIZE_KB']*2/1024,1), round(x['WRITE_SIZE_KB']/1024,1), x['launches_per_step']) for k, x in d['kernels'].items()})"
done
