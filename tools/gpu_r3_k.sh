#!/bin/bash
# Staged bounce loop, same box: late rays placed from records (default) against every stage writing the planes itself.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 1 0 1 0; do
  timeout -k 10 200 python bench.py --workload naive_sim --no-cpu-baseline --steps 20 --tune-late-records $v 2>gpurun_out/r3k.err | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('late records $v: step_ms %.3f event_ms %.3f algorithmic MB %.1f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['algorithmic_bytes_per_launch']/1e6))" || exit 1
done
for v in 1 0; do
  bash profiles/run_pmc.sh naive_sim_$v --workload naive_sim --tune-late-records $v > gpurun_out/pmc_naive_sim_$v.txt 2>&1 || { tail -5 gpurun_out/pmc_naive_sim_$v.txt; exit 1; }
  python3 profiles/make_traffic.py gpurun_out/prof_naive_sim_$v naive_sim 4194304 gpurun_out/traffic_naive_sim_$v.json 7 > /dev/null || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/traffic_naive_sim_$v.json')); print('late records $v: HBM MB per simulate %.1f' % (d['traffic_bytes_per_launch']/1e6), {k[:16]: (round(x['FETCH_SIZE_KB']*2/1024,1), round(x['WRITE_SIZE_KB']/1024,1), x['launches_per_step']) for k, x in d['kernels'].items()})"
done
