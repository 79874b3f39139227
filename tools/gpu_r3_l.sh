#!/bin/bash
# Staged bounce loop: 4 wavefronts per SIMD with 40 B of scratch per lane (default) against 3 wavefronts without scratch.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
EXP=$PWD/raytracetorch_amd/csrc/librtt_hip_exp.so
run() {  # label, env
  env $2 timeout -k 10 200 python bench.py --workload naive_sim --no-cpu-baseline --steps 20 --tune-late-records $3 2>gpurun_out/r3l.err | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 late records $3: step_ms %.3f event_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" || exit 1
}
for rep in 1 2; do
  run "4 waves" "RTT_X=0" 1; run "3 waves" "RTT_HIP_LIB=$EXP" 1; run "4 waves" "RTT_X=0" 0; run "3 waves" "RTT_HIP_LIB=$EXP" 0
done
export RTT_HIP_LIB=$EXP
for v in 1 0; do
  bash profiles/run_pmc.sh naive_sim_w3_$v --workload naive_sim --tune-late-records $v > gpurun_out/pmc_naive_sim_w3_$v.txt 2>&1 || { tail -5 gpurun_out/pmc_naive_sim_w3_$v.txt; exit 1; }
  python3 profiles/make_traffic.py gpurun_out/prof_naive_sim_w3_$v naive_sim 4194304 gpurun_out/traffic_naive_sim_w3_$v.json 7 > /dev/null || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/traffic_naive_sim_w3_$v.json')); print('3 waves, late records $v: HBM MB per simulate %.1f' % (d['traffic_bytes_per_launch']/1e6), {k[:16]: (round(x['FETCH_SIZE_KB']*2/1024,1), round(x['WRITE_SIZE_KB']/1024,1), x['launches_per_step']) for k, x in d['kernels'].items()})"
done
