#!/bin/bash
# Profiles: kernel stats + counters of every workload with the current library, traffic files with its hash.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out profiles/r02
for spec in "singlet 67108864" "system5 16777216" "naive32 4194304" "spot 33554432" "naive_sim 4194304" "sample_trace 67108864"; do
  set -- $spec
  bash profiles/run_pmc.sh $1 --workload $1 > gpurun_out/pmc_$1.txt 2>&1 || { echo "pmc $1 failed"; tail -5 gpurun_out/pmc_$1.txt; exit 1; }
  steps=0; [ $1 = naive_sim ] && steps=7   # every kernel of the staged loop runs several times per step: run_pmc.sh makes 5 steps + 2 warm-up
  RTT_PROFILE_STEPS=$steps python3 profiles/make_traffic.py gpurun_out/prof_$1 $1 $2 gpurun_out/traffic_$1.json || exit 1
  grep -v "at::native\|rocclr\|Cijk" gpurun_out/prof_$1/summary.txt > gpurun_out/$1_rocprofv3_summary.txt
done
