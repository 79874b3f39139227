#!/bin/bash
# Collect rocprofv3 summaries for bench.py (run on the GPU box from the repo root).
# usage: profiles/run_pmc.sh <tag> [bench args...]
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
STEPS=5; WARMUP=2   # make_traffic.py divides dispatch counts by STEPS + WARMUP
ARGS="--steps $STEPS --warmup $WARMUP --no-cpu-baseline $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $ARGS > $out/trace.log 2>&1 || exit 1
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$name -- python3 bench.py $ARGS > $out/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/pmc_$name.log; }
done
python3 profiles/summarize.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
