#!/bin/bash
# Counter passes of the spot step with the reverse sweep's states in registers (0) and in LDS (1).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 60 rocprofv3 --list-avail > gpurun_out/r3b_avail.txt 2>&1
grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_ACTIVE[A-Z_]*" gpurun_out/r3b_avail.txt | sort -u | tr '\n' ' '
echo
ARGS="--workload spot --no-cpu-baseline --steps 4 --warmup 1"
for st in 0 1; do
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
              "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_VMEM_RD" \
              "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES GRBM_GUI_ACTIVE"; do
    name=$(echo $pass | cut -d' ' -f1)
    timeout -s KILL 90 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/r3b_pmc_${st}_$name -- python3 bench.py $ARGS --tune-adjoint-state $st > gpurun_out/r3b_pmc_${st}_$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/r3b_pmc_${st}_$name.log; }
  done
  echo "=== state $st"
  python3 profiles/summarize.py gpurun_out | grep "r3b_pmc_${st}_" -A40 | grep "adjoint" | sed 's/void (anonymous namespace):://' | cut -c1-30,50-140
done
