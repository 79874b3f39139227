#!/bin/bash
# Latency counters of the reverse sweep: with its own trace (compact states) and from a tape.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--workload spot --no-cpu-baseline --steps 4 --warmup 1"
for mode in retrace adjoint; do
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
              "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
              "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
              "GRBM_GUI_ACTIVE"; do
    name=$(echo $pass | cut -d' ' -f1)
    RTT_BACKWARD=$mode timeout -s KILL 90 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/r3d_pmc_${mode}_$name -- python3 bench.py $ARGS > gpurun_out/r3d_pmc_${mode}_$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/r3d_pmc_${mode}_$name.log; }
  done
done
python3 tools/adjoint_counters.py gpurun_out r3d_pmc_retrace r3d_pmc_adjoint
