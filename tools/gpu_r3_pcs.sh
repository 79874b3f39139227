#!/bin/bash
# Try PC sampling of the spot step (reverse sweep with its own trace).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --help 2>&1 | grep -i -A3 "pc-sampling" | head -40
ARGS="--workload spot --no-cpu-baseline --steps 20 --warmup 2"
RTT_BACKWARD=retrace timeout -s KILL 120 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 65536 --output-format csv -d gpurun_out/r3_pcs_stoch -- python3 bench.py $ARGS > gpurun_out/r3_pcs_stoch.log 2>&1
echo "stochastic rc=$?"; tail -5 gpurun_out/r3_pcs_stoch.log; find gpurun_out/r3_pcs_stoch -type f | head; 
RTT_BACKWARD=retrace timeout -s KILL 120 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 1 --output-format csv -d gpurun_out/r3_pcs_trap -- python3 bench.py $ARGS > gpurun_out/r3_pcs_trap.log 2>&1
echo "host_trap rc=$?"; tail -5 gpurun_out/r3_pcs_trap.log; find gpurun_out/r3_pcs_trap -type f | head; 
for f in $(find gpurun_out/r3_pcs_stoch gpurun_out/r3_pcs_trap -name "*.csv" | head -6); do echo "== $f"; head -3 $f; wc -l $f; done
