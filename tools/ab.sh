#!/bin/bash
# A/B of two builds of librtt_hip.so on one box: tools/ab.sh <lib_a> <lib_b> [workloads...]
# Alternates the two libraries three times per workload and prints kernel ms for each run.
A=$1; B=$2; shift 2
WL=${@:-singlet system5 naive32 spot}
for w in $WL; do
  for rep in 1 2 3; do
    for L in $A $B; do
      RTT_HIP_LIB=$PWD/$L timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline 2>&1 | \
        python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w', '$L'.split('/')[-1], 'kernel_ms %.4f step_ms %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))" || exit 1
    done
  done
done
