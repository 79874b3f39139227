#!/bin/bash
# Spot step under several grid sizes of the second-moment pass (every block ends in float64 atomics on the same addresses).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for rep in 1 2; do
for b in 512 256 384 768 192; do
  timeout -k 10 200 python bench.py --workload spot --no-cpu-baseline --steps 30 --tune-moments-blocks $b 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('moment blocks $b: step event ms mean %.4f min %.4f median %.4f' % (r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median']))" || exit 1
done
done
