#!/bin/bash
# spot step in both backward modes and lane widths; prints one line each
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "adjoint 0" "adjoint 1" "adjoint 2" "tangent 0"; do
  set -- $cfg
  RTT_BACKWARD=$1 timeout -k 10 200 python bench.py --workload spot --no-cpu-baseline --tune-adjoint-lane $2 2>gpurun_out/spot_$1.err | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('spot $1 lane=$2', 'kernel_ms %.4f min %.4f step_ms %.4f' % (d['roofline']['kernel_ms'], d['roofline']['kernel_ms_min'], d['ms_per_step']))" || exit 1
done
