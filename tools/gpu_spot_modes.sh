#!/bin/bash
# spot step for the backward variants given as "mode lane block" triples (default: a sweep); prints one line each
set -u
cd "${GRAFT_REPO_ROOT:-.}"
CFGS=${1:-"adjoint,0,256 adjoint,1,256 adjoint,2,256 adjoint,1,128 adjoint,2,128 tangent,0,256"}
for cfg in $CFGS; do
  IFS=, read mode lane block <<< "$cfg"
  RTT_BACKWARD=$mode timeout -k 10 200 python bench.py --workload spot --no-cpu-baseline --tune-adjoint-lane $lane --tune-adjoint-block $block 2>gpurun_out/spot_$mode.err | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('spot $mode lane=$lane block=$block', 'kernel_ms %.4f min %.4f step_ms %.4f' % (d['roofline']['kernel_ms'], d['roofline']['kernel_ms_min'], d['ms_per_step']))" || exit 1
done
