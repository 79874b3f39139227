#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
  for rounds in 1 2 3; do
    RTT_BACKWARD=retrace timeout -k 10 120 python bench.py --workload spot --no-cpu-baseline --tune-adjoint-rounds $rounds 2>>gpurun_out/r3g_spot.err | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('spot retrace rounds $rounds', 'event_ms %.4f step_ms %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))" || exit 1
  done
done
