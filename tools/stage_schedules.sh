#!/bin/bash
# bench.py --workload naive_sim under several stage schedules of the staged bounce loop (a later first hand-over moves
# fewer rays through the compact planes and leaves fewer rays to be placed one by one)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for sch in ${SCHEDULES:-"7,2,2,3,5,8,16" "8,2,3,5,8,16" "9,2,3,5,8,16" "10,3,5,8,16" "9,3,4,8,16" "8,3,5,8,16"}; do
  RTT_STAGE_BOUNCES=$sch timeout -k 10 120 python bench.py --workload naive_sim --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$sch', 'step_ms %.3f event_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" || exit 1
done
