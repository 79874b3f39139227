#!/bin/bash
# bench.py --workload naive_sim under several stage schedules of the staged bounce loop
for sch in "7,2,2,3,5,8,16" "4,3,2,2,3,5,8,16" "3,2,2,2,2,3,5,8,16" "5,2,2,2,3,5,8,16" "2,2,2,2,2,2,3,5,8,16" "3,2,2,1,1,1,2,3,5,8,16" "2,1,1,1,1,1,1,1,2,2,3,5,8,16"; do
  RTT_STAGE_BOUNCES=$sch timeout -k 10 120 python bench.py --workload naive_sim --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$sch', 'step_ms %.3f' % d['ms_per_step'])" || exit 1
done
