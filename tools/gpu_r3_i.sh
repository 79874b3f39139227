#!/bin/bash
# C3 regrouping bound, share of rays still moving per bounce of the 100-bounce loop, collective time of a one-rank RCCL group.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 python -u tools/c3_sorted_experiment.py > gpurun_out/r3i_c3_regroup.txt 2>gpurun_out/r3i.err || { tail -20 gpurun_out/r3i.err; exit 1; }
cat gpurun_out/r3i_c3_regroup.txt
BOUNCES=30 timeout -k 10 200 python -u tools/bounce_coherence.py > gpurun_out/r3i_bounce_coherence.txt 2>>gpurun_out/r3i.err || { tail -20 gpurun_out/r3i.err; exit 1; }
cut -c1-60 gpurun_out/r3i_bounce_coherence.txt
timeout -k 10 300 python bench.py --workload spot --dist-single --no-cpu-baseline 2>>gpurun_out/r3i.err | tee gpurun_out/r3i_spot_dist_single.json | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config'], d['ms_per_step'], d['roofline']['kernel_ms'])" || exit 1
