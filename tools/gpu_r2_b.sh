#!/bin/bash
# GPU pass B: reverse-mode tests, spot step in both modes, kernel trace of the spot step.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_adjoint.py tests/test_gpu_two_ranks.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/adj2.log 2>&1
rc=$?; tail -30 gpurun_out/adj2.log; [ $rc -ne 0 ] && exit $rc
for mode in adjoint tangent; do
  for lane in 0 1 2; do
  [ $mode = tangent ] && [ $lane != 0 ] && continue
  RTT_BACKWARD=$mode timeout -k 10 200 python bench.py --workload spot --no-cpu-baseline --tune-adjoint-lane $lane 2>gpurun_out/spot_$mode.err | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('spot $mode lane=$lane', 'kernel_ms %.4f min %.4f step_ms %.4f' % (d['roofline']['kernel_ms'], d['roofline']['kernel_ms_min'], d['ms_per_step']))" || exit 1
  done
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_spot_adj -- python3 bench.py --workload spot --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/prof_spot_adj.log 2>&1 || exit 1
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_spot_adj/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        print("  {Name:70.70s} calls={Calls:>5s} avg_ns={AverageNs:>12s} pct={Percentage}".format(**row))
for f in glob.glob('gpurun_out/prof_spot_adj/**/*kernel_trace.csv', recursive=True):
    seen=set()
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'][:60]
        if k in seen: continue
        seen.add(k)
        print(k, 'VGPR', row.get('VGPR_Count'), 'accum', row.get('Accum_VGPR_Count'), 'SGPR', row.get('SGPR_Count'), 'LDS', row.get('LDS_Block_Size'), 'scratch', row.get('Scratch_Size'), 'grid', row.get('Grid_Size'))
PY
