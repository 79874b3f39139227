#!/bin/bash
# Nearest-hit iteration: tests that pin the search (culling bit-exactness, step/simulate parity, gradients through the
# bounce loop, renderer), then kernel times of the 32-surface step and of the 100-bounce simulate.
# Usage: tools/gpu_c4.sh [notest]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "${1:-}" != "notest" ]; then
  timeout -k 10 900 python -u -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_render.py tests/test_gpu_autograd.py tests/test_physics_extra.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/c4_tests.log 2>&1
  rc=$?; tail -5 gpurun_out/c4_tests.log; [ $rc -ne 0 ] && exit $rc
fi
for w in naive32 naive_sim; do
  rm -rf gpurun_out/prof_c4_$w
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4_$w -- python3 bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/prof_c4_$w.log 2>&1 || exit 1
  grep '^{' gpurun_out/prof_c4_$w.log | python3 -c "import sys,json
d=json.loads(sys.stdin.readline()); print('$w step_ms %.4f event_ms %.4f min %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_min']))"
  python3 - "$w" <<'PY'
import csv,glob,sys
w=sys.argv[1]
for f in glob.glob(f'gpurun_out/prof_c4_{w}/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'nearest' in row['Name']:
            print("  {Name:60.60s} calls={Calls:>5s} avg_ns={AverageNs:>12s} min_ns={MinNs:>10s}".format(**row))
PY
done
