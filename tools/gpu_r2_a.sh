#!/bin/bash
# GPU pass A: full GPU test suite, default bench, 2-rank gloo rehearsal, other workloads, counters of system5 / naive32.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?; tail -15 gpurun_out/gpu_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || exit 1
cut -c1-600 gpurun_out/bench_default.json
timeout -k 10 200 python bench.py --gpus 2 --backend gloo --rays 4194304 --no-cpu-baseline > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err || { tail -20 gpurun_out/bench_gloo2.err; exit 1; }
cut -c1-400 gpurun_out/bench_gloo2.json
: > gpurun_out/all_workloads.txt
for w in system5 naive32 naive_sim spot sample_trace; do
  timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline 2>>gpurun_out/all_workloads.err | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w', 'kernel_ms %.4f min %.4f step_ms %.4f value %.4g frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['kernel_ms_min'], d['ms_per_step'], d['value'], d['roofline']['frac']))" >> gpurun_out/all_workloads.txt || exit 1
done
cat gpurun_out/all_workloads.txt
bash profiles/run_pmc.sh system5 --workload system5 > gpurun_out/pmc_system5.txt 2>&1 || exit 1
bash profiles/run_pmc.sh naive32 --workload naive32 > gpurun_out/pmc_naive32.txt 2>&1 || exit 1
tail -40 gpurun_out/pmc_system5.txt
