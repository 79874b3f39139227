#!/bin/bash
# GPU tests and the bench lines of every workload with the current library (no counter passes).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?; tail -4 gpurun_out/gpu_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || exit 1
: > gpurun_out/all_workloads.jsonl
for w in system5 naive32 naive_sim spot sample_trace; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline 2>>gpurun_out/all_workloads.err >> gpurun_out/all_workloads.jsonl || exit 1
done
timeout -k 10 200 python bench.py --rays 1048576 --no-cpu-baseline >> gpurun_out/all_workloads.jsonl 2>>gpurun_out/all_workloads.err || exit 1
python3 - <<'PY'
import json
for f in ("gpurun_out/bench_default.json", "gpurun_out/all_workloads.jsonl"):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print(d["config"]["workload"][:24], "rays", d["config"].get("rays_per_gpu"),
                  "kernel_ms %.4f min %.4f step_ms %.4f value %.4g frac %.3f" % (
                      r["kernel_ms"], r["kernel_ms_min"] or float("nan"), d["ms_per_step"], d["value"], r["frac"]))
PY
