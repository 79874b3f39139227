#!/bin/bash
# Evidence pass: GPU tests, bench lines of every workload, 2-rank rehearsal, single-rank RCCL, profiles + traffic files.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?; tail -4 gpurun_out/gpu_tests.log; [ $rc -ne 0 ] && exit $rc
bash tools/gpu_profiles_r2.sh > gpurun_out/profiles_r2.log 2>&1 || { tail -20 gpurun_out/profiles_r2.log; exit 1; }
mkdir -p profiles/r02
cp gpurun_out/traffic_*.json profiles/r02/ 2>/dev/null
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || exit 1
: > gpurun_out/all_workloads.jsonl
for w in system5 naive32 naive_sim spot sample_trace; do
  timeout -k 10 300 python bench.py --workload $w 2>>gpurun_out/all_workloads.err >> gpurun_out/all_workloads.jsonl || exit 1
done
timeout -k 10 200 python bench.py --rays 1048576 --no-cpu-baseline >> gpurun_out/all_workloads.jsonl 2>>gpurun_out/all_workloads.err || exit 1
timeout -k 10 200 python bench.py --rays 1048576 --graph --no-cpu-baseline >> gpurun_out/all_workloads.jsonl 2>>gpurun_out/all_workloads.err || exit 1
timeout -k 10 200 python bench.py --gpus 2 --backend gloo --rays 4194304 --no-cpu-baseline > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err || exit 1
timeout -k 10 200 python bench.py --gpus 2 --backend gloo --workload spot --rays 4194304 --no-cpu-baseline > gpurun_out/bench_gloo2_spot.json 2>> gpurun_out/bench_gloo2.err || exit 1
timeout -k 10 200 python bench.py --dist-single --workload spot --no-cpu-baseline > gpurun_out/bench_rccl1_spot.json 2> gpurun_out/bench_rccl1.err || exit 1
RTT_BACKWARD=tangent timeout -k 10 200 python bench.py --workload spot --no-cpu-baseline > gpurun_out/bench_spot_tangent.json 2>/dev/null || exit 1
timeout -k 10 100 python tools/small_n_latency.py > gpurun_out/small_n_latency.txt 2>&1 || exit 1
timeout -k 10 200 python bench.py --workload sample_trace --two-launches --no-cpu-baseline >> gpurun_out/all_workloads.jsonl 2>>gpurun_out/all_workloads.err || exit 1
timeout -k 10 200 python bench.py --workload naive_sim --step-loop --no-cpu-baseline --steps 5 --warmup 1 >> gpurun_out/all_workloads.jsonl 2>>gpurun_out/all_workloads.err || exit 1
timeout -k 10 100 python tools/step_latency.py > gpurun_out/step_latency.txt 2>&1 || exit 1
timeout -k 10 200 python tools/sim_benchmark.py --fused > gpurun_out/sim_benchmark.jsonl 2>/dev/null || exit 1
bash tools/stage_times.sh > gpurun_out/stage_times.txt 2>&1 || exit 1
timeout -k 10 200 python tools/c3_breakdown.py > gpurun_out/c3_breakdown.txt 2>&1 || exit 1
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1 || { cat gpurun_out/smoke.txt; exit 1; }
python3 - <<'PY'
import json
for f in ("gpurun_out/bench_default.json", "gpurun_out/all_workloads.jsonl", "gpurun_out/bench_gloo2.json", "gpurun_out/bench_gloo2_spot.json",
          "gpurun_out/bench_rccl1_spot.json", "gpurun_out/bench_spot_tangent.json"):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print(f.split("/")[-1], d["config"]["workload"][:24], "n_gpus", d["n_gpus"], "rays", d["config"].get("rays_per_gpu"),
                  "kernel_ms %.4f min %.4f med %.4f step_ms %.4f value %.4g frac %.3f traffic %s" % (
                      r["kernel_ms"], r["kernel_ms_min"] or float("nan"), r["kernel_ms_median"] or float("nan"), d["ms_per_step"],
                      d["value"], r["frac"], r["traffic"]),
                  "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
cat gpurun_out/small_n_latency.txt gpurun_out/step_latency.txt gpurun_out/stage_times.txt gpurun_out/c3_breakdown.txt gpurun_out/smoke.txt | grep -v amdgpu.ids
grep fused gpurun_out/sim_benchmark.jsonl | cut -c1-200
