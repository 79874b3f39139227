"""Turn the rocprofv3 passes of profiles/run_pmc.sh into the traffic file bench.py attaches to its roofline:
    python profiles/make_traffic.py gpurun_out/prof_<tag> <workload> <rays per GPU> <out.json> <steps>
<steps> = the steps the profiled command executed, warm-up included (run_pmc.sh: --steps 5 --warmup 2 -> 7); the
dispatches of a kernel divided by it are its launches per step.
HBM bytes per step = sum over the workload's own kernels of (FETCH_SIZE x 2 + WRITE_SIZE) KB per dispatch x dispatches
per step (FETCH_SIZE counts 128-byte requests as 64 bytes on gfx950; WRITE_SIZE is exact), from separate --pmc passes.
The file records the SHA-256 of the kernel library the passes ran with; bench.py ignores it for any other build."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

root, workload, rays, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
if len(sys.argv) < 6 and not os.environ.get("RTT_PROFILE_STEPS"):
    raise SystemExit("make_traffic.py: give the number of steps the profiled command ran (steps + warm-up) as the fifth argument")
steps = int(sys.argv[5]) if len(sys.argv) > 5 else int(os.environ["RTT_PROFILE_STEPS"])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("RTT_HIP_LIB") or os.path.join(ROOT, "raytracetorch_amd", "csrc", "librtt_hip.so")
OWN = ("k_trace_seq", "k_simulate_nearest", "k_place_late", "k_ray_cast", "k_spot_moments", "k_sum_partials", "k_sample_bundle")


def short(name):
    for k in OWN:
        if k in name:
            tail = name.split(k, 1)[1]
            return k + (tail.split("(")[0] if tail.startswith(("_", "<")) else "")
    return None


counters = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k and not k.startswith("k_sample_bundle"):
            counters[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
stats = {}
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Name"])
        if k and not k.startswith("k_sample_bundle"):
            stats[k] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "min_us": float(row["MinNs"]) / 1e3}
# The step count is given, never inferred from the dispatch counts: a command may dispatch a kernel outside its steps,
# and the stages of the bounce loop launch every kernel several times per step.
kernels, total = {}, 0.0
for k, cs in counters.items():
    mean = {c: sum(v) / len(v) for c, v in cs.items()}
    per_step = stats.get(k, {}).get("calls", steps) / steps
    bytes_launch = (2.0 * mean.get("FETCH_SIZE", 0.0) + mean.get("WRITE_SIZE", 0.0)) * 1024.0
    entry = {"launches_per_step": per_step, "FETCH_SIZE_KB": mean.get("FETCH_SIZE"), "WRITE_SIZE_KB": mean.get("WRITE_SIZE"),
             "hbm_bytes_per_launch": bytes_launch, **stats.get(k, {})}
    if "SQ_ACTIVE_INST_VALU" in mean and "GRBM_GUI_ACTIVE" in mean:
        # SQ_ACTIVE_INST_* count quad-cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs
        cyc = mean["GRBM_GUI_ACTIVE"] / 8.0
        entry["valu_busy"] = mean["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc
        if "SQ_ACTIVE_INST_SCA" in mean:
            entry["salu_busy"] = mean["SQ_ACTIVE_INST_SCA"] * 4.0 / 1024.0 / cyc
        entry["valu_insts_per_ray"] = mean.get("SQ_INSTS_VALU", 0.0) * 64.0 / max(rays, 1) if "SQ_INSTS_VALU" in mean else None
        if "SQ_WAVES" in mean:
            entry["waves"] = mean["SQ_WAVES"]
    kernels[k] = entry
    total += bytes_launch * per_step
doc = {"workload": workload, "rays_per_gpu": rays, "steps_profiled": steps,
       "backward_mode": os.environ.get("RTT_BACKWARD", "adjoint") if workload == "spot" else None, "lib_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],
       "source": f"rocprofv3 --kernel-trace --stats and separate --pmc passes of `bench.py --workload {workload}` (profiles/run_pmc.sh), "
                 f"means per dispatch",
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2; WRITE_SIZE exact",
       "traffic_bytes_per_launch": int(round(total)), "kernels": kernels,
       "note": "traffic_bytes_per_launch sums the HBM bytes of every kernel of one step (hbm_bytes_per_launch x launches_per_step); valu_busy / salu_busy = issue cycles of the "
               "vector / scalar unit per SIMD over the kernel's cycles"}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else "...") for k, v in doc.items()}))
for k, e in kernels.items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items()})
