#!/bin/bash
# per-stage kernel times of the staged bounce loop: rocprofv3 kernel trace of bench.py --workload naive_sim, one line per stage
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/prof_st
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/prof_st -o st --output-format csv -- python bench.py --workload naive_sim --no-cpu-baseline --steps 3 --warmup 1 > /tmp/st.log 2>&1 || { tail -5 /tmp/st.log; exit 1; }
f=$(find /tmp/prof_st -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_simulate_nearest" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
grids = [int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)) for r in rows]
n = 8   # stages per call of the default schedule
calls = [durs[i:i + n] for i in range(0, len(durs), n)]
print("kernel dispatches", len(durs), "-> calls of", n, "stages")
for c in calls[-2:]:
    print("stage us:", " ".join(f"{d:8.1f}" for d in c), " total %.1f" % sum(c))
PY
