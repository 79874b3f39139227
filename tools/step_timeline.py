"""Timeline of one optimisation step from a rocprofv3 kernel trace: python tools/step_timeline.py <dir> [anchor]
Prints, for the last repetition of the anchor kernel (default k_trace_seq), every kernel up to the next anchor with
its start offset, duration and the idle gap in front of it."""
import csv
import glob
import sys

root = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_trace_seq<"
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {r['Kernel_Name'][:90]}")
    prev_end = e
    busy += e - s
print(f"step {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {b - a} launches")
