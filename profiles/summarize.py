"""Summarise rocprofv3 CSV output (kernel trace stats + PMC passes) into one text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


print(f"# rocprofv3 summary of {root}")
for f in find("*kernel_stats.csv"):
    print(f"\n## kernel stats ({os.path.relpath(f, root)})")
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("  {Name:60.60s} calls={Calls:>5s} avg_ns={AverageNs:>12s} min_ns={MinNs:>10s} max_ns={MaxNs:>10s} pct={Percentage}".format(**row))

for f in find("*counter_collection.csv"):
    agg = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f"\n## counters ({os.path.relpath(f, root)})  [mean per dispatch]")
    for k, cs in agg.items():
        if not any(t in k for t in ("trace_seq", "simulate_nearest", "ray_cast", "spot_moments", "sample_bundle", "tangent", "adjoint", "sum_partials")):
            continue
        for c, v in cs.items():
            print(f"  {k[:50]:50s} {c:28s} n={len(v):3d} mean={sum(v) / len(v):.6g}")
