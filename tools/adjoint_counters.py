"""Per-kernel latency picture from rocprofv3 --pmc passes: tools/adjoint_counters.py <dir> <prefix> [<prefix> ...] [kernel substring]"""
import csv, glob, sys
from collections import defaultdict
root, prefixes = sys.argv[1], [a for a in sys.argv[2:] if a.startswith("r")]
which = [a for a in sys.argv[2:] if not a.startswith("r")] or ["adjoint"]
for pre in prefixes:
    for sub in which:
        agg = defaultdict(list)
        for f in glob.glob(f"{root}/{pre}_*/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if sub in row["Kernel_Name"]:
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        m = {k: sum(v) / len(v) for k, v in agg.items()}
        if "GRBM_GUI_ACTIVE" not in m:
            continue
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        g = lambda k: m.get(k, float("nan"))
        wc = g("SQ_WAVE_CYCLES")
        print(f"== {pre} kernel~{sub}: cycles/XCD {cyc:.4g}")
        print("   busy per SIMD: valu %.3f sca %.3f lds %.3f vmem %.3f misc %.3f" % tuple(g(k) * 4 / 1024 / cyc for k in
              ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC")))
        print("   per wave-cycle: active %.3f wait_any %.3f wait_inst %.3f wait_lds %.3f; waves resident per SIMD %.2f" % (
            g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc, wc * 4 / 1024 / cyc))
        print("   per wave: valu %.0f salu %.0f smem %.0f lds %.0f vmem_rd %.0f vmem_wr %.0f branch %.0f ifetch %.0f" % tuple(
            g(k) / g("SQ_WAVES") for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH", "SQ_IFETCH")))
        print("   mean latency (level / count, cycles): smem %.0f lds %.0f vmem %.0f ifetch %.0f" % (
            g("SQ_INST_LEVEL_SMEM") / g("SQ_INSTS_SMEM"), g("SQ_INST_LEVEL_LDS") / g("SQ_INSTS_LDS"),
            g("SQ_INST_LEVEL_VMEM") / (g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")), g("SQ_IFETCH_LEVEL") / g("SQ_IFETCH")))
