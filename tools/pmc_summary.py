"""Dev tool: per-kernel averages of rocprofv3 --pmc counter_collection.csv rows, joined with the kernel trace's durations.
usage: pmc_summary.py <dir> [<dir> ...] > summary.csv   (kernels whose name contains 'czc::' only; one row per kernel name)
Derived columns (when the counters are present): mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8),
clock_ghz = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS section: the counter sums the 8 XCDs)."""
import csv, glob, os, sys
from collections import defaultdict

def short(k):
    return k.split("(")[0].replace("void ", "").strip()

acc = defaultdict(lambda: [0.0, 0])
dur = defaultdict(lambda: [0.0, 0])
for arg in sys.argv[1:]:
    for f in glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "czc::" not in k:
                    continue
                a = acc[(short(k), row["Counter_Name"])]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    for f in glob.glob(os.path.join(arg, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "czc::" not in k:
                    continue
                d = dur[short(k)]
                d[0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3; d[1] += 1
kernels = sorted({k for k, _ in acc}, key=lambda k: -(dur[k][0]))
counters = sorted({c for _, c in acc})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches", "avg_us_profiled", "total_us_profiled"] + counters + ["mfma_util", "clock_ghz"])
for k in kernels:
    n = max(acc[(k, c)][1] for c in counters if (k, c) in acc)
    val = {c: acc[(k, c)][0] / acc[(k, c)][1] for c in counters if (k, c) in acc and acc[(k, c)][1]}
    us = dur[k][0] / dur[k][1] if dur[k][1] else 0.0
    util = clock = ""
    if "GRBM_GUI_ACTIVE" in val and val["GRBM_GUI_ACTIVE"] > 0:
        cyc = val["GRBM_GUI_ACTIVE"] / 8.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in val:
            util = f"{val['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc):.4f}"
        if us > 0:
            clock = f"{cyc / (us * 1e3):.3f}"
    w.writerow([k, n, f"{us:.2f}", f"{dur[k][0]:.1f}"] + [f"{val[c]:.1f}" if c in val else "" for c in counters] + [util, clock])
