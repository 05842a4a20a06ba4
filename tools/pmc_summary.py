"""Dev tool: average rocprofv3 --pmc counter_collection.csv rows per (kernel, counter).
usage: pmc_summary.py <dir-or-csv> [<dir-or-csv> ...] > summary.csv   (kernels whose name contains 'czc::' only)"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for arg in sys.argv[1:]:
    files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "czc::" not in k:
                    continue
                k = k.split("(")[0].replace("void ", "")
                a = acc[(k, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"]); a[1] += 1
kernels = sorted({k for k, _ in acc})
counters = sorted({c for _, c in acc})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches"] + counters)
for k in kernels:
    n = max(acc[(k, c)][1] for c in counters if (k, c) in acc)
    w.writerow([k, n] + [f"{acc[(k, c)][0] / acc[(k, c)][1]:.1f}" if (k, c) in acc and acc[(k, c)][1] else "" for c in counters])
