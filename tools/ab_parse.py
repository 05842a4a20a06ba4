"""Dev helper: summarise gpurun_out/ab.txt (alternating '== old' / '== new' runs of tools/bench_gemm.py)."""
import re, collections, sys
cur = None; d = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ab.txt"):
    if l.startswith("=="): cur = l.split()[1]; continue
    m = re.match(r"(.+?)\s*\| \w+:\s+([\d.]+) us", l)
    if m: d[m.group(1).strip()][cur].append(float(m.group(2)))
names = sorted({k for v in d.values() for k in v})
tot = {n: 0.0 for n in names}
for k, v in d.items():
    avg = {n: sum(v[n]) / len(v[n]) for n in names}
    for n in names: tot[n] += avg[n]
    print(f"{k:34s} " + "  ".join(f"{n} {avg[n]:7.1f}" for n in names) + f"  {100 * (avg[names[0]] / avg[names[-1]] - 1):+.1f}%")
print({n: round(t, 1) for n, t in tot.items()}, f"{names[0]} vs {names[-1]}: {100 * (tot[names[0]] / tot[names[-1]] - 1):+.2f}%")
