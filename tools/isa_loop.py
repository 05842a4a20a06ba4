#!/usr/bin/env python3
"""Dev helper: print the hottest loop (most MFMAs) of one kernel from a hipcc -S listing, MFMA runs collapsed."""
import sys, re
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
body = [l for l in s[i:j].splitlines() if not re.match(r"\s*;", l)]
labels = {}
for n, l in enumerate(body):
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = n
best = None
for n, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        cnt = sum(1 for x in body[labels[m.group(1)]:n] if "v_mfma" in x)
        if best is None or cnt > best[2]: best = (labels[m.group(1)], n, cnt)
a, b, cnt = best
k = 0
for l in body[a - int(sys.argv[3]) if len(sys.argv) > 3 else a:b + 1]:
    if "v_mfma" in l: k += 1; continue
    if k: print(f"      [{k} mfma]"); k = 0
    print(l[:100])
