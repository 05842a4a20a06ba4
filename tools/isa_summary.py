#!/usr/bin/env python3
"""Dev helper: summarise one kernel's ISA (instruction counts, waits, barriers, branches)."""
import sys, re, collections
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
body = s[i:j].splitlines()
print(len(body), "lines")
c = collections.Counter()
for l in body:
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if m: c[m.group(1)] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:40]:
    print(f"  {k:40s} {v}")
if len(sys.argv) > 3:
    for n, l in enumerate(body):
        if re.search(sys.argv[3], l): print(n, l.strip())
