"""Dev tool: in-kernel s_memtime stamps of the resident attention forward (diagnostic build: python -m czech_contriever_amd.build
--variant fstamp -DATTF_STAMPS=0, run with CZC_LIB_VARIANT=fstamp).  Where a workgroup's 18 us go: requests issued, data
arrived, images written + barrier, the eight key tiles, the stores; and how the workgroups of a launch are spread in time."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
B, S, nh = 256, 256, 12; H = nh * 64; p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).to(torch.bfloat16)
mask = torch.ones(B, S, dtype=torch.bool, device=dev)
for _ in range(4):
    ctx, lse = ops.attention_forward(qkv, mask, B, S, nh, p, 5, 3)
torch.cuda.synchronize()
st = lse.view(torch.int64).view(B * nh, S // 2)[:, :12].cpu()
# co-residency: group by (xcc, se, sh, cu); s_memtime is comparable inside one XCD only
import collections
cus = collections.defaultdict(list)
for w in range(B * nh):
    hw, xcc = st[w, 10].item(), st[w, 11].item() & 0xf
    cus[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)].append((st[w, 0].item(), st[w, 3].item(), st[w, 4].item(), st[w, 6].item(), w))
print("CUs seen:", len(cus), " workgroups per CU:", sorted(collections.Counter(len(v) for v in cus.values()).items()))
key = sorted(cus)[5]
print("timeline of one CU", key, "(cycles from its first start: start, images ready, tiles done, end, dispatch index)")
base = min(v[0] for v in cus[key])
for v in sorted(cus[key]):
    print("   ", [x - base for x in v[:4]], v[4])
# overlap statistic: for each CU, fraction of a workgroup's load phase (start..images ready) during which the CU's other resident workgroup is in its tile phase
ov = []
for v in cus.values():
    v = sorted(v)
    for i, a_ in enumerate(v):
        L0, L1 = a_[0], a_[1]
        best = 0
        for j, b_ in enumerate(v):
            if i == j: continue
            best = max(best, max(0, min(L1, b_[2]) - max(L0, b_[1])))
        ov.append(best / max(1, L1 - L0))
print(f"load phase covered by the CU's other workgroup's tile phase: median {statistics.median(ov):.2f}")
names = ["requests issued", "data arrived", "images + barrier", "8 key tiles", "epilogue to stores issued", "stores acknowledged"]
d = [[(st[w, k + 1] - st[w, k]).item() for w in range(B * nh)] for k in range(6)]
clk = statistics.median((st[w, 6] - st[w, 0]).item() / max(1, (st[w, 9] - st[w, 8]).item()) * 0.1 for w in range(B * nh))
print(f"clock {clk:.2f} GHz; cycles per workgroup (wave {os.environ.get('WAVE', '0')}): median / p10 / p90")
for n, v in zip(names, d):
    v = sorted(v)
    print(f"  {n:28s} {statistics.median(v):8.0f} {v[len(v)//10]:8.0f} {v[9*len(v)//10]:8.0f}")
tot = sorted((st[w, 6] - st[w, 0]).item() for w in range(B * nh))
print(f"  {'workgroup total':28s} {statistics.median(tot):8.0f} {tot[len(tot)//10]:8.0f} {tot[9*len(tot)//10]:8.0f}   = {statistics.median(tot)/clk/1e3:.1f} us")
