"""Dev tool: in-kernel s_memtime stamps of the streaming attention backward (needs the diagnostic build:
hipcc ... -DATT_STAMPS=<wave> attention.hip, linked into the libczc.so this process loads)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
B, S, nh = 256, 256, 12; H = nh * 64; p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
BF, F32 = torch.bfloat16, torch.float32
qkv = (torch.randn(B * S, 3 * H, device=dev) * 0.5).to(BF)
dctx = (torch.randn(B * S, H, device=dev) * 0.5).to(BF)
mask = torch.ones(B, S, dtype=torch.bool, device=dev)
from czech_contriever_amd import ops
ctx, lse = ops.attention_forward(qkv, mask, B, S, nh, p, 5, 3)
dqkv = torch.empty_like(qkv)
delta = torch.zeros(B, nh, S, dtype=F32, device=dev)
db = torch.zeros(3 * H, device=dev)
part = torch.empty(lib.czc_attention_bias_scratch_floats(B, S, nh), dtype=F32, device=dev)
for _ in range(5):
    L.check(lib.czc_attention_backward(L.ptr(qkv, BF), L.ptr(mask), L.ptr(ctx, BF), L.ptr(dctx, BF), L.ptr(lse, F32), B, S, nh, p, 5, 3,
                                       L.ptr(delta), L.ptr(dqkv), L.ptr(db), L.ptr(part), L.stream_ptr()))
torch.cuda.synchronize()
st = delta.view(torch.int64).view(-1)[: 256 * 128].view(256, 128).cpu()
def med(i, j): return statistics.median((st[w, j] - st[w, i]).item() for w in range(256) if st[w, i] and st[w, j])
names = ["dma issue", "dq block+stores", "S dP softmax dV dK", "delta", "vmcnt", "barrier"]
print("per step (cycles, median over workgroups):")
for s_ in range(8):
    base = 6 * s_
    row = [med(base + k, base + k + 1) for k in range(6)]
    print(f"  step {s_}: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, row)) + f"   total {sum(row):6.0f}")
print("seam: to vmcnt", med(48, 60), " vmcnt+bar", med(60, 61), " dqstore/pack/take+bar", med(61, 62), " image write+bar", med(62, 63),
      " row stores+psum+bar", med(63, 64), " dbias", med(64, 65))
print("item total", med(0, 65), " clock GHz", statistics.median((st[w, 65] - st[w, 0]).item() / max(1, (st[w, 67] - st[w, 66]).item()) * 0.1 for w in range(256) if st[w, 65]))
