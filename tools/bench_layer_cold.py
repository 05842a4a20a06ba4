"""Dev tool: every GEMM shape of one encoder layer (forward + backward, B 256 x S 256) on COLD operands -- inputs and outputs
rotate over buffers far beyond the 256 MiB Infinity Cache with 1 GiB of unrelated traffic between launches, which is what a
layer sees inside the step (tools/bench_gemm_cold.py explains why back-to-back loops mislead).  Prints one line per shape:
median us and TFLOP/s.  Run under CZC_LIB_VARIANT=<name> for an A/B of compile-time switches on one device.
python tools/bench_layer_cold.py [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from czech_contriever_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
bf = torch.bfloat16
M, H, I = 65536, 768, 3072
NB = 3
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if os.environ.get("CZC_GEMM_CFG"):   # czc_gemm_set_tile: 2 = 128x128 tiles at two workgroups per CU, +16 one workgroup per tile, ...
    L.check(lib.czc_gemm_set_tile(int(os.environ["CZC_GEMM_CFG"])))
def r(*s): return [(torch.randn(*s, device=dev) * 0.5).to(bf) for _ in range(NB)]
def e(*s): return [torch.empty(*s, dtype=bf, device=dev) for _ in range(NB)]
X, XI = r(M, H), r(M, I)
W = {n: r(*s) for n, s in (("qkv", (3 * H, H)), ("o", (H, H)), ("w1", (I, H)), ("w2", (H, I)), ("w1t", (H, I)), ("w2t", (I, H)), ("qkvt", (H, 3 * H)))}
X3 = r(M, 3 * H)
O_H, O_3H, O_I, O_I2 = e(M, H), e(M, 3 * H), e(M, I), e(M, I)
bias = {n: torch.randn(n, device=dev) for n in (H, 3 * H, I)}
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
p = lambda t: None if t is None else t.data_ptr()
def gemm(epi, Mx, N, K, A, B, C, b=None, aux=None, C2=None):
    L.check(lib.czc_gemm_bf16(0, 0, L.EPI[epi], Mx, N, K, p(A), K, p(B), K, p(C), N, p(b), p(aux), p(C2), 1, L.stream_ptr()))
cases = [
    ("QKV fwd      bias      2304x768 ", 2304, 768, lambda i: gemm("bias", M, 3 * H, H, X[i], W["qkv"][i], O_3H[i], bias[3 * H])),
    ("Wo fwd       bias       768x768 ", 768, 768, lambda i: gemm("bias", M, H, H, X[i], W["o"][i], O_H[i], bias[H])),
    ("FFN-up q     bias_gelu 3072x768 ", 3072, 768, lambda i: gemm("bias_gelu", M, I, H, X[i], W["w1"][i], O_I[i], bias[I], None, O_I2[i])),
    ("FFN-up k     gelu_only 3072x768 ", 3072, 768, lambda i: gemm("bias_gelu_only", M, I, H, X[i], W["w1"][i], O_I[i], bias[I])),
    ("FFN-down fwd bias       768x3072", 768, 3072, lambda i: gemm("bias", M, H, I, XI[i], W["w2"][i], O_H[i], bias[H])),
    ("dGELU bwd    dgelu     3072x768 ", 3072, 768, lambda i: gemm("dgelu", M, I, H, X[i], W["w2t"][i], O_I[i], None, XI[i])),
    ("dX(W1) bwd   add        768x3072", 768, 3072, lambda i: gemm("add", M, H, I, XI[i], W["w1t"][i], O_H[i], None, X[i])),
    ("dX(Wo) bwd   store      768x768 ", 768, 768, lambda i: gemm("store", M, H, H, X[i], W["o"][i], O_H[i])),
    ("dX(QKV) bwd  add        768x2304", 768, 2304, lambda i: gemm("add", M, H, 3 * H, X3[i], W["qkvt"][i], O_H[i], None, X[i])),
]
res = {c[0]: [] for c in cases}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rd in range(rounds + 1):
    for name, N, K, fn in cases:
        ts = []
        for i in range(6):
            junk.add_(1)
            e0.record(); fn(i % NB); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        if rd:
            res[name].append(statistics.median(ts))
tot = 0.0
for name, N, K, fn in cases:
    t = statistics.median(res[name]) * 1e3
    tot += t
    print(f"{name} | cold: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF/s", flush=True)
print(f"sum {tot:.1f} us")
