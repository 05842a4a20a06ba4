"""Dev tool: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this step's shapes, for headroom only.
Nothing in the product path calls it."""
import torch, statistics
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts)
M = 65536
for name, (m, n, k, ta, tb) in {
    "qkv fwd   x[M,768] w[2304,768]^T": (M, 2304, 768, False, True),
    "ffn-up    x[M,768] w[3072,768]^T": (M, 3072, 768, False, True),
    "ffn-down  x[M,3072] w[768,3072]^T": (M, 768, 3072, False, True),
    "dX ffn-down  dy[M,768] w[768,3072]": (M, 3072, 768, False, False),
    "dX ffn-up    dy[M,3072] w[3072,768]": (M, 768, 3072, False, False),
    "dW ffn-up    dy[M,3072]^T x[M,768]": (3072, 768, M, True, False),
    "dW ffn-down  dy[M,768]^T x[M,3072]": (768, 3072, M, True, False),
}.items():
    a = torch.randn((k, m) if ta else (m, k), device=dev).to(torch.bfloat16)
    b = torch.randn((n, k) if tb else (k, n), device=dev).to(torch.bfloat16)
    A = a.t() if ta else a
    Bm = b.t() if tb else b
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: torch.matmul(A, Bm, out=out))
    print(f"{name:42s} {ms*1e3:8.1f} us  {2*m*n*k/ms/1e9:7.0f} TFLOP/s")
