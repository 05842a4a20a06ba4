"""Dev tool: random GEMM problems through the C ABI against an fp32 matmul -- layouts, epilogues, ragged M / N / K,
padded leading dimensions, split-K, and every tile / K-loop setting czc_gemm_set_tile offers (256x256 persistent or one
workgroup per tile, 128x128, ping-pong on / off).  The parametrised tests pin chosen shapes; this walks the space between
them.  Prints the failures and a summary, exit code 1 on any failure.   python tools/fuzz_gemm.py [cases] [seed]
tests/test_gpu_fullsize.py runs a seeded 40-case slice of it (`run(40, seed)`) inside pytest -m gpu."""
import math
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from czech_contriever_amd import lib as L, ops



def run(n_cases=300, seed=0, verbose=True):
    """-> (cases checked, [(tag, why)])"""
    lib = L.load()
    dev = torch.device("cuda:0")
    BF16 = torch.bfloat16
    rng = random.Random(seed)
    g = torch.Generator(device="cpu").manual_seed(rng.randrange(1 << 30))


    def rnd(*shape, scale=0.5):
        return (torch.randn(*shape, generator=g) * scale).to(dev)


    def pick_dim(mult):
        kind = rng.random()
        if kind < 0.35:
            v = rng.randrange(1, 40) * mult                       # small
        elif kind < 0.7:
            v = rng.choice([64, 128, 192, 256, 320, 512, 768, 1024]) + rng.choice([0, 0, mult, -mult, 8 * mult])
        else:
            v = rng.randrange(1, 260) * mult                      # up to ~2000
        return max(mult, v)


    def raw_gemm(A, Bm, al, bl, M, N, K, epi, bias, aux, splitk):
        """ops.gemm without its contiguity check: the operands here may have leading dimensions longer than their rows"""
        F32 = torch.float32
        if epi == "f32":
            out = torch.empty(splitk, M, N, dtype=F32, device=dev)
        else:
            out = torch.empty(M, N, dtype=BF16, device=dev)
        out2 = torch.empty(M, N, dtype=BF16, device=dev) if epi == "bias_gelu" else None
        p = lambda t: None if t is None else t.data_ptr()
        ld = lambda t: t.stride(0) if t.shape[0] > 1 else t.shape[1]      # (a one-row tensor reports an arbitrary stride)
        L.check(lib.czc_gemm_bf16(al, bl, L.EPI[epi], M, N, K, p(A), ld(A), p(Bm), ld(Bm), p(out), N, p(bias), p(aux),
                                  p(out2), splitk, L.stream_ptr()))
        return (out, out2) if epi == "bias_gelu" else out


    def padded(t, pad):
        """the same values with a leading dimension `pad` elements longer than the row"""
        if pad == 0:
            return t
        buf = torch.zeros(t.shape[0], t.shape[1] + pad, dtype=t.dtype, device=t.device)
        buf[:, :t.shape[1]] = t
        return buf[:, :t.shape[1]]


    fails, done = [], 0
    for case in range(n_cases):
        al, bl = rng.choice([(0, 0), (0, 0), (0, 1), (1, 1)])
        epis = ["store", "bias", "add", "dgelu", "f32"] + (["bias_gelu", "bias_gelu_only"] if (al, bl) == (0, 0) else [])
        epi = rng.choice(epis)
        M = pick_dim(8 if al == 1 else 1)
        N = pick_dim(8)
        K = pick_dim(8 if (al == 0 or bl == 0) else 1)
        if M * N > 6_000_000:
            M = max(8, (6_000_000 // N) // 8 * 8)
        splitk = rng.choice([1, 1, 2, 3, 7]) if epi == "f32" else 1
        cfg = rng.choice([0, 1, 2, 17, 18, 33, 65, 65 + 16, 1 + 1024, 1 + 2048, 1 + 512])
        a32, b32 = rnd(M, K), rnd(N, K)
        a, b = a32.to(BF16), b32.to(BF16)
        A = a if al == 0 else a.t().contiguous()
        Bm = b if bl == 0 else b.t().contiguous()
        A, Bm = padded(A, rng.choice([0, 0, 8, 24])), padded(Bm, rng.choice([0, 0, 8, 40]))
        bias = rnd(N, scale=1.0) if epi in ("bias", "bias_gelu", "bias_gelu_only") else None
        aux = rnd(M, N, scale=1.0).to(BF16) if epi in ("add", "dgelu") else None
        acc = a.float() @ b.float().t()
        tag = f"#{case} al{al} bl{bl} {epi} {M}x{N}x{K} splitk{splitk} cfg{cfg} lda{A.stride(0)} ldb{Bm.stride(0)}"
        L.check(lib.czc_gemm_set_tile(cfg))
        try:
            out = raw_gemm(A, Bm, al, bl, M, N, K, epi, bias, aux, splitk)
            torch.cuda.synchronize()
        except RuntimeError as e:
            L.check(lib.czc_gemm_set_tile(0))
            fails.append((tag, f"raised {e}"))
            continue
        finally:
            L.check(lib.czc_gemm_set_tile(0))
        tol_acc = 0.02 * math.sqrt(K) * 0.25 + 1e-2
        checks = []
        if epi == "store":
            checks.append((out, acc, tol_acc, 8e-3))
        elif epi == "bias":
            checks.append((out, acc + bias, tol_acc + 2e-2, 8e-3))
        elif epi == "f32":
            checks.append((out.sum(0), acc, 2e-3 * math.sqrt(K) + 1e-3, 1e-4))
        elif epi == "add":
            checks.append((out, acc.to(BF16).float() + aux.float(), tol_acc + 3e-2, 8e-3))
        elif epi == "dgelu":
            checks.append((out, acc.to(BF16).float() * aux.float(), (tol_acc + 3e-2) * 4, 1e-2))
        elif epi == "bias_gelu":
            u = (acc + bias).to(BF16).float().requires_grad_(True)
            h = torch.nn.functional.gelu(u)
            h.sum().backward()
            # u itself is off by the accumulation-order rounding: compare through a tolerance on u (|gelu'| <= 1.13, |gelu''| <= 0.8)
            checks.append((out[1], h.detach(), tol_acc + 2e-2, 8e-3))
            checks.append((out[0], u.grad, tol_acc + 2e-2, 8e-3))
        elif epi == "bias_gelu_only":
            checks.append((out, torch.nn.functional.gelu(acc + bias), tol_acc + 2e-2, 8e-3))
        for got, ref, atol, rtol in checks:
            got = got.float()
            if got.shape != ref.shape or not torch.isfinite(got).all():
                fails.append((tag, f"shape {tuple(got.shape)} vs {tuple(ref.shape)} or non-finite"))
                break
            err = (got - ref).abs() - rtol * ref.abs()
            if float(err.max()) > atol:
                i = int(err.argmax())
                fails.append((tag, f"max excess {float(err.max()):.4f} > {atol:.4f} at flat {i} (row {i // ref.shape[1]}, col {i % ref.shape[1]})"))
                break
        done += 1
        if case % 50 == 49:
            if verbose:
                print(f"{case + 1} cases, {len(fails)} failures", flush=True)
    return done, fails


if __name__ == "__main__":
    done, fails = run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    for t, why in fails:
        print("FAIL", t, why)
    print(f"fuzz_gemm: {done} cases checked, {len(fails)} failures")
    sys.exit(1 if fails else 0)
