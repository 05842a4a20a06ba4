"""GPU parity tests for every kernel behind the C ABI (libczc.so), each against a plain fp32
reference of the same op computed from the SAME bf16-rounded inputs.  Run with `-m gpu` on MI355X.
Tolerances are written next to each check: bf16 outputs carry 2^-9 relative rounding; fp32
accumulations over K terms differ from torch's by summation order only."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from czech_contriever_amd import ops as _ops
    from czech_contriever_amd import lib
    lib.load()
    return _ops


def dev():
    return torch.device("cuda:0")


def rnd(*shape, scale=1.0, seed=0, dtype=F32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def report(name, got, ref, atol, rtol):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any() or not torch.isfinite(got).all():
        idx = torch.nonzero(bad)[:5].tolist()
        worst = err.argmax()
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.numel()} out of tolerance; max err {float(err.max()):.4e} at flat {int(worst)} "
            f"(got {float(got.flatten()[worst]):.6f} ref {float(ref.flatten()[worst]):.6f}); first bad idx {idx}; "
            f"finite={bool(torch.isfinite(got).all())}")


# ------------------------------------------------------------------------------------------------
# flat elementwise kernels
# ------------------------------------------------------------------------------------------------
def test_ema_exact(ops):
    n = 1_000_003
    pk, pq = rnd(n, seed=1), rnd(n, seed=2)
    nb = 4096
    pkb = torch.zeros(nb, dtype=BF16, device=dev())
    m = 0.9995
    ref = pk * m + pq * (1.0 - m)      # torch on the GPU: two rounded products, one add
    ref_cpu = pk.cpu() * m + pq.cpu() * (1.0 - m)
    ops.ema_update(pk, pq, m, pkb, nb)
    assert torch.equal(pk.cpu(), ref_cpu), (pk.cpu() - ref_cpu).abs().max()
    assert torch.equal(pk, ref)
    assert torch.equal(pkb, ref[:nb].to(BF16))


def test_adamw_matches_torch(ops):
    n = 100_003
    p0 = rnd(n, seed=3)
    p = p0.clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    pb = torch.zeros(1024, dtype=BF16, device=dev())
    tp = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([tp], lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01)
    for step in range(1, 6):
        g = rnd(n, seed=10 + step, scale=0.5)
        tp.grad = g.clone()
        opt.step()
        ops.adamw_step(p, g, m, v, 1e-2, 0.9, 0.98, 1e-6, 0.01, step, 1.0, None, pb, 1024)
        report(f"adamw step {step}", p, tp.detach(), 1e-6, 1e-5)
    assert torch.equal(pb, p[:1024].to(BF16))


def test_grad_norm_clip_and_scale(ops):
    n = 300_001
    g = rnd(n, seed=5)
    partials = torch.empty(4096, dtype=F32, device=dev())
    ss = torch.empty(1, dtype=F32, device=dev())
    coef = torch.empty(1, dtype=F32, device=dev())
    ops.sumsq(g, partials, ss)
    ref = float((g.double() ** 2).sum())
    assert abs(float(ss) - ref) <= 1e-5 * ref
    ops.clip_coef(ss, 1.0, 2.0, coef)
    assert abs(float(coef) - min(1.0, 2.0 / (math.sqrt(ref) + 1e-6))) < 1e-6
    # the coefficient is applied inside AdamW: compare with torch clipping then stepping
    p = rnd(n, seed=6)
    tp = torch.nn.Parameter(p.clone())
    tp.grad = g.clone()
    torch.nn.utils.clip_grad_norm_([tp], 2.0)
    opt = torch.optim.AdamW([tp], lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01)
    opt.step()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    ops.adamw_step(p, g, m, v, 1e-3, 0.9, 0.98, 1e-6, 0.01, 1, 1.0, coef)
    report("clipped adamw", p, tp.detach(), 1e-6, 1e-5)
    g2 = g.clone()
    ops.clamp_(g2, -0.5, 0.25)
    assert torch.equal(g2, g.clamp(-0.5, 0.25))
    dst = torch.empty(n, dtype=BF16, device=dev())
    ops.cast_bf16(g, dst)
    assert torch.equal(dst, g.to(BF16))


def test_enqueue_ring_buffer(ops):
    C, K, n = 64, 512, 64
    qf = torch.nn.functional.normalize(rnd(C, K, seed=7), dim=0).contiguous()
    qb = torch.empty(K, C, dtype=BF16, device=dev())
    ops.queue_import(qf, qb)
    assert torch.equal(qb, qf.t().contiguous().to(BF16))
    ptr = torch.zeros(1, dtype=torch.int64, device=dev())
    ref = qf.clone()
    p = 0
    for step in range(10):  # wraps after 8
        keys = rnd(n, C, seed=100 + step)
        ops.enqueue(keys, qb, qf, ptr)
        ref[:, p:p + n] = keys.t()
        p = (p + n) % K
        assert int(ptr) == p
    assert torch.equal(qf, ref)
    assert torch.equal(qb, ref.t().contiguous().to(BF16))
    with pytest.raises(RuntimeError):
        ops.enqueue(rnd(48, C), qb, qf, ptr)  # 512 % 48 != 0: the reference asserts (src/moco.py:279-281)


def test_dropout_mask_rate_and_numpy_restatement(ops):
    from tests.dropout_ref import keep_mask
    n = 1 << 20
    keep = ops.dropout_mask(1234, 17, 0.1, n, dev()).cpu().numpy()
    assert abs(keep.mean() - 0.9) < 2e-3
    assert np.array_equal(keep, keep_mask(1234, 17, 0.1, n))


# ------------------------------------------------------------------------------------------------
# row kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,H", [(37, 128), (1000, 768), (5, 1024)])
def test_residual_ln_fwd_bwd(ops, rows, H):
    y, resid = rnd(rows, H, seed=1, dtype=BF16), rnd(rows, H, seed=2, dtype=BF16)
    gamma = 1.0 + 0.1 * rnd(H, seed=3)
    beta = 0.1 * rnd(H, seed=4)
    eps = 1e-12
    x, stats = ops.residual_ln_forward(y, resid, gamma, beta, eps)
    z = (y.float() + resid.float()).requires_grad_(True)
    gp, bp = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(z, (H,), gp, bp, eps)
    report("ln fwd", x, ref, 2e-2, 8e-3)  # bf16 output
    report("ln mean", stats[:, 0], z.mean(-1), 1e-5, 1e-5)
    dx = rnd(rows, H, seed=5, dtype=BF16)
    dx2 = rnd(rows, H, seed=6, dtype=BF16)
    ref.backward(dx.float() + dx2.float())
    dg, db, dbias = (torch.zeros(H, device=dev()) for _ in range(3))
    dz, dy = ops.residual_ln_backward(dx, dx2, y, resid, stats, gamma, dg, db, dbias)
    report("ln dz", dz, z.grad, 3e-2, 1e-2)
    report("ln dgamma", dg, gp.grad, 2e-3 * math.sqrt(rows), 1e-3)
    report("ln dbeta", db, bp.grad, 2e-3 * math.sqrt(rows), 1e-3)
    report("ln dbias", dbias, z.grad.sum(0), 3e-3 * math.sqrt(rows), 1e-3)


def test_residual_ln_dropout_consistency(ops):
    from tests.dropout_ref import keep_mask
    rows, H, p, seed, site = 64, 768, 0.1, 99, 33
    y, resid = rnd(rows, H, seed=1, dtype=BF16), rnd(rows, H, seed=2, dtype=BF16)
    gamma, beta = 1.0 + 0.1 * rnd(H, seed=3), 0.1 * rnd(H, seed=4)
    keep = torch.from_numpy(keep_mask(seed, site, p, rows * H).reshape(rows, H)).to(dev()).float()
    x, stats = ops.residual_ln_forward(y, resid, gamma, beta, 1e-12, p, seed, site)
    yv = y.float().requires_grad_(True)
    z = yv * keep / (1 - p) + resid.float()
    ref = torch.nn.functional.layer_norm(z, (H,), gamma, beta, 1e-12)
    report("ln+dropout fwd", x, ref, 2e-2, 8e-3)
    dx = rnd(rows, H, seed=5, dtype=BF16)
    ref.backward(dx.float())
    dg, db, dbias = (torch.zeros(H, device=dev()) for _ in range(3))
    dz, dy = ops.residual_ln_backward(dx, None, y, resid, stats, gamma, dg, db, dbias, p, seed, site)
    report("ln+dropout dy", dy, yv.grad, 3e-2, 1e-2)


def test_embed_ln_fwd_bwd(ops):
    B, S, H, V = 3, 40, 768, 1000
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, V, (B, S), generator=g).to(dev())
    ids[0, :5] = 7  # repeated ids: scatter-add collisions
    word, pos, typ = rnd(V, H, seed=1, scale=0.05), rnd(64, H, seed=2, scale=0.05), rnd(2, H, seed=3, scale=0.05)
    gamma, beta = 1.0 + 0.1 * rnd(H, seed=4), 0.1 * rnd(H, seed=5)
    x, stats = ops.embed_ln_forward(ids, word, pos, typ[0].contiguous(), gamma, beta, 1e-12)
    wp, pp, tp = (t.clone().requires_grad_(True) for t in (word, pos, typ))
    gp, bp = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = wp[ids] + tp[0][None, None] + pp[:S][None]
    ref = torch.nn.functional.layer_norm(z, (H,), gp, bp, 1e-12)
    report("embed fwd", x.view(B, S, H), ref, 2e-2, 8e-3)
    dx = rnd(B * S, H, seed=6, dtype=BF16)
    ref.backward(dx.float().view(B, S, H))
    dw, dp, dt = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros(H, device=dev())
    dg, db = torch.zeros(H, device=dev()), torch.zeros(H, device=dev())
    ops.embed_ln_backward(ids, word, pos, typ[0].contiguous(), gamma, stats, dx, dw, dp, dt, dg, db)
    report("d_word", dw, wp.grad, 5e-4, 1e-3)
    report("d_pos", dp, pp.grad, 5e-4, 1e-3)
    report("d_type0", dt, tp.grad[0], 2e-3, 1e-3)
    report("d_gamma", dg, gp.grad, 2e-3, 1e-3)
    report("d_beta", db, bp.grad, 2e-3, 1e-3)


@pytest.mark.parametrize("B", [40, 300])   # 40: chunks of 2-3 sequences per unit; 300: sixteen chunks of 19 per position
def test_embed_ln_bwd_equal_id_runs_and_padding(ops, B):
    """Text-shaped ids: every sequence starts with the same id (runs of equal ids inside a unit are summed before the
    scatter-add), padded tails (id 0) carry an exactly-zero incoming gradient (their rows add nothing), a frequent id
    in the middle; and the gradients ACCUMULATE into what is already there (update_freq > 1)."""
    S, H, V = 24, 768, 500
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(5, V, (B, S), generator=g)
    ids[:, 0] = 3
    ids[:, 7] = torch.where(torch.rand(B, generator=g) < 0.6, torch.tensor(9), ids[:, 7])
    lens = torch.randint(9, S + 1, (B,), generator=g)
    valid = torch.arange(S)[None, :] < lens[:, None]
    ids = torch.where(valid, ids, torch.zeros_like(ids)).to(dev())
    valid = valid.to(dev())
    word, pos, typ = rnd(V, H, seed=1, scale=0.05), rnd(64, H, seed=2, scale=0.05), rnd(2, H, seed=3, scale=0.05)
    gamma, beta = 1.0 + 0.1 * rnd(H, seed=4), 0.1 * rnd(H, seed=5)
    x, stats = ops.embed_ln_forward(ids, word, pos, typ[0].contiguous(), gamma, beta, 1e-12)
    wp, pp, tp = (t.clone().requires_grad_(True) for t in (word, pos, typ))
    gp, bp = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(wp[ids] + tp[0][None, None] + pp[:S][None], (H,), gp, bp, 1e-12)
    dx = (rnd(B * S, H, seed=6).view(B, S, H) * valid[..., None]).to(BF16).view(B * S, H).contiguous()
    ref.backward(dx.float().view(B, S, H))
    start = 0.25
    dw, dp, dt = torch.full_like(word, start), torch.full_like(pos, start), torch.full((H,), start, device=dev())
    dg, db = torch.full((H,), start, device=dev()), torch.full((H,), start, device=dev())
    ops.embed_ln_backward(ids, word, pos, typ[0].contiguous(), gamma, stats, dx, dw, dp, dt, dg, db)
    report("d_word", dw - start, wp.grad, 2e-3, 1e-3)
    report("d_pos", dp - start, pp.grad, 2e-3, 1e-3)
    report("d_type0", dt - start, tp.grad[0], 5e-3, 1e-3)
    report("d_gamma", dg - start, gp.grad, 5e-3, 1e-3)
    report("d_beta", db - start, bp.grad, 5e-3, 1e-3)
    assert torch.equal(dw[0], torch.full((H,), start, device=dev()))     # the padding id's row was not touched


@pytest.mark.parametrize("pooling,normalize", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_pool_fwd_bwd(ops, pooling, normalize):
    B, S, H = 5, 24, 768
    x = rnd(B * S, H, seed=1, dtype=BF16)
    mask = torch.zeros(B, S, dtype=torch.bool, device=dev())
    for b, n in enumerate([24, 17, 1, 9, 24]):
        mask[b, :n] = True
    emb, raw = ops.pool_forward(x, mask, B, S, pooling, normalize)
    xv = x.float().view(B, S, H).requires_grad_(True)
    lh = xv.masked_fill(~mask[..., None], 0.0)
    ref = lh.sum(1) / mask.sum(1)[..., None] if pooling == 0 else lh[:, 0]
    if normalize:
        ref = torch.nn.functional.normalize(ref, dim=-1)
    report("pool fwd", emb, ref, 1e-5, 1e-5)
    de = rnd(B, H, seed=2)
    ref.backward(de)
    dx = ops.pool_backward(de, raw, mask, B, S, pooling, normalize)
    report("pool bwd", dx.view(B, S, H), xv.grad, 1e-3, 8e-3)


def test_colsum(ops):
    x = rnd(1000, 2304, seed=1, dtype=BF16)
    col = torch.zeros(2304, device=dev())
    ops.colsum(x, col)
    report("colsum", col, x.float().sum(0), 2e-3, 1e-4)


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
def _mk(M, N, K, al, bl, seed):
    a = rnd(M, K, seed=seed, scale=0.5, dtype=BF16)
    b = rnd(N, K, seed=seed + 1, scale=0.5, dtype=BF16)
    A = a if al == 0 else a.t().contiguous()   # [K, M]
    Bm = b if bl == 0 else b.t().contiguous()  # [K, N]
    return a, b, A, Bm


def test_gemm_identity_asymmetric(ops):
    # A = I picks out B^T exactly: catches a transposed or permuted C write
    M = N = K = 256
    eye = torch.eye(256, dtype=BF16, device=dev())
    b = (torch.arange(N, device=dev())[:, None] * 3 + torch.arange(K, device=dev())[None, :] * 0.25 - 40).to(BF16)
    for al, bl in ((0, 0), (0, 1), (1, 1)):
        A = eye
        Bm = b if bl == 0 else b.t().contiguous()
        c = ops.gemm(A, Bm, a_layout=al, b_layout=bl, M=M, N=N, K=K)
        assert torch.equal(c, b.t().contiguous()), (al, bl, (c.float() - b.t().float()).abs().max())


@pytest.mark.parametrize("al,bl", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 768), (300, 128, 192), (1000, 2304, 768), (24, 384, 128),
                                   (768, 768, 1000), (128, 256, 65)])
def test_gemm_layouts_and_tails(ops, al, bl, M, N, K):
    if K % 8 and (al == 0 or bl == 0):
        pytest.skip("K-major operands need K % 8 == 0")
    if al == 1 and M % 8:
        pytest.skip("M-major A needs M % 8 == 0")
    a, b, A, Bm = _mk(M, N, K, al, bl, 11)
    ref = a.float() @ b.float().t()
    c = ops.gemm(A, Bm, a_layout=al, b_layout=bl, M=M, N=N, K=K)
    report(f"gemm {al}{bl} {M}x{N}x{K}", c, ref, 0.02 * math.sqrt(K) * 0.25 + 1e-2, 8e-3)


def test_gemm_epilogues(ops):
    M, N, K = 520, 768, 256
    a, b, A, Bm = _mk(M, N, K, 0, 0, 21)
    bias = rnd(N, seed=30)
    aux = rnd(M, N, seed=31, dtype=BF16)
    acc = a.float() @ b.float().t()
    report("bias", ops.gemm(A, Bm, epilogue="bias", bias=bias), acc + bias, 5e-2, 8e-3)
    # FFN-up forward: C = gelu'(u), C2 = gelu(u), u = acc + bias rounded to bf16 (what the kernel evaluates both on)
    dg, h = ops.gemm(A, Bm, epilogue="bias_gelu", bias=bias)
    u = (acc + bias).to(BF16).float().requires_grad_(True)
    ref_h = torch.nn.functional.gelu(u)
    ref_h.sum().backward()
    report("gelu(u)", h, ref_h.detach(), 1e-2, 8e-3)
    report("gelu'(u)", dg, u.grad, 1e-2, 8e-3)
    # the one-output form (key encoder, inference) evaluates gelu on the fp32 sum, as the reference does
    report("gelu only", ops.gemm(A, Bm, epilogue="bias_gelu_only", bias=bias), torch.nn.functional.gelu(acc + bias), 1e-2, 8e-3)
    # its backward: C = acc * aux, aux = the stored derivative
    report("dgelu", ops.gemm(A, Bm, epilogue="dgelu", aux=aux), acc.to(BF16).float() * aux.float(), 5e-2, 1e-2)
    report("add", ops.gemm(A, Bm, epilogue="add", aux=aux), acc.to(BF16).float() + aux.float(), 5e-2, 8e-3)


@pytest.mark.parametrize("al,bl", [(0, 0), (0, 1)])
def test_gemm_workgroups_walking_over_tiles(ops, al, bl):
    """More output tiles (65 x 5 = 325) than resident workgroups (256): every workgroup of the persistent launch takes a
    second tile whose first K-tile it requested during the first tile's epilogue.  Must equal the one-workgroup-
    per-tile launch bit for bit, and the fp32 reference within the bf16 bar, for every bf16 epilogue."""
    from czech_contriever_amd import lib as L
    M, N, K = 16600, 1280, 192
    a, b, A, Bm = _mk(M, N, K, al, bl, 77)
    bias = rnd(N, seed=78)
    aux = rnd(M, N, seed=79, dtype=BF16)
    acc = a.float() @ b.float().t()
    cases = [("store", {}), ("bias", dict(bias=bias)), ("add", dict(aux=aux)), ("dgelu", dict(aux=aux))]
    if al == 0 and bl == 0:
        cases.append(("bias_gelu", dict(bias=bias)))
        cases.append(("bias_gelu_only", dict(bias=bias)))
    lib = L.load()
    for epi, kw in cases:
        outs = []
        for cfg in (1, 17, 33, 65 + 16):   # 256x256 tiles: persistent, one workgroup per tile, ping-pong K-loop, plain per-tile
            L.check(lib.czc_gemm_set_tile(cfg))
            try:
                o = ops.gemm(A, Bm, a_layout=al, b_layout=bl, M=M, N=N, K=K, epilogue=epi, **kw)
            finally:
                L.check(lib.czc_gemm_set_tile(0))
            outs.append(o[1] if isinstance(o, tuple) else o)
        for o in outs[1:]:
            assert torch.equal(outs[0], o), (epi, (outs[0].float() - o.float()).abs().max())
        if epi == "store":
            report("walk store", outs[0], acc, 0.02 * math.sqrt(K) * 0.25 + 1e-2, 8e-3)
        elif epi == "bias":
            report("walk bias", outs[0], acc + bias, 5e-2, 8e-3)
        elif epi == "add":
            report("walk add", outs[0], acc.to(BF16).float() + aux.float(), 5e-2, 8e-3)


def test_gemm_operands_beyond_two_gib(ops):
    """An operand that spans 2 GiB and more leaves the scalar-base / 32-bit-offset addressing of the DMA pieces (`fast_addr`,
    gemm_bf16.hip) for the per-lane 64-bit pointers: the ping-pong K-loop's slow staging, the plain loop's, and the
    transposing-read kernel's, each against fp32 rows of the reference; and a bf16 output a buffer descriptor cannot address
    (4 GiB) is refused before any launch."""
    from czech_contriever_amd import lib as L
    M, N, K = 70016, 256, 16384                       # A: 70016 x 16384 bf16 = 2.29 GB
    g = torch.Generator(device=dev()).manual_seed(91)  # (drawn on the device: 1.1 G values)
    a = (torch.randn(M, K, device=dev(), generator=g) * 0.25).to(BF16)
    b = (torch.randn(N, K, device=dev(), generator=g) * 0.25).to(BF16)
    assert a.numel() * 2 > (1 << 31)
    rows = torch.tensor([0, 1, 255, 256, 40000, 65535, 65536, M - 1], device=dev())
    ref = a[rows].float() @ b.float().t()
    lib = L.load()
    for cfg in (0, 65):                               # ping-pong (K >= 1536) and the plain K-loop
        L.check(lib.czc_gemm_set_tile(cfg))
        try:
            c = ops.gemm(a, b)
        finally:
            L.check(lib.czc_gemm_set_tile(0))
        report(f"gemm beyond 2 GiB cfg {cfg}", c[rows], ref, 0.02 * math.sqrt(K) * 0.0625 + 2e-2, 8e-3)
    at = a.t().contiguous()                           # [K, M]: the M-major operand of the weight-gradient products
    del a
    bt = b.t().contiguous()
    slabs = ops.gemm(at, bt, a_layout=1, b_layout=1, epilogue="f32", M=M, N=N, K=K, splitk=2)
    report("gemm beyond 2 GiB, transposing reads", slabs.sum(0)[rows], ref, 2e-3 * math.sqrt(K) * 0.0625 + 1e-3, 1e-4)
    # M x ldc x 2 bytes >= 4 GiB: refused on the arguments alone (the pointers are never touched)
    rc = lib.czc_gemm_bf16(0, 0, 0, 70016, 256, 64, L.ptr(bt), 64, L.ptr(bt), 64, L.ptr(c), 32768, None, None, None, 1, L.stream_ptr())
    assert rc != 0 and b"4 GiB" in lib.czc_last_error()


def test_gemm_splitk_weight_gradient(ops):
    # dW[N',K'] = dY^T X, reduction over 3000 tokens, split 7 ways, then the slab reduction
    Mtok, No, Ko = 3000, 768, 384
    dy, xx = rnd(Mtok, No, seed=41, scale=0.5, dtype=BF16), rnd(Mtok, Ko, seed=42, scale=0.5, dtype=BF16)
    slabs = ops.gemm(dy, xx, a_layout=1, b_layout=1, epilogue="f32", M=No, N=Ko, K=Mtok, splitk=7)
    dst = torch.ones(No, Ko, device=dev())
    ops.reduce_slabs(slabs, dst, True)
    ref = dy.float().t() @ xx.float() + 1.0
    report("dW", dst, ref, 2e-3 * math.sqrt(Mtok), 1e-4)
    ops.reduce_slabs(slabs, dst, False)
    report("dW overwrite", dst, ref - 1.0, 2e-3 * math.sqrt(Mtok), 1e-4)


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def _attn_ref(qkv, mask, B, S, nh, keep=None, p=0.0):
    H = nh * 64
    x = qkv.float().view(B, S, 3, nh, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))  # [B,nh,S,64]
    s = (q @ k.transpose(-1, -2)) * 0.125
    s = s.masked_fill(~mask[:, None, None, :], torch.finfo(torch.float32).min)
    pr = torch.softmax(s, dim=-1)
    if keep is not None:
        pr = pr * keep / (1 - p)
    return (pr @ v).transpose(1, 2).reshape(B * S, H)


@pytest.mark.parametrize("B,S,nh,lengths", [(2, 256, 12, [256, 130]), (5, 24, 2, [24, 17, 1, 9, 24]),
                                            (3, 100, 4, [100, 64, 33]), (2, 320, 2, [320, 200]),
                                            (3, 27, 2, [27, 13, 1]), (2, 1, 1, [1, 1]), (1, 512, 1, [500]),
                                            # more (batch, head) items than CUs: the persistent forward (workgroups walk over
                                            # items, the next item's K / V arrive by LDS DMA meanwhile); 288, 300 and 1030 items
                                            (24, 256, 12, [256, 130, 27, 1] * 6), (30, 100, 10, [100, 64, 33] * 10),
                                            (103, 27, 10, [27, 13, 1, 20] * 25 + [27, 5, 9])])
def test_attention_fwd_bwd(ops, B, S, nh, lengths):
    H = nh * 64
    qkv = rnd(B * S, 3 * H, seed=1, scale=1.0, dtype=BF16)
    mask = torch.zeros(B, S, dtype=torch.bool, device=dev())
    for b, n in enumerate(lengths):
        mask[b, :n] = True
    ctx, lse = ops.attention_forward(qkv, mask, B, S, nh)
    qv = qkv.float().requires_grad_(True)
    ref = _attn_ref(qv, mask, B, S, nh)
    report("attn fwd", ctx, ref, 2e-2, 1e-2)
    dctx = rnd(B * S, H, seed=2, dtype=BF16)
    ref.backward(dctx.float())
    dbias = torch.zeros(3 * H, device=dev())
    dqkv = ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, d_bias=dbias)
    report("attn d_bias (column sums of dqkv)", dbias, qv.grad.sum(0), 2e-2 * math.sqrt(B * S), 2e-2)
    g = qv.grad.view(B, S, 3, H)
    d = dqkv.float().view(B, S, 3, H)
    # gradients of K/V rows at padded positions are exactly zero in both
    report("attn dQ", d[:, :, 0], g[:, :, 0], 3e-2, 2e-2)
    report("attn dK", d[:, :, 1], g[:, :, 1], 3e-2, 2e-2)
    report("attn dV", d[:, :, 2], g[:, :, 2], 3e-2, 2e-2)


@pytest.mark.parametrize("B,S,nh,seed", [(40, 33, 8, 0), (36, 65, 8, 1), (30, 127, 10, 2), (27, 129, 10, 3), (25, 191, 11, 4),
                                         (24, 255, 12, 5), (44, 224, 6, 6), (70, 31, 4, 7), (26, 160, 10, 8)])
def test_attention_backward_streaming_shapes(ops, B, S, nh, seed):
    """The persistent backward across its seams: more (batch, head) items than CUs at sequence lengths around the 32-query
    block and 64-row group boundaries, odd lengths (clamped rows, the mask bytes' unaligned path), random padding."""
    g = torch.Generator().manual_seed(seed)
    H = nh * 64
    qkv = rnd(B * S, 3 * H, seed=seed + 10, scale=1.0, dtype=BF16)
    lengths = torch.randint(1, S + 1, (B,), generator=g).tolist()
    lengths[0] = S
    mask = torch.zeros(B, S, dtype=torch.bool, device=dev())
    for b, n in enumerate(lengths):
        mask[b, :n] = True
    ctx, lse = ops.attention_forward(qkv, mask, B, S, nh)
    qv = qkv.float().requires_grad_(True)
    ref = _attn_ref(qv, mask, B, S, nh)
    report("attn fwd", ctx, ref, 2e-2, 1e-2)
    dctx = rnd(B * S, H, seed=seed + 20, dtype=BF16)
    ref.backward(dctx.float())
    dbias = torch.zeros(3 * H, device=dev())
    dqkv = ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, d_bias=dbias)
    report("attn d_bias", dbias, qv.grad.sum(0), 2e-2 * math.sqrt(B * S), 2e-2)
    # 5e-2: ten million unit-scale elements per case, the bf16 rounding tail of one or two of them passes 3e-2 (the fused
    # kernel gives the same numbers to 2e-3: tools/dbg_attn_stream.py)
    report("attn dqkv", dqkv, qv.grad, 5e-2, 2e-2)
    again = ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh)
    assert torch.equal(dqkv, again)


@pytest.mark.parametrize("S,B,nh", [(64, 4, 12), (128, 2, 2), (252, 2, 2), (320, 2, 2), (128, 23, 12), (256, 2, 2), (256, 23, 12)])   # 64: config 1's chunk length; (multiples of 4: the draw index pitch is S rounded up to 4) <= 256: the single-pass backward; 320: the dQ / dK+dV kernel pair; 23 x 12 = 276 items: the persistent forward; 256: the backward's 256-token specialisation, with fewer and with more items than CUs
def test_attention_dropout_matches_masked_reference(ops, S, B, nh):
    from tests.dropout_ref import keep_mask
    p, seed, site = 0.1, 7, 16
    H = nh * 64
    qkv = rnd(B * S, 3 * H, seed=3, dtype=BF16)
    mask = torch.ones(B, S, dtype=torch.bool, device=dev())
    mask[1, S - 38:] = False
    keep = torch.from_numpy(keep_mask(seed, site, p, B * nh * S * S).reshape(B, nh, S, S)).to(dev()).float()
    ctx, lse = ops.attention_forward(qkv, mask, B, S, nh, p, seed, site)
    qv = qkv.float().requires_grad_(True)
    ref = _attn_ref(qv, mask, B, S, nh, keep, p)
    report("attn+dropout fwd", ctx, ref, 2e-2, 1e-2)
    dctx = rnd(B * S, H, seed=4, dtype=BF16)
    ref.backward(dctx.float())
    dqkv = ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, p, seed, site)
    report("attn+dropout dqkv", dqkv, qv.grad, 3e-2, 2e-2)
    if S <= 256:
        # the keep decisions handed from the forward to the backward as bit words: the same draws, so the same numbers
        ctx2, lse2, words = ops.attention_forward(qkv, mask, B, S, nh, p, seed, site, keep_words=True)
        assert torch.equal(ctx2, ctx) and torch.equal(lse2, lse)
        w = words.view(B, nh, 8, 256)
        for b_, h_, q_, k_ in [(0, 0, 0, 0), (B - 1, nh - 1, S - 1, S - 1), (1, 0, S // 2, 3), (0, nh - 1, 33 % S, S - 2)]:
            pos = 32 * (k_ // 32) + 8 * ((k_ % 32) // 8) + 2 * (k_ % 4) + (k_ // 4) % 2   # include/czc.h
            assert ((int(w[b_, h_, q_ // 32, pos]) >> (q_ % 32)) & 1) == int(keep[b_, h_, q_, k_]), (b_, h_, q_, k_)
        dq2 = ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, p, seed, site, keep_words=words)
        assert torch.equal(dq2, dqkv), float((dq2.float() - dqkv.float()).abs().max())


# ------------------------------------------------------------------------------------------------
# InfoNCE head
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,K,T,ls", [(8, 768, 512, 0.05, 0.0), (8, 768, 512, 0.05, 0.1), (256, 768, 4096, 0.05, 0.0),
                                        (24, 128, 1024, 1.0, 0.0)])
def test_infonce_head(ops, B, C, K, T, ls):
    q = rnd(B, C, seed=1, scale=0.4)
    k = rnd(B, C, seed=2, scale=0.4)
    k[: B // 2] = q[: B // 2] * 0.9 + 0.02
    queue = (torch.nn.functional.normalize(rnd(C, K, seed=3), dim=0) * 2.0).contiguous()
    qb = torch.empty(K, C, dtype=BF16, device=dev())
    ops.queue_import(queue, qb)
    ws = ops.infonce_workspace(B, C, K, dev())
    lse, stats = ops.infonce_forward(q, k, qb, T, ls, ws)
    # reference on the bf16-rounded operands the MFMA sees; the positive logit stays fp32
    qq = q.clone().requires_grad_(True)
    l_pos = (qq * k).sum(1, keepdim=True)
    q16 = qq + (qq.to(BF16).float() - qq).detach()  # straight-through bf16 rounding
    l_neg = q16 @ qb.float().t()
    logits = torch.cat([l_pos, l_neg], 1) / T
    loss = torch.nn.functional.cross_entropy(logits, torch.zeros(B, dtype=torch.long, device=dev()), label_smoothing=ls)
    loss.backward()
    assert abs(float(stats[0]) - float(loss)) < 2e-3 * max(1.0, abs(float(loss))), (float(stats[0]), float(loss))
    acc = 100 * (logits.argmax(-1) == 0).float().mean()
    assert abs(float(stats[1]) - float(acc)) < 1e-4
    assert abs(float(stats[2]) - float(torch.std(q, dim=0).mean())) < 1e-5
    assert abs(float(stats[3]) - float(torch.std(k, dim=0).mean())) < 1e-5
    report("lse", lse, torch.logsumexp(logits, 1), 2e-3, 1e-4)
    dq = ops.infonce_backward(q, k, qb, T, ls, lse, 1.0, ws)
    report("dq", dq, qq.grad, 2e-2 * float(qq.grad.abs().max()), 2e-2)


@pytest.mark.parametrize("ls", [0.0, 0.1])
def test_infonce_head_headline_size_vs_fp32_oracle(ops, ls):
    """The bench / BASELINE config-2 head: B 256, C 768, K 131 072, T 0.05 (512 N-tiles of LSE partials, a 67 MB bf16
    softmax-gradient matrix, a split-K dq) against the FP32 CPU oracle (oracle.infonce_loss on the fp32 q, k, queue --
    not on the bf16-rounded operands the MFMA sees): loss, lse, accuracy, dq.  Reference: src/moco.py:289-294,323-330."""
    from oracle import moco_bert_oracle as O
    B, C, K, T = 256, 768, 131072, 0.05
    q = (O.det_uniform((B, C), 901) * 0.7)
    k = (O.det_uniform((B, C), 902) * 0.7)
    k[: B // 2] = q[: B // 2] * 0.9 + 0.02          # half of the rows have an easy positive
    queue = O.init_queue(C, K, 0)                   # normalize(randn(C, K), dim=0), seed 0: the bench's queue
    qd, kd, queued = q.to(dev()), k.to(dev()), queue.to(dev())
    qb = torch.empty(K, C, dtype=BF16, device=dev())
    ops.queue_import(queued, qb)
    ws = ops.infonce_workspace(B, C, K, dev())
    lse, stats = ops.infonce_forward(qd, kd, qb, T, ls, ws)
    dq = ops.infonce_backward(qd, kd, qb, T, ls, lse, 1.0, ws)
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    qq = q.clone().requires_grad_(True)
    loss, logits = O.infonce_loss(qq, k, queue, T, ls)
    loss.backward()
    ref_lse = torch.logsumexp(logits.detach(), 1)
    # bf16 q and queue: each negative logit carries ~2^-9 |q| / T ~ 0.05 of noise; the lse of 131 073 of them much less
    assert abs(float(stats[0]) - float(loss)) < 2e-2 * max(1.0, abs(float(loss))), (float(stats[0]), float(loss))
    report("lse @ K=131072", lse.cpu(), ref_lse, 5e-2, 1e-3)
    acc = float(100 * (logits.argmax(-1) == 0).float().mean())
    assert abs(float(stats[1]) - acc) <= 100.0 / B + 1e-4, (float(stats[1]), acc)     # at most one near-tie row differs
    g, r = dq.cpu().double(), qq.grad.double()
    cos_all = float(torch.nn.functional.cosine_similarity(g.reshape(-1), r.reshape(-1), dim=0))
    assert cos_all >= 1 - 1e-3, cos_all
    # row by row, for the rows that carry gradient (an easy positive leaves ~0)
    big = r.norm(dim=1) > 1e-3 * float(r.norm(dim=1).max())
    cos_rows = torch.nn.functional.cosine_similarity(g[big], r[big], dim=1)
    assert float(cos_rows.min()) >= 1 - 5e-3, float(cos_rows.min())
    assert abs(float(g.norm()) / float(r.norm()) - 1) < 2e-2


@pytest.mark.parametrize("B,S", [(4, 256), (24, 256), (24, 224)])   # 288 items: workgroups walk over two items (seams, rows stored a step late); 224: the general kernel
def test_attention_fused_backward_is_bitwise_reproducible(ops, B, S):
    """the single-pass backward (seq <= 256) sums every dQ element in one MFMA chain over the keys and uses no
    atomics for dQ / dK / dV: two launches on the same inputs must agree bit for bit (dropout on)."""
    nh, p, seed, site = 12, 0.1, 11, 16
    H = nh * 64
    qkv = rnd(B * S, 3 * H, seed=5, dtype=BF16)
    mask = torch.ones(B, S, dtype=torch.bool, device=dev())
    mask[1, 130:] = False
    mask[3, 27:] = False
    ctx, lse = ops.attention_forward(qkv, mask, B, S, nh, p, seed, site)
    ctx2, lse2 = ops.attention_forward(qkv, mask, B, S, nh, p, seed, site)
    assert torch.equal(ctx, ctx2) and torch.equal(lse, lse2)
    dctx = rnd(B * S, H, seed=6, dtype=BF16)
    a = ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, p, seed, site)
    b = ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, p, seed, site)
    assert torch.equal(a, b), float((a.float() - b.float()).abs().max())


def test_infonce_saturated_positive_is_exactly_zero(ops):
    """README flags (un-normalised embeddings, T = 0.05): while the encoder is near its initialisation every positive
    logit towers over the negatives, softmax is exactly one-hot in fp32 and the reference's loss AND gradient are exactly
    0 (torch cross_entropy: z0 - lse(z) with lse == z0).  The head must reproduce the exact zeros: AdamW turns any
    non-zero gradient, however small, into a full lr-sized step (found at config-1 size: a fused multiply-subtract left
    the rounding residual of lpos * (1 / T) as the "gradient")."""
    from oracle import moco_bert_oracle as O
    B, C, K, T = 4, 768, 512, 0.05
    base = O.det_uniform((1, C), 5) * 0.8 + 0.3
    q = (base + 0.1 * O.det_uniform((B, C), 6)).to(dev())        # |q| ~ 22, q.k ~ 480: z0 ~ 9600 against negatives ~ +-50
    k = (base + 0.1 * O.det_uniform((B, C), 7)).to(dev())
    queue = O.init_queue(C, K, 0).to(dev())
    qb = torch.empty(K, C, dtype=BF16, device=dev())
    ops.queue_import(queue, qb)
    ws = ops.infonce_workspace(B, C, K, dev())
    for ls in (0.0,):
        lse, stats = ops.infonce_forward(q, k, qb, T, ls, ws)
        dq = ops.infonce_backward(q, k, qb, T, ls, lse, 1.0, ws)
        qq = q.cpu().clone().requires_grad_(True)
        loss, _ = O.infonce_loss(qq, k.cpu(), queue.cpu(), T, ls)
        loss.backward()
        assert float(loss) == 0.0 and float(qq.grad.abs().max()) == 0.0          # what the reference computes
        assert float(stats[0]) == 0.0, float(stats[0])
        assert float(dq.abs().max()) == 0.0, float(dq.abs().max())
        assert float(stats[1]) == 100.0


@pytest.mark.parametrize("M1,M2,N,K,sp", [(2304, 768, 768, 3000, 7), (3072, 1024, 1024, 81, 2), (256, 40, 320, 700, 3)])
def test_gemm_two_weight_gradients_in_one_launch(ops, M1, M2, N, K, sp):
    """czc_gemm_wgrad_pair: rows [0, M1) = A1^T . B1, rows [M1, M1 + M2) = A2^T . B2 in one split-K launch (the encoder's
    dW[qkv | o]) must equal the two products launched on their own, slab for slab and bit for bit."""
    from czech_contriever_amd import lib as L
    lib = L.load()
    a1, b1 = rnd(K, M1, seed=61, scale=0.5, dtype=BF16), rnd(K, N, seed=62, scale=0.5, dtype=BF16)
    a2, b2 = rnd(K, M2, seed=63, scale=0.5, dtype=BF16), rnd(K, N, seed=64, scale=0.5, dtype=BF16)
    out = torch.full((sp, M1 + M2, N), float("nan"), device=dev())
    L.check(lib.czc_gemm_wgrad_pair(M1, M2, N, K, a1.data_ptr(), M1, b1.data_ptr(), N, a2.data_ptr(), M2, b2.data_ptr(), N,
                                    out.data_ptr(), sp, L.stream_ptr()))
    r1 = ops.gemm(a1, b1, a_layout=1, b_layout=1, epilogue="f32", M=M1, N=N, K=K, splitk=sp)
    r2 = ops.gemm(a2, b2, a_layout=1, b_layout=1, epilogue="f32", M=M2, N=N, K=K, splitk=sp)
    assert torch.isfinite(out).all()
    # the split-K ranges are those of the joint launch (same K, same split count): slab for slab the same MFMA chains
    assert torch.equal(out[:, :M1], r1) and torch.equal(out[:, M1:], r2)
    ref = torch.cat([a1.float().t() @ b1.float(), a2.float().t() @ b2.float()])
    report("pair", out.sum(0), ref, 2e-3 * math.sqrt(K), 1e-4)


def test_gemm_output_store_policies_are_only_policies(ops):
    """czc_gemm_set_tile + 512 / + 1024 / + 2048 (sc1 / nt / plain bf16 output stores; the default is nt from N = 2048 on):
    a cache policy, nothing else -- every epilogue's outputs bit-identical, on a wide product with ragged edges."""
    from czech_contriever_amd import lib as L
    M, N, K = 700, 2312, 200
    a, b, A, Bm = _mk(M, N, K, 0, 0, 71)
    bias = rnd(N, seed=72)
    aux = rnd(M, N, seed=73, dtype=BF16)
    lib = L.load()
    for epi, kw in [("store", {}), ("bias", dict(bias=bias)), ("add", dict(aux=aux)), ("dgelu", dict(aux=aux)),
                    ("bias_gelu", dict(bias=bias)), ("bias_gelu_only", dict(bias=bias))]:
        outs = []
        for cfg in (0, 2048, 1024, 512):
            L.check(lib.czc_gemm_set_tile(cfg))
            try:
                o = ops.gemm(A, Bm, epilogue=epi, **kw)
            finally:
                L.check(lib.czc_gemm_set_tile(0))
            outs.append(torch.cat([x.reshape(-1) for x in o]) if isinstance(o, tuple) else o.reshape(-1))
        for o in outs[1:]:
            assert torch.equal(outs[0], o), epi


def test_adamw_with_folded_ema_is_bit_identical_to_the_two_passes(ops):
    """czc_adamw_ema_step == czc_adamw_step followed by czc_ema_update, bit for bit (weights, moments, key weights, both bf16
    mirrors), over several steps, n not a multiple of 4."""
    n, nb, em = 1_000_003, 4096, 0.9995
    pa, ka = rnd(n, seed=3), rnd(n, seed=4)
    pb_, kb = pa.clone(), ka.clone()
    ma, va, mb, vb = (torch.zeros(n, device=dev()) for _ in range(4))
    mir = [torch.zeros(nb, dtype=BF16, device=dev()) for _ in range(4)]    # p / k mirrors of path a, then of path b
    for step in range(1, 4):
        g = rnd(n, seed=10 + step, scale=0.5)
        ops.adamw_step(pa, g, ma, va, 1e-2, 0.9, 0.98, 1e-6, 0.01, step, 1.0, None, mir[0], nb)
        ops.ema_update(ka, pa, em, mir[1], nb)
        ops.adamw_ema_step(pb_, g, mb, vb, 1e-2, 0.9, 0.98, 1e-6, 0.01, step, kb, em, 1.0, None, mir[2], nb, mir[3])
        for a, b, what in ((pa, pb_, "p"), (ka, kb, "k"), (ma, mb, "m"), (va, vb, "v"), (mir[0], mir[2], "p mirror"), (mir[1], mir[3], "k mirror")):
            assert torch.equal(a, b), (step, what)
    assert not torch.equal(ka, rnd(n, seed=4))


# ------------------------------------------------------------------------------------------------
# in-batch head (row f4) and the L2-from-init penalty (row a13) on libczc
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,C,off,T,ls", [(6, 6, 128, 0, 1.0, 0.0), (8, 24, 768, 8, 0.05, 0.1), (256, 2048, 768, 1792, 0.05, 0.0),
                                            (33, 99, 320, 33, 0.5, 0.1)])
def test_inbatch_head_matches_torch(ops, B, N, C, off, T, ls):
    """czc_inbatch_head against the reference's arithmetic (src/inbatch.py:71-88) in fp32 torch on the device: loss, accuracy,
    stdq, stdk, d loss / d q and d loss / d keys; world sizes 1 (N = B) to 8 (N = 8 B), this rank's block at `off`."""
    q = rnd(B, C, seed=1, scale=0.4)
    keys = rnd(N, C, seed=2, scale=0.4)
    keys[off:off + B // 2] = q[:B // 2] * 0.9 + 0.02            # some easy positives: accuracy neither 0 nor 100
    if B >= 8:
        keys[3] = keys[off + 1]                                  # an exact tie with a positive's column: argmax takes the first
    stats, dq, dk = ops.inbatch_head(q, keys, off, T, ls)
    qv, kv = q.clone().requires_grad_(True), keys.clone().requires_grad_(True)
    scores = (qv / T) @ kv.t()
    labels = torch.arange(B, device=dev()) + off
    loss = torch.nn.functional.cross_entropy(scores, labels, label_smoothing=ls)
    loss.backward()
    # a logit z carries an fp32 rounding error of a few ulp(z) whatever the summation order, and exp() turns it into that
    # RELATIVE error of the probabilities: with un-normalised embeddings at T = 0.05 the logits reach ~2 500 (ulp 2.4e-4)
    zerr = 16 * 1.2e-7 * float(scores.detach().abs().max())
    report("inbatch loss", stats[0], loss.detach(), (2e-5 + zerr) * max(1.0, abs(float(loss.detach()))), 1e-5)
    assert round(float(stats[1]) * B / 100) == int((scores.argmax(-1) == labels).sum())      # the same rows hit (ties: first index)
    report("inbatch stdq", stats[2], torch.std(q, dim=0).mean(), 1e-6, 1e-5)
    report("inbatch stdk", stats[3], torch.std(keys[off:off + B], dim=0).mean(), 1e-6, 1e-5)
    report("inbatch dq", dq, qv.grad, (1e-5 + zerr) * float(qv.grad.abs().max()) + 1e-9, 1e-4)
    report("inbatch dkeys", dk, kv.grad, (1e-5 + zerr) * float(kv.grad.abs().max()) + 1e-9, 1e-4)
    again = ops.inbatch_head(q, keys, off, T, ls)
    assert torch.equal(again[0][:2], stats[:2]) and torch.equal(again[1], dq) and torch.equal(again[2], dk)   # fixed summation order
    only = ops.inbatch_head(q, keys, off, T, ls, want_dq=False, want_dkeys=False)
    assert only[1] is None and only[2] is None and torch.equal(only[0][:2], stats[:2])


def test_l2_from_init_kernels_match_torch(ops):
    n = 1_000_003                                                # not a multiple of 4: the scalar tail
    p, p0 = rnd(n, seed=1, scale=0.05), rnd(n, seed=2, scale=0.05)
    wd = 0.01
    g = rnd(n, seed=3, scale=0.01)
    g0 = g.clone()
    up = torch.tensor([0.5], device=dev())
    pen = ops.l2_from_init(p, p0, wd, grads=g, upstream=up, accumulate=True)
    ref = wd * ((p.double() - p0.double()) ** 2).sum()
    assert abs(float(pen) - float(ref)) <= 1e-5 * float(ref)
    report("l2 grad (accumulated)", g, g0 + 2 * wd * 0.5 * (p - p0), 1e-8, 1e-6)
    ops.l2_from_init(p, p0, wd, grads=g, upstream=None, accumulate=False)
    report("l2 grad (overwritten)", g, 2 * wd * (p - p0), 1e-8, 1e-6)
