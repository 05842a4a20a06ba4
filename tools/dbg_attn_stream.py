"""Dev tool: streaming vs fused attention backward on the same inputs (two child processes), pattern of the differences."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from czech_contriever_amd import lib as L, ops
    L.load()
    dev = torch.device("cuda:0")
    B, S, nh = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]); p = float(sys.argv[6]); H = nh * 64
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn(B * S, 3 * H, device=dev, generator=g).to(torch.bfloat16)
    dctx = torch.randn(B * S, H, device=dev, generator=g).to(torch.bfloat16)
    mask = torch.ones(B, S, dtype=torch.bool, device=dev)
    if B > 1: mask[1, S // 2:] = False
    ctx, lse = ops.attention_forward(qkv, mask, B, S, nh, p, 5, 3)
    outs = []
    for rep in range(3):
        db = torch.zeros(3 * H, device=dev)
        dq = ops.attention_backward(qkv, mask, ctx, dctx, lse, B, S, nh, p, 5, 3, d_bias=db)
        outs.append((dq.clone(), db.clone()))
    torch.cuda.synchronize()
    torch.save({"dqkv": [o[0].cpu() for o in outs], "db": [o[1].cpu() for o in outs]}, sys.argv[2])
    sys.exit(0)
import torch
shape = sys.argv[1:5] if len(sys.argv) > 4 else ["2", "256", "12", "0.0"]
for mode in ("0", "1"):
    env = dict(os.environ, CZC_ATTN_STREAM_BWD=mode)
    subprocess.check_call([sys.executable, __file__, "child", f"gpurun_out/dbg_attn_{mode}.pt"] + shape, env=env)
a = torch.load("gpurun_out/dbg_attn_0.pt"); b = torch.load("gpurun_out/dbg_attn_1.pt")
B, S, nh = int(shape[0]), int(shape[1]), int(shape[2]); H = nh * 64
ref = a["dqkv"][0].float().view(B, S, 3, nh, 64)
for rep in range(3):
    got = b["dqkv"][rep].float().view(B, S, 3, nh, 64)
    err = (got - ref).abs()
    tol = 0.03 + 0.02 * ref.abs()
    bad = err > tol
    print(f"rep {rep}: bad per part (Q, K, V): {[int(bad[:, :, i].sum()) for i in range(3)]}  max err {float(err.max()):.3e}  same as rep0: {torch.equal(b['dqkv'][rep], b['dqkv'][0])}")
    idx = torch.nonzero(bad)
    if rep == 0 and len(idx):
        import collections
        print("  by batch:", collections.Counter(idx[:, 0].tolist()))
        print("  by part:", collections.Counter(idx[:, 2].tolist()))
        print("  by head:", sorted(collections.Counter(idx[:, 3].tolist()).items()))
        print("  by seq pos % 32:", sorted(collections.Counter((idx[:, 1] % 32).tolist()).items()))
        print("  by seq pos // 32:", sorted(collections.Counter((idx[:, 1] // 32).tolist()).items()))
        print("  by d:", sorted(collections.Counter(idx[:, 4].tolist()).items()))
        for t in idx[:24].tolist():
            print("   ", t, float(got[tuple(t)]), float(ref[tuple(t)]), hex(b["dqkv"][0].view(torch.int16).view(B, S, 3, nh, 64)[tuple(t)].item() & 0xffff))
print("dbias max diff:", float((a["db"][0] - b["db"][0]).abs().max()), " fused self-repro:", torch.equal(a["dqkv"][0], a["dqkv"][1]))
