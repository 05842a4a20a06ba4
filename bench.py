#!/usr/bin/env python3
"""bench.py — contrastive pairs/sec of the MoCo training micro-step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, or bare:
   `python bench.py --gpus N` starts that launcher itself as a child process, before anything here touches a GPU)

One "step" = one full training micro-step of train.py's loop on one synthetic batch per GPU
(BASELINE.json config 2/3): query-encoder forward + backward, momentum EMA, key-encoder forward,
InfoNCE logits + cross-entropy forward/backward against the 131 072-slot queue, key all-gather +
enqueue, gradient all-reduce, AdamW (update_freq = 1 since target_batch_size = 256 * N).
bert-base-sized random-init weights (no checkpoints offline), chunk_length 256, per_gpu_batch_size 256,
T 0.05, momentum 0.9995, dropout 0.1 ON in both encoders as the reference trains.
Prints ONE JSON line on rank 0 (contract: the task statement; extra objects `roofline`, `cpu_baseline`).
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBPS = 8000.0             # HBM3E spec (same table); ~6.3 TB/s is what a float4 copy reaches
FLOP_PER_PAIR = 184.01e9           # 4 x encoder fwd (45.902 G) + head 0.403 G

BERT_BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, hidden_act="gelu",
                 initializer_range=0.02)


def make_opt(model_dir, batch, world, queue):
    return types.SimpleNamespace(
        retriever_model_id=model_dir, random_init=True, pooling="average", projection_size=768,
        queue_size=queue, momentum=0.9995, temperature=0.05, label_smoothing=0.0,
        norm_query=False, norm_doc=False, moco_train_mode_encoder_k=False,
        weight_decay=0.01, weight_decay_from_init=False, optim="adamw", scheduler="linear",
        lr=1e-5, beta1=0.9, beta2=0.98, eps=1e-6, warmup_steps=20000, total_steps=500000, lr_min_ratio=0.0,
        per_gpu_batch_size=batch, target_batch_size=batch * world, chunk_length=256, seed=0)


def synthetic_batches(rank, batch, seq, n, device, ragged=False):
    """SURVEY §8 d2.  Headline: all-true masks at `seq`.  ragged: per-sample lengths as the reference's crops produce
    them -- int(seq * U[0.1, 0.5]) + 2 tokens (src/data.py:553-560, 635-652) -- each side padded with id 0 to ITS OWN
    batch maximum (src/data.py:563-573), so queries and keys have different widths."""
    out = []
    for t in range(n):
        g = torch.Generator().manual_seed(1234 + 1000 * rank + t)
        sides = []
        for _ in range(2):
            ids = torch.randint(1000, 30522, (batch, seq), generator=g, dtype=torch.int64)
            m = torch.ones(batch, seq, dtype=torch.bool)
            if ragged:
                lens = (seq * (0.1 + 0.4 * torch.rand(batch, generator=g))).to(torch.int64) + 2
                width = int(lens.max())
                m = torch.arange(width)[None, :] < lens[:, None]
                ids = ids[:, :width] * m
            sides.append((ids.to(device), m.to(device)))
        out.append(dict(q_tokens=sides[0][0], q_mask=sides[0][1], k_tokens=sides[1][0], k_mask=sides[1][1]))
    return out


def host_cpu_share():
    """CPUs this process may actually use: affinity, capped by the cgroup CPU quota (a GPU box gives a
    one-GPU job a 16-CPU share of a much larger host; spawning one thread per visible core there
    oversubscribes the quota and stalls OpenMP)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                quota = max(1, int(int(q) / int(per)))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    quota = max(1, q // per)
        except Exception:
            pass
    if quota is not None:
        n = min(n, quota)
    env = os.environ.get("CZC_CPU_THREADS")
    if env:
        return max(1, int(env))
    return min(n, 16) if quota is None else n


def cpu_baseline(seq, cpu_batch=8, steps=3, queue_size=131072):
    """The CPU oracle (fp32, the reference's arithmetic) timed on this host: identical micro-step against the full
    131 072-slot queue, B reduced to 8 (pairs/s is batch-insensitive on CPU), 1 warm-up + `steps` >= 3 measured steps,
    median.  kind = "port"."""
    from oracle import moco_bert_oracle as O
    cores = host_cpu_share()
    torch.set_num_threads(cores)
    log(f"cpu baseline: {cores} threads (affinity {len(os.sched_getaffinity(0))}, os.cpu_count {os.cpu_count()})")
    cfg = O.BertShape()
    P = O.make_params(cfg, 0)
    queue = O.init_queue(768, queue_size, 0)
    st = O.MoCoState(cfg, P, queue, momentum=0.9995, temperature=0.05)
    g = torch.Generator().manual_seed(99)
    times = []
    log(f"cpu baseline: oracle built, {cores} threads")
    for i in range(steps + 1):
        q = torch.randint(1000, 30522, (cpu_batch, seq), generator=g)
        k = torch.randint(1000, 30522, (cpu_batch, seq), generator=g)
        m = torch.ones(cpu_batch, seq, dtype=torch.bool)
        t0 = time.perf_counter()
        st.forward_backward(q, m, k, m, training=True)
        st.optimizer_step(1e-5)
        times.append(time.perf_counter() - t0)
        log(f"cpu baseline step {i}: {times[-1]:.2f} s")
    import statistics
    med = statistics.median(times[1:])
    return {"value": round(cpu_batch / med, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{steps} full fp32 micro-steps (after 1 warm-up) of the CPU oracle at B={cpu_batch}, S={seq}, "
                      f"bert-base, queue {queue_size}, dropout on; median {med:.2f} s/step; {cores} threads = the CPU share this job "
                      f"is allowed (cgroup quota; the host shows {os.cpu_count()} logical CPUs, which a one-GPU job may not occupy)"}


PEAK_FP32_MATRIX_TFLOPS = 157.3    # MI355X fp32 MFMA (MI355X_MICROARCH.md): 256 CUs x 256 FLOP/clk x 2.4 GHz


def retrieval_bench(args):
    """SURVEY §8 f3 / BASELINE.json config 5 on ONE GPU, not the headline: (A) eval-mode encoder forward over synthetic
    256-token passages in batches of 512 (generate_passage_embeddings.py's default), (B) exact top-k inner-product search
    of `--index-queries` queries against `--index-rows` fp32 vectors resident in HBM.  A "step" of the reported value is
    one search batch."""
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        raise SystemExit("--objective retrieval is a single-GPU diagnostic")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    from czech_contriever_amd.flat_index import FlatIPIndex
    from src.contriever import Contriever, bert_config
    torch.manual_seed(0)
    enc = Contriever(bert_config(**BERT_BASE)).to(device).eval().half()
    g = torch.Generator().manual_seed(1234)
    ids = [torch.randint(1000, 30522, (512, args.seq), generator=g, dtype=torch.int64).to(device) for _ in range(4)]
    mask = torch.ones(512, args.seq, dtype=torch.bool, device=device)
    with torch.no_grad():
        for i in range(3):
            enc(input_ids=ids[i % 4], attention_mask=mask)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_enc = max(4, args.steps)
        for i in range(n_enc):
            enc(input_ids=ids[i % 4], attention_mask=mask)
        torch.cuda.synchronize()
        enc_s = time.perf_counter() - t0
    passages_per_s = 512 * n_enc / enc_s
    log(f"encode: {passages_per_s:.0f} passages/s ({args.seq} tokens, batch 512)")
    del enc
    torch.cuda.empty_cache()

    dim, rows, nq, k = 768, args.index_rows, args.index_queries, args.top_k
    gd = torch.Generator(device=device).manual_seed(0)
    ix = FlatIPIndex(dim)
    for lo in range(0, rows, 1_000_000):           # passage_retrieval.py --indexing_batch_size
        ix.add(torch.randn(min(1_000_000, rows - lo), dim, device=device, generator=gd) * 0.05)
    q = torch.randn(nq, dim, device=device, generator=gd)
    log(f"index: {rows} x {dim} fp32 = {rows * dim * 4 / 2**30:.1f} GiB in HBM")
    def timed(mode, steps):
        for _ in range(max(1, args.warmup // 3)):
            ix.search(q, k, mode=mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            out = ix.search(q, k, mode=mode)
        e1.record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, e0.elapsed_time(e1) / steps, out

    flop = 2.0 * rows * nq * dim
    # the fp32 matrix-core scan (every score exact as computed), then the default: bf16 scan with error margins + fp32
    # rescoring of the survivors — same answer
    _, exact_ms, (scores_x, found_x) = timed("exact", max(2, args.steps // 4))
    elapsed, dev_ms, (scores, found) = timed("coarse", args.steps)
    same = float((found == found_x).float().mean())
    tf = flop / (dev_ms * 1e-3) / 1e12
    tf_x = flop / (exact_ms * 1e-3) / 1e12

    # parity inside the bench: a sample of the result against the oracle's float64 scores of the returned rows
    sys.path.insert(0, ROOT)
    from oracle import flat_index_oracle
    import numpy as np
    rsel = found[:8].cpu().numpy()
    qs = q[:8].cpu().numpy().astype(np.float64)
    vec = lambda r: ix.blocks[r // 1_000_000][r % 1_000_000]          # blocks were added 1 M rows at a time
    exact = np.stack([torch.stack([vec(int(r)) for r in rsel[i]]).cpu().numpy().astype(np.float64) @ qs[i] for i in range(8)])
    err = float(np.abs(exact - scores[:8].cpu().numpy()).max())
    # CPU baseline: the oracle's fp32 sgemm + partial sort on a bounded slice of the same index, scaled by rows
    sample = min(rows, 200_000)
    dbs = ix.blocks[0][:sample].cpu().numpy()
    qh = q.cpu().numpy()
    threads = host_cpu_share()
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=threads):
        t1 = time.perf_counter()
        flat_index_oracle.search_fp32_blas(dbs, qh, k)
        cpu_s = time.perf_counter() - t1
    out = {
        "metric": "passage retrieval queries/sec (exact top-k over a flat fp32 inner-product index)",
        "value": round(nq * args.steps / elapsed, 1), "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 scoring + f32 rescoring (results as the f32 scan)", "data": "synthetic",
        "config": {"workload": f"NON-HEADLINE (BASELINE.json config 5): {rows} x {dim} fp32 passages in HBM, {nq} queries per search, top {k}; "
                               f"encoder forward over synthetic {args.seq}-token passages, batch 512",
                   "index_rows": rows, "dim": dim, "queries": nq, "top_k": k},
        "encode_passages_per_s": round(passages_per_s, 1),
        "encode_frac_bf16_peak": round(passages_per_s * 45.902e9 * (args.seq / 256.0) / (PEAK_BF16_TFLOPS * 1e12), 4),
        "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf / PEAK_BF16_TFLOPS, 4), "traffic": None,
                     "kernel": f"whole search = czc::gemm_kernel<0,0,9> (bf16 MFMA, filter epilogue) + margin_compact + rescore + topk_compact: "
                               f"{flop / 1e12:.2f} TFLOP of scoring per step",
                     "kernel_ms": round(dev_ms, 3)},
        "fp32_scan": {"ms_per_step": round(exact_ms, 3), "achieved": round(tf_x, 1), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                      "frac": round(tf_x / PEAK_FP32_MATRIX_TFLOPS, 4), "queries_per_s": round(nq / exact_ms * 1e3, 1),
                      "kernel": "czc::ip_scan_kernel (v_mfma_f32_32x32x2_f32) + czc::topk_compact_kernel",
                      "rows_identical_to_default_mode": same},
        "max_abs_score_error_vs_f64": err,
        "cpu_baseline": {"value": round(nq / (cpu_s * rows / sample), 2), "unit": "queries/s", "cores": threads, "kind": "port",
                         "sample": f"oracle/flat_index_oracle.py search_fp32_blas (numpy sgemm + argpartition) on {sample} of the {rows} rows, "
                                   f"{nq} queries: {cpu_s:.2f} s, scaled by rows"},
    }
    print(json.dumps(out), flush=True)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: one rank process per GPU through torch.distributed.run (the
    reference's launcher contract: torchrun env, train.py:428-430), started as a CHILD of this process -- which has not
    touched a GPU and never will (a process that has initialised the GPU must not exec or fork ranks) -- on a free
    loopback port.  Rank 0's JSON line goes to our stdout unchanged; our exit code is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"starting {n} ranks: {' '.join(cmd)}")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def newest_profile(suffix):
    """profiles/r<N>_<suffix> with the largest round NUMBER (r10 after r9, not before r2)"""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", f"r*_{suffix}")):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    return best[1] if best else None


def gemm_source_sha1():
    import hashlib
    with open(os.path.join(ROOT, "czech_contriever_amd", "csrc", "gemm_bf16.hip"), "rb") as f:
        return hashlib.sha1(f.read()).hexdigest()


def attach_traffic(kernels):
    """fetch_mb / write_mb per launch (rocprofv3 --pmc FETCH_SIZE x 2 and WRITE_SIZE, separate passes: tools/final_session.sh
    -> tools/make_profiles.py -> profiles/r<N>_traffic_kernels.csv) next to each operation's algorithmic bytes.  Only while
    gemm_bf16.hip still hashes to the source the counters were collected on (the file's last line names it)."""
    path = newest_profile("traffic_kernels.csv")
    if not path or not kernels:
        return
    import csv
    with open(path) as f:
        lines = f.read().splitlines()
    meta = next((l for l in lines if l.startswith("#")), "")
    if f"gemm_bf16.hip sha1 {gemm_source_sha1()}" not in meta:
        return
    rows = list(csv.DictReader([l for l in lines if not l.startswith("#")]))
    by_op = {}
    for r in rows:
        for op in filter(None, (r.get("bench_ops") or "").split("|")):
            by_op[op] = r
    for k in kernels:
        r = by_op.get(k["name"])
        if r is not None:
            k["fetch_mb"], k["write_mb"] = float(r["fetch_MB_per_launch"]), float(r["write_MB_per_launch"])
            shared = [o for o in r["bench_ops"].split("|") if o and o != k["name"]]
            if shared:
                k["traffic_is_kernel_average_with"] = shared


def kernel_table(rows, steps):
    """czc_prof rows -> per-operation dicts (ms per step, achieved TFLOP/s or GB/s, fraction of the bound's peak),
    largest first.  MFMA work is priced against the dense bf16 peak, byte-moving work against 8 TB/s HBM."""
    out = []
    for name, calls, ms, flops, nbytes in rows:
        if calls == 0 or ms <= 0:
            continue
        per = ms / calls
        d = {"name": name, "calls_per_step": round(calls / steps, 2), "ms_per_step": round(ms / steps, 4), "us_per_call": round(per * 1e3, 2),
             "algorithmic_mb": round(nbytes / 1e6, 2)}
        if flops > 0:
            tf = flops / (per * 1e-3) / 1e12
            d.update(bound="mfma", tflops=round(tf, 1), frac=round(tf / PEAK_BF16_TFLOPS, 4), gflop_per_call=round(flops / 1e9, 2))
        else:
            gbs = nbytes / (per * 1e-3) / 1e9
            d.update(bound="hbm", gbps=round(gbs, 1), frac=round(gbs / PEAK_HBM_GBPS, 4), mbytes_per_call=round(nbytes / 1e6, 2))
        out.append(d)
    out.sort(key=lambda d: -d["ms_per_step"])
    return out


class GpuTelemetry:
    """Shader clock and package power of the device this rank runs on, read from the amdgpu sysfs files (what rocm-smi
    reads; nothing is written, no setting is changed).  A background thread samples every `period` seconds."""

    def __init__(self, device_index, period=0.25):
        import glob
        import threading
        self.dir, self.samples, self.period = None, [], period
        self._stop = threading.Event()
        self._thread = None
        try:
            p = torch.cuda.get_device_properties(device_index)
            want = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}."
            for d in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(d)).startswith(want):
                    self.dir = d
        except Exception:
            pass

    def read(self):
        """(sclk_mhz, package_w) right now; None for what cannot be read"""
        import glob
        mhz = watts = None
        if self.dir:
            try:
                with open(os.path.join(self.dir, "pp_dpm_sclk")) as f:
                    cur = [l for l in f.read().splitlines() if l.rstrip().endswith("*")]
                mhz = float(cur[0].split(":")[1].lower().replace("mhz", "").replace("*", "").strip()) if cur else None
            except Exception:
                pass
            for name in ("power1_input", "power1_average"):
                try:
                    with open(glob.glob(os.path.join(self.dir, "hwmon", "hwmon*", name))[0]) as f:
                        watts = float(f.read()) / 1e6
                    break
                except Exception:
                    continue
        return mhz, watts

    def start(self):
        import threading
        self.samples = []
        self._stop.clear()

        def loop():
            while not self._stop.wait(self.period):
                self.samples.append(self.read())
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        """median (sclk_mhz, package_w) over the second half of the samples taken since start()"""
        import statistics
        self._stop.set()
        if self._thread:
            self._thread.join()
        half = self.samples[len(self.samples) // 2:]
        med = lambda v: round(statistics.median(v), 1) if v else None
        return med([a for a, _ in half if a is not None]), med([b for _, b in half if b is not None])


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (headline: 256)")
    ap.add_argument("--seq", type=int, default=256, help="chunk_length (headline: 256)")
    ap.add_argument("--queue", type=int, default=131072)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-table", type=int, default=3,
                    help="steps run AFTER the timed region with per-operation hipEvents, for roofline.kernels (0 = skip)")
    ap.add_argument("--steady-steps", type=int, default=200,
                    help="when --steps < 50: steps run AFTER the timed region (never part of `value`) for the line's `steady_state` "
                         "object -- the chip reaches its package power cap about a second into the load, so a short timed region "
                         "straddles the clock ramp (0 = skip)")
    ap.add_argument("--no-dropout", action="store_true", help="diagnostic only: the headline keeps dropout on")
    ap.add_argument("--ragged", action="store_true",
                    help="informational: sample lengths as the reference's random crops give them (27..130 of 256), pad id 0")
    ap.add_argument("--objective", choices=("moco", "inbatch", "retrieval", "distill"), default="moco",
                    help="inbatch = SURVEY §8 row f4 (one encoder, queries and keys both differentiable); retrieval = row f3 "
                         "(BASELINE.json config 5: passage encoding, then exact top-k over a flat fp32 index); distill = row f2 "
                         "(the student's step of distill.py with a synthetic 3584-d teacher called in-process); none is the headline")
    ap.add_argument("--index-rows", type=int, default=10_000_000, help="retrieval: passages in the flat index (config 5: 10 M)")
    ap.add_argument("--index-queries", type=int, default=2048, help="retrieval: queries per search (Indexer.search_knn's batch)")
    ap.add_argument("--top-k", type=int, default=100, help="retrieval: passages returned per query (passage_retrieval.py --n_docs)")
    args = ap.parse_args()
    if args.objective == "retrieval":
        return retrieval_bench(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args.gpus)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} under a launcher with WORLD_SIZE={world}: the two must agree")
    # CZC_BENCH_REHEARSE=1: a dry run of the N > 1 path on a box with fewer GPUs than ranks (ranks share the cards,
    # gloo instead of RCCL, which refuses two ranks on one device).  Its line says so in `data`; it is not a measurement.
    rehearse = world > 1 and os.environ.get("CZC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from czech_contriever_amd import lib as L
    from czech_contriever_amd.parallel import DataParallel
    from src import inbatch, moco, utils
    lib = L.load()

    mdir = tempfile.mkdtemp(prefix="czc_bench_")
    cfg = dict(BERT_BASE)
    if args.no_dropout:
        cfg.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    with open(os.path.join(mdir, "config.json"), "w") as f:
        json.dump(cfg, f)
    opt = make_opt(mdir, args.batch, world, args.queue)
    torch.manual_seed(opt.seed)
    if args.objective == "distill":
        import distill
        opt.distill_weight = 0.5
        model = moco.MoCoDistill(opt).to(device)
        teacher = distill.SyntheticTeacher(3584, cfg["vocab_size"], 0, device)
    else:
        model = (moco.MoCo(opt) if args.objective == "moco" else inbatch.InBatch(opt)).to(device)
    model.train()
    optimizer, scheduler = utils.set_optim(opt, model)
    if hasattr(model, "allow_ema_fold"):
        # as train.py does on every step that no checkpoint follows: the next forward's momentum EMA rides in the AdamW pass
        # (same arithmetic, bit for bit; CZC_EMA_FOLD=0 keeps the separate launch for A/B)
        model.allow_ema_fold(os.environ.get("CZC_EMA_FOLD", "1") != "0")
    model = DataParallel(model, optimizer)
    batches = synthetic_batches(rank, args.batch, args.seq, 16, device, ragged=args.ragged)
    if rank == 0:
        log(f"model on {device}, {sum(p.numel() for p in model.module.get_encoder().parameters())/1e6:.1f} M params/encoder; warm-up {args.warmup}")

    def step(i):
        if args.objective == "distill":   # distill.py's student: contrastive + similarity-matrix loss against the teacher
            b = batches[i % len(batches)]
            loss, q_emb, _ = model(**b, process_group=None, stats_prefix="train")
            aux = distill.distill_loss(q_emb, teacher.encode_ids(b["q_tokens"]))
            (0.5 * loss + 0.5 * aux).backward()
        else:
            loss, _ = model(**batches[i % len(batches)], stats_prefix="train")
            loss.backward()
        optimizer.step()
        scheduler.step()
        model.zero_grad()
        return loss

    if os.environ.get("CZC_GEMM_CFG"):   # tuning A/B only (czc_gemm_set_tile, include/czc.h)
        L.check(lib.czc_gemm_set_tile(int(os.environ["CZC_GEMM_CFG"])))
    for i in range(args.warmup):
        last = step(i)
        torch.cuda.synchronize()
        if rank == 0:
            log(f"warm-up step {i} done, loss {float(last.detach()):.4f}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    torch.cuda.synchronize()
    if args.warmup and not torch.isfinite(last.detach()).item():
        raise SystemExit("non-finite loss in warm-up")

    # dominant kernel = the FFN-up GEMM (bias+GELU epilogue): hipEvents around each of its launches
    M, N, K = args.batch * args.seq, cfg["intermediate_size"], cfg["hidden_size"]
    L.check(lib.czc_probe_arm(0, 0, L.EPI["bias_gelu"], M, N, K, 24 * args.steps))

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step times (no host sync inside)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        last = step(args.warmup + i)
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    tot_ms, cnt = ctypes.c_float(0), ctypes.c_int(0)
    L.check(lib.czc_probe_read(ctypes.byref(tot_ms), ctypes.byref(cnt)))
    L.check(lib.czc_probe_arm(0, 0, 0, 0, 0, 0, 0))

    # steady state (outside the timed region, every rank: the steps hold collectives): BASELINE.md §3 asks for >= 50 measured
    # steps; the driver's 5 + 20 straddle the power ramp.  Clock and package power are sampled from sysfs beside it.
    # The same steps time the main stream's waits for the two exchanges (0 on one rank; on N ranks: what the collectives cost
    # beyond the compute that covers them).
    steady = None
    n_tail = args.steady_steps if args.steps < 50 else 0
    if n_tail > 0:
        tele = GpuTelemetry(local_rank)
        kxs = list(getattr(model.module, "_key_exchange", {}).values())
        model.exposed = []
        for kx in kxs:
            kx.exposed = []
        skip = min(20, n_tail // 4)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tele.start()
        for i in range(n_tail):
            if i == skip:
                ev0.record()
            step(args.warmup + args.steps + i)
        ev1.record()
        torch.cuda.synchronize()
        sclk, watts = tele.stop()
        n_meas = n_tail - skip
        steady = {"ms_per_step": round(ev0.elapsed_time(ev1) / n_meas, 3), "steps": n_meas,
                  "pairs_per_s": round(args.batch * world * n_meas / (ev0.elapsed_time(ev1) * 1e-3), 1),
                  "sclk_mhz": sclk, "package_w": watts,
                  "allreduce_exposed_ms": round(model.exposed_ms() / n_tail, 4),
                  "key_exchange_exposed_ms": round(sum(kx.exposed_ms() for kx in kxs) / n_tail, 4),
                  "note": "rank 0's device and stream; steps run after the timed region, not part of `value`"}
        model.exposed = None
        for kx in kxs:
            kx.exposed = None

    # per-operation table: a few more steps, OUTSIDE the timed region, with a hipEvent pair around every operation
    kernels = None
    # (the per-operation event pairs mean something only while operations run one after another: the key encoder's
    #  second stream is switched off for these extra steps)
    overlap_was = getattr(model.module, "key_overlap", None)
    if overlap_was is not None and args.kernel_table > 0:
        model.module.key_overlap = False
    if rank == 0 and args.kernel_table > 0:
        L.check(lib.czc_prof_enable(1))
        for i in range(args.kernel_table):
            step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        kernels = kernel_table(L.prof_report(), args.kernel_table)
        attach_traffic(kernels)
        if os.environ.get("CZC_BENCH_ALLOPS"):       # dev: every operation of a step, largest first
            for k in kernels:
                log(f"op {k['name']:44s} x{k['calls_per_step']:6.1f} {k['us_per_call']:9.2f} us  {k['ms_per_step']:8.3f} ms/step  "
                    + (f"{k['tflops']:7.1f} TF/s" if k['bound'] == 'mfma' else f"{k['gbps']:7.1f} GB/s"))
        L.check(lib.czc_prof_enable(0))
    elif args.kernel_table > 0:
        for i in range(args.kernel_table):     # the other ranks take the same collectives
            step(args.warmup + args.steps + i)
    if overlap_was is not None:
        model.module.key_overlap = overlap_was

    if rank == 0:
        log(f"timed region: {args.steps} steps in {elapsed:.3f} s")
        pairs = args.batch * world * args.steps
        value = pairs / elapsed
        kern_ms = tot_ms.value / max(1, cnt.value)
        kern_tflops = (2.0 * M * N * K) / (kern_ms * 1e-3) / 1e12 if cnt.value else None
        # HBM bytes of the roofline kernel come from separate rocprofv3 --pmc passes (tools/pmc_ffn_up.py); the file names
        # the commit of gemm_bf16.hip it was collected on, and a figure collected on another version of the kernel is
        # reported as null rather than passed off as current
        traffic, traffic_note = None, None
        tpath = newest_profile("pmc_ffn_up.json") or ""
        if os.path.exists(tpath):
            with open(tpath) as f:
                pmc = json.load(f)
            now = gemm_source_sha1()
            if pmc.get("gemm_bf16_sha1") == now:
                traffic = pmc.get("hbm_bytes_per_launch")
            traffic_note = (f"rocprofv3 --pmc passes of commit {pmc.get('commit')}, gemm_bf16.hip sha1 {pmc.get('gemm_bf16_sha1')}"
                            + ("" if traffic is not None else f" -- STALE (kernel source is now {now}): not reported"))
        headline = (args.batch == 256 and args.seq == 256 and args.queue == 131072 and not args.no_dropout
                    and args.objective == "moco" and not args.ragged)
        # in-batch: q fwd+bwd and k fwd+bwd = 6 encoder forwards' worth (275.4 G) + a 2*B*W*B*768 score product
        flop_per_pair = FLOP_PER_PAIR if args.objective in ("moco", "distill") else 6 * 45.902e9 * (args.seq / 256.0)
        out = {
            "metric": "contrastive pairs/sec (seq256, bsz256/GPU, queue131k)",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic" if not rehearse else "synthetic; REHEARSAL: ranks share GPUs over gloo, not a measurement",
            "config": {"workload": "MoCo CZERT (bert-base-sized) bf16, chunk_length=256, per_gpu_batch_size=256, "
                                   "queue_size=131072, temperature=0.05" if headline else
                                   f"NON-HEADLINE diagnostic: objective={args.objective} batch={args.batch} seq={args.seq} ragged={args.ragged} queue={args.queue} dropout={not args.no_dropout}",
                       "global_batch": args.batch * world, "seq_len": args.seq, "queue_size": args.queue,
                       "parallelism": f"dp{world}", "dp": model.settings(), "weights": "random-init bert-base (seed 0)", "dropout": 0.0 if args.no_dropout else 0.1,
                       "step": "q fwd+bwd, EMA, k fwd, InfoNCE fwd+bwd, all-gather+enqueue, grad all-reduce, AdamW"
                               if args.objective == "moco" else
                               "MoCo step + synthetic teacher embeddings (3584-d) + similarity-matrix MSE fwd+bwd"
                               if args.objective == "distill" else
                               "q fwd+bwd, k fwd+bwd (same encoder), key all-gather with grad, in-batch CE, grad all-reduce, AdamW"},
            "roofline": {"bound": "mfma", "achieved": None if kern_tflops is None else round(kern_tflops, 1),
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": None if kern_tflops is None else round(kern_tflops / PEAK_BF16_TFLOPS, 4),
                         "traffic": traffic,
                         "kernel": "czc::gemm_kernel<0,0,2> FFN-up x.W1^T + bias -> (gelu'(u), gelu(u)), "
                                   f"M={M} N={N} K={K}, {2.0 * M * N * K / 1e9:.1f} GFLOP/launch",
                         "kernel_ms": round(kern_ms, 4), "launches_timed": cnt.value, "traffic_source": traffic_note,
                         "kernel_choice": "the largest single GEMM shape below its siblings' rate (FFN-up with its two-output "
                                          "GELU epilogue); by time the largest operations are the ones listed first in `kernels`",
                         "kernels": None if kernels is None else kernels[:16],
                         "hbm_kernels": None if kernels is None else [k for k in kernels if k["bound"] == "hbm"][:10],
                         "whole_step_frac": None if args.ragged else round(value * flop_per_pair / (world * PEAK_BF16_TFLOPS * 1e12), 4)},
            "step_ms": (lambda v: {"p10": round(v[len(v) // 10], 3), "median": round(v[len(v) // 2], 3), "p90": round(v[(9 * len(v)) // 10], 3)})(
                sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))),
            "final_loss": round(float(last.detach()), 4),
            "steady_state": steady,
        }
        if world == 1 and not args.no_cpu_baseline and args.objective == "moco":
            out["cpu_baseline"] = cpu_baseline(args.seq)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
