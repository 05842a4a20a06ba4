"""Dev tool: turn one GPU session's outputs (gpurun_out/r2_final/*) into the tracked summaries under profiles/.
usage: python tools/make_profiles.py gpurun_out/r2_final r2"""
import csv, glob, hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
prof = os.path.join(ROOT, "profiles")
commit = os.environ.get("CZC_COMMIT") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()   # (the GPU box has no .git: the caller passes the hash)
with open(os.path.join(ROOT, "czech_contriever_amd", "csrc", "gemm_bf16.hip"), "rb") as f:
    gemm_sha = hashlib.sha1(f.read()).hexdigest()

def first(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    return g[0] if g else None

# 1. the bench line
bj = os.path.join(src, "bench_default.json")
bench = None
if os.path.exists(bj):
    bench = json.load(open(bj))
    json.dump(bench, open(os.path.join(prof, f"{tag}_bench_default.json"), "w"), indent=1)

# 2. rocprofv3 --kernel-trace --stats: per-kernel totals
ks = first("stats/**/*kernel_stats.csv")
if ks:
    rows = list(csv.reader(open(ks)))
    with open(os.path.join(prof, f"{tag}_final_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            w.writerow(r)
        w.writerow([f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --kernel-table 0 (10 steps in the trace; durations in ns as rocprofv3 writes them); commit {commit}"])

# 3. HBM-bound operations: bench's per-operation table (hipEvents, algorithmic bytes) next to rocprof's average per kernel
if bench and bench["roofline"].get("hbm_kernels"):
    avg = {}
    if ks:
        for r in csv.DictReader(open(ks)):
            avg[r["Name"]] = float(r["AverageNs"]) / 1e3
    kern_of = {"residual_ln_fwd": "residual_ln_fwd_kernel", "residual_ln_bwd": "residual_ln_bwd_kernel", "embed_ln_fwd": "embed_ln_fwd_kernel",
               "embed_ln_bwd": "embed_ln_bwd_kernel", "pool_fwd": "pool_fwd_kernel", "pool_bwd": "pool_bwd_kernel", "ema": "ema_kernel",
               "adamw": "adamw_kernel", "enqueue": "enqueue_kernel", "reduce_slabs": "reduce_slabs_kernel", "reduce_partials": "reduce_partials_kernel"}
    with open(os.path.join(prof, f"{tag}_hbm_kernels.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["operation", "calls_per_step", "algorithmic_MB_per_call", "avg_us_hipEvent", "GB_per_s", "frac_of_8TBps", "rocprof_kernel", "rocprof_avg_us"])
        allk = bench["roofline"]["hbm_kernels"] + [k for k in bench["roofline"].get("kernels", []) if k["name"].startswith("infonce")]
        for k in allk:
            kn = kern_of.get(k["name"], "")
            ra = next((v for n, v in avg.items() if kn and kn in n), "")
            if k["bound"] == "hbm":
                w.writerow([k["name"], k["calls_per_step"], k["mbytes_per_call"], k["us_per_call"], k["gbps"], k["frac"], kn, f"{ra:.2f}" if ra != "" else ""])
        w.writerow([f"# bench.py --kernel-table pass (hipEvent pairs around each operation, outside the timed region); algorithmic bytes as DESIGN.md prices them; commit {commit}"])

# 4. FFN-up HBM traffic from the two --pmc passes
def counter_avg(dirname, counter, kernel_sub):
    tot = n = 0
    for f in glob.glob(os.path.join(src, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]:
                tot += float(r["Counter_Value"]); n += 1
    return (tot / n if n else None), n
ffn = "gemm_kernel<0, 0, 2"
fetch, nf = counter_avg("pmc_fetch", "FETCH_SIZE", ffn)
write, nw = counter_avg("pmc_write", "WRITE_SIZE", ffn)
if fetch is not None and write is not None:
    M, N, K = 65536, 3072, 768
    algo = (M * K + N * K) * 2 + 2 * M * N * 2
    json.dump({
        "kernel": "czc::gemm_kernel<0,0,2,Cfg<2,4,8>,PP> (FFN-up: u = x.W1^T + b -> gelu'(u), gelu(u); M=65536 N=3072 K=768)",
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes with --kernel-trace), bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-table 0; {nf} / {nw} launches averaged",
        "commit": commit, "gemm_bf16_sha1": gemm_sha,
        "FETCH_SIZE_KB_raw": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
        "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B/lane stores",
        "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
        "algorithmic_bytes_per_launch": algo,
    }, open(os.path.join(prof, f"{tag}_pmc_ffn_up.json"), "w"), indent=1)
    # the session's own bench line was printed before these passes existed for this binary: fill its traffic field from them
    # (same session, same binary: the line's other figures are untouched)
    if bench and bench["roofline"].get("traffic") is None:
        bench["roofline"]["traffic"] = int((2 * fetch + write) * 1024)
        bench["roofline"]["traffic_source"] = (f"rocprofv3 --pmc passes of this session (commit {commit}, gemm_bf16.hip sha1 {gemm_sha}), "
                                               "filled in by tools/make_profiles.py")
        json.dump(bench, open(os.path.join(prof, f"{tag}_bench_default.json"), "w"), indent=1)

# 5. SQ counters per kernel over the bench (MFMA busy, waits) and over the attention driver
for name, out in (("pmc_sq", f"{tag}_mfma_busy.csv"), ("pmc_attn", f"{tag}_pmc_attention.csv")):
    d = os.path.join(src, name)
    if os.path.isdir(d):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), d], capture_output=True, text=True)
        if r.stdout.strip():
            with open(os.path.join(prof, out), "w") as f:
                f.write(r.stdout)
                f.write(f"# rocprofv3 --pmc <SQ counters> GRBM_GUI_ACTIVE --kernel-trace; per-dispatch values averaged per kernel; quad-cycle counters (SQ_WAVE_CYCLES, SQ_WAIT_*) as reported; commit {commit}\n")
# 6. beyond-L2 traffic of EVERY kernel of a step (the same two --pmc passes): launches per step, corrected FETCH and WRITE
#    per launch, distinct per-launch values where a kernel runs several shapes
def per_kernel(dirname, counter, scale):
    import collections, re
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[re.sub(r"\(.*", "", r["Kernel_Name"])].append(float(r["Counter_Value"]) * scale * 1024 / 1e6)
    return out
F, W = per_kernel("pmc_fetch", "FETCH_SIZE", 2.0), per_kernel("pmc_write", "WRITE_SIZE", 1.0)


def bench_ops_of(kernel):
    """bench.py's operation names (czc_prof: "gemm a<al> b<bl> epi<e> MxNxK[ splitk][ (two products)]") that launch this
    kernel template in the headline step -- the template's <AL, BL, EPI, Cfg, PP, ANT> against gemm_dispatch's choices
    (csrc/gemm_bf16.hip: ANT = streaming A rows for K >= 1536 and N <= 1024; PP = ping-pong K-loop for the long-K / wide-N
    K-major products).  Several names = the figures are this kernel's average over those operations."""
    import re
    m = re.search(r"gemm_kernel<(\d), (\d), (\d+), .*?>, (true|false), (true|false)>", kernel)
    if not m or not bench:
        return ""
    al, bl, epi, pp, ant = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4) == "true", m.group(5) == "true"
    names = []
    for k in (bench["roofline"].get("kernels_all") or bench["roofline"].get("kernels") or []):
        g = re.match(r"gemm a(\d) b(\d) epi(\d+) (\d+)x(\d+)x(\d+)", k["name"])
        if not g or (int(g.group(1)), int(g.group(2)), int(g.group(3))) != (al, bl, epi):
            continue
        Kd, Nd = int(g.group(6)), int(g.group(5))
        splitk = "splitk" in k["name"]
        k_pp = al == 0 and bl == 0 and not splitk and (Kd >= 1536 or Nd >= 2304)
        k_ant = k_pp and epi in (1, 4) and Kd >= 1536 and Nd <= 1024
        if (k_ant, k_pp) == (ant, pp):
            names.append(k["name"])
    return "|".join(names)


if F:
    steps = 3   # bench.py --steps 2 --warmup 1
    rows = []
    for k in F:
        f_, w_ = F[k], W.get(k, [0.0])
        rows.append((sum(f_) + sum(w_), k, len(f_) / steps, sum(f_) / len(f_), sum(w_) / max(1, len(w_)),
                     " ".join(str(v) for v in sorted(set(round(x) for x in f_))[:6])))
    rows.sort(reverse=True)
    with open(os.path.join(prof, f"{tag}_traffic_kernels.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches_per_step", "fetch_MB_per_launch", "write_MB_per_launch", "GB_per_step", "distinct_fetch_MB", "bench_ops"])
        for tot, k, n, fm, wm, dist in rows:
            if tot / steps < 1.0:
                continue
            w.writerow([k.strip(), round(n, 2), round(fm, 1), round(wm, 1), round(tot / steps / 1e3, 2), dist, bench_ops_of(k)])
        w.writerow([f"# gemm_bf16.hip sha1 {gemm_sha}; total {sum(r[0] for r in rows) / steps / 1e3:.1f} GB per step; rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) over bench.py --steps 2 --warmup 1; FETCH_SIZE doubled (gfx950: 64 B counted per 128-B request); L2 misses, i.e. traffic from the Infinity Cache and HBM together; commit {commit}"])
print("profiles written for commit", commit)
