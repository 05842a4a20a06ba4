"""Builds libczc.so (the gfx950 kernels + C ABI) in-tree with hipcc.  No GPU needed: hipcc cross-compiles."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libczc.so")
SOURCES = ["api.cpp", "gemm_bf16.hip", "attention.hip", "rowops.hip", "head.hip", "optim.hip", "encoder.hip", "flat_index.hip", "distill.hip"]
HEADERS = ["czc_common.h", "gemm_args.h", os.path.join("..", "..", "include", "czc.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# per-file additions.  gemm_bf16.hip: no SLP vectorisation -- it packed the GELU epilogue's fp32 polynomial into
# v_pk_fma_f32 chains (half rate on gfx950, a wait state between dependent ones, register-pair moves) and serialised the
# eight values of a segment; scalar v_fmaak_f32 with literal constants, stage by stage, is 25 % fewer VALU cycles
# attention.hip: the same packing of the softmax / dS arithmetic beside the MFMAs (backward 326 -> 322 us)
EXTRA_FLAGS = {"gemm_bf16.hip": ["-fno-slp-vectorize"], "attention.hip": ["-fno-slp-vectorize"]}


def source_hash(extra=()) -> str:
    """sha1 over every source and header of the library plus the compiler flags: compiled into the library
    (czc_source_hash) so that lib.load() can tell a binary built from OTHER sources than the ones beside it."""
    h = hashlib.sha1()
    for rel in SOURCES + HEADERS:
        path = os.path.normpath(os.path.join(CSRC, rel))
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(repr((FLAGS, sorted(EXTRA_FLAGS.items()), list(extra))).encode())
    return h.hexdigest()


def _hash_flag(digest: str):
    return [f'-DCZC_SOURCE_HASH="{digest}"']


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libczc.so cannot be built")
    return exe


def _stale(target: str, deps) -> bool:
    deps = list(deps) + [os.path.abspath(__file__)]      # flags live in this file
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(suffix: str, defines, verbose: bool = True) -> str:
    """A second library libczc_<suffix>.so from the same sources with extra -D switches or compiler flags (kernel A/B on one device:
    CZC_LIB_VARIANT=<suffix> makes lib.load() take it).  Objects go to build_<suffix>/; nothing of the product build is touched."""
    obj = OBJ + "_" + suffix
    os.makedirs(obj, exist_ok=True)
    out = os.path.join(HERE, f"libczc_{suffix}.so")
    digest = source_hash(defines)

    def one(src):
        sp = os.path.join(CSRC, src)
        op = os.path.join(obj, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc()] + FLAGS + list(defines) + EXTRA_FLAGS.get(src, []) + (_hash_flag(digest) if src == "api.cpp" else []) \
            + (["-x", "hip"] if sp.endswith(".cpp") else []) + ["-c", sp, "-o", op]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp}:\n{r.stdout}\n{r.stderr}")
        return op
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, SOURCES))
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[czc build] linked {out}", file=sys.stderr)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    digest = source_hash()
    stamp = os.path.join(OBJ, "source_hash.txt")
    old = open(stamp).read().strip() if os.path.exists(stamp) else ""
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        # api.cpp carries the hash of ALL sources: it is recompiled whenever any of them (or a flag) changed
        if force or _stale(op, [sp] + hdrs) or (src == "api.cpp" and old != digest):
            jobs.append((sp, op))

    def compile_one(job):
        sp, op = job
        base = os.path.basename(sp)
        cmd = [_hipcc()] + FLAGS + EXTRA_FLAGS.get(base, []) + (_hash_flag(digest) if base == "api.cpp" else []) \
            + (["-x", "hip"] if sp.endswith(".cpp") else []) + ["-c", sp, "-o", op]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp}:\n{r.stdout}\n{r.stderr}")
        return sp

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for sp in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[czc build] compiled {os.path.basename(sp)}", file=sys.stderr)
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[czc build] linked {LIB}", file=sys.stderr)
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:       # python -m czech_contriever_amd.build --variant nt -DATT_BWD_NT=1
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))    # -D switches and / or compiler flags (-mllvm X ...)
    else:
        print(build(force="--force" in sys.argv))
