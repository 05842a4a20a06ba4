"""ctypes binding of libczc.so (include/czc.h).  Fails loudly: there is no CPU fallback.

Every wrapper takes torch tensors (device memory is torch's), checks dtype / device / contiguity,
passes raw pointers plus the CURRENT torch stream, and raises RuntimeError with czc_last_error()
when the library refuses a call."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libczc.so")

vp, i32, i64, u32, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_float


class BertShape(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("num_layers", C.c_int32),
                ("num_heads", C.c_int32), ("intermediate_size", C.c_int32), ("max_position", C.c_int32),
                ("type_vocab_size", C.c_int32), ("layer_norm_eps", C.c_float), ("hidden_dropout", C.c_float),
                ("attn_dropout", C.c_float)]


SP = C.POINTER(BertShape)

# name -> (restype, argtypes); mirrors include/czc.h one to one
SIGNATURES = {
    "czc_last_error": (C.c_char_p, []),
    "czc_abi_version": (i32, []),
    "czc_param_count": (i64, [SP]),
    "czc_matrix_count": (i64, [SP]),
    "czc_param_offset": (i64, [SP, i32, i32]),
    "czc_encoder_workspace_bytes": (i64, [SP, i32, i32, i32]),
    "czc_encoder_backward_scratch_bytes": (i64, [SP, i32, i32]),
    "czc_encoder_forward": (i32, [SP, vp, vp, vp, vp, i32, i32, i32, i32, i32, u32, i32, vp, i64, vp, i32, vp, vp]),
    "czc_event_create": (i32, [C.POINTER(vp)]),
    "czc_event_destroy": (i32, [vp]),
    "czc_event_record": (i32, [vp, vp]),
    "czc_stream_wait_event": (i32, [vp, vp]),
    "czc_encoder_backward": (i32, [SP, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, u32, vp, i64, vp, i64, vp, vp, i32, i32, i32, vp]),
    "czc_transpose_matrices": (i32, [SP, vp, vp, vp]),
    "czc_infonce_workspace_bytes": (i64, [i32, i32, i32]),
    "czc_infonce_forward": (i32, [vp, vp, vp, i32, i32, i32, f32, f32, vp, i64, vp, vp, vp]),
    "czc_infonce_backward": (i32, [vp, vp, vp, i32, i32, i32, f32, f32, vp, f32, vp, i64, vp, vp]),
    "czc_enqueue": (i32, [vp, i32, i32, i32, vp, vp, vp, vp]),
    "czc_queue_import": (i32, [vp, i32, i32, vp, vp]),
    "czc_ema_update": (i32, [vp, vp, i64, f32, f32, vp, i64, vp]),
    "czc_adamw_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp, vp, i64, vp]),
    "czc_adamw_ema_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp, vp, i64, vp, vp, f32, f32, vp]),
    "czc_sumsq": (i32, [vp, i64, vp, vp, vp]),
    "czc_l2_from_init_forward": (i32, [vp, vp, i64, f32, vp, vp, vp]),
    "czc_l2_from_init_backward": (i32, [vp, vp, i64, f32, vp, vp, i32, vp]),
    "czc_inbatch_workspace_floats": (i64, [i32, i32]),
    "czc_inbatch_head": (i32, [vp, vp, i32, i32, i32, i32, f32, f32, vp, i64, vp, vp, vp, vp]),
    "czc_clip_coef": (i32, [vp, f32, f32, vp, vp]),
    "czc_clamp": (i32, [vp, i64, f32, f32, vp]),
    "czc_cast_bf16": (i32, [vp, vp, i64, vp]),
    "czc_gemm_bf16": (i32, [i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, i32, vp]),
    "czc_gemm_set_tile": (i32, [i32]),
    "czc_gemm_wgrad_pair": (i32, [i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp]),
    "czc_probe_arm": (i32, [i32, i32, i32, i32, i32, i32, i32]),
    "czc_probe_read": (i32, [C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "czc_prof_enable": (i32, [i32]),
    "czc_prof_read": (i64, [C.c_char_p, i64]),
    "czc_reduce_slabs": (i32, [vp, i32, i64, vp, i32, vp]),
    "czc_attention_forward": (i32, [vp, vp, i32, i32, i32, f32, u32, u32, vp, vp, vp]),
    "czc_attention_backward": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f32, u32, u32, vp, vp, vp, vp, vp]),
    "czc_attention_bias_scratch_floats": (i64, [i32, i32, i32]),
    "czc_attention_keep_words": (i64, [i32, i32, i32]),
    "czc_attention_forward_keep": (i32, [vp, vp, i32, i32, i32, f32, u32, u32, vp, vp, vp, vp]),
    "czc_attention_backward_keep": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f32, u32, u32, vp, vp, vp, vp, vp, vp]),
    "czc_colsum_scratch_floats": (i64, [i32]),
    "czc_reduce_partials": (i32, [vp, i32, i32, i32, vp, vp]),
    "czc_embed_ln_forward": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, f32, u32, u32, vp, vp, vp]),
    "czc_embed_ln_backward": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "czc_residual_ln_forward": (i32, [vp, vp, i32, i32, vp, vp, f32, f32, u32, u32, vp, vp, vp]),
    "czc_residual_ln_backward": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, f32, u32, u32, vp, vp, vp, vp, vp, vp, vp]),
    "czc_colsum_bf16": (i32, [vp, i32, i32, i32, vp, vp]),
    "czc_pool_forward": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    "czc_pool_backward": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "czc_dropout_mask": (i32, [u32, u32, f32, i64, vp, vp]),
    "czc_distill_workspace_floats": (i64, [i32, i32, i32]),
    "czc_distill_sim_loss": (i32, [vp, vp, i32, i32, i32, vp, i64, vp, vp, vp]),
    "czc_index_candidate_capacity": (i64, [i32]),
    "czc_index_begin": (i32, [vp, vp, i32, vp]),
    "czc_index_scan": (i32, [vp, i64, i32, i64, vp, i32, i32, i32, vp, vp, vp, vp, i64, vp]),
    "czc_source_hash": (C.c_char_p, []),
    "czc_index_finish": (i32, [vp, vp, vp, i64, i32, i32, vp, vp, vp]),
    "czc_index_coarse_capacity": (i64, [i32]),
    "czc_index_coarse_epsilon": (f32, []),
    "czc_index_prepare": (i32, [vp, i64, i32, f32, vp, vp, vp]),
    "czc_index_coarse_begin": (i32, [vp, vp, vp, vp, i32, vp]),
    "czc_index_coarse_scan": (i32, [vp, vp, i64, i32, i64, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "czc_index_rescore": (i32, [vp, vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, i64, vp]),
}

_lib = None


def load():
    """Load libczc.so (building it first if hipcc is present and it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        from . import build as _b
        _b.build()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m czech_contriever_amd.build`")
    # CZC_LIB_VARIANT=<suffix>: kernel A/B on one device -- a second build of the SAME sources with other compile-time
    # switches (python -m czech_contriever_amd.build --variant <suffix> -D...) sits beside the product as libczc_<suffix>.so
    variant = os.environ.get("CZC_LIB_VARIANT")
    path = LIB_PATH if not variant else os.path.join(os.path.dirname(LIB_PATH), f"libczc_{variant}.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing (CZC_LIB_VARIANT={variant})")
    if not variant:
        path = _fresh_or_rebuilt(path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = a symbol czc.h declares is not exported
        fn.restype, fn.argtypes = res, args
    if lib.czc_abi_version() != 6:
        raise RuntimeError("libczc.so ABI version mismatch")
    _lib = lib
    return lib


def _embedded_hash(path):
    """czc_source_hash() of the binary at `path`, read from the file's bytes (api.cpp keeps it behind a marker): a stale
    library is never mapped, so the rebuilt file of the same name is the one that gets loaded."""
    import mmap
    marker = b"CZC_SOURCE_HASH="
    with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as m:
        i = m.find(marker)
        if i < 0:
            return "none"
        j = m.find(b"\0", i)
        return m[i + len(marker):j].decode("ascii", "replace")


def _fresh_or_rebuilt(path):
    """The binary must come from the sources beside it (the .so is git-ignored and travels prebuilt: a stale one would
    otherwise run silently).  CZC_SKIP_HASH_CHECK=1 opts out (a deployment that ships the binary without csrc/)."""
    if os.environ.get("CZC_SKIP_HASH_CHECK") == "1":
        return path
    from . import build as _b
    if not all(os.path.exists(os.path.normpath(os.path.join(_b.CSRC, rel))) for rel in _b.SOURCES + _b.HEADERS):
        return path                       # no sources to compare with
    want = _b.source_hash()
    have = _embedded_hash(path)
    if have == want:
        return path
    try:
        _b.build(verbose=False)
    except Exception as e:
        raise RuntimeError(f"{path} was built from other sources (hash {have}, sources {want}) and cannot be rebuilt: {e}")
    if _embedded_hash(path) != want:
        raise RuntimeError(f"{path} is stale (hash {have}, sources {want}) even after a rebuild")
    return path


def prof_report():
    """[(name, calls, total_ms, flops_per_call, bytes_per_call)] collected since czc_prof_enable(1)"""
    lib = load()
    n = int(lib.czc_prof_read(None, 0))
    buf = C.create_string_buffer(n)
    lib.czc_prof_read(buf, n)
    rows = []
    for line in buf.value.decode().splitlines():
        name, calls, ms, fl, by = line.split("\t")
        rows.append((name, int(calls), float(ms), float(fl), float(by)))
    return rows


def last_error() -> str:
    return load().czc_last_error().decode("utf-8", "replace")


def check(rc: int):
    if rc != 0:
        raise RuntimeError(f"libczc: {last_error()} (code {rc})")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor], dtype=None) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libczc needs device tensors (there is no CPU path)")
    if not t.is_contiguous():
        raise RuntimeError("libczc needs contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"libczc: expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def shape_struct(cfg, hidden_dropout=None, attn_dropout=None) -> BertShape:
    """cfg: anything with HF BertConfig attribute names."""
    g = lambda k, d=None: getattr(cfg, k, d)
    return BertShape(
        int(g("vocab_size")), int(g("hidden_size")), int(g("num_hidden_layers")), int(g("num_attention_heads")),
        int(g("intermediate_size")), int(g("max_position_embeddings")), int(g("type_vocab_size", 2)),
        float(g("layer_norm_eps", 1e-12)),
        float(g("hidden_dropout_prob", 0.1) if hidden_dropout is None else hidden_dropout),
        float(g("attention_probs_dropout_prob", 0.1) if attn_dropout is None else attn_dropout))


P_KINDS = ["WQKV", "WO", "W1", "W2", "BQKV", "BO", "LN1W", "LN1B", "B1", "B2", "LN2W", "LN2B",
           "WORD", "POS", "TYPE", "ELNW", "ELNB"]
P = {k: i for i, k in enumerate(P_KINDS)}
EPI = {"store": 0, "bias": 1, "bias_gelu": 2, "dgelu": 3, "add": 4, "f32": 5, "bias_gelu_only": 8}
