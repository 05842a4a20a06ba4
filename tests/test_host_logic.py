"""CPU-only checks of the host side: the C ABI library loads and exports every symbol include/czc.h
declares, the command line matches the reference's flag table, the flat parameter layout, and loud
failure without a GPU (no CPU fallback exists)."""
import ctypes
import json
import os
import re
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from czech_contriever_amd import lib
    L = lib.load()
    header = open(os.path.join(ROOT, "include", "czc.h")).read()
    declared = set(re.findall(r"\b(czc_[a-z0-9_]+)\s*\(", header))
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    raw = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        getattr(raw, name)
    assert L.czc_abi_version() == 2


def test_argument_checks_fail_without_launching():
    from czech_contriever_amd import lib
    L = lib.load()
    # M-major A with a K-major B is not built; refused before any launch (no GPU needed)
    rc = L.czc_gemm_bf16(1, 0, 0, 64, 64, 64, None, 64, None, 64, None, 64, None, None, None, 1, None)
    assert rc != 0 and b"czc_gemm" in L.czc_last_error()
    s = lib.BertShape(512, 128, 2, 4, 256, 64, 2, 1e-12, 0.1, 0.1)  # head_dim 32
    assert L.czc_encoder_forward(ctypes.byref(s), None, None, None, None, 1, 8, 0, 0, 0, 0, 0, None, 0, None, None) != 0
    assert b"head_dim 64" in L.czc_last_error()


def test_flag_table_matches_reference_fixture():
    from src.options import Options
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "options_flags.json")))
    got = {a.dest: [repr(a.default), type(a).__name__, getattr(a.type, "__name__", None), a.nargs]
           for a in Options().parser._actions if a.dest != "help"}
    assert got == want
    opt = Options().parse(["--queue_size", "131072", "--unknown_flag", "1"])  # parse_known_args, as the reference
    assert opt.queue_size == 131072 and opt.momentum == 0.999 and opt.retriever_model_id == "czert"


def test_param_layout_is_bert_base_sized():
    from czech_contriever_amd import lib
    L = lib.load()
    s = lib.BertShape(30522, 768, 12, 12, 3072, 512, 2, 1e-12, 0.1, 0.1)
    assert L.czc_param_count(ctypes.byref(s)) == 108_891_648          # P
    assert L.czc_matrix_count(ctypes.byref(s)) == 12 * 7_077_888
    offs = [L.czc_param_offset(ctypes.byref(s), k, 3) for k in range(17)]
    assert all(o >= 0 and o % 64 == 0 for o in offs) and len(set(offs)) == 17


def test_no_cpu_fallback():
    from src.contriever import Contriever, bert_config
    enc = Contriever(bert_config(vocab_size=64, hidden_size=64, num_hidden_layers=1, num_attention_heads=1,
                                 intermediate_size=64, max_position_embeddings=16))
    with pytest.raises(RuntimeError, match="no CPU path"):
        enc(input_ids=torch.zeros(1, 4, dtype=torch.long), attention_mask=torch.ones(1, 4, dtype=torch.bool))


def test_scheduler_and_data_contract():
    from src import utils, data
    import random
    p = torch.nn.Parameter(torch.zeros(4))
    o = torch.optim.SGD([p], lr=1.0)
    s = utils.WarmupLinearScheduler(o, warmup=3, total=10, ratio=0.1)
    lam = [s.lr_lambda(i) for i in range(12)]
    assert lam[0] == 0.0 and abs(lam[3] - 1.0) < 1e-12 and abs(lam[10] - 0.1) < 1e-12 and lam[11] < lam[10]
    random.seed(0)
    c = data.randomcrop(torch.arange(100), 0.1, 0.5).tolist()
    assert 10 <= len(c) <= 50 and c == list(range(c[0], c[0] + len(c)))
    ids, mask = data.build_mask([torch.tensor([5, 6, 7]), torch.tensor([9])])
    assert ids.tolist() == [[5, 6, 7], [9, 0, 0]] and mask.tolist() == [[True, True, True], [True, False, False]]
    assert ids.dtype == torch.int64 and mask.dtype == torch.bool
