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
    assert L.czc_abi_version() == 6


def test_argument_checks_fail_without_launching():
    from czech_contriever_amd import lib
    L = lib.load()
    # M-major A with a K-major B is not built; refused before any launch (no GPU needed)
    rc = L.czc_gemm_bf16(1, 0, 0, 64, 64, 64, None, 64, None, 64, None, 64, None, None, None, 1, None)
    assert rc != 0 and b"czc_gemm" in L.czc_last_error()
    s = lib.BertShape(512, 128, 2, 4, 256, 64, 2, 1e-12, 0.1, 0.1)  # head_dim 32
    assert L.czc_encoder_forward(ctypes.byref(s), None, None, None, None, 1, 8, 0, 0, 0, 0, 0, None, 0, None, 0, None, None) != 0
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


def test_bench_starts_its_own_ranks_when_invoked_bare(monkeypatch):
    """`python bench.py --gpus N` without a launcher (how the driver calls it) must start N ranks through
    torch.distributed.run as a child process and pass the flags on; under a launcher (RANK set) it must not."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                        # the launcher's exit code is ours
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_profiler_report_is_empty_without_launches():
    from czech_contriever_amd import lib
    L = lib.load()
    assert L.czc_prof_enable(1) == 0
    assert lib.prof_report() == []
    assert L.czc_prof_enable(0) == 0


def test_mirror_goes_stale_on_any_parameter_write():
    """the bf16 weight mirror must be rebuilt after load_state_dict / in-place parameter writes, also once the
    parameters have been re-pointed (p.data = view) and no longer share the flat array's version counter"""
    from src.contriever import Contriever, bert_config
    enc = Contriever(bert_config(vocab_size=64, hidden_size=64, num_hidden_layers=1, num_attention_heads=1,
                                 intermediate_size=64, max_position_embeddings=16))
    enc._flat = enc._flat.clone()      # what .cuda() does: new storage, parameters re-pointed at it
    enc._rebind()
    enc._mirror_version = enc._write_stamp()
    w = enc.get_parameter("encoder.layer.0.output.dense.weight")
    with torch.no_grad():
        w.mul_(2.0)
    assert enc._mirror_version != enc._write_stamp()
    enc._mirror_version = enc._write_stamp()
    enc.load_state_dict(enc.state_dict())
    assert enc._mirror_version == -1
    enc._mirror_version = enc._write_stamp()
    enc.reset_parameters()
    assert enc._mirror_version == -1
    import copy
    k = copy.deepcopy(enc)
    assert torch.equal(k.flat_params, enc.flat_params) and k._salt != enc._salt
    g = torch.Generator().manual_seed(0)
    state = torch.get_rng_state()
    copy.deepcopy(enc)
    assert torch.equal(torch.get_rng_state(), state)       # the copy draws nothing from the torch RNG


def test_dropout_stream_is_pinned_to_roles_and_resumable(tmp_path):
    """the dropout seed is (--seed, salt, training forwards so far).  The salt belongs to the role (MoCo:
    query 0, key 1; InBatch: 0) whatever other encoders the process built first, and a resumed run is put at the
    position the checkpointed run stood at (train.py: step * update_freq micro-steps), so it continues the stream
    instead of replaying the masks of steps 0..n.  The checkpoint keeps the reference's five keys."""
    import json
    import types
    from src import inbatch, moco
    from src.contriever import Contriever, bert_config
    d = tmp_path / "m"
    d.mkdir()
    cfg = dict(vocab_size=64, hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=64,
               max_position_embeddings=16)
    (d / "config.json").write_text(json.dumps(cfg))
    opt = types.SimpleNamespace(queue_size=8, momentum=0.9, temperature=1.0, label_smoothing=0.0, norm_doc=False,
                                norm_query=False, moco_train_mode_encoder_k=False, weight_decay=0.01,
                                weight_decay_from_init=False, retriever_model_id=str(d), pooling="average",
                                random_init=True, projection_size=64)
    Contriever(bert_config(**cfg))          # a "teacher" built first must not shift the students' salts
    m = moco.MoCo(opt)
    assert m.encoder_q.dropout_state() == {"salt": 0, "calls": 0}
    assert m.encoder_k.dropout_state() == {"salt": 1, "calls": 0}
    m.set_dropout_position(14)              # e.g. step 7 at update_freq 2
    assert m.encoder_q.dropout_state() == {"salt": 0, "calls": 14}
    assert m.encoder_k.dropout_state() == {"salt": 1, "calls": 14}
    assert not any("extra_state" in k for k in m.state_dict())       # the reference's state_dict keys are untouched
    ib = inbatch.InBatch(opt)
    ib.set_dropout_position(3)
    assert ib.encoder.dropout_state() == {"salt": 0, "calls": 6}     # queries and keys: two training forwards per micro-step


def test_stale_library_is_detected(monkeypatch):
    """lib.load() compares the hash compiled into libczc.so with the sources beside it: a binary built
    from other sources is rebuilt, and refused if the rebuild does not produce the expected hash."""
    from czech_contriever_amd import build, lib
    L = lib.load()
    assert L.czc_source_hash().decode() == build.source_hash() == lib._embedded_hash(lib.LIB_PATH)
    assert lib._fresh_or_rebuilt(lib.LIB_PATH) == lib.LIB_PATH
    calls = []
    monkeypatch.setattr(build, "source_hash", lambda extra=(): "0" * 40)      # "the sources changed"
    monkeypatch.setattr(build, "build", lambda force=False, verbose=True: calls.append(1))   # a rebuild that changes nothing
    with pytest.raises(RuntimeError, match="stale"):
        lib._fresh_or_rebuilt(lib.LIB_PATH)
    assert calls == [1]
    monkeypatch.setenv("CZC_SKIP_HASH_CHECK", "1")
    assert lib._fresh_or_rebuilt(lib.LIB_PATH) == lib.LIB_PATH
