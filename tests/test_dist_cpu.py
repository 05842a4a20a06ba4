"""N > 1 path on CPU: world_size 2 over gloo (the RCCL calls are the same torch.distributed calls).
Checks the pieces of the multi-GPU design that do not need a GPU: rank-order key all-gather (so the
replicated queues stay bit-identical), the autograd all-gather (InBatch-style), flat-gradient averaging,
and the cross-rank stats reduction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import moco_bert_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from src import dist_utils, utils
        from czech_contriever_amd import parallel
        torch.manual_seed(rank)
        C, K, B = 16, 64, 4
        # replicated queue, identical start on every rank
        queue = torch.nn.functional.normalize(O.det_uniform((C, K), 5), dim=0)
        ptr = 0
        for step in range(10):   # 10 * 2 * 4 = 80 > 64: wraps
            keys = O.det_uniform((B, C), 100 + 10 * step + rank)
            allk = dist_utils.gather_nograd(keys)
            assert allk.shape == (world * B, C)
            for r in range(world):  # rank order
                assert torch.equal(allk[r * B:(r + 1) * B], O.det_uniform((B, C), 100 + 10 * step + r))
            queue, ptr = O.enqueue(queue, ptr, allk)
        digest = queue.double().sum().item()
        all_digest = [None] * world
        dist.all_gather_object(all_digest, (digest, ptr))
        assert all(d == all_digest[0] for d in all_digest), all_digest

        # the key exchange MoCo uses under data parallelism (czech_contriever_amd.parallel.KeyExchange): gathered in two
        # batch ranges (first half early, second half late) or in one piece, the rows come out in the SAME rank-major
        # order, so queue and pointer stay bit-identical to the unsplit path (and to the reference's cat(all_gather))
        kx_split, kx_whole = parallel.KeyExchange(), parallel.KeyExchange()
        assert parallel.KeyExchange.split_for(8, 256) == 4 and parallel.KeyExchange.split_for(1, 256) == 0
        assert parallel.KeyExchange.split_for(6, 5) == 0 and parallel.KeyExchange.split_for(6, 16) == 3
        qa, qb_, pa, pb = queue.clone(), queue.clone(), ptr, ptr
        for step in range(10):
            keys = O.det_uniform((B, C), 300 + 10 * step + rank)
            got = {}
            for name, kx, n in (("split", kx_split, B // 2), ("whole", kx_whole, 0)):
                kx.gather(keys, n)
                kx.enqueue(lambda rows, name=name: got.__setitem__(name, rows.clone()))
            ref = torch.cat([O.det_uniform((B, C), 300 + 10 * step + r) for r in range(world)])   # rank order
            assert torch.equal(got["split"], ref) and torch.equal(got["whole"], ref)
            qa, pa = O.enqueue(qa, pa, got["split"])
            qb_, pb = O.enqueue(qb_, pb, got["whole"])
        assert torch.equal(qa, qb_) and pa == pb == (ptr + 10 * world * B) % K

        # autograd all-gather: each rank gets the SUM over ranks of the gradient of its slice
        x = O.det_uniform((B, C), 7 + rank).requires_grad_(True)
        g = dist_utils.gather(x)
        w = O.det_uniform((world * B, C), 9)
        ((g * w).sum() * (rank + 1)).backward()
        expect = w[rank * B:(rank + 1) * B] * sum(r + 1 for r in range(world))
        assert torch.allclose(x.grad, expect)

        # flat gradient averaging (what DataParallel launches after the encoder backward)
        flat = O.det_uniform((1001,), 11 + rank)
        ref = sum(O.det_uniform((1001,), 11 + r) for r in range(world)) / world
        parallel.average_(flat)
        assert torch.allclose(flat, ref, atol=1e-7)

        # DataParallel's ranged reduction: the flat gradient is averaged range by range (matrix slices of
        # 3 layers, top first, then the vector + embedding tail) and every element exactly once
        from src.contriever import Contriever, bert_config
        enc = Contriever(bert_config(vocab_size=64, hidden_size=64, num_hidden_layers=4, num_attention_heads=1,
                                     intermediate_size=128, max_position_embeddings=16))
        dp = parallel.DataParallel(enc, chunk_layers=3)
        assert enc.backward_chunk_layers == 3
        n = enc.flat_params.numel()
        enc._flat_grad = O.det_uniform((n,), 21 + rank)
        want = sum(O.det_uniform((n,), 21 + r) for r in range(world)) / world
        la = enc._n_mat // 4
        seen = torch.zeros(n, dtype=torch.int32)
        hi = 4
        while hi > 0:
            lo = max(0, hi - 3)
            dp._reduce_range(enc, lo * la, hi * la); seen[lo * la:hi * la] += 1
            hi = lo
        dp._reduce_range(enc, enc._n_mat, n); seen[enc._n_mat:] += 1
        assert int(seen.min()) == 1 and int(seen.max()) == 1
        assert torch.allclose(enc._flat_grad, want, atol=1e-7)

        # stats: weighted mean over ranks lands on rank 0
        st = utils.WeightedAvgStats()
        st.update({"train/loss": (float(rank + 1), 4 * (rank + 1))})
        avg = st.average_stats
        if rank == 0:
            assert abs(avg["train/loss"] - (1 * 4 + 2 * 8) / 12) < 1e-6
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _distill_worker(rank, world, port, out):
    """Teacher (rank 0) + two students over gloo: the per-step exchange of distill.py on host tensors."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import distill
        from src import dist_utils, utils
        everyone = dist.new_group(list(range(world)))
        control = dist.new_group(list(range(world)), backend="gloo")
        students = dist.new_group(list(range(1, world)))
        B, dim, pad, vocab, cpu = 4, 24, 0, 50, torch.device("cpu")
        teacher = distill.SyntheticTeacher(dim, vocab, pad, cpu)
        widths = {1: [7, 5, 9], 2: [6, 8, 9]}          # students crop to different lengths
        if rank == 0:
            served = distill.serve_teacher(teacher, everyone, control, B, cpu)
            assert served == 3
        else:
            link = distill.StudentLink(everyone, control, B, dim, pad)
            for step, w in enumerate(widths[rank]):
                ids = O.det_ints((B, w), 300 + 10 * step + rank, 1, vocab)
                assert link.send_ids(ids)
                got = link.recv_embeddings(cpu)
                # what the teacher saw: this student's ids padded to the longest of the step; padding is masked out
                assert torch.allclose(got, teacher.encode_ids(ids), atol=1e-6)
            # the students' own group: key gather in student order and the stats reduction onto global rank 1
            keys = O.det_uniform((B, 8), 40 + rank)
            allk = dist_utils.gather_nograd(keys, students)
            assert torch.equal(allk, torch.cat([O.det_uniform((B, 8), 41), O.det_uniform((B, 8), 42)]))
            st = utils.WeightedAvgStatsDistill(students)
            st.update({"train/loss": (float(rank), 4 * rank)})
            avg = st.average_stats
            if rank == 1:
                assert abs(avg["train/loss"] - (1 * 4 + 2 * 8) / 12) < 1e-6
            assert not link.send_ids(None, stop=True)
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        out.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_teacher_student_exchange_world3_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_distill_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok"), (2, "ok")], res


def _student_dp_worker(rank, world, port, out):
    """distill.py's layout -- rank 0 the teacher, ranks 1..W-1 student replicas -- with the students' gradient exchange under
    DataParallel(process_group=students) and update_freq = 2 (train.py / distill.py accumulate micro-steps and, like the
    reference's DDP without no_sync (train.py:263-266, distill.py:466-470), all-reduce on EVERY micro-step).  The encoder
    backward itself needs a GPU; here the hook protocol of FlatBertEncoder._backward_impl is driven by hand with known
    per-rank gradients: pre-access wait, accumulate, ranges top-down, tail, post-backward."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      CZC_DP_CHUNK_LAYERS="2", CZC_DP_PER_TILE="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from czech_contriever_amd import parallel
        students = dist.new_group(list(range(1, world)))      # every rank creates the group, as distill.py does
        if rank == 0:
            dist.barrier()                                    # the teacher takes no part in the students' collectives
            out.put((rank, "ok"))
            return
        from src.contriever import Contriever, bert_config
        enc = Contriever(bert_config(vocab_size=64, hidden_size=64, num_hidden_layers=4, num_attention_heads=1,
                                     intermediate_size=128, max_position_embeddings=16))
        dp = parallel.DataParallel(enc, process_group=students)
        assert dp.settings() == {"chunk_layers": 2, "per_tile": "0", "world": world - 1}   # the env knobs, read once
        assert enc.backward_chunk_layers == 2
        n, nl = enc.flat_params.numel(), 4
        la = enc._n_mat // nl
        ranges = []
        orig = dp._reduce_range
        dp._reduce_range = lambda e, lo, hi: (ranges.append((lo, hi)), orig(e, lo, hi))[1]
        enc._range_hooks = [dp._reduce_range]

        def micro_backward(m, accumulate):
            for hook in enc._pre_grad_access:                 # an earlier all-reduce may still own the array
                hook()
            g = O.det_uniform((n,), 500 + 10 * m + rank)
            if accumulate:
                enc._flat_grad += g
            else:
                enc._flat_grad = g.clone()
            hi = nl
            while hi > 0:                                     # FlatBertEncoder._backward_impl: ranges of chunk layers, top first
                lo = max(0, hi - enc.backward_chunk_layers)
                for hook in enc._range_hooks:
                    hook(enc, lo * la, hi * la)
                hi = lo
            for hook in enc._range_hooks:
                hook(enc, enc._n_mat, n)
            for hook in enc._post_backward:
                hook(enc)

        micro_backward(0, False)
        micro_backward(1, True)
        dp.wait()
        mean = lambda m: sum(O.det_uniform((n,), 500 + 10 * m + r) for r in range(1, world)) / (world - 1)
        # averaging the accumulated array on every micro-step: mean_r(mean(g0) + g1_r) = mean(g0) + mean(g1)
        assert torch.allclose(enc._flat_grad, mean(0) + mean(1), atol=1e-6)
        assert ranges[:3] == [(2 * la, 4 * la), (0, 2 * la), (enc._n_mat, n)] and len(ranges) == 6
        assert dp.exposed_ms() == 0.0                         # no device, nothing timed
        dist.barrier()
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        out.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_student_group_data_parallel_world3_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_student_dp_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok"), (2, "ok")], res
