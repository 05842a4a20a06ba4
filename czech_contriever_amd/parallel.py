"""Data parallelism for the MoCo path: one process per GPU, RCCL over xGMI (torch.distributed "nccl").

Replaces torch DDP at train.py:483-488 of the reference.  DDP cannot be used here: parameter
gradients are written by libczc straight into one flat fp32 array, not through autograd hooks.
Two exchanges per micro-step, both on side HIP streams:

`DataParallel` -- gradient mean all-reduce of the flat gradient array in 5 collectives: the encoder
    backward is issued in ranges of 3 layers (top first); as soon as a range has been enqueued its weight-matrix
    gradients (one contiguous 85 MB slice of the flat array) are averaged while the next range computes; the
    vectors + embeddings tail (96 MB) goes last.  Anything that touches the flat gradients on the main stream afterwards
    (the optimizer, clipping, an accumulating backward, the L2-from-init penalty) first waits for the communication
    stream.  (The reference all-reduces on every micro-step as well: it never enters no_sync(), train.py:263-266.)
    No buffer broadcast: DDP re-broadcasts `queue` (403 MB) from rank 0 on every forward; the replicas are identical
    by construction (same gathered keys, same order), so nothing is sent.

`KeyExchange` -- the key all-gather of MoCo._dequeue_and_enqueue (src/moco.py:272-287 via src/dist_utils.py:29-37)
    and the enqueue behind it.  The key encoder runs its last layer in two batch halves (czc_encoder_forward,
    split_batch): the first half's embeddings are gathered while the second half is still being encoded, the second
    half's while the InfoNCE head runs; the rows are put in rank-major order (rank r's B keys in batch order, as
    torch.cat(all_gather(...)) gives) so every replica enqueues the same rows in the same slots.  The enqueue waits for
    this step's logits (they read the columns it overwrites) and the main stream waits for the enqueue before it goes on."""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import lib as L
from .engine import FlatBertEncoder


def world(group=None):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


@torch.no_grad()
def average_(flat: torch.Tensor, group=None):
    """In-place mean over ranks of a flat gradient array."""
    if world(group) == 1:
        return flat
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
    elif flat.is_cuda:
        # gloo rehearsal of the multi-GPU path on a one-GPU box: stage through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host.div_(dist.get_world_size(group)))
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))
    return flat


@torch.no_grad()
def _all_gather_rows(out: torch.Tensor, x: torch.Tensor, group):
    """out [W * n, C] <- every rank's x [n, C], rank order (device tensors under gloo are staged through the host)."""
    if x.is_cuda and dist.get_backend(group) != "nccl":
        host = torch.empty(out.shape, dtype=x.dtype)
        dist.all_gather_into_tensor(host, x.detach().cpu().contiguous(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


class KeyExchange:
    """Rank-order all-gather of a replica's key embeddings in (up to) two batch ranges, off the main stream.

        kx = KeyExchange(group)
        n = kx.split_for(B, S)                       # 0: no split (odd batch, one sequence, unaligned rows)
        k = encoder_k.encode(..., split=(n, kx.split_event))
        kx.gather(k, n)                              # side stream: rows [0, n) as soon as the event fires, then the rest
        ... InfoNCE head on the main stream ...
        kx.enqueue(fn)                               # side stream: after the head, fn(gathered [W * B, C]); main waits

    Host tensors (the gloo tests) take the same path without streams."""

    def __init__(self, group=None):
        self.group = group
        self.world = world(group)
        self.stream = None
        self.split_event = self.keys_event = self.head_event = self.done_event = None
        self._bufs = {}
        self.gathered = None
        self.exposed = None        # a list: (start, stop) timing events around the main stream's wait for the enqueue

    def exposed_ms(self):
        """Main-stream time spent waiting for the key exchange + enqueue since `exposed` was set to a list (after a device
        synchronisation): what the exchange costs the step beyond what the InfoNCE head covers; 0 on one rank."""
        ev, self.exposed = self.exposed or [], [] if self.exposed is not None else None
        return sum(a.elapsed_time(b) for a, b in ev)

    def _events(self):
        if self.split_event is None:
            lib = L.load()
            evs = []
            for _ in range(4):
                e = C.c_void_p()
                L.check(lib.czc_event_create(C.byref(e)))
                evs.append(e)
            self.split_event, self.keys_event, self.head_event, self.done_event = evs
            self.stream = torch.cuda.Stream()

    def __deepcopy__(self, memo):
        return KeyExchange(self.group)   # events, stream and buffers belong to one object: a copy makes its own on first use

    def __del__(self):
        try:
            for e in (self.split_event, self.keys_event, self.head_event, self.done_event):
                if e is not None:
                    L.load().czc_event_destroy(e)
        except Exception:
            pass

    @staticmethod
    def split_for(batch: int, seq: int) -> int:
        """Sequences in the first range: half the batch (0 = no split: a single sequence, or token counts that would leave
        the second range's rows off the 16-token grid the row kernels were measured on)."""
        n = batch // 2
        return n if (n > 0 and (n * seq) % 16 == 0) else 0

    def _buf(self, name, rows, like):
        b = self._bufs.get(name)
        if b is None or b.shape != (rows, like.shape[1]) or b.device != like.device or b.dtype != like.dtype:
            b = torch.empty(rows, like.shape[1], dtype=like.dtype, device=like.device)
            self._bufs[name] = b
        return b

    @torch.no_grad()
    def _gather_ranges(self, k, n):
        """gathered [W * B, C], rank-major, from the ranges [0, n) and [n, B) of every rank's k"""
        W, B = self.world, k.shape[0]
        out = self._buf("out", W * B, k)
        if n == 0:
            return _all_gather_rows(out, k, self.group)
        view = out.view(W, B, k.shape[1])
        for name, lo, hi in (("lo", 0, n), ("hi", n, B)):
            if name == "hi" and k.is_cuda:
                L.check(L.load().czc_stream_wait_event(self.stream.cuda_stream, self.keys_event))
            part = _all_gather_rows(self._buf(name, W * (hi - lo), k), k[lo:hi], self.group)
            view[:, lo:hi].copy_(part.view(W, hi - lo, k.shape[1]))
        return out

    @torch.no_grad()
    def gather(self, k: torch.Tensor, n: int = 0):
        """Start the exchange of this rank's keys k [B, C]; call right after the key forward has been enqueued."""
        if not k.is_cuda:
            self.gathered = self._gather_ranges(k, n)
            return
        self._events()
        lib = L.load()
        L.check(lib.czc_event_record(self.keys_event, L.stream_ptr()))        # all of k is complete behind this point
        first = self.split_event if n else self.keys_event
        L.check(lib.czc_stream_wait_event(self.stream.cuda_stream, first))
        with torch.cuda.stream(self.stream):
            self.gathered = self._gather_ranges(k, n)

    @torch.no_grad()
    def enqueue(self, fn):
        """fn(gathered) is enqueued on the side stream behind everything the main stream has issued so far (this step's
        logits: they read the queue columns fn overwrites), and the main stream then waits for it: whatever reads the
        queue or its pointer next -- the next step's logits, a checkpoint, int(queue_ptr) -- sees the enqueued keys."""
        if self.stream is None:
            return fn(self.gathered)
        lib = L.load()
        L.check(lib.czc_event_record(self.head_event, L.stream_ptr()))
        L.check(lib.czc_stream_wait_event(self.stream.cuda_stream, self.head_event))
        with torch.cuda.stream(self.stream):
            fn(self.gathered)
            L.check(lib.czc_event_record(self.done_event, self.stream.cuda_stream))
        pair = _timing_pair() if self.exposed is not None else None
        if pair:
            pair[0].record()
        L.check(lib.czc_stream_wait_event(L.stream_ptr(), self.done_event))
        if pair:
            pair[1].record()
            self.exposed.append(pair)


def _timing_pair():
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


class DataParallel(nn.Module):
    """`.module` is the wrapped MoCo (same attribute the reference reads from its DDP wrapper).

    Two settings no 8-GPU run has tuned yet are read ONCE here from the environment, so the first such run can sweep
    them without a code change (bench.py echoes them in its line's `config`):
      CZC_DP_CHUNK_LAYERS=<n>   layers per backward range = per gradient all-reduce (default 3: 85 MB slices, 5 collectives)
      CZC_DP_PER_TILE=0|1|auto  GEMM launches beside a collective: 1 = one workgroup per output tile from the first range's
                                all-reduce to the end of the backward (czc_gemm_set_tile(16)), 0 = keep the persistent
                                launches, auto = 1 (the conservative choice until it has been measured)"""

    def __init__(self, module: nn.Module, optimizer=None, chunk_layers: int = 3, process_group=None):
        """process_group: the ranks that hold replicas (default: all; the students of a teacher/student run otherwise)."""
        super().__init__()
        self.module = module
        self._group = process_group
        self._world = world(process_group)
        env_chunk = os.environ.get("CZC_DP_CHUNK_LAYERS")
        if env_chunk:
            chunk_layers = max(1, int(env_chunk))
        mode = os.environ.get("CZC_DP_PER_TILE", "auto").lower()
        if mode not in ("0", "1", "auto"):
            raise ValueError(f"CZC_DP_PER_TILE={mode!r}: 0, 1 or auto")
        self.chunk_layers = chunk_layers
        self.per_tile_mode = mode
        self._use_per_tile = mode != "0"
        self.exposed = None        # a list: (start, stop) timing events around the main stream's waits for the all-reduces
        self._encoders = [m for m in module.modules() if isinstance(m, FlatBertEncoder)
                          and any(p.requires_grad for p in m._plist)]
        self._comm = None
        self._done = None          # newest event of the communication stream (it is in order: covers every earlier range)
        self._done_mat = None      # event behind the last WEIGHT-MATRIX range of the current backward
        self._per_tile = False
        for enc in self._encoders:
            enc._range_hooks = [self._reduce_range]
            enc._pre_grad_access = [self.wait]
            enc._post_backward = [self._backward_enqueued]
            enc.backward_chunk_layers = chunk_layers if self._world > 1 else 0
        if optimizer is not None:
            self.attach(optimizer)

    def attach(self, optimizer):
        """The optimizer waits for the matrix ranges before it updates the weight matrices and for the last slice (vectors +
        embeddings, 96 MB: the one all-reduce no backward work is left to cover) only before it updates those."""
        if self._world > 1:
            optimizer._pre_step.append(self.wait_matrices)
            optimizer._pre_tail.append(self.wait)
        else:
            optimizer._pre_step.append(self.wait)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def _reduce_range(self, enc, lo, hi):
        """gradient elements [lo, hi) of the flat array are final: average them across ranks on the side stream"""
        if self._world == 1 or hi <= lo:
            return
        g = enc.flat_grads(create=False)[lo:hi]
        if g.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream()
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(ready)
                average_(g, self._group)
                self._done = torch.cuda.Event()   # the stream is in order: the newest event covers every earlier range
                self._done.record()
                if hi <= enc._n_mat:
                    self._done_mat = self._done
            if self._use_per_tile and not self._per_tile:
                # The GEMMs' persistent launch sizes its grid to ALL the CUs; with RCCL's kernels holding some of them
                # the last workgroups would start a whole pass late.  One workgroup per output tile degrades gracefully
                # instead (include/czc.h, czc_gemm_set_tile) -- only for the launches that can run beside a collective:
                # from the first range's all-reduce to the end of this backward.
                L.check(L.load().czc_gemm_set_tile(16))
                self._per_tile = True
        else:
            average_(g, self._group)

    def _backward_enqueued(self, enc):
        if self._per_tile:
            L.check(L.load().czc_gemm_set_tile(0))
            self._per_tile = False

    def settings(self) -> dict:
        return {"chunk_layers": self.chunk_layers, "per_tile": self.per_tile_mode, "world": self._world}

    def exposed_ms(self):
        """Main-stream time spent waiting for gradient all-reduces since `exposed` was set to a list (after a device
        synchronisation): the part of the collectives the backward did not cover; 0 on one rank."""
        ev, self.exposed = self.exposed or [], [] if self.exposed is not None else None
        return sum(a.elapsed_time(b) for a, b in ev)

    def _wait_event(self, event):
        pair = _timing_pair() if self.exposed is not None else None
        if pair:
            pair[0].record()
        torch.cuda.current_stream().wait_event(event)
        if pair:
            pair[1].record()
            self.exposed.append(pair)

    def wait(self):
        """The main stream waits for every gradient all-reduce issued so far."""
        if self._done is not None:
            self._wait_event(self._done)
            self._done = None
        self._done_mat = None

    def wait_matrices(self):
        """The main stream waits for the all-reduces of the weight-matrix ranges (not for the vectors + embeddings slice)."""
        if self._done_mat is not None:
            self._wait_event(self._done_mat)
            self._done_mat = None
