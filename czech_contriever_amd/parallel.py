"""Data parallelism for the MoCo path: one process per GPU, RCCL over xGMI (torch.distributed "nccl").

Replaces torch DDP at train.py:483-488 of the reference.  DDP cannot be used here: parameter
gradients are written by libczc straight into one flat fp32 array, not through autograd hooks.
What this wrapper does instead, per micro-step:
  * gradient mean all-reduce of the flat gradient array in 5 collectives on a side stream: the encoder
    backward is issued in ranges of 3 layers (top first); as soon as a range has been enqueued its weight-matrix
    gradients (one contiguous 85 MB slice of the flat array) are averaged while the next range computes; the
    vectors + embeddings tail (96 MB) goes last.  The optimizer waits on the last event.  (The reference all-reduces on every micro-step as well:
    it never enters no_sync(), train.py:263-266.)
  * no buffer broadcast: DDP re-broadcasts `queue` (403 MB) from rank 0 on every forward; the
    replicas are identical by construction (same gathered keys, same order), so nothing is sent.
The key all-gather is in src.moco / src.dist_utils."""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn

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


class DataParallel(nn.Module):
    """`.module` is the wrapped MoCo (same attribute the reference reads from its DDP wrapper)."""

    def __init__(self, module: nn.Module, optimizer=None, chunk_layers: int = 3, process_group=None):
        """process_group: the ranks that hold replicas (default: all; the students of a teacher/student run otherwise)."""
        super().__init__()
        self.module = module
        self._group = process_group
        self._world = world(process_group)
        self._encoders = [m for m in module.modules() if isinstance(m, FlatBertEncoder)
                          and any(p.requires_grad for p in m._plist)]
        self._comm = None
        self._done = None
        for enc in self._encoders:
            enc._range_hooks = [self._reduce_range]
            enc.backward_chunk_layers = chunk_layers if self._world > 1 else 0
        if self._world > 1 and torch.cuda.is_available():
            # The GEMMs' persistent launch sizes its grid to ALL the CUs; with RCCL's kernels holding some of them
            # during the overlapped gradient all-reduce the last workgroups would start a whole pass late.  One
            # workgroup per output tile degrades gracefully instead (include/czc.h, czc_gemm_set_tile).
            from . import lib as L
            L.check(L.load().czc_gemm_set_tile(16))
        if optimizer is not None:
            self.attach(optimizer)

    def attach(self, optimizer):
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
                self._done = torch.cuda.Event()
                self._done.record()
        else:
            average_(g, self._group)

    def wait(self):
        if self._done is not None:
            torch.cuda.current_stream().wait_event(self._done)
            self._done = None
