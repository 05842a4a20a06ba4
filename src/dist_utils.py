"""`src.dist_utils` for the MI355X path: cross-GPU key exchange over RCCL (torch.distributed "nccl"
on ROCm) and the rank helpers the training loop uses.

Reference surface (src/dist_utils.py:7-148): gather, gather_nograd, varsize_gather_nograd,
get_rank, is_main, get_world_size, barrier, average_main, sum_main, weighted_average.
Deviations, both listed in DESIGN.md: gather_nograd(x, process_group=None) gathers from ALL ranks
of the group in rank order (the reference allocates world_size-1 slots, src/dist_utils.py:33, which
is only right for distill.py's sub-group), and it is ONE all_gather_into_tensor into a preallocated
[W*B, C] buffer instead of W tensors plus a cat."""
from __future__ import annotations

import torch
import torch.distributed as dist


def _on():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if _on() else 0


def is_main():
    return get_rank() == 0


def get_world_size(group=None):
    return dist.get_world_size(group) if _on() else 1


def barrier():
    if _on():
        dist.barrier()


class _AllGatherWithGrad(torch.autograd.Function):
    """all-gather whose backward hands each rank the summed gradient of its own slice
    (reference Gather: all_gather forward, all_reduce + slice backward, src/dist_utils.py:7-18)."""

    @staticmethod
    def forward(ctx, x):
        w = dist.get_world_size()
        out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous())
        ctx.n = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        w, r = dist.get_world_size(), dist.get_rank()
        g = g.contiguous()
        if g.is_cuda:
            mine = torch.empty((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(mine, g)  # == all_reduce(g)[rank slice], 1/W of the traffic
            return mine
        dist.all_reduce(g)
        return g[r * ctx.n:(r + 1) * ctx.n]


def gather(x: torch.Tensor):
    if not _on():
        return x
    return _AllGatherWithGrad.apply(x)


@torch.no_grad()
def gather_nograd(x: torch.Tensor, process_group=None, out: torch.Tensor = None):
    """[B, ...] per rank -> [W*B, ...], rank order.  `out` lets the caller reuse one buffer per step."""
    if not _on():
        return x
    w = dist.get_world_size(process_group)
    if out is None:
        out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if x.is_cuda and dist.get_backend(process_group) != "nccl":
        # gloo rehearsal on a one-GPU box: gather on the host
        host = torch.empty(out.shape, dtype=x.dtype)
        dist.all_gather_into_tensor(host, x.detach().cpu().contiguous(), group=process_group)
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, x.contiguous(), group=process_group)
    return out


@torch.no_grad()
def varsize_gather_nograd(x: torch.Tensor):
    if not _on():
        return x
    w = dist.get_world_size()
    size = torch.tensor([x.shape[0]], device=x.device, dtype=torch.int64)
    sizes = torch.empty(w, device=x.device, dtype=torch.int64)
    dist.all_gather_into_tensor(sizes, size)
    sizes = sizes.tolist()
    mx = max(sizes)
    padded = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    padded[: x.shape[0]] = x
    out = torch.empty((w * mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[i * mx: i * mx + n] for i, n in enumerate(sizes)], dim=0)


@torch.no_grad()
def get_varsize(x: torch.Tensor):
    if not _on():
        return [x.shape[0]]
    size = torch.tensor([x.shape[0]], device=x.device, dtype=torch.int64)
    sizes = torch.empty(dist.get_world_size(), device=x.device, dtype=torch.int64)
    dist.all_gather_into_tensor(sizes, size)
    return sizes


def sum_main(x):
    if _on() and dist.get_world_size() > 1:
        dist.reduce(x, 0, op=dist.ReduceOp.SUM)
    return x


def sum_main_distill(x, process_group):
    """Sum over the STUDENT ranks onto global rank 1, the students' main (reference :121-126; with a single student
    there is nothing to add)."""
    if _on() and dist.get_world_size() > 2:
        dist.reduce(x, 1, op=dist.ReduceOp.SUM, group=process_group)
    return x


def average_main(x):
    if _on() and dist.get_world_size() > 1:
        dist.reduce(x, 0, op=dist.ReduceOp.SUM)
        if is_main():
            x = x / dist.get_world_size()
    return x


def _stat_device():
    if _on() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def weighted_average(x, count):
    """Cross-rank weighted mean of a scalar stat, result valid on rank 0 (reference :139-148)."""
    if not _on():
        if isinstance(x, torch.Tensor):
            x = x.item()
        return x, count
    d = _stat_device()
    t = torch.tensor([float(x) * float(count), float(count)], dtype=torch.float64 if d.type == "cpu" else torch.float32, device=d)
    sum_main(t)
    return (t[0] / t[1]).item(), t[1].item()


def weighted_average_distill(x, count, process_group):
    """weighted_average over the student group, valid on global rank 1 (reference :128-137)."""
    if not _on():
        if isinstance(x, torch.Tensor):
            x = x.item()
        return x, count
    d = _stat_device()
    t = torch.tensor([float(x) * float(count), float(count)], dtype=torch.float64 if d.type == "cpu" else torch.float32, device=d)
    sum_main_distill(t, process_group)
    return (t[0] / t[1]).item(), t[1].item()
