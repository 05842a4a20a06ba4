"""`src.slurm.init_distributed_mode` for the MI355X path (reference src/slurm.py:43-113).

Fills `params.{local_rank, global_rank, world_size, is_distributed, is_main}` and binds the process to its GPU.
torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK) and plain single-process runs are built; under a bare SLURM
allocation the ranks are read from SLURM_PROCID / SLURM_NTASKS / SLURM_LOCALID, but the master address must already be
in MASTER_ADDR / MASTER_PORT (the reference shells out to `scontrol` for it, :65-71 — cluster control plane, not
built).  Process group backend: "nccl", which is RCCL on ROCm."""
import os

import torch


def init_distributed_mode(params):
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        params.global_rank = int(os.environ["RANK"])
        params.world_size = int(os.environ["WORLD_SIZE"])
        params.local_rank = int(os.environ.get("LOCAL_RANK", getattr(params, "local_rank", 0) or 0))
    elif "SLURM_JOB_ID" in os.environ and "SLURM_PROCID" in os.environ:
        params.global_rank = int(os.environ["SLURM_PROCID"])
        params.world_size = int(os.environ["SLURM_NTASKS"])
        params.local_rank = int(os.environ["SLURM_LOCALID"])
        if params.world_size > 1 and "MASTER_ADDR" not in os.environ:
            raise RuntimeError("multi-task SLURM run: export MASTER_ADDR / MASTER_PORT (scontrol lookup is not built)")
        os.environ.setdefault("RANK", str(params.global_rank))
        os.environ.setdefault("WORLD_SIZE", str(params.world_size))
    else:
        params.global_rank, params.world_size, params.local_rank = 0, 1, 0
    params.is_distributed = params.world_size > 1
    params.is_main = params.global_rank == 0
    if torch.cuda.is_available():
        torch.cuda.set_device(params.local_rank)
    if params.is_distributed and not torch.distributed.is_initialized():
        torch.distributed.init_process_group(backend="nccl", init_method="env://", world_size=params.world_size,
                                             rank=params.global_rank)
