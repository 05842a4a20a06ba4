"""`src.utils` for the MI355X path: optimiser/scheduler construction, checkpoint save/load, running
stats.  Reference: Ajchler/czech-contriever src/utils.py:19-291.

Kept: `set_optim(opt, model) -> (optimizer, scheduler)`, `WarmupLinearScheduler`, `CosineScheduler`,
`save(model, optimizer, scheduler, step, opt, dir_path, name)` writing
`<dir>/checkpoint/<name>/checkpoint.pth` = {"step","model","optimizer","scheduler","opt"} plus the
`latest` symlink, `load(model_class, dir_path, opt, reset_params)`, `WeightedAvgStats`, `mean_pooling`,
`init_logger`, `init_tb_logger`, `get_parameters`.
Changed (each is a defect of the reference as committed): CosineScheduler imports
math; checkpoints are read with torch.load(weights_only=True) plus an allow-list of the one non-tensor class the file
holds (the options Namespace under "opt") -- the reference's bare torch.load fails on torch >= 2.6 and a full unpickle
would run whatever a downloaded checkpoint contains; init_tb_logger's second argument is optional.  The optimiser is FusedAdamW: same hyper-parameters and
state_dict layout as torch.optim.AdamW, one HIP launch over the flat parameter array."""
from __future__ import annotations

import errno
import logging
import math
import os
import sys
from collections import defaultdict
from typing import Dict, Tuple, Union

import torch

from src import dist_utils

Number = Union[float, int]
logger = logging.getLogger(__name__)


def mean_pooling(token_embeddings, mask):
    """Masked mean over the sequence axis (the pooling of src/contriever.py:45-50, for callers that hold hidden states)."""
    keep = mask.bool().unsqueeze(-1)
    return torch.where(keep, token_embeddings, torch.zeros_like(token_embeddings)).sum(1) / mask.sum(1, keepdim=True)


def init_logger(args, stdout_only=False):
    """Root logging to stdout (+ <output_dir>/run.log), INFO on the main rank and WARN elsewhere (reference :25-38)."""
    dist_utils.barrier()
    sinks = [logging.StreamHandler(sys.stdout)]
    if not stdout_only:
        sinks.append(logging.FileHandler(filename=os.path.join(args.output_dir, "run.log")))
    level = logging.INFO if dist_utils.is_main() else logging.WARN
    logging.basicConfig(level=level, handlers=sinks, datefmt="%m/%d/%Y %H:%M:%S",
                        format="[%(asctime)s] {%(filename)s:%(lineno)d} %(levelname)s - %(message)s")
    return logger


def symlink_force(target, link_name):
    """`ln -sfn`: an existing link (or file) of that name is replaced."""
    if os.path.lexists(link_name):
        os.remove(link_name)
    os.symlink(target, link_name)


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def read_checkpoint(path, unsafe=False):
    """torch.load of a `checkpoint.pth` (reference src/utils.py:78, src/contriever.py:108) that executes nothing from the
    file: tensors, containers and numbers, plus the options object under "opt" (argparse.Namespace as src/options.py
    parses it; types.SimpleNamespace as tests and bench build it).  Anything else in the pickle is refused; a caller who
    trusts the file can opt in with unsafe=True (env CZC_UNSAFE_LOAD=1)."""
    import argparse
    import types
    if unsafe or os.environ.get("CZC_UNSAFE_LOAD") == "1":
        return torch.load(path, map_location="cpu", weights_only=False)
    with torch.serialization.safe_globals([argparse.Namespace, types.SimpleNamespace]):
        return torch.load(path, map_location="cpu", weights_only=True)


def save(model, optimizer, scheduler, step, opt, dir_path, name):
    root = os.path.join(dir_path, "checkpoint")
    target = os.path.join(root, name)
    os.makedirs(target, exist_ok=True)
    if getattr(_unwrap(model), "_ema_prefolded", None) is not None:
        # the optimizer's pass already applied the NEXT forward's momentum update to the key encoder (MoCo.allow_ema_fold):
        # a checkpoint now would hold key weights one EMA ahead of the reference's and a resumed run would apply it twice
        raise RuntimeError("utils.save after an optimizer step with the momentum update folded in: call "
                           "model.allow_ema_fold(False) before the optimizer step that precedes a checkpoint")
    payload = {
        "step": step,
        "model": _unwrap(model).state_dict(),
        "optimizer": optimizer.state_dict(),
        "scheduler": scheduler.state_dict(),
        "opt": opt,
    }
    torch.save(payload, os.path.join(target, "checkpoint.pth"))
    symlink_force(target, os.path.join(root, "latest"))
    if name != "lastlog":
        logger.info(f"Saving model to {target}")


def load(model_class, dir_path, opt, reset_params=False):
    epoch_path = os.readlink(dir_path) if os.path.islink(dir_path) else dir_path
    checkpoint_path = os.path.join(epoch_path, "checkpoint.pth")
    logger.info(f"loading checkpoint {checkpoint_path}")
    checkpoint = read_checkpoint(checkpoint_path)
    opt_checkpoint = checkpoint["opt"]
    model = model_class(opt_checkpoint)
    model.load_state_dict(checkpoint["model"], strict=True)
    model = model.cuda()
    step = checkpoint["step"]
    if reset_params:
        optimizer, scheduler = set_optim(opt, model)
    else:
        optimizer, scheduler = set_optim(opt_checkpoint, model)
        scheduler.load_state_dict(checkpoint["scheduler"])
        optimizer.load_state_dict(checkpoint["optimizer"])
    return model, optimizer, scheduler, opt_checkpoint, step


# ----------------------------------------------------------------------------------------------
# schedules
# ----------------------------------------------------------------------------------------------
class WarmupLinearScheduler(torch.optim.lr_scheduler.LambdaLR):
    """lambda(step): linear 0 -> (1-ratio) over `warmup`, then linear 1 -> ratio at `total`."""

    def __init__(self, optimizer, warmup, total, ratio, last_epoch=-1):
        self.warmup, self.total, self.ratio = warmup, total, ratio
        super().__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup:
            return (1 - self.ratio) * step / float(max(1, self.warmup))
        decay = (self.ratio - 1) * (step - self.warmup) / float(max(1.0, self.total - self.warmup))
        return max(0.0, 1.0 + decay)


class CosineScheduler(torch.optim.lr_scheduler.LambdaLR):
    def __init__(self, optimizer, warmup, total, ratio=0.1, last_epoch=-1):
        self.warmup, self.total, self.ratio = warmup, total, ratio
        super().__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup:
            return float(step) / self.warmup
        s = float(step - self.warmup) / (self.total - self.warmup)
        return self.ratio + (1.0 - self.ratio) * math.cos(0.5 * math.pi * s)


def set_optim(opt, model):
    if opt.optim != "adamw":
        raise NotImplementedError("optimizer class not implemented")
    from czech_contriever_amd.optim import FusedAdamW
    # --weight_decay_from_init: the reference builds AdamW WITHOUT a weight_decay argument (src/utils.py:139-142), so
    # torch's default decoupled decay of 0.01 applies on top of the L2-from-init penalty; kept as it is there
    wd = 0.01 if opt.weight_decay_from_init else opt.weight_decay
    optimizer = FusedAdamW(model, lr=opt.lr, betas=(opt.beta1, opt.beta2), eps=opt.eps, weight_decay=wd)
    core = model.module if hasattr(model, "module") else model
    if hasattr(core, "_momentum_update_key_encoder") and hasattr(core, "encoder_k"):
        optimizer.attach_ema(core)      # MoCo / MoCoDistill: see FusedAdamW.attach_ema (off until the loop allows it per step)
    kinds = {"linear": WarmupLinearScheduler, "cosine": CosineScheduler}
    if opt.scheduler not in kinds:
        raise ValueError
    scheduler = kinds[opt.scheduler](optimizer, warmup=opt.warmup_steps, total=opt.total_steps, ratio=opt.lr_min_ratio)
    return optimizer, scheduler


def get_parameters(net, verbose=False):
    n = sum(p.numel() for p in net.parameters())
    return "[Network] Total number of parameters : %.6f M" % (n / 1e6)


class WeightedAvgStats:
    """Weighted running means of named stats; `average_stats` reduces them across ranks (to rank 0)."""

    def __init__(self):
        self.reset()

    def update(self, vals: Dict[str, Tuple[Number, Number]]) -> None:
        for key, (value, weight) in vals.items():
            self.raw_stats[key] += value * weight
            self.total_weights[key] += weight

    @property
    def stats(self) -> Dict[str, float]:
        return {k: self.raw_stats[k] / self.total_weights[k] for k in self.raw_stats}

    @property
    def tuple_stats(self) -> Dict[str, Tuple[float, float]]:
        return {k: (self.raw_stats[k] / self.total_weights[k], self.total_weights[k]) for k in self.raw_stats}

    def reset(self) -> None:
        self.raw_stats: Dict[str, float] = defaultdict(float)
        self.total_weights: Dict[str, float] = defaultdict(float)

    @property
    def average_stats(self) -> Dict[str, float]:
        keys = sorted(self.raw_stats.keys())
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.broadcast_object_list(keys, src=0)
        out = {}
        for k in keys:
            v = self.raw_stats[k] / self.total_weights[k] if k in self.total_weights else 0.0
            out[k], _ = dist_utils.weighted_average(v, self.total_weights[k])
        return out


class WeightedAvgStatsDistill(WeightedAvgStats):
    """The same over the student ranks of a teacher/student run: keys come from global rank 1, sums go to it
    (reference src/utils.py:177-220)."""

    def __init__(self, process_group):
        super().__init__()
        self.process_group = process_group

    @property
    def average_stats(self) -> Dict[str, float]:
        keys = sorted(self.raw_stats.keys())
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 2:
            torch.distributed.broadcast_object_list(keys, src=1, group=self.process_group)
        out = {}
        for k in keys:
            v = self.raw_stats[k] / self.total_weights[k] if k in self.total_weights else 0.0
            out[k], _ = dist_utils.weighted_average_distill(v, self.total_weights[k], self.process_group)
        return out


def load_hf(object_class, model_name):
    """Local files only: there is no network on the target.  (reference :268-276 falls back to a download)"""
    if model_name == "czert":
        return object_class.from_pretrained("models/czert")
    return object_class.from_pretrained(model_name, local_files_only=True)


def init_tb_logger(output_dir, is_main=None):
    if is_main is None:
        is_main = dist_utils.is_main()
    try:
        from torch.utils import tensorboard
        return tensorboard.SummaryWriter(output_dir) if is_main else None
    except Exception:
        logger.warning("Tensorboard is not available.")
        return None
