#!/usr/bin/env python3
"""train.py — MoCo contrastive pre-training on MI355X, same command line as the reference.

    torchrun --nproc_per_node=N train.py --retriever_model_id czert --random_init \\
        --chunk_length 256 --per_gpu_batch_size 256 --target_batch_size 2048 --queue_size 131072 \\
        --temperature 0.05 --momentum 0.9995 --lr 1e-5 --total_steps 500000 ...

Loop skeleton of Ajchler/czech-contriever train.py:246-327 (accumulate `update_freq` micro-steps,
log every log_freq, optional clipping, optimizer/scheduler step, zero_grad, periodic save) on the
libczc.so path.  Differences from the file as committed upstream (SURVEY.md "five facts" #5):
  * no clearml import at module load; TensorBoard is optional (utils.init_tb_logger)
  * torch DDP is replaced by czech_contriever_amd.parallel.DataParallel (flat-gradient all-reduce
    over RCCL, no per-forward 403 MB queue broadcast)
  * the batch is moved with non_blocking copies; stats stay on the device until log_freq
  * per-parameter grad norms (197 host syncs) are replaced by ONE squared-norm reduction
  * data: `--train_data` goes through src/data.py (the reference's load_data / Dataset / Collator surface and
    sampling order) and a torch DataLoader; with no `--train_data` the loop runs on synthetic token ids
    (seed 1234 + 1000*rank + step).
  * BEIR / validation-set evaluation (eval_model, eval_loss) are outside the hot path and not built.
"""
import logging
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from src import dist_utils, inbatch, moco, utils
from src.options import Options

logger = logging.getLogger(__name__)


class SyntheticPairs:
    """Endless (q, k) token-id batches of the shape the reference's Collator produces (src/data.py:531-573)."""

    def __init__(self, opt, rank, vocab_size):
        self.B, self.S, self.rank, self.V = opt.per_gpu_batch_size, opt.chunk_length, rank, vocab_size
        self.step = 0

    def __iter__(self):
        return self

    def __next__(self):
        g = torch.Generator().manual_seed(1234 + 1000 * self.rank + self.step)
        self.step += 1
        lo = min(1000, self.V // 2)
        q = torch.randint(lo, self.V, (self.B, self.S), generator=g, dtype=torch.int64)
        k = torch.randint(lo, self.V, (self.B, self.S), generator=g, dtype=torch.int64)
        m = torch.ones(self.B, self.S, dtype=torch.bool)
        return {"q_tokens": q, "q_mask": m, "k_tokens": k, "k_mask": m.clone()}, None


class _Epochs:
    """The reference's epoch loop (train.py:248-256): a DataLoader over src.data's dataset, reshuffled per epoch."""

    def __init__(self, loader, sampler):
        self.loader, self.sampler, self.epoch = loader, sampler, 0

    def __iter__(self):
        while True:
            self.epoch += 1
            if hasattr(self.sampler, "set_epoch"):
                self.sampler.set_epoch(self.epoch)
            for item in self.loader:
                yield item


def make_batches(opt, model, rank, world):
    core = model.module if hasattr(model, "module") else model
    if opt.train_data:
        # reference train.py:183-206: load_data -> RandomSampler / DistributedSampler -> DataLoader(collate_fn=Collator)
        from src import data
        if core.tokenizer is None:
            raise RuntimeError("--train_data needs a local tokenizer (vocab.txt next to the model's config.json)")
        train_dataset, _ = data.load_data(opt, core.tokenizer, is_main=dist_utils.is_main())
        if world > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(train_dataset, num_replicas=world, rank=rank, shuffle=True)
        else:
            sampler = torch.utils.data.RandomSampler(train_dataset)
        loader = torch.utils.data.DataLoader(train_dataset, sampler=sampler, batch_size=opt.per_gpu_batch_size, drop_last=True,
                                             num_workers=getattr(opt, "num_workers", 0), collate_fn=data.Collator(opt=opt))
        return _Epochs(loader, sampler)
    return SyntheticPairs(opt, rank, core.get_encoder().config.vocab_size)


def train(opt, model, optimizer, scheduler, step):
    run_stats = utils.WeightedAvgStats()
    tb_logger = utils.init_tb_logger(opt.output_dir)
    rank, world = dist_utils.get_rank(), dist_utils.get_world_size()
    device = torch.device("cuda", torch.cuda.current_device())
    batches = make_batches(opt, model, rank, world)

    if opt.target_batch_size % (opt.per_gpu_batch_size * world) != 0:
        raise ValueError("target_batch_size must be divisible by per_gpu_batch_size * dist.get_world_size()")
    update_freq = opt.target_batch_size // (opt.per_gpu_batch_size * world)

    model.train()
    accumulate_steps = 0
    t_last, step_last = time.time(), step
    for batch, _ in batches:
        if step >= opt.total_steps:
            break
        accumulate_steps += 1
        batch = {k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        train_loss, iter_stats = model(**batch, stats_prefix="train")
        train_loss.backward()

        if accumulate_steps % update_freq != 0:
            continue
        run_stats.update(iter_stats)
        if step % opt.log_freq == 0:
            log = f"{step} / {opt.total_steps}"
            for k, v in sorted(run_stats.average_stats.items()):
                log += f" | {k}: {v:.3f}"
                if tb_logger:
                    tb_logger.add_scalar(k, v, step)
            lr = scheduler.get_last_lr()[0]
            log += f" | lr: {lr:0.3g}"
            log += f" | Memory: {torch.cuda.max_memory_allocated() // 1e9} GiB"
            enc = (model.module if hasattr(model, "module") else model).get_encoder()
            g = enc.flat_grads(create=False)
            if g is not None:
                if hasattr(model, "wait"):
                    model.wait()
                gnorm = float(torch.linalg.vector_norm(g))
                log += f" | grad: {gnorm:.3g}"
                if tb_logger:
                    tb_logger.add_scalar("train/global_grad", gnorm, step)
            now = time.time()
            if step > step_last:
                pairs = (step - step_last) * opt.target_batch_size / (now - t_last)
                log += f" | pairs/s: {pairs:.0f}"
            t_last, step_last = now, step
            if tb_logger:
                tb_logger.add_scalar("train/lr", lr, step)
            logger.info(log)
            run_stats.reset()

        if opt.clip_gradients:
            if opt.max_grad_value is not None:
                optimizer.clip_grad_value_(opt.max_grad_value)
            elif opt.max_grad_norm is not None:
                optimizer.clip_grad_norm_(opt.max_grad_norm)

        # the next forward's momentum update rides in the optimizer's pass unless a checkpoint is written first (MoCo.allow_ema_fold)
        core = model.module if hasattr(model, "module") else model
        if hasattr(core, "allow_ema_fold"):
            nxt = step + 1
            core.allow_ema_fold(nxt < opt.total_steps and nxt % opt.eval_freq != 0 and nxt % opt.save_freq != 0)
        optimizer.step()
        scheduler.step()
        model.zero_grad()
        step += 1

        if dist_utils.is_main() and step % opt.eval_freq == 0 and step > 0:
            utils.save(model, optimizer, scheduler, step, opt, opt.save_dir, "lastlog")
        if dist_utils.is_main() and step % opt.save_freq == 0 and step > 0:
            utils.save(model, optimizer, scheduler, step, opt, opt.save_dir, f"step-{step}")
    return step


def main(argv=None):
    opt = Options().parse(argv)
    torch.manual_seed(opt.seed)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X (there is no CPU path: the product fails loudly instead)")
    torch.cuda.set_device(local_rank)
    if world > 1 or "RANK" in os.environ:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    directory_exists = os.path.isdir(os.path.join(opt.output_dir, "checkpoint", "latest")) or os.path.islink(
        os.path.join(opt.output_dir, "checkpoint", "latest"))
    dist_utils.barrier()
    os.makedirs(opt.output_dir, exist_ok=True)
    if dist_utils.is_main():
        Options().print_options(opt)
    dist_utils.barrier()
    utils.init_logger(opt)
    os.environ["TOKENIZERS_PARALLELISM"] = "false"

    if opt.contrastive_mode == "moco":        # reference train.py:449-454
        model_class = moco.MoCo
    elif opt.contrastive_mode == "inbatch":
        model_class = inbatch.InBatch
    else:
        raise ValueError(f"contrastive_mode: {opt.contrastive_mode} not recognised")

    if not directory_exists and opt.model_path == "none":
        model = model_class(opt).cuda()
        optimizer, scheduler = utils.set_optim(opt, model)
        step = 0
    elif directory_exists:
        model_path = os.path.join(opt.output_dir, "checkpoint", "latest")
        model, optimizer, scheduler, opt_checkpoint, step = utils.load(model_class, model_path, opt, reset_params=False)
        logger.info(f"Model loaded from {opt.output_dir}")
    else:
        model, optimizer, scheduler, opt_checkpoint, step = utils.load(
            model_class, opt.model_path, opt, reset_params=False if opt.continue_training else True)
        if not opt.continue_training:
            step = 0
        logger.info(f"Model loaded from {opt.model_path}")
    logger.info(utils.get_parameters(model))
    if step > 0:   # resumed: continue the counter-based dropout streams instead of replaying the masks of steps 0..n
        world = dist.get_world_size() if dist.is_initialized() else 1
        model.set_dropout_position(step * max(1, opt.target_batch_size // (opt.per_gpu_batch_size * world)))

    if dist.is_initialized():
        from czech_contriever_amd.parallel import DataParallel
        model = DataParallel(model, optimizer)
        dist.barrier()

    logger.info("Start training")
    step = train(opt, model, optimizer, scheduler, step)
    if dist_utils.is_main():
        utils.save(model, optimizer, scheduler, step, opt, opt.save_dir, f"step-{step}")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
