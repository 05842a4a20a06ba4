#!/usr/bin/env python3
"""distill.py — teacher/student contrastive distillation on MI355X, the reference's command line.

    torchrun --nproc_per_node=N distill.py --retriever_model_id czert --random_init --train_data corpus.jsonl \\
        --per_gpu_batch_size 256 --target_batch_size <256*(N-1)> --queue_size 131072 --distill_weight 0.5 ...

What Ajchler/czech-contriever distill.py does (:225-419, :451-570): global rank 0 holds the TEACHER and never trains;
ranks 1..N-1 are STUDENTS, each a `MoCoDistill` replica (data parallel over the student group).  Per micro-step every
student sends its query token ids to the teacher, runs the MoCo step, receives the teacher's embeddings of its
queries and adds  DISTILL_LOSS_MULTIPLIER * mse(sim(student q), sim(teacher q))  weighted by --distill_weight.

On this path
  * the student step is libczc.so: `src.moco.MoCoDistill` (key all-gather over the student group), the similarity
    loss is `czc_distill_sim_loss`, gradients are averaged by czech_contriever_amd.parallel.DataParallel over the
    student group (RCCL);
  * the exchange is ONE max-length/stop all-reduce, one gather of token ids and one send/recv of embeddings per
    micro-step on the global group; the ids leave BEFORE the student's forward so the teacher encodes while the
    students train, the receive is posted after it (the reference brackets each of these with barriers, seven per
    step);
  * the teacher stops when the students say so (the reference's teacher loops forever and is killed with the job);
  * the reference's teacher, BAAI/bge-multilingual-gemma2 (3584-d, fetched by name, :506-507), cannot be downloaded
    here: `--teacher` names a LOCAL teacher instead — `synthetic:<dim>` (a fixed random token table, mean-pooled: a
    stand-in with the right shapes and traffic) or a directory with a `checkpoint.pth` of this repo (a BERT teacher
    on the same kernels; student ids are decoded to text and re-tokenised with the teacher's vocabulary, as the
    reference does for its teacher, :381-392);
  * one process (no torchrun) = one student with the teacher called in-process, for tests and single-GPU runs;
  * samplers split the data over the students only (the reference also counts the teacher rank, whose share of the
    corpus is then never read).  BEIR evaluation / eval_loss are outside the path and not built.
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

DISTILL_LOSS_MULTIPLIER = 100
INSTRUCTION = "Given a web search query, retrieve relevant passages that answer the query."

logger = logging.getLogger(__name__)


# ---- teachers (local stand-ins for the fetched model) --------------------------------------------------------------
class SyntheticTeacher:
    """mean over the non-pad tokens of a fixed N(0,1) table [vocab, dim]: deterministic, text-free, the reference
    teacher's output shape (3584 by default).  Host glue in torch — the teacher is not on the measured path."""

    def __init__(self, dim, vocab_size, pad_id, device):
        g = torch.Generator().manual_seed(20240517)
        self.table = torch.randn(vocab_size, dim, generator=g).to(device)
        self.dim, self.pad_id = dim, pad_id

    @torch.no_grad()
    def encode_ids(self, ids):
        mask = (ids != self.pad_id).to(self.table.dtype).unsqueeze(-1)
        e = self.table[ids.clamp(0, self.table.shape[0] - 1)] * mask
        return (e.sum(1) / mask.sum(1).clamp(min=1.0)).float()


class LocalBertTeacher:
    """A `checkpoint.pth` of this repo as the teacher (a larger / further-trained BERT), encoded by libczc.so."""

    def __init__(self, path, student_tokenizer, device):
        from src.contriever import load_retriever
        self.model, self.tokenizer, _ = load_retriever(path)
        self.model = self.model.eval().to(device)
        self.dim = self.model.config.hidden_size
        self.student_tokenizer = student_tokenizer
        self.device = device

    @torch.no_grad()
    def encode_ids(self, ids):
        texts = self.student_tokenizer.batch_decode(ids, skip_special_tokens=True)
        texts = [f"<instruct>{INSTRUCTION}\n<query>{t}" for t in texts]     # distill.py:51-52, 386-389
        enc = self.tokenizer(texts, max_length=self.model.config.max_position_embeddings, padding=True, truncation=True, return_tensors="pt")
        return self.model(input_ids=enc["input_ids"].to(self.device), attention_mask=enc["attention_mask"].to(self.device)).float()


def make_teacher(spec, student_tokenizer, vocab_size, pad_id, device):
    if spec.startswith("synthetic"):
        dim = int(spec.split(":")[1]) if ":" in spec else 3584
        return SyntheticTeacher(dim, vocab_size, pad_id, device)
    if os.path.isdir(spec):
        return LocalBertTeacher(spec, student_tokenizer, device)
    raise FileNotFoundError(f"--teacher {spec}: not `synthetic[:dim]` and not a local directory (models cannot be fetched here)")


def teacher_dim(spec):
    if spec.startswith("synthetic"):
        return int(spec.split(":")[1]) if ":" in spec else 3584
    ck = utils.read_checkpoint(os.path.join(spec, "checkpoint.pth"))
    from src.moco import resolve_model_config
    return resolve_model_config(ck["opt"].retriever_model_id)[0].hidden_size


# ---- the exchange --------------------------------------------------------------------------------------------------
def _dev(group):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


class StudentLink:
    """A student's side of the per-step exchange with global rank 0."""

    def __init__(self, group, control, batch, dim, pad_id):
        """group: all ranks, carries ids and embeddings (RCCL); control: the same ranks on gloo, carries the two-word
        header from host memory, so that agreeing on the padded width does not drain the GPU queue every step."""
        self.group, self.control, self.batch, self.dim, self.pad_id = group, control, batch, dim, pad_id
        self.dev = _dev(group)

    def send_ids(self, q_tokens, stop=False):
        """Before the forward: this step's query ids go to the teacher.  stop=True (no ids) tells everyone the run is
        over.  Returns False once the run is over."""
        head = torch.tensor([0 if stop else q_tokens.shape[1], int(stop)], dtype=torch.int64)
        dist.all_reduce(head, op=dist.ReduceOp.MAX, group=self.control)
        if int(head[1]):
            return False
        width = int(head[0])
        ids = q_tokens.to(self.dev)
        if ids.shape[1] < width:      # students crop to different lengths: pad to the longest (distill.py:247-250)
            ids = torch.cat([ids, torch.full((ids.shape[0], width - ids.shape[1]), self.pad_id, dtype=ids.dtype, device=self.dev)], 1)
        dist.gather(ids.contiguous(), None, dst=0, group=self.group)
        return True

    def recv_embeddings(self, device):
        """After the forward: the teacher's embeddings of the ids sent above, fp32 [batch, dim]."""
        buf = torch.empty(self.batch, self.dim, dtype=torch.float32, device=self.dev)
        dist.recv(buf, src=0, group=self.group)
        return buf.to(device)


def serve_teacher(teacher, group, control, batch, device):
    """Global rank 0: answer the students until they signal the end (reference :366-419, a `while True` there)."""
    dev = _dev(group)
    world = dist.get_world_size(group)
    served = 0
    while True:
        head = torch.zeros(2, dtype=torch.int64)
        dist.all_reduce(head, op=dist.ReduceOp.MAX, group=control)
        if int(head[1]):
            return served
        width = int(head[0])
        mine = torch.zeros(batch, width, dtype=torch.int64, device=dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.gather(mine, parts, dst=0, group=group)
        emb = teacher.encode_ids(torch.cat(parts[1:], 0).to(device)).to(dev)
        for r in range(1, world):
            dist.send(emb[(r - 1) * batch:r * batch].contiguous(), dst=r, group=group)
        served += 1


# ---- the student's loop --------------------------------------------------------------------------------------------
def distill_loss(student_embeddings, teacher_embeddings):
    """DISTILL_LOSS_MULTIPLIER * mse(sim(student), sim(teacher)) (reference :272-275) on czc_distill_sim_loss."""
    from czech_contriever_amd.ops import distill_sim_loss
    return distill_sim_loss(student_embeddings, teacher_embeddings) * DISTILL_LOSS_MULTIPLIER


def train_student(opt, model, optimizer, scheduler, step, link, local_teacher, student_group, n_students, student_rank):
    import train as train_script
    run_stats = utils.WeightedAvgStatsDistill(student_group)
    is_main = student_rank == 0
    tb_logger = utils.init_tb_logger(opt.output_dir, is_main)
    device = torch.device("cuda", torch.cuda.current_device())
    batches = train_script.make_batches(opt, model, student_rank, n_students)

    if opt.target_batch_size % (opt.per_gpu_batch_size * n_students) != 0:
        raise ValueError("target_batch_size must be divisible by per_gpu_batch_size * number of students")
    update_freq = opt.target_batch_size // (opt.per_gpu_batch_size * n_students)

    model.train()
    accumulate_steps = 0
    t_last, step_last = time.time(), step
    for batch, _ in batches:
        if step >= opt.total_steps:
            break
        accumulate_steps += 1
        batch = {k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        if link is not None:
            link.send_ids(batch["q_tokens"])

        train_loss, student_embeddings, iter_stats = model(**batch, process_group=student_group, stats_prefix="train")
        if link is not None:
            teacher_embeddings = link.recv_embeddings(device)
        else:
            teacher_embeddings = local_teacher.encode_ids(batch["q_tokens"])
        aux_loss = distill_loss(student_embeddings, teacher_embeddings)

        bsz = batch["q_tokens"].size(0)
        iter_stats["train/loss_contrastive"] = (train_loss.detach(), bsz)
        iter_stats["train/distill_loss"] = (aux_loss.detach(), bsz)
        iter_stats["train/loss"] = (train_loss.detach(), bsz)
        total = (1 - opt.distill_weight) * train_loss + opt.distill_weight * aux_loss
        total.backward()

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
            log += f" | lr: {lr:0.3g} | Memory: {torch.cuda.max_memory_allocated() // 1e9} GiB"
            now = time.time()
            if step > step_last:
                log += f" | pairs/s: {(step - step_last) * opt.target_batch_size / (now - t_last):.0f}"
            t_last, step_last = now, step
            if tb_logger:
                tb_logger.add_scalar("train/lr", lr, step)
            if is_main:
                logger.info(log)
            run_stats.reset()

        if opt.clip_gradients:
            if opt.max_grad_value is not None:
                optimizer.clip_grad_value_(opt.max_grad_value)
            elif opt.max_grad_norm is not None:
                optimizer.clip_grad_norm_(opt.max_grad_norm)
        core = model.module if hasattr(model, "module") else model
        nxt = step + 1       # (as train.py: no folded momentum update in front of a checkpoint or the end of the run)
        core.allow_ema_fold(nxt < opt.total_steps and nxt % opt.eval_freq != 0 and nxt % opt.save_freq != 0)
        optimizer.step()
        scheduler.step()
        model.zero_grad()
        step += 1

        if is_main and step % opt.eval_freq == 0 and step > 0:
            utils.save(model, optimizer, scheduler, step, opt, opt.save_dir, "lastlog")
        if is_main and step % opt.save_freq == 0 and step > 0:
            utils.save(model, optimizer, scheduler, step, opt, opt.save_dir, f"step-{step}")
    if link is not None:
        link.send_ids(None, stop=True)
    return step


def build_student(opt, model_class):
    latest = os.path.join(opt.output_dir, "checkpoint", "latest")
    if os.path.isdir(latest) or os.path.islink(latest):
        model, optimizer, scheduler, _, step = utils.load(model_class, latest, opt, reset_params=False)
        logger.info(f"Model loaded from {opt.output_dir}")
    elif opt.model_path == "none":
        model = model_class(opt).cuda()
        optimizer, scheduler = utils.set_optim(opt, model)
        step = 0
    else:
        model, optimizer, scheduler, _, step = utils.load(model_class, opt.model_path, opt,
                                                          reset_params=not opt.continue_training)
        if not opt.continue_training:
            step = 0
        logger.info(f"Model loaded from {opt.model_path}")
    return model, optimizer, scheduler, step


def get_options():
    options = Options()
    options.parser.add_argument("--teacher", type=str, default="synthetic:3584",
                                help="LOCAL teacher: synthetic[:dim] or a directory with this repo's checkpoint.pth "
                                     "(the reference fetches BAAI/bge-multilingual-gemma2 by name)")
    return options


def main(argv=None):
    options = get_options()
    opt = options.parse(argv)
    torch.manual_seed(opt.seed)
    if not torch.cuda.is_available():
        raise SystemExit("distill.py needs an MI355X (there is no CPU path)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if opt.contrastive_mode != "moco":
        raise ValueError(f"contrastive mode: {opt.contrastive_mode} is not built for distillation (MoCoDistill only)")
    model_class = moco.MoCoDistill
    os.environ["TOKENIZERS_PARALLELISM"] = "false"

    global_group = control_group = student_group = None
    if world > 1:
        dist.init_process_group(backend=os.environ.get("CZC_DIST_BACKEND", "nccl"),
                                **({"device_id": device} if os.environ.get("CZC_DIST_BACKEND", "nccl") == "nccl" else {}))
        global_group = dist.new_group(list(range(world)))
        control_group = dist.new_group(list(range(world)), backend="gloo")
        student_group = dist.new_group(list(range(1, world)))     # every rank must take part in the creation
    rank = dist.get_rank() if world > 1 else 0
    os.makedirs(opt.output_dir, exist_ok=True)
    os.makedirs(opt.save_dir, exist_ok=True)
    if rank == (1 if world > 1 else 0):
        options.print_options(opt)
    utils.init_logger(opt)

    from src.moco import resolve_model_config
    cfg, student_tokenizer = resolve_model_config(opt.retriever_model_id)
    pad_id = student_tokenizer.pad_token_id if student_tokenizer is not None and student_tokenizer.pad_token_id is not None else 0

    if world > 1 and rank == 0:
        teacher = make_teacher(opt.teacher, student_tokenizer, cfg.vocab_size, pad_id, device)
        logger.warning(f"rank 0: teacher {opt.teacher} ({teacher.dim}-d)")
        served = serve_teacher(teacher, global_group, control_group, opt.per_gpu_batch_size, device)
        logger.warning(f"teacher served {served} micro-steps")
        dist.destroy_process_group()
        return

    model, optimizer, scheduler, step = build_student(opt, model_class)
    logger.info(utils.get_parameters(model))
    if step > 0:   # resumed: continue the counter-based dropout streams instead of replaying the masks of steps 0..n (as train.py)
        model.set_dropout_position(step * max(1, opt.target_batch_size // (opt.per_gpu_batch_size * max(1, world - 1))))
    if world > 1:
        from czech_contriever_amd.parallel import DataParallel
        model = DataParallel(model, optimizer, process_group=student_group)
        link = StudentLink(global_group, control_group, opt.per_gpu_batch_size, teacher_dim(opt.teacher), pad_id)
        local_teacher = None
        n_students, student_rank = world - 1, rank - 1
    else:
        link = None
        local_teacher = make_teacher(opt.teacher, student_tokenizer, cfg.vocab_size, pad_id, device)
        n_students, student_rank = 1, 0

    logger.info("Start training")
    step = train_student(opt, model, optimizer, scheduler, step, link, local_teacher, student_group, n_students, student_rank)
    if student_rank == 0:
        utils.save(model, optimizer, scheduler, step, opt, opt.save_dir, f"step-{step}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
