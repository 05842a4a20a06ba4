"""`src.options` for the MI355X path: the reference's command line, flag for flag.

Every flag name, type and default of Ajchler/czech-contriever src/options.py:16-244 is kept (the
checkpoint pickles the parsed Namespace under "opt", so names are part of the file format).
The flags are declared as a table; `Options().parse()` uses parse_known_args like the reference
(:263-266), and `print_options` writes `<output_dir>/opt.txt` in the same "name: value [default]"
form (:246-261)."""
from __future__ import annotations

import argparse
import os

_S, _I, _F = str, int, float
FLAG = lambda name, **kw: (name, kw)
ON = dict(action="store_true")
ON_F = dict(action="store_true", default=False)
LIST = dict(nargs="+", default=[])

# (flag, argparse kwargs).  Groups follow the reference's order.
FLAGS = [
    # --- basic: where things live, what data to read ---
    FLAG("output_dir", type=_S, default="./checkpoint/my_experiments", help="models are saved here"),
    FLAG("save_dir", type=_S, default="./checkpoint/my_experiments", help="models are saved here"),
    FLAG("train_data", help="jsonl files with the training texts", **LIST),
    FLAG("offsets_file", type=_S, default=None, help="byte offsets of the lines of the training data"),
    FLAG("cumsums_file", type=_S, default="data/line-offsets.pkl", help="token cumsums of the training lines"),
    FLAG("valid_data", help="jsonl files held out of the training data", **LIST),
    FLAG("val_data_ratio", type=_F, default=0.05, help="share of the training data used for validation"),
    FLAG("eval_data", help="evaluation data (fine-tuning only)", **LIST),
    FLAG("eval_datasets", help="BEIR-format evaluation sets", **LIST),
    FLAG("eval_datasets_dir", type=_S, default="./", help="where the evaluation sets are stored"),
    FLAG("orig_sampling", help="sample documents as contriever does", **ON_F),
    FLAG("model_path", type=_S, default="none", help="path for retraining"),
    FLAG("continue_training", **ON),
    FLAG("num_workers", type=_I, default=5),
    FLAG("num_workers_valid", type=_I, default=1),
    FLAG("chunk_length", type=_I, default=256),
    FLAG("loading_mode", type=_S, default="split"),
    FLAG("data_preprocessed", **ON_F),
    FLAG("lower_case", help="evaluate after lowercasing", **ON),
    FLAG("normalize_text", **ON_F),
    FLAG("sampling_coefficient", type=_F, default=0.0, help="dataset sampling coefficient (0 = uniform)"),
    FLAG("augmentation", type=_S, default="none"),
    FLAG("prob_augmentation", type=_F, default=0.0),
    FLAG("dropout", type=_F, default=0.1),
    FLAG("rho", type=_F, default=0.05),
    # --- contrastive objective ---
    FLAG("contrastive_mode", type=_S, default="moco"),
    FLAG("queue_size", type=_I, default=65536),
    FLAG("temperature", type=_F, default=1.0),
    FLAG("momentum", type=_F, default=0.999),
    FLAG("distill_weight", type=_F, default=0.5, help="distillation loss weight (contrastive weight is 1 - this)"),
    FLAG("moco_train_mode_encoder_k", **ON),
    FLAG("eval_normalize_text", **ON),
    FLAG("norm_query", **ON),
    FLAG("norm_doc", **ON),
    FLAG("projection_size", type=_I, default=768),
    FLAG("ratio_min", type=_F, default=0.1),
    FLAG("ratio_max", type=_F, default=0.5),
    FLAG("score_function", type=_S, default="dot"),
    FLAG("retriever_model_id", type=_S, default="czert"),
    FLAG("teacher_model_id", type=_S, default="facebook/contriever"),
    FLAG("translator_model_id", type=_S, default="Helsinki-NLP/opus-mt-cs-en"),
    FLAG("pooling", type=_S, default="average"),
    FLAG("random_init", help="init model with random weights", **ON),
    # --- batch sizes and schedule length ---
    FLAG("target_batch_size", type=_I, default=2048, help="batch size combined across all GPUs"),
    FLAG("per_gpu_batch_size", type=_I, default=64, help="training batch size per GPU"),
    FLAG("per_gpu_eval_batch_size", type=_I, default=256, help="evaluation batch size per GPU"),
    FLAG("total_steps", type=_I, default=1000),
    FLAG("warmup_steps", type=_I, default=-1),
    FLAG("local_rank", type=_I, default=-1, help="for distributed training: local_rank"),
    FLAG("main_port", type=_I, default=10001, help="master port (multi-node SLURM jobs)"),
    FLAG("sge", help="claim GPUs in a SGE environment", **ON_F),
    FLAG("seed", type=_I, default=0, help="random seed for initialization"),
    # --- optimisation ---
    FLAG("optim", type=_S, default="adamw"),
    FLAG("scheduler", type=_S, default="linear"),
    FLAG("lr", type=_F, default=1e-4, help="learning rate"),
    FLAG("lr_min_ratio", type=_F, default=0.0, help="final learning rate as a ratio of lr"),
    FLAG("weight_decay_from_init", help="penalise the L2 distance from the initial weights, not from 0", **ON_F),
    FLAG("weight_decay", type=_F, default=0.01, help="weight decay"),
    FLAG("beta1", type=_F, default=0.9, help="beta1"),
    FLAG("beta2", type=_F, default=0.98, help="beta2"),
    FLAG("eps", type=_F, default=1e-6, help="eps"),
    FLAG("clip_gradients", help="enable gradient clipping", **ON_F),
    FLAG("max_grad_norm", type=_F, default=None, help="max grad norm for clipping"),
    FLAG("max_grad_value", type=_F, default=None, help="max grad value for clipping"),
    FLAG("log_freq", type=_I, default=100, help="log train stats every <log_freq> steps"),
    FLAG("eval_freq", type=_I, default=1000, help="evaluate every <eval_freq> steps"),
    FLAG("save_freq", type=_I, default=50000),
    FLAG("maxload", type=_I, default=None),
    FLAG("label_smoothing", type=_F, default=0.0),
    # --- fine-tuning ---
    FLAG("negative_ctxs", type=_I, default=1),
    FLAG("negative_hard_min_idx", type=_I, default=0),
    FLAG("negative_hard_ratio", type=_F, default=0.0),
]


class Options:
    def __init__(self):
        self.parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        self.initialize()

    def initialize(self):
        for name, kw in FLAGS:
            self.parser.add_argument("--" + name, **kw)

    def print_options(self, opt):
        lines = []
        for key, value in sorted(vars(opt).items()):
            default = self.parser.get_default(key)
            note = "" if value == default else "\t[default: %s]" % str(default)
            lines.append(f"{str(key):>40}: {str(value):<40}{note}\n")
        message = "".join(lines)
        print(message, flush=True)
        os.makedirs(os.path.join(opt.output_dir, "models"), exist_ok=True)
        with open(os.path.join(opt.output_dir, "opt.txt"), "wt") as f:
            f.write(message + "\n")

    def parse(self, args=None):
        opt, _unknown = self.parser.parse_known_args(args)
        return opt
