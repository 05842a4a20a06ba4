"""FusedAdamW: torch.optim.AdamW's interface and state_dict layout over the flat parameter array.

`step()` is one czc_adamw_step launch over the query encoder's flat fp32 parameters / gradients /
moments (and it refreshes the bf16 weight mirror in the same pass).  `param_groups` and `state` keep
the per-parameter view torch's optimizers expose, so LambdaLR schedulers and checkpoints behave as
with torch.optim.AdamW (reference: utils.set_optim, src/utils.py:137-150).  Parameters that never
receive a gradient (the frozen key encoder, which the reference also hands to AdamW) are skipped."""
from __future__ import annotations

import torch

from . import lib as L
from .engine import FlatBertEncoder, _numel

F32 = torch.float32


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        root = model.module if hasattr(model, "module") else model
        params = list(root.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._encoders = [m for m in root.modules() if isinstance(m, FlatBertEncoder)
                          and any(p.requires_grad for p in m._plist)]
        if not self._encoders:
            raise RuntimeError("FusedAdamW: the model has no trainable FlatBertEncoder")
        owned = {id(p) for e in self._encoders for p in e._plist}
        loose = [p for p in params if p.requires_grad and id(p) not in owned]
        if loose:
            raise RuntimeError("FusedAdamW only updates encoder parameters stored in the flat array")
        self._m = {}
        self._v = {}
        self._steps = 0
        self.grad_scale = 1.0          # e.g. 1/update_freq is NOT applied by the reference; kept at 1
        self.clip_coef = None          # device scalar set by clip_grad_norm_()
        self._pre_step = []            # callables run before the update (gradient all-reduce wait)
        self._pre_tail = []            # callables run between the update of the matrix region and that of the vectors +
                                       # embeddings (data parallel: the last all-reduce, 96 MB, finishes under the first update)
        self._ema_owner = None         # weakref to the MoCo module whose key encoder's EMA may ride in this pass (attach_ema)

    def attach_ema(self, moco_module):
        """`moco_module` (src.moco.MoCo): its momentum EMA -- applied before EVERY key forward, reference src/moco.py:314 --
        reads the query weights this optimizer has just written.  While the module's `ema_fold` is set (the training loop
        sets it for steps after which nothing but the next training forward looks at the key encoder), step() applies that
        forward's EMA in the AdamW pass itself (czc_adamw_ema_step: bit-identical, 12 B per parameter less traffic, one launch
        fewer) and tells the module, which then skips it once."""
        import weakref
        self._ema_owner = weakref.ref(moco_module)

    def _moments(self, enc):
        k = id(enc)
        if k not in self._m or self._m[k].device != enc.flat_params.device:
            self._m[k] = torch.zeros_like(enc.flat_params)
            self._v[k] = torch.zeros_like(enc.flat_params)
            for (name, shape, off), p in zip(enc._table, enc._plist):
                if p.requires_grad:
                    n = _numel(shape)
                    self.state[p] = {"step": torch.tensor(float(self._steps)),
                                     "exp_avg": self._m[k][off:off + n].view(shape),
                                     "exp_avg_sq": self._v[k][off:off + n].view(shape)}
        return self._m[k], self._v[k]

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        split = bool(self._pre_tail) and self.clip_coef is None      # (a clip coefficient was computed from ALL gradients)
        for fn in (self._pre_tail if not split else []) + self._pre_step:
            fn()
        g0 = self.param_groups[0]
        self._steps += 1
        for enc in self._encoders:
            grads = enc.flat_grads(create=False)
            if grads is None or all(p.grad is None for p in enc._plist):
                continue
            m, v = self._moments(enc)
            b1, b2 = g0["betas"]
            mirror = enc.mirror()
            n, n_mat, esz = enc.flat_params.numel(), mirror.numel(), 4
            owner = self._ema_owner() if self._ema_owner is not None else None
            ek = None
            if owner is not None and getattr(owner, "ema_fold", False) and owner.encoder_q is enc:
                ek = owner.encoder_k
                if ek.flat_params.numel() != n or ek.flat_params.device != enc.flat_params.device:
                    ek = None
            k_mirror = ek.mirror() if ek is not None else None

            def launch(lo, hi, n_bf16):
                args = (enc.flat_params.data_ptr() + lo * esz, grads.data_ptr() + lo * esz, m.data_ptr() + lo * esz,
                        v.data_ptr() + lo * esz, hi - lo, float(g0["lr"]), float(b1), float(b2), float(g0["eps"]),
                        float(g0["weight_decay"]), self._steps, float(self.grad_scale),
                        None if self.clip_coef is None else self.clip_coef.data_ptr(), mirror.data_ptr(), n_bf16)
                if ek is None:
                    L.check(L.load().czc_adamw_step(*args, L.stream_ptr()))
                else:
                    L.check(L.load().czc_adamw_ema_step(*args, ek.flat_params.data_ptr() + lo * esz, k_mirror.data_ptr(),
                                                        float(owner.momentum), float(1.0 - owner.momentum), L.stream_ptr()))

            if split and 0 < n_mat < n and n_mat % 4 == 0:
                launch(0, n_mat, n_mat)            # weight matrices (their all-reduces were waited for by _pre_step)
                for fn in self._pre_tail:          # ... the vectors + embeddings slice has been averaged by now
                    fn()
                launch(n_mat, n, 0)
            else:
                if split:                          # no split point in this array: the tail all-reduce must have finished first
                    for fn in self._pre_tail:
                        fn()
                launch(0, n, n_mat)
            enc.mark_mirror_fresh()
            if ek is not None:
                ek.mark_mirror_fresh()
                owner._ema_prefolded = (enc._write_stamp(), ek._write_stamp())   # valid while torch writes neither encoder
            for p in enc._plist:
                st = self.state.get(p)
                if st is not None:
                    st["step"] = torch.tensor(float(self._steps))
        self.clip_coef = None
        return loss

    # ---- gradient clipping on the flat array (train.py:314-322) ----
    @torch.no_grad()
    def clip_grad_norm_(self, max_norm):
        for fn in self._pre_step + self._pre_tail:   # every overlapped gradient all-reduce must have finished averaging what is measured here
            fn()
        enc = self._encoders[0]
        g = enc.flat_grads(create=False)
        if g is None:
            return None
        if not hasattr(self, "_clip_ws") or self._clip_ws.device != g.device:
            self._clip_ws = torch.empty(4096 + 2, dtype=F32, device=g.device)
        part, ss, coef = self._clip_ws[:4096], self._clip_ws[4096:4097], self._clip_ws[4097:4098]
        lib = L.load()
        L.check(lib.czc_sumsq(g.data_ptr(), g.numel(), part.data_ptr(), ss.data_ptr(), L.stream_ptr()))
        L.check(lib.czc_clip_coef(ss.data_ptr(), float(self.grad_scale), float(max_norm), coef.data_ptr(), L.stream_ptr()))
        self.clip_coef = coef
        return ss  # squared norm (device); sqrt on demand

    @torch.no_grad()
    def clip_grad_value_(self, value):
        for fn in self._pre_step + self._pre_tail:
            fn()
        for enc in self._encoders:
            g = enc.flat_grads(create=False)
            if g is not None:
                L.check(L.load().czc_clamp(g.data_ptr(), g.numel(), -float(value), float(value), L.stream_ptr()))

    # ---- checkpoints: torch's layout; moments are copied into the flat arrays on load ----
    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        saved = state_dict["state"]
        all_params = [p for g in self.param_groups for p in g["params"]]
        for g_mine, g_saved in zip(self.param_groups, groups):
            for k, v in g_saved.items():
                if k != "params":
                    g_mine[k] = v
        steps = 0
        for enc in self._encoders:
            if not enc.flat_params.is_cuda:
                continue
            self._moments(enc)
        for idx, st in saved.items():
            p = all_params[int(idx)]
            mine = self.state.get(p)
            if mine is None:
                continue
            mine["exp_avg"].copy_(st["exp_avg"])
            mine["exp_avg_sq"].copy_(st["exp_avg_sq"])
            steps = max(steps, int(float(st["step"])))
        self._steps = steps
        for p, st in self.state.items():
            st["step"] = torch.tensor(float(steps))
