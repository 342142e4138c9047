"""Optimizer side of the train step on the GPU: AdamW over one flat fp32 buffer, gradient-norm clipping and the EMA of
the weights fused into two HIP launches (csrc/optim.hip).

Mirrors what the reference's drivers assemble around the hot path:
  * `add_weight_decay` -- the name-based parameter groups of optim_factory.py:27-42 (biases, affine_weight / affine_bias,
    mean_shift, ParameterList entries `bias.N` and `model.no_weight_decay()` names get weight_decay 0);
  * `FlatAdamW` -- torch.optim.AdamW semantics (optim_factory.py:126-127) + `dispatch_clip_grad(mode='norm')`
    (engine.py:76-78) + `ModelEmaV2.update` (engine.py:89-90, main_qm9.py:169-175), as ONE optimizer object.  It is a
    torch.optim.Optimizer (param_groups with 'lr' for the schedulers, state_dict with exp_avg / exp_avg_sq / step per
    parameter), but parameters, moments and the EMA copy live in flat buffers and a step is: one multi-tensor copy of
    the gradients into the flat gradient buffer (skipped when `FlatGradAllReduce` already owns one), `eqf_sumsq` (only
    when clipping) and `eqf_adamw_step` -- instead of ~230 parameter tensors x (AdamW + clip + EMA) element-wise ops.
"""
import copy
import ctypes

import torch

from .lib import call


def add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    """[ref: optim_factory.py:27-42] -> [{'params': no_decay, 'weight_decay': 0.}, {'params': decay, ...}]"""
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if (name.endswith(".bias") or name.endswith(".affine_weight") or name.endswith(".affine_bias")
                or name.endswith(".mean_shift") or "bias." in name or name in skip_list):
            no_decay.append(param)
        else:
            decay.append(param)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def _P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class FlatAdamW(torch.optim.Optimizer):
    """AdamW (+ optional clip_grad_norm and EMA) on flat buffers.  `params`: iterable of tensors or param-group dicts
    (every group may set its own weight_decay; lr / betas / eps must agree across groups, as in the reference's
    drivers).  `reducer`: an `equiformer_amd.parallel.FlatGradAllReduce` over the same parameters whose flat gradient
    buffer is then used directly.  `ema_decay`: keep an exponential moving average of the weights; `ema_module(model)`
    returns a copy of `model` whose parameters alias it (what the drivers evaluate as `model_ema.module`)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, clip_grad=None,
                 ema_decay=None, reducer=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.clip_grad = clip_grad
        self.ema_decay = ema_decay
        ps = [p for g in self.param_groups for p in g["params"]]
        if not ps:
            raise ValueError("FlatAdamW got no parameters")
        self._check_params(ps)
        if reducer is not None:
            if [id(p) for p in reducer.params] != [id(p) for p in self._ordered(ps, reducer.params)]:
                raise ValueError("reducer and optimizer must hold the same parameters")
            ps = list(reducer.params)  # adopt the reducer's layout
        self._params = ps
        self._reducer = reducer
        dev = ps[0].device
        self.sizes = [p.numel() for p in ps]
        from .parallel import flat_offsets
        self.offsets, n = flat_offsets(self.sizes)  # every slice starts 256-byte aligned (same layout as the reducer's)
        self.n = n
        if reducer is not None and (list(reducer.offsets) != self.offsets or reducer.flat.numel() != n):
            raise ValueError("reducer and optimizer flat layouts differ")
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_wd = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = reducer.flat if reducer is not None else torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_ema = None
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        wd_of = {id(p): g["weight_decay"] for g in self.param_groups for p in g["params"]}
        self._gviews = []
        with torch.no_grad():
            for p, k, off in zip(ps, self.sizes, self.offsets):
                self.flat_p[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)  # the parameter now aliases the flat buffer
                self.flat_wd[off:off + k].fill_(float(wd_of[id(p)]))
                self._gviews.append(self.flat_g[off:off + k].view_as(p))
                st = self.state[p]
                st["step"] = 0
                st["exp_avg"] = self.flat_m[off:off + k].view_as(p)
                st["exp_avg_sq"] = self.flat_v[off:off + k].view_as(p)
        if ema_decay is not None:
            self.flat_ema = self.flat_p.clone()
        self._step = 0
        self._hyper_dev = None  # device {lr, 1 - b1^t, sqrt(1 - b2^t)} of a captured step (equiformer_amd/capture.py)
        self._hyper_host = None
        self._lag = [0] * len(ps)  # steps a parameter missed because it had no gradient (torch keeps a per-parameter count)
        self._wd_value = [float(wd_of[id(p)]) for p in ps]
        self._skipped = set()  # parameters whose slice of flat_wd currently holds the kernel's "skip" mark (-1)

    @staticmethod
    def _check_params(ps):
        if any((not p.is_cuda) or p.dtype != torch.float32 for p in ps):
            from .ops import HipOnlyError
            raise HipOnlyError("FlatAdamW runs on GPU fp32 parameters only")

    @staticmethod
    def _ordered(ps, like):
        ids = {id(p) for p in ps}
        return [p for p in like if id(p) in ids]

    def ema_module(self, model):
        """Deep copy of `model` whose parameters are views of the EMA buffer (buffers are copied as they are)."""
        if self.flat_ema is None:
            raise RuntimeError("FlatAdamW was built without ema_decay")
        m = copy.deepcopy(model)
        by_id = {id(p): i for i, p in enumerate(self._params)}
        for (_, src), (_, dst) in zip(model.named_parameters(), m.named_parameters()):
            i = by_id.get(id(src))
            if i is not None:
                dst.data = self.flat_ema[self.offsets[i]:self.offsets[i] + self.sizes[i]].view_as(src)
            dst.requires_grad_(False)
        return m

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g0 = self.param_groups[0]
        for g in self.param_groups[1:]:
            if g["lr"] != g0["lr"] or g["betas"] != g0["betas"] or g["eps"] != g0["eps"]:
                raise ValueError("FlatAdamW: lr / betas / eps must agree across parameter groups")
        # gradients -> flat buffer (already there when the reducer re-pointed .grad at its slices)
        src, dst, zero, slow, nograd = [], [], [], [], set()
        offs = self.offsets
        for i, (p, v, k, off) in enumerate(zip(self._params, self._gviews, self.sizes, self.offsets)):
            if p.grad is None:
                zero.append(v)
                nograd.add(i)
                self._lag[i] += 1
            else:
                if p.grad.data_ptr() != v.data_ptr():
                    src.append(p.grad)
                    dst.append(v)
                if self._lag[i] > 0:
                    slow.append((i, off, k, True))
        # torch.optim.AdamW skips a parameter without a gradient entirely (no decay, no moment decay, its own step
        # count is not advanced).  The kernel does the same for elements whose weight-decay entry is negative; the
        # marks are rewritten only when the SET of gradient-less parameters changes (in practice never after the
        # first step: a model's unused parameters are the same every step), so a step issues no per-parameter copies.
        # (Before round 3 every such parameter cost eight device-to-device copies per step: 58 per QM9 step.)
        if nograd != self._skipped:
            for i in nograd - self._skipped:
                self.flat_wd[offs[i]:offs[i] + self.sizes[i]].fill_(-1.0)
            for i in self._skipped - nograd:
                self.flat_wd[offs[i]:offs[i] + self.sizes[i]].fill_(self._wd_value[i])
            self._skipped = set(nograd)
        # Parameters that were skipped BEFORE and have a gradient now lag behind the global step count (their bias
        # corrections differ): saved here and redone below with tensor ops on their slices.
        saved = [(i, o, k, has, self.flat_p[o:o + k].clone(), self.flat_m[o:o + k].clone(), self.flat_v[o:o + k].clone(),
                  None if self.flat_ema is None else self.flat_ema[o:o + k].clone()) for i, o, k, has in slow]
        if zero:
            torch._foreach_zero_(zero)
        if src:
            torch._foreach_copy_(dst, src)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.flat_p.device).cuda_stream)
        clip = self.clip_grad is not None
        if clip:
            call("eqf_sumsq", _P(self.flat_g), self.n, _P(self._sumsq), st)
        self._step += 1
        b1, b2 = g0["betas"]
        lr, eps = float(g0["lr"]), float(g0["eps"])
        if self._hyper_dev is not None:  # captured steps (and eager ones beside them): lr and the bias corrections from the device
            if not torch.cuda.is_current_stream_capturing():  # (a replay is preceded by advance_captured())
                self.write_hyper()
            call("eqf_adamw_step_dev", _P(self.flat_p), _P(self.flat_g), _P(self.flat_m), _P(self.flat_v), _P(self.flat_wd),
                 _P(self.flat_ema), _P(self._sumsq) if clip else None, self.n, _P(self._hyper_dev), float(b1), float(b2), eps,
                 float(self.clip_grad or 0.0), float(self.ema_decay or 0.0), st)
        else:
            call("eqf_adamw_step", _P(self.flat_p), _P(self.flat_g), _P(self.flat_m), _P(self.flat_v), _P(self.flat_wd),
                 _P(self.flat_ema), _P(self._sumsq) if clip else None, self.n, lr, float(b1), float(b2), eps, self._step,
                 float(self.clip_grad or 0.0), float(self.ema_decay or 0.0), st)
        for i, o, k, has, sp, sm, sv, se in saved:
            if has:  # same arithmetic as the kernel, with this parameter's own step count
                t = self._step - self._lag[i]
                g = self.flat_g[o:o + k]
                if clip:
                    g = g * torch.clamp(self.clip_grad / (self._sumsq.sqrt() + 1e-6), max=1.0)
                sp.mul_(1.0 - lr * self.flat_wd[o:o + k])
                sm.mul_(b1).add_(g, alpha=1.0 - b1)
                sv.mul_(b2).addcmul_(g, g, value=1.0 - b2)
                denom = (sv.sqrt() / (1.0 - b2 ** t) ** 0.5).add_(eps)
                sp.addcdiv_(sm, denom, value=-lr / (1.0 - b1 ** t))
            self.flat_p[o:o + k].copy_(sp)
            self.flat_m[o:o + k].copy_(sm)
            self.flat_v[o:o + k].copy_(sv)
            if se is not None:  # the EMA follows the weights, updated or not
                self.flat_ema[o:o + k].copy_(se.mul_(self.ema_decay).add_(sp, alpha=1.0 - self.ema_decay))
        for i, p in enumerate(self._params):
            self.state[p]["step"] = self._step - self._lag[i]
        return loss

    def device_hyper(self, on=True):
        """Captured steps (equiformer_amd/capture.py): the step reads lr and the bias corrections from a device array that
        `write_hyper()` refreshes (the same double-precision host arithmetic as the by-value launch, so the update is bit-equal)."""
        if on and self._hyper_dev is None:
            self._hyper_dev = torch.zeros(4, dtype=torch.float32, device=self.flat_p.device)
            self._hyper_host = torch.zeros(4, dtype=torch.float32).pin_memory()
        elif not on:
            self._hyper_dev = self._hyper_host = None

    def write_hyper(self):
        """{lr, 1 - b1^t, sqrt(1 - b2^t)} of the CURRENT step count -> device (asynchronous copy on the current stream)."""
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        t = max(self._step, 1)
        # fp32 roundings of the double-precision values, as csrc/optim.hip computes them for the by-value launch
        self._hyper_host[0] = float(g0["lr"])
        self._hyper_host[1] = 1.0 - float(b1) ** t
        self._hyper_host[2] = (1.0 - float(b2) ** t) ** 0.5
        self._hyper_dev.copy_(self._hyper_host, non_blocking=True)

    def advance_captured(self):
        """Host-side bookkeeping of one REPLAY of a captured step (the Python of step() does not run then)."""
        self._step += 1
        self.write_hyper()
        for i, p in enumerate(self._params):
            self.state[p]["step"] = self._step - self._lag[i]

    def load_state_dict(self, state_dict):
        """Restores lr / betas / eps / weight decay of the groups and copies exp_avg / exp_avg_sq / step INTO the flat
        buffers (the per-parameter state tensors stay views of them)."""
        groups = state_dict["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        for g, sg in zip(self.param_groups, groups):
            if len(g["params"]) != len(sg["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of the group")
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in sg:
                    g[k] = tuple(sg[k]) if k == "betas" else sg[k]
        order = [p for g in self.param_groups for p in g["params"]]
        ids = [i for sg in groups for i in sg["params"]]
        step = 0
        with torch.no_grad():
            for p, i in zip(order, ids):
                st = state_dict["state"].get(i)
                if st is None:
                    continue
                self.state[p]["exp_avg"].copy_(st["exp_avg"])
                self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                step = max(step, int(st["step"]))
            wd_of = {id(p): g["weight_decay"] for g in self.param_groups for p in g["params"]}
            for p, k, off in zip(self._params, self.sizes, self.offsets):
                self.flat_wd[off:off + k].fill_(float(wd_of[id(p)]))
            self._wd_value = [float(wd_of[id(p)]) for p in self._params]
            self._skipped = set()  # the skip marks went with the rewrite; step() puts them back
        self._step = step
        own = {id(p): int(state_dict["state"][i]["step"]) for p, i in zip(order, ids) if i in state_dict["state"]}
        for i, p in enumerate(self._params):
            self._lag[i] = step - own.get(id(p), step)  # parameters that missed steps keep their own (smaller) count
            self.state[p]["step"] = step - self._lag[i]

    def grad_norm(self):
        """Global gradient norm of the last clipped step (device scalar, no sync)."""
        return self._sumsq.sqrt()
