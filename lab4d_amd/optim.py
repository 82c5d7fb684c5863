"""Optimizer step on the gfx950 kernels of csrc/optim.hip (include/lab4d_optim.h): the reference's
`torch.nn.utils.clip_grad_norm_` + `torch.optim.AdamW` over one group per parameter (lab4d/engine/trainer.py:164-190,350,
581-604) as three launches over ONE flat fp32 buffer.

`FlatAdamW(params, lr)` moves the parameters into a flat buffer (each padded to a multiple of 4 elements) and re-points
`p.data` / `p.grad` at views of it, so autograd keeps accumulating into the flat gradient buffer and `flat_grad` is directly
what a data-parallel job all-reduces (one RCCL call, no gather / scatter copies).
"""
import torch

from . import _lib

vp, ci, cf, i64 = _lib.vp, _lib.ci, _lib.cf, __import__("ctypes").c_int64
_lib.register("lab4d_grad_norm_clip", [vp, i64, cf, vp, vp, vp, vp])
_lib.register("lab4d_adamw_step", [vp, vp, vp, vp, i64, vp, vp, ci, cf, cf, cf, cf, ci, vp, vp])
_lib.register("lab4d_check_grad", [vp, i64, cf, cf, vp, vp, vp, vp, vp, vp])
_lib.register("lab4d_adamw_step_guarded", [vp, vp, vp, vp, i64, vp, vp, ci, cf, cf, cf, cf, vp, vp, vp, vp])


class FlatAdamW:
    """torch.optim.AdamW semantics (decoupled weight decay, bias-corrected moments) with one learning rate per parameter.

    params: list of leaf tensors on one device; lr: float or one float per parameter (the reference's per-parameter
    OneCycleLR rates; update with set_lr).  betas / eps / weight_decay default to the reference's (trainer.py:185-190)."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4):
        self.params = list(params)
        if not self.params:
            raise RuntimeError("FlatAdamW: no parameters")
        # the flat layout (and with it the all-reduce bucket `flat_grad`) is plain tensor bookkeeping and works on any device
        # -- the CPU tests of the data-parallel path use exactly this object; step() / grad_norm_clip() need the HIP library
        dev = self.params[0].device
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        ends, off = [], 0
        self.offsets = []
        for p in self.params:
            if p.dtype != torch.float32:
                raise RuntimeError("FlatAdamW: fp32 parameters only")
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
            ends.append(off)
        self.n = off
        self.flat = torch.zeros(off, device=dev)
        self.flat_grad = torch.zeros(off, device=dev)
        self.m = torch.zeros(off, device=dev)
        self.v = torch.zeros(off, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
        self.seg_end = torch.tensor(ends, dtype=torch.int64, device=dev)
        self.seg_lr = torch.empty(len(ends), device=dev)
        self.set_lr(lr)
        self.work = torch.empty(512, device=dev)
        self.norm = torch.zeros(1, device=dev)
        self.coef = torch.ones(1, device=dev)
        self.steps = 0  # host-side count of step() calls
        # check_grad state on the device: 1 when the last step was discarded; the number of steps actually taken; how many were discarded
        self.skipped = torch.zeros(1, dtype=torch.int32, device=dev)
        self.dev_step = torch.zeros(1, dtype=torch.int32, device=dev)

    def set_lr(self, lr):
        lrs = [float(lr)] * len(self.params) if not hasattr(lr, "__len__") else [float(x) for x in lr]
        if len(lrs) != len(self.params):
            raise RuntimeError("FlatAdamW: %d learning rates for %d parameters" % (len(lrs), len(self.params)))
        self.seg_lr.copy_(torch.tensor(lrs), non_blocking=True)

    def zero_grad(self):
        self.flat_grad.zero_()

    def grad_norm_clip(self, max_norm):
        """Global gradient norm and clip_grad_norm_'s coefficient, both left on the device (no host sync); returns the norm."""
        _lib.require_device(self.flat_grad)
        _lib.check(_lib.lib().lab4d_grad_norm_clip(_lib.ptr(self.flat_grad), self.n, float(max_norm), _lib.ptr(self.work), _lib.ptr(self.norm),
                                                   _lib.ptr(self.coef), _lib.stream()), "grad_norm_clip")
        return self.norm

    def check_grad(self, max_norm, skip_above=None):
        """Trainer.check_grad (engine/trainer.py:581-604) on the device: the global gradient norm, clip_grad_norm_'s coefficient and the
        discard decision (pre-clip norm > skip_above, default max_norm like the reference, or not finite) -- `norm`, `coef`, `skipped` stay on
        the device, the NEXT step() applies them; no host synchronisation.  Returns the norm tensor."""
        _lib.require_device(self.flat_grad)
        thresh = float(max_norm if skip_above is None else skip_above)
        _lib.check(_lib.lib().lab4d_check_grad(_lib.ptr(self.flat_grad), self.n, float(max_norm), thresh, _lib.ptr(self.work), _lib.ptr(self.norm),
                                               _lib.ptr(self.coef), _lib.ptr(self.skipped), _lib.ptr(self.dev_step), _lib.stream()), "check_grad")
        self._checked = True
        self._guarded_used = True
        return self.norm

    def step(self, max_norm=None, skip_above=None):
        """One AdamW step; with max_norm the gradients are scaled by min(1, max_norm / (norm + 1e-6)) inside the update.

        skip_above (with max_norm) = Trainer.check_grad's discard rule (engine/trainer.py:581-604): when the pre-clip norm exceeds it,
        or is not finite, the step is a no-op on the device -- parameters, moments and the step count of the bias corrections stay
        untouched -- and `self.skipped` (device int32) is 1; no host synchronisation.  The reference then reloads the weights cached two
        rounds ago when it has any (trainer.py:598-604): a caller does that from `self.skipped` at a point where it synchronises anyway.
        After a separate check_grad() call (the reference's order: check_grad(), then optimizer.step()) step() takes no arguments."""
        _lib.require_device(self.flat, self.flat_grad)
        self.steps += 1
        if skip_above is not None:
            if max_norm is None:
                raise RuntimeError("FlatAdamW.step: skip_above needs max_norm (check_grad clips and checks in one pass)")
            self.check_grad(max_norm, skip_above)
        if getattr(self, "_checked", False):
            self._checked = False
            _lib.check(_lib.lib().lab4d_adamw_step_guarded(_lib.ptr(self.flat), _lib.ptr(self.flat_grad), _lib.ptr(self.m), _lib.ptr(self.v), self.n,
                                                           _lib.ptr(self.seg_end), _lib.ptr(self.seg_lr), len(self.params), self.betas[0],
                                                           self.betas[1], self.eps, self.weight_decay, _lib.ptr(self.coef), _lib.ptr(self.skipped),
                                                           _lib.ptr(self.dev_step), _lib.stream()), "adamw_step_guarded")
            torch.autograd.graph.increment_version(self.params)
            return
        if max_norm is not None:
            self.grad_norm_clip(max_norm)
        _lib.check(_lib.lib().lab4d_adamw_step(_lib.ptr(self.flat), _lib.ptr(self.flat_grad), _lib.ptr(self.m), _lib.ptr(self.v), self.n,
                                               _lib.ptr(self.seg_end), _lib.ptr(self.seg_lr), len(self.params), self.betas[0], self.betas[1], self.eps,
                                               self.weight_decay, self._taken(), _lib.ptr(self.coef) if max_norm is not None else None, _lib.stream()),
                   "adamw_step")
        # the kernel wrote through raw pointers: tell autograd (and the packed-weight caches keyed on it) that the data changed
        torch.autograd.graph.increment_version(self.params)

    def _taken(self):
        """Step number of the unguarded path (host-side): steps taken so far.  Mixing guarded and unguarded steps on one optimizer would
        need the device count on the host -- refused."""
        if getattr(self, "_guarded_used", False):
            raise RuntimeError("FlatAdamW: step() without skip_above after guarded steps (the step count lives on the device)")
        return self.steps


class TorchFlatAdamW(torch.optim.Optimizer):
    """FlatAdamW behind the torch.optim.Optimizer interface the reference's trainer programs against (engine/trainer.py:185-210,
    255-270, 343-350, 581-604): `param_groups` with one group per parameter (OneCycleLR writes group["lr"] every step and stores
    "initial_lr" / "max_lr" / "min_lr" there), step() / zero_grad(), state_dict() / load_state_dict() (the two-rounds-back cache of
    check_grad's rollback and the checkpoints).  The update itself is FlatAdamW's three launches; the groups' learning rates are uploaded
    when they changed.  Build it directly from a list of {"params": [p]} groups like torch.optim.AdamW, or adopt an existing AdamW (and
    whatever scheduler already points at it) with TorchFlatAdamW.adopt(opt)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._build()

    @classmethod
    def adopt(cls, opt):
        """Turn a freshly constructed torch.optim.AdamW (no step taken yet) into a TorchFlatAdamW IN PLACE: the object identity, its
        param_groups list and the group dicts survive, so a learning-rate scheduler created on it keeps working."""
        if not isinstance(opt, torch.optim.AdamW):
            raise RuntimeError("TorchFlatAdamW.adopt: expected torch.optim.AdamW, got %s" % type(opt).__name__)
        if any(len(st) for st in opt.state.values()):
            raise RuntimeError("TorchFlatAdamW.adopt: the optimizer has already stepped")
        opt.__class__ = cls
        # A scheduler built on the AdamW before the swap (Trainer.optimizer_init does: OneCycleLR, engine/trainer.py:199-207) has left an INSTANCE
        # attribute opt.step behind -- LRScheduler.__init__ wraps the bound AdamW.step it found to set _opt_called -- and an instance attribute
        # outlives the class swap: optimizer.step() would run torch's Adam.step on this object.  Replace it with the same kind of wrapper around
        # THIS class's step (the scheduler's "step() before optimizer.step()" warning keeps working).
        stale = opt.__dict__.pop("step", None)
        if stale is not None and getattr(stale, "_wrapped_by_lr_sched", False):
            def step(*args, **kwargs):
                opt._opt_called = True
                return cls.step(opt, *args, **kwargs)
            step._wrapped_by_lr_sched = True
            opt.step = step
        opt._build()
        return opt

    def _build(self):
        ps, seen = [], set()
        for g in self.param_groups:
            if g.get("amsgrad") or g.get("maximize"):
                raise NotImplementedError("TorchFlatAdamW: amsgrad / maximize are not implemented")
            for p in g["params"]:
                if id(p) in seen:
                    raise RuntimeError("TorchFlatAdamW: a parameter appears in two groups")
                seen.add(id(p))
                ps.append(p)
        g0 = self.param_groups[0]
        for g in self.param_groups:
            if (tuple(g["betas"]), g["eps"], g["weight_decay"]) != (tuple(g0["betas"]), g0["eps"], g0["weight_decay"]):
                raise NotImplementedError("TorchFlatAdamW: betas / eps / weight_decay must be the same for every group (the reference's are)")
        self.flat = FlatAdamW(ps, self._lrs(), betas=tuple(g0["betas"]), eps=g0["eps"], weight_decay=g0["weight_decay"])
        self._lr_sent = self._lrs()
        for p, o in zip(self.flat.params, self.flat.offsets):  # torch's per-parameter state, as views of the flat moment buffers
            self.state[p] = {"step": torch.zeros((), dtype=torch.float32), "exp_avg": self.flat.m[o:o + p.numel()].view_as(p),
                             "exp_avg_sq": self.flat.v[o:o + p.numel()].view_as(p)}

    def _lrs(self):
        return [float(g["lr"]) for g in self.param_groups for _ in g["params"]]

    @property
    def skipped(self):
        return self.flat.skipped

    def check_grad(self, thresh):
        return self.flat.check_grad(thresh)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None if closure is None else closure()
        lrs = self._lrs()
        if lrs != self._lr_sent:
            self.flat.set_lr(lrs)
            self._lr_sent = lrs
        self.flat.step()
        return loss

    def zero_grad(self, set_to_none=True):
        """Zeroes the flat gradient buffer; the parameters keep their .grad VIEWS of it whatever set_to_none says (the weight-gradient
        kernels and the data-parallel all-reduce work on that buffer)."""
        self.flat.zero_grad()
        for p, o in zip(self.flat.params, self.flat.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat.flat_grad[o:o + p.numel()].view_as(p)

    def state_dict(self):
        step = float(int(self.flat.dev_step)) if getattr(self.flat, "_guarded_used", False) else float(self.flat.steps)
        for st in self.state.values():
            st["step"] = torch.tensor(step)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        views = {p: (st["exp_avg"], st["exp_avg_sq"]) for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        step = 0
        with torch.no_grad():
            for p, (m, v) in views.items():
                st = self.state.get(p)
                if st:  # loaded moments are copies: put them back into the flat buffers and re-point the state at the views
                    m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                    step = int(st["step"])
                else:
                    m.zero_(); v.zero_()
                self.state[p] = {"step": torch.tensor(float(step)), "exp_avg": m, "exp_avg_sq": v}
        self.flat.steps = step
        self.flat.dev_step.fill_(step)
        self._lr_sent = None
