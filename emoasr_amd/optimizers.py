"""Optimizer front end with the reference's interface (asr/optimizers.py:6-117, asr/train_asr.py:228-229):

    optimizer_base = Adam(model.parameters(), lr=0, weight_decay=params.weight_decay)
    optimizer = ScheduledOptimizer(optimizer_base, params)
    ...
    optimizer.step(); optimizer.zero_grad(); optimizer.update_epoch(); optimizer.state_dict()

`Adam` is the fused HIP Adam (csrc/optim.hip) over the model's flat parameter / gradient arena: one
squared-norm launch and one update launch per step, global-norm clipping and the NaN/Inf skip
(train_asr.py:84-89) inside the update kernel, no host synchronisation.  It binds to the arena lazily (the
reference constructs its optimizer before `model.to(device)`), has ONE parameter group, and reads the
learning rate that ScheduledOptimizer writes into `param_groups[0]["lr"]`.  state_dict()/load_state_dict()
speak torch.optim.Adam's per-parameter layout, so `optim.ep{N}` files move between the two.
"""
import logging

import torch

from . import checkpoint


class Adam:
    def __init__(self, params, lr=0.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(params)
        if params and isinstance(params[0], dict):
            if len(params) != 1:
                raise NotImplementedError("emoasr_amd.optimizers.Adam: one parameter group (flat arena)")
            params = list(params[0]["params"])
        self.param_groups = [{"params": params, "lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay,
                              "amsgrad": False}]
        self._core = None
        self._pending = None  # a state dict loaded before the arena exists
        self.clip_grad_norm = 0.0  # set by train_step (the reference clips outside the optimizer)
        self.grad_mult = 1.0
        self._unclaimed = 0  # skipped updates already folded out of the core's counter, not yet out of a scheduler's

    # -- binding ---------------------------------------------------------------------------
    def _bind(self):
        from .engine import arena_of
        from .train import ArenaAdam
        arena = arena_of(self.param_groups[0]["params"])
        if self._core is not None:
            assert self._core.arena is arena, "the model was re-bound to a new arena after the optimizer took its first step"
            return self._core
        g = self.param_groups[0]
        self._core = ArenaAdam(arena, lambda step: self.param_groups[0]["lr"], betas=g["betas"], eps=g["eps"],
                               weight_decay=g["weight_decay"])
        if self._pending is not None:
            self._load(self._pending)
            self._pending = None
        return self._core

    @property
    def state(self):
        """torch.optim.Adam-style {param: {...}} view, for code that walks optimizer.state (optimizer_to)"""
        return {}

    # -- torch.optim.Optimizer surface -------------------------------------------------------
    def step(self):
        core = self._bind()
        core.clip = float(self.clip_grad_norm)
        core.step(grad_mult=self.grad_mult)

    def zero_grad(self, set_to_none=False):
        if self._core is not None or self._can_bind():
            self._bind().zero_grad()

    def fold_skipped(self):
        """-> number of updates the fused kernel skipped (NaN / Inf gradient norm) since the last call; the step counter
        behind the bias correction is corrected by it (one host synchronisation).  The count is also kept pending for the
        scheduler that wraps this optimizer (take_skipped), whoever triggered the fold."""
        n = self._core.fold_skipped() if self._core is not None else 0
        self._unclaimed += n
        return n

    def take_skipped(self):
        """fold, then hand over (and forget) every skipped update no scheduler has been told about yet"""
        self.fold_skipped()
        n, self._unclaimed = self._unclaimed, 0
        return n

    def _can_bind(self):
        from .engine import arena_of
        try:
            arena_of(self.param_groups[0]["params"])
            return True
        except LookupError:
            return False

    def state_dict(self):
        if self._core is None and not self._can_bind():
            return self._pending or {"state": {}, "param_groups": [dict(self.param_groups[0], params=list(
                range(len(self.param_groups[0]["params"]))))]}
        core = self._bind()
        core.lr = self.param_groups[0]["lr"]
        return checkpoint.optimizer_state_dict(core, 0.0, 0)["optimizer"]

    def load_state_dict(self, sd):
        if self._core is None and not self._can_bind():
            self._pending = sd
        else:
            self._bind()
            self._load(sd)
        if sd.get("param_groups"):
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in sd["param_groups"][0]:
                    self.param_groups[0][k] = sd["param_groups"][0][k]

    def _load(self, sd):
        steps = [int(v["step"]) for v in sd.get("state", {}).values()]
        checkpoint.load_optimizer_state_dict(self._core, {"optimizer": sd, "_step": max(steps) if steps else 0})


def _warmup_ramp(peak, warm, step):
    return (peak / max(1.0, warm)) * step


# schedule name -> rate(sched, step) for step >= 1 (asr/optimizers.py:47-75); `sched` carries the constants
_SCHEDULES = {
    # Vaswani et al.: peak * d^-0.5 * min(step^-0.5, step * warm^-1.5)
    "noam": lambda o, t: o.base_lr * o.model_dim ** (-0.5) * min(t ** (-0.5), t * o.num_warmup_steps ** (-1.5)),
    # linear ramp, then flat (the per-epoch decay happens in update_epoch)
    "epdecay": lambda o, t: _warmup_ramp(o.base_lr, o.num_warmup_steps, t) if t <= o.num_warmup_steps else o.base_lr,
    # linear ramp, then a straight line to zero at num_total_steps (transformers.get_linear_schedule_with_warmup)
    "lindecay": lambda o, t: _warmup_ramp(o.base_lr, o.num_warmup_steps, t) if t <= o.num_warmup_steps else
    o.base_lr * max(0.0, float(o.num_total_steps - t) / float(max(1.0, o.num_total_steps - o.num_warmup_steps))),
}

# the checkpoint boundary: keys of the reference's `optim.ep{N}` files (asr/optimizers.py:99-108), in their order
_CKPT_KEYS = ("_step", "_epoch", "base_lr", "_lr", "num_warmup_steps", "num_total_steps")


class ScheduledOptimizer:
    """Learning-rate schedules of asr/optimizers.py:45-97 ("noam", "epdecay", "lindecay") around a wrapped optimizer:
    before every step the rate for the new schedule position is written into the optimizer's parameter groups.

    Attribute names (`_step`, `_epoch`, `_lr`, `base_lr`, ...) and the state_dict layout are the reference's (they are
    what its checkpoints and its training log read); the schedules themselves are table entries (_SCHEDULES)."""

    def __init__(self, optimizer, params, num_total_steps=None):
        has_steps, has_prop = hasattr(params, "num_warmup_steps"), hasattr(params, "warmup_proportion")
        assert has_steps != has_prop, "give exactly one of num_warmup_steps / warmup_proportion"
        self.optimizer = optimizer
        self.schedule_type = params.lr_schedule_type
        self.base_lr = params.learning_rate
        self.num_total_steps = num_total_steps
        self.num_warmup_steps = int(num_total_steps * params.warmup_proportion) if has_prop else params.num_warmup_steps
        self._step = self._epoch = 0
        self._lr = 0
        if self.schedule_type == "noam":
            self.model_dim = getattr(params, "enc_hidden_size", None) or params.hidden_size
        elif self.schedule_type == "epdecay":
            self.lr_decay_start_epoch, self.lr_decay_rate = params.lr_decay_start_epoch, params.lr_decay_rate
        logging.info(f"lr scheduling type: {self.schedule_type}" + (f", warmup #steps: {self.num_warmup_steps:d}" if has_prop else ""))

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def _publish(self, lr):
        """make `lr` the rate of every parameter group (skipped when it did not change)"""
        if lr != self._lr:
            for group in self.optimizer.param_groups:
                group["lr"] = lr
        self._lr = lr

    def rate(self, step):
        """learning rate at schedule position `step`; None for an unknown schedule (as the reference leaves it).  Positions
        below 1 (every update so far was skipped and folded out) are clamped: noam's step^-0.5 has no value at 0."""
        fn = _SCHEDULES.get(self.schedule_type)
        return fn(self, max(int(step), 1)) if fn is not None else None

    def step(self):
        self._step += 1
        self._publish(self.rate(self._step))
        self.optimizer.step()

    def update_epoch(self):
        self.fold_skipped()
        self._epoch += 1
        if self.schedule_type == "epdecay" and self._epoch >= self.lr_decay_start_epoch:
            decayed = self._lr * self.lr_decay_rate
            logging.info(f"learning rate decreased: {self._lr:.6f} -> {decayed:.6f}")
            self._lr = None  # force the write: the epoch decay applies even when the product equals the old rate
            self._publish(decayed)

    def zero_grad(self):
        self.optimizer.zero_grad()

    def fold_skipped(self):
        """Take the updates that the fused Adam kernel skipped on the device (NaN / Inf gradient norm) out of the schedule
        position: the reference never calls step() for them (train_asr.py:88-91).  One host synchronisation; called at every
        epoch boundary, at the training loop's log-step synchronisation and before the state is saved.  The wrapped Adam
        keeps a pending count, so skips it folded on its own (a direct Adam.state_dict()) reach the schedule here too."""
        take = getattr(self.optimizer, "take_skipped", None)
        n = take() if take is not None else 0
        if n:
            logging.warning(f"{n:d} update(s) skipped because of nan grad_norm")
            self._step = max(self._step - n, 0)
        return n

    def state_dict(self):
        self.fold_skipped()
        out = {k: getattr(self, k) for k in _CKPT_KEYS}
        out["optimizer"] = self.optimizer.state_dict()
        return out

    def load_state_dict(self, state_dict):
        inner = state_dict.get("optimizer")
        if inner is not None:
            self.optimizer.load_state_dict(inner)
        total = state_dict.get("num_total_steps")
        assert total is None or total == self.num_total_steps, (total, self.num_total_steps)
        for key, value in state_dict.items():  # (unknown keys become attributes, as the reference's loop does)
            if key not in ("optimizer", "num_total_steps"):
                setattr(self, key, value)


def optimizer_to(optimizer, device):
    """asr/optimizers.py:120-125 moves torch.optim state tensors; the HIP Adam's moments live next to the
    parameter arena on its device already, so there is nothing to move"""
    return optimizer
