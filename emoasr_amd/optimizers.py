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
        behind the bias correction is corrected by it (one host synchronisation)"""
        return self._core.fold_skipped() if self._core is not None else 0

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


class ScheduledOptimizer:
    """learning-rate schedules of asr/optimizers.py:45-97 ("noam", "epdecay", "lindecay"); the rate is written
    into the wrapped optimizer's param groups before every step"""

    def __init__(self, optimizer, params, num_total_steps=None):
        self.optimizer = optimizer
        self.schedule_type = params.lr_schedule_type
        self._step = 0
        self._epoch = 0
        self.base_lr = params.learning_rate
        self.num_total_steps = num_total_steps
        assert hasattr(params, "num_warmup_steps") ^ hasattr(params, "warmup_proportion")
        if hasattr(params, "warmup_proportion"):
            self.num_warmup_steps = int(num_total_steps * params.warmup_proportion)
            logging.info(f"warmup #steps: {self.num_warmup_steps:d}")
        else:
            self.num_warmup_steps = params.num_warmup_steps
        self._lr = 0
        logging.info(f"lr scheduling type: {self.schedule_type}")
        if self.schedule_type == "epdecay":
            self.lr_decay_start_epoch = params.lr_decay_start_epoch
            self.lr_decay_rate = params.lr_decay_rate
        elif self.schedule_type == "noam":
            self.model_dim = params.enc_hidden_size if hasattr(params, "enc_hidden_size") else params.hidden_size

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def _set_lr(self, lr):
        for group in self.optimizer.param_groups:
            group["lr"] = lr

    def rate(self, step):
        warm = self.num_warmup_steps
        if self.schedule_type == "noam":
            return self.base_lr * self.model_dim ** (-0.5) * min(step ** (-0.5), step * warm ** (-1.5))
        if step <= warm:  # epdecay / lindecay: linear warm-up
            return (self.base_lr / max(1.0, warm)) * step
        if self.schedule_type == "epdecay":
            return self.base_lr
        if self.schedule_type == "lindecay":
            return self.base_lr * max(0.0, float(self.num_total_steps - step) / float(max(1.0, self.num_total_steps - warm)))
        return None

    def step(self):
        self._step += 1
        new_lr = self.rate(self._step)
        if new_lr != self._lr:
            self._set_lr(new_lr)
        self._lr = new_lr
        self.optimizer.step()

    def update_epoch(self):
        self.fold_skipped()
        self._epoch += 1
        if self.schedule_type == "epdecay" and self._epoch >= self.lr_decay_start_epoch:
            new_lr = self._lr * self.lr_decay_rate
            self._set_lr(new_lr)
            logging.info(f"learning rate decreased: {self._lr:.6f} -> {new_lr:.6f}")
            self._lr = new_lr

    def zero_grad(self):
        self.optimizer.zero_grad()

    def fold_skipped(self):
        """Take the updates that the fused Adam kernel skipped on the device (NaN / Inf gradient norm) out of the schedule
        position: the reference never calls step() for them (train_asr.py:88-91).  One host synchronisation; called at every
        epoch boundary and before the state is saved."""
        base = getattr(self.optimizer, "fold_skipped", None)
        n = base() if base is not None else 0
        if n:
            logging.warning(f"{n:d} update(s) skipped because of nan grad_norm")
            self._step -= n
        return n

    def state_dict(self):
        self.fold_skipped()
        return {"_step": self._step, "_epoch": self._epoch, "base_lr": self.base_lr, "_lr": self._lr,
                "num_warmup_steps": self.num_warmup_steps, "num_total_steps": self.num_total_steps,
                "optimizer": self.optimizer.state_dict()}

    def load_state_dict(self, state_dict):
        for key, value in state_dict.items():
            if key == "optimizer":
                self.optimizer.load_state_dict(value)
            elif key == "num_total_steps" and value is not None:
                assert self.num_total_steps == value
            else:
                setattr(self, key, value)


def optimizer_to(optimizer, device):
    """asr/optimizers.py:120-125 moves torch.optim state tensors; the HIP Adam's moments live next to the
    parameter arena on its device already, so there is nothing to move"""
    return optimizer
