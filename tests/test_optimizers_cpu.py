"""ScheduledOptimizer on the CPU (no GPU needed): the three schedules against the reference's formulas
(asr/optimizers.py:45-82, restated here), the `optim.ep{N}` key layout (asr/optimizers.py:99-108), the epoch decay, and the
bookkeeping of updates the fused Adam skipped on the device (NaN gradient norm: train_asr.py:88-91 never calls step())."""
from types import SimpleNamespace

import pytest
import torch

from emoasr_amd.optimizers import ScheduledOptimizer


class _Opt:
    """stands in for the wrapped optimizer: records the rate it is stepped with, reports skipped updates on request"""

    def __init__(self):
        self.param_groups = [{"lr": 0.0}, {"lr": 0.0}]
        self.rates, self.skipped = [], 0

    def step(self):
        self.rates.append(self.param_groups[0]["lr"])

    def zero_grad(self):
        pass

    def state_dict(self):
        return {"state": {}, "param_groups": [{"lr": g["lr"]} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.loaded = sd

    def take_skipped(self):
        n, self.skipped = self.skipped, 0
        return n


def _noam(base, d, warm, t):
    return base * d ** (-0.5) * min(t ** (-0.5), t * warm ** (-1.5))


def test_noam_schedule_and_state_layout():
    p = SimpleNamespace(lr_schedule_type="noam", learning_rate=5.0, num_warmup_steps=4, enc_hidden_size=256)
    opt = _Opt()
    sch = ScheduledOptimizer(opt, p)
    for _ in range(9):
        sch.step()
    assert opt.rates == [_noam(5.0, 256, 4, t) for t in range(1, 10)]
    assert all(g["lr"] == opt.rates[-1] for g in opt.param_groups)
    sd = sch.state_dict()
    assert list(sd) == ["_step", "_epoch", "base_lr", "_lr", "num_warmup_steps", "num_total_steps", "optimizer"]
    assert sd["_step"] == 9 and sd["_lr"] == opt.rates[-1]
    other = ScheduledOptimizer(_Opt(), p)
    other.load_state_dict(sd)
    assert other._step == 9 and other._lr == sd["_lr"] and other.optimizer.loaded == sd["optimizer"]
    other.step()
    assert other.optimizer.rates == [_noam(5.0, 256, 4, 10)]


def test_epdecay_and_lindecay():
    p = SimpleNamespace(lr_schedule_type="epdecay", learning_rate=1e-3, num_warmup_steps=3, lr_decay_start_epoch=2,
                        lr_decay_rate=0.5, hidden_size=8)
    opt = _Opt()
    sch = ScheduledOptimizer(opt, p)
    for _ in range(5):
        sch.step()
    want = [(1e-3 / 3.0) * t for t in (1, 2, 3)] + [1e-3, 1e-3]
    assert opt.rates == want
    sch.update_epoch()                      # epoch 1: below lr_decay_start_epoch
    assert sch._lr == 1e-3
    sch.update_epoch()                      # epoch 2: decays, and writes the rate even though no step was taken
    assert sch._lr == 5e-4 and opt.param_groups[1]["lr"] == 5e-4
    p2 = SimpleNamespace(lr_schedule_type="lindecay", learning_rate=2.0, warmup_proportion=0.25, hidden_size=8)
    opt2 = _Opt()
    sch2 = ScheduledOptimizer(opt2, p2, num_total_steps=8)
    assert sch2.num_warmup_steps == 2
    for _ in range(8):
        sch2.step()
    want2 = [(2.0 / 2.0) * 1, (2.0 / 2.0) * 2] + [2.0 * max(0.0, float(8 - t) / float(8 - 2)) for t in range(3, 9)]
    assert opt2.rates == want2
    with pytest.raises(AssertionError):
        sch2.load_state_dict({"num_total_steps": 9})


def test_skipped_updates_leave_the_schedule_position():
    p = SimpleNamespace(lr_schedule_type="noam", learning_rate=1.0, num_warmup_steps=10, enc_hidden_size=16)
    opt = _Opt()
    sch = ScheduledOptimizer(opt, p)
    for _ in range(3):
        sch.step()
    opt.skipped = 2                         # the device skipped two of the three updates (NaN gradient norm)
    assert sch.fold_skipped() == 2 and sch._step == 1
    sch.step()
    assert opt.rates[-1] == _noam(1.0, 16, 10, 2)
    opt.skipped = 5                         # more skips than steps: the position never goes below zero ...
    sch.fold_skipped()
    assert sch._step == 0
    assert sch.rate(0) == _noam(1.0, 16, 10, 1)   # ... and the rate there is the first step's (no 0 ** -0.5)
    assert sch.state_dict()["_step"] == 0
