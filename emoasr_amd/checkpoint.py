"""Checkpoint I/O in the reference's formats: `model.ep{N}` = torch.save(model.state_dict())
(asr/train_asr.py:290-303; the module tree here has the same state_dict keys, so reference-trained
checkpoints load directly), checkpoint averaging (utils/average_checkpoints.py:16-52), latest-epoch
discovery for resume (utils/paths.py:81-112) and the optimizer file `optim.ep{N}`
(asr/optimizers.py:99-117: schedule counters + torch.optim.Adam.state_dict()).

The HIP optimizer (train.ArenaAdam) keeps its moments in two flat buffers laid out like the parameter
arena; `optimizer_state_dict` / `load_optimizer_state_dict` convert to and from torch.optim.Adam's
per-parameter layout (param index = position in model.parameters(): the module tree here registers
its parameters in the reference's order, which tests/test_host_io.py checks against the golden
state_dict key order).
"""
import logging
import os
import re

import torch


def save_model(model, path):
    torch.save(model.state_dict(), path)


def load_model(model, path, strict=True):
    model.load_state_dict(torch.load(path, map_location="cpu"), strict=strict)
    return model


def parse_epochs(ep):
    """"3-5" -> [3, 4, 5]; "1+4+7" -> [1, 4, 7]; a single epoch -> None (nothing to average)"""
    ep = str(ep)
    if "-" in ep:
        a, b = ep.split("-")
        return list(range(int(a), int(b) + 1))
    if "+" in ep:
        return [int(e) for e in ep.split("+")]
    return None


def average_state_dicts(states):
    """key-wise sum, then torch.div by the count -- integer buffers (num_batches_tracked) therefore come
    out as floats, exactly like the reference's file (average_checkpoints.py:38-50)"""
    avg = {k: v.clone() for k, v in states[0].items()}
    for sd in states[1:]:
        for k in avg:
            avg[k] += sd[k]
    return {k: (torch.div(v, len(states)) if v is not None else None) for k, v in avg.items()}


def model_average(save_dir, ep):
    """average `model.ep{e}` for the epochs named by `ep` into `model.ep{ep}` (skipped if it exists);
    returns the path, or None when `ep` is a single epoch"""
    epochs = parse_epochs(ep)
    if epochs is None:
        return None
    out = os.path.join(save_dir, f"model.ep{ep}")
    if os.path.exists(out):
        logging.info(f"checkpoint: {out} already exists!")
        return out
    states = [torch.load(os.path.join(save_dir, f"model.ep{e}"), map_location="cpu") for e in epochs]
    torch.save(average_state_dicts(states), out)
    return out


def resume_paths(save_dir, epoch=0):
    """-> (model_path, optim_path, epoch); epoch 0 = latest `model.epN` with a matching `optim.epN`
    ("" / "" / 0 when the directory holds none), as utils/paths.py:81-112"""
    if epoch <= 0:
        found = {"model": 0, "optim": 0}
        for name in os.listdir(save_dir) if os.path.isdir(save_dir) else []:
            m = re.fullmatch(r"(model|optim)\.ep([0-9]+)", name)
            if m:
                found[m.group(1)] = max(found[m.group(1)], int(m.group(2)))
        assert found["model"] == found["optim"], "latest model and optimizer checkpoints differ"
        epoch = found["model"]
    if epoch <= 0:
        return "", "", 0
    return os.path.join(save_dir, f"model.ep{epoch:d}"), os.path.join(save_dir, f"optim.ep{epoch:d}"), epoch


# ---- optimizer: flat arena moments <-> torch.optim.Adam layout --------------------------------------
def optimizer_state_dict(opt, base_lr, num_warmup_steps, epoch=0, num_total_steps=None):
    """the dict asr/optimizers.py:99-108 writes, from a train.ArenaAdam"""
    arena = opt.arena
    state, params = {}, []
    for i, n in enumerate(arena.module_order):
        o, v = arena.offsets[n], arena.pviews[n]
        if opt._step > 0:
            state[i] = {"step": torch.tensor(float(opt._step)),
                        "exp_avg": opt.m[o:o + v.numel()].view(v.shape).detach().cpu().clone(),
                        "exp_avg_sq": opt.v[o:o + v.numel()].view(v.shape).detach().cpu().clone()}
        params.append(i)
    group = {"lr": opt.lr, "betas": tuple(opt.betas), "eps": opt.eps, "weight_decay": opt.wd, "amsgrad": False,
             "params": params}
    return {"_step": opt._step, "_epoch": epoch, "base_lr": base_lr, "_lr": opt.lr,
            "num_warmup_steps": num_warmup_steps, "num_total_steps": num_total_steps,
            "optimizer": {"state": state, "param_groups": [group]}}


def load_optimizer_state_dict(opt, sd):
    """restore a train.ArenaAdam from a reference `optim.ep{N}` dict (or one written by the function
    above); returns the schedule fields (_epoch, base_lr, num_warmup_steps) for the caller's lr_fn"""
    arena = opt.arena
    st = sd["optimizer"]["state"]
    n_params = len(arena.module_order)
    bad = [i for i in st if not (isinstance(i, int) and 0 <= i < n_params)]
    assert not bad, f"optimizer state holds entries for parameters the model does not have: {bad[:5]}"
    # torch.optim.Adam creates state only for parameters that ever received a gradient (a frozen or unused head has none):
    # missing entries start from zero moments, which is what torch does at their first gradient.  (The fused step updates the
    # whole arena every step, so from here on they also see weight decay and moment decay like every other parameter.)
    opt.m.zero_()
    opt.v.zero_()
    for i, n in enumerate(arena.module_order):
        if i in st:
            o, v = arena.offsets[n], arena.pviews[n]
            assert tuple(st[i]["exp_avg"].shape) == tuple(v.shape), f"optimizer state of {n}: shape {tuple(st[i]['exp_avg'].shape)}"
            opt.m[o:o + v.numel()].view(v.shape).copy_(st[i]["exp_avg"])
            opt.v[o:o + v.numel()].view(v.shape).copy_(st[i]["exp_avg_sq"])
    opt._step = int(sd["_step"])
    opt.lr = float(sd.get("_lr", 0.0))
    return {k: sd.get(k) for k in ("_epoch", "base_lr", "num_warmup_steps", "num_total_steps")}
