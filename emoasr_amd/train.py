"""Training-step glue on the flat parameter arena: fused Adam (HIP), noam schedule, global-norm
clipping and NaN skip on device, one RCCL all-reduce of the gradient arena per optimizer step.

Reference semantics: asr/train_asr.py:35-97 (micro-batch / accumulate / clip / NaN skip / step),
asr/optimizers.py:45-82 (ScheduledOptimizer: lr = base_lr * d^-0.5 * min(step^-0.5, step*warmup^-1.5)
written to the param groups before Adam.step), torch.optim.Adam with coupled weight decay
(train_asr.py:228).  Data parallelism replaces nn.DataParallel (train_asr.py:236-243): one
process per GPU, loss = mean of per-replica batch-mean losses => all-reduce(sum) / world.
"""
import torch



def noam_lr(base_lr, d_model, warmup, step):
    return base_lr * d_model ** (-0.5) * min(step ** (-0.5), step * warmup ** (-1.5))


def allreduce_sum_(flat, group=None):
    """one collective per optimizer step: sum the flat gradient buffer over the data-parallel ranks
    (backend nccl = RCCL over xGMI on the GPU box; gloo in the CPU tests)"""
    import torch.distributed as dist
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class GradBuckets:
    """Overlap of the gradient all-reduce with the backward sweep.  The backward finishes gradients from
    the END of the flat arena towards its start (decoder / head first, then encoder layers 11..0, then
    the front-end), so `ready(lo)` -- "everything at or above offset lo is final" -- lets the tail
    [lo, done) go out as an asynchronous all-reduce while the layers below are still being
    differentiated.  Ranges are merged until they reach `min_elems` (a few large collectives: xGMI rings
    are per-link bound).  `finish()` sends what is left and waits; the sum over all ranges is exactly
    one all-reduce of the whole buffer (train_asr.py:67-71 semantics are applied by the optimizer's
    grad_mult = 1/world, as before)."""

    def __init__(self, flat_grad, group=None, min_elems=4 << 20, comm_dtype=None):
        """comm_dtype=torch.bfloat16: each range travels as a bf16 copy (half the bytes on the per-link-bound xGMI
        rings: 47 MB instead of 94 MB per step for L2) and is widened back into the f32 arena in finish(); the sum
        itself then rounds to bf16 per hop, so this is an option, not the default."""
        self.flat, self.group, self.min_elems = flat_grad, group, min_elems
        self.comm_dtype = comm_dtype if comm_dtype not in (None, flat_grad.dtype) else None
        self.done = flat_grad.numel()
        self.handles = []

    def ready(self, lo, force=False):
        import torch.distributed as dist
        lo = max(0, int(lo))
        if lo >= self.done or (self.done - lo < self.min_elems and not force and lo > 0):
            return
        view = self.flat[lo:self.done]
        buf = view.to(self.comm_dtype) if self.comm_dtype is not None else view
        h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.handles.append((h, view, buf))
        self.done = lo

    def finish(self):
        self.ready(0, force=True)
        for h, view, buf in self.handles:
            h.wait()
            if buf is not view:
                view.copy_(buf)
        self.handles = []
        self.done = self.flat.numel()


class ArenaAdam:
    def __init__(self, arena, lr_fn, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_grad_norm=0.0):
        self.arena = arena
        self.lr_fn = lr_fn
        self.betas, self.eps, self.wd, self.clip = betas, eps, weight_decay, clip_grad_norm
        self.m = torch.zeros_like(arena.flat)
        self.v = torch.zeros_like(arena.flat)
        self.nsq = torch.zeros(1, device=arena.flat.device, dtype=torch.float32)
        self.skipped = torch.zeros(1, device=arena.flat.device, dtype=torch.int32)  # steps the kernel skipped (NaN / Inf norm)
        self._step = 0
        self.lr = 0.0

    def zero_grad(self):
        self.arena.grad.zero_()

    def allreduce(self, group=None):
        allreduce_sum_(self.arena.grad, group)

    def step(self, grad_mult=1.0):
        """grad_mult: 1/world_size after a sum all-reduce (and/or 1/accum if not folded in the loss)."""
        self._step += 1
        self.lr = self.lr_fn(self._step)
        self.nsq.zero_()
        from . import ops
        ops.sqnorm(self.arena.grad, self.nsq)
        ops.adam_step(self.arena.flat, self.arena.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                      self.eps, self.wd, self._step, gnorm_sq=self.nsq, clip=self.clip, grad_mult=grad_mult,
                      skipped=self.skipped)

    def fold_skipped(self):
        """Host synchronisation point: take the steps the kernel skipped since the last call out of the step counter
        (train_asr.py:88-91 does not call optimizer.step() on a NaN gradient norm, so neither the bias correction nor a
        schedule position advance) -> number of skipped steps."""
        n = int(self.skipped.item())
        if n:
            self._step -= n
            self.skipped.zero_()
        return n

    def state_dict(self):
        self.fold_skipped()
        return {"_step": self._step, "m": self.m, "v": self.v, "lr": self.lr}

    def load_state_dict(self, sd):
        self._step = sd["_step"]
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])


def _world(group=None):
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank_dropout_seed(model, group=None):
    """Offset the engine's dropout seed by this process's rank (idempotent).  Call it once the process group exists and
    before the first training forward; train_step / train do."""
    import torch.distributed as dist
    eng = model.engine() if hasattr(model, "engine") else None
    if eng is not None and _world(group) > 1 and not getattr(eng, "_seed_ranked", False):
        eng.seed += 7919 * dist.get_rank(group)
        eng._seed_ranked = True


def train_step(model, optimizer, data, params, device, no_grad=False, empty_cache=False, group=None, sync=True,
               specaug=None):
    """One micro-batch of asr/train_asr.py:35-97: forward, loss / accum_grad, backward; unless `no_grad`
    (= still accumulating), clip to params.clip_grad_norm, skip the update on a NaN gradient norm, step,
    zero_grad.  -> loss_dict of floats divided by accum_grad (`sync=False`: 0-dim device tensors, no
    host synchronisation).

    With `emoasr_amd.optimizers.Adam` underneath, norm / clip / NaN-skip / update are the fused HIP step
    (no host round trip, so the reference's "do not update because of nan grad_norm" warning is not
    logged); with any torch optimizer the reference's sequence runs literally.  One process per GPU: the
    gradient arena is summed over the ranks once per optimizer step and scaled by 1/world, which is
    nn.DataParallel's mean of replica losses (train_asr.py:67-71).

    `specaug`: a data.SpecAugment; the batch is masked on the device before the forward (the reference masks each
    utterance in its data loader, asr/datasets.py:94-95)."""
    import math

    from .optimizers import Adam as HipAdam
    to = lambda k: data[k].to(device) if k in data else None
    xs = to("xs")
    if specaug is not None:
        xs = specaug(xs.float(), data["xlens"])
    world = _world(group)
    if world > 1:
        # one process per GPU: every replica gets its own dropout stream (masks are a pure hash of seed / site / element
        # index; identical seeds would drop identical positions on all ranks, unlike nn.DataParallel's replicas).  Applied
        # BEFORE the first forward, once per engine.
        rank_dropout_seed(model, group)
    loss, loss_dict = model(xs=xs, xlens=data["xlens"], ys=data["ys"], ylens=data["ylens"], ys_in=data["ys_in"],
                            ys_out=data["ys_out"], soft_labels=to("soft_labels"), ps=data.get("ps"),
                            plens=data.get("plens"))
    accum = params.accum_grad
    if sync:
        loss_dict = {k: v.item() / accum for k, v in loss_dict.items()}
    else:
        loss_dict = {k: v.detach() / accum for k, v in loss_dict.items()}
    (loss / accum).backward()
    if not no_grad:
        base = getattr(optimizer, "optimizer", optimizer)
        if isinstance(base, HipAdam):
            if world > 1:
                allreduce_sum_(model.engine().arena.grad, group)
            base.clip_grad_norm, base.grad_mult = params.clip_grad_norm, 1.0 / world
            optimizer.step()
        else:
            if world > 1:
                for p in model.parameters():
                    if p.grad is not None:
                        allreduce_sum_(p.grad, group)
                        p.grad.div_(world)
            grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), params.clip_grad_norm)
            if math.isnan(grad_norm):
                import logging
                logging.warning("do not update because of nan grad_norm")
            else:
                optimizer.step()
        optimizer.zero_grad()
        if empty_cache:
            torch.cuda.empty_cache()
    return loss_dict


def stacked_ok(model, optimizer, params):
    """can the accum_grad micro-batches of one optimizer step go through the engine TOGETHER (train_group)?
    -> "ctc" (encoder + CTC head + lattices stacked: engine.ctc_train_stacked), "encoder" (the encoder stacked, any decoder per
    micro-batch on its slice of the stacked output) or False"""
    from .optimizers import Adam as HipAdam
    from . import lib
    base = getattr(optimizer, "optimizer", optimizer)
    if not isinstance(base, HipAdam) or not hasattr(model, "engine"):
        return False
    if not (1 < params.accum_grad <= lib.MAX_SEGMENTS) or not model.training:
        return False
    if not next(model.parameters()).is_cuda:
        return False
    eng = model.engine()
    if not eng.encoder_stacked_ok():
        return False
    dec = model.decoder
    plain_ctc = (getattr(model, "decoder_type", "") == "ctc" and not (dec.kd_weight > 0 or dec.mtl_phone_ctc_weight > 0 or
                                                                      dec.mtl_inter_ctc_weight > 0))
    if plain_ctc and eng.stacked_ok():
        return "ctc"
    if getattr(model, "decoder_type", "") == "ctc":
        return False   # (auxiliary CTC branches read the encoder's intermediate output: one-by-one passes)
    return "encoder"


def train_group(model, optimizer, datas, params, device, group=None, sync=True, specaug=None, empty_cache=False):
    """The `accum_grad` micro-batches of ONE optimizer step (asr/train_asr.py:106-128: forward, loss / accum_grad, backward for
    each, then clip / NaN skip / step / zero_grad) as one stacked pass through the engine (engine.ctc_train_stacked: every
    row-wise kernel runs once over all micro-batches' rows; attention, convolution padding and BatchNorm statistics stay per
    micro-batch).  Models with another decoder stack the ENCODER only (modeling/functions.py: encoder_apply_stacked) and run the
    decoder per micro-batch on its slice.  Same result as len(datas) calls of train_step up to summation order; needs
    stacked_ok(...).  -> list of the micro-batches' loss_dicts (values / accum_grad, as train_step returns them)."""
    world = _world(group)
    if world > 1:
        rank_dropout_seed(model, group)
    eng = model.engine()
    accum = params.accum_grad
    mode = "ctc" if (getattr(model, "decoder_type", "") == "ctc" and eng.stacked_ok()) else "encoder"
    xs_list, xl_list = [], []
    for data in datas:
        xs = data["xs"].to(device)
        xlens = [int(v) for v in data["xlens"]]
        if specaug is not None:
            xs = specaug(xs.float(), data["xlens"])
        xs_list.append(xs[:, : max(xlens)].to(torch.float32).contiguous())
        xl_list.append(xlens)
    if mode == "ctc":
        batches = []
        for xs, xlens, data in zip(xs_list, xl_list, datas):
            ylens = [int(v) for v in data["ylens"]]
            batches.append((xs, xlens, data["ys"][:, : max(ylens)], ylens))
        losses = eng.ctc_train_stacked(batches, model.decoder.blank_id, scales=[1.0 / accum] * len(batches))
        if sync:
            host = (losses / accum).tolist()
            dicts = [{"loss_ctc": v, "loss_total": v} for v in host]
        else:
            dicts = [{"loss_ctc": losses[k] / accum, "loss_total": losses[k] / accum} for k in range(len(batches))]
    else:
        from .modeling.functions import encoder_apply_stacked
        outs = encoder_apply_stacked(model.encoder, xs_list, xl_list)
        total, dicts = None, []
        # a transducer's prediction network reads the labels only: all micro-batches in one pass (None: not applicable)
        preds = None
        if hasattr(model.decoder, "prediction_stacked") and all(d.get("ys_in") is not None for d in datas):
            preds = model.decoder.prediction_stacked([d["ys_in"] for d in datas], [d["ylens"] for d in datas])
        for k, ((eouts, elens, _), data) in enumerate(zip(outs, datas)):
            ymax = int(max(data["ylens"]))   # the targets are trimmed to the batch as ASR.forward does (asr.py:57-62)
            ys = data["ys"][:, :ymax]
            ys_in = data["ys_in"][:, : ymax + 1] if data.get("ys_in") is not None else None
            ys_out = data["ys_out"][:, : ymax + 1] if data.get("ys_out") is not None else None
            soft = data["soft_labels"].to(device) if data.get("soft_labels") is not None else None
            ps, plens = data.get("ps"), data.get("plens")
            if ps is not None:
                ps = ps[:, : int(max(plens))]
            extra = {} if preds is None else {"pred": preds[k]}
            loss, loss_dict, _ = model.decoder(eouts, elens, None, ys, data["ylens"], ys_in, ys_out, soft, ps, plens, **extra)
            total = loss / accum if total is None else total + loss / accum
            dicts.append({k: (v.item() / accum if sync else v.detach() / accum) for k, v in loss_dict.items()})
        total.backward()
    base = getattr(optimizer, "optimizer", optimizer)
    if world > 1:
        allreduce_sum_(eng.arena.grad, group)
    base.clip_grad_norm, base.grad_mult = params.clip_grad_norm, 1.0 / world
    optimizer.step()
    optimizer.zero_grad()
    if empty_cache:
        torch.cuda.empty_cache()
    return dicts


def train(model, optimizer, dataloader, params, device, epoch, empty_cache=False, group=None, log=None, stacked="auto"):
    """One epoch of asr/train_asr.py:100-143: every accum_grad-th micro-batch steps the optimizer; the
    running loss_dict sums are logged every params.log_step optimizer steps (the only host
    synchronisation of the loop).

    stacked: "auto" (default; also params.stacked when present) sends the accum_grad micro-batches of an optimizer step through the
    engine together whenever stacked_ok(...) allows it (activation memory is accum_grad times one micro-batch's; dropout masks and
    the rounding of the BatchNorm running variance differ from the one-by-one passes in the last bits); False keeps the
    reference's one-by-one loop; True asserts that the stacked path is taken.  A group that runs out of device memory is re-run
    one by one from zeroed gradients (the BatchNorm running statistics may then have moved once more for that group)."""
    import logging
    from .hostenv import respect_cpu_quota
    respect_cpu_quota()   # (an oversized CPU pool under a cgroup quota freezes the launching thread for 20-50 ms at a time)
    log = log or logging.info
    optimizer.update_epoch()
    step, sums = 0, {}
    specaug = None
    if getattr(params, "spec_augment", False):  # asr/datasets.py:37-38
        from .data import SpecAugment
        specaug = SpecAugment(params)
    n_total = len(dataloader) // params.accum_grad if hasattr(dataloader, "__len__") else -1
    # the accum_grad micro-batches of an optimizer step go through the engine together when the model allows it (a trailing
    # incomplete group is accumulated and never stepped, exactly as the one-by-one loop leaves it)
    choice = getattr(params, "stacked", stacked)
    assert choice in ("auto", True, False), "train: stacked must be 'auto', True or False"
    stacked = stacked_ok(model, optimizer, params) if choice in ("auto", True) else False
    assert stacked or choice is not True, "train(stacked=True): this model / optimizer cannot take the stacked path (stacked_ok)"
    pending = []
    for accum_step, data in enumerate(dataloader):
        stepping = (accum_step + 1) % params.accum_grad == 0
        if stacked:
            pending.append(data)
            if not stepping:
                continue
            try:
                dicts = train_group(model, optimizer, pending, params, device, group=group, sync=False, specaug=specaug,
                                    empty_cache=empty_cache)
            except torch.cuda.OutOfMemoryError:
                # the stacked pass holds accum_grad micro-batches' activations at once: drop what it had allocated and whatever
                # it had accumulated, then run this group (and the rest of the epoch) the reference's way
                log("train: the stacked pass ran out of device memory; continuing with one-by-one micro-batches")
                model.engine().arena.grad.zero_()
                torch.cuda.empty_cache()
                stacked = False
                dicts = [train_step(model, optimizer, d, params, device, no_grad=k + 1 < len(pending), group=group,
                                    sync=False, specaug=specaug) for k, d in enumerate(pending)]
            pending = []
        else:
            dicts = [train_step(model, optimizer, data, params, device, no_grad=not stepping,
                                empty_cache=empty_cache and stepping, group=group, sync=False, specaug=specaug)]
        step += int(stepping)
        for loss_dict in dicts:
            for k, v in loss_dict.items():
                sums[k] = sums[k] + v if k in sums else v
        if stepping and step % params.log_step == 0:
            from . import lib
            if lib.size_query("emoasr_lstm_coop_status") > 0:   # (this is the loop's host synchronisation point anyway; -1: no device)
                raise RuntimeError("lstm_coop: a grid barrier gave up waiting (csrc/lstm_coop.hip); emoasr_set_option('lstm_coop', 0) "
                                   "selects the per-position launch chain")
            if hasattr(optimizer, "fold_skipped"):   # skipped (NaN) updates leave the schedule position here, not only at epoch ends
                optimizer.fold_skipped()
            detail = " ".join(f"{k}: {float(v) / params.log_step:.3f}" for k, v in sums.items())
            log(f"epoch = {(epoch + 1):>2} step = {step:>6} / {n_total:>6} lr = {optimizer._lr:.5f} " + detail)
            sums = {}
    for data in pending:   # an incomplete last group: gradients accumulate, no update (train_asr.py:106-128)
        loss_dict = train_step(model, optimizer, data, params, device, no_grad=True, group=group, sync=False, specaug=specaug)
        for k, v in loss_dict.items():   # (the one-by-one loop adds these micro-batches' losses to the running sums too)
            sums[k] = sums[k] + v if k in sums else v
    return step
