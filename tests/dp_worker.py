"""Child process of tests/test_00_dp_two_process_gpu.py: one data-parallel rank on cuda:0 (two ranks share the one GPU of the
test box; the collective backend is gloo, the arena / bucket / optimizer code is exactly what runs over RCCL on 8 GPUs).

usage: python -m tests.dp_worker RANK WORLD PORT OUT_JSON"""
import json
import os
import sys
from types import SimpleNamespace

import torch
import torch.distributed as dist


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from emoasr_amd.modeling.asr import ASR
    from emoasr_amd.optimizers import Adam, ScheduledOptimizer
    from emoasr_amd.train import GradBuckets, rank_dropout_seed, train_step
    from tests.util import CONFIGS, load_golden
    dev = torch.device("cuda:0")
    _, sd, g = load_golden("l2_tiny")
    cfg = dict(CONFIGS["l2_tiny"], dropout_enc_rate=0.1, dropout_attn_rate=0.1, lr_schedule_type="noam", learning_rate=0.02,
               num_warmup_steps=4, accum_grad=1, clip_grad_norm=5.0, weight_decay=1e-6, log_step=100)
    params = SimpleNamespace(**cfg)
    model = ASR(params, compute_dtype=torch.bfloat16)
    model.load_state_dict(sd)
    optimizer = ScheduledOptimizer(Adam(model.parameters(), lr=0, weight_decay=params.weight_decay), params)
    model = model.to(dev).train()
    eng = model.engine()
    seed0 = eng.seed
    rank_dropout_seed(model)
    assert eng.seed == seed0 + 7919 * rank, (eng.seed, seed0)   # applied before the first forward, once
    rank_dropout_seed(model)
    assert eng.seed == seed0 + 7919 * rank
    # this rank's shard: utterances rank, rank + world, ... of the golden batch
    B = g["xs"].shape[0]
    idx = list(range(rank, B, world))
    data = {k: g[k][idx] for k in ("xs", "xlens", "ys", "ylens", "ys_in", "ys_out")}

    def fwd_bwd(with_buckets):
        eng.step_count = 3                  # the same dropout masks in both runs (they are a hash of seed / step / element)
        eng.arena.grad.zero_()
        buckets = None
        if with_buckets:
            buckets = GradBuckets(eng.arena.grad, min_elems=1 << 12)
            eng.grad_hook = buckets.ready
        loss, _ = model(data["xs"].to(dev), data["xlens"], data["ys"], data["ylens"], data["ys_in"], data["ys_out"])
        loss.backward()
        n = 0
        if buckets is not None:
            n = len(buckets.handles)
            buckets.finish()
            eng.grad_hook = None
        torch.cuda.synchronize()
        return eng.arena.grad.clone(), n

    local, _ = fwd_bwd(False)
    reduced, n_async = fwd_bwd(True)
    assert n_async >= 1, "no gradient range went out during the backward sweep"
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    total = torch.stack(gathered).sum(0)
    scale = total.abs().max().item()
    assert scale > 0
    # bucketed all-reduce through the engine's arena == sum of the ranks' own gradients (weight gradients accumulate with
    # float atomics: two runs of the same rank differ in the last bits)
    err = (reduced - total).abs().max().item() / scale
    assert err < 1e-4, err
    differ = (gathered[0] - gathered[1]).abs().max().item() / scale
    assert differ > 1e-3, differ           # the ranks really worked on different utterances / dropout masks
    # three optimizer steps through train_step: all-reduce, 1 / world inside the fused Adam, identical parameters everywhere
    eng.arena.grad.zero_()
    optimizer.update_epoch()
    for _ in range(3):
        train_step(model, optimizer, data, params, dev)
    flat = eng.arena.flat.clone()
    everyone = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(everyone, flat)
    for r in range(1, world):
        assert torch.equal(everyone[0], everyone[r]), f"parameters of rank {r} differ from rank 0 after 3 steps"
    # ---- the stacked path (accum_grad micro-batches through the engine in one pass) under the same two ranks: a d = 256 model
    #      (the large-tile front-end kernels want 256 channels), identical initial weights, different micro-batches per rank
    from emoasr_amd.train import stacked_ok, train_group
    cfg2 = dict(input_layer="conv2d", feat_dim=80, num_framestacks=1, encoder_type="conformer", decoder_type="ctc",
                pos_encode_type="rel", enc_hidden_size=256, enc_num_attention_heads=4, enc_num_layers=2,
                enc_intermediate_size=512, dropout_enc_rate=0.1, dropout_attn_rate=0.1, vocab_size=96, blank_id=0, eos_id=2,
                kd_weight=0, lr_schedule_type="noam", learning_rate=1.0, num_warmup_steps=10, accum_grad=2, clip_grad_norm=5.0,
                weight_decay=1e-6, log_step=100)
    p2 = SimpleNamespace(**cfg2)
    torch.manual_seed(0)
    m2 = ASR(p2, compute_dtype=torch.bfloat16)
    o2 = ScheduledOptimizer(Adam(m2.parameters(), lr=0, weight_decay=p2.weight_decay), p2)
    m2 = m2.to(dev).train()
    assert stacked_ok(m2, o2, p2) == "ctc"
    gen = torch.Generator().manual_seed(100 + rank)

    def micro(xlens):
        xl = torch.tensor(xlens)
        yl = torch.clamp(xl // 40, min=1)
        xs = torch.randn(len(xlens), int(xl.max()), 80, generator=gen)
        ys = torch.randint(3, 96, (len(xlens), int(yl.max())), generator=gen)
        return dict(xs=xs, xlens=xl, ys=ys, ylens=yl, ys_in=None, ys_out=None)

    o2.update_epoch()
    for _ in range(2):
        train_group(m2, o2, [micro([203, 150, 96]), micro([303, 280])], p2, dev)
    flat2 = m2.engine().arena.flat.clone()
    both = [torch.zeros_like(flat2) for _ in range(world)]
    dist.all_gather(both, flat2)
    for r in range(1, world):
        assert torch.equal(both[0], both[r]), f"stacked path: parameters of rank {r} differ from rank 0 after 2 steps"
    assert torch.isfinite(flat2).all()
    if rank == 0:
        with open(out, "w") as f:
            json.dump({"ok": True, "allreduce_err": err, "rank_grad_difference": differ, "async_ranges": n_async,
                       "step": optimizer._step, "stacked_step": o2._step}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
