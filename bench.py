#!/usr/bin/env python
"""Headline benchmark: training frames/s (and greedy-decode RTF) of the 23M Conformer-CTC (`L2`)
on synthetic LibriSpeech-shaped batches, 80-dim features resident in HBM, SpecAugment on GPU,
bf16 MFMA compute, fused Adam, one RCCL all-reduce of the flat gradient arena per step.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
`roofline` (dominant kernel family: algorithmic FLOPs / HIP-event time over the timed region)
and `cpu_baseline` (the CPU oracle, oracle/model.py, timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import random
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L2 = dict(input_layer="conv2d", feat_dim=80, num_framestacks=1, encoder_type="conformer", decoder_type="ctc",
          pos_encode_type="rel", enc_hidden_size=256, enc_num_attention_heads=4, enc_num_layers=12,
          enc_intermediate_size=1024, dropout_enc_rate=0.1, dropout_attn_rate=0.1, vocab_size=10000, blank_id=0,
          eos_id=2, kd_weight=0)
OPT = dict(lr=5.0, warmup=25000, weight_decay=1e-6, clip_grad_norm=5.0)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def fwd_flops_per_utt(T, F=1024, V=10000, d=256, layers=12):
    """closed form of SURVEY.md section 8d (exactly torch.utils.flop_counter on the reference)"""
    T1 = (T - 1) // 2
    Tp = (T1 - 1) // 2
    conv1 = 2 * 9 * d * T1 * 39
    conv2 = 2 * 9 * d * d * Tp * 19
    lin = 2 * (d * 19) * d * Tp
    layer = (2 * (4 * d * F) * Tp + 8 * d * d * Tp + 2 * d * d * (2 * Tp - 1) + 2 * d * Tp * Tp +
             2 * d * Tp * (2 * Tp - 1) + 2 * d * Tp * Tp + 4 * d * d * Tp + 2 * 31 * d * Tp + 2 * d * d * Tp)
    head = 2 * d * V * Tp
    return conv1 + conv2 + lin + layers * layer + head


PMC_FILE = os.path.join("profiles", "r06_pmc.json")


def pmc_kernel(kernel, key):
    """per-launch figure of one kernel from the committed PMC passes (separate rocprofv3 --pmc runs of THIS bench command,
    summarised by tools/pmc_summary.py): key = "traffic_bytes" (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of
    /opt/skills/guides/MI355X_MICROARCH.md) or "mfma_util" (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CU cycles the kernel was
    resident)).  Counters cannot be collected inside a timed run, so these two figures are NOT live: they are None unless the
    profile file of this round exists, and the bench line names the file they came from (`pmc_source`)."""
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        fams = json.load(f).get("kernels", {})
    hit = [v for k, v in fams.items() if k == kernel] or [v for k, v in fams.items() if kernel in k]
    if not hit or key not in hit[0]:
        return None
    n = sum(v["launches"] for v in hit)
    return sum(v[key] * v["launches"] for v in hit) / max(n, 1)


def pmc_source():
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        meta = json.load(f)
    return {"file": PMC_FILE, "commit": meta.get("commit"), "command": meta.get("command")}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# kernel families timed inside the library (emoasr_timer_read_ex); value: the kernel-table symbol(s) of profiles/*_kernel_stats.csv
FAMILIES = {"gemm_nt_nn": "gemm_nt_kernel / gemm_nn (big_nt_kernel for wide products): forward and data-gradient products",
            "gemm_tn": "gemm_tn_grouped_kernel / gemm_tn_kernel: weight gradients",
            "attn_bwd_fused_kernel": "attn_bwd_kv_kernel + attn_bwd_q_kernel (the two-pass backward; the family keeps its round-2 timer name; the keep-mask bits are hashed up front by attn_dropmask_kernel on the side stream)", "attn_bwd_dpos_kernel": "attn_bwd_dpos3_kernel (position-table gradient, side stream)",
            "attn_fwd_kernel": "attn_fwd_kernel", "layernorm": "ln_fwd_kernel + ln_bwd8_kernel",
            "conv_module": "cf_dwconv / bn_* / cf_conv_bwd kernels (convolution module, per-utterance part)"}
RIDGE_FLOP_PER_BYTE = 2500e12 / 8000e9  # bf16 dense MFMA peak / HBM peak


def family_table(emo_lib, attn_work, elapsed_s=None):
    """{family: calls, ms, algorithmic GFLOP / GB, achieved TFLOP/s and GB/s, fractions of both peaks} from the library's timers.
    attn_work: (fwd flops, bwd-main flops, dpos flops) of the recorded steps -- the attention kernels' algorithmic work depends on
    the utterance lengths, which only the caller knows."""
    out = {}
    for name in FAMILIES:
        calls, ms, fl, by = emo_lib.timer_read_ex(name)
        if not calls:
            continue
        if name == "attn_fwd_kernel":
            fl = attn_work[0]
        elif name == "attn_bwd_fused_kernel":
            fl = attn_work[1]
        elif name == "attn_bwd_dpos_kernel":
            fl = attn_work[2]
        row = {"calls": calls, "ms": ms, "avg_us": 1e3 * ms / calls}
        if fl:
            row["gflop"] = fl / 1e9
            row["tflops"] = fl / (ms * 1e-3) / 1e12
            row["mfma_frac"] = row["tflops"] / MFMA_PEAK_TFLOPS["bf16"]
        if by:
            row["gbytes"] = by / 1e9
            row["gbps"] = by / (ms * 1e-3) / 1e9
            row["hbm_frac"] = row["gbps"] / HBM_PEAK_GBS
        if elapsed_s:
            row["share_of_step"] = ms * 1e-3 / elapsed_s
        # HBM bytes per launch from the committed counter passes (not live: see pmc_kernel): what the family ACTUALLY moved against
        # its algorithmic bytes, and the bandwidth that traffic amounts to at this run's launch time
        tr = pmc_kernel(name, "traffic_bytes")
        if tr:
            row["pmc_traffic_MB"] = tr / 1e6
            row["pmc_traffic_gbps"] = tr / (1e-3 * ms / calls) / 1e9
            if by:
                row["traffic_over_algorithmic"] = tr / (by / calls)
        out[name] = row
    return out


class CallTimer:
    """HIP-event timing of selected C-ABI entry points on the stream they are launched on."""

    def __init__(self, lib_mod, names=None):
        self.lib, self.names, self.rec = lib_mod, names, {}
        self.attn_pairs = 0.0  # sum over the batch of (valid queries x valid keys), set per step by the caller
        self._orig = lib_mod.call

    def __enter__(self):
        def call(name, *args):
            if self.names is not None and name not in self.names:
                return self._orig(name, *args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._orig(name, *args)
            e1.record()
            flops = None
            if name in ("emoasr_gemm_nt", "emoasr_gemm_nn", "emoasr_gemm_tn"):
                flops = 2.0 * args[1] * args[2] * args[3]
            elif name in ("emoasr_conv2_fwd", "emoasr_conv2_wgrad"):
                B, T1, F1, C = args[1], args[2], args[3], args[4]
                flops = 2.0 * B * ((T1 - 3) // 2 + 1) * ((F1 - 3) // 2 + 1) * C * 9 * C
            elif name in ("emoasr_attn_fwd", "emoasr_attn_bwd") and self.attn_pairs:
                # algorithmic matmul count over the VALID (query, key) pairs only (padding excluded):
                # fwd: Q K^T, Q pos^T (band), P V; bwd adds the recomputation of both score products plus
                # dP = dO V^T, dV = P^T dO, dQ = dS K (+ dS pos), dK = dS^T Q, dpos = dS^T Q
                a = args[1]._obj
                relpos = bool(a.pos)
                nmm = ((3 if relpos else 2) if name == "emoasr_attn_fwd" else (8 if relpos else 5))
                flops = 2.0 * nmm * a.H * a.DK * self.attn_pairs
            self.rec.setdefault(name, []).append((e0, e1, flops))
        self.lib.call = call
        import emoasr_amd.ops as ops_mod
        ops_mod.lib.call = call
        return self

    def __exit__(self, *a):
        self.lib.call = self._orig

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.rec.items():
            ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
            fl = sum(f for _, _, f in evs if f is not None)
            out[name] = dict(calls=len(evs), ms=ms, flops=fl)
        return out


def make_batches(rank, world, need, dev, seed=0):
    from emoasr_amd.data import libri_shaped_lengths, pack_batches
    xlens, ylens = libri_shaped_lengths(2000, seed)
    batches = pack_batches(xlens, ylens, 30000, 3000, 50, 1)
    order = list(range(len(batches)))
    random.Random(seed).shuffle(order)
    mine = [batches[order[(i * world + rank) % len(order)]] for i in range(need)]
    g = torch.Generator().manual_seed(seed * 1000 + rank)
    out = []
    for idx in mine:
        xl, yl = xlens[idx], ylens[idx]
        B, T, L = len(idx), int(xl.max()), int(yl.max())
        xs = torch.randn(B, T, 80, generator=g)
        ys = torch.randint(3, L2["vocab_size"], (B, L), generator=g)
        for b in range(B):
            xs[b, xl[b]:] = 0
            ys[b, yl[b]:] = L2["eos_id"]
        out.append(SimpleNamespace(xs=xs.to(dev), xlens=[int(v) for v in xl], ys=ys, ylens=[int(v) for v in yl]))
    return out


def cpu_baseline(model, max_seconds=14.0):
    """CPU oracle (oracle/model.py, fp32 torch eager on the host cores), bounded samples of the same workload:
    all threads -- fwd + bwd of one 4-utterance L2 batch (the headline `value`); one thread (the reference's RTF protocol,
    asr/test_asr.py:227) -- fwd + bwd of one 640-frame utterance and greedy CTC decoding of two utterances."""
    from oracle import model as om
    cfg = SimpleNamespace(**dict(L2, dropout_enc_rate=0.0, dropout_attn_rate=0.0))
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k]
    g = torch.Generator().manual_seed(0)
    xlens = torch.tensor([1200, 1037, 911, 640])
    ylens = torch.tensor([40, 33, 29, 20])
    xs = torch.randn(4, 1200, 80, generator=g)
    ys = torch.randint(3, 10000, (4, 40), generator=g)
    for b in range(4):
        xs[b, xlens[b]:] = 0

    # the same step on both sides (asr/train_asr.py:35-97): forward + backward + global-norm clip + Adam
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9)

    def train_rate(xs_, xl_, ys_, yl_, budget, max_steps):
        steps, t_total = 0, 0.0
        for i in range(max_steps + 1):
            t0 = time.perf_counter()
            loss, _, _ = om.asr_ctc_forward(sd, cfg, xs_, xl_, ys_, yl_, training=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 5.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            dt = time.perf_counter() - t0
            if i == 0 and max_steps > 1:
                continue  # warm-up
            steps += 1
            t_total += dt
            if t_total > budget:
                break
        return int(xl_.sum()) * steps / t_total, steps, t_total

    nthr = torch.get_num_threads()
    rate, steps, t_total = train_rate(xs, xlens, ys, ylens, max_seconds, 24)   # (about 12 s of host work on the GPU box)
    out = dict(value=rate, unit="frames/s", cores=nthr, kind="port", cpu_model=cpu_model(),
               host_cores=os.cpu_count(),
               sample=f"{steps} fwd+bwd+clip+Adam steps of one L2 batch (4 utts, xlens 1200/1037/911/640, fp32, dropout 0) in "
                      f"{t_total:.1f}s on {nthr} torch threads = the container's CPU quota on this host ({cpu_model()}, "
                      f"{os.cpu_count()} logical CPUs visible); BASELINE.md asks for n in {{1, all physical cores}}: the "
                      f"1-thread figure is value_1thread, all physical cores are not available to this process")
    torch.set_num_threads(1)
    try:
        r1, s1, t1 = train_rate(xs[3:4, :640], xlens[3:4], ys[3:4, :20], ylens[3:4], 6.0, 1)
        out["value_1thread"] = r1
        out["sample_1thread"] = f"{s1} fwd+bwd+clip+Adam step(s) of one 640-frame utterance in {t1:.1f}s on 1 thread"
        t0 = time.perf_counter()
        with torch.no_grad():
            for b in (2, 3):
                om.asr_ctc_greedy(sd, cfg, xs[b:b + 1, : int(xlens[b])], xlens[b:b + 1])
        out["decode_rtf_1thread"] = (time.perf_counter() - t0) / (float(xlens[2] + xlens[3]) * 0.010)
    finally:
        torch.set_num_threads(nthr)
    return out


class _NpyLoader:
    """the reference's decode-time data path: TSV manifest + one .npy feature file per utterance, batch size 1, read from
    disk on every pass (asr/datasets.py:25-177 via emoasr_amd.datasets.ASRDataset)"""

    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def __iter__(self):
        for i in range(len(self.ds)):
            yield self.ds.collate_fn([self.ds[i]])


def rtf_fixture(tmpdir, n_utts, seed, feat_dim=80):
    """n_utts synthetic LibriSpeech-length utterances as .npy files + manifest + vocabulary -> (loader, vocab, ylens)"""
    from emoasr_amd.data import libri_shaped_lengths
    from emoasr_amd.datasets import ASRDataset, Vocab
    xlens, ylens = libri_shaped_lengths(2000, 0)
    rs = np.random.RandomState(seed)
    pick = rs.choice(len(xlens), n_utts, replace=False)
    os.makedirs(tmpdir, exist_ok=True)
    lines = ["feat_path\tutt_id\ttoken_id\ttext\txlen\tylen"]
    for k, i in enumerate(pick):
        fp = os.path.join(tmpdir, f"utt{k}.npy")
        np.save(fp, rs.randn(int(xlens[i]), feat_dim).astype(np.float32))
        lines.append(f"{fp}\tutt{k}\t3 4 5\tref\t{int(xlens[i])}\t{int(ylens[i])}")
    tsv = os.path.join(tmpdir, "test.tsv")
    with open(tsv, "w") as f:
        f.write("\n".join(lines) + "\n")
    vp = os.path.join(tmpdir, "vocab.txt")
    with open(vp, "w") as f:
        f.write("<blank> 0\n<unk> 1\n<eos> 2\n" + "".join(f"\u2581w{i} {i}\n" for i in range(3, 10000)))
    ds = ASRDataset(SimpleNamespace(feat_dim=feat_dim, eos_id=2), tsv, phase="test")
    return _NpyLoader(ds), Vocab(vp), [int(ylens[i]) for i in pick]


def decode_rtf(model, dev, tmpdir):
    """greedy CTC decode with the reference's RTF protocol (asr/test_asr.py:226-263 = emoasr_amd.decode.measure_rtf): 20
    utterances, batch 1, feature files read from disk inside the timed region, mean of 5 repeats"""
    from emoasr_amd import decode as dec
    loader, vocab, _ = rtf_fixture(os.path.join(tmpdir, "greedy"), 20, 1)
    model.eval()
    dec.test(model, loader, vocab, 1, 0.0, 0.0, False, None, 0.0, dev, num_samples=3)  # warm-up
    _, rtf = dec.measure_rtf(model, loader, vocab, 1, 0.0, 0.0, False, None, 0.0, dev, num_samples=20, num_repeats=5)
    model.train()
    return rtf


def ctc_beam_rtf(model, dev, dtype, tmpdir, n_utts=3):
    """CTC prefix beam search with LM shallow fusion for the CTC model itself (asr/modeling/decoders/ctc.py:203-344): beam 10,
    12-layer Transformer LM at weight 0.3, batch 1, the reference's RTF protocol on a few utterances (random-init weights: a
    full beam of distinct prefixes at every frame).  Acoustic side on the GPU, one LM call per NEW prefix (rows cached on the device),
    the per-frame bookkeeping in native host code (csrc/ctc_beam_host.hip)."""
    import logging
    from emoasr_amd import decode as dec
    from emoasr_amd.modeling.lm import LM
    logging.disable(logging.WARNING)
    torch.manual_seed(3)
    lm = LM(SimpleNamespace(**LM12), compute_dtype=dtype).to(dev).eval()
    loader, vocab, _ = rtf_fixture(os.path.join(tmpdir, "ctcbeam"), n_utts, 1)
    model.eval()
    dec.test(model, loader, vocab, 10, 0.0, 0.0, False, lm, 0.3, dev, num_samples=1)  # warm-up
    _, rtf = dec.measure_rtf(model, loader, vocab, 10, 0.0, 0.0, False, lm, 0.3, dev, num_samples=n_utts, num_repeats=1)
    model.train()
    logging.disable(logging.NOTSET)
    return {"rtf": rtf, "beam": 10, "lm_weight": 0.3, "utts": n_utts}


def logmel_rate(dev, batch_xlens, repeats=3):
    """on-GPU Kaldi log-mel of one training batch's raw audio (N = 160 (T - 1) + 400 samples of N(0, 0.05) per
    utterance): feature frames per second of the fbank kernel alone"""
    from emoasr_amd.features import LogMel
    lm = LogMel(dev)
    g = torch.Generator().manual_seed(9)
    wavs = [(0.05 * torch.randn(160 * (int(t) - 1) + 400, generator=g)).to(dev) for t in batch_xlens]
    for w in wavs[:2]:
        lm(w)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(repeats):
        t0 = time.perf_counter()
        for w in wavs:
            lm(w)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return sum(int(t) for t in batch_xlens) / best


def decode_rtf_batched(model, dev, batch=32, repeats=3):
    """greedy CTC decode at a batched operating point: `batch` length-sorted neighbours per call"""
    from emoasr_amd.data import libri_shaped_lengths
    xlens, _ = libri_shaped_lengths(2000, 0)
    lens = [int(v) for v in xlens[1000:1000 + batch]]
    xs = torch.randn(batch, max(lens), 80)
    for b, n in enumerate(lens):
        xs[b, n:] = 0
    xs = xs.to(dev)
    model.eval()
    model.decode(xs, lens)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(repeats):
        t0 = time.perf_counter()
        model.decode(xs, lens)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (sum(lens) * 0.010))
    model.train()
    return best


L4 = dict(L2, decoder_type="rnn_transducer", vocab_size=1000, embedding_size=256, dec_hidden_size=512, dec_num_layers=2,
          joint_hidden_size=512, dropout_emb_rate=0.1, dropout_dec_rate=0.1, mtl_ctc_weight=0.3, lsm_prob=0.0)


def l4_rnnt(dev, dtype, steps=8, warmup=2, n_dec=5, accum=5):
    """config 5 (`L4`): RNN-T (Conformer) 26 M -- training frames/s (optimizer steps of `accum` micro-batches as the reference
    runs them: the ENCODER of all micro-batches in one stacked pass, the prediction network / joint / transducer lattice per
    micro-batch on its slice, loss / accum each, then Adam; LibriSpeech-shaped batches at the standard 30 000-frame budget: the
    joint logits [B,T',U+1,1000] are ~0.9 GB in bf16 per micro-batch) and streaming greedy decode RTF at batch 1 (the whole search
    of an utterance as one device-resident launch; decode steps capped by the model's own max-symbols rule)."""
    from emoasr_amd.data import libri_shaped_lengths, pack_batches
    from emoasr_amd.modeling.asr import ASR
    from emoasr_amd.modeling.functions import encoder_apply_stacked
    from emoasr_amd.train import ArenaAdam, noam_lr
    torch.manual_seed(2)
    model = ASR(SimpleNamespace(**L4), compute_dtype=dtype).to(dev).train()
    eng = model.engine()
    opt = ArenaAdam(eng.arena, lambda s: noam_lr(OPT["lr"], 256, OPT["warmup"], s), weight_decay=OPT["weight_decay"],
                    clip_grad_norm=OPT["clip_grad_norm"])
    xlens, ylens = libri_shaped_lengths(2000, 0)
    batches = pack_batches(xlens, ylens, 30000, 3000, 50, 1)  # the standard sampler budget (asr/datasets.py:159-213)
    rs = random.Random(3)
    rs.shuffle(batches)
    g = torch.Generator().manual_seed(5)
    stacked = accum > 1 and dtype == torch.bfloat16 and eng.encoder_stacked_ok()
    if not stacked:
        accum = 1
    data = []
    for idx in batches[: (steps + warmup) * accum]:
        xl, yl = [int(xlens[i]) for i in idx], [int(ylens[i]) for i in idx]
        xs = torch.randn(len(idx), max(xl), 80, generator=g)
        ys = torch.randint(3, L4["vocab_size"], (len(idx), max(yl)), generator=g)
        eos = torch.full((len(idx), 1), L4["eos_id"])
        for b in range(len(idx)):
            xs[b, xl[b]:] = 0
            ys[b, yl[b]:] = L4["eos_id"]
        data.append((xs.to(dev), xl, ys, yl, torch.cat([eos, ys], 1), torch.cat([ys, eos], 1)))
    groups = [data[i * accum:(i + 1) * accum] for i in range(steps + warmup)]

    def step(grp):
        opt.zero_grad()
        if stacked:
            outs = encoder_apply_stacked(model.encoder, [bt[0] for bt in grp], [bt[1] for bt in grp])
            # (as emoasr_amd/train.py: train_group) the prediction network of all micro-batches in one pass
            preds = model.decoder.prediction_stacked([bt[4] for bt in grp], [bt[3] for bt in grp])
            total = None
            for k, ((eouts, elens, _), bt) in enumerate(zip(outs, grp)):
                extra = {} if preds is None else {"pred": preds[k]}
                loss, _, _ = model.decoder(eouts, elens, None, bt[2], bt[3], bt[4], bt[5], None, None, None, **extra)
                total = loss / accum if total is None else total + loss / accum
            total.backward()
        else:
            for bt in grp:
                loss, _ = model(*bt)
                (loss / accum).backward()
        opt.step()
        return loss

    for grp in groups[:warmup]:
        step(grp)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for grp in groups[warmup:]:
        loss = step(grp)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    frames = sum(sum(bt[1]) for grp in groups[warmup:] for bt in grp)
    out = dict(train_frames_per_s=frames / el, ms_per_step=1e3 * el / steps, frames_per_step=frames / steps, accum_grad=accum,
               stacked_encoder=bool(stacked), params_M=sum(p.numel() for p in model.parameters()) / 1e6,
               final_loss=float(loss.detach()),
               # the output layer + transducer loss without the [B,T,U,V] logits (emoasr_rnnt_head_fwd / _grad)
               fused_output_layer=bool(model.engine().rnnt_fused), peak_mem_GiB=torch.cuda.max_memory_allocated() / 2 ** 30)
    model.eval()
    rs2 = np.random.RandomState(4)
    pick = rs2.choice(len(xlens), n_dec, replace=False)
    utts = [(torch.randn(1, int(xlens[i]), 80).to(dev), [int(xlens[i])]) for i in pick]
    model.decode(*utts[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x, l in utts:
        model.decode(x, l)
    torch.cuda.synchronize()
    out["greedy_rtf"] = (time.perf_counter() - t0) / (sum(l[0] for _, l in utts) * 0.010)
    # alignment-length synchronous beam search (rnn_transducer.py:242-325), beam 4, on the two shortest picks
    short = sorted(utts, key=lambda u: u[1][0])[:2]
    model.decode(*short[0], beam_width=4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x, l in short:
        model.decode(x, l, beam_width=4)
    torch.cuda.synchronize()
    out["beam4_rtf"] = (time.perf_counter() - t0) / (sum(l[0] for _, l in short) * 0.010)
    return out


L3 = dict(L2, decoder_type="transformer", dec_hidden_size=256, dec_num_attention_heads=4, dec_num_layers=6,
          dec_intermediate_size=1024, dropout_dec_rate=0.1, mtl_ctc_weight=0.3, lsm_prob=0.1,
          loss_normalize_length=False, loss_normalize_batch=True, max_decode_ylen=256)
LM12 = dict(lm_type="transformer", vocab_size=10000, hidden_size=256, num_layers=12, num_attention_heads=4,
            intermediate_size=1024, max_seq_len=256)


def decode_rtf_l33(dev, dtype, tmpdir, n_utts=20, repeats=5, out_steps=36, eos_biased=False):  # noqa: C901
    """config 4 (`L3-3`): joint CTC+attention beam 10 with Transformer-LM shallow fusion, batch 1, the reference's RTF
    protocol (20 utterances x 5 repeats, feature files read inside the timed region; emoasr_amd.decode.measure_rtf).
    Random-init weights never emit <eos>, so every utterance is decoded for exactly `out_steps` output steps with a full
    beam (36 = mean of round(xlen / 30) + 1 over the set): the per-step work of a real decode, without early termination."""
    import logging
    from emoasr_amd import decode as dec
    from emoasr_amd.modeling.asr import ASR
    from emoasr_amd.modeling.lm import LM
    logging.disable(logging.WARNING)  # ("cannot decode": no hypothesis ever ends with <eos> here, by construction)
    torch.manual_seed(1)
    model = ASR(SimpleNamespace(**L3), compute_dtype=dtype)
    lm = LM(SimpleNamespace(**LM12), compute_dtype=dtype)
    if eos_biased:
        # the second leg: heads biased as tests/test_fullsize_l3_l4_gpu.py does, so that hypotheses END (<eos> likely in the
        # attention head, blank-dominated CTC posteriors, a flattened LM head): finished hypotheses, the results list and the
        # search's own early stop are inside the timed region; the number of output steps is whatever the search takes
        with torch.no_grad():
            model.decoder.output.weight.mul_(6.0)
            model.decoder.output.bias[2] += 8.0
            model.decoder.ctc.output.weight.mul_(3.0)
            model.decoder.ctc.output.bias[0] += 14.0
            for n, p in lm.named_parameters():
                if n.endswith("predictions.transform.LayerNorm.weight"):
                    p.mul_(0.1)
    model, lm = model.to(dev).eval(), lm.to(dev).eval()
    loader, vocab, _ = rtf_fixture(os.path.join(tmpdir, "l33e" if eos_biased else "l33"), n_utts, 2)
    model.decoder.max_decode_ylen = 8
    dec.test(model, loader, vocab, 10, 0.0, 0.3, False, lm, 0.3, dev, num_samples=2)  # warm-up
    model.decoder.max_decode_ylen = 200 if eos_biased else out_steps
    model.engine()._beam_stats = None
    runtime, rtf = dec.measure_rtf(model, loader, vocab, 10, 0.0, 0.3, False, lm, 0.3, dev, num_samples=n_utts,
                                   num_repeats=repeats)
    logging.disable(logging.NOTSET)
    st = getattr(model.engine(), "_beam_stats", None)
    steps_per_utt = (st["steps"] / max(st["utts"], 1)) if st and st["steps"] else float(out_steps)
    out = dict(rtf=rtf, out_steps=steps_per_utt, ms_per_step=1e3 * runtime / steps_per_utt, utts=n_utts, repeats=repeats, beam=10,
               lm_weight=0.3, decode_ctc_weight=0.3, forced_steps=not eos_biased)
    # what one output step has to read: every weight of the decoder and the LM once (bf16) -- the HBM floor of a step
    wbytes = 2 * (sum(p.numel() for n, p in model.named_parameters() if n.startswith("decoder.")) +
                  sum(p.numel() for p in lm.parameters()))
    out["weight_bytes_per_step"] = wbytes
    if st and st["steps"]:
        # inside the device-resident search loop only (graph refresh per utterance included; encoder, feature loading and the
        # per-utterance set-up excluded): ms_per_step above is the whole utterance / steps, the RTF protocol's view
        ms = 1e3 * st["loop_s"] / st["steps"]
        out["search_loop_ms_per_step"] = ms
        out["steps_per_s"] = 1e3 / ms
        out["achieved_GBps"] = wbytes / (ms * 1e-3) / 1e9
        out["hbm_frac"] = out["achieved_GBps"] / HBM_PEAK_GBS
    return out


def parity_mode(dev, batches, steps=4, warmup=2):
    """the f32 (parity) mode on the same workload: training frames/s with exact-f32 MFMA, and the measured distance of the
    bf16 engine from it on one full-size batch (same weights, dropout 0): relative loss error, max logits error over the
    logits' range, greedy token agreement.  The f32 engine is what tests/ hold to the oracle at 1e-3 / bit-exact ids."""
    from emoasr_amd.modeling.asr import ASR
    from emoasr_amd.train import ArenaAdam, noam_lr
    def timed(mode, accum=1):
        """accum micro-batches per optimizer step (accum > 1: through engine.ctc_train_stacked, as the headline's are)"""
        torch.manual_seed(0)
        m32 = ASR(SimpleNamespace(**L2), compute_dtype=mode).to(dev).train()
        eng = m32.engine()
        opt = ArenaAdam(eng.arena, lambda s: noam_lr(OPT["lr"], L2["enc_hidden_size"], OPT["warmup"], s),
                        weight_decay=OPT["weight_decay"], clip_grad_norm=OPT["clip_grad_norm"])
        assert accum == 1 or eng.stacked_ok()

        def step(group):
            opt.zero_grad()
            if accum > 1:
                eng.ctc_train_stacked([(bt.xs, bt.xlens, bt.ys, bt.ylens) for bt in group], L2["blank_id"])
            else:
                loss, _ = m32(group[0].xs, group[0].xlens, group[0].ys, group[0].ylens, None, None)
                loss.backward()
            opt.step()

        groups = [batches[i * accum:(i + 1) * accum] for i in range(len(batches) // accum)]
        nw, ns = (warmup, steps) if accum == 1 else (1, min(3, len(groups) - 1))
        for g in groups[:nw]:
            step(g)
        torch.cuda.synchronize()
        if accum > 1:
            # every group has its own stacked shapes: its first pass grows the caching allocator's pool by gigabytes (hipMalloc inside
            # the region: one run read 119 ms per step for 79 ms of kernels) -- the timed groups run once untimed first
            for g in groups[nw:nw + ns]:
                step(g)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g in groups[nw:nw + ns]:
            step(g)
            if os.environ.get("EMOASR_BENCH_TRACE"):
                from emoasr_amd import ops as _ops
                torch.cuda.synchronize()
                print("[parity %s x%d] %.1f ms since start, split flag %d, mem %.1f / %.1f GiB" % (
                    mode, accum, 1e3 * (time.perf_counter() - t0), int(_ops.split_products()), torch.cuda.memory_allocated() / 2**30,
                    torch.cuda.memory_reserved() / 2**30), file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return sum(sum(b.xlens) for g in groups[nw:nw + ns] for b in g) / el, 1e3 * el / ns

    f32_fps, f32_ms = timed(torch.float32)
    x3_fps, x3_ms = timed("f32x3")
    out = {"f32_frames_per_s": f32_fps, "f32_ms_per_step": f32_ms,
           # the throughput mode that meets north_star's tolerances: f32 storage, every matrix product as three bf16 MFMAs over
           # (hi, lo) operand pairs (compute_dtype "f32x3"); one micro-batch per optimizer step, like the f32 leg
           "parity_mode_frames_per_s": x3_fps, "parity_mode_ms_per_step": x3_ms, "parity_mode": "f32x3"}
    acc = 5
    if len(batches) >= 2 * acc:
        # ... and with the headline's accumulation: five micro-batches per optimizer step in one stacked pass
        s_fps, s_ms = timed("f32x3", acc)
        out.update({"parity_mode_one_by_one_frames_per_s": x3_fps, "parity_mode_one_by_one_ms_per_step": x3_ms,
                    "parity_mode_frames_per_s": s_fps, "parity_mode_ms_per_step": s_ms, "parity_mode_accum_grad": acc,
                    "parity_mode_stacked_micro_batches": True})
        f_fps, f_ms = timed(torch.float32, acc)
        out.update({"f32_stacked_frames_per_s": f_fps, "f32_stacked_ms_per_step": f_ms})
    # bf16 vs f32 on identical weights and inputs, no dropout
    cfg0 = dict(L2, dropout_enc_rate=0.0, dropout_attn_rate=0.0)
    torch.manual_seed(0)
    a32 = ASR(SimpleNamespace(**cfg0), compute_dtype=torch.float32)
    sd = {k: v.clone() for k, v in a32.state_dict().items()}
    a16 = ASR(SimpleNamespace(**cfg0), compute_dtype=torch.bfloat16)
    a16.load_state_dict(sd)
    ax3 = ASR(SimpleNamespace(**cfg0), compute_dtype="f32x3")
    ax3.load_state_dict(sd)
    a32, a16, ax3 = a32.to(dev).eval(), a16.to(dev).eval(), ax3.to(dev).eval()
    bt = batches[0]
    res = []
    with torch.no_grad():
        for m in (a32, a16, ax3):
            eouts, elens, _ = m.encoder(bt.xs, bt.xlens)
            logits = m.decoder(eouts, elens).float()
            loss, _ = m(bt.xs, bt.xlens, bt.ys, bt.ylens, None, None)
            hyps = m.decode(bt.xs, bt.xlens)[0]
            res.append((float(loss), logits, hyps, [int(e) for e in elens]))
        # the mixed mode of greedy decoding: bf16 encoder, f32 logits out of the head product (engine.f32_head, the default)
        eng16 = a16.engine()
        e16, el16, _ = a16.encoder(bt.xs, bt.xlens)
        z16f = eng16.head_logits(e16, "decoder.output", out_f32=True).float()
        eng16.f32_head = False
        h16_bf16_head = a16.decode(bt.xs, bt.xlens)[0]
        eng16.f32_head = True
    (l32, z32, h32, el32), (l16, z16, h16, _), (lx3, zx3, hx3, _) = res
    mask = torch.zeros(z32.shape[:2], dtype=torch.bool, device=dev)
    for b, e in enumerate(el32):
        mask[b, :e] = True
    a1 = z32.argmax(-1)[mask]
    a2 = z16.argmax(-1)[mask]
    out["bf16_vs_f32"] = {"loss_rel": abs(l16 - l32) / abs(l32),
                          "logits_rel": float((z16 - z32)[mask].abs().max() / (z32[mask].max() - z32[mask].min())),
                          "greedy_frame_agreement": float((a1 == a2).float().mean()),
                          "greedy_hyp_exact": float(np.mean([x == y for x, y in zip(h32, h16)])),
                          "f32_head": {"note": "bf16 encoder, head product with f32 output (what greedy decoding uses)",
                                       "greedy_frame_agreement": float((a1 == z16f.argmax(-1)[mask]).float().mean()),
                                       "logits_rel": float((z16f - z32)[mask].abs().max() / (z32[mask].max() - z32[mask].min()))},
                          "bf16_head_greedy_frame_agreement": float((a1 == a2).float().mean()),
                          "batch": f"B={len(bt.xlens)}, {sum(bt.xlens)} frames, random-init weights, dropout 0"}
    out["f32x3_vs_f32"] = {"loss_rel": abs(lx3 - l32) / abs(l32),
                           "logits_rel": float((zx3 - z32)[mask].abs().max() / (z32[mask].max() - z32[mask].min())),
                           "greedy_frame_agreement": float((a1 == zx3.argmax(-1)[mask]).float().mean()),
                           "greedy_hyp_exact": float(np.mean([x == y for x, y in zip(h32, hx3)])),
                           "batch": f"B={len(bt.xlens)}, {sum(bt.xlens)} frames, random-init weights, dropout 0"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--accum", type=int, default=5,
                    help="micro-batches per optimizer step (the reference trains with accum_grad: 5, asr/correct/exps/csj/asr.yaml:52, "
                         "asr/train_asr.py:106-128); a `step` of this benchmark is one optimizer step = this many micro-batches")
    ap.add_argument("--timer-stride", type=int, default=17,
                    help="in the timed region every n-th launch of the dominant kernel family is timed with a HIP event pair")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print a per-entry-point GPU time table to stderr")
    args = ap.parse_args()

    # the intra-op CPU pool must fit the container's CPU quota, or the kernel throttles the whole process -- the launching and
    # polling thread included -- for tens of milliseconds at a time (emoasr_amd/hostenv.py)
    from emoasr_amd.hostenv import respect_cpu_quota
    respect_cpu_quota()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    # (EMOASR_BENCH_ONE_GPU=1 + EMOASR_DIST_BACKEND=gloo: rehearsal of the N > 1 code path on a 1-GPU box)
    if os.environ.get("EMOASR_BENCH_ONE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("EMOASR_DIST_BACKEND", "nccl")  # nccl = RCCL over xGMI
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from emoasr_amd import lib as emo_lib, ops
    from emoasr_amd.data import specaug_spans
    from emoasr_amd.engine import h2d_i32, h2d_pack
    from emoasr_amd.modeling.asr import ASR
    from emoasr_amd.train import ArenaAdam, noam_lr

    torch.manual_seed(0)  # identical initial weights on every rank
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = ASR(SimpleNamespace(**L2), compute_dtype=dtype).to(dev)
    model.train()
    eng = model.engine()
    eng.seed = 1234 + rank  # per-replica dropout streams
    opt = ArenaAdam(eng.arena, lambda s: noam_lr(OPT["lr"], L2["enc_hidden_size"], OPT["warmup"], s),
                    weight_decay=OPT["weight_decay"], clip_grad_norm=OPT["clip_grad_norm"])
    accum = max(1, args.accum)
    micro = make_batches(rank, world, (args.warmup + args.steps) * accum, dev)
    batches = [micro[i * accum:(i + 1) * accum] for i in range(args.warmup + args.steps)]   # one entry = one optimizer step
    np_rng, py_rng = np.random.RandomState(rank), random.Random(rank)

    buckets = None
    if world > 1 and os.environ.get("EMOASR_DP_OVERLAP", "1") != "0":
        from emoasr_amd.train import GradBuckets
        buckets = GradBuckets(eng.arena.grad)
        eng.grad_hook = buckets.ready

    stacked = accum > 1 and dtype == torch.bfloat16 and eng.stacked_ok()

    def step(group):
        """one optimizer step: `accum` micro-batches (SpecAugment on the device each), gradients of loss / accum summed -- through
        the engine together when it can stack them (engine.ctc_train_stacked), else one after the other --, all-reduce, Adam"""
        datas = []
        # the SpecAugment span tables and lengths of all micro-batches of the step in one pinned upload
        tabs = h2d_pack([torch.as_tensor(t, dtype=torch.int32) for bt in group
                         for t in (specaug_spans(bt.xlens, 80, np_rng=np_rng, py_rng=py_rng), bt.xlens)], dev)
        for k, bt in enumerate(group):
            xs = bt.xs.clone()
            ops.specaug_apply(xs, tabs[2 * k], 2, 2, tabs[2 * k + 1])
            datas.append((xs, bt.xlens, bt.ys, bt.ylens))
        opt.zero_grad()
        if stacked:
            loss = eng.ctc_train_stacked(datas, L2["blank_id"]).mean()
        else:
            for xs, xlens, ys, ylens in datas:
                loss, _ = model(xs, xlens, ys, ylens, None, None)
                (loss / accum).backward()
        if buckets is not None:
            buckets.finish()   # the tail ranges went out during the backward sweep (train.GradBuckets)
        elif world > 1:
            opt.allreduce()
        opt.step(grad_mult=1.0 / world)
        return loss

    def attn_pairs(group):
        sub = [((t - 3) // 2 + 1 - 3) // 2 + 1 for bt in group for t in bt.xlens]  # Conv2dEncoder: two stride-2 3x3 convs
        return float(sum(t * t for t in sub))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # algorithmic work of the attention kernels for a group of micro-batches (they depend on the utterance lengths):
    # per valid (query, key, head) pair -- forward 3 products of 2*DK flop (Q K^T, Q pos^T, P V); backward main kernel 7
    # (Q K^T, Q pos^T, dO V^T, dV, dK, dQ from K, dQ from pos); dpos kernel 1 (dS^T (Q+v))
    H_, DK_, NL_ = L2["enc_num_attention_heads"], L2["enc_hidden_size"] // L2["enc_num_attention_heads"], L2["enc_num_layers"]

    def attn_work(pairs):
        unit = 2.0 * DK_ * H_ * pairs * NL_
        return 3 * unit, 7 * unit, 1 * unit

    def analytic_families(group, nparams):
        """algorithmic GFLOP / GB per optimizer step of the kernel families that the library's timers do not cover (they carry no
        work counters): formulas of DESIGN.md section 4.2, evaluated on the micro-batch shapes of the instrumented step.  The rows
        of DESIGN.md section 4.1 (tools/evidence_table.py) divide them by the kernel table's times."""
        C, V, d = L2["enc_hidden_size"], L2["vocab_size"], L2["enc_hidden_size"]
        fe_fl = fe_by = ctc_fl = ctc_by = aux_fl = aux_by = 0.0
        for bt in group:
            B, T = len(bt.xlens), max(bt.xlens)
            T1, F1 = (T - 3) // 2 + 1, (80 - 3) // 2 + 1
            T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
            y1, y2, M = B * T1 * F1 * C * 2.0, B * T2 * F2 * C * 2.0, B * T2
            # Conv2d(1 -> C) forward + weight gradient (VALU), Conv2d(C -> C) forward / data gradient / weight gradient (MFMA)
            fe_fl += 2 * (2.0 * B * T1 * F1 * C * 9) + 3 * (2.0 * B * T2 * F2 * C * 9 * C)
            fe_by += (B * T * 80 * 4.0 + y1) * 2 + (y1 + y2) + (y2 + 2 * y1) + (y1 + y2)
            # vocabulary head (logits stored once, read by the gradient kernel, gradient rows written) + its soft-max partials
            ctc_fl += 2.0 * M * V * d
            ctc_by += 3.0 * M * V * 2 + M * d * 2.0
            # attention side kernels of the 12 layers: keep-mask words written, Q + bias copies / delta, the table gradient
            # (dS image read once + (Q+v) rows; 1 product of 2 DK flop per (query, key, head) pair over the PADDED square)
            nw = (T2 + 31) // 32
            aux_fl += NL_ * 2.0 * DK_ * H_ * B * T2 * T2
            aux_by += NL_ * (M * H_ * nw * 4.0 + 5 * M * d * 2.0 + M * H_ * 4.0 + B * H_ * T2 * T2 * 2.0 + M * d * 2.0)
        return {"front_end_conv2d": {"gflop": fe_fl / 1e9, "gbytes": fe_by / 1e9},
                "ctc_head_loss": {"gflop": ctc_fl / 1e9, "gbytes": ctc_by / 1e9},
                "attention_aux_side_stream": {"gflop": aux_fl / 1e9, "gbytes": aux_by / 1e9},
                "optimizer": {"gflop": 0.0, "gbytes": nparams * 32.0 / 1e9}}

    # warm-up; the LAST warm-up step is instrumented (every kernel family timed inside the library with HIP events on the launch
    # stream): it gives the per-family table and picks the dominant family, which alone is then timed over the timed region
    breakdown = None
    families, families_analytic = {}, {}
    for i in range(args.warmup):
        if i == args.warmup - 1:
            sync()
            emo_lib.set_option("timers", 1)
            for fam in FAMILIES:
                emo_lib.timer_read_ex(fam)
            t_w = time.perf_counter()
            if args.breakdown:
                with CallTimer(emo_lib) as ct:
                    ct.attn_pairs = attn_pairs(batches[i])
                    step(batches[i])
                breakdown = ct.summary()
            else:
                step(batches[i])
            sync()
            t_w = time.perf_counter() - t_w
            emo_lib.set_option("timers", 0)
            families = family_table(emo_lib, attn_work(attn_pairs(batches[i])), t_w)
            families_analytic = analytic_families(batches[i], sum(p.numel() for p in model.parameters()))
        else:
            step(batches[i])
    # the roofline object is about ONE kernel family, chosen at run time: the one with the most device time in the instrumented
    # step.  It is timed inside the library over exactly the timed region (only its own events are recorded there).
    dominant = max(families, key=lambda k: families[k]["ms"]) if families else "gemm_nt_nn"
    sync()
    frames = sum(sum(b.xlens) for grp in batches[args.warmup:] for b in grp)
    # an event pair per launch is not free (~10 us of lost overlap with the neighbouring kernels): families with hundreds of
    # launches per step are SAMPLED, every 17th launch (17 is coprime with the per-step launch counts -- 208 = 16 x 13 for the
    # GEMMs -- so over the timed steps every call site is sampled equally often)
    stride = args.timer_stride if families.get(dominant, {}).get("calls", 0) > 48 else 1
    emo_lib.set_option("timer_stride", stride)
    emo_lib.set_option("timers", emo_lib.timer_mask(dominant))
    emo_lib.timer_read_ex(dominant)
    stats0 = torch.cuda.memory_stats(dev) if os.environ.get("EMOASR_BENCH_TRACE") else None
    t0 = time.perf_counter()
    pairs = 0.0
    for bt in batches[args.warmup:]:
        pairs += attn_pairs(bt)
        loss = step(bt)
    sync()
    elapsed = time.perf_counter() - t0
    if stats0 is not None:   # did the caching allocator call the driver inside the timed region?
        st1 = torch.cuda.memory_stats(dev)
        print("[headline] hipMalloc calls in the timed region: %d (segments %d -> %d, reserved %.1f -> %.1f GiB)" % (
            st1["num_device_alloc"] - stats0["num_device_alloc"], stats0["segment.all.current"], st1["segment.all.current"],
            stats0["reserved_bytes.all.current"] / 2**30, st1["reserved_bytes.all.current"] / 2**30), file=sys.stderr, flush=True)
    # a rate measured on broken arithmetic is no measurement: the last loss and every parameter must be finite after the timed steps
    if not (bool(torch.isfinite(loss.detach()).all()) and bool(torch.isfinite(eng.arena.flat).all())):
        raise SystemExit("bench.py: non-finite loss or parameters after the timed steps (loss %r)" % (float(loss.detach()),))
    emo_lib.set_option("timers", 0)
    emo_lib.set_option("timer_stride", 1)
    dom = family_table(emo_lib, attn_work(pairs)).get(dominant, {"calls": 0, "ms": 0.0})
    tt = torch.tensor([elapsed, float(frames)], device=dev, dtype=torch.float64)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed, frames = tmax[0].item(), tt[1].item()
    value = frames / elapsed
    dp_info = None
    if world > 1:
        # the N > 1 line validates itself: what the collective backend reports, every rank's own frame count, and -- data-parallel
        # replicas apply the SAME reduced gradient with the same fused Adam -- the parameters must be bit-identical on all ranks
        # after the timed steps (a checksum over the raw bits of the flat f32 arena + its f64 sum, gathered from every rank)
        flat = eng.arena.flat
        # bit identity is decided element by element: rank 0's arena is broadcast and every rank counts the words that differ
        ref0 = flat.clone()
        dist.broadcast(ref0, 0)
        ndiff = (flat.view(torch.int32) != ref0.view(torch.int32)).sum().to(torch.int64)
        dist.all_reduce(ndiff, op=dist.ReduceOp.SUM)
        del ref0
        # reported checksum: position-weighted wrap-around sum of the raw words, gathered as int64 (no float round trip)
        words = flat.view(torch.int32).to(torch.int64)
        weight = torch.arange(words.numel(), device=dev, dtype=torch.int64) % 65521 + 1
        bits = (words * weight).sum()
        mine_i = torch.stack([bits])
        alli = [torch.zeros_like(mine_i) for _ in range(world)]
        dist.all_gather(alli, mine_i)
        alli = torch.stack(alli).cpu()
        mine = torch.stack([torch.tensor(float(sum(sum(b.xlens) for grp in batches[args.warmup:] for b in grp)), device=dev,
                                         dtype=torch.float64), flat.double().sum()])
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        dp_info = {"backend": dist.get_backend(), "nranks": dist.get_world_size(),
                   "one_gpu_rehearsal": os.environ.get("EMOASR_BENCH_ONE_GPU") == "1",
                   "frames_per_rank": [float(v) for v in allr[:, 0]],
                   "param_bits_checksum": [int(v) for v in alli[:, 0]], "param_sum": [float(v) for v in allr[:, 1]],
                   "param_words_differing_from_rank0": int(ndiff.item()),
                   "params_identical_across_ranks": bool(int(ndiff.item()) == 0 and (alli[:, 0] == alli[0, 0]).all()),
                   "overlapped_allreduce": buckets is not None,
                   "note": "no scaling value has been measured on multi-GPU hardware by the builder: the driver's SCALE run is the "
                           "only N > 1 execution over RCCL / xGMI"}

    if rank == 0:
        res = {
            "metric": "train frames/sec + decode RTF, Conformer-CTC 23M, 80-mel, 1/2/4/8 GPU",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / max(args.steps, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "CTC(Conformer) 23M `L2`, bf16, SpecAugment on-GPU, synthetic LibriSpeech-shaped "
                                   "batches (2000 utts, lognormal lengths, ASRBatchSampler packing 30000 frames/50 utts "
                                   f"per GPU), one step = accum_grad {accum} micro-batches (the reference's accum_grad: 5, "
                                   "asr/train_asr.py:106-128" + (", stacked through the engine in one pass" if stacked else "") +
                                   ") fwd+bwd, then all-reduce+clip+Adam; dropout 0.1",
                       "accum_grad": accum, "stacked_micro_batches": bool(stacked),
                       "params_M": sum(p.numel() for p in model.parameters()) / 1e6,
                       "parallelism": f"dp{world}", "final_loss": float(loss.detach())},
        }
        if dp_info is not None:
            res["dp"] = dp_info
        mean_T = frames / world / args.steps / accum / max(1, np.mean([len(b.xlens) for grp in batches[args.warmup:] for b in grp]))
        train_flops = 3.0 * fwd_flops_per_utt(int(mean_T)) / max(mean_T, 1) * frames
        res["model_mfma_frac"] = train_flops / elapsed / (MFMA_PEAK_TFLOPS[args.dtype] * 1e12 * world)
        if dom["calls"]:
            # which roof bounds the family: its algorithmic intensity against the ridge of the two peaks
            ai = (dom.get("gflop", 0.0) / dom["gbytes"]) if dom.get("gbytes") else None
            bound = "hbm" if (ai is not None and ai < RIDGE_FLOP_PER_BYTE) or "tflops" not in dom else "mfma"
            if bound == "hbm" and "gbps" in dom:
                ach, peak, unit = dom["gbps"], HBM_PEAK_GBS, "GB/s"
            else:
                bound, ach, peak, unit = "mfma", dom["tflops"], MFMA_PEAK_TFLOPS[args.dtype], "TFLOP/s"
            res["roofline"] = {"kernel": dominant, "symbols": FAMILIES[dominant], "chosen": "largest device time of the "
                               "instrumented warm-up step (in-library HIP-event timers)", "bound": bound, "achieved": ach,
                               "peak": peak, "unit": unit, "frac": ach / peak,
                               "mfma_frac": dom.get("mfma_frac"), "hbm_frac": dom.get("hbm_frac"),
                               "flop_per_byte": ai, "ridge_flop_per_byte": RIDGE_FLOP_PER_BYTE,
                               "traffic": pmc_kernel(dominant, "traffic_bytes"), "mfma_util": pmc_kernel(dominant, "mfma_util"),
                               "pmc_source": pmc_source(),
                               "launches": dom["calls"], "sampled": f"every {stride}th launch" if stride > 1 else "every launch",
                               "avg_us": dom["avg_us"],
                               "flop_per_launch": 1e9 * dom.get("gflop", 0.0) / dom["calls"],
                               "bytes_per_launch": 1e9 * dom.get("gbytes", 0.0) / dom["calls"],
                               "share_of_step": stride * dom["ms"] * 1e-3 / elapsed}
        else:
            res["roofline"] = {"kernel": dominant, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": None, "traffic": None, "launches": 0}
        res["families"] = {"step": "the last warm-up step, every family timed (HIP events inside the library)", **families}
        res["families_analytic"] = {"what": "algorithmic GFLOP / GB per optimizer step of the families without in-library counters "
                                            "(formulas: bench.py analytic_families; times: the kernel table)", **families_analytic}
        if args.breakdown and breakdown:
            tot = sum(v["ms"] for v in breakdown.values())
            for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms"]):
                tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0.0
                print(f"  {k:28s} calls {v['calls']:5d}  {v['ms']:9.3f} ms  {100 * v['ms'] / tot:5.1f}%  {tf:7.1f} TF/s",
                      file=sys.stderr)
            print(f"  total instrumented GPU time {tot:.2f} ms (one step, {accum} micro-batches)", file=sys.stderr)
        if world == 1 and not args.no_decode:
            # the same optimizer steps with the features handed over as HOST buffers (pinned, as a DataLoader with pin_memory
            # yields them): the H2D copies are inside the timed region.  Never `value`; noted in DESIGN.md section 4.
            host_steps = batches[args.warmup:args.warmup + 4]
            pinned = [[bt.xs.cpu().pin_memory() for bt in grp] for grp in host_steps]

            def host_step(grp, pins):
                return step([SimpleNamespace(xs=px.to(dev, non_blocking=True), xlens=bt.xlens, ys=bt.ys, ylens=bt.ylens)
                             for bt, px in zip(grp, pins)])

            host_step(host_steps[0], pinned[0])
            sync()
            t0 = time.perf_counter()
            for grp, pins in zip(host_steps[1:], pinned[1:]):
                host_step(grp, pins)
            sync()
            res["host_inputs"] = {"frames_per_s": sum(sum(b.xlens) for grp in host_steps[1:] for b in grp) / (time.perf_counter() - t0),
                                  "steps": len(host_steps) - 1,
                                  "MB_per_step": sum(px.numel() * 4 for px in pinned[1]) / 1e6,
                                  "note": "features start in pinned host memory, H2D inside the timed region (PCIe-inclusive rate)"}
        if world == 1 and not args.no_decode:
            import tempfile
            with tempfile.TemporaryDirectory() as tmpdir:
                res["decode_rtf"] = decode_rtf(model, dev, tmpdir)
                res["ctc_beam"] = ctc_beam_rtf(model, dev, dtype, tmpdir)
                res["decode_rtf_batch32"] = decode_rtf_batched(model, dev)
                res["decode_l33"] = decode_rtf_l33(dev, dtype, tmpdir)
                res["decode_l33_eos"] = decode_rtf_l33(dev, dtype, tmpdir, n_utts=10, repeats=2, eos_biased=True)
            # log-mel in the loop: the same steps starting from raw 16 kHz audio (fbank kernel -> SpecAugment -> model)
            from emoasr_amd.features import LogMel
            fb = logmel_rate(dev, batches[-1][-1].xlens)
            lmel = LogMel(dev)
            g = torch.Generator().manual_seed(9)
            sub = batches[args.warmup:args.warmup + 3]
            wavs = [[[(0.05 * torch.randn(160 * (t - 1) + 400, generator=g)).to(dev) for t in bt.xlens] for bt in grp] for grp in sub]

            def wav_step(grp, ws_grp):
                feats = []
                for bt, ws_ in zip(grp, ws_grp):
                    xs = torch.zeros_like(bt.xs)
                    for b, w in enumerate(ws_):
                        f = lmel(w)
                        xs[b, : f.shape[0]] = f[: xs.shape[1]]
                    feats.append(SimpleNamespace(xs=xs, xlens=bt.xlens, ys=bt.ys, ylens=bt.ylens))
                return step(feats)

            wav_step(sub[0], wavs[0])
            sync()
            t0 = time.perf_counter()
            for grp, ws_grp in zip(sub[1:], wavs[1:]):
                wav_step(grp, ws_grp)
            sync()
            res["logmel"] = {"frames_per_s": fb, "train_frames_per_s_with_logmel":
                             sum(sum(b.xlens) for grp in sub[1:] for b in grp) / (time.perf_counter() - t0),
                             "steps": len(sub) - 1}
            # (the secondary legs start from an empty caching allocator: the blocks the L2 legs left behind fragment the
            # pool the transducer step's 0.3-1 GB tensors are carved from)
            del wavs, sub
            torch.cuda.empty_cache()
            res["l4_rnnt"] = l4_rnnt(dev, dtype)
            torch.cuda.empty_cache()
            if args.dtype == "bf16":
                res.update(parity_mode(dev, [b for grp in batches for b in grp]))
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(model)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
