"""Generate golden vectors by importing the reference (emonosuke/emoASR at /root/reference).

Runs ONLY in the authoring container (the reference cannot travel).  Outputs small .npz
fixtures next to this script; tests/test_oracle_golden.py pins oracle/ against them on CPU
and tests/test_model_gpu.py pins the HIP engine against them on the GPU.

    python tests/golden/make_golden.py
"""
import os
import sys
import types
from collections import namedtuple

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
# warp_rnnt (third-party CUDA RNN-T loss) is imported at module import time by
# asr/modeling/decoders/rnn_transducer.py:14 and is not installed here.
sys.modules.setdefault("warp_rnnt", types.ModuleType("warp_rnnt"))

from asr.modeling.asr import ASR  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

COMMON = dict(input_layer="conv2d", feat_dim=40, num_framestacks=1, enc_hidden_size=128,
              enc_num_attention_heads=2, enc_num_layers=2, enc_intermediate_size=256,
              dropout_enc_rate=0.0, dropout_attn_rate=0.0, dropout_dec_rate=0.0, vocab_size=40,
              blank_id=0, eos_id=2, kd_weight=0, lsm_prob=0.1)
CONFIGS = {
    "l2_tiny": dict(COMMON, encoder_type="conformer", decoder_type="ctc", pos_encode_type="rel"),
    "l1_tiny": dict(COMMON, encoder_type="transformer", decoder_type="ctc"),
}


def make_params(d):
    return namedtuple("Params", d.keys())(**d)


def make_batch(seed, feat_dim, vocab):
    g = torch.Generator().manual_seed(seed)
    xlens = torch.tensor([203, 167, 131, 64])
    ylens = torch.tensor([9, 7, 5, 3])
    B, T, L = len(xlens), int(xlens.max()), int(ylens.max())
    xs = torch.randn(B, T, feat_dim, generator=g)
    ys = torch.randint(3, vocab, (B, L), generator=g)
    for b in range(B):
        xs[b, xlens[b]:] = 0.0
        ys[b, ylens[b]:] = 2  # padded with eos, as collate_fn does (datasets.py:160-170)
    ys[0, 3] = ys[0, 2]  # a repeated label
    eos = torch.full((B, 1), 2)
    ys_in = torch.cat([eos, ys], 1)
    ys_out = torch.cat([ys, eos], 1)
    for b in range(B):
        ys_out[b, ylens[b]] = 2
        ys_out[b, ylens[b] + 1:] = 2
    return xs, xlens, ys, ylens, ys_in, ys_out


def run_ctc(name, cfg):
    torch.manual_seed(0)
    params = make_params(cfg)
    model = ASR(params, phase="train")
    # give BatchNorm non-trivial running statistics / affine parameters and the output
    # layer a wider spread, so eval-mode greedy decoding is not dominated by ties
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("batch_norm.weight") or (".norm" in n and n.endswith("weight")):
                p.add_(0.1 * torch.randn_like(p))
            if n.endswith("batch_norm.bias") or (".norm" in n and n.endswith(".bias")):
                p.add_(0.1 * torch.randn_like(p))
        model.decoder.output.weight.mul_(3.0)
    xs, xlens, ys, ylens, ys_in, ys_out = make_batch(1, cfg["feat_dim"], cfg["vocab_size"])
    out = {}
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    for k, v in sd0.items():
        out["sd/" + k] = v.numpy()
    out.update(xs=xs.numpy(), xlens=xlens.numpy(), ys=ys.numpy(), ylens=ylens.numpy(),
               ys_in=ys_in.numpy(), ys_out=ys_out.numpy())
    # ---- train mode (dropout 0, BatchNorm batch statistics), loss + all gradients
    model.train()
    loss, loss_dict = model(xs, xlens, ys, ylens, ys_in, ys_out)
    loss.backward()
    out["train/loss"] = loss.detach().numpy()
    for n, p in model.named_parameters():
        out["grad/" + n] = p.grad.numpy()
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            out["sd_after/" + k] = v.clone().numpy()
    # ---- eval mode from the ORIGINAL state: encoder outputs, logits, greedy decode
    model.load_state_dict(sd0)
    model.eval()
    with torch.no_grad():
        eouts, elens, _ = model.encoder(xs, xlens)
        logits = model.decoder(eouts, elens)
        loss_eval, _ = model(xs, xlens, ys, ylens, ys_in, ys_out)
        hyps, _, _, aligns = model.decode(xs, xlens, beam_width=1)
    out["eval/eouts"] = eouts.numpy()
    out["eval/elens"] = elens.numpy()
    out["eval/logits"] = logits.numpy()
    out["eval/loss"] = loss_eval.numpy()
    out["eval/hyp_lens"] = np.array([len(h) for h in hyps])
    out["eval/hyps"] = np.array(sum(hyps, []), dtype=np.int64)
    out["eval/aligns"] = np.array(sum(aligns, []), dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "loss", float(loss), "eval loss", float(loss_eval), "hyp lens", [len(h) for h in hyps],
          "params", sum(p.numel() for p in model.parameters()))


L3 = dict(COMMON, encoder_type="conformer", decoder_type="transformer", pos_encode_type="rel",
          dec_hidden_size=128, dec_num_attention_heads=2, dec_num_layers=2, dec_intermediate_size=256,
          mtl_ctc_weight=0.3, loss_normalize_length=False, loss_normalize_batch=True, max_decode_ylen=20)
LM_CFG = dict(lm_type="transformer", vocab_size=40, hidden_size=128, num_layers=2, num_attention_heads=2,
              intermediate_size=256, max_seq_len=64)
DECODE_SETTINGS = [dict(beam_width=4, len_weight=0.0, lm_weight=0.0, decode_ctc_weight=0.0),
                   dict(beam_width=4, len_weight=0.0, lm_weight=0.0, decode_ctc_weight=0.3),
                   dict(beam_width=4, len_weight=0.1, lm_weight=0.3, decode_ctc_weight=0.3),
                   dict(beam_width=3, len_weight=0.2, lm_weight=0.5, decode_ctc_weight=0.0)]


def run_l3():
    from lm.modeling.lm import LM
    torch.manual_seed(0)
    model = ASR(make_params(L3), phase="train")
    lm = LM(make_params(LM_CFG))
    lm.eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".norm" in n or "batch_norm" in n:
                p.add_(0.1 * torch.randn_like(p))
        model.decoder.output.weight.mul_(2.0)
        model.decoder.ctc.output.bias[L3["blank_id"]] += 5.0  # blank-dominated CTC: short prefixes keep non-negligible mass
        model.decoder.output.bias[L3["eos_id"]] += 2.5  # random-init decoders never emit <eos> otherwise
        for n, p in lm.named_parameters():
            if "LayerNorm" in n:
                p.add_(0.1 * torch.randn_like(p))
        lm.lm.transformer.cls.predictions.bias.add_(0.5 * torch.randn(LM_CFG["vocab_size"]))
    xs, xlens, ys, ylens, ys_in, ys_out = make_batch(1, L3["feat_dim"], L3["vocab_size"])
    out = {}
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    for k, v in sd0.items():
        out["sd/" + k] = v.numpy()
    for k, v in lm.state_dict().items():
        out["lm/" + k] = v.clone().numpy()
    out.update(xs=xs.numpy(), xlens=xlens.numpy(), ys=ys.numpy(), ylens=ylens.numpy(), ys_in=ys_in.numpy(),
               ys_out=ys_out.numpy())
    model.train()
    loss, loss_dict = model(xs, xlens, ys, ylens, ys_in, ys_out)
    loss.backward()
    out["train/loss"] = loss.detach().numpy()
    out["train/loss_att"] = loss_dict["loss_att"].detach().numpy()
    out["train/loss_ctc"] = loss_dict["loss_ctc"].detach().numpy()
    for n, p in model.named_parameters():
        out["grad/" + n] = p.grad.clone().numpy()
    model.load_state_dict(sd0)
    model.eval()
    with torch.no_grad():
        eouts, elens, _ = model.encoder(xs, xlens)
        logits = model.decoder(eouts, elens, None, ys, ylens, ys_in, None)
        out["eval/eouts"] = eouts.numpy()
        out["eval/att_logits"] = logits.numpy()
        ystest = torch.randint(3, LM_CFG["vocab_size"], (3, 9))
        yl = torch.tensor([9, 6, 2])
        lp, _ = lm.predict(ystest, yl)
        out["lm_test/ys"], out["lm_test/ylens"], out["lm_test/logp"] = ystest.numpy(), yl.numpy(), lp.numpy()
        import warnings
        for si, st in enumerate(DECODE_SETTINGS):
            for b in range(2):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    hyps, scores, _, _ = model.decode(xs[b:b + 1, : xlens[b]], xlens[b:b + 1], lm=lm, **st)
                out[f"decode/{si}/{b}/lens"] = np.array([len(h) for h in hyps])
                out[f"decode/{si}/{b}/hyps"] = np.array(sum(hyps, []), dtype=np.int64)
                out[f"decode/{si}/{b}/scores"] = np.array(scores, dtype=np.float64)
                print("decode", si, b, [len(h) for h in hyps], [round(s, 3) for s in scores])
    np.savez_compressed(os.path.join(OUT, "l3_tiny.npz"), **out)
    print("l3_tiny loss", float(loss.detach()), {k: float(v) for k, v in loss_dict.items()})


L4 = dict(COMMON, encoder_type="conformer", decoder_type="rnn_transducer", pos_encode_type="rel",
          embedding_size=64, dec_hidden_size=128, dec_num_layers=2, joint_hidden_size=128, dropout_emb_rate=0.0,
          mtl_ctc_weight=0.3)


def run_l4():
    """RNN-T.  warp_rnnt is absent: the reference network runs with oracle.rnnt.rnnt_loss plugged in as
    warp_rnnt.rnnt_loss (loss value unpinned; LSTM / joint / aux CTC / greedy decode pinned)."""
    sys.path.insert(0, "/root/repo")
    from oracle import rnnt as orn
    sys.modules["warp_rnnt"].rnnt_loss = orn.rnnt_loss
    sys.modules["warp_rnnt"].__version__ = "oracle-restatement"
    import asr.modeling.decoders.rnn_transducer as rt
    rt.warp_rnnt = sys.modules["warp_rnnt"]
    torch.manual_seed(0)
    model = ASR(make_params(L4), phase="train")
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".norm" in n or "batch_norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    xs, xlens, ys, ylens, ys_in, ys_out = make_batch(1, L4["feat_dim"], L4["vocab_size"])
    # a random-init transducer either never or always emits blank; fit the tiny model to this batch for a
    # couple of hundred Adam steps so that greedy decoding produces real label/blank interleavings
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    model.train()
    for it in range(int(os.environ.get("L4_FIT_STEPS", 120))):
        opt.zero_grad()
        l, _ = model(xs, xlens, ys, ylens, ys_in, ys_out)
        l.backward()
        opt.step()
        if it % 50 == 0:
            print("  fit", it, float(l.detach()))
    opt.zero_grad()
    out = {}
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    for k, v in sd0.items():
        out["sd/" + k] = v.numpy()
    out.update(xs=xs.numpy(), xlens=xlens.numpy(), ys=ys.numpy(), ylens=ylens.numpy(), ys_in=ys_in.numpy(),
               ys_out=ys_out.numpy())
    model.train()
    loss, loss_dict = model(xs, xlens, ys, ylens, ys_in, ys_out)
    loss.backward()
    out["train/loss"] = loss.detach().numpy()
    out["train/loss_rnnt"] = loss_dict["loss_rnnt"].detach().numpy()
    out["train/loss_ctc"] = loss_dict["loss_ctc"].detach().numpy()
    for n, p in model.named_parameters():
        out["grad/" + n] = p.grad.clone().numpy()
    model.load_state_dict(sd0)
    model.eval()
    with torch.no_grad():
        eouts, elens, _ = model.encoder(xs, xlens)
        douts, _ = model.decoder.recurrency(ys_in, None)
        out["eval/douts"] = douts.numpy()
        out["eval/joint_logits_b0"] = model.decoder.joint(eouts[:1, :20], douts[:1]).numpy()
        hyps, _, _, aligns = model.decoder._greedy(eouts, elens)
    out["eval/hyp_lens"] = np.array([len(h) for h in hyps])
    out["eval/hyps"] = np.array(sum(hyps, []), dtype=np.int64)
    out["eval/align_lens"] = np.array([len(a) for a in aligns])
    out["eval/aligns"] = np.array(sum(aligns, []), dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "l4_tiny.npz"), **out)
    print("l4_tiny loss", float(loss.detach()), {k: float(v) for k, v in loss_dict.items()}, "hyp lens",
          [len(h) for h in hyps], "align lens", [len(a) for a in aligns])


CTC_BEAM_SETTINGS = [dict(beam_width=4, len_weight=0.0, lm_weight=0.0),
                     dict(beam_width=4, len_weight=0.1, lm_weight=0.3),
                     dict(beam_width=3, len_weight=0.2, lm_weight=0.5)]


def run_ctc_beam():
    """CTC prefix beam search with LM shallow fusion (decoders/ctc.py:203-344,372-397) on the l2_tiny
    model and the l3_tiny LM; weights are read back from those fixtures (not stored twice) except the
    sharpened output layer (a random-init CTC head is near-uniform: every beam decision would be a tie)."""
    from lm.modeling.lm import LM
    ctc = np.load(os.path.join(OUT, "l2_tiny.npz"))
    l3 = np.load(os.path.join(OUT, "l3_tiny.npz"))
    model = ASR(make_params(CONFIGS["l2_tiny"]), phase="test")
    model.load_state_dict({k[3:]: torch.from_numpy(ctc[k]) for k in ctc.files if k.startswith("sd/")})
    lm = LM(make_params(LM_CFG))
    lm.load_state_dict({k[3:]: torch.from_numpy(l3[k]) for k in l3.files if k.startswith("lm/")})
    model.eval()
    lm.eval()
    torch.manual_seed(3)
    with torch.no_grad():
        model.decoder.output.weight.mul_(4.0)
        model.decoder.output.bias.add_(0.5 * torch.randn(model.decoder.output.bias.shape))
        model.decoder.output.bias[0] += 6.0  # blank-dominated frames, like a trained CTC model
    out = {"sd_override/decoder.output.weight": model.decoder.output.weight.detach().clone().numpy(),
           "sd_override/decoder.output.bias": model.decoder.output.bias.detach().clone().numpy()}
    xs, xlens = torch.from_numpy(ctc["xs"]), torch.from_numpy(ctc["xlens"])
    with torch.no_grad():
        for si, st in enumerate(CTC_BEAM_SETTINGS):
            for b in (1, 2, 3):
                hyps, scores, logits, _ = model.decode(xs[b:b + 1, : xlens[b]], xlens[b:b + 1], lm=lm, **st)
                out[f"decode/{si}/{b}/lens"] = np.array([len(h) for h in hyps])
                out[f"decode/{si}/{b}/hyps"] = np.array(sum([list(map(int, h)) for h in hyps], []), dtype=np.int64)
                out[f"decode/{si}/{b}/scores"] = np.array(scores, dtype=np.float64)
                if si == 0:
                    out[f"logits/{b}"] = logits.numpy()
                print("ctc beam", si, b, [len(h) for h in hyps], [round(float(s), 3) for s in scores])
    np.savez_compressed(os.path.join(OUT, "ctcbeam_tiny.npz"), **out)


def run_hostio():
    """known answers for the host-side formats, from the reference's own functions: WER alignment
    (asr/metrics.py:20-105), subword -> word joining (utils/vocab.py:45-64), batch packing
    (asr/datasets.py:200-234).  Stored as JSON (small, readable)."""
    import json
    import random as pyrandom
    from asr.datasets import ASRBatchSampler
    from asr.metrics import compute_wer
    from utils.vocab import Vocab
    rng = pyrandom.Random(7)
    words = ["a", "b", "c", "d", "e", "f"]
    wer_cases = []
    for _ in range(40):
        ref = [rng.choice(words) for _ in range(rng.randint(1, 9))]
        hyp = [rng.choice(words) for _ in range(rng.randint(0, 9))]
        cer = rng.random() < 0.3
        wer, w = compute_wer(list(hyp), list(ref), cer=cer)
        wer_cases.append(dict(hyp=hyp, ref=ref, cer=cer, wer=wer, n_sub=w["n_sub"], n_ins=w["n_ins"], n_del=w["n_del"],
                              n_ref=w["n_ref"], error_list=w["error_list"]))
    pieces = ["\u2581he", "llo", "\u2581wor", "ld", "<eos>", "\u2581a", "b", "<unk>", "c", "\u2581", "x>"]
    sw_cases = []
    for _ in range(30):
        seq = [rng.choice(pieces) for _ in range(rng.randint(1, 8))]
        sw_cases.append(dict(subwords=seq, words=Vocab.subwords_to_words(None, seq)))
    pack_cases = []
    for _ in range(8):
        n = rng.randint(5, 60)
        xlens = sorted(rng.randint(50, 900) for _ in range(n))
        ylens = [max(1, x // 30 + rng.randint(-2, 2)) for x in xlens]
        prm = make_params(dict(max_xlens_batch=rng.choice([1000, 2500, 4000]), max_ylens_batch=rng.choice([40, 90, 300]),
                               batch_size=rng.choice([3, 8, 50])))
        mbs = rng.choice([1, 1, 2, 3])
        import pandas as pd
        dset = types.SimpleNamespace(data=pd.DataFrame(dict(xlen=xlens, ylen=ylens)))
        sampler = ASRBatchSampler(dset, prm, min_batch_size=mbs)
        pack_cases.append(dict(xlens=xlens, ylens=ylens, max_xlens_batch=prm.max_xlens_batch,
                               max_ylens_batch=prm.max_ylens_batch, batch_size=prm.batch_size, min_batch_size=mbs,
                               batches=sorted(sampler.indices_batches)))
    # ---- dataset items with phone targets and distillation soft labels (datasets.py:25-192,248-263):
    # a tiny manifest + feature files + kd pickle written to a temp dir, read back by the reference
    import pickle
    import tempfile
    from asr.datasets import ASRDataset
    nrng = np.random.RandomState(5)
    kd_case = {}
    with tempfile.TemporaryDirectory() as tmp:
        rows, kd = [], {}
        utts = [("sp0.9-utt-a", 7, [5, 9, 4]), ("utt-b", 5, [11, 3]), ("sp1.1-utt-c", 9, [6, 6, 8, 13])]
        feats = {}
        for uid, T, toks in utts:
            x = nrng.randn(T, 4).astype(np.float32)
            fp = os.path.join(tmp, uid + ".npy")
            np.save(fp, x)
            feats[uid] = x.tolist()
            ph = [int(v) for v in nrng.randint(3, 9, size=len(toks) + 2)]
            rows.append(dict(feat_path=fp, utt_id=uid, token_id=" ".join(map(str, toks)), text="t " + uid, xlen=T,
                             ylen=len(toks), phone_token_id=" ".join(map(str, ph)), phone_text="p " + uid))
            key = "-".join(uid.split("-")[1:]) if uid.startswith("sp") else uid
            if uid != "utt-b":  # one utterance has no teacher output
                kd[key] = [[(int(v), np.float32(pv)) for v, pv in zip(nrng.choice(16, 3, replace=False),
                                                                        nrng.dirichlet(np.ones(3)) * 0.9)]
                           for _ in toks]
        import pandas as pd
        tsv = os.path.join(tmp, "train.tsv")
        pd.DataFrame(rows).to_csv(tsv, sep="\t", index=False)
        kdp = os.path.join(tmp, "kd.pkl")
        with open(kdp, "wb") as f:
            pickle.dump(kd, f)
        for dec in ("ctc", "transformer"):
            prm = make_params(dict(feat_dim=4, num_framestacks=1, vocab_size=16, lsm_prob=0.1, eos_id=2, spec_augment=False,
                                   mtl_phone_ctc_weight=0.3, phone_eos_id=1, kd_weight=0.5, kd_label_path=kdp,
                                   decoder_type=dec))
            dset = ASRDataset(prm, tsv, phase="train")
            batch = ASRDataset.collate_fn([dset[i] for i in range(3)])
            kd_case[dec] = {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in batch.items()}
        kd_case["rows"] = [dict(r, feat_path=os.path.basename(r["feat_path"])) for r in rows]
        kd_case["feats"] = feats
        kd_case["kd"] = {k: [[(v, float(p)) for v, p in pos] for pos in val] for k, val in kd.items()}
    # ---- decode result rows (test_asr.py:96-117: strip <eos>, ids -> string, ids -> text through the vocab)
    # for the greedy hypotheses the reference produced on l2_tiny (tests/golden/l2_tiny.npz)
    from utils.converters import ints2str, strip_eos
    g2 = np.load(os.path.join(OUT, "l2_tiny.npz"))
    toks = ["<blank>", "<unk>", "<eos>"] + [("\u2581" if i % 3 == 0 else "") + "w%d" % i for i in range(3, 40)]
    with tempfile.TemporaryDirectory() as tmp:
        vp = os.path.join(tmp, "vocab.txt")
        with open(vp, "w") as f:
            for i, t in enumerate(toks):
                f.write(f"{t} {i}\n")
        vocab = Vocab(vp)
        hyps, o = [], 0
        for n in g2["eval/hyp_lens"].tolist():
            hyps.append([int(v) for v in g2["eval/hyps"][o:o + n]])
            o += n
        hyps.append([2, 5, 2, 2])   # only <eos> around one token
        hyps.append([])             # nothing decoded
        row_case = dict(vocab=toks, hyps=hyps,
                        rows=[[ints2str(strip_eos(h, 2)), vocab.ids2text(strip_eos(h, 2))] for h in hyps])
    with open(os.path.join(OUT, "hostio.json"), "w") as f:
        json.dump(dict(wer=wer_cases, subwords=sw_cases, packing=pack_cases, dataset_kd=kd_case, result_rows=row_case), f)
    print("hostio:", len(wer_cases), "wer,", len(sw_cases), "subword,", len(pack_cases), "packing cases")


RNNT_BEAM_WIDTHS = [2, 4]


def run_rnnt_beam():
    """RNN-T alignment-length synchronous beam search (rnn_transducer.py:242-325) on the fitted l4_tiny
    weights, one utterance at a time (the reference asserts batch size 1) -> rnntbeam_tiny.npz."""
    g = np.load(os.path.join(OUT, "l4_tiny.npz"))
    model = ASR(make_params(L4), phase="test")
    model.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")})
    model.eval()
    xs, xlens = torch.from_numpy(g["xs"]), torch.from_numpy(g["xlens"])
    out = {}
    with torch.no_grad():
        for bw in RNNT_BEAM_WIDTHS:
            flat, lens, nhyp = [], [], []
            for b in range(xs.shape[0]):
                hyps, _, _, _ = model.decode(xs[b:b + 1, :int(xlens[b])], xlens[b:b + 1], beam_width=bw)
                nhyp.append(len(hyps))
                for h in hyps:
                    lens.append(len(h))
                    flat += [int(v) for v in h]
            out[f"bw{bw}/n_hyps"] = np.array(nhyp)
            out[f"bw{bw}/hyp_lens"] = np.array(lens)
            out[f"bw{bw}/hyps"] = np.array(flat, dtype=np.int64)
            print("rnnt beam", bw, "n_hyps", nhyp, "first hyp lens", lens[:4])
    np.savez_compressed(os.path.join(OUT, "rnntbeam_tiny.npz"), **out)


KD_CTC_CASES = {
    "ctc_all": dict(kd_weight=0.5, reduce_main_loss_kd=False),
    "ctc_mid": dict(kd_weight=0.3, reduce_main_loss_kd=True, kd_ctc_soft_label_weight=0.6, kd_ctc_position="mid"),
}
KD_INTER_CASES = {
    "inter": dict(mtl_inter_ctc_weight=0.3, inter_ctc_layer_id=1),
    "inter_kd": dict(mtl_inter_ctc_weight=0.3, inter_ctc_layer_id=1, inter_kd_weight=0.5, kd_weight=0.5,
                     reduce_main_loss_kd=True),
    "inter_kd_noreduce": dict(mtl_inter_ctc_weight=0.3, inter_ctc_layer_id=2, inter_kd_weight=0.5,
                              reduce_main_loss_kd=False),
    "phone": dict(mtl_phone_ctc_weight=0.3, hie_mtl_phone=True, phone_vocab_size=12, inter_ctc_layer_id=1),
    "phone_top": dict(mtl_phone_ctc_weight=0.2, hie_mtl_phone=False, phone_vocab_size=12, inter_ctc_layer_id=1),
}
KD_GRAD_KEYS = ["decoder.output.weight", "decoder.output.bias", "encoder.norm.weight", "encoder.conv.conv.0.weight",
                "encoder.transformers.1.feed_forward.w2.weight", "encoder.transformers.0.self_attn.linear_q.weight"]


def _soft_labels(seed, B, L, V, topk=5, lsm=0.1):
    """soft labels shaped like datasets.create_soft_label (datasets.py:248-263): top-k teacher probabilities
    scaled by 1-lsm, the remaining mass spread evenly"""
    g = torch.Generator().manual_seed(seed)
    soft = torch.full((B, L, V), lsm / (V - topk))
    for b in range(B):
        for i in range(L):
            idx = torch.randperm(V, generator=g)[:topk]
            pr = torch.softmax(2.0 * torch.randn(topk, generator=g), 0)
            soft[b, i, idx] = pr * (1 - lsm)
    return soft


def run_kd():
    """knowledge-distillation path -> kd_tiny.npz:
    criteria.py losses + ctc_aligner.py on random inputs, and the CTC / Transformer decoders with
    kd_weight > 0 on the l2_tiny / l3_tiny weights (loss, loss_dict, aligns, selected gradients,
    the L2 norm of every gradient)."""
    from asr.criteria import CTCAlignDistillLoss, DistillLoss, RNNTAlignDistillLoss, RNNTWordDistillLoss
    from asr.modeling.decoders.ctc_aligner import CTCForcedAligner
    out = {}
    # ---- forced aligner + label mapping on random posteriors
    g = torch.Generator().manual_seed(11)
    B, T, V, L = 5, 23, 7, 6
    logits = 3.0 * torch.randn(B, T, V, generator=g)
    elens = torch.tensor([23, 19, 12, 7, 23])
    ylens = torch.tensor([6, 4, 3, 1, 2])
    ys = torch.randint(1, V, (B, L), generator=g)
    ys[0, 2] = ys[0, 1]  # a repeated label: needs a blank in between
    for b in range(B):
        ys[b, ylens[b]:] = 0
    aligns = CTCForcedAligner(blank_id=0)(torch.log_softmax(logits, -1), elens, ys, ylens)
    out.update({"align/logits": logits.numpy(), "align/elens": elens.numpy(), "align/ys": ys.numpy(),
                "align/ylens": ylens.numpy(), "align/aligns": aligns.numpy()})
    for pos in ("all", "left", "mid", "right"):
        fn = CTCAlignDistillLoss(vocab_size=V, position=pos)
        maps = torch.full((B, T), -2, dtype=torch.long)
        for b in range(B):
            maps[b, :elens[b]] = fn._frame_to_label_mapping(aligns[b][:elens[b]].long(), int(elens[b]), int(ylens[b]))
        out["align/map_" + pos] = maps.numpy()
    # ---- the four distillation losses: values and logits gradients
    soft = _soft_labels(3, B, L, V, topk=3)
    for name, kw in (("cad_all", dict(soft_label_weight=1.0, position="all", lsm_prob=0.1)),
                     ("cad_right", dict(soft_label_weight=0.4, position="right", lsm_prob=0.1)),
                     ("cad_left_nonorm", dict(soft_label_weight=0.0, position="left", lsm_prob=0.2,
                                              normalize_length=False, normalize_batch=False))):
        z = logits.clone().requires_grad_(True)
        loss = CTCAlignDistillLoss(vocab_size=V, blank_id=0, **kw)(z, ys, soft, aligns, elens, ylens)
        loss.backward()
        out[f"loss/{name}"] = loss.detach().numpy()
        out[f"loss/{name}_grad"] = z.grad.numpy()
    zl = (2.0 * torch.randn(B, L, V, generator=g))
    for name, kw in (("distill", dict(soft_label_weight=0.3, lsm_prob=0.1)),
                     ("distill_len", dict(soft_label_weight=0.7, lsm_prob=0.0, normalize_length=True,
                                          normalize_batch=False))):
        z = zl.clone().requires_grad_(True)
        l, ls, lh = DistillLoss(vocab_size=V, **kw)(z, ys, soft, ylens)
        l.backward()
        out[f"loss/{name}"] = np.array([float(l), float(ls), float(lh)])
        out[f"loss/{name}_grad"] = z.grad.numpy()
    out["loss/dec_logits"] = zl.numpy()
    out["loss/soft"] = soft.numpy()
    Tr = 9
    z4 = 2.0 * torch.randn(B, Tr, L + 1, V, generator=g)
    xl4 = torch.tensor([9, 8, 6, 4, 9])
    z = z4.clone().requires_grad_(True)
    l = RNNTWordDistillLoss()(z, soft, xl4, ylens)
    l.backward()
    out["loss/rnnt_word"] = l.detach().numpy()
    out["loss/rnnt_word_grad"] = z.grad.numpy()
    al4 = torch.stack([torch.sort(torch.randint(0, int(xl4[b]), (L,), generator=g))[0] for b in range(B)])
    z = z4.clone().requires_grad_(True)
    l = RNNTAlignDistillLoss()(z, ys, soft, al4, xl4, ylens)
    l.backward()
    out["loss/rnnt_align"] = l.detach().numpy()
    out["loss/rnnt_align_grad"] = z.grad.numpy()
    out.update({"loss/rnnt_logits": z4.numpy(), "loss/rnnt_xlens": xl4.numpy(), "loss/rnnt_aligns": al4.numpy()})
    # ---- CTC decoder with kd_weight > 0 on the l2_tiny weights
    g2 = np.load(os.path.join(OUT, "l2_tiny.npz"))
    sd2 = {k[3:]: torch.from_numpy(g2[k]) for k in g2.files if k.startswith("sd/")}
    xs, xlens, ys2, ylens2, ys_in, ys_out = make_batch(1, COMMON["feat_dim"], COMMON["vocab_size"])
    soft2 = _soft_labels(5, 4, ys2.shape[1], COMMON["vocab_size"])
    out["model/soft_ctc"] = soft2.numpy()
    for name, extra in KD_CTC_CASES.items():
        model = ASR(make_params(dict(CONFIGS["l2_tiny"], **extra)), phase="train")
        model.load_state_dict(sd2)
        model.train()
        loss, ld = model(xs, xlens, ys2, ylens2, ys_in, ys_out, soft_labels=soft2)
        loss.backward()
        out[f"model/{name}/loss"] = loss.detach().numpy()
        for k, v in ld.items():
            out[f"model/{name}/ld/{k}"] = v.detach().numpy()
        grads = dict((n, p.grad) for n, p in model.named_parameters())
        for k in KD_GRAD_KEYS:
            out[f"model/{name}/grad/{k}"] = grads[k].numpy()
        out[f"model/{name}/gnorms"] = np.array([float(v.norm()) for v in grads.values()])
        with torch.no_grad():
            model.load_state_dict(sd2)
            eouts, elens2, _ = model.encoder(xs, xlens)
            lg = model.decoder.output(eouts)
            out[f"model/{name}/aligns"] = model.decoder.forced_aligner(torch.log_softmax(lg, -1), elens2, ys2,
                                                                      ylens2).numpy()
        print("kd", name, float(loss), {k: float(v) for k, v in ld.items()})
    # ---- intermediate CTC (+ KD on it) and phone-level CTC (ctc.py:129-170), l2_tiny weights
    gph = torch.Generator().manual_seed(21)
    ps = torch.randint(1, 12, (4, 12), generator=gph)
    plens = torch.tensor([12, 9, 7, 3])
    for b in range(4):
        ps[b, plens[b]:] = 0
    out["model/ps"], out["model/plens"] = ps.numpy(), plens.numpy()
    for name, extra in KD_INTER_CASES.items():
        torch.manual_seed(31)
        model = ASR(make_params(dict(CONFIGS["l2_tiny"], **extra)), phase="train")
        missing = model.load_state_dict(sd2, strict=False)
        for k in missing.missing_keys:  # the phone head is not part of l2_tiny: keep its seeded init
            out[f"model/{name}/sd/{k}"] = model.state_dict()[k].clone().numpy()
        model.train()
        loss, ld = model(xs, xlens, ys2, ylens2, ys_in, ys_out, soft_labels=soft2, ps=ps, plens=plens)
        loss.backward()
        out[f"model/{name}/loss"] = loss.detach().numpy()
        for k, v in ld.items():
            out[f"model/{name}/ld/{k}"] = v.detach().numpy()
        grads = dict((n, p.grad) for n, p in model.named_parameters())
        for k in KD_GRAD_KEYS + [k for k in grads if "phone_output" in k]:
            out[f"model/{name}/grad/{k}"] = grads[k].numpy()
        out[f"model/{name}/gnorms"] = np.array([float(v.norm()) for v in grads.values()])
        print("kd", name, float(loss), {k: float(v) for k, v in ld.items()})
    # ---- Transformer decoder with kd_weight > 0 (DistillLoss on ys_out, aux CTC without KD) on l3_tiny
    g3 = np.load(os.path.join(OUT, "l3_tiny.npz"))
    sd3 = {k[3:]: torch.from_numpy(g3[k]) for k in g3.files if k.startswith("sd/") and not k.startswith("sd/lm.")}
    sd3 = {k: v for k, v in sd3.items() if k.startswith("encoder.") or k.startswith("decoder.")}
    soft3 = _soft_labels(6, 4, ys_out.shape[1], COMMON["vocab_size"])
    out["model/soft_att"] = soft3.numpy()
    model = ASR(make_params(dict(L3, kd_weight=0.4, reduce_main_loss_kd=False)), phase="train")
    model.load_state_dict(sd3)
    model.train()
    loss, ld = model(xs, xlens, ys2, ylens2, ys_in, ys_out, soft_labels=soft3)
    loss.backward()
    out["model/att/loss"] = loss.detach().numpy()
    for k, v in ld.items():
        out[f"model/att/ld/{k}"] = v.detach().numpy()
    grads = dict((n, p.grad) for n, p in model.named_parameters())
    for k in ("decoder.output.weight", "decoder.embed.weight", "decoder.transformers.1.src_attn.linear_k.weight",
              "encoder.norm.weight"):
        out[f"model/att/grad/{k}"] = grads[k].numpy()
    out["model/att/gnorms"] = np.array([float(v.norm()) for v in grads.values()])
    print("kd att", float(loss), {k: float(v) for k, v in ld.items()})
    # ---- RNN-T with word-level distillation (rnn_transducer.py:127-141) on the l4_tiny weights; the
    # transducer loss itself is oracle.rnnt.rnnt_loss plugged in for the absent warp_rnnt, as in run_l4
    sys.path.insert(0, "/root/repo")
    from oracle import rnnt as orn
    sys.modules["warp_rnnt"].rnnt_loss = orn.rnnt_loss
    sys.modules["warp_rnnt"].__version__ = "oracle-restatement"
    import asr.modeling.decoders.rnn_transducer as rt
    rt.warp_rnnt = sys.modules["warp_rnnt"]
    g4 = np.load(os.path.join(OUT, "l4_tiny.npz"))
    sd4 = {k[3:]: torch.from_numpy(g4[k]) for k in g4.files if k.startswith("sd/")}
    soft4 = _soft_labels(8, 4, ys2.shape[1], COMMON["vocab_size"])
    out["model/soft_rnnt"] = soft4.numpy()
    for name, extra in (("rnnt_word", dict(kd_weight=0.3, kd_type="word", reduce_main_loss_kd=False)),
                        ("rnnt_word_reduce", dict(kd_weight=0.5, kd_type="word", reduce_main_loss_kd=True))):
        model = ASR(make_params(dict(L4, **extra)), phase="train")
        model.load_state_dict(sd4)
        model.train()
        loss, ld = model(xs, xlens, ys2, ylens2, ys_in, ys_out, soft_labels=soft4)
        loss.backward()
        out[f"model/{name}/loss"] = loss.detach().numpy()
        for k, v in ld.items():
            out[f"model/{name}/ld/{k}"] = v.detach().numpy()
        grads = dict((n, p.grad) for n, p in model.named_parameters())
        for k in ("decoder.output.weight", "decoder.w_dec.weight", "decoder.rnns.1.weight_hh_l0", "decoder.embed.weight",
                  "encoder.norm.weight", "decoder.ctc.output.bias"):
            out[f"model/{name}/grad/{k}"] = grads[k].numpy()
        out[f"model/{name}/gnorms"] = np.array([float(v.norm()) for v in grads.values()])
        print("kd", name, float(loss), {k: float(v) for k, v in ld.items()})
    np.savez_compressed(os.path.join(OUT, "kd_tiny.npz"), **out)


TRAIN_TRACE = dict(lr_schedule_type="noam", learning_rate=0.02, num_warmup_steps=4, accum_grad=2, clip_grad_norm=5.0,
                   weight_decay=1e-6, log_step=100)


def run_train_trace():
    """the micro-batch / accumulate / clip / NaN-skip / noam sequence of asr/train_asr.py:35-97 with the
    reference's ScheduledOptimizer (asr/optimizers.py) around torch.optim.Adam(lr=0, weight_decay) on the
    l2_tiny weights: 12 optimizer steps x accum_grad 2 over three fixed batches -> train_trace.npz
    (train_asr.py itself imports GitPython, which is absent: its train_step sequence is driven from here)."""
    import math
    from asr.optimizers import ScheduledOptimizer
    g2 = np.load(os.path.join(OUT, "l2_tiny.npz"))
    sd2 = {k[3:]: torch.from_numpy(g2[k]) for k in g2.files if k.startswith("sd/")}
    params = make_params(dict(CONFIGS["l2_tiny"], **TRAIN_TRACE))
    model = ASR(params, phase="train")
    model.load_state_dict(sd2)
    model.train()
    optimizer = ScheduledOptimizer(torch.optim.Adam(model.parameters(), lr=0, weight_decay=params.weight_decay), params)
    batches = [make_batch(seed, COMMON["feat_dim"], COMMON["vocab_size"]) for seed in (1, 2, 3)]
    out = {}
    for i, (xs, xlens, ys, ylens, ys_in, ys_out) in enumerate(batches):
        out.update({f"batch{i}/xs": xs.numpy(), f"batch{i}/xlens": xlens.numpy(), f"batch{i}/ys": ys.numpy(),
                    f"batch{i}/ylens": ylens.numpy(), f"batch{i}/ys_in": ys_in.numpy(), f"batch{i}/ys_out": ys_out.numpy()})
    losses, lrs, gnorms = [], [], []
    optimizer.update_epoch()
    for micro in range(24):
        xs, xlens, ys, ylens, ys_in, ys_out = batches[micro % 3]
        loss, loss_dict = model(xs, xlens, ys, ylens, ys_in, ys_out)
        losses.append(loss_dict["loss_total"].item() / params.accum_grad)
        (loss / params.accum_grad).backward()
        if (micro + 1) % params.accum_grad == 0:
            grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), params.clip_grad_norm)
            gnorms.append(float(grad_norm))
            if not math.isnan(grad_norm):
                optimizer.step()
            optimizer.zero_grad()
            lrs.append(optimizer._lr)
    out["losses"], out["lrs"], out["gnorms"] = np.array(losses), np.array(lrs), np.array(gnorms)
    sd_end = model.state_dict()
    for k in ("decoder.output.weight", "encoder.norm.weight", "encoder.transformers.0.conv.batch_norm.running_var",
              "encoder.transformers.1.self_attn.pos_bias_u", "encoder.conv.conv.0.weight",
              "encoder.transformers.0.conv.batch_norm.num_batches_tracked"):
        out["end/" + k] = sd_end[k].numpy()
    osd = optimizer.state_dict()
    names = [n for n, _ in model.named_parameters()]
    idx = names.index("decoder.output.weight")
    out["optim/_step"] = np.array(osd["_step"])
    out["optim/_lr"] = np.array(osd["_lr"])
    out["optim/exp_avg/decoder.output.weight"] = osd["optimizer"]["state"][idx]["exp_avg"].numpy()
    out["optim/exp_avg_sq/decoder.output.weight"] = osd["optimizer"]["state"][idx]["exp_avg_sq"].numpy()
    np.savez_compressed(os.path.join(OUT, "train_trace.npz"), **out)
    print("train trace: losses", [round(v, 3) for v in losses[:4]], "...", [round(v, 3) for v in losses[-2:]],
          "lrs", [round(v, 6) for v in lrs[:5]], "gnorms", [round(v, 2) for v in gnorms[:4]])


TRACE_D256 = dict(input_layer="conv2d", feat_dim=40, num_framestacks=1, encoder_type="conformer", decoder_type="ctc",
                  pos_encode_type="rel", enc_hidden_size=256, enc_num_attention_heads=4, enc_num_layers=2,
                  enc_intermediate_size=256, dropout_enc_rate=0.0, dropout_attn_rate=0.0, dropout_dec_rate=0.0, vocab_size=40,
                  blank_id=0, eos_id=2, kd_weight=0, lsm_prob=0.1, **TRAIN_TRACE)


def run_train_trace_d256():
    """as run_train_trace at enc_hidden_size 256 -- the width the stacked bf16 engine (engine.ctc_train_stacked, the benched
    path) accepts -- so that tests/test_stacked_oracle_gpu.py can replay the REFERENCE's optimizer trace through it.  The model
    has 4 M parameters: the fixture holds no weights, the initial state is tests/util.py: synthetic_state (a seeded rule both
    sides apply), a few of its tensors are stored to prove the rule reproduced them -> train_trace_d256.npz"""
    import math
    sys.path.insert(1, os.path.dirname(os.path.dirname(OUT)))
    from tests.util import synthetic_state
    from asr.optimizers import ScheduledOptimizer
    params = make_params(TRACE_D256)
    model = ASR(params, phase="train")
    sd0 = synthetic_state({k: v.shape for k, v in model.state_dict().items()})
    model.load_state_dict(sd0)
    model.train()
    optimizer = ScheduledOptimizer(torch.optim.Adam(model.parameters(), lr=0, weight_decay=params.weight_decay), params)
    batches = [make_batch(seed, 40, 40) for seed in (11, 12, 13)]
    out = {}
    for i, (xs, xlens, ys, ylens, ys_in, ys_out) in enumerate(batches):
        out.update({f"batch{i}/xs": xs.numpy(), f"batch{i}/xlens": xlens.numpy(), f"batch{i}/ys": ys.numpy(),
                    f"batch{i}/ylens": ylens.numpy(), f"batch{i}/ys_in": ys_in.numpy(), f"batch{i}/ys_out": ys_out.numpy()})
    for k in ("encoder.norm.weight", "encoder.transformers.1.self_attn.pos_bias_u", "decoder.output.bias",
              "encoder.transformers.0.conv.depthwise_conv.weight"):
        out["init/" + k] = sd0[k].numpy()
    losses, lrs, gnorms = [], [], []
    optimizer.update_epoch()
    for micro in range(24):
        xs, xlens, ys, ylens, ys_in, ys_out = batches[micro % 3]
        loss, loss_dict = model(xs, xlens, ys, ylens, ys_in, ys_out)
        losses.append(loss_dict["loss_total"].item() / params.accum_grad)
        (loss / params.accum_grad).backward()
        if (micro + 1) % params.accum_grad == 0:
            grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), params.clip_grad_norm)
            gnorms.append(float(grad_norm))
            if not math.isnan(grad_norm):
                optimizer.step()
            optimizer.zero_grad()
            lrs.append(optimizer._lr)
    out["losses"], out["lrs"], out["gnorms"] = np.array(losses), np.array(lrs), np.array(gnorms)
    sd_end = model.state_dict()
    for k in ("decoder.output.weight", "encoder.norm.weight", "encoder.transformers.0.conv.batch_norm.running_var",
              "encoder.transformers.0.conv.batch_norm.running_mean", "encoder.transformers.1.self_attn.pos_bias_u",
              "encoder.conv.conv.0.weight", "encoder.transformers.1.feed_forward.w2.weight",
              "encoder.transformers.0.self_attn.linear_pos.weight", "encoder.transformers.1.conv.depthwise_conv.weight",
              "encoder.transformers.0.conv.batch_norm.num_batches_tracked"):
        out["end/" + k] = sd_end[k].numpy()
    out["optim/_step"] = np.array(optimizer.state_dict()["_step"])
    np.savez_compressed(os.path.join(OUT, "train_trace_d256.npz"), **out)
    print("train trace d256: losses", [round(v, 3) for v in losses[:4]], "...", [round(v, 3) for v in losses[-2:]],
          "lrs", [round(v, 6) for v in lrs[:5]], "gnorms", [round(v, 2) for v in gnorms[:4]])


def run_rnnt_align_xcheck():
    """CROSS-CHECK, not a pin: asr/modeling/decoders/rnnt_aligner.py:14-198 holds the only restatement of the transducer lattice
    inside the reference (two Numba CUDA kernels + the alpha+beta walk); numba is absent and there is no GPU here.  The kernels'
    Python bodies are executed as they stand, one (block, thread) after the other: `numba.cuda` is replaced by a shim whose
    `jit` returns a launcher that sets blockIdx / threadIdx and calls the function -- threads u = 0..U in order for the forward
    kernel, u = U..0 for the backward kernel, so that every spin-lock test (`cuda.atomic.add(lock, ., 0) < 0`) succeeds at its
    first try (the shim raises if one would spin).  RNNTForcedAligner.__call__ then runs unchanged on CPU tensors.
    -> rnnt_align_xcheck.npz: inputs, alpha, beta, log_p (both directions), best alignments."""
    import importlib

    class _Dim:
        x = 0

    class _Atomic:
        spins = 0

        @staticmethod
        def add(arr, idx, val):
            old = int(arr[idx])
            if val == 0 and old >= 0:
                _Atomic.spins += 1
                if _Atomic.spins > 10000:
                    raise RuntimeError("a thread would spin: the sequential order does not satisfy the lock protocol")
            arr[idx] = old + val
            return old

    cuda = types.ModuleType("numba.cuda")
    cuda.blockIdx, cuda.threadIdx, cuda.atomic = _Dim(), _Dim(), _Atomic

    def jit(_sig):
        def deco(fn):
            class Launcher:
                def __getitem__(self, cfg):
                    nblk, nthr = cfg

                    def launch(*args):
                        order = range(nthr) if "forward" in fn.__name__ else reversed(range(nthr))
                        order = list(order)
                        for b in range(nblk):
                            for u in order:
                                cuda.blockIdx.x, cuda.threadIdx.x = b, u
                                fn(*args)
                    return launch
            return Launcher()
        return deco

    cuda.jit = jit
    numba = types.ModuleType("numba")
    numba.cuda = cuda
    sys.modules["numba"], sys.modules["numba.cuda"] = numba, cuda
    ra = importlib.import_module("asr.modeling.decoders.rnnt_aligner")
    # two sets: the round-4 lattices (3 x 7 x 5 x 6), and a larger ragged set -- 6 utterances up to 23 frames x 11 labels over 13
    # classes, among them a single-frame utterance, a single-label one and one whose labels fill every frame but one
    sets = [("rnnt_align_xcheck.npz", 7, (3, 7, 4, 6), [7, 5, 3], [4, 2, 3]),
            ("rnnt_align_xcheck2.npz", 11, (6, 23, 11, 13), [23, 17, 1, 9, 12, 20], [11, 5, 1, 1, 11, 3])]
    for fname, seed, (B, T, L, V), el, yl in sets:
        g = torch.Generator().manual_seed(seed)
        lp = torch.log_softmax(2.0 * torch.randn(B, T, L + 1, V, generator=g), -1)
        ys = torch.randint(1, V, (B, L), generator=g)
        elens, ylens = torch.tensor(el), torch.tensor(yl)
        # the aligner's own buffers, reproduced here so that alpha / beta / log_p can be stored as well (rnnt_aligner.py:158-181)
        alpha = torch.zeros(B, T, L + 1)
        beta = torch.zeros(B, T, L + 1)
        lpa, lpb = torch.zeros(B), torch.zeros(B)
        lock = torch.zeros(B, L + 1, dtype=torch.int32)
        ra.cu_kernel_forward[B, L + 1](lp, ys.int(), alpha, lpa, elens.int(), ylens.int(), 0, lock)
        lock = lock * 0
        ra.cu_kernel_backward[B, L + 1](lp, ys.int(), beta, lpb, elens.int(), ylens.int(), 0, lock)
        aligns = ra.RNNTForcedAligner(blank_id=0)(lp, elens, ys, ylens)
        np.savez_compressed(os.path.join(OUT, fname), log_probs=lp.numpy(), ys=ys.numpy(), elens=elens.numpy(),
                            ylens=ylens.numpy(), alpha=alpha.numpy(), beta=beta.numpy(), log_p_alpha=lpa.numpy(),
                            log_p_beta=lpb.numpy(), aligns=aligns.numpy())
        print(fname, "rnnt aligner cross-check: log_p alpha", lpa.tolist(), "beta", lpb.tolist(), "aligns", aligns.tolist(),
              "lock tests that found the lock closed:", _Atomic.spins)


if __name__ == "__main__":
    which = sys.argv[1:] or ["ctc", "ctcabs", "l3", "l4"]
    if "ctc" in which:
        for name, cfg in CONFIGS.items():
            run_ctc(name, cfg)
    if "ctcabs" in which:
        # Conformer with ABSOLUTE positions: the layer runs convolution BEFORE self-attention and uses the plain
        # MultiHeadedAttention (conformer.py:209-219), the encoder adds the absolute table (encoders/transformer.py:96-99)
        run_ctc("l2abs_tiny", dict(CONFIGS["l2_tiny"], pos_encode_type="abs"))
    if "l3" in which:
        run_l3()
    if "l4" in which:
        run_l4()
    if "hostio" in which:
        run_hostio()
    if "traintrace" in which:  # needs l2_tiny.npz
        run_train_trace()
    if "rnntalign" in which:
        run_rnnt_align_xcheck()
    if "traintrace256" in which:
        run_train_trace_d256()
    if "kd" in which:  # needs l2_tiny.npz and l3_tiny.npz (reads their weights)
        run_kd()
    if "rnntbeam" in which:  # needs l4_tiny.npz (reads its weights)
        run_rnnt_beam()
    if "ctcbeam" in which:  # needs l2_tiny.npz and l3_tiny.npz (reads their weights)
        run_ctc_beam()
