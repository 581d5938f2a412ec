"""On-disk data formats of the reference, host side: the TSV manifest + per-utterance `.npy` feature
files (asr/datasets.py:25-177), the length-budgeted batch sampler (:180-245), the decode result TSV
(asr/test_asr.py:265-313) and the vocabulary file (utils/vocab.py).

    manifest columns   feat_path  utt_id  token_id  text  xlen  ylen     (tab separated, header row)
    feat_path          float array [frames, >= feat_dim] saved with np.save; the first feat_dim columns are used
    token_id           space-separated integer ids (no <eos>)

Differences from the reference are deliberate and local: SpecAugment is NOT applied per utterance on
the CPU here -- the batch's mask spans are sampled with the reference's draws (data.specaug_spans) and
applied by one HIP kernel on the padded batch (ops.specaug_apply).

Optional inputs of the auxiliary branches, same formats as the reference: phone targets (manifest columns
phone_token_id / phone_text, datasets.py:43-64,108-116 -> batch keys ps / plens / ptexts, padded with
params.phone_eos_id) and knowledge-distillation soft labels (params.kd_label_path: a pickle
{utt_id without speed-perturbation prefix: [per label position: [(token id, probability), ...]]},
datasets.py:70-79,118-131,248-263 -> batch key soft_labels [B, L(+1), V]).
"""
import logging
import random

import numpy as np
import torch

from .data import pack_batches


def _read_table(path):
    import pandas as pd
    return pd.read_table(path, comment="#")


def get_utt_id_nosp(utt_id):
    """utils/converters.py:17-26: drop the speed-perturbation prefix "sp0.9-" / "sp1.0-" / "sp1.1-" """
    if utt_id.startswith(("sp0.9", "sp1.0", "sp1.1")):
        return "-".join(utt_id.split("-")[1:])
    return utt_id


def create_soft_label(data_kd_utt, ylen, vocab_size, lsm_prob, add_eos=False, eos_id=2):
    """datasets.py:248-263: per label position the teacher's top-k probabilities scaled by 1-lsm_prob, the
    smoothing mass spread over the other classes; positions the teacher does not cover stay all-zero;
    with add_eos one more row for <eos> (attention decoders: same length as ys_out)."""
    soft = torch.zeros(ylen + 1 if add_eos else ylen, vocab_size)
    for i, topk in enumerate(data_kd_utt):
        soft[i, :] = lsm_prob / (vocab_size - len(topk))
        for v, prob in topk:
            soft[i, v] = float(np.float64(prob)) * (1 - lsm_prob)
    if add_eos:
        soft[-1, :] = lsm_prob / (vocab_size - 1)
        soft[-1, eos_id] = 1.0 * (1 - lsm_prob)
    return soft


class ASRDataset:
    """Items: (utt_id, x float32 [T, feat_dim * num_framestacks], xlen, y int64 [L], ylen, text,
    p int64 [P] | None, plen | None, ptext | None, soft_label f32 [L(+1), V] | None)."""

    COLUMNS = ["feat_path", "utt_id", "token_id", "text", "xlen", "ylen"]

    def __init__(self, params, data_path, phase="train", size=-1, decode_phone=False):
        self.feat_dim = params.feat_dim
        self.num_framestacks = getattr(params, "num_framestacks", 1)
        self.vocab_size = getattr(params, "vocab_size", None)
        self.lsm_prob = getattr(params, "lsm_prob", 0.0)
        self.eos_id = params.eos_id
        self.phase = phase
        columns = list(self.COLUMNS)
        self.with_phones = (phase == "train" and getattr(params, "mtl_phone_ctc_weight", 0) > 0) or decode_phone
        if self.with_phones:
            columns += ["phone_token_id", "phone_text"]
            self.phone_eos_id = params.phone_eos_id
        self.data = _read_table(data_path)[columns]
        self.data_kd = None
        if phase == "train" and (getattr(params, "kd_weight", 0) > 0 or getattr(params, "inter_kd_weight", 0) > 0):
            import pickle
            with open(params.kd_label_path, "rb") as f:
                self.data_kd = pickle.load(f)
            logging.info(f"kd labels: {params.kd_label_path}")
            self.add_eos = params.decoder_type in ["transformer", "las"]
        if size > 0:
            self.data = self.data[:size]
        self.data = self.data.reset_index(drop=True)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        row = self.data.loc[idx]
        x = torch.from_numpy(np.load(row["feat_path"])[:, : self.feat_dim].astype(np.float32))
        if self.num_framestacks > 1:  # datasets.py:134-142: drop the ragged tail, concatenate consecutive frames
            n = x.shape[0] // self.num_framestacks
            x = x[: n * self.num_framestacks].reshape(n, self.feat_dim * self.num_framestacks)
        y = torch.tensor([int(t) for t in str(row["token_id"]).split()], dtype=torch.long)
        p = plen = ptext = soft = None
        if self.with_phones:
            p = torch.tensor([int(t) for t in str(row["phone_token_id"]).split()], dtype=torch.long)
            plen, ptext = p.shape[0], row["phone_text"]
        if self.data_kd is not None:
            key = get_utt_id_nosp(row["utt_id"])
            if key not in self.data_kd:
                logging.warning(f"soft label: {key} not found")
            soft = create_soft_label(self.data_kd.get(key, []), y.shape[0], self.vocab_size, self.lsm_prob,
                                     add_eos=self.add_eos, eos_id=self.eos_id)
        return row["utt_id"], x, x.shape[0], y, y.shape[0], row["text"], p, plen, ptext, soft

    def collate_fn(self, batch):
        """dict with the reference's keys (datasets.py:144-177): xs zero-padded, ys padded with <eos>
        (no <eos> appended), ys_in = <eos> + y, ys_out = y + <eos> (both <eos>-padded, length ylen+1)."""
        utt_ids, xs, xlens, ys, ylens, texts, ps, plens, ptexts, softs = zip(*batch)
        eos = self.eos_id
        B, T, L = len(xs), max(xlens), max(ylens)
        xpad = torch.zeros(B, T, xs[0].shape[1])
        ypad = torch.full((B, L), eos, dtype=torch.long)
        yin = torch.full((B, L + 1), eos, dtype=torch.long)
        yout = torch.full((B, L + 1), eos, dtype=torch.long)
        for b in range(B):
            xpad[b, : xlens[b]] = xs[b]
            ypad[b, : ylens[b]] = ys[b]
            yin[b, 1: ylens[b] + 1] = ys[b]
            yout[b, : ylens[b]] = ys[b]
        ret = {"utt_ids": list(utt_ids), "texts": list(texts), "xs": xpad, "xlens": torch.tensor(xlens),
               "ys": ypad, "ylens": torch.tensor(ylens, dtype=torch.long), "ys_in": yin, "ys_out": yout}
        if ps[0] is not None:
            ppad = torch.full((B, max(plens)), self.phone_eos_id, dtype=torch.long)
            for b in range(B):
                ppad[b, : plens[b]] = ps[b]
            ret.update(ps=ppad, plens=torch.tensor(plens), ptexts=list(ptexts))
        if softs[0] is not None:
            spad = torch.zeros(B, max(s.shape[0] for s in softs), softs[0].shape[1])
            for b in range(B):
                spad[b, : softs[b].shape[0]] = softs[b]
            ret["soft_labels"] = spad
        return ret


class ASRBatchSampler:
    """Consecutive utterances of the (length-sorted) manifest are packed until the next one would exceed
    max_xlens_batch input frames, max_ylens_batch labels or batch_size utterances; batches with fewer
    than min_batch_size utterances are dropped (datasets.py:200-234).  The batch ORDER is reshuffled at
    every epoch (:236-242), the composition of each batch never changes."""

    def __init__(self, dataset, params, min_batch_size=1, seed=0):
        xlens, ylens = dataset.data["xlen"].values, dataset.data["ylen"].values
        assert xlens.max(initial=0) <= params.max_xlens_batch and ylens.max(initial=0) <= params.max_ylens_batch, \
            "an utterance exceeds the per-batch budget"
        self.indices_batches = pack_batches(xlens, ylens, params.max_xlens_batch, params.max_ylens_batch,
                                            params.batch_size, min_batch_size)
        dropped = len(xlens) - sum(len(b) for b in self.indices_batches)
        if dropped:
            logging.warning(f"{dropped} utterances are skipped because their batch is smaller than min_batch_size")
        self._rng = random.Random(seed)

    def __iter__(self):
        self._rng.shuffle(self.indices_batches)
        return iter(self.indices_batches)

    def __len__(self):
        return len(self.indices_batches)

    def shard(self, rank, world):
        """this rank's batches for one-process-per-GPU data parallelism: batch i goes to rank i % world
        (every rank sees the same shuffled order; the tail is dropped so all ranks step equally often)"""
        n = len(self.indices_batches) // world * world
        return [self.indices_batches[i] for i in range(rank, n, world)]


def batches(dataset, sampler):
    for idx in sampler:
        yield dataset.collate_fn([dataset[i] for i in idx])


class Vocab:
    """`token id` per line (utils/vocab.py:5-43); sentencepiece-style subwords -> words (:45-64)."""

    def __init__(self, vocab_path):
        self.i2t, self.t2i = {}, {}
        with open(vocab_path) as f:
            for line in f:
                if not line.strip():
                    continue
                token, idx = line.split()
                self.i2t[int(idx)] = token
                self.t2i[token] = int(idx)
        self.unk_id = self.t2i["<unk>"]

    def id2token(self, idx):
        return self.i2t[idx]

    def ids2tokens(self, ids):
        return [self.i2t[i] for i in ids]

    def token2id(self, token):
        return self.t2i.get(token, self.unk_id)

    def tokens2ids(self, tokens):
        return [self.token2id(t) for t in tokens]

    def ids2words(self, ids):
        return self.subwords_to_words(self.ids2tokens(ids))

    def ids2text(self, ids):
        return " ".join(self.ids2words(ids))

    @staticmethod
    def subwords_to_words(subwords):
        """a new word starts at a piece beginning with the sentencepiece marker or with `<`, and right
        after a piece ending in `>` (special tokens stand alone)"""
        words, cur = [], ""
        for piece in subwords:
            if piece[0] in ("▁", "<") or (cur and cur[-1] == ">"):
                if cur:
                    words.append(cur)
                cur = piece[1:] if piece[0] == "▁" else piece
            else:
                cur += piece
        if cur:
            words.append(cur)
        return words


def write_results_tsv(path, rows, comment=None):
    """decode results in the reference's layout (test_asr.py:265-313): utt_id, token_id, text, reftext
    (+ any extra keys of the row dicts), tab separated; an optional `# ...` comment line (WER summary)
    goes first, as utils/log.py:insert_comment does."""
    keys = ["utt_id", "token_id", "text", "reftext"]
    extra = [k for k in rows[0] if k not in keys] if rows else []
    with open(path, "w") as f:
        if comment:
            f.write("# " + comment + "\n")
        f.write("\t".join(keys + extra) + "\n")
        for r in rows:
            tok = r["token_id"]
            tok = tok if isinstance(tok, str) else " ".join(str(int(t)) for t in tok)
            f.write("\t".join([str(r["utt_id"]), tok, str(r["text"]), str(r["reftext"])] + [str(r[k]) for k in extra]) + "\n")
