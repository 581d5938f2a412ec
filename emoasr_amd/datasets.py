"""On-disk data formats of the reference, host side: the TSV manifest + per-utterance `.npy` feature
files (asr/datasets.py:25-177), the length-budgeted batch sampler (:180-245), the decode result TSV
(asr/test_asr.py:265-313) and the vocabulary file (utils/vocab.py).

    manifest columns   feat_path  utt_id  token_id  text  xlen  ylen     (tab separated, header row)
    feat_path          float array [frames, >= feat_dim] saved with np.save; the first feat_dim columns are used
    token_id           space-separated integer ids (no <eos>)

Differences from the reference are deliberate and local: SpecAugment is NOT applied per utterance on
the CPU here -- the batch's mask spans are sampled with the reference's draws (data.specaug_spans) and
applied by one HIP kernel on the padded batch (ops.specaug_apply); the knowledge-distillation soft
labels and the phone targets are outside the hot path.
"""
import logging
import random

import numpy as np
import torch

from .data import pack_batches


def _read_table(path):
    import pandas as pd
    return pd.read_table(path, comment="#")


class ASRDataset:
    """Items: (utt_id, x float32 [T, feat_dim * num_framestacks], xlen, y int64 [L], ylen, text)."""

    COLUMNS = ["feat_path", "utt_id", "token_id", "text", "xlen", "ylen"]

    def __init__(self, params, data_path, phase="train", size=-1):
        self.feat_dim = params.feat_dim
        self.num_framestacks = getattr(params, "num_framestacks", 1)
        self.eos_id = params.eos_id
        self.phase = phase
        self.data = _read_table(data_path)[self.COLUMNS]
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
        return row["utt_id"], x, x.shape[0], y, y.shape[0], row["text"]

    def collate_fn(self, batch):
        """dict with the reference's keys (datasets.py:144-177): xs zero-padded, ys padded with <eos>
        (no <eos> appended), ys_in = <eos> + y, ys_out = y + <eos> (both <eos>-padded, length ylen+1)."""
        utt_ids, xs, xlens, ys, ylens, texts = zip(*batch)
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
        return {"utt_ids": list(utt_ids), "texts": list(texts), "xs": xpad, "xlens": torch.tensor(xlens),
                "ys": ypad, "ylens": torch.tensor(ylens, dtype=torch.long), "ys_in": yin, "ys_out": yout}


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
