"""Host-side formats of the reference: manifest + .npy reader and batch
sampler, vocabulary, result TSV, WER, checkpoint averaging / resume discovery.  CPU only.
Known answers come from the reference's own behaviour restated by hand (small enough to verify by eye)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from emoasr_amd import checkpoint as ck
from emoasr_amd import datasets as ds
from emoasr_amd import metrics as mt


def _write_corpus(tmp_path, n=7, feat_dim=6):
    rows = []
    rng = np.random.RandomState(0)
    for i in range(n):
        T = 20 + 7 * i
        np.save(tmp_path / f"u{i}.npy", rng.randn(T, feat_dim + 2).astype(np.float32))
        toks = rng.randint(3, 30, size=2 + i % 3)
        rows.append((str(tmp_path / f"u{i}.npy"), f"utt{i}", " ".join(map(str, toks)), f"text {i}", T, len(toks)))
    path = tmp_path / "train.tsv"
    with open(path, "w") as f:
        f.write("feat_path\tutt_id\ttoken_id\ttext\txlen\tylen\n")
        for r in rows:
            f.write("\t".join(map(str, r)) + "\n")
    return path, rows


def test_dataset_collate_and_sampler(tmp_path):
    path, rows = _write_corpus(tmp_path)
    params = SimpleNamespace(feat_dim=6, num_framestacks=1, eos_id=2, max_xlens_batch=100, max_ylens_batch=7, batch_size=3)
    data = ds.ASRDataset(params, str(path))
    assert len(data) == 7
    utt, x, xlen, y, ylen, text, p, plen, ptext, soft = data[2]  # the reference's 10-tuple (datasets.py:133)
    assert p is None and plen is None and ptext is None and soft is None
    assert utt == "utt2" and x.shape == (34, 6) and xlen == 34 and ylen == 4 and text == "text 2"
    assert torch.equal(x, torch.from_numpy(np.load(rows[2][0])[:, :6]))
    sampler = ds.ASRBatchSampler(data, params)
    # xlens 20,27,34,41,48,55,62 / ylens 2,3,4,2,3,4,2 with budgets 100 frames / 7 labels / 3 utterances:
    # [0,1] (5 labels; +4 > 7 labels), [2,3] (6 labels; +3 > 7), [4] (48 frames; +55 > 100 frames), [5] (55; +62 > 100), [6]
    assert sorted(map(tuple, sampler.indices_batches)) == [(0, 1), (2, 3), (4,), (5,), (6,)]
    assert len(ds.ASRBatchSampler(data, params, min_batch_size=2)) == 2  # single-utterance batches are dropped
    wide = SimpleNamespace(**dict(vars(params), max_ylens_batch=100))
    # frame budget: [0,1,2] (81; +41 > 100), [3,4] (89; +55 > 100), [5] (55; +62 > 100), [6]
    assert sorted(map(tuple, ds.ASRBatchSampler(data, wide).indices_batches)) == [(0, 1, 2), (3, 4), (5,), (6,)]
    b = data.collate_fn([data[i] for i in (0, 1, 2)])
    assert b["xs"].shape == (3, 34, 6) and b["xlens"].tolist() == [20, 27, 34]
    assert float(b["xs"][0, 20:].abs().max()) == 0.0
    y0 = [int(t) for t in rows[0][2].split()]
    assert b["ys"][0].tolist() == y0 + [2] * (4 - len(y0))
    assert b["ys_in"][0].tolist() == [2] + y0 + [2] * (4 - len(y0))
    assert b["ys_out"][0].tolist() == y0 + [2] * (5 - len(y0))
    assert b["ys_in"].shape == (3, 5) and b["ylens"].tolist() == [2, 3, 4]
    sh0, sh1 = sampler.shard(0, 2), sampler.shard(1, 2)
    assert len(sh0) == len(sh1) == 2 and not set(map(tuple, sh0)) & set(map(tuple, sh1))


def test_frame_stacking(tmp_path):
    path, rows = _write_corpus(tmp_path, n=1)
    data = ds.ASRDataset(SimpleNamespace(feat_dim=6, num_framestacks=3, eos_id=2), str(path))
    _, x, xlen = data[0][:3]
    raw = np.load(rows[0][0])[:, :6]
    assert x.shape == (6, 18) and xlen == 6  # 20 frames -> 6 stacks of 3 (2 dropped)
    assert np.array_equal(x[1].numpy(), raw[3:6].reshape(-1))


def test_vocab_and_results_tsv(tmp_path):
    vp = tmp_path / "vocab.txt"
    toks = ["<unk>", "<pad>", "<eos>", "▁he", "llo", "▁wor", "ld", "▁a"]
    vp.write_text("".join(f"{t} {i}\n" for i, t in enumerate(toks)))
    v = ds.Vocab(str(vp))
    assert v.ids2words([3, 4, 5, 6, 7]) == ["hello", "world", "a"]
    assert v.ids2text([7, 2, 3]) == "a <eos> he"  # special tokens stand alone
    assert v.tokens2ids(["▁a", "zzz"]) == [7, 0]
    out = tmp_path / "res.tsv"
    ds.write_results_tsv(str(out), [dict(utt_id="u1", token_id=[3, 4], text="hello", reftext="hello world")],
                         comment="WER: 50.00")
    import pandas as pd
    df = pd.read_table(out, comment="#")
    assert list(df.columns) == ["utt_id", "token_id", "text", "reftext"] and df.loc[0, "token_id"] == "3 4"
    wer, w = mt.compute_wers_df(df)
    assert wer == 50.0 and (w["n_del"], w["n_sub"], w["n_ins"], w["n_ref"]) == (1, 0, 0, 2)


def test_wer_known_answers():
    wer, w = mt.compute_wer("a b c d".split(), "a x c d e".split())
    assert w["error_list"] == ["C", "S", "C", "C", "D"] and wer == pytest.approx(40.0)
    wer, w = mt.compute_wer([], ["a", "b"])  # empty hypothesis = one dummy word: 1 sub + 1 del
    assert (w["n_sub"], w["n_del"], w["n_ins"]) == (1, 1, 0) and wer == 100.0
    wer, w = mt.compute_wer(["ab", "c"], ["a", "bc"], cer=True)  # characters: "abc" vs "abc"
    assert wer == 0.0 and w["n_ref"] == 3
    wer, w = mt.compute_wers([[1, 2, 3], [4]], [[1, 3], [4, 5]])
    assert (w["n_ins"], w["n_del"], w["n_ref"]) == (1, 1, 4) and wer == 50.0
    assert mt.wer_summary(wer, w) == "WER: 50.00 [D=1, S=0, I=1, N=4]"


def test_checkpoint_average_and_resume(tmp_path):
    sds = []
    for e in (1, 2, 3):
        sd = {"w": torch.full((2, 2), float(e)), "bn.num_batches_tracked": torch.tensor(10 * e)}
        torch.save(sd, tmp_path / f"model.ep{e}")
        torch.save({"_step": e}, tmp_path / f"optim.ep{e}")
        sds.append(sd)
    assert ck.parse_epochs("2-4") == [2, 3, 4] and ck.parse_epochs("1+3") == [1, 3] and ck.parse_epochs("5") is None
    out = ck.model_average(str(tmp_path), "1-3")
    avg = torch.load(out)
    assert torch.equal(avg["w"], torch.full((2, 2), 2.0)) and float(avg["bn.num_batches_tracked"]) == 20.0
    assert ck.model_average(str(tmp_path), "2") is None
    mp, op, ep = ck.resume_paths(str(tmp_path))
    assert ep == 3 and mp.endswith("model.ep3") and op.endswith("optim.ep3")  # the averaged file is not an epoch
    assert ck.resume_paths(str(tmp_path / "nothing"))[2] == 0


def test_parameter_order_matches_reference():
    """torch optimizers index parameters by position in model.parameters(); the golden fixtures keep the
    reference's state_dict key order, so equal parameter-name order means optim.ep{N} files map 1:1"""
    from tests.util import CONFIGS
    from emoasr_amd.modeling.asr import ASR
    for name in ("l1_tiny", "l2_tiny", "l3_tiny", "l4_tiny"):
        z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
        ref_keys = [k[3:] for k in z.files if k.startswith("sd/")]
        model = ASR(SimpleNamespace(**CONFIGS[name]), compute_dtype=torch.float32)
        assert list(model.state_dict().keys()) == ref_keys, name
        ref_params = [k for k in ref_keys if "running_" not in k and "num_batches_tracked" not in k]
        assert [n for n, _ in model.named_parameters()] == ref_params, name


def test_against_reference_known_answers():
    """tests/golden/hostio.json: outputs of the reference's compute_wer / subwords_to_words / ASRBatchSampler
    on seeded random inputs (tests/golden/make_golden.py hostio)"""
    import json
    from emoasr_amd.data import pack_batches
    with open(os.path.join(os.path.dirname(__file__), "golden", "hostio.json")) as f:
        g = json.load(f)
    for c in g["wer"]:
        wer, w = mt.compute_wer(c["hyp"], c["ref"], cer=c["cer"])
        assert w["error_list"] == c["error_list"], c
        assert (w["n_sub"], w["n_ins"], w["n_del"], w["n_ref"]) == (c["n_sub"], c["n_ins"], c["n_del"], c["n_ref"])
        assert wer == pytest.approx(c["wer"])
    for c in g["subwords"]:
        assert ds.Vocab.subwords_to_words(c["subwords"]) == c["words"], c
    for c in g["packing"]:
        got = pack_batches(np.array(c["xlens"]), np.array(c["ylens"]), c["max_xlens_batch"], c["max_ylens_batch"],
                           c["batch_size"], c["min_batch_size"])
        assert sorted(got) == c["batches"], c


def test_scheduled_optimizer_rates_and_state():
    """asr/optimizers.py:45-117: the noam rates of the reference's 12-step trace (train_trace.npz), the
    other two schedules by their formulas, and the state_dict round trip"""
    import numpy as np
    from types import SimpleNamespace
    from emoasr_amd.optimizers import ScheduledOptimizer

    class Dummy:
        def __init__(self):
            self.param_groups = [{"lr": 0.0}]
            self.steps = 0

        def step(self):
            self.steps += 1

        def zero_grad(self):
            pass

        def state_dict(self):
            return {"steps": self.steps}

        def load_state_dict(self, sd):
            self.steps = sd["steps"]

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_trace.npz"))
    prm = SimpleNamespace(lr_schedule_type="noam", learning_rate=0.02, num_warmup_steps=4, enc_hidden_size=128)
    opt = ScheduledOptimizer(Dummy(), prm)
    got = []
    for _ in range(12):
        opt.step()
        got.append(opt.param_groups[0]["lr"])
    assert np.allclose(got, z["lrs"], rtol=1e-12, atol=0)
    sd = opt.state_dict()
    assert sd["_step"] == 12 and sd["optimizer"] == {"steps": 12} and sd["num_warmup_steps"] == 4
    opt2 = ScheduledOptimizer(Dummy(), prm)
    opt2.load_state_dict(sd)
    opt2.step()
    opt.step()
    assert opt2._lr == opt._lr and opt2.optimizer.steps == 13
    ep = ScheduledOptimizer(Dummy(), SimpleNamespace(lr_schedule_type="epdecay", learning_rate=1.0, num_warmup_steps=2,
                                                     lr_decay_start_epoch=2, lr_decay_rate=0.5))
    rates = []
    for _ in range(3):
        ep.step()
        rates.append(ep._lr)
    assert rates == [0.5, 1.0, 1.0]
    ep.update_epoch()
    assert ep._lr == 1.0
    ep.update_epoch()
    assert ep._lr == 0.5 and ep.param_groups[0]["lr"] == 0.5
    lin = ScheduledOptimizer(Dummy(), SimpleNamespace(lr_schedule_type="lindecay", learning_rate=1.0, warmup_proportion=0.2),
                             num_total_steps=10)
    rates = []
    for _ in range(10):
        lin.step()
        rates.append(lin._lr)
    assert np.allclose(rates, [0.5, 1.0, 0.875, 0.75, 0.625, 0.5, 0.375, 0.25, 0.125, 0.0])


@pytest.mark.parametrize("decoder_type", ["ctc", "transformer"])
def test_dataset_phone_targets_and_soft_labels(tmp_path, decoder_type):
    """items + collate with phone targets and distillation soft labels (datasets.py:25-192,248-263) against
    the batch the reference's ASRDataset built from the same manifest / features / kd pickle"""
    import json
    import pickle
    import pandas as pd
    with open(os.path.join(os.path.dirname(__file__), "golden", "hostio.json")) as f:
        case = json.load(f)["dataset_kd"]
    rows = []
    for r in case["rows"]:
        fp = str(tmp_path / r["feat_path"])
        np.save(fp, np.array(case["feats"][r["utt_id"]], dtype=np.float32))
        rows.append(dict(r, feat_path=fp))
    tsv = str(tmp_path / "train.tsv")
    pd.DataFrame(rows).to_csv(tsv, sep="\t", index=False)
    kdp = str(tmp_path / "kd.pkl")
    with open(kdp, "wb") as f:
        pickle.dump({k: [[(int(v), np.float32(p)) for v, p in pos] for pos in val] for k, val in case["kd"].items()}, f)
    prm = SimpleNamespace(feat_dim=4, num_framestacks=1, vocab_size=16, lsm_prob=0.1, eos_id=2, mtl_phone_ctc_weight=0.3,
                          phone_eos_id=1, kd_weight=0.5, kd_label_path=kdp, decoder_type=decoder_type)
    data = ds.ASRDataset(prm, tsv, phase="train")
    batch = data.collate_fn([data[i] for i in range(3)])
    want = case[decoder_type]
    assert sorted(batch) == sorted(want)
    for k, w in want.items():
        got = batch[k].tolist() if torch.is_tensor(batch[k]) else batch[k]
        if k in ("xs", "soft_labels"):
            assert np.allclose(np.array(got), np.array(w), rtol=1e-6, atol=1e-7), k
        else:
            assert got == w, k
    # test phase: neither phones nor soft labels are read
    plain = ds.ASRDataset(prm, tsv, phase="test")
    assert sorted(plain.collate_fn([plain[0]])) == ["texts", "utt_ids", "xlens", "xs", "ylens", "ys", "ys_in", "ys_out"]


class _FakeModel:
    """stands in for ASR.decode: returns the captured hypotheses, one utterance per call"""

    def __init__(self, hyps):
        self.hyps, self.calls = hyps, []

    def decode(self, xs, xlens, beam_width, len_weight, lm=None, lm_weight=0, decode_ctc_weight=0, decode_phone=False):
        i = len(self.calls)
        self.calls.append((tuple(xs.shape), int(xlens[0]), beam_width, len_weight, lm_weight, decode_ctc_weight))
        h = self.hyps[i]
        return ([h] if h else []), [None], None, None


def test_decode_driver_rows(tmp_path):
    """test_asr.py:63-121 result rows against the strings the reference's Vocab / strip_eos / ints2str give
    for the hypotheses it decoded on l2_tiny (+ an <eos>-only and an empty case)"""
    import json
    from emoasr_amd import decode as dec
    with open(os.path.join(os.path.dirname(__file__), "golden", "hostio.json")) as f:
        case = json.load(f)["result_rows"]
    vp = tmp_path / "vocab.txt"
    vp.write_text("".join(f"{t} {i}\n" for i, t in enumerate(case["vocab"])))
    vocab = ds.Vocab(str(vp))
    loader = [{"utt_ids": [f"utt{i}"], "texts": [f"ref {i}"], "xs": torch.zeros(1, 10 + i, 4), "xlens": torch.tensor([10 + i])}
              for i in range(len(case["hyps"]))]
    model = _FakeModel(case["hyps"])
    rows = dec.test(model, loader, vocab, 1, 0.0, 0.0, False, None, 0.0, "cpu")
    assert len(rows) == len(case["hyps"])
    for i, (row, (tok, text), hyp) in enumerate(zip(rows, case["rows"], case["hyps"])):
        if hyp:
            assert row == [f"utt{i}", tok, text, f"ref {i}"]
        else:
            assert row == [f"utt{i}", None, "", f"ref {i}"]  # "cannot decode"
    assert model.calls[0] == ((1, 10, 4), 10, 1, 0.0, 0.0, 0.0)
    # num_samples / sample_utt_id selection
    assert len(dec.test(_FakeModel(case["hyps"]), loader, vocab, 1, 0, 0, False, None, 0, "cpu", num_samples=2)) == 2
    only = dec.test(_FakeModel(case["hyps"][1:]), loader, vocab, 1, 0, 0, False, None, 0, "cpu", sample_utt_id="utt1")
    assert [r[0] for r in only] == ["utt1"]
    # result file: WER comment first, then the reference's four columns
    wer, info = dec.save_results(rows, str(tmp_path / "result.tsv"))
    lines = (tmp_path / "result.tsv").read_text().splitlines()
    assert lines[0] == "# " + info and lines[1].split("\t") == ["utt_id", "token_id", "text", "reftext"]
    assert info.startswith("WER: ") and len(lines) == 2 + len(rows)
    assert dec.wavtime_of("spk-utt_0012300_0015800") == 3.5 and dec.wavtime_of("utt7") is None
