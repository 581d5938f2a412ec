"""Oracle part 4 (CTC prefix beam search with LM shallow fusion) pinned to the reference's outputs
(tests/golden/ctcbeam_tiny.npz, made by tests/golden/make_golden.py ctcbeam)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import ctc_beam as ob
from oracle import decoder as od
from oracle import model as om
from tests.util import CTC_BEAM_SETTINGS, LM_CFG, load_ctc_beam_golden, split_ragged


@pytest.mark.parametrize("si", range(len(CTC_BEAM_SETTINGS)))
def test_ctc_beam_search_matches_reference(si):
    cfg, sd, lmsd, g2, gb = load_ctc_beam_golden()
    st = CTC_BEAM_SETTINGS[si]
    lmcfg = SimpleNamespace(**LM_CFG)

    def lm_predict(batch, lens):
        with torch.no_grad():
            return od.lm_predict(lmsd, lmcfg, torch.from_numpy(batch), torch.from_numpy(lens)).numpy()

    for b in (1, 2, 3):
        n = int(g2["xlens"][b])
        with torch.no_grad():
            eouts, elens = om.encoder_forward(sd, cfg, g2["xs"][b:b + 1, :n], g2["xlens"][b:b + 1])
            logits = om.ctc_decoder_forward(sd, cfg, eouts, elens)
        if si == 0:
            ref_logits = gb[f"logits/{b}"]
            assert ((logits - ref_logits).abs().max() / ref_logits.abs().max()).item() < 1e-4
        lp = torch.log_softmax(logits[0].double(), -1).numpy()
        hyps, scores = ob.ctc_prefix_beam_search(lp, cfg.blank_id, cfg.eos_id, st["beam_width"], st["len_weight"],
                                                 lm_predict, st["lm_weight"])
        ref_hyps = split_ragged(gb[f"decode/{si}/{b}/hyps"], gb[f"decode/{si}/{b}/lens"])
        assert hyps == ref_hyps, (si, b)
        np.testing.assert_allclose(scores, gb[f"decode/{si}/{b}/scores"].numpy(), rtol=2e-4, atol=2e-4)
