"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/emoasr_hip.h declares (no compute calls without a GPU), and the module API mirrors the
reference's state_dict layout."""
import os
import re
from types import SimpleNamespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "emoasr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(emoasr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from emoasr_amd import lib
    handle = lib.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(handle, n), f"{n} declared in emoasr_hip.h but not exported"
    for n in lib.SIGNATURES:
        assert n in names, f"{n} bound in lib.py but not declared in the header"
    assert handle.emoasr_version() >= 1


def test_missing_option_is_an_error():
    from emoasr_amd import lib
    with pytest.raises(lib.EmoasrHipError):
        lib.set_option("no_such_option", 1)


@pytest.mark.parametrize("name", ["l2_tiny", "l1_tiny", "l3_tiny", "l4_tiny"])
def test_state_dict_layout_matches_reference(name):
    from emoasr_amd.modeling.asr import ASR
    from tests.util import CONFIGS, load_golden
    cfg, sd, g = load_golden(name)
    model = ASR(SimpleNamespace(**CONFIGS[name]))
    mine = model.state_dict()
    assert set(mine) == set(sd)
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k


def test_no_cpu_fallback():
    from emoasr_amd.modeling.asr import ASR
    from tests.util import CONFIGS
    model = ASR(SimpleNamespace(**CONFIGS["l1_tiny"]))
    xs = torch.zeros(1, 64, 40)
    with pytest.raises((AssertionError, RuntimeError)):
        model(xs, [64], torch.zeros(1, 3, dtype=torch.long), [3], None, None)
    with pytest.raises(RuntimeError):
        model.encoder.transformers[0](xs)


def test_unbuilt_variants_fail_loudly():
    from emoasr_amd.modeling.asr import ASR
    from tests.util import CONFIGS
    with pytest.raises(NotImplementedError):
        ASR(SimpleNamespace(**dict(CONFIGS["l2_tiny"], encoder_type="rnn")))
    with pytest.raises(NotImplementedError):
        ASR(SimpleNamespace(**dict(CONFIGS["l4_tiny"], kd_weight=0.5, kd_type="sequence", reduce_main_loss_kd=False)))
