"""ASR facade -- drop-in for asr/modeling/asr.py:21-95 with the compute on MI355X HIP kernels.

    model = ASR(params, phase="train").cuda()
    loss, loss_dict = model(xs, xlens, ys, ylens, ys_in, ys_out)       # loss.backward() works
    hyps, scores, logits, aligns = model.decode(xs, xlens, beam_width=1)

`compute_dtype`: torch.bfloat16 (throughput mode; f32 accumulation, statistics and lattices),
torch.float32 (exact mode: f32 MFMA) or "f32x3" (f32 storage, every matrix product as three bf16
MFMAs over (hi, lo) operand pairs: 16 significand bits per operand -- meets the 1e-3 bars of the reference
comparison at 1.6-1.7 x the exact mode's training rate, 0.37 of the bf16 rate; csrc/gemm.hip SplitCfg).  The constructor
logs ONE line naming the mode and its measured distance from the reference arithmetic (MODE_NOTES): the default, bf16, is
BASELINE.json's benchmark mode and is OUTSIDE north_star's 1e-3 / bit-exact-ids tolerance -- a caller who swaps the import
for parity passes compute_dtype="f32x3".
"""
import logging

import torch
import torch.nn as nn

from .decoders.ctc import CTCDecoder
from .encoders.transformer import TransformerEncoder


F32X3 = "f32x3"   # compute_dtype of the split mode (see the module docstring)

# what a constructor call is told about its mode: measured at the full L2 size against oracle/model.py
# (tests/test_fullsize_gpu.py::test_full_model_against_oracle, test_bench_sized_batch_against_oracle)
MODE_NOTES = {
    "bf16": "compute_dtype=bf16 (the benchmark's mode): logits within 1.3e-2 of their range of the f32 reference, greedy CTC frames agree "
            "0.96-0.98 on random-init weights (identical hypotheses on fitted weights) -- OUTSIDE the 1e-3 / bit-exact-ids tolerance; "
            "pass compute_dtype='f32x3' for parity",
    "f32x3": "compute_dtype='f32x3': f32 storage, split-bf16 products -- loss 2e-7, logits 9e-6 of range, greedy ids bit-exact against "
             "the f32 reference (inside the 1e-3 tolerance) at 0.37 of the bf16 training rate",
    "f32": "compute_dtype=float32: exact f32 MFMA chains -- loss / logits 1e-6 of the f32 reference, greedy ids bit-exact, at 0.22 of "
           "the bf16 training rate",
}


class ASR(nn.Module):
    def __init__(self, params, phase="train", compute_dtype=torch.bfloat16):
        super().__init__()
        self.encoder_type = params.encoder_type
        self.decoder_type = params.decoder_type
        self.params = params
        self.compute_dtype = compute_dtype   # (property: "f32x3" -> torch.float32 + f32_split)
        if self.encoder_type not in ("transformer", "conformer"):
            raise NotImplementedError(f"emoasr_amd: encoder_type={self.encoder_type!r} is outside the HIP hot path")
        self.encoder = TransformerEncoder(params, is_conformer=(self.encoder_type == "conformer"))
        if self.decoder_type == "ctc":
            self.decoder = CTCDecoder(params)
        elif self.decoder_type == "transformer":
            from .decoders.transformer import TransformerDecoder
            self.decoder = TransformerDecoder(params)
            if hasattr(self.decoder, "ctc"):
                self.decoder.ctc._owner = [self]
        elif self.decoder_type == "rnn_transducer":
            from .decoders.rnn_transducer import RNNTDecoder
            self.decoder = RNNTDecoder(params, phase)
            if hasattr(self.decoder, "ctc"):
                self.decoder.ctc._owner = [self]
        else:
            raise NotImplementedError(f"emoasr_amd: decoder_type={self.decoder_type!r} is not built yet")
        self.encoder._owner = [self]  # list: keeps the back-reference out of nn.Module registration
        self.decoder._owner = [self]
        self._engine = None
        n = sum(p.numel() for p in self.parameters())
        logging.info(f"ASR model #parameters: {n}")
        logging.info("emoasr_amd: " + MODE_NOTES["f32x3" if self.f32_split else ("bf16" if self.compute_dtype == torch.bfloat16 else "f32")])

    @property
    def compute_dtype(self):
        return self._compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, value):
        # "f32x3": f32 storage and statistics, every matrix product as three bf16 MFMAs over (hi, lo) operand pairs; the engine is
        # rebuilt at its next use when the mode changed (engine())
        split = isinstance(value, str) and value == F32X3
        assert split or value in (torch.float32, torch.bfloat16), f"compute_dtype={value!r}: torch.bfloat16, torch.float32 or 'f32x3'"
        object.__setattr__(self, "f32_split", split)
        object.__setattr__(self, "_compute_dtype", torch.float32 if split else value)

    def engine(self):
        from ..engine import CTCEngine
        if self._engine is None or self._engine.dtype != self.compute_dtype or self._engine.split != self.f32_split:
            self._engine = CTCEngine(self.params, self, self.compute_dtype, f32_split=self.f32_split)
        self._engine._apply_mode()
        return self._engine

    def forward(self, xs, xlens, ys, ylens, ys_in, ys_out, soft_labels=None, ps=None, plens=None):
        xs = xs[:, : int(max(xlens))]
        ys = ys[:, : int(max(ylens))]
        if ys_in is not None:
            ys_in = ys_in[:, : int(max(ylens)) + 1]
            ys_out = ys_out[:, : int(max(ylens)) + 1]
        if ps is not None:  # phone targets are trimmed to the batch like the word targets (asr.py:61-62)
            ps = ps[:, : int(max(plens))]
        eouts, elens, eouts_inter = self.encoder(xs, xlens)
        loss, loss_dict, _ = self.decoder(eouts, elens, eouts_inter, ys, ylens, ys_in, ys_out, soft_labels, ps, plens)
        return loss, loss_dict

    def decode(self, xs, xlens, beam_width=1, len_weight=0, lm=None, lm_weight=0, decode_ctc_weight=0,
               decode_phone=False):
        with torch.no_grad():
            eouts, elens, eouts_inter = self.encoder(xs, xlens)
            return self.decoder.decode(eouts, elens, eouts_inter, beam_width, len_weight, lm, lm_weight,
                                       decode_ctc_weight, decode_phone)
