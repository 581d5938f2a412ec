"""RNNTDecoder -- module API of asr/modeling/decoders/rnn_transducer.py:24-346 on the HIP engine.

    decoder(eouts, elens, eouts_inter, ys, ylens, ys_in, ys_out) -> (loss, loss_dict, logits [B,T,L+1,V])
    decoder.decode(eouts, elens, beam_width=1, ...)              -> (hyps, None, None, None)

The transducer loss is the HIP lattice kernel (`emoasr_rnnt_forward/_grad`) -- the reference calls the
third-party warp_rnnt package here (rnn_transducer.py:106-115).  beam_width <= 1: time-synchronous greedy
search; > 1: alignment-length synchronous beam search (:242-325, batch size 1, hypotheses keep the leading
<sos>, LM arguments unused -- all as in the reference).  Like the reference, decode() discards
scores/logits/aligns (:339-346).
"""
import torch.nn as nn

from ..functions import rnnt_apply, rnnt_beam_apply, rnnt_greedy_apply, rnnt_prediction_stacked
from ...criteria import RNNTAlignDistillLoss, RNNTWordDistillLoss
from .ctc import CTCDecoder
from .rnnt_aligner import RNNTForcedAligner


class RNNTDecoder(nn.Module):
    def __init__(self, params, phase="train"):
        super().__init__()
        self.dec_num_layers = params.dec_num_layers
        self.dec_hidden_size = params.dec_hidden_size
        self.eos_id = params.eos_id
        self.blank_id = params.blank_id
        self.max_seq_len = 256
        self.mtl_ctc_weight = params.mtl_ctc_weight
        self.kd_weight = params.kd_weight
        self.embed = nn.Embedding(params.vocab_size, params.embedding_size)
        self.rnns = nn.ModuleList()
        nin = params.embedding_size
        for _ in range(self.dec_num_layers):
            self.rnns.append(nn.LSTM(input_size=nin, hidden_size=params.dec_hidden_size, num_layers=1, batch_first=True))
            nin = params.dec_hidden_size
        self.w_enc = nn.Linear(params.enc_hidden_size, params.joint_hidden_size)
        self.w_dec = nn.Linear(params.dec_hidden_size, params.joint_hidden_size)
        self.output = nn.Linear(params.joint_hidden_size, params.vocab_size)
        if self.mtl_ctc_weight > 0:
            self.ctc = CTCDecoder(params, prefix="decoder.ctc")
        if params.kd_weight > 0 and phase == "train":
            self.kd_type = params.kd_type
            self.reduce_main_loss_kd = params.reduce_main_loss_kd
            if self.kd_type == "word":
                self.transducer_kd_loss = RNNTWordDistillLoss()
            elif self.kd_type == "align":
                self.transducer_kd_loss = RNNTAlignDistillLoss()
                self.forced_aligner = RNNTForcedAligner(blank_id=self.blank_id)
            else:
                raise NotImplementedError(f"emoasr_amd: unknown kd_type {self.kd_type!r}")
        # the reference returns the joint logits [B,T,U,V] as the third value of forward(); nothing in its drivers reads them in
        # training, and forming them costs 0.9 GB per micro-batch at the L4 sizes: in train mode they are None unless this is set
        # (or distillation needs them); eval-mode forward returns them as the reference does
        self.return_logits = False
        self._owner = None

    def prediction_stacked(self, ys_in_list, ylens_list):
        """the prediction network (embedding + LSTM stack: it reads the labels only) of several micro-batches in ONE pass ->
        a list of per-micro-batch outputs to hand to forward(..., pred=...), or None when the stacked pass does not apply
        (not in the reference: its train loop runs the micro-batches one by one, train_asr.py:106-128)"""
        return rnnt_prediction_stacked(self, ys_in_list, ylens_list)

    def forward(self, eouts, elens, eouts_inter=None, ys=None, ylens=None, ys_in=None, ys_out=None,
                soft_labels=None, ps=None, plens=None, pred=None):
        kd = None
        if self.kd_weight > 0 and soft_labels is not None:
            kd = (soft_labels, self.kd_weight, self.reduce_main_loss_kd, self.kd_type)
        loss, loss_rnnt, loss_ctc, logits, loss_kd = rnnt_apply(self, eouts, elens, ys, ylens, ys_in, kd, pred)
        loss_dict = {"loss_rnnt": loss_rnnt}
        if self.mtl_ctc_weight > 0:
            loss_dict["loss_ctc"] = loss_ctc
        if kd is not None:
            loss_dict["loss_kd"] = loss_kd
        loss_dict["loss_total"] = loss
        return loss, loss_dict, logits

    def decode(self, eouts, elens, eouts_inter=None, beam_width=1, len_weight=0, lm=None, lm_weight=0,
               decode_ctc_weight=0, decode_phone=False):
        if beam_width > 1:
            return self._beam_search(eouts, elens, beam_width, len_weight, lm, lm_weight), None, None, None
        if decode_ctc_weight == 1:
            self.ctc._owner = self._owner
            return self.ctc.decode(eouts, elens, beam_width=1)
        hyps, aligns = rnnt_greedy_apply(self, eouts, elens)
        return hyps, None, None, None

    def _greedy(self, eouts, elens, decode_ctc_weight=0):
        hyps, aligns = rnnt_greedy_apply(self, eouts, elens)
        return hyps, [None] * len(hyps), None, aligns

    def _beam_search(self, eouts, elens, beam_width=1, len_weight=0, lm=None, lm_weight=0):
        assert eouts.size(0) == 1  # rnn_transducer.py:251
        return rnnt_beam_apply(self, eouts, beam_width)
