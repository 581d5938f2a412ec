"""Joint CTC / attention beam search (+ Transformer-LM shallow fusion) with the WHOLE output step on the device.

Same behaviour as modeling/beam_search.py:joint_beam_search (the reference's decoders/transformer.py:161-294 with its
quirks), but per output step the host makes ONE C-ABI call (csrc/decode_rt.hip:emoasr_joint_beam_step) and reads ONE
flag back:
  * the decoder and the LM process one position per hypothesis against self-attention K / V caches, re-ordered on the
    device by parent beam (no prefix recomputation);
  * candidate selection, CTC prefix re-scoring, the per-beam and global prunes, <eos> handling and the result list are
    device kernels (emoasr_beam_update: numpy-float32 candidate scores, double hypothesis scores, stable orders);
  * the token sequences are rebuilt once, at the end, from the per-step (parent, token) history.
"""
import ctypes
import gc
import math
import os
import time

import numpy as np
import torch

from .. import lib, ops
from ..decode_rt import DecoderStepRuntime, LMStepRuntime, _lin, _ln
from ..engine import h2d_i32
from .functions import _engine_of

_STEP_DEADLINE_S = 30.0   # a search step takes ~0.5 ms; this only bounds the wait for a launch that will never report

CTC_BEAM_WIDTH_RATIO = 1.5


class _Buffers:
    """device buffers of one search configuration (beam, candidates, Lmax, model sizes), reused across utterances"""

    def __init__(self, dev, dtype, bw, cw, Lmax, V, dnl, dd, lnl, ld):
        i32 = lambda *s: torch.zeros(*s, device=dev, dtype=torch.int32)
        f32 = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.state = i32(4)                       # pos, n_alive, n_results, done
        self.ids, self.parent, self.pcand = i32(bw), i32(bw), i32(bw)
        self.last, self.outlen, self.klens = i32(bw), i32(bw), i32(bw)
        self.score = torch.zeros(bw, device=dev, dtype=torch.float64)
        self.score_ctc = f32(bw)
        self.hist_parent, self.hist_token = i32(Lmax, bw), i32(Lmax, bw)
        self.res_score = torch.zeros(bw, device=dev, dtype=torch.float64)
        self.res_step, self.res_parent = i32(bw), i32(bw)
        self.vals, self.cands, self.lm_at, self.psi = f32(bw, cw), i32(bw, cw), f32(bw, cw), f32(bw, cw)
        self.scores_pre = f32(bw, V)
        self.logits = torch.zeros(bw, V, device=dev, dtype=dtype)
        self.lm_logp = f32(bw, V)
        self.dec_k = torch.zeros(2, dnl, bw, Lmax, dd, device=dev, dtype=dtype)
        self.dec_v = torch.zeros_like(self.dec_k)
        self.lm_k = torch.zeros(2, max(lnl, 1), bw, Lmax, ld, device=dev, dtype=dtype)
        self.lm_v = torch.zeros_like(self.lm_k)
        # the flag the host polls: a pinned copy of `state`
        self.state_host = torch.zeros(4, dtype=torch.int32).pin_memory()


_yield = os.sched_yield if os.environ.get("EMOASR_BEAM_YIELD", "1") != "0" else (lambda: None)


def joint_beam_search_device(dec, eouts, elens, beam_width, len_weight=0, lm=None, lm_weight=0, decode_ctc_weight=0):
    eng = _engine_of(dec)
    assert eouts.shape[0] == 1, "beam search decodes one utterance at a time (decoders/transformer.py:181)"
    V, eos, blank = dec.vocab_size, dec.eos_id, dec.blank_id
    dev = eouts.device
    T = eouts.shape[1]
    use_ctc = decode_ctc_weight > 0
    use_lm = lm is not None and lm_weight > 0 and hasattr(lm, "predict_device")
    bw = beam_width
    cw = min(V, int(bw * CTC_BEAM_WIDTH_RATIO)) if use_ctc else bw
    assert bw <= 32 and cw <= 32, "device beam bookkeeping handles up to 32 beams x 32 candidates"
    max_steps = dec.max_decode_ylen
    Lmax = max_steps + 1
    A = eng.arena
    timing = os.environ.get("EMOASR_BEAM_TIMING") == "1"
    if os.environ.get("EMOASR_DECODE_COOP"):    # A/B switch of csrc/decode_coop.hip (one cooperative launch per network and step)
        lib.set_option("decode_coop", int(os.environ["EMOASR_DECODE_COOP"]))
    if timing:
        torch.cuda.synchronize()
        t_start = time.perf_counter()
    # the search runs on its own stream: the legacy default stream cannot be captured into a graph
    bstream = getattr(eng, "_beam_stream", None)
    if bstream is None:
        bstream = eng._beam_stream = torch.cuda.Stream(device=dev)
    caller = torch.cuda.current_stream()
    bstream.wait_stream(caller)
    with torch.no_grad(), torch.cuda.stream(bstream), ops.stream_scope():
        rt = getattr(eng, "_dec_rt", None)
        if rt is None:
            rt = eng._dec_rt = DecoderStepRuntime(eng)
        dec_layers = rt._params()
        rt.begin(eouts, bw)                     # cross-attention K / V of the memory, once per utterance
        dtype = A.shadow.dtype
        dd, dh = eng.dd, eng.dh
        lnl = ld = lH = lF = 0
        if use_lm:
            lmrt = getattr(lm, "_step_rt", None)
            if lmrt is None:
                lmrt = lm._step_rt = LMStepRuntime(lm)
            LA, lm_layers = lmrt._params()
            P = lm.params
            lnl, ld, lH, lF = P.num_layers, P.hidden_size, P.num_attention_heads, P.intermediate_size
            assert P.vocab_size == V and LA.shadow.dtype == dtype
            assert lm._pe.shape[0] >= Lmax, "LM position table shorter than max_decode_ylen + 1"
        key = (str(dev), dtype, bw, cw, Lmax, V, eng.dnl, dd, lnl, ld)
        cache = getattr(eng, "_beam_bufs", None)
        if cache is None or cache[0] != key:
            cache = eng._beam_bufs = (key, _Buffers(dev, dtype, bw, cw, Lmax, V, eng.dnl, dd, lnl, ld))
        Bf = cache[1]
        if use_ctc:
            ctc_logits = eng.head_logits(eouts, "decoder.ctc.output")
            x = ops.log_softmax(ctc_logits.view(T, V))
            init_state = ops.ctc_prefix_init(x, blank)
            states = torch.empty(2, bw, cw, T, 2, device=dev, dtype=torch.float32)
            states[1, 0, 0].copy_(init_state)    # step 0 reads "previous" = index 1 at (parent 0, candidate 0)
        # ---- initial beams: one hypothesis [<sos> = eos] ----
        Bf.state.copy_(torch.tensor([0, 1, 0, 0], dtype=torch.int32), non_blocking=True)
        Bf.ids.fill_(eos); Bf.last.fill_(eos)
        Bf.parent.zero_(); Bf.pcand.zero_(); Bf.outlen.zero_(); Bf.klens.fill_(1)
        Bf.score.zero_(); Bf.score_ctc.zero_()
        esz = A.shadow.element_size()
        F = rt._F
        ws_dec = torch.empty(lib.size_query("emoasr_decode_step_ws_bytes", ops.dt(Bf.logits), bw, dd, dh, F, V), device=dev,
                             dtype=torch.uint8)
        ws_lm = torch.empty(lib.size_query("emoasr_decode_step_ws_bytes", ops.dt(Bf.logits), bw, max(ld, 8), max(lH, 1),
                                           max(lF, 8), V), device=dev, dtype=torch.uint8) if use_lm else None
        pe_dec = eng._abs_table(Lmax, dev, dd)
        steps = []
        for cur in (0, 1):
            prev = cur ^ 1
            js = lib.JointStep()
            js.dec_nl, js.dec_layers = eng.dnl, dec_layers
            d = js.dec
            d.nb, d.Lmax, d.T, d.dd, d.H, d.F, d.V = bw, Lmax, T, dd, dh, F, V
            d.ids, d.pos, d.klens = Bf.ids.data_ptr(), Bf.state.data_ptr(), Bf.klens.data_ptr()
            d.embed, d.pe, d.emb_scale = A.w("decoder.embed.weight").data_ptr(), pe_dec.data_ptr(), math.sqrt(dd)
            d.kmem, d.kv = rt.kmem.data_ptr(), ctypes.cast(rt.kv_ptrs, ctypes.POINTER(ctypes.c_void_p))
            d.kcache, d.vcache = Bf.dec_k[cur].data_ptr(), Bf.dec_v[cur].data_ptr()
            _ln(d.ln_out, A.p("decoder.norm.weight"), A.p("decoder.norm.bias"))
            _lin(d.out, A.w("decoder.output.weight"), A.p("decoder.output.bias"))
            d.logits_last, d.ws, d.ws_bytes = Bf.logits.data_ptr(), ws_dec.data_ptr(), ws_dec.numel()
            js.dec_k_prev, js.dec_v_prev = Bf.dec_k[prev].data_ptr(), Bf.dec_v[prev].data_ptr()
            js.lm_nl = lnl if use_lm else 0
            if use_lm:
                js.lm_layers = lm_layers
                l = js.lm
                pre, cp = "lm.transformer.bert.", "lm.transformer.cls.predictions."
                l.nb, l.Lmax, l.d, l.H, l.F, l.V = bw, Lmax, ld, lH, lF, V
                l.ids, l.pos, l.klens = Bf.ids.data_ptr(), Bf.state.data_ptr(), Bf.klens.data_ptr()
                l.word_emb, l.pe = LA.w(pre + "embeddings.word_embeddings.weight").data_ptr(), lm._pe.data_ptr()
                _ln(l.ln_emb, LA.p(pre + "embeddings.LayerNorm.weight"), LA.p(pre + "embeddings.LayerNorm.bias"))
                l.kcache, l.vcache = Bf.lm_k[cur].data_ptr(), Bf.lm_v[cur].data_ptr()
                _lin(l.transform, LA.w(cp + "transform.dense.weight"), LA.p(cp + "transform.dense.bias"))
                _ln(l.ln_transform, LA.p(cp + "transform.LayerNorm.weight"), LA.p(cp + "transform.LayerNorm.bias"))
                l.out_bias, l.logp = LA.p(cp + "bias").data_ptr(), Bf.lm_logp.data_ptr()
                l.raw_logits = 1   # the scoring kernel does both log-softmaxes (emoasr_beam_scores_topk)
                l.ws, l.ws_bytes = ws_lm.data_ptr(), ws_lm.numel()
                js.lm_k_prev, js.lm_v_prev = Bf.lm_k[prev].data_ptr(), Bf.lm_v[prev].data_ptr()
            js.parent, js.scores_pre = Bf.parent.data_ptr(), Bf.scores_pre.data_ptr()
            js.T, js.blank = T, blank
            if use_ctc:
                js.ctc_x = x.data_ptr()
                js.states_prev, js.states_cur = states[prev].data_ptr(), states[cur].data_ptr()
            u = js.upd
            u.bw, u.cw, u.eos = bw, cw, eos
            lam, mu = float(decode_ctc_weight), float(lm_weight)
            u.one_minus_lam, u.lam, u.mu = float(np.float32(1 - lam)), float(np.float32(lam)), float(np.float32(mu)) if use_lm else 0.0
            u.len_weight = float(len_weight)
            u.vals, u.cands = Bf.vals.data_ptr(), Bf.cands.data_ptr()
            u.lm_at = Bf.lm_at.data_ptr() if (use_lm and use_ctc) else None
            u.psi = Bf.psi.data_ptr() if use_ctc else None
            u.score, u.score_ctc = Bf.score.data_ptr(), Bf.score_ctc.data_ptr()
            u.n_ids, u.n_parent, u.n_pcand = Bf.ids.data_ptr(), Bf.parent.data_ptr(), Bf.pcand.data_ptr()
            u.n_last, u.n_outlen, u.n_klens = Bf.last.data_ptr(), Bf.outlen.data_ptr(), Bf.klens.data_ptr()
            u.hist_parent, u.hist_token = Bf.hist_parent.data_ptr(), Bf.hist_token.data_ptr()
            u.res_score, u.res_step, u.res_parent = Bf.res_score.data_ptr(), Bf.res_step.data_ptr(), Bf.res_parent.data_ptr()
            u.state = Bf.state.data_ptr()
            u.host_mirror = Bf.state_host.data_ptr()   # pinned: the kernel writes it, the host polls it (no copy engine)
            steps.append(js)
        side = None
        if use_lm:
            side = getattr(eng, "_lm_stream", None)
            if side is None:
                side = eng._lm_stream = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream()
        dt_code = ops.dt(Bf.logits)
        if timing:
            torch.cuda.synchronize()
            t_loop = time.perf_counter()
        main_p, side_p = main.cuda_stream, (side.cuda_stream if side is not None else None)
        t_loop0 = time.perf_counter()
        # HIP graphs pay off for the launch chains (~185 kernels per step); with the cooperative step kernels a step is ~25 launches
        # and plain launches are ~3 % faster (no graph boundaries: 0.476-0.607 against 0.494-0.625 ms per step over T' 190-600)
        g_env = os.environ.get("EMOASR_BEAM_GRAPH", "auto")
        use_graph = (os.environ.get("EMOASR_DECODE_COOP", "1") == "0") if g_env == "auto" else g_env != "0"
        if use_graph:
            if not getattr(eng, "_beam_graph_warm", False):
                # first use in this process: one eager pass with the search marked finished, so that every kernel's lazy
                # one-time setup (function attributes, device queries) happens outside stream capture
                Bf.state.copy_(torch.tensor([0, 1, 0, 1], dtype=torch.int32))
                lib.call("emoasr_joint_beam_step", dt_code, ctypes.byref(steps[0]), main_p, side_p)
                main.synchronize()
                Bf.state.copy_(torch.tensor([0, 1, 0, 0], dtype=torch.int32))
                eng._beam_graph_warm = True
            for k in (0, 1):
                for part in ((0, 1, 2) if use_lm else (0, 2)):
                    lib.call("emoasr_joint_beam_graph_build", dt_code, ctypes.byref(steps[k]), k, part, main_p)
            if timing:
                print(f"[beam timing] graph build / update {1e3 * (time.perf_counter() - t_loop0):.2f} ms", flush=True)
        # Steps are issued one ahead of the flag they depend on: a step launched after the search has finished changes
        # nothing (emoasr_beam_update returns at once), and the GPU never waits for the host's round trip.
        ev_tail, ev_lm = torch.cuda.Event(), torch.cuda.Event()
        # device time of the search loop for the running totals (bench.py: search_loop_ms_per_step): events on the search stream, so
        # that the encoder pass / cross-attention K, V / CTC prefix setup still queued AHEAD of the first step are not charged to
        # the loop (host wall time from here charged ~4 ms of them per utterance: 0.94 ms per step on ten-step searches, 0.53 on
        # forced 36-step ones, for the same 0.55 ms steps)
        ev_loop0, ev_loop1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev_loop0.record(main)
        two_streams = os.environ.get("EMOASR_BEAM_TWO_STREAMS", "1") != "0"
        mirror = Bf.state_host.numpy()               # [pos, n_alive, n_results, done], written by emoasr_beam_update
        mirror[:] = (0, 1, 0, 0)
        # a generation-2 garbage collection in the middle of the loop stalls a 0.65 ms step for 10-15 ms; the loop allocates
        # next to nothing, so collection simply waits until it is over
        gc_was_on = gc.isenabled()
        gc.disable()
        dbg = os.environ.get("EMOASR_BEAM_STEP_TIMING") == "1"
        t_l = t_s = 0.0
        worst = (0.0, -1, "")
        try:
            for i in range(max_steps):
                if dbg:
                    _t0 = time.perf_counter()
                if use_graph:
                    k = i & 1
                    if use_lm and two_streams:   # the LM chain on the side stream, concurrent with the decoder chain
                        ev_tail.record(main)             # the previous step's tail (ids / parent / pos) is complete
                        side.wait_event(ev_tail)
                        lib.call("emoasr_joint_beam_graph_launch", k, 1, side_p)
                        ev_lm.record(side)
                    elif use_lm:
                        lib.call("emoasr_joint_beam_graph_launch", k, 1, main_p)
                    lib.call("emoasr_joint_beam_graph_launch", k, 0, main_p)
                    if use_lm and two_streams:
                        main.wait_event(ev_lm)
                    lib.call("emoasr_joint_beam_graph_launch", k, 2, main_p)
                else:
                    lib.call("emoasr_joint_beam_step", dt_code, ctypes.byref(steps[i & 1]), main_p, side_p)
                if dbg:
                    _t1 = time.perf_counter()
                # Steps are issued one ahead of the flag they depend on (a step launched after the search has finished changes
                # nothing: emoasr_beam_update returns at once), so the GPU never waits for the host.  The state of step i - 1
                # arrives in pinned memory by the kernel's own store; polling it involves no copy engine and no event.
                if i >= 1:
                    spins, t_wait = 0, None
                    while mirror[0] < i and not mirror[3]:
                        _yield()   # the runtime's own threads (signal handling, graph bookkeeping) may share this core
                        spins += 1
                        if spins & 0xFFF == 0:   # a lost launch / a faulted device must surface, not spin forever
                            now = time.perf_counter()
                            t_wait = now if t_wait is None else t_wait
                            if now - t_wait > _STEP_DEADLINE_S:
                                raise RuntimeError(f"beam search: step {i - 1} did not report within {_STEP_DEADLINE_S:.0f} s "
                                                   f"(stream idle: {main.query()}; last library error: {lib.load().emoasr_last_error().decode()!r})")
                    if mirror[3]:
                        break
                if dbg:
                    _t2 = time.perf_counter()
                    t_l += _t1 - _t0; t_s += _t2 - _t1
                    if _t1 - _t0 > worst[0]:
                        worst = (_t1 - _t0, i, "launch")
                    if _t2 - _t1 > worst[0]:
                        worst = (_t2 - _t1, i, "sync")
            ev_loop1.record(main)
            main.synchronize()
            if lib.size_query("emoasr_decode_coop_status") > 0:
                raise RuntimeError("decode_coop: a grid barrier gave up waiting (csrc/decode_coop.hip); emoasr_set_option('decode_coop', 0) "
                                   "selects the launch chain")
        finally:
            if gc_was_on:   # (also on an exception: the collector must not stay off for the rest of the process)
                gc.enable()
        if dbg:
            print(f"   host: launches {1e3 * t_l:.2f} ms, waits {1e3 * t_s:.2f} ms, worst {1e3 * worst[0]:.2f} ms at step {worst[1]} ({worst[2]})", flush=True)
        n_done = int(mirror[0])                      # effective steps (pos advances only while the search is live)
        stats = getattr(eng, "_beam_stats", None)    # running totals for bench.py: steps and wall time of the search loops
        if stats is None:
            stats = eng._beam_stats = {"steps": 0, "loop_s": 0.0, "utts": 0}
        # (the loop issues one step beyond the last effective one: its device time is part of the figure)
        stats["steps"] += n_done; stats["loop_s"] += 1e-3 * ev_loop0.elapsed_time(ev_loop1); stats["utts"] += 1
        if side is not None:
            main.wait_stream(side)
        if timing:
            torch.cuda.synchronize()
            t_end = time.perf_counter()
            print(f"[beam timing] T' {T}: setup {1e3 * (t_loop - t_start):.2f} ms, {n_done} steps {1e3 * (t_end - t_loop):.2f} ms "
                  f"({1e3 * (t_end - t_loop) / max(n_done, 1):.3f} ms/step)", flush=True)
        # ---- results: token sequences from the (parent, token) history ----
        n_res = int(mirror[2])
        hp, ht = Bf.hist_parent[:n_done].cpu().numpy(), Bf.hist_token[:n_done].cpu().numpy()
        rs, rstep, rpar = Bf.res_score[:n_res].cpu().numpy(), Bf.res_step[:n_res].cpu().numpy(), Bf.res_parent[:n_res].cpu().numpy()
    caller.wait_stream(bstream)
    results = []
    for k in range(n_res):
        toks, slot = [], int(rpar[k])
        for p in range(int(rstep[k]) - 1, -1, -1):
            toks.append(int(ht[p, slot]))
            slot = int(hp[p, slot])
        results.append(dict(hyp=toks[::-1], score=float(rs[k])))
    results = sorted(results, key=lambda r: r["score"], reverse=True)
    return [r["hyp"] for r in results], [r["score"] for r in results], None, None
