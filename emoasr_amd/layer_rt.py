"""Python side of the C++ layer runtime (csrc/layer.hip: emoasr_conformer_layer_fwd).

One FFI call per Conformer layer instead of ~24: per layer this module keeps a cached parameter struct
(pointers into the flat arenas, stable across steps) and, per (B, T) shape, the layout of two
workspaces (compute dtype / f32) that receive every intermediate.  The backward needs those
intermediates as tensors; the views are created lazily, when the backward sweep reaches the layer.
"""
import ctypes
import os

import torch

from . import lib, ops

_ALIGN_T, _ALIGN_F = 64, 16  # elements: 128-byte (bf16) / 64-byte (f32) aligned slots


def _round(n, a):
    return (n + a - 1) // a * a


class _Layout:
    """offsets (in elements) of every forward intermediate of one layer for a (B, T) shape, or for a tuple of stacked
    micro-batches `segs` = ((B_0, T_0), (B_1, T_1), ...) whose rows are concatenated (emoasr_segments_t)"""

    def __init__(self, B, T, d, H, F, segs=None):
        self.segs = segs
        if segs is None:
            M, R = B * T, 2 * T - 1
            qkv_shape, o_shape, lse_shape, nstat = (B, T, 3 * d), (B, T, d), (B, H, T), 1
            npart = lib.size_query("emoasr_dwconv_stats_floats", B, T, d)
        else:
            M, R = sum(b * t for b, t in segs), sum(2 * t - 1 for _, t in segs)
            qkv_shape, o_shape, lse_shape, nstat = (M, 3 * d), (M, d), (H * M,), len(segs)
            npart = sum(lib.size_query("emoasr_dwconv_stats_floats", b, t, d) for b, t in segs)
        self.M, self.R = M, R
        # attention keep mask as bits (uint32 [M, H, nw], emoasr_attn_t::keep_mask): hashed once per layer in the forward
        self.mask_nw = lib.size_query("emoasr_attn_dropmask_words", T if segs is None else max(t for _, t in segs))
        t_fields = [("ffm_h", (M, d)), ("ffm_u", (M, F)), ("ffm_a", (M, F)), ("ffm_y", (M, d)),
                    ("at_h", (M, d)), ("qkv", qkv_shape), ("pp", (R, d)), ("o", o_shape), ("at_y", (M, d)),
                    ("cv_h", (M, d)), ("g", (M, 2 * d)), ("gl", (M, d)), ("c", (M, d)), ("z", (M, d)), ("cv_y", (M, d)),
                    ("ff_h", (M, d)), ("ff_u", (M, F)), ("ff_a", (M, F)), ("ff_y", (M, d)), ("y", (M, d))]
        f_fields = [("ffm_mean", (M,)), ("ffm_rstd", (M,)), ("lse", lse_shape), ("at_mean", (M,)), ("at_rstd", (M,)),
                    ("bmean", (nstat * d,)), ("bvar", (nstat * d,)), ("bn_part", (npart,)),
                    ("cv_mean", (M,)), ("cv_rstd", (M,)), ("ff_mean", (M,)), ("ff_rstd", (M,)),
                    ("fin_mean", (M,)), ("fin_rstd", (M,)), ("att_mask", (M * H * self.mask_nw,))]
        self.t, self.f = {}, {}
        off = 0
        for name, shape in t_fields:
            n = 1
            for s in shape:
                n *= s
            self.t[name] = (off, n, shape)
            off += _round(n, _ALIGN_T)
        self.nt = off
        off = 0
        for name, shape in f_fields:
            n = 1
            for s in shape:
                n *= s
            self.f[name] = (off, n, shape)
            off += _round(n, _ALIGN_F)
        self.nf = off


class LayerStash:
    """What one layer's forward left behind; iterating yields the (s_ffm, s_att, s_conv, s_ff, s_fin) tuples
    engine._backward expects, with tensor views built on first use."""
    __slots__ = ("wt", "wf", "lay", "x_in", "seeds", "_tup", "io", "B", "T", "sweep")

    def __init__(self, wt, wf, lay, x_in, seeds):
        self.wt, self.wf, self.lay, self.x_in, self.seeds, self._tup = wt, wf, lay, x_in, seeds, None
        self.io = None  # the emoasr_conformer_fwd_t of the forward call (the C++ backward takes it as the stash)

    def tv(self, name):
        o, n, shape = self.lay.t[name]
        return self.wt[o:o + n].view(shape)

    def fv(self, name):
        o, n, shape = self.lay.f[name]
        return self.wf[o:o + n].view(shape)

    def input(self):
        """the layer input as a tensor (a tensor, or the previous layer's stash whose output it is)"""
        x = self.x_in
        if isinstance(x, LayerStash):
            x = self.x_in = x.tv("y")
        return x

    def __iter__(self):
        if self.lay.segs is not None:
            raise RuntimeError("a stacked layer pass is differentiated by emoasr_conformer_layer_bwd only (no per-kernel path)")
        if self._tup is None:
            tv, fv, s = self.tv, self.fv, self.seeds
            x0 = self.input()
            x1, x2, x3, x4 = tv("ffm_y"), tv("at_y"), tv("cv_y"), tv("ff_y")
            s_ffm = (x0, fv("ffm_mean"), fv("ffm_rstd"), tv("ffm_h"), tv("ffm_u"), tv("ffm_a"), s[0], s[1])
            s_att = (x1, fv("at_mean"), fv("at_rstd"), tv("at_h"), tv("qkv"), tv("pp"), tv("o"), fv("lse"), s[2], s[3], None)
            s_conv = (x2, fv("cv_mean"), fv("cv_rstd"), tv("cv_h"), tv("g"), tv("gl"), tv("c"), fv("bmean"), fv("bvar"),
                      tv("z"), s[4])
            s_ff = (x3, fv("ff_mean"), fv("ff_rstd"), tv("ff_h"), tv("ff_u"), tv("ff_a"), s[5], s[6])
            self._tup = (s_ffm, s_att, s_conv, s_ff, (x4, fv("fin_mean"), fv("fin_rstd")))
        return iter(self._tup)

    def __getitem__(self, i):
        return tuple(self)[i]


class ConformerLayerRuntime:
    def __init__(self, eng):
        self.eng = eng
        self.params = {}    # layer index -> (lib.ConformerLayer, guard)
        self.grads = {}     # layer index -> (lib.ConformerLayer of gradient pointers, guard)
        self.layouts = {}   # (B, T) -> _Layout
        # backward workspaces: one shared by all layers (they run one after the other) -- two, taken in turn, when the layers' weight-
        # gradient launches run on the side stream (option "wgrad_side": a launch reads its layer's workspace under the next layer)
        self.bwd_ws = [None, None]
        self.wgrad_side = os.environ.get("EMOASR_CPP_WGRAD_SIDE", "0") != "0"
        lib.set_option("wgrad_side", int(self.wgrad_side))
        self.calls = 0
        self.attn_img, self.attn_img_key = None, None  # f32: zero-filled attention images of the current backward sweep
        self.sweeps = 0

    def _layer_params(self, li):
        eng, A = self.eng, self.eng.arena
        name = f"encoder.transformers.{li}"
        bn = name + ".conv.batch_norm"
        rm = eng._buffers(bn + ".running_mean")
        guard = (A.flat.data_ptr(), A.shadow.data_ptr(), rm.data_ptr())
        hit = self.params.get(li)
        if hit is not None and hit[1] == guard:
            return hit[0]
        d = eng.d
        L = lib.ConformerLayer()
        L.d, L.H = d, eng.h
        w1 = A.w(name + ".feed_forward.w1.weight")
        L.F = w1.shape[0]
        wd = A.p(name + ".conv.depthwise_conv.weight")
        L.K = wd.shape[-1]

        def ffn(dst, pre, norm):
            dst.ln_g, dst.ln_b = A.p(norm + ".weight").data_ptr(), A.p(norm + ".bias").data_ptr()
            dst.w1, dst.b1 = A.w(pre + ".w1.weight").data_ptr(), A.p(pre + ".w1.bias").data_ptr()
            dst.w2, dst.b2 = A.w(pre + ".w2.weight").data_ptr(), A.p(pre + ".w2.bias").data_ptr()

        ffn(L.ffm, name + ".feed_forward_macaron", name + ".norm_ff_macaron")
        ffn(L.ff, name + ".feed_forward", name + ".norm_ff")
        sa = name + ".self_attn"
        L.att_ln_g, L.att_ln_b = A.p(name + ".norm_self_attn.weight").data_ptr(), A.p(name + ".norm_self_attn.bias").data_ptr()
        L.wqkv = A.w_span(sa + ".linear_q.weight", sa + ".linear_v.weight", (3 * d, d)).data_ptr()
        L.bqkv = A.p_span(sa + ".linear_q.bias", sa + ".linear_v.bias", (3 * d,)).data_ptr()
        L.wpos = A.w(sa + ".linear_pos.weight").data_ptr()
        L.bias_u, L.bias_v = A.p(sa + ".pos_bias_u").data_ptr(), A.p(sa + ".pos_bias_v").data_ptr()
        L.wout, L.bout = A.w(sa + ".linear_out.weight").data_ptr(), A.p(sa + ".linear_out.bias").data_ptr()
        cv = name + ".conv"
        L.cv_ln_g, L.cv_ln_b = A.p(name + ".norm_conv.weight").data_ptr(), A.p(name + ".norm_conv.bias").data_ptr()
        L.pw1, L.pw1_b = A.w(cv + ".pointwise_conv1.weight", (2 * d, d)).data_ptr(), A.p(cv + ".pointwise_conv1.bias").data_ptr()
        L.dw_w, L.dw_b = wd.data_ptr(), A.p(cv + ".depthwise_conv.bias").data_ptr()
        L.bn_g, L.bn_b = A.p(bn + ".weight").data_ptr(), A.p(bn + ".bias").data_ptr()
        L.bn_rm, L.bn_rv = rm.data_ptr(), eng._buffers(bn + ".running_var").data_ptr()
        L.bn_nbt = eng._buffers(bn + ".num_batches_tracked").data_ptr()
        L.pw2, L.pw2_b = A.w(cv + ".pointwise_conv2.weight", (d, d)).data_ptr(), A.p(cv + ".pointwise_conv2.bias").data_ptr()
        L.fin_ln_g, L.fin_ln_b = A.p(name + ".norm_final.weight").data_ptr(), A.p(name + ".norm_final.bias").data_ptr()
        if A.shadow.dtype == torch.bfloat16 and os.environ.get("EMOASR_DGRAD_NT", "1") != "0":
            # transposed copies of the weights whose data gradients are long reductions (csrc/layer.hip uses them when present)
            L.ffm_w1t = A.transposed(name + ".feed_forward_macaron.w1.weight").data_ptr()
            L.ff_w1t = A.transposed(name + ".feed_forward.w1.weight").data_ptr()
            L.wqkv_t = A.transposed(sa + ".linear_q.weight", sa + ".linear_v.weight", (3 * d, d)).data_ptr()
            L.pw1_t = A.transposed(cv + ".pointwise_conv1.weight", None, (2 * d, d)).data_ptr()
            # the K = d products too (A/B switch; round 5, three same-box pairs inside the step: 28.19 / 28.23 / 28.55 ms with,
            # 28.19 / 28.12 / 28.26 without -- the NN kernel's transposing LDS reads are not what those products wait for: off)
            if os.environ.get("EMOASR_DGRAD_NT2", "0") != "0":
                L.ffm_w2t = A.transposed(name + ".feed_forward_macaron.w2.weight").data_ptr()
                L.ff_w2t = A.transposed(name + ".feed_forward.w2.weight").data_ptr()
                L.wout_t = A.transposed(sa + ".linear_out.weight").data_ptr()
                L.pw2_t = A.transposed(cv + ".pointwise_conv2.weight", None, (d, d)).data_ptr()
        self.params[li] = (L, guard)
        return L

    def hash_attn_masks(self, nl, B, T, elens, p_att, segs):
        """the attention keep masks of all nl layers of one stacked training pass, hashed NOW on the attention's side stream (call
        before the convolution front-end: the hashing runs under its MFMA-bound products) -> int32 [nl, M * H * nw]; forward() is
        then given layer li's slice (att_mask=...)"""
        eng = self.eng
        lay = self._layout(B, T, self._layer_params(0).F, segs)
        masks = torch.empty(nl, lay.M * eng.h * lay.mask_nw, device=elens.device, dtype=torch.int32)
        seeds = (ctypes.c_uint64 * nl)(*[eng._seed(100 + li * 20 + 2) for li in range(nl)])
        seg = lib.Segments()
        seg.n = len(segs)
        for k, (sb, stt) in enumerate(segs):
            seg.B[k], seg.T[k] = sb, stt
        lib.call("emoasr_conformer_attn_masks", lib.BF16, nl, ctypes.byref(seg), B, T, eng.h, eng.d, elens.data_ptr(), p_att, seeds,
                 masks.data_ptr(), masks.stride(0), lay.mask_nw, ops._stream())
        return masks

    def _layout(self, B, T, F, segs):
        key = (B, T) if segs is None else tuple(segs)
        lay = self.layouts.get(key)
        if lay is None:
            if len(self.layouts) > 64:
                self.layouts.clear()
            lay = self.layouts[key] = _Layout(B, T, self.eng.d, self.eng.h, F, segs)
        return lay

    def forward(self, li, x, B, T, elens, pos_t, p_enc, p_att, training, keep, segs=None, att_mask=None):
        """x: tensor [B*T, d] or the previous layer's LayerStash.  -> LayerStash (its tv("y") is the output).
        segs: ((B_0, T_0), ...) -- stacked micro-batches (x has sum B_s T_s rows, pos_t the segments' tables back to back,
        elens all utterances in order); B / T are then the totals / maximum."""
        eng = self.eng
        L = self._layer_params(li)
        lay = self._layout(B, T, L.F, segs)
        dev = pos_t.device
        wt = torch.empty(lay.nt, device=dev, dtype=eng.dtype)
        wf = torch.empty(lay.nf, device=dev, dtype=torch.float32)
        site = 100 + li * 20
        seeds = tuple(eng._seed(site + k) for k in (0, 1, 2, 3, 4, 6, 7))
        st = LayerStash(wt, wf, lay, x, seeds)
        # serial number of the forward sweep this layer belongs to (a sweep starts at the layer that is given a tensor): all its
        # layers share the utterance lengths, hence the masks of the attention backward's images (backward: attn_img)
        if isinstance(x, LayerStash):
            st.sweep = x.sweep
        else:
            self.sweeps += 1
            st.sweep = self.sweeps
        esz = wt.element_size()
        tb, fb = wt.data_ptr(), wf.data_ptr()
        t, f = lay.t, lay.f
        io = lib.ConformerFwd()
        io.B, io.T = B, T
        if segs is not None:
            assert len(segs) <= lib.MAX_SEGMENTS, f"at most {lib.MAX_SEGMENTS} stacked micro-batches"
            io.seg.n = len(segs)
            for k, (sb, stt) in enumerate(segs):
                io.seg.B[k], io.seg.T[k] = sb, stt
        io.x = tb_prev(x, esz)
        io.pos_t, io.klens = pos_t.data_ptr(), elens.data_ptr()
        io.training, io.p_enc, io.p_att = int(training), p_enc, p_att
        io.seed[:] = seeds
        io.ffm.h, io.ffm.u, io.ffm.a, io.ffm.y = (tb + t["ffm_h"][0] * esz, tb + t["ffm_u"][0] * esz if keep else None,
                                                  tb + t["ffm_a"][0] * esz, tb + t["ffm_y"][0] * esz)
        io.ff.h, io.ff.u, io.ff.a, io.ff.y = (tb + t["ff_h"][0] * esz, tb + t["ff_u"][0] * esz if keep else None,
                                              tb + t["ff_a"][0] * esz, tb + t["ff_y"][0] * esz)
        for k in ("at_h", "qkv", "pp", "o", "at_y", "cv_h", "g", "gl", "c", "z", "cv_y", "y"):
            setattr(io, k, tb + t[k][0] * esz)
        io.lse = fb + f["lse"][0] * 4
        if training and p_att > 0 and wt.dtype == torch.bfloat16:
            if att_mask is not None:   # hashed up front for all layers (hash_attn_masks)
                io.att_mask, io.att_mask_nw, io.att_mask_ready = att_mask.data_ptr(), lay.mask_nw, 1
            else:
                io.att_mask, io.att_mask_nw = fb + f["att_mask"][0] * 4, lay.mask_nw
        io.bmean, io.bvar, io.bn_part = fb + f["bmean"][0] * 4, fb + f["bvar"][0] * 4, fb + f["bn_part"][0] * 4
        if keep:
            io.ffm.mean, io.ffm.rstd = fb + f["ffm_mean"][0] * 4, fb + f["ffm_rstd"][0] * 4
            io.ff.mean, io.ff.rstd = fb + f["ff_mean"][0] * 4, fb + f["ff_rstd"][0] * 4
            io.at_mean, io.at_rstd = fb + f["at_mean"][0] * 4, fb + f["at_rstd"][0] * 4
            io.cv_mean, io.cv_rstd = fb + f["cv_mean"][0] * 4, fb + f["cv_rstd"][0] * 4
            io.fin_mean, io.fin_rstd = fb + f["fin_mean"][0] * 4, fb + f["fin_rstd"][0] * 4
        lib.call("emoasr_conformer_layer_fwd", ops.dt(wt), ctypes.byref(L), ctypes.byref(io), ops._stream())
        st.io, st.B, st.T = io, B, T
        return st

    # ---------------------------------------------------------------------------------------------- backward
    def _layer_grads(self, li):
        """the layer struct again, every parameter pointer replaced by its f32 gradient's address"""
        eng, A = self.eng, self.eng.arena
        guard = A.grad.data_ptr()
        hit = self.grads.get(li)
        if hit is not None and hit[1] == guard:
            return hit[0]
        name = f"encoder.transformers.{li}"
        d = eng.d
        G = lib.ConformerLayer()

        def gp(n):
            return A.g(n).data_ptr()

        def ffn(dst, pre, norm):
            dst.ln_g, dst.ln_b = gp(norm + ".weight"), gp(norm + ".bias")
            dst.w1, dst.b1, dst.w2, dst.b2 = gp(pre + ".w1.weight"), gp(pre + ".w1.bias"), gp(pre + ".w2.weight"), gp(pre + ".w2.bias")

        ffn(G.ffm, name + ".feed_forward_macaron", name + ".norm_ff_macaron")
        ffn(G.ff, name + ".feed_forward", name + ".norm_ff")
        sa, cv, bn = name + ".self_attn", name + ".conv", name + ".conv.batch_norm"
        G.att_ln_g, G.att_ln_b = gp(name + ".norm_self_attn.weight"), gp(name + ".norm_self_attn.bias")
        G.wqkv = A.g_span(sa + ".linear_q.weight", sa + ".linear_v.weight", (3 * d, d)).data_ptr()
        G.bqkv = A.g_span(sa + ".linear_q.bias", sa + ".linear_v.bias", (3 * d,)).data_ptr()
        G.wpos = gp(sa + ".linear_pos.weight")
        G.bias_u, G.bias_v = gp(sa + ".pos_bias_u"), gp(sa + ".pos_bias_v")
        G.wout, G.bout = gp(sa + ".linear_out.weight"), gp(sa + ".linear_out.bias")
        G.cv_ln_g, G.cv_ln_b = gp(name + ".norm_conv.weight"), gp(name + ".norm_conv.bias")
        G.pw1, G.pw1_b = gp(cv + ".pointwise_conv1.weight"), gp(cv + ".pointwise_conv1.bias")
        G.dw_w, G.dw_b = gp(cv + ".depthwise_conv.weight"), gp(cv + ".depthwise_conv.bias")
        G.bn_g, G.bn_b = gp(bn + ".weight"), gp(bn + ".bias")
        G.pw2, G.pw2_b = gp(cv + ".pointwise_conv2.weight"), gp(cv + ".pointwise_conv2.bias")
        G.fin_ln_g, G.fin_ln_b = gp(name + ".norm_final.weight"), gp(name + ".norm_final.bias")
        self.grads[li] = (G, guard)
        return G

    def backward(self, li, st, dy, dx, ln_part, deferred):
        """one C-ABI call for the whole layer (csrc/layer.hip: emoasr_conformer_layer_bwd).  dy / dx: [M, d] gradient at
        the layer's output / input (dx is written); ln_part: f32 [5, stride] scratch of this layer's LayerNorm partial
        sums, recorded in `deferred` for ops.layernorm_bwd_finalize."""
        eng, A = self.eng, self.eng.arena
        L, G = self._layer_params(li), self._layer_grads(li)
        B, T = st.B, st.T
        if st.lay.segs is None:
            nb = lib.size_query("emoasr_conformer_layer_bwd_ws_bytes", ops.dt(st.wt), B, T, eng.d, eng.h, L.F, L.K)
        else:
            nb = lib.ws_bytes_seg(ops.dt(st.wt), st.io.seg, eng.d, eng.h, L.F, L.K)
        k = (self.calls & 1) if self.wgrad_side else 0
        self.calls += 1
        if self.bwd_ws[k] is None or self.bwd_ws[k].numel() < nb:
            self.join_wgrads()   # (a launch in flight may still read the buffer that is being replaced)
            self.bwd_ws[k] = torch.empty(int(nb * 1.1) + 256, device=dy.device, dtype=torch.uint8)
        io = lib.ConformerBwd()
        io.dy, io.dx = dy.data_ptr(), dx.data_ptr()
        io.ws, io.ws_bytes = self.bwd_ws[k].data_ptr(), self.bwd_ws[k].numel()
        if st.wt.dtype != torch.bfloat16:
            # f32: the materialised attention backward's P^T / dS^T / dBD images, zero-filled ONCE per backward sweep (every layer of
            # a sweep masks the same entries) -- the sweep is known by the forward pass that made its stashes
            seg = st.io.seg
            if st.lay.segs is None:
                seg = lib.Segments()
                seg.n, seg.B[0], seg.T[0] = 1, B, T
            ni = lib.img_bytes_seg(ops.dt(st.wt), seg, eng.d, eng.h)
            key = st.sweep
            if self.attn_img is None or self.attn_img.numel() < ni:
                self.attn_img, self.attn_img_key = torch.empty(int(ni * 1.1) + 256, device=dy.device, dtype=torch.uint8), None
            if self.attn_img_key != key:
                self.attn_img[:ni].zero_()
                self.attn_img_key = key
            io.attn_img, io.attn_img_bytes = self.attn_img.data_ptr(), self.attn_img.numel()
        io.ln_part, io.ln_part_stride = ln_part.data_ptr(), ln_part.stride(0)
        lib.call("emoasr_conformer_layer_bwd", ops.dt(st.wt), ctypes.byref(L), ctypes.byref(G), ctypes.byref(st.io),
                 ctypes.byref(io), ops._stream())
        name = f"encoder.transformers.{li}"
        M = st.lay.M
        for k, norm in enumerate(("norm_final", "norm_ff", "norm_conv", "norm_self_attn", "norm_ff_macaron")):
            deferred.append((M, eng.d, ln_part[k], A.g(f"{name}.{norm}.weight"), A.g(f"{name}.{norm}.bias")))

    def join_wgrads(self, keep=0):
        """the current stream waits for the weight-gradient launches on the side stream: all (keep = 0) or all but the latest"""
        if self.wgrad_side:
            lib.call("emoasr_wgrad_side_join", int(keep), ops._stream())


def tb_prev(x, esz):
    """device address of a layer input given as a tensor or as the previous layer's stash"""
    if isinstance(x, LayerStash):
        return x.wt.data_ptr() + x.lay.t["y"][0] * esz
    return x.data_ptr()
