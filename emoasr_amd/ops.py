"""Tensor-level wrappers over the C ABI (emoasr_amd/lib.py).

torch is used only as the owner of device memory and of the current HIP stream; every
function here enqueues hand-written HIP kernels from libemoasr_hip.so.  Nothing in this
module has autograd: forward and backward ops are separate entry points and the model
code (emoasr_amd/engine.py) sequences them explicitly.
"""
import ctypes
from ctypes import byref, c_void_p

import os
import threading

import torch

from . import lib
from .lib import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SWISH, BF16, F32, F32X3, AttnArgs, Epilogue  # noqa: F401

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt(t):
    """dtype code of a call on tensor t: F32 / BF16, or F32X3 for an f32 tensor while the calling thread's engine runs split products"""
    if t.dtype == torch.float32 and split_products():
        return F32X3
    return _DT[t.dtype]


def _p(t):
    return None if t is None else c_void_p(t.data_ptr())


_STREAM_CACHE = None


def _stream():
    if _STREAM_CACHE is not None:
        return _STREAM_CACHE
    return c_void_p(torch.cuda.current_stream().cuda_stream)


# "f32x3" (lib.F32X3): f32 tensors whose matrix products run as three bf16 MFMAs over (hi, lo) operand pairs.  The mode is an
# argument of every C call (the dtype code dt() returns); on this side it is a property of the calling engine, held per THREAD for
# the duration of the engine's entry point (an autograd thread or a second engine never sees another's mode; there is no library
# state to go stale -- rounds 4-5 had a process-wide option).
# EMOASR_FORCE_SPLIT=1: EVERY f32 product of the process runs split, whatever an engine asks for (test aid: the whole GPU suite's f32
# cases under the split arithmetic, profiles/r05_forced_split_suite.txt)
_FORCE_SPLIT = os.environ.get("EMOASR_FORCE_SPLIT", "0") == "1"
_mode = threading.local()


def split_products(on=None):
    """the calling thread's f32 product mode: split_products(True / False) sets it, split_products() reads it"""
    if on is not None:
        _mode.split = bool(on)
    return bool(getattr(_mode, "split", False)) or _FORCE_SPLIT


class stream_scope:
    """Resolve torch's current stream once for a whole forward/backward sequence (the lookup costs
    more than a kernel launch when done per op).  f32_split: also assert that mode of the f32 products (None: leave it)."""

    def __init__(self, f32_split=None):
        self.f32_split = f32_split

    def __enter__(self):
        global _STREAM_CACHE
        self.prev = _STREAM_CACHE
        _STREAM_CACHE = c_void_p(torch.cuda.current_stream().cuda_stream)
        if self.f32_split is not None:
            split_products(self.f32_split)
        return self

    def __exit__(self, *exc):
        global _STREAM_CACHE
        _STREAM_CACHE = self.prev
        return False


def _chk(t, dtype=None):
    assert t.is_cuda, "emoasr_amd ops need device tensors (no CPU fallback)"
    if dtype is not None:
        assert t.dtype == dtype, f"expected {dtype}, got {t.dtype}"
    return t


def _rows(t):
    """(rows, cols, ld) of a 2-D view whose last dim is contiguous."""
    assert t.stride(-1) == 1
    if t.dim() == 1:
        return 1, t.shape[0], t.shape[0]
    assert t.dim() == 2
    return t.shape[0], t.shape[1], t.stride(0)


def make_epilogue(bias=None, act=ACT_NONE, alpha=1.0, residual=None, res_scale=1.0, pre_out=None,
                  dact_pre=None, dact=ACT_NONE, drop_p=0.0, seed=0, out_f32=False):
    return Epilogue(None if bias is None else bias.data_ptr(),
                    None if residual is None else residual.data_ptr(),
                    None if pre_out is None else pre_out.data_ptr(),
                    None if dact_pre is None else dact_pre.data_ptr(),
                    alpha, res_scale, act, dact, 0 if residual is None else residual.stride(0),
                    1 if out_f32 else 0, drop_p, seed)


def gemm_nt(a, b, out=None, **epi):
    """out[M,N] = epilogue(a[M,K] @ b[N,K]^T)"""
    M, K, lda = _rows(_chk(a))
    N, Kb, ldb = _rows(_chk(b, a.dtype))
    assert K == Kb, (a.shape, b.shape)
    out_f32 = epi.get("out_f32", False)
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    ep = make_epilogue(**epi)
    lib.call("emoasr_gemm_nt", dt(a), M, N, K, _p(a), lda, _p(b), ldb, _p(out), out.stride(0), byref(ep),
             _stream())
    return out


def gemm_nn(a, b, out=None, **epi):
    """out[M,N] = epilogue(a[M,K] @ b[K,N])   (b row-major [K,N]: dgrad with b = weight [out,in])"""
    M, K, lda = _rows(_chk(a))
    Kb, N, ldb = _rows(_chk(b, a.dtype))
    assert K == Kb, (a.shape, b.shape)
    out_f32 = epi.get("out_f32", False)
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    ep = make_epilogue(**epi)
    lib.call("emoasr_gemm_nn", dt(a), M, N, K, _p(a), lda, _p(b), ldb, _p(out), out.stride(0), byref(ep),
             _stream())
    return out


def gemm_tn(a, b, out=None, alpha=1.0, accumulate=False, colsum=None, colsum_scale=1.0):
    """out[N1,N2] (+)= alpha * a[K,N1]^T @ b[K,N2]  (f32 out);
    colsum[N1] (+)= colsum_scale * a.sum(0) (fused bias gradient)"""
    K, N1, lda = _rows(_chk(a))
    Kb, N2, ldb = _rows(_chk(b, a.dtype))
    assert K == Kb
    if out is None:
        assert not accumulate
        out = torch.empty(N1, N2, device=a.device, dtype=torch.float32)
    _chk(out, torch.float32)
    lib.call("emoasr_gemm_tn", dt(a), N1, N2, K, _p(a), lda, _p(b), ldb, _p(out), out.stride(0), alpha,
             int(accumulate), _p(colsum), colsum_scale, _stream())
    return out


def gemm_tn_grouped(problems, stream=None):
    """problems: list of (a[K,N1], b[K,N2], out[N1,N2] f32, alpha, colsum or None, colsum_scale); every
    out (and colsum) is accumulated into, all in one launch per lib.TN_GROUP_MAX problems.
    stream: raw HIP stream handle (int) to enqueue on instead of torch's current stream."""
    sptr = _stream() if stream is None else c_void_p(stream)
    for i0 in range(0, len(problems), lib.TN_GROUP_MAX):
        chunk = problems[i0:i0 + lib.TN_GROUP_MAX]
        arr = (lib.TnProblem * len(chunk))()
        for q, (a, b, out, alpha, colsum, colsum_scale) in zip(arr, chunk):
            K, N1, lda = _rows(_chk(a))
            Kb, N2, ldb = _rows(_chk(b, a.dtype))
            assert K == Kb and out.shape == (N1, N2) and a.dtype == chunk[0][0].dtype
            _chk(out, torch.float32)
            q.N1, q.N2, q.K, q.A, q.lda, q.B, q.ldb = N1, N2, K, a.data_ptr(), lda, b.data_ptr(), ldb
            q.C, q.ldc, q.alpha = out.data_ptr(), out.stride(0), alpha
            q.colsum = None if colsum is None else colsum.data_ptr()
            q.colsum_scale = colsum_scale
        lib.call("emoasr_gemm_tn_grouped", dt(chunk[0][0]), len(chunk), arr, sptr)


def gemm_nn_batched(a, b, out, M, N, K, lda, sa, ldb, sb, ldc, sc, nb, nh, alpha=1.0, accumulate=False):
    """out[b,h] (+)= alpha * a[b,h] (M x K, k-contiguous) @ b[b,h] (K x N, k-major); s* = (outer, inner)
    batch strides in elements; base pointers are the tensors' data pointers."""
    lib.call("emoasr_gemm_nn_batched", dt(a), M, N, K, _p(a), lda, sa[0], sa[1], _p(b), ldb, sb[0], sb[1],
             _p(out), ldc, sc[0], sc[1], nb, nh, alpha, int(accumulate), _stream())
    return out


def colsum(x, out=None, scale=1.0, accumulate=False):
    M, N, ld = _rows(_chk(x))
    if out is None:
        out = torch.empty(N, device=x.device, dtype=torch.float32)
    lib.call("emoasr_colsum", dt(x), M, N, _p(x), ld, _p(_chk(out, torch.float32)), scale, int(accumulate),
             _stream())
    return out


# ---- front-end -----------------------------------------------------------------------
def conv1_fwd(x, w1, b1, dtype):
    B, T, F = x.shape
    C = w1.shape[0]
    T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    y1 = torch.empty(B, T1, F1, C, device=x.device, dtype=dtype)
    lib.call("emoasr_conv1_fwd", _DT[dtype], B, T, F, C, _p(_chk(x, torch.float32)), _p(w1), _p(b1), _p(y1),
             _stream())
    return y1


def conv1_wgrad(x, dy1, dw1, db1, accumulate=False):
    B, T, F = x.shape
    C = dy1.shape[-1]
    scratch = torch.empty(lib.size_query("emoasr_conv1_wgrad_scratch_floats", B, T, C), device=x.device, dtype=torch.float32)
    lib.call("emoasr_conv1_wgrad", dt(dy1), B, T, F, C, _p(x), _p(dy1), _p(dw1), _p(db1), int(accumulate),
             _p(scratch), _stream())


def conv2_fwd(y1, w, out=None, **epi):
    B, T1, F1, C = y1.shape
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    if out is None:
        y2 = torch.empty(B, T2, F2, C, device=y1.device, dtype=y1.dtype)
    else:  # (a contiguous slice of a larger buffer: the stacked micro-batches of engine.ctc_train_stacked)
        assert out.is_contiguous() and out.numel() == B * T2 * F2 * C and out.dtype == y1.dtype
        y2 = out.view(B, T2, F2, C)
    ep = make_epilogue(**epi)
    lib.call("emoasr_conv2_fwd", dt(y1), B, T1, F1, C, _p(y1), _p(_chk(w, y1.dtype)), _p(y2), byref(ep),
             _stream())
    return y2


def conv2_wgrad(dy2, y1, dw, dbias=None, accumulate=False):
    B, T1, F1, C = y1.shape
    lib.call("emoasr_conv2_wgrad", dt(y1), B, T1, F1, C, _p(dy2), _p(y1), _p(_chk(dw, torch.float32)),
             _p(dbias), int(accumulate), _stream())


def conv2_dgrad(dy2, w, y1):
    """dy2 [B,T2,F2,C] (or [B*T2*F2, C]), w [C, 9C] (conv2_fwd's layout), y1 [B,T1,F1,C] -> dy1 = relu'(y1) * conv2^T(dy2)"""
    B, T1, F1, C = y1.shape
    dy1 = torch.empty_like(y1)
    lib.call("emoasr_conv2_dgrad", dt(y1), B, T1, F1, C, _p(_chk(dy2, y1.dtype)), _p(_chk(w, y1.dtype)), _p(y1), _p(dy1),
             _stream())
    return dy1


def conv2_dgrad_kc(dy2, wt, y1):
    """as conv2_dgrad on the large-tile kernel: wt [C, 9C] = conv.2.weight.permute(1, 2, 3, 0) (c, kh, kw, n); bf16, C % 256 == 0"""
    B, T1, F1, C = y1.shape
    dy1 = torch.empty_like(y1)
    lib.call("emoasr_conv2_dgrad_kc", dt(y1), B, T1, F1, C, _p(_chk(dy2, y1.dtype)), _p(_chk(wt, y1.dtype)), _p(y1), _p(dy1),
             _stream())
    return dy1


def gemm_nt_big(a, b, out=None, bias=None, relu=False):
    """out[M,N] = relu?(a[M,K] @ b[N,K]^T + bias) on the large-tile kernel (bf16, N % 8 == 0, K % 64 == 0)"""
    M, K, lda = _rows(_chk(a, torch.bfloat16))
    N, Kb, ldb = _rows(_chk(b, a.dtype))
    assert K == Kb, (a.shape, b.shape)
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=a.dtype)
    lib.call("emoasr_gemm_nt_big", dt(a), M, N, K, _p(a), lda, _p(b), ldb, _p(out), out.stride(0), _p(bias), int(relu),
             _stream())
    return out


def conv2_col2im(dcol, y1):
    B, T1, F1, C = y1.shape
    dy1 = torch.empty_like(y1)
    lib.call("emoasr_conv2_col2im", dt(y1), B, T1, F1, C, _p(dcol), _p(y1), _p(dy1), _stream())
    return dy1


# ---- LayerNorm -------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, want_stats=True):
    M, N, ld = _rows(_chk(x))
    assert ld == N
    y = torch.empty_like(x)
    mean = torch.empty(M, device=x.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(M, device=x.device, dtype=torch.float32) if want_stats else None
    lib.call("emoasr_layernorm_fwd", dt(x), M, N, _p(x), _p(gamma), _p(beta), eps, _p(y), _p(mean), _p(rstd),
             _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, branch=None, deferred=None):
    """dx = dres + LN'(dy); dgamma / dbeta accumulated.
    branch=(scale, p, seed): also return dropout(dx * scale) -- the gradient entering the next residual
    branch of the backward sweep (replaces a scale_dropout pass); returns (dx, dy_branch).
    deferred: a list; the dgamma/dbeta fold is postponed and recorded there for layernorm_bwd_finalize."""
    M, N, ld = _rows(_chk(x))
    dx = torch.empty_like(x)
    scratch = None
    if dgamma is not None or dbeta is not None:
        scratch = torch.empty(lib.size_query("emoasr_layernorm_bwd_scratch_floats", N), device=x.device, dtype=torch.float32)
    if branch is None and deferred is None:
        lib.call("emoasr_layernorm_bwd", dt(x), M, N, _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx),
                 _p(dgamma), _p(dbeta), _p(scratch), _stream())
        return dx
    o = lib.LnBwdOpts()
    dy2 = None
    if branch is not None:
        dy2 = torch.empty_like(x)
        o.dy2, o.scale2, o.drop_p2, o.seed2 = dy2.data_ptr(), branch[0], branch[1], branch[2]
    if deferred is not None and scratch is not None:
        o.defer_finalize = 1
        deferred.append((M, N, scratch, dgamma, dbeta))
    lib.call("emoasr_layernorm_bwd_ex", dt(x), M, N, _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx),
             _p(dgamma), _p(dbeta), _p(scratch), byref(o), _stream())
    return dx if branch is None else (dx, dy2)


def layernorm_bwd_finalize(deferred):
    """fold the dgamma / dbeta partials of deferred layernorm_bwd calls (one launch per 64 of them)"""
    for i0 in range(0, len(deferred), lib.LN_FINALIZE_MAX):
        chunk = deferred[i0:i0 + lib.LN_FINALIZE_MAX]
        arr = (lib.LnFinalizeItem * len(chunk))()
        for it, (M, N, scratch, dgamma, dbeta) in zip(arr, chunk):
            it.M, it.N, it.part = M, N, scratch.data_ptr()
            it.dgamma = None if dgamma is None else dgamma.data_ptr()
            it.dbeta = None if dbeta is None else dbeta.data_ptr()
        lib.call("emoasr_layernorm_bwd_finalize", len(chunk), arr, _stream())
    del deferred[:]


# ---- attention -------------------------------------------------------------------------
def _attn_args(q, k, v, H, pos, bias_u, bias_v, klens, causal, scale, drop_p, seed):
    B, Tq, D = q.shape
    Tk = k.shape[1]
    a = AttnArgs()
    a.B, a.H, a.DK, a.Tq, a.Tk = B, H, D // H, Tq, Tk
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    assert q.stride(0) == Tq * q.stride(1) and k.stride(0) == Tk * k.stride(1) and v.stride(0) == Tk * v.stride(1)
    a.ldq, a.ldk, a.ldv = q.stride(1), k.stride(1), v.stride(1)
    a.q, a.k, a.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    if pos is not None:
        a.pos, a.ldp = pos.data_ptr(), pos.stride(0)
    a.bias_u = None if bias_u is None else bias_u.data_ptr()
    a.bias_v = None if bias_v is None else bias_v.data_ptr()
    a.klens = None if klens is None else klens.data_ptr()
    a.causal = int(causal)
    a.scale, a.drop_p, a.seed = scale, drop_p, seed
    return a


def attn_dropmask(q, k, H, klens=None, drop_p=0.0, seed=0):
    """the attention-dropout keep mask of an attn_fwd / attn_bwd call with these arguments as bits: uint32 [B * Tq, H, ceil(Tk / 32)]
    (emoasr_attn_t::keep_mask: hashed once, tested by the forward and both passes of the fused backward)"""
    B, Tq, _ = q.shape
    nw = lib.size_query("emoasr_attn_dropmask_words", k.shape[1])
    a = _attn_args(q, k, k, H, None, None, None, klens, False, 1.0, drop_p, seed)
    mask = torch.zeros(B * Tq, H, nw, device=q.device, dtype=torch.int32)
    lib.call("emoasr_attn_dropmask", dt(q), byref(a), mask.data_ptr(), nw, _stream())
    return mask


def attn_fwd(q, k, v, H, scale, pos=None, bias_u=None, bias_v=None, klens=None, causal=False, drop_p=0.0,
             seed=0, store_scores=False, keep_mask=None):
    """q [B,Tq,D], k/v [B,Tk,D] (views with a row stride are fine) -> out [B,Tq,D], lse [B,H,Tq]
    (+ st f32 [B,H,Tk,ldst], the scaled scores S^T kept for the backward, if store_scores)"""
    B, Tq, D = q.shape
    a = _attn_args(q, k, v, H, pos, bias_u, bias_v, klens, causal, scale, drop_p, seed)
    out = torch.empty(B, Tq, D, device=q.device, dtype=q.dtype)
    lse = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
    a.out, a.ldo, a.lse = out.data_ptr(), D, lse.data_ptr()
    if keep_mask is not None:
        a.keep_mask, a.keep_nw = keep_mask.data_ptr(), keep_mask.shape[-1]
    st = None
    if store_scores:
        st = torch.empty(B, H, k.shape[1], (Tq + 7) // 8 * 8, device=q.device, dtype=torch.float32)
        a.st, a.ldst = st.data_ptr(), st.shape[-1]
    lib.call("emoasr_attn_fwd", dt(q), byref(a), _stream())
    return (out, lse, st) if store_scores else (out, lse)


class AttnScratch:
    """Zero-initialised HBM scratch of the materialised attention backward (P^T, dS^T, dBD band).
    Reusable across calls that share (B, H, Tq, Tk, klens) -- e.g. all layers of one step."""

    def __init__(self, B, H, Tq, Tk, dtype, device, rel):
        self.key = (B, H, Tq, Tk, dtype, rel)
        self.ldpd = (Tq + 7) // 8 * 8
        self.ldbd = (2 * Tq - 1 + 7) // 8 * 8
        self.pdT = torch.zeros(B, H, Tk, self.ldpd, device=device, dtype=dtype)
        self.dsT = torch.zeros(B, H, Tk, self.ldpd, device=device, dtype=dtype)
        self.dbd = torch.zeros(H, B, Tq, self.ldbd, device=device, dtype=dtype) if rel else None
        self.cs = torch.empty(H, self.ldbd, device=device, dtype=torch.float32) if rel else None
        # Q + pos_bias_u / Q + pos_bias_v and the per-query-tile dbias partial sums (emoasr_attn_t.qu/qv/dbias_part)
        self.qu = torch.empty(B, Tq, H * 64, device=device, dtype=dtype) if rel else None
        self.qv = torch.empty(B, Tq, H * 64, device=device, dtype=dtype) if rel else None
        self.dbias_part = torch.empty(B * ((Tq + 31) // 32), H, 2, 64, device=device, dtype=torch.float32)


_FUSED_WS = {}


def fused_attn_bwd_ok(q, pos, bias_u, bias_v, causal):
    """can emoasr_attn_bwd_fused take this call? (bf16, no causal mask, relative positions with both biases or none)"""
    return (q.dtype == torch.bfloat16 and not causal and (pos is None or (bias_u is not None and bias_v is not None))
            and (pos is not None or (bias_u is None and bias_v is None)))


def _fused_ws(device, nbytes):
    """scratch of the single-pass attention backward: one buffer per device, grown on demand (no initialisation needed;
    every call runs on the current stream, so calls never overlap)"""
    w = _FUSED_WS.get(device)
    if w is None or w.numel() < nbytes:
        w = _FUSED_WS[device] = torch.empty(int(nbytes * 1.25) + 256, device=device, dtype=torch.uint8)
    return w


def attn_bwd(dout, out, lse, q, k, v, H, scale, dq, dk, dv, pos=None, bias_u=None, bias_v=None, klens=None,
             causal=False, drop_p=0.0, seed=0, dpos=None, dbias_u=None, dbias_v=None, scratch=None,
             materialise=True, st=None, keep_mask=None):
    """dq/dk/dv are written (same strides as q/k/v); dpos/dbias_* are accumulated into.
    materialise="fused": the single-pass kernel (bf16, no causal mask; see fused_attn_bwd_ok);
    materialise=True: dV/dK/dpos through batched GEMMs over stored P^T/dS^T (scratch is allocated
    here unless an AttnScratch for this shape/mask is passed); False: score-recompute kernels."""
    B, Tq, D = q.shape
    if materialise == "fused":
        assert fused_attn_bwd_ok(q, pos, bias_u, bias_v, causal)
        a = _attn_args(q, k, v, H, pos, bias_u, bias_v, klens, causal, scale, drop_p, seed)
        if keep_mask is not None:
            a.keep_mask, a.keep_nw = keep_mask.data_ptr(), keep_mask.shape[-1]
        assert dq.stride() == q.stride() and dk.stride() == k.stride() and dv.stride() == v.stride()
        assert dout.is_contiguous() and out.is_contiguous()
        delta = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
        a.out, a.ldo, a.lse = out.data_ptr(), D, lse.data_ptr()
        a.dout, a.delta = dout.data_ptr(), delta.data_ptr()
        a.dq, a.dk, a.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
        a.dpos = None if dpos is None else dpos.data_ptr()
        a.dbias_u = None if dbias_u is None else dbias_u.data_ptr()
        a.dbias_v = None if dbias_v is None else dbias_v.data_ptr()
        nb = lib.size_query("emoasr_attn_bwd_fused_ws_bytes", dt(q), B, H, Tq, k.shape[1], int(pos is not None))
        ws = _fused_ws(q.device, nb)
        lib.call("emoasr_attn_bwd_fused", dt(q), byref(a), ws.data_ptr(), ws.numel(), _stream())
        return
    a = _attn_args(q, k, v, H, pos, bias_u, bias_v, klens, causal, scale, drop_p, seed)
    assert dq.stride() == q.stride() and dk.stride() == k.stride() and dv.stride() == v.stride()
    assert dout.is_contiguous() and out.is_contiguous()
    delta = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
    a.out, a.ldo, a.lse = out.data_ptr(), D, lse.data_ptr()
    a.dout, a.delta = dout.data_ptr(), delta.data_ptr()
    a.dq, a.dk, a.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    a.dpos = None if dpos is None else dpos.data_ptr()
    a.dbias_u = None if dbias_u is None else dbias_u.data_ptr()
    a.dbias_v = None if dbias_v is None else dbias_v.data_ptr()
    if materialise:
        rel = pos is not None
        if scratch is None:
            scratch = AttnScratch(B, H, Tq, k.shape[1], q.dtype, q.device, rel)
        assert scratch.key == (B, H, Tq, k.shape[1], q.dtype, rel), "AttnScratch built for another shape"
        a.pdT, a.dsT, a.ldpd = scratch.pdT.data_ptr(), scratch.dsT.data_ptr(), scratch.ldpd
        if rel:
            a.dbd, a.ldbd, a.cs = scratch.dbd.data_ptr(), scratch.ldbd, scratch.cs.data_ptr()
            if bias_u is not None and bias_v is not None and q.shape[2] == H * 64:
                a.qu, a.qv = scratch.qu.data_ptr(), scratch.qv.data_ptr()
        a.dbias_part = scratch.dbias_part.data_ptr()
        if st is not None:  # scores stored by attn_fwd(store_scores=True): no recomputation
            a.st, a.ldst = st.data_ptr(), st.shape[-1]
    lib.call("emoasr_attn_bwd", dt(q), byref(a), _stream())


# ---- conv module -----------------------------------------------------------------------
def glu_fwd(x):
    M, C2, ld = _rows(_chk(x))
    out = torch.empty(M, C2 // 2, device=x.device, dtype=x.dtype)
    lib.call("emoasr_glu_fwd", dt(x), M, C2 // 2, _p(x), _p(out), _stream())
    return out


def glu_bwd(x, dout):
    M, C2, ld = _rows(_chk(x))
    din = torch.empty_like(x)
    lib.call("emoasr_glu_bwd", dt(x), M, C2 // 2, _p(x), _p(dout), _p(din), _stream())
    return din


def dwconv_fwd(x, w, bias):
    B, T, C = x.shape
    y = torch.empty_like(x)
    lib.call("emoasr_dwconv_fwd", dt(x), B, T, C, w.shape[-1], _p(_chk(x)), _p(w), _p(bias), _p(y), _stream())
    return y


def dwconv_bn_stats_fwd(x, w, bias, running_mean=None, running_var=None, momentum=0.1, num_batches_tracked=None):
    """depthwise conv + training-mode BatchNorm statistics of its output in two launches:
    -> (y, batch mean [C], biased batch var [C]); running stats / counter updated in place."""
    B, T, C = x.shape
    y = torch.empty_like(x)
    part = torch.empty(lib.size_query("emoasr_dwconv_stats_floats", B, T, C), device=x.device, dtype=torch.float32)
    lib.call("emoasr_dwconv_fwd_stats", dt(x), B, T, C, w.shape[-1], _p(_chk(x)), _p(w), _p(bias), _p(y), _p(part),
             _stream())
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    var = torch.empty(C, device=x.device, dtype=torch.float32)
    if num_batches_tracked is not None:
        _chk(num_batches_tracked, torch.int64)
    lib.call("emoasr_bn_stats_finalize", B, T, C, _p(part), _p(mean), _p(var), _p(running_mean), _p(running_var),
             momentum, _p(num_batches_tracked), _stream())
    return y, mean, var


def glu_dwconv_fwd(g, B, T, w, bias, running_mean=None, running_var=None, momentum=0.1, num_batches_tracked=None,
                   training=True):
    """c = depthwise_conv(GLU(g)) in one launch (bf16; g [B*T, 2C]); training: + the BatchNorm batch statistics of c
    -> (c [B,T,C], mean, var) like dwconv_bn_stats_fwd, else (c, running_mean, running_var)."""
    C = g.shape[-1] // 2
    c = torch.empty(B, T, C, device=g.device, dtype=g.dtype)
    if not training:
        lib.call("emoasr_glu_dwconv_fwd", dt(g), B, T, C, w.shape[-1], _p(_chk(g)), _p(w), _p(bias), _p(c), None, _stream())
        return c, running_mean, running_var
    part = torch.empty(lib.size_query("emoasr_dwconv_stats_floats", B, T, C), device=g.device, dtype=torch.float32)
    lib.call("emoasr_glu_dwconv_fwd", dt(g), B, T, C, w.shape[-1], _p(_chk(g)), _p(w), _p(bias), _p(c), _p(part), _stream())
    mean = torch.empty(C, device=g.device, dtype=torch.float32)
    var = torch.empty(C, device=g.device, dtype=torch.float32)
    if num_batches_tracked is not None:
        _chk(num_batches_tracked, torch.int64)
    lib.call("emoasr_bn_stats_finalize", B, T, C, _p(part), _p(mean), _p(var), _p(running_mean), _p(running_var),
             momentum, _p(num_batches_tracked), _stream())
    return c, mean, var


def conv_bwd_fused(ds, c, mean, var, gamma, beta, eps, dgamma, dbeta, g, w, dw, dbias, B, T):
    """backward of GLU -> depthwise conv -> BatchNorm(training) -> Swish in three launches (bf16): BatchNorm sums + fold,
    then apply / depthwise data gradient / GLU backward / depthwise weight-gradient partials fused, then their fold.
    ds: gradient w.r.t. the Swish output [B*T, C]; c: the convolution's output; g: the GLU input [B*T, 2C] -> dg."""
    import ctypes
    M, C, _ = _rows(_chk(c))
    K = w.shape[-1]
    scr = torch.empty(lib.size_query("emoasr_bn_swish_bwd_scratch_floats", M, C), device=c.device, dtype=torch.float32)
    tot = ctypes.c_void_p()
    lib.call("emoasr_bn_swish_bwd_sums", dt(c), M, C, _p(_chk(ds, c.dtype)), _p(c), _p(mean), _p(var), _p(gamma), _p(beta),
             eps, _p(dgamma), _p(dbeta), _p(scr), ctypes.byref(tot), _stream())
    dg = torch.empty_like(g)
    wscr = torch.empty(lib.size_query("emoasr_dwconv_bwd_w_scratch_floats", B, T, C, K), device=c.device, dtype=torch.float32)
    lib.call("emoasr_conv_bwd_fused", dt(c), B, T, C, K, _p(ds), _p(c), _p(mean), _p(var), _p(gamma), _p(beta), eps, tot,
             _p(_chk(g, c.dtype)), _p(w), _p(dg), _p(dw), _p(dbias), _p(wscr), _stream())
    return dg


def dwconv_bwd_x(dy, w):
    B, T, C = dy.shape
    dx = torch.empty_like(dy)
    lib.call("emoasr_dwconv_bwd_x", dt(dy), B, T, C, w.shape[-1], _p(dy), _p(w), _p(dx), _stream())
    return dx


def dwconv_bwd_w(dy, x, dw, dbias, accumulate=False):
    B, T, C = dy.shape
    K = dw.shape[-1]
    scratch = torch.empty(lib.size_query("emoasr_dwconv_bwd_w_scratch_floats", B, T, C, K), device=dy.device,
                          dtype=torch.float32)
    lib.call("emoasr_dwconv_bwd_w", dt(dy), B, T, C, K, _p(dy), _p(x), _p(dw), _p(dbias),
             int(accumulate), _p(scratch), _stream())


def bn_stats(y, running_mean=None, running_var=None, momentum=0.1):
    M, C, ld = _rows(_chk(y))
    mean = torch.empty(C, device=y.device, dtype=torch.float32)
    var = torch.empty(C, device=y.device, dtype=torch.float32)
    lib.call("emoasr_bn_stats", dt(y), M, C, _p(y), _p(mean), _p(var), _p(running_mean), _p(running_var),
             momentum, _stream())
    return mean, var


def bn_swish_fwd(y, mean, var, gamma, beta, eps):
    M, C, ld = _rows(_chk(y))
    z = torch.empty_like(y)
    lib.call("emoasr_bn_swish_fwd", dt(y), M, C, _p(y), _p(mean), _p(var), _p(gamma), _p(beta), eps, _p(z),
             _stream())
    return z


def bn_swish_bwd(dz, y, mean, var, gamma, beta, eps, dgamma, dbeta):
    M, C, ld = _rows(_chk(y))
    dy = torch.empty_like(y)
    scratch = torch.empty(lib.size_query("emoasr_bn_swish_bwd_scratch_floats", M, C), device=y.device, dtype=torch.float32)
    lib.call("emoasr_bn_swish_bwd", dt(y), M, C, _p(dz), _p(y), _p(mean), _p(var), _p(gamma), _p(beta), eps,
             _p(dy), _p(dgamma), _p(dbeta), _p(scratch), _stream())
    return dy


# ---- element-wise ----------------------------------------------------------------------
def strided_copy(src, out=None, out_dtype=None, accumulate=False):
    """out (contiguous, src.shape) (+)= src (any strides, <= 4 dims); casts f32 <-> bf16."""
    assert src.dim() <= 4
    shape = [1] * (4 - src.dim()) + list(src.shape)
    strides = [0] * (4 - src.dim()) + list(src.stride())
    if out is None:
        out = torch.empty(src.shape, device=src.device, dtype=out_dtype or src.dtype)
    assert out.is_contiguous()
    lib.call("emoasr_strided_copy", dt(src), dt(out), _p(src), _p(out), *shape, *strides, int(accumulate),
             _stream())
    return out


def transpose_cast_batched(pairs):
    """pairs: [(src f32 [rows, cols] contiguous, dst [cols, rows] compute dtype, last dim contiguous)] -> every dst = src^T, one
    launch per lib.TC_MAX pairs (the transposed weight copies of the layer runtime, refreshed with the parameter shadow)"""
    for i0 in range(0, len(pairs), lib.TC_MAX):
        chunk = pairs[i0:i0 + lib.TC_MAX]
        arr = (lib.TcItem * len(chunk))()
        for q, (src, dst) in zip(arr, chunk):
            rows, cols = src.shape
            assert src.is_contiguous() and src.dtype == torch.float32 and tuple(dst.shape) == (cols, rows) and dst.stride(1) == 1
            assert dst.dtype == chunk[0][1].dtype
            q.src, q.dst, q.rows, q.cols, q.ld_dst = src.data_ptr(), dst.data_ptr(), rows, cols, dst.stride(0)
        lib.call("emoasr_transpose_cast_batched", dt(chunk[0][1]), len(chunk), arr, _stream())


def scale_dropout(x, scale=1.0, drop_p=0.0, seed=0):
    y = torch.empty_like(x)
    lib.call("emoasr_scale_dropout", dt(x), x.numel(), _p(_chk(x)), _p(y), scale, drop_p, seed, _stream())
    return y


def posenc(x, pe, scale, drop_p=0.0, seed=0):
    B, T, N = x.shape
    y = torch.empty_like(x)
    lib.call("emoasr_posenc", dt(x), B, T, N, _p(_chk(x)), _p(pe), scale, drop_p, seed, _p(y), _stream())
    return y


def add(a, b):
    y = torch.empty_like(a)
    lib.call("emoasr_add", dt(a), a.numel(), _p(a), _p(b), _p(y), _stream())
    return y


# ---- CTC -------------------------------------------------------------------------------
def gemm_nt_lse(a, b, bias):
    """logits = a @ b^T + bias and their row log-sum-exp in ONE pass over the logits (the CTC head: emoasr_gemm_nt_lse) when the
    large-tile kernel takes the shape (bf16, N % 8 == 0, K % 64 == 0, operands below 4 GiB); else the product + a row pass"""
    M, K, lda = _rows(_chk(a))
    N, Kb, ldb = _rows(_chk(b, a.dtype))
    assert K == Kb
    if (a.dtype == torch.bfloat16 and N % 8 == 0 and N >= 256 and K % 64 == 0 and lda % 8 == 0 and ldb % 8 == 0
            and M * lda * 2 < (1 << 32) and N * ldb * 2 < (1 << 32) and M >= 2048):
        # rows padded to a multiple of 64 columns (V = 10 000: 20 000-byte rows put every 128-byte store of a tile astride two lines)
        Np = (N + 63) // 64 * 64
        out = torch.empty(M, Np, device=a.device, dtype=a.dtype)[:, :N]
        part = torch.empty((N + 63) // 64, M, 2, device=a.device, dtype=torch.float32)
        lse = torch.empty(M, device=a.device, dtype=torch.float32)
        lib.call("emoasr_gemm_nt_lse", dt(a), M, N, K, _p(a), lda, _p(b), ldb, _p(out), Np, _p(bias), _p(part), _p(lse), _stream())
        return out, lse
    out = gemm_nt(a, b, bias=bias)
    return out, row_lse(out)


def row_lse(logits):
    M, V, ld = _rows(_chk(logits))
    lse = torch.empty(M, device=logits.device, dtype=torch.float32)
    lib.call("emoasr_row_lse", dt(logits), M, V, _p(logits), ld, _p(lse), _stream())
    return lse


def ctc_forward(logits, lse, labels, elens, ylens, blank):
    """logits [B,T,V]; labels int32 [B,Lmax]; elens/ylens int32 [B] -> (lp, alpha, beta, nll)"""
    B, T, V = logits.shape
    Lmax = labels.shape[1]
    S = 2 * Lmax + 1
    dev = logits.device
    lp = torch.empty(B, T, S, device=dev, dtype=torch.float32)
    alpha = torch.empty(B, T, S, device=dev, dtype=torch.float32)
    beta = torch.empty(B, T, S, device=dev, dtype=torch.float32)
    nll = torch.empty(B, device=dev, dtype=torch.float32)
    lib.call("emoasr_ctc_forward", dt(logits), B, T, V, Lmax, _p(logits), logits.stride(1), _p(lse), _p(labels),
             _p(elens), _p(ylens), blank, _p(lp), _p(alpha), _p(beta), _p(nll), _stream())
    return lp, alpha, beta, nll


def ctc_grad(logits, lse, labels, elens, ylens, blank, lp, alpha, beta, nll, gscale, gscale_dev=None, out=None):
    B, T, V = logits.shape
    ld = logits.stride(1)  # rows of a ragged vocabulary are padded (engine.head_logits): keep that layout
    if out is not None:
        assert out.shape == logits.shape and out.dtype == logits.dtype and out.stride(2) == 1
        grad = out
    else:
        grad = torch.empty_like(logits) if ld == V else torch.empty(B, T, ld, device=logits.device, dtype=logits.dtype)[..., :V]
    lib.call("emoasr_ctc_grad", dt(logits), B, T, V, labels.shape[1], _p(logits), logits.stride(1), _p(lse),
             _p(labels), _p(elens), _p(ylens), blank, _p(lp), _p(alpha), _p(beta), _p(nll), gscale, _p(gscale_dev),
             _p(grad),
             grad.stride(1), _stream())
    return grad


def ctc_forward_rows(logits2, lse, labels, elens, ylens, blank, row0, Tmax):
    """CTC lattices of utterances whose logits rows are row0[b] + t in the stacked [M, V] `logits2` (several micro-batches in
    one set of launches); the lattice tables are [B, Tmax, S].  -> (lp, alpha, beta, nll)"""
    M, V = logits2.shape
    B, Lmax = labels.shape
    S = 2 * Lmax + 1
    dev = logits2.device
    lp = torch.empty(B, Tmax, S, device=dev, dtype=torch.float32)
    alpha = torch.empty(B, Tmax, S, device=dev, dtype=torch.float32)
    beta = torch.empty(B, Tmax, S, device=dev, dtype=torch.float32)
    nll = torch.empty(B, device=dev, dtype=torch.float32)
    assert row0.dtype == torch.int64
    lib.call("emoasr_ctc_forward_rows", dt(logits2), B, Tmax, V, Lmax, _p(logits2), logits2.stride(0), _p(lse), _p(labels),
             _p(elens), _p(ylens), blank, _p(row0), _p(lp), _p(alpha), _p(beta), _p(nll), _stream())
    return lp, alpha, beta, nll


def ctc_grad_rows(logits2, lse, labels, elens, ylens, blank, lp, alpha, beta, nll, gscale, row0, tpad, uscale, out):
    """gradient rows of ctc_forward_rows' utterances written into `out` [M, V] (every row of every utterance, zeros on padding)"""
    M, V = logits2.shape
    B, Tmax, _ = lp.shape
    assert out.shape == logits2.shape and out.dtype == logits2.dtype and tpad.dtype == torch.int32 and uscale.dtype == torch.float32
    lib.call("emoasr_ctc_grad_rows", dt(logits2), B, Tmax, V, labels.shape[1], _p(logits2), logits2.stride(0), _p(lse), _p(labels),
             _p(elens), _p(ylens), blank, _p(lp), _p(alpha), _p(beta), _p(nll), gscale, None, _p(row0), _p(tpad), _p(uscale),
             _p(out), out.stride(0), _stream())
    return out


def ctc_greedy(logits, elens, blank):
    B, T, V = logits.shape
    dev = logits.device
    # one buffer, three views: the caller reads all of them back with ONE device-to-host copy (ctc_greedy_apply)
    buf = torch.empty(2 * B * T + B, device=dev, dtype=torch.int32)
    best, hyp, hyplen = buf[:B * T].view(B, T), buf[B * T:2 * B * T].view(B, T), buf[2 * B * T:]
    lib.call("emoasr_ctc_greedy", dt(logits), B, T, V, _p(logits), logits.stride(1), _p(elens), blank, _p(best),
             _p(hyp), _p(hyplen), _stream())
    return best, hyp, hyplen


# ---- Transformer decoder side ------------------------------------------------------------
def embed_fwd(ids, table, pe, scale, drop_p=0.0, seed=0):
    """ids int32 [B,L]; table [V,d] (compute dtype); pe f32 [>=L,d] or None -> [B,L,d]"""
    B, L = ids.shape
    d = table.shape[1]
    out = torch.empty(B, L, d, device=table.device, dtype=table.dtype)
    lib.call("emoasr_embed_fwd", dt(table), B * L, L, d, _p(ids), _p(table), _p(pe), scale, drop_p, seed, _p(out),
             _stream())
    return out


def embed_bwd(ids, dout, scale, dtable, drop_p=0.0, seed=0):
    B, L = ids.shape
    lib.call("emoasr_embed_bwd", dt(dout), B * L, dout.shape[-1], _p(ids), _p(dout), scale, drop_p, seed,
             _p(_chk(dtable, torch.float32)), _stream())


def lsm_loss(logits, labels, w, lsm_prob, want_grad=False, gscale=1.0, gscale_dev=None):
    """logits [M,V]; labels int32 [M]; w f32 [M] row weights (0 = padded) -> (loss_rows f32 [M], grad | None)"""
    M, V, ld = _rows(_chk(logits))
    loss = torch.empty(M, device=logits.device, dtype=torch.float32)
    grad = torch.empty_like(logits) if want_grad else None
    lib.call("emoasr_lsm_loss", dt(logits), M, V, _p(logits), ld, _p(labels), _p(w), lsm_prob, _p(loss), gscale,
             _p(gscale_dev), _p(grad), 0 if grad is None else grad.stride(0), _stream())
    return loss, grad


# ---- knowledge distillation ---------------------------------------------------------------
def soft_ce(logits, soft=None, src=None, hard=None, w_soft=None, w_hard=None, lsm_prob=0.0, lrow=None, want_grad=False,
            gscale=1.0, gscale_dev=None, grad=None):
    """logits [M,V]; soft f32 [N,V]; src/hard/lrow int32 [R]; w_* f32 [R] -> (loss rows f32 [R], grad [M,V] | None).
    With `lrow` only the listed logits rows are touched: pass a zero-filled `grad`."""
    M, V, ld = _rows(_chk(logits))
    R = M if lrow is None else lrow.numel()
    loss = torch.empty(R, device=logits.device, dtype=torch.float32)
    if want_grad and grad is None:
        grad = torch.zeros_like(logits) if lrow is not None else torch.empty_like(logits)
    if soft is not None:
        _chk(soft, torch.float32)
        assert soft.shape[-1] == V and soft.is_contiguous()
    lib.call("emoasr_soft_ce", dt(logits), R, V, _p(logits), ld, _p(lrow), _p(soft), V, _p(src), _p(hard), _p(w_soft),
             _p(w_hard), lsm_prob, _p(loss), gscale, _p(gscale_dev), _p(grad) if want_grad else None,
             grad.stride(-2) if want_grad else 0, _stream())
    return loss, grad if want_grad else None


def ctc_best_path(lp, alpha, beta, labels, elens, ylens, blank):
    """lattices of ctc_forward [B,T,S] -> aligns int32 [B,T] (ctc_aligner.py:139-221)"""
    B, T, S = lp.shape
    aligns = torch.empty(B, T, device=lp.device, dtype=torch.int32)
    lib.call("emoasr_ctc_best_path", B, T, labels.shape[1], _p(lp), _p(alpha), _p(beta), _p(labels), _p(elens),
             _p(ylens), blank, _p(aligns), _stream())
    return aligns


def rnnt_best_path(alpha, beta, elens, ylens):
    """lattices of rnnt_forward f32 [B,T,U] -> aligns int32 [B,U-1] (rnnt_aligner.py:186-196)"""
    B, T, U = alpha.shape
    aligns = torch.zeros(B, max(U - 1, 0), device=alpha.device, dtype=torch.int32)
    lib.call("emoasr_rnnt_best_path", B, T, U, _p(alpha), _p(beta), _p(elens), _p(ylens), _p(aligns), _stream())
    return aligns


LABEL_POSITIONS = {"all": 0, "left": 1, "mid": 2, "right": 3}


def ctc_label_map(aligns, xlens, blank, position="all"):
    """aligns int32 [B,T], xlens int32 [B] -> (label_map int32 [B,T] (-1: none), count int32 [B])"""
    B, T = aligns.shape
    lmap = torch.empty(B, T, device=aligns.device, dtype=torch.int32)
    count = torch.empty(B, device=aligns.device, dtype=torch.int32)
    lib.call("emoasr_ctc_label_map", B, T, _p(aligns), _p(xlens), blank, LABEL_POSITIONS[position], _p(lmap), _p(count),
             _stream())
    return lmap, count


# ---- beam search -----------------------------------------------------------------------------
def log_softmax(x, add=None, mu=0.0):
    """x [M,V] (row stride allowed) -> f32 [M,V] = log_softmax(x) + mu*add[:, :V]"""
    M, V, ld = _rows(_chk(x))
    out = torch.empty(M, V, device=x.device, dtype=torch.float32)
    lib.call("emoasr_log_softmax", dt(x), M, V, _p(x), ld, _p(add), 0 if add is None else add.stride(0), mu, _p(out),
             V, _stream())
    return out


def topk(x, k, aux=None):
    """x f32 [M,V] -> (vals [M,k], idx int32 [M,k], aux gathered [M,k] | None)"""
    M, V, ld = _rows(_chk(x, torch.float32))
    vals = torch.empty(M, k, device=x.device, dtype=torch.float32)
    idx = torch.empty(M, k, device=x.device, dtype=torch.int32)
    aux_out = torch.empty(M, k, device=x.device, dtype=torch.float32) if aux is not None else None
    lib.call("emoasr_topk", M, V, k, _p(x), ld, _p(aux), 0 if aux is None else aux.stride(0), _p(vals), _p(idx),
             _p(aux_out), _stream())
    return vals, idx, aux_out


def ctc_prefix_init(x, blank):
    T, V = x.shape
    r = torch.empty(T, 2, device=x.device, dtype=torch.float32)
    lib.call("emoasr_ctc_prefix_init", T, V, _p(_chk(x, torch.float32)), blank, _p(r), _stream())
    return r


def ctc_prefix_score(x, cands, last, out_len, blank, eos, prev_states=None, parent=None, pcand=None, init_state=None):
    """x f32 [T,V]; cands int32 [nb,cw]; last/out_len int32 [nb] -> (log_psi [nb,cw], states [nb,cw,T,2])"""
    T, V = x.shape
    nb, cw = cands.shape
    log_psi = torch.empty(nb, cw, device=x.device, dtype=torch.float32)
    states = torch.empty(nb, cw, T, 2, device=x.device, dtype=torch.float32)
    cw_prev = 0 if prev_states is None else prev_states.shape[1]
    lib.call("emoasr_ctc_prefix_score", nb, T, V, cw, _p(x), _p(prev_states), cw_prev, _p(parent), _p(pcand),
             _p(init_state), _p(last), _p(out_len), _p(cands), blank, eos, _p(log_psi), _p(states), _stream())
    return log_psi, states


# ---- RNN-T -------------------------------------------------------------------------------------
DACT_TANH_OUT = 5
DACT_MUL = 6            # epilogue dact: multiply by the saved factor (ACT_SAVE_DACT forward)
ACT_SAVE_DACT = 0x100   # epilogue act flag: pre_out <- act'(pre) * dropout_scale


def lstm_cell_fwd(gates_pre, c_prev, h_out, c_out, gates_act):
    """gates_pre [B,4H]; c_prev f32 [B,H] | None; h_out: [B,H] view (row stride allowed)"""
    B, H4 = gates_pre.shape
    lib.call("emoasr_lstm_cell_fwd", dt(gates_pre), B, H4 // 4, _p(gates_pre), _p(c_prev), _p(h_out), h_out.stride(0),
             _p(c_out), _p(gates_act), _stream())


def lstm_seq_supported(x, B, H):
    """does the cooperative whole-sequence recurrence (csrc/lstm_coop.hip) take this layer?"""
    return lib.size_query("emoasr_lstm_seq_supported", dt(x), B, H) == 1


def lstm_seq_fwd(pre, w_hh, h0, c0, hseq, cseq, gact):
    """pre [U,B,4H] (input projection + biases) -> hseq [U,B,H], cseq f32 [U,B,H], gact [U,B,4H] in one launch"""
    U, B, H4 = pre.shape
    lib.call("emoasr_lstm_seq_fwd", dt(pre), U, B, H4 // 4, _p(pre), _p(w_hh), _p(h0), _p(c0), _p(hseq), _p(cseq), _p(gact), _stream())


_lstm_ws = {}


def lstm_seq_bwd(dh_seq, gact, cseq, c0, w_hh, dgp):
    """dh_seq [U,B,H] (gradient w.r.t. the layer outputs) -> dgp [U,B,4H] (w.r.t. the gate pre-activations) in one launch"""
    U, B, H = dh_seq.shape
    need = lib.size_query("emoasr_lstm_seq_bwd_ws_bytes", B, H)
    key = (dh_seq.device, need)
    ws = _lstm_ws.get(key)
    if ws is None:
        ws = _lstm_ws[key] = torch.empty(need, device=dh_seq.device, dtype=torch.uint8)
    lib.call("emoasr_lstm_seq_bwd", dt(dh_seq), U, B, H, _p(dh_seq), _p(gact), _p(cseq), _p(c0), _p(w_hh), _p(dgp), _p(ws), need,
             _stream())


def lstm_cell_bwd(dh_out, dh_rec, dc, gates_act, c_prev, c, dgates_pre):
    B, H4 = gates_act.shape
    lib.call("emoasr_lstm_cell_bwd", dt(gates_act), B, H4 // 4, _p(dh_out), dh_out.stride(0), _p(dh_rec), _p(dc),
             _p(gates_act), _p(c_prev), _p(c), _p(dgates_pre), _stream())


def joint_tanh(e, g):
    """e [B,T,J], g [B,U,J] -> tanh(e[:, :, None] + g[:, None]) [B,T,U,J]"""
    B, T, J = e.shape
    U = g.shape[1]
    h = torch.empty(B, T, U, J, device=e.device, dtype=e.dtype)
    lib.call("emoasr_joint_tanh", dt(e), B, T, U, J, _p(e), _p(g), _p(h), _stream())
    return h


def joint_reduce(d):
    B, T, U, J = d.shape
    de = torch.empty(B, T, J, device=d.device, dtype=d.dtype)
    dg = torch.empty(B, U, J, device=d.device, dtype=d.dtype)
    lib.call("emoasr_joint_reduce", dt(d), B, T, U, J, _p(d), _p(de), _p(dg), _stream())
    return de, dg


def rnnt_forward(logits, labels, elens, ylens, blank):
    """logits [B,T,U,V]; labels int32 [B,Lmax] (U = Lmax+1) -> ctx tuple, nll f32 [B]"""
    B, T, U, V = logits.shape
    dev = logits.device
    f = lambda: torch.empty(B, T, U, device=dev, dtype=torch.float32)
    lse, lpb, lpy, alpha, beta = f(), f(), f(), f(), f()
    nll = torch.empty(B, device=dev, dtype=torch.float32)
    lib.call("emoasr_rnnt_forward", dt(logits), B, T, U, V, labels.shape[1], _p(logits), _p(labels), _p(elens), _p(ylens),
             blank, _p(lse), _p(lpb), _p(lpy), _p(alpha), _p(beta), _p(nll), _stream())
    return (lse, lpb, lpy, alpha, beta), nll


def rnnt_head_forward(h, w, bias, B, T, U, labels, elens, ylens, blank, lattice_stream=None):
    """the transducer's output layer + loss lattice WITHOUT the [B,T,U,V] logits: h [B*T*U, J] (bf16), w [V, J]
    -> ctx tuple (lse, lpb, lpy, alpha, beta) f32 [B,T,U], nll f32 [B]   (as rnnt_forward)
    lattice_stream: a torch side stream for the fold + lattice launches (2 B blocks of one wave row each: ~190 us of a mostly
    idle chip); -> (ctx, nll, event, keep) then -- the caller waits for `event` on its own stream before it reads ctx / nll and
    holds `keep` (scratch the side stream still reads) until then"""
    N, J = h.shape
    V = w.shape[0]
    dev = h.device
    nchunk = (V + 63) // 64
    part = torch.empty(nchunk, N, 2, device=dev, dtype=torch.float32)   # chunk-major: every launch fills its rows
    f = lambda: torch.empty(B, T, U, device=dev, dtype=torch.float32)
    lse, zb, zy, alpha, beta = f(), f(), f(), f(), f()
    zy.fill_(0.0)   # (cells beyond ylens never get a label logit)
    ycol = torch.empty(N, device=dev, dtype=torch.int32)
    lib.call("emoasr_rnnt_ycol", B, T, U, labels.shape[1], _p(labels), _p(ylens), _p(ycol), _stream())
    step = max(1, min(N, ((1 << 31) // (J * 2)) // 256 * 256))   # rows per launch: the operand stays below 4 GiB
    for r0 in range(0, N, step):
        n = min(step, N - r0)
        lib.call("emoasr_rnnt_head_fwd", dt(h), r0, n, T, U, V, J, labels.shape[1], _p(h[r0:r0 + n]), _p(w), _p(bias), _p(labels),
                 _p(ylens), blank, _p(part), N, _p(zb.view(-1)[r0:r0 + n]), _p(zy.view(-1)[r0:r0 + n]), _p(ycol[r0:r0 + n]), _stream())
    nll = torch.empty(B, device=dev, dtype=torch.float32)
    if lattice_stream is not None:
        main = torch.cuda.current_stream()
        ev0 = torch.cuda.Event()
        ev0.record(main)
        lattice_stream.wait_event(ev0)
        lib.call("emoasr_rnnt_forward_parts", B, T, U, V, _p(part), _p(elens), _p(ylens), _p(lse), _p(zb), _p(zy), _p(alpha),
                 _p(beta), _p(nll), c_void_p(lattice_stream.cuda_stream))
        ev1 = torch.cuda.Event()
        ev1.record(lattice_stream)
        return (lse, zb, zy, alpha, beta), nll, ev1, (part, ycol)
    lib.call("emoasr_rnnt_forward_parts", B, T, U, V, _p(part), _p(elens), _p(ylens), _p(lse), _p(zb), _p(zy), _p(alpha),
             _p(beta), _p(nll), _stream())
    return (lse, zb, zy, alpha, beta), nll


def rnnt_coef(ctx, nll, labels, elens, ylens, gscale, gscale_dev=None):
    """per-cell constants of the output layer's gradient -> coef f32 [cells, 4], ycol int32 [cells]"""
    lse, lpb, lpy, alpha, beta = ctx
    B, T, U = lse.shape
    coef = torch.empty(B * T * U, 4, device=lse.device, dtype=torch.float32)
    ycol = torch.empty(B * T * U, device=lse.device, dtype=torch.int32)
    lib.call("emoasr_rnnt_coef", B, T, U, labels.shape[1], _p(lse), _p(lpb), _p(lpy), _p(alpha), _p(beta), _p(labels), _p(elens),
             _p(ylens), _p(nll), gscale, _p(gscale_dev), _p(coef), _p(ycol), _stream())
    return coef, ycol


def rnnt_head_grad(h, w, bias, coef, ycol, blank, out):
    """out [n, V] (bf16) = gradient rows of the output layer for the cells of h [n, J], logits recomputed"""
    n, J = h.shape
    V = w.shape[0]
    lib.call("emoasr_rnnt_head_grad", dt(h), n, V, J, _p(h), _p(w), _p(bias), _p(coef), _p(ycol), blank, _p(out), out.stride(0),
             _stream())
    return out


def rnnt_grad(logits, ctx, nll, labels, elens, ylens, blank, gscale, gscale_dev=None, out=None):
    B, T, U, V = logits.shape
    lse, lpb, lpy, alpha, beta = ctx
    dz = torch.empty_like(logits) if out is None else out
    lib.call("emoasr_rnnt_grad", dt(logits), B, T, U, V, labels.shape[1], _p(logits), _p(lse), _p(lpb), _p(lpy), _p(alpha),
             _p(beta), _p(labels), _p(elens), _p(ylens), _p(nll), blank, gscale, _p(gscale_dev), _p(dz), _stream())
    return dz


def argmax_rows(x):
    M, V, ld = _rows(_chk(x))
    out = torch.empty(M, device=x.device, dtype=torch.int32)
    lib.call("emoasr_argmax_rows", dt(x), M, V, _p(x), ld, _p(out), _stream())
    return out


def first_not_equal(x, value):
    """x int32 [n] -> int32 [2] = (first index with x != value or -1, x there or value)"""
    out = torch.empty(2, device=x.device, dtype=torch.int32)
    lib.call("emoasr_first_not_equal", x.numel(), _p(_chk(x, torch.int32)), int(value), _p(out), _stream())
    return out


# ---- optimizer -------------------------------------------------------------------------
def sqnorm(x, out):
    lib.call("emoasr_sqnorm", x.numel(), _p(_chk(x, torch.float32)), _p(out), _stream())


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, gnorm_sq=None, clip=0.0, grad_mult=1.0, skipped=None):
    lib.call("emoasr_adam_step_ex", p.numel(), _p(p), _p(g), _p(m), _p(v), lr, beta1, beta2, eps, weight_decay,
             step, _p(gnorm_sq), clip, grad_mult, _p(skipped), _stream())


# ---- features --------------------------------------------------------------------------
def specaug_apply(x, spans, nf, nt, xlens=None, fill=None):
    B, T, F = x.shape
    lib.call("emoasr_specaug_apply", B, T, F, _p(_chk(x, torch.float32)), _p(spans), nf, nt, _p(xlens), _p(fill),
             _stream())
    return x


def cmvn(x, mean, std):
    M = x.numel() // x.shape[-1]
    lib.call("emoasr_cmvn", M, x.shape[-1], _p(x), _p(mean), _p(std), _stream())
    return x


def fbank(wav, frame_len, frame_shift, n_fft, n_mel, preemph, window, mel_fb):
    n = wav.numel()
    T = 1 + (n - frame_len) // frame_shift if n >= frame_len else 0
    feats = torch.empty(T, n_mel, device=wav.device, dtype=torch.float32)
    lib.call("emoasr_fbank", _p(wav), n, frame_len, frame_shift, n_fft, n_mel, preemph, _p(window), _p(mel_fb),
             _p(feats), T, _stream())
    return feats


# ---- small-M decode-step kernels (csrc/rowlin.hip) -------------------------------------------------------------------
def rowlin(x, w, bias=None, act=ACT_NONE, ln_a=None, res=None, ln_r=None, out_f32=False, eps=1e-12):
    """y[M <= 16, N] = act(LN_a?(x) @ w^T + bias) (+ res | LN_r(res)); bf16; ln_a / ln_r = (gamma, beta)"""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    ga, ba = ln_a if ln_a is not None else (None, None)
    gr, br = ln_r if ln_r is not None else (None, None)
    lib.call("emoasr_rowlin", M, N, K, _p(_chk(x, torch.bfloat16)), x.stride(0), _p(ga), _p(ba), eps, _p(_chk(w, torch.bfloat16)),
             _p(bias), act, _p(res), 0 if res is None else res.stride(0), _p(gr), _p(br), eps, _p(y), int(out_f32), N, _stream())
    return y


def attn_step(qkv, kcache, vcache, pos, H):
    """single-query attention of every row of qkv [nb, 3d] against its cache [nb, Lmax, d]; appends k, v at *pos (device int)"""
    nb, d3 = qkv.shape
    d = d3 // 3
    out = torch.empty(nb, d, device=qkv.device, dtype=qkv.dtype)
    lib.call("emoasr_attn_step", nb, d, H, kcache.shape[1], _p(_chk(qkv, torch.bfloat16)), _p(kcache), _p(vcache), _p(pos), _p(out),
             _stream())
    return out
