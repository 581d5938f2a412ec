// Host-side sequencing of one Conformer encoder layer (forward) in C++: the same kernels, in the same
// order and with the same arguments as emoasr_amd/engine.py:_layer_fwd issues them one FFI call at a
// time -- so results are bit-identical -- but a single crossing of the C ABI per layer.
// Reference: ConformerEncoderLayer.forward, asr/modeling/conformer.py:146-225 (macaron FFN, rel-pos
// MHA, convolution module, FFN, final LayerNorm; residual scales 0.5 / 1 / 1 / 0.5).
#include <math.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
#include "common.h"
#include "../../include/emoasr_hip.h"

void emo_attn_bwd_fused_extras(float* zero, long zn, const float* cast_src, void* cast_dst, long cn);  // csrc/attention.hip
void emo_attn_bwd_defer_join(int v);
void emo_attn_bwd_join(void* stream);
int emo_attn_bwd_prelaunch(int dtype, const emoasr_attn_t* a, void* ws, size_t ws_bytes, float* zero, long zn, void* stream);

// csrc/convmodule.hip: the BatchNorm kernels over ALL stacked micro-batches in one launch each (any dtype)
int emo_bn_stats_finalize_seg(const RowSegs& sg, int C, const float* part, float* mean, float* var, float* running_mean,
                              float* running_var, float momentum, long long* nbt, hipStream_t s);
int emo_bn_swish_fwd_seg_dt(int dtype, const RowSegs& sg, int C, const void* y, const float* mean, const float* var,
                            const float* gamma, const float* beta, float eps, void* z, hipStream_t s);
int emo_bn_swish_bwd_sums_seg_dt(int dtype, const RowSegs& sg, int C, const void* dz, const void* y, const float* mean, const float* var,
                                 const float* gamma, const float* beta, float eps, float* dgamma, float* dbeta, float* scratch,
                                 float** tot_out, hipStream_t s);
int emo_bn_bwd_apply_seg(int dtype, const RowSegs& sg, int C, const void* dz, const void* y, const float* mean, const float* var,
                         const float* gamma, const float* beta, float eps, const float* tot, void* dy, hipStream_t s);
int emo_dwconv_seg(int dtype, const RowSegs& sg, int tmax, int C, int K, const void* x, const float* w, const float* bias, void* y,
                   int flip, float* part, hipStream_t s);
int emo_dwconv_bwd_w_seg(int dtype, const RowSegs& sg, int tmax, int C, int K, const void* dy, const void* x, float* dw, float* dbias,
                         float* scratch, hipStream_t s);

namespace {

emoasr_epilogue_t plain_ep() {
  emoasr_epilogue_t e{};
  e.alpha = 1.f;
  e.res_scale = 1.f;
  return e;
}

// x + res_scale * drop(W2 act(W1 LN(x) + b1) + b2): LayerNorm + two products.  (The block as ONE launch -- 64 rows per workgroup, the
// F-wide intermediate consumed on chip -- was built in round 2 and measured again at the stacked size in round 4: 263 us against
// 120 us at M = 35 145, 90 against 36 at M = 7 029; every CU streams both weight matrices for its 64 rows.  Deleted.)
int g_att_bits = 1;       // option "attn_mask_bits": the layer hashes its attention keep mask once, as bits, in the forward
int g_ffn_save_dact = 1;  // option "ffn_save_dact": the feed-forward blocks save Swish'(u) * dropout_scale instead of u (common.h:
                          // EMO_ACT_SAVE_DACT); must not change between a forward and its backward
int g_stack_launch = 1;  // stacked micro-batches: 1 = the per-utterance kernels take all segments in ONE launch (segment table in
                         // their arguments), 0 = one launch per segment (same arithmetic; A/B switch, option "stack_launch")
int g_conv_fused = 1;  // bf16: the fused convolution-module kernels of csrc/convfused.hip (bit-identical to the separate launches)
bool conv_fused_ok(int dtype, int d) { return g_conv_fused && dtype == EMO_BF16 && d % 8 == 0; }

// Option "wgrad_side": the layer backward's grouped weight-gradient launch (nine products, ~0.2 ms at the stacked row count, 0.7
// rounds of workgroups) goes to a SIDE stream and runs under the NEXT layer's gradient chain, most of whose launches leave CUs
// idle; nothing on that chain reads a weight gradient.  One side stream + two completion events per (device, caller stream).  The
// launch reads the layer's workspace, so callers alternate between TWO workspaces from call to call (emoasr_amd/layer_rt.py); a
// call first waits for the launch issued two calls earlier (the last reader of its workspace).  emoasr_wgrad_side_join makes
// the caller's stream wait for the launches still in flight (before the optimizer, a gradient hook or a workspace release).
int g_wgrad_side = 0;
struct WgSide {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, done[2] = {nullptr, nullptr};
  bool pending[2] = {false, false};
  unsigned n = 0;   // launches so far: launch k signals done[k & 1]
};
std::mutex g_wg_mu;
std::map<std::pair<int, hipStream_t>, WgSide> g_wg;

WgSide* wg_side_of(hipStream_t s, bool create) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;   // (captured launches stay inline)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_wg_mu);
  auto it = g_wg.find({dev, s});
  if (it != g_wg.end()) return &it->second;
  if (!create) return nullptr;
  WgSide w;
  if (hipStreamCreateWithFlags(&w.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&w.fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&w.done[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&w.done[1], hipEventDisableTiming) != hipSuccess)
    return nullptr;
  return &(g_wg[{dev, s}] = w);
}

int ffn_fwd(int dtype, int M, int d, int F, const emoasr_ffn_params_t& p, const void* x, float res_scale,
            float p_enc, uint64_t s_in, uint64_t s_out, const emoasr_ffn_stash_t& st, void* stream) {
  if (emoasr_layernorm_fwd(dtype, M, d, x, p.ln_g, p.ln_b, 1e-5f, st.h, st.mean, st.rstd, stream)) return 1;
  emoasr_epilogue_t e1 = plain_ep();
  e1.bias = p.b1; e1.act = EMOASR_ACT_SWISH | (g_ffn_save_dact ? EMOASR_ACT_SAVE_DACT : 0); e1.pre_out = st.u; e1.drop_p = p_enc; e1.seed = s_in;
  if (emoasr_gemm_nt(dtype, M, F, d, st.h, d, p.w1, d, st.a, F, &e1, stream)) return 1;
  emoasr_epilogue_t e2 = plain_ep();
  e2.bias = p.b2; e2.residual = x; e2.ldr = d; e2.res_scale = res_scale; e2.drop_p = p_enc; e2.seed = s_out;
  return emoasr_gemm_nt(dtype, M, d, F, st.a, F, p.w2, F, st.y, d, &e2, stream);
}

// The stacked micro-batches of a call (include/emoasr_hip.h: emoasr_segments_t) with their offsets: first row, first row of
// the relative-position table, first utterance, first float of the BatchNorm partial-sum area.
struct SegView {
  int n = 0, Btot = 0;
  int B[EMOASR_MAX_SEGMENTS], T[EMOASR_MAX_SEGMENTS], b0[EMOASR_MAX_SEGMENTS];
  long row[EMOASR_MAX_SEGMENTS + 1], prow[EMOASR_MAX_SEGMENTS + 1], part[EMOASR_MAX_SEGMENTS + 1];
  long M() const { return row[n]; }
  long R() const { return prow[n]; }
};

bool seg_view(const emoasr_conformer_fwd_t* io, int d, SegView* v) {
  const emoasr_segments_t& s = io->seg;
  if (s.n < 0 || s.n > EMOASR_MAX_SEGMENTS) return false;
  v->n = s.n > 0 ? s.n : 1;
  v->row[0] = v->prow[0] = v->part[0] = 0;
  v->Btot = 0;
  for (int i = 0; i < v->n; ++i) {
    v->B[i] = s.n > 0 ? s.B[i] : io->B;
    v->T[i] = s.n > 0 ? s.T[i] : io->T;
    if (v->B[i] <= 0 || v->T[i] <= 0) return false;
    v->b0[i] = v->Btot;
    v->Btot += v->B[i];
    v->row[i + 1] = v->row[i] + (long)v->B[i] * v->T[i];
    v->prow[i + 1] = v->prow[i] + 2L * v->T[i] - 1;
    v->part[i + 1] = v->part[i] + emoasr_dwconv_stats_floats(v->B[i], v->T[i], d);
  }
  return true;
}

RowSegs row_segs_of(const SegView& sv, int C) {   // (as csrc/convfused.hip: row_segs)
  RowSegs sg{};
  sg.n = sv.n;
  for (int i = 0; i < sv.n; ++i) {
    sg.b0[i + 1] = sg.b0[i] + sv.B[i];
    sg.T[i] = sv.T[i];
    sg.row[i + 1] = sv.row[i + 1];
    sg.part[i + 1] = sv.part[i + 1];
    sg.sums[i + 1] = sg.sums[i] + (emoasr_bn_swish_bwd_scratch_floats(sv.B[i] * sv.T[i], C) / (2 * C) - 1);
  }
  return sg;
}

// attention arguments of segments [s0, s1) of a stacked pass (s1 - s0 == 1: a dense batch at the segment's offsets)
void attn_args_for(emoasr_attn_t& a, const SegView& sv, int s0, int s1, int H, int d, size_t esz, const void* qkv, const void* pp,
                   const int* klens, uint64_t seed) {
  const size_t ro = (size_t)sv.row[s0];
  const char* base = (const char*)qkv + ro * 3 * d * esz;
  int nb = 0, tmax = 0;
  for (int k = s0; k < s1; ++k) { nb += sv.B[k]; tmax = std::max(tmax, sv.T[k]); }
  a.B = nb; a.H = H; a.DK = d / H; a.Tq = tmax; a.Tk = tmax;
  a.ldq = a.ldk = a.ldv = 3 * d; a.ldo = d; a.ldp = d;
  a.q = base; a.k = base + (size_t)d * esz; a.v = base + (size_t)2 * d * esz;
  a.pos = (const char*)pp + (size_t)sv.prow[s0] * d * esz;
  a.klens = klens ? klens + sv.b0[s0] : nullptr;
  a.causal = 0; a.scale = 1.f / sqrtf((float)(d / H));
  a.seed = seed + 0x9E3779B97F4A7C15ull * (uint64_t)s0;   // (the mask index restarts in every segment)
  a.nseg = 0;
  if (s1 - s0 > 1) {
    a.nseg = s1 - s0;
    for (int k = s0; k <= s1; ++k) {
      a.seg_b0[k - s0] = k < s1 ? sv.b0[k] - sv.b0[s0] : nb;
      a.seg_row[k - s0] = sv.row[k] - sv.row[s0];
      a.seg_prow[k - s0] = sv.prow[k] - sv.prow[s0];
      if (k < s1) a.seg_T[k - s0] = sv.T[k];
    }
  }
}

}  // namespace

void emo_layer_set_wgrad_side(int v) { g_wgrad_side = v ? 1 : 0; }
extern "C" int emoasr_wgrad_side_join(int keep, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  WgSide* w = wg_side_of(s, false);
  if (!w) return 0;
  // launch n - 1 signals done[(n - 1) & 1] (the latest), launch n - 2 the other one
  for (int back = 2; back > (keep > 0 ? 1 : 0); --back) {
    if (w->n < (unsigned)back) continue;
    const int slot = (w->n - back) & 1;
    if (!w->pending[slot]) continue;
    if (hipStreamWaitEvent(s, w->done[slot], 0) != hipSuccess) { emo_set_error("wgrad_side_join: hipStreamWaitEvent failed"); return 1; }
    w->pending[slot] = false;
  }
  return 0;
}
void emo_layer_set_conv_fused(int v) { g_conv_fused = v; }
void emo_layer_set_stack_launch(int v) { g_stack_launch = v; }
void emo_layer_set_ffn_save_dact(int v) { g_ffn_save_dact = v ? 1 : 0; }
void emo_layer_set_att_bits(int v) { g_att_bits = v ? 1 : 0; }

extern "C" int emoasr_conformer_attn_masks(int dtype, int nl, const emoasr_segments_t* seg, int B, int T, int H, int d, const int* klens,
                                           float p_att, const uint64_t* seeds, unsigned* masks, long layer_stride_words, int nw,
                                           void* stream) {
  EMO_CHECK(seg && seeds && masks && nl >= 0 && H > 0 && d % H == 0, "conformer_attn_masks: bad arguments");
  if (p_att <= 0.f) return 0;
  emoasr_conformer_fwd_t io{};
  io.B = B; io.T = T; io.seg = *seg;
  SegView sv;
  EMO_CHECK(seg_view(&io, d, &sv), "conformer_attn_masks: bad batch / segment shapes");
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  const int astep = g_stack_launch ? sv.n : 1;
  for (int l = 0; l < nl; ++l)
    for (int si = 0; si < sv.n; si += astep) {
      emoasr_attn_t am{};
      attn_args_for(am, sv, si, si + astep, H, d, esz, nullptr, nullptr, klens, seeds[l]);
      am.drop_p = p_att;
      if (emoasr_attn_dropmask(dtype, &am, masks + (size_t)l * layer_stride_words + (size_t)sv.row[si] * H * nw, nw, stream)) return 1;
    }
  return 0;
}

extern "C" int emoasr_conformer_layer_fwd(int dtype, const emoasr_conformer_layer_t* L,
                                          const emoasr_conformer_fwd_t* io, void* stream) {
  EMO_CHECK(L && io && io->x && io->pos_t, "conformer_layer_fwd: missing arguments");
  const int d = L->d, H = L->H, F = L->F;
  EMO_CHECK(d > 0 && H > 0 && d % H == 0, "conformer_layer_fwd: bad dims d=%d H=%d", d, H);
  SegView sv;
  EMO_CHECK(seg_view(io, d, &sv), "conformer_layer_fwd: bad batch / segment shapes (B=%d T=%d segments=%d)", io->B, io->T, io->seg.n);
  const int M = (int)sv.M(), R = (int)sv.R();
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  const float p_enc = io->p_enc;
  // the attention-dropout keep mask of this layer as bits, hashed now (on the attention's side stream: it runs under the macaron
  // block's products) for the attention forward below and both passes of its backward
  const bool att_bits = g_att_bits && io->att_mask && io->training && io->p_att > 0.f && dtype == EMO_BF16;
  const int astep_m = g_stack_launch ? sv.n : 1;
  for (int si = 0; si < sv.n && att_bits && !io->att_mask_ready; si += astep_m) {
    emoasr_attn_t am{};
    attn_args_for(am, sv, si, si + astep_m, H, d, esz, io->qkv, io->pp, io->klens, io->seed[2]);
    am.drop_p = io->p_att;
    if (emoasr_attn_dropmask(dtype, &am, io->att_mask + (size_t)sv.row[si] * H * io->att_mask_nw, io->att_mask_nw, stream)) return 1;
  }
  // ---- macaron feed-forward ---------------------------------------------------------------------
  if (ffn_fwd(dtype, M, d, F, L->ffm, io->x, 0.5f, p_enc, io->seed[0], io->seed[1], io->ffm, stream)) return 1;
  const void* x1 = io->ffm.y;
  // ---- relative-position multi-head self-attention ----------------------------------------------
  if (emoasr_layernorm_fwd(dtype, M, d, x1, L->att_ln_g, L->att_ln_b, 1e-5f, io->at_h, io->at_mean, io->at_rstd, stream))
    return 1;
  {
    emoasr_epilogue_t e = plain_ep();
    e.bias = L->bqkv;
    if (emoasr_gemm_nt(dtype, M, 3 * d, d, io->at_h, d, L->wqkv, d, io->qkv, 3 * d, &e, stream)) return 1;
    emoasr_epilogue_t ep = plain_ep();
    if (emoasr_gemm_nt(dtype, R, d, d, io->pos_t, d, L->wpos, d, io->pp, d, &ep, stream)) return 1;
    // attention is per utterance: all stacked micro-batches in one launch (segment table in the arguments), or one each
    const int astep = g_stack_launch ? sv.n : 1;
    for (int si = 0; si < sv.n; si += astep) {
      const size_t ro = (size_t)sv.row[si];
      emoasr_attn_t a{};
      attn_args_for(a, sv, si, si + astep, H, d, esz, io->qkv, io->pp, io->klens, io->seed[2]);
      a.bias_u = L->bias_u; a.bias_v = L->bias_v; a.drop_p = io->p_att;
      a.out = (char*)io->o + ro * d * esz; a.lse = io->lse + ro * H;
      if (att_bits) { a.keep_mask = io->att_mask + ro * H * io->att_mask_nw; a.keep_nw = io->att_mask_nw; }
      if (emoasr_attn_fwd(dtype, &a, stream)) return 1;
    }
    emoasr_epilogue_t eo = plain_ep();
    eo.bias = L->bout; eo.residual = x1; eo.ldr = d; eo.drop_p = p_enc; eo.seed = io->seed[3];
    if (emoasr_gemm_nt(dtype, M, d, d, io->o, d, L->wout, d, io->at_y, d, &eo, stream)) return 1;
  }
  const void* x2 = io->at_y;
  // ---- convolution module -----------------------------------------------------------------------
  if (emoasr_layernorm_fwd(dtype, M, d, x2, L->cv_ln_g, L->cv_ln_b, 1e-5f, io->cv_h, io->cv_mean, io->cv_rstd, stream))
    return 1;
  {
    emoasr_epilogue_t e = plain_ep();
    e.bias = L->pw1_b;
    if (emoasr_gemm_nt(dtype, M, 2 * d, d, io->cv_h, d, L->pw1, d, io->g, 2 * d, &e, stream)) return 1;
    // bf16: GLU + depthwise convolution (+ BatchNorm partial statistics) in one launch, the GLU output never stored
    const bool fused = conv_fused_ok(dtype, d);
    if (!fused && emoasr_glu_fwd(dtype, M, d, io->g, io->gl, stream)) return 1;
    if (io->training) EMO_CHECK(io->bn_part && io->bmean && io->bvar, "conformer_layer_fwd: training needs the BatchNorm buffers");
    const bool conv_one_launch = fused && sv.n > 1 && g_stack_launch;   // all stacked micro-batches in one launch per kernel
    if (conv_one_launch &&
        emoasr_conv_module_fwd_seg(dtype, &io->seg, d, L->K, io->g, L->dw_w, L->dw_b, io->c, io->bn_part, io->bmean, io->bvar,
                                   L->bn_rm, L->bn_rv, 0.1f, L->bn_nbt, L->bn_g, L->bn_b, 1e-5f, io->z, io->training, stream))
      return 1;
    // f32 / f32x3 (the separate kernels), training: the depthwise convolution per micro-batch, then the BatchNorm statistics'
    // finalize and the BatchNorm + Swish apply over all micro-batches in ONE launch each
    const bool bn_one_launch = !fused && io->training && sv.n > 1 && g_stack_launch;
    if (bn_one_launch) {
      const RowSegs sg = row_segs_of(sv, d);
      int tmax = 0;
      for (int si = 0; si < sv.n; ++si) tmax = std::max(tmax, sv.T[si]);
      if (emo_dwconv_seg(dtype, sg, tmax, d, L->K, io->gl, L->dw_w, L->dw_b, io->c, 0, io->bn_part, (hipStream_t)stream)) return 1;
      if (emo_bn_stats_finalize_seg(sg, d, io->bn_part, io->bmean, io->bvar, L->bn_rm, L->bn_rv, 0.1f, L->bn_nbt, (hipStream_t)stream))
        return 1;
      if (emo_bn_swish_fwd_seg_dt(dtype, sg, d, io->c, io->bmean, io->bvar, L->bn_g, L->bn_b, 1e-5f, io->z, (hipStream_t)stream)) return 1;
    }
    for (int si = 0; si < sv.n && !conv_one_launch && !bn_one_launch; ++si) {   // each micro-batch: its own zero padding, its own batch statistics
      const size_t ro = (size_t)sv.row[si];
      const int B = sv.B[si], T = sv.T[si];
      const char* g = (const char*)io->g + ro * 2 * d * esz;
      const char* gl = io->gl ? (const char*)io->gl + ro * d * esz : nullptr;
      char* c = (char*)io->c + ro * d * esz;
      const float *bmean = L->bn_rm, *bvar = L->bn_rv;
      if (io->training) {
        float* part = io->bn_part + sv.part[si];
        if (fused) {
          if (emoasr_glu_dwconv_fwd(dtype, B, T, d, L->K, g, L->dw_w, L->dw_b, c, part, stream)) return 1;
        } else if (emoasr_dwconv_fwd_stats(dtype, B, T, d, L->K, gl, L->dw_w, L->dw_b, c, part, stream)) {
          return 1;
        }
        // (the running statistics move once per micro-batch, in order, as in the separate passes)
        if (emoasr_bn_stats_finalize(B, T, d, part, io->bmean + (size_t)si * d, io->bvar + (size_t)si * d, L->bn_rm, L->bn_rv, 0.1f,
                                     L->bn_nbt, stream))
          return 1;
        bmean = io->bmean + (size_t)si * d; bvar = io->bvar + (size_t)si * d;
      } else if (fused) {
        if (emoasr_glu_dwconv_fwd(dtype, B, T, d, L->K, g, L->dw_w, L->dw_b, c, nullptr, stream)) return 1;
      } else if (emoasr_dwconv_fwd(dtype, B, T, d, L->K, gl, L->dw_w, L->dw_b, c, stream)) {
        return 1;
      }
      if (emoasr_bn_swish_fwd(dtype, B * T, d, c, bmean, bvar, L->bn_g, L->bn_b, 1e-5f, (char*)io->z + ro * d * esz, stream)) return 1;
    }
    emoasr_epilogue_t eo = plain_ep();
    eo.bias = L->pw2_b; eo.residual = x2; eo.ldr = d; eo.drop_p = p_enc; eo.seed = io->seed[4];
    if (emoasr_gemm_nt(dtype, M, d, d, io->z, d, L->pw2, d, io->cv_y, d, &eo, stream)) return 1;
  }
  // ---- feed-forward, final LayerNorm ----------------------------------------------------------------
  if (ffn_fwd(dtype, M, d, F, L->ff, io->cv_y, 0.5f, p_enc, io->seed[5], io->seed[6], io->ff, stream)) return 1;
  return emoasr_layernorm_fwd(dtype, M, d, io->ff.y, L->fin_ln_g, L->fin_ln_b, 1e-5f, io->y, io->fin_mean, io->fin_rstd, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of one Conformer layer (relative positions) as ONE crossing of the C ABI: the gradient kernels in
// the order emoasr_amd/engine.py:_backward issues them (final LayerNorm, feed-forward, convolution module,
// self-attention, macaron feed-forward; reference autograd of conformer.py:146-225), the layer's nine weight-gradient
// products as one grouped launch, intermediates in one caller-provided workspace that the next layer reuses.
// The LayerNorm dgamma / dbeta partial sums are left in `ln_part` for one emoasr_layernorm_bwd_finalize over the
// whole sweep (slots: 0 final, 1 feed-forward, 2 convolution, 3 attention, 4 macaron).
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct BwdLayout {
  size_t off = 0;
  size_t take(size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; }
};

struct BwdBufs {
  size_t dx1, dx2, dx3, dx4, pre1, pre2, pre3, pre4, dh, du_ff, du_ffm, dz, dc, dgl, dg, dout, dqkv, dpos_t;
  size_t dpos, delta, bn_scr, dw_scr, attn_ws, attn_ws_bytes, total;
  size_t img[EMOASR_MAX_SEGMENTS + 1];   // f32: offsets of the segments' attention images in the caller's zero-filled area
};

BwdBufs bwd_layout(int dtype, const SegView& sv, int d, int H, int F, int K) {
  const size_t esz = dtype == EMO_BF16 ? 2 : 4, M = (size_t)sv.M(), R = (size_t)sv.R();
  BwdLayout L;
  BwdBufs b{};
  b.dx1 = L.take(M * d * esz); b.dx2 = L.take(M * d * esz); b.dx3 = L.take(M * d * esz); b.dx4 = L.take(M * d * esz);
  b.pre1 = L.take(M * d * esz); b.pre2 = L.take(M * d * esz); b.pre3 = L.take(M * d * esz); b.pre4 = L.take(M * d * esz);
  b.dh = L.take(M * d * esz);
  b.du_ff = L.take(M * F * esz); b.du_ffm = L.take(M * F * esz);
  b.dz = L.take(M * d * esz); b.dc = L.take(M * d * esz); b.dgl = L.take(M * d * esz); b.dg = L.take(M * 2 * d * esz);
  b.dout = L.take(M * d * esz); b.dqkv = L.take(M * 3 * d * esz); b.dpos_t = L.take(R * d * esz);
  b.dpos = L.take(R * d * 4); b.delta = L.take(M * H * 4);
  // per-segment scratch is reused by the segments one after the other: sized for the largest
  size_t bn_scr = 0, dw_scr = 0, attn_ws = 0;
  int tmax = 0;
  for (int i = 0; i < sv.n; ++i) {
    bn_scr = std::max(bn_scr, (size_t)emoasr_bn_swish_bwd_scratch_floats(sv.B[i] * sv.T[i], d) * 4);
    dw_scr = std::max(dw_scr, (size_t)emoasr_dwconv_bwd_w_scratch_floats(sv.B[i], sv.T[i], d, K) * 4);
    // bf16: the single-pass backward's workspace; f32: the plain scratch of the materialised backward (one segment at a time)
    attn_ws = std::max(attn_ws, dtype == EMO_BF16 ? emoasr_attn_bwd_fused_ws_bytes(dtype, sv.B[i], H, sv.T[i], sv.T[i], 1)
                                                  : emoasr_attn_bwd_mat_bytes(dtype, sv.B[i], H, sv.T[i], sv.T[i], 1, 1));
    tmax = std::max(tmax, sv.T[i]);
    b.img[i + 1] = b.img[i] + (dtype == EMO_BF16 ? 0 : emoasr_attn_bwd_mat_bytes(dtype, sv.B[i], H, sv.T[i], sv.T[i], 1, 0));
  }
  if (dtype == EMO_BF16)
    attn_ws = std::max(attn_ws, emoasr_attn_bwd_fused_ws_bytes_rows(dtype, (long)M, H, tmax, 1));   // all segments in one launch
  if (sv.n > 1) {
    emoasr_segments_t sg{};
    sg.n = sv.n;
    for (int i = 0; i < sv.n; ++i) { sg.B[i] = sv.B[i]; sg.T[i] = sv.T[i]; }
    bn_scr = std::max(bn_scr, (size_t)emoasr_conv_module_bwd_seg_scratch_floats(&sg, d, K, 0) * 4);
    dw_scr = std::max(dw_scr, (size_t)emoasr_conv_module_bwd_seg_scratch_floats(&sg, d, K, 1) * 4);
  }
  b.bn_scr = L.take(bn_scr);
  b.dw_scr = L.take(dw_scr);
  b.attn_ws_bytes = attn_ws;
  b.attn_ws = L.take(b.attn_ws_bytes);
  b.total = L.off;
  return b;
}

}  // namespace

extern "C" size_t emoasr_conformer_layer_bwd_ws_bytes_seg(int dtype, const emoasr_segments_t* seg, int d, int H, int F, int K) {
  emoasr_conformer_fwd_t io{};
  if (!seg) return 0;
  io.seg = *seg;
  SegView sv;
  if (seg->n < 1 || !seg_view(&io, d, &sv)) return 0;
  return bwd_layout(dtype, sv, d, H, F, K).total;
}

extern "C" size_t emoasr_conformer_layer_bwd_img_bytes_seg(int dtype, const emoasr_segments_t* seg, int d, int H) {
  emoasr_conformer_fwd_t io{};
  if (!seg || dtype == EMO_BF16) return 0;
  io.seg = *seg;
  SegView sv;
  if (seg->n < 1 || !seg_view(&io, d, &sv)) return 0;
  return bwd_layout(dtype, sv, d, H, 4 * d, 1).img[sv.n];
}

extern "C" size_t emoasr_conformer_layer_bwd_ws_bytes(int dtype, int B, int T, int d, int H, int F, int K) {
  emoasr_segments_t seg{};
  seg.n = 1; seg.B[0] = B; seg.T[0] = T;
  return emoasr_conformer_layer_bwd_ws_bytes_seg(dtype, &seg, d, H, F, K);
}

extern "C" int emoasr_conformer_layer_bwd(int dtype, const emoasr_conformer_layer_t* L, const emoasr_conformer_layer_t* G,
                                          const emoasr_conformer_fwd_t* st, const emoasr_conformer_bwd_t* io, void* stream) {
  EMO_CHECK(L && G && st && io && io->dy && io->dx && io->ws && io->ln_part, "conformer_layer_bwd: missing arguments");
  const int d = L->d, H = L->H, F = L->F, K = L->K;
  SegView sv;
  EMO_CHECK(seg_view(st, d, &sv), "conformer_layer_bwd: bad batch / segment shapes");
  const int M = (int)sv.M(), R = (int)sv.R();
  const BwdBufs bb = bwd_layout(dtype, sv, d, H, F, K);
  EMO_CHECK(io->ws_bytes >= bb.total, "conformer_layer_bwd: workspace %zu < %zu bytes", io->ws_bytes, bb.total);
  char* ws = static_cast<char*>(io->ws);
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  EMO_CHECK(dtype == EMO_BF16 || (io->attn_img && io->attn_img_bytes >= bb.img[sv.n]),
            "conformer_layer_bwd: f32 needs the zero-filled attention image area (%zu bytes, emoasr_conformer_layer_bwd_img_bytes_seg)",
            bb.img[sv.n]);
  const float p = st->p_enc;
  hipStream_t s = (hipStream_t)stream;
  emoasr_tn_problem_t pr[EMOASR_TN_GROUP_MAX];
  int npr = 0;
  WgSide* wgs = g_wgrad_side ? wg_side_of(s, true) : nullptr;
  if (wgs && wgs->pending[wgs->n & 1]) {   // the launch of two calls ago read THIS workspace (callers alternate between two)
    if (hipStreamWaitEvent(s, wgs->done[wgs->n & 1], 0) != hipSuccess) return 1;
    wgs->pending[wgs->n & 1] = false;
  }
  auto wgrad = [&](const void* dy, long lddy, int N1, const void* x, long ldx, int N2, int Kred, const void* gw, float alpha,
                   const void* gb) {
    pr[npr++] = emoasr_tn_problem_t{N1, N2, Kred, dy, lddy, x, ldx, (float*)gw, (long)N2, alpha, (float*)gb, alpha};
  };
  auto ln_bwd = [&](int slot, const void* dh, const void* x, const float* gamma, const float* mean, const float* rstd,
                    const void* dres, void* dx, const void* ggam, const void* gbet, void* pre, float scale, uint64_t seed) {
    emoasr_ln_bwd_opts_t o{};
    if (pre && p > 0.f) { o.dy2 = pre; o.scale2 = scale; o.drop_p2 = p; o.seed2 = seed; }
    o.defer_finalize = 1;
    return emoasr_layernorm_bwd_ex(dtype, M, d, dh, x, gamma, mean, rstd, dres, dx, (float*)ggam, (float*)gbet,
                                   io->ln_part + (long)slot * io->ln_part_stride, &o, stream);
  };
  // x + res_scale * drop(W2 act(W1 LN(x))): gradient chain of one feed-forward block.  dx_in: gradient at the block's
  // output; pre_in: dropout(dx_in * res_scale) from the LayerNorm backward that produced dx_in (p > 0)
  // dX = dY . W for W [n_out, d]: as the NT product against the transposed copy Wt [d, n_out] when the layer carries one (a long
  // reduction onto one 256-column tile: the large-tile kernel), else the NN product on the 64 x 64 kernel
  auto dgrad = [&](int n_out, const void* dy_, const void* W, const void* Wt, void* dx_) {
    emoasr_epilogue_t e_ = plain_ep();
    if (Wt) return emoasr_gemm_nt(dtype, M, d, n_out, dy_, n_out, Wt, n_out, dx_, d, &e_, stream);
    return emoasr_gemm_nn(dtype, M, d, n_out, dy_, n_out, W, d, dx_, d, &e_, stream);
  };
  auto ffn_bwd = [&](const emoasr_ffn_params_t& P, const emoasr_ffn_params_t& Gp, const emoasr_ffn_stash_t& S, const void* x_in,
                     const void* dx_in, const void* pre_in, void* du, uint64_t s_in, const void* w1t, const void* w2t) {
    const void* dy = p > 0.f ? pre_in : dx_in;
    const float alpha = p > 0.f ? 1.f : 0.5f;
    wgrad(dy, d, d, S.a, F, F, M, Gp.w2, alpha, Gp.b2);
    emoasr_epilogue_t e = plain_ep();
    e.alpha = alpha; e.dact_pre = S.u; e.seed = s_in;
    if (g_ffn_save_dact) { e.dact = EMOASR_DACT_MUL; e.drop_p = 0.f; }   // S.u holds Swish'(u) * dropout_scale (ffn_fwd)
    else { e.dact = EMOASR_ACT_SWISH; e.drop_p = p; }
    // du = (dy . W2) * factor: against the transposed copy W2^T [F, d] as an NT product when the layer carries one
    if (w2t ? emoasr_gemm_nt(dtype, M, F, d, dy, d, w2t, d, du, F, &e, stream)
            : emoasr_gemm_nn(dtype, M, F, d, dy, d, P.w2, F, du, F, &e, stream)) return 1;
    wgrad(du, F, F, S.h, d, d, M, Gp.w1, 1.f, Gp.b1);
    return dgrad(F, du, P.w1, w1t, ws + bb.dh);
  };

  // What the attention backward needs of FORWARD data only -- the dropout keep mask, the dense Q + pos_bias copies, the cleared
  // position-table gradient -- is prepared now, on the attention's side stream, under the feed-forward / convolution backward
  // that comes first (csrc/attention.hip: emo_attn_bwd_prelaunch)
  if (g_stack_launch && dtype == EMO_BF16) {
    emoasr_attn_t am{};
    attn_args_for(am, sv, 0, sv.n, H, d, esz, st->qkv, st->pp, st->klens, st->seed[2]);
    am.bias_u = L->bias_u; am.bias_v = L->bias_v; am.drop_p = st->p_att;
    if (g_att_bits && st->att_mask && st->p_att > 0.f) { am.keep_mask = st->att_mask; am.keep_nw = st->att_mask_nw; }
    if (emo_attn_bwd_prelaunch(dtype, &am, ws + bb.attn_ws, bb.attn_ws_bytes, (float*)(ws + bb.dpos), (long)sv.R() * d, stream)) return 1;
  }
  // ---- final LayerNorm ------------------------------------------------------------------------------------------
  if (ln_bwd(0, io->dy, st->ff.y, L->fin_ln_g, st->fin_mean, st->fin_rstd, nullptr, ws + bb.dx1, G->fin_ln_g, G->fin_ln_b,
             ws + bb.pre1, 0.5f, st->seed[6])) return 1;
  // ---- feed-forward -----------------------------------------------------------------------------------------------
  if (ffn_bwd(L->ff, G->ff, st->ff, st->cv_y, ws + bb.dx1, ws + bb.pre1, ws + bb.du_ff, st->seed[5], L->ff_w1t, L->ff_w2t)) return 1;
  if (ln_bwd(1, ws + bb.dh, st->cv_y, L->ff.ln_g, st->ff.mean, st->ff.rstd, ws + bb.dx1, ws + bb.dx2, G->ff.ln_g, G->ff.ln_b,
             ws + bb.pre2, 1.f, st->seed[4])) return 1;
  // ---- convolution module -----------------------------------------------------------------------------------------
  {
    const void* dy = p > 0.f ? ws + bb.pre2 : ws + bb.dx2;
    wgrad(dy, d, d, st->z, d, d, M, G->pw2, 1.f, G->pw2_b);
    emoasr_epilogue_t e = plain_ep();
    if (L->pw2_t ? emoasr_gemm_nt(dtype, M, d, d, dy, d, L->pw2_t, d, ws + bb.dz, d, &e, stream)
                 : emoasr_gemm_nn(dtype, M, d, d, dy, d, L->pw2, d, ws + bb.dz, d, &e, stream)) return 1;
    if (conv_fused_ok(dtype, d)) {
      // BatchNorm sums + fold, then ONE launch for BatchNorm/Swish apply -> depthwise data gradient -> GLU backward and the
      // depthwise weight-gradient partials (the GLU output is recomputed from g)
      const bool conv_one_launch = sv.n > 1 && g_stack_launch;   // all stacked micro-batches in one launch per kernel
      if (conv_one_launch &&
          emoasr_conv_module_bwd_seg(dtype, &st->seg, d, K, ws + bb.dz, st->c, st->bmean, st->bvar, L->bn_g, L->bn_b, 1e-5f,
                                     (float*)G->bn_g, (float*)G->bn_b, st->g, L->dw_w, ws + bb.dg, (float*)G->dw_w, (float*)G->dw_b,
                                     (float*)(ws + bb.bn_scr), (float*)(ws + bb.dw_scr), stream))
        return 1;
      for (int si = 0; si < sv.n && !conv_one_launch; ++si) {   // per micro-batch: own statistics / padding; scratch reused in order
        const size_t ro = (size_t)sv.row[si];
        const float *bmean = st->bmean + (size_t)si * d, *bvar = st->bvar + (size_t)si * d;
        const char* c = (const char*)st->c + ro * d * esz;
        float* tot = nullptr;
        if (emoasr_bn_swish_bwd_sums(dtype, sv.B[si] * sv.T[si], d, ws + bb.dz + ro * d * esz, c, bmean, bvar, L->bn_g, L->bn_b, 1e-5f,
                                     (float*)G->bn_g, (float*)G->bn_b, (float*)(ws + bb.bn_scr), &tot, stream)) return 1;
        if (emoasr_conv_bwd_fused(dtype, sv.B[si], sv.T[si], d, K, ws + bb.dz + ro * d * esz, c, bmean, bvar, L->bn_g, L->bn_b, 1e-5f,
                                  tot, (const char*)st->g + ro * 2 * d * esz, L->dw_w, ws + bb.dg + ro * 2 * d * esz, (float*)G->dw_w,
                                  (float*)G->dw_b, (float*)(ws + bb.dw_scr), stream)) return 1;
      }
    } else {
      // f32 / f32x3: the BatchNorm gradient sums + fold of all micro-batches in one launch each, then per micro-batch the apply
      // pass and the depthwise convolution's two gradients
      const bool bn_one_launch = sv.n > 1 && g_stack_launch;
      if (bn_one_launch) {   // every kernel of the chain over all micro-batches in one launch
        const RowSegs sg = row_segs_of(sv, d);
        int tmax = 0;
        for (int si = 0; si < sv.n; ++si) tmax = std::max(tmax, sv.T[si]);
        float* tot = nullptr;
        if (emo_bn_swish_bwd_sums_seg_dt(dtype, sg, d, ws + bb.dz, st->c, st->bmean, st->bvar, L->bn_g, L->bn_b, 1e-5f, (float*)G->bn_g,
                                         (float*)G->bn_b, (float*)(ws + bb.bn_scr), &tot, s)) return 1;
        if (emo_bn_bwd_apply_seg(dtype, sg, d, ws + bb.dz, st->c, st->bmean, st->bvar, L->bn_g, L->bn_b, 1e-5f, tot, ws + bb.dc, s)) return 1;
        if (emo_dwconv_seg(dtype, sg, tmax, d, K, ws + bb.dc, L->dw_w, nullptr, ws + bb.dgl, 1, nullptr, s)) return 1;
        if (emo_dwconv_bwd_w_seg(dtype, sg, tmax, d, K, ws + bb.dc, st->gl, (float*)G->dw_w, (float*)G->dw_b, (float*)(ws + bb.dw_scr), s))
          return 1;
      }
      for (int si = 0; si < sv.n && !bn_one_launch; ++si) {
        const size_t ro = (size_t)sv.row[si];
        const int B = sv.B[si], T = sv.T[si];
        const float *bmean = st->bmean + (size_t)si * d, *bvar = st->bvar + (size_t)si * d;
        if (emoasr_bn_swish_bwd(dtype, B * T, d, ws + bb.dz + ro * d * esz, (const char*)st->c + ro * d * esz, bmean, bvar, L->bn_g,
                                L->bn_b, 1e-5f, ws + bb.dc + ro * d * esz, (float*)G->bn_g, (float*)G->bn_b,
                                (float*)(ws + bb.bn_scr), stream)) return 1;
        if (emoasr_dwconv_bwd_x(dtype, B, T, d, K, ws + bb.dc + ro * d * esz, L->dw_w, ws + bb.dgl + ro * d * esz, stream)) return 1;
        if (emoasr_dwconv_bwd_w(dtype, B, T, d, K, ws + bb.dc + ro * d * esz, (const char*)st->gl + ro * d * esz, (float*)G->dw_w,
                                (float*)G->dw_b, 1, (float*)(ws + bb.dw_scr), stream)) return 1;
      }
      if (emoasr_glu_bwd(dtype, M, d, st->g, ws + bb.dgl, ws + bb.dg, stream)) return 1;
    }
    wgrad(ws + bb.dg, 2 * d, 2 * d, st->cv_h, d, d, M, G->pw1, 1.f, G->pw1_b);
    if (dgrad(2 * d, ws + bb.dg, L->pw1, L->pw1_t, ws + bb.dh)) return 1;
    if (ln_bwd(2, ws + bb.dh, st->at_y, L->cv_ln_g, st->cv_mean, st->cv_rstd, ws + bb.dx2, ws + bb.dx3, G->cv_ln_g, G->cv_ln_b,
               ws + bb.pre3, 1.f, st->seed[3])) return 1;
  }
  // ---- relative-position self-attention -----------------------------------------------------------------------------
  {
    const void* dy = p > 0.f ? ws + bb.pre3 : ws + bb.dx3;
    wgrad(dy, d, d, st->o, d, d, M, G->wout, 1.f, G->bout);
    emoasr_epilogue_t e = plain_ep();
    if (L->wout_t ? emoasr_gemm_nt(dtype, M, d, d, dy, d, L->wout_t, d, ws + bb.dout, d, &e, stream)
                  : emoasr_gemm_nn(dtype, M, d, d, dy, d, L->wout, d, ws + bb.dout, d, &e, stream)) return 1;
    // f32: the materialised backward (P^T / dS^T / dBD images + batched products, csrc/attention.hip: launch_bwd_tr), one
    // micro-batch at a time; the position-table gradient is f32 already
    if (dtype != EMO_BF16) {
      if (hipMemsetAsync(ws + bb.dpos, 0, (size_t)R * d * 4, s) != hipSuccess) return 1;
      for (int si = 0; si < sv.n; ++si) {
        const size_t ro = (size_t)sv.row[si], po = (size_t)sv.prow[si];
        char* dqkv = ws + bb.dqkv + ro * 3 * d * esz;
        emoasr_attn_t a{};
        attn_args_for(a, sv, si, si + 1, H, d, esz, st->qkv, st->pp, st->klens, st->seed[2]);
        a.bias_u = L->bias_u; a.bias_v = L->bias_v; a.drop_p = st->p_att;
        a.out = (char*)st->o + ro * d * esz; a.lse = st->lse + ro * H;
        a.dout = ws + bb.dout + ro * d * esz; a.delta = (float*)(ws + bb.delta) + ro * H;
        a.dq = dqkv; a.dk = dqkv + (size_t)d * esz; a.dv = dqkv + (size_t)2 * d * esz;
        a.dpos = (float*)(ws + bb.dpos) + po * d; a.dbias_u = (float*)G->bias_u; a.dbias_v = (float*)G->bias_v;
        if (emoasr_attn_bwd_mat_bind(dtype, &a, (char*)io->attn_img + bb.img[si], ws + bb.attn_ws)) return 1;
        if (emoasr_attn_bwd(dtype, &a, stream)) return 1;
      }
    }
    // bf16: all stacked micro-batches in one set of launches (segment table in the arguments), or one set each (workspace reused)
    const int astep = g_stack_launch ? sv.n : 1;
    for (int si = 0; si < sv.n && dtype == EMO_BF16; si += astep) {
      const size_t ro = (size_t)sv.row[si], po = (size_t)sv.prow[si];
      const long Rs = sv.prow[si + astep] - sv.prow[si];
      char* dqkv = ws + bb.dqkv + ro * 3 * d * esz;
      emoasr_attn_t a{};
      attn_args_for(a, sv, si, si + astep, H, d, esz, st->qkv, st->pp, st->klens, st->seed[2]);
      a.bias_u = L->bias_u; a.bias_v = L->bias_v; a.drop_p = st->p_att;
      a.out = (char*)st->o + ro * d * esz; a.lse = st->lse + ro * H;
      if (g_att_bits && st->att_mask && st->p_att > 0.f) { a.keep_mask = st->att_mask + ro * H * st->att_mask_nw; a.keep_nw = st->att_mask_nw; }
      a.dout = ws + bb.dout + ro * d * esz; a.delta = (float*)(ws + bb.delta) + ro * H;
      a.dq = dqkv; a.dk = dqkv + (size_t)d * esz; a.dv = dqkv + (size_t)2 * d * esz;
      float* dpos = (float*)(ws + bb.dpos) + po * d;
      a.dpos = dpos; a.dbias_u = (float*)G->bias_u; a.dbias_v = (float*)G->bias_v;
      // the position-table gradient is cleared by the attention backward's prologue launch and rounded to the compute dtype
      // by its finalize launch (two launches less per layer than a memset + a cast)
      emo_attn_bwd_fused_extras(dpos, Rs * d, dpos, ws + bb.dpos_t + po * d * esz, Rs * d);
      // the position-table gradient may stay on the attention's side stream until the weight gradients need it (joined below)
      emo_attn_bwd_defer_join(astep == sv.n);
      if (emoasr_attn_bwd_fused(dtype, &a, ws + bb.attn_ws, bb.attn_ws_bytes, stream)) return 1;
    }
    wgrad(dtype == EMO_BF16 ? ws + bb.dpos_t : ws + bb.dpos, d, d, st->pos_t, d, d, R, G->wpos, 1.f, nullptr);
    wgrad(ws + bb.dqkv, 3 * d, 3 * d, st->at_h, d, d, M, G->wqkv, 1.f, G->bqkv);
    if (dgrad(3 * d, ws + bb.dqkv, L->wqkv, L->wqkv_t, ws + bb.dh)) return 1;
    if (ln_bwd(3, ws + bb.dh, st->ffm.y, L->att_ln_g, st->at_mean, st->at_rstd, ws + bb.dx3, ws + bb.dx4, G->att_ln_g,
               G->att_ln_b, ws + bb.pre4, 0.5f, st->seed[1])) return 1;
  }
  // ---- macaron feed-forward ---------------------------------------------------------------------------------------
  if (ffn_bwd(L->ffm, G->ffm, st->ffm, st->x, ws + bb.dx4, ws + bb.pre4, ws + bb.du_ffm, st->seed[0], L->ffm_w1t, L->ffm_w2t)) return 1;
  if (ln_bwd(4, ws + bb.dh, st->x, L->ffm.ln_g, st->ffm.mean, st->ffm.rstd, ws + bb.dx4, io->dx, G->ffm.ln_g, G->ffm.ln_b,
             nullptr, 0.f, 0)) return 1;
  // ---- the layer's weight gradients, one launch -----------------------------------------------------------------------
  emo_attn_bwd_join(stream);
  if (!wgs) return emoasr_gemm_tn_grouped(dtype, npr, pr, stream);
  // fork: everything this call put on the caller's stream precedes the side launch; its completion is an event of its own
  const int slot = wgs->n & 1;
  if (hipEventRecord(wgs->fork, s) != hipSuccess || hipStreamWaitEvent(wgs->side, wgs->fork, 0) != hipSuccess) {
    emo_set_error("conformer_layer_bwd: fork to the weight-gradient stream failed");
    return 1;
  }
  if (emoasr_gemm_tn_grouped(dtype, npr, pr, wgs->side)) return 1;
  if (hipEventRecord(wgs->done[slot], wgs->side) != hipSuccess) { emo_set_error("conformer_layer_bwd: hipEventRecord failed"); return 1; }
  wgs->pending[slot] = true;
  ++wgs->n;
  return 0;
}
