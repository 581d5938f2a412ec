// Host-side sequencing of one Conformer encoder layer (forward) in C++: the same kernels, in the same
// order and with the same arguments as emoasr_amd/engine.py:_layer_fwd issues them one FFI call at a
// time -- so results are bit-identical -- but a single crossing of the C ABI per layer.
// Reference: ConformerEncoderLayer.forward, asr/modeling/conformer.py:146-225 (macaron FFN, rel-pos
// MHA, convolution module, FFN, final LayerNorm; residual scales 0.5 / 1 / 1 / 0.5).
#include <math.h>
#include "common.h"
#include "../../include/emoasr_hip.h"

namespace {

emoasr_epilogue_t plain_ep() {
  emoasr_epilogue_t e{};
  e.alpha = 1.f;
  e.res_scale = 1.f;
  return e;
}

// x + res_scale * drop(W2 act(W1 LN(x) + b1) + b2)
int ffn_fwd(int dtype, int M, int d, int F, const emoasr_ffn_params_t& p, const void* x, float res_scale,
            float p_enc, uint64_t s_in, uint64_t s_out, const emoasr_ffn_stash_t& st, void* stream) {
  if (emoasr_layernorm_fwd(dtype, M, d, x, p.ln_g, p.ln_b, 1e-5f, st.h, st.mean, st.rstd, stream)) return 1;
  emoasr_epilogue_t e1 = plain_ep();
  e1.bias = p.b1; e1.act = EMOASR_ACT_SWISH; e1.pre_out = st.u; e1.drop_p = p_enc; e1.seed = s_in;
  if (emoasr_gemm_nt(dtype, M, F, d, st.h, d, p.w1, d, st.a, F, &e1, stream)) return 1;
  emoasr_epilogue_t e2 = plain_ep();
  e2.bias = p.b2; e2.residual = x; e2.ldr = d; e2.res_scale = res_scale; e2.drop_p = p_enc; e2.seed = s_out;
  return emoasr_gemm_nt(dtype, M, d, F, st.a, F, p.w2, F, st.y, d, &e2, stream);
}

}  // namespace

extern "C" int emoasr_conformer_layer_fwd(int dtype, const emoasr_conformer_layer_t* L,
                                          const emoasr_conformer_fwd_t* io, void* stream) {
  EMO_CHECK(L && io && io->x && io->pos_t, "conformer_layer_fwd: missing arguments");
  const int d = L->d, H = L->H, F = L->F, B = io->B, T = io->T, M = B * T;
  EMO_CHECK(d > 0 && H > 0 && d % H == 0 && M > 0, "conformer_layer_fwd: bad dims d=%d H=%d B=%d T=%d", d, H, B, T);
  const size_t esz = dtype == EMO_BF16 ? 2 : 4;
  const float p_enc = io->p_enc;
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
    if (emoasr_gemm_nt(dtype, 2 * T - 1, d, d, io->pos_t, d, L->wpos, d, io->pp, d, &ep, stream)) return 1;
    emoasr_attn_t a{};
    a.B = B; a.H = H; a.DK = d / H; a.Tq = T; a.Tk = T;
    a.ldq = a.ldk = a.ldv = 3 * d; a.ldo = d; a.ldp = d;
    a.q = io->qkv;
    a.k = (const char*)io->qkv + (size_t)d * esz;
    a.v = (const char*)io->qkv + (size_t)2 * d * esz;
    a.pos = io->pp; a.bias_u = L->bias_u; a.bias_v = L->bias_v; a.klens = io->klens;
    a.causal = 0; a.scale = 1.f / sqrtf((float)(d / H)); a.drop_p = io->p_att; a.seed = io->seed[2];
    a.out = io->o; a.lse = io->lse;
    if (emoasr_attn_fwd(dtype, &a, stream)) return 1;
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
    if (emoasr_glu_fwd(dtype, M, d, io->g, io->gl, stream)) return 1;
    const float *bmean = L->bn_rm, *bvar = L->bn_rv;
    if (io->training) {
      EMO_CHECK(io->bn_part && io->bmean && io->bvar, "conformer_layer_fwd: training needs the BatchNorm buffers");
      if (emoasr_dwconv_fwd_stats(dtype, B, T, d, L->K, io->gl, L->dw_w, L->dw_b, io->c, io->bn_part, stream)) return 1;
      if (emoasr_bn_stats_finalize(B, T, d, io->bn_part, io->bmean, io->bvar, L->bn_rm, L->bn_rv, 0.1f, L->bn_nbt, stream))
        return 1;
      bmean = io->bmean; bvar = io->bvar;
    } else if (emoasr_dwconv_fwd(dtype, B, T, d, L->K, io->gl, L->dw_w, L->dw_b, io->c, stream)) {
      return 1;
    }
    if (emoasr_bn_swish_fwd(dtype, M, d, io->c, bmean, bvar, L->bn_g, L->bn_b, 1e-5f, io->z, stream)) return 1;
    emoasr_epilogue_t eo = plain_ep();
    eo.bias = L->pw2_b; eo.residual = x2; eo.ldr = d; eo.drop_p = p_enc; eo.seed = io->seed[4];
    if (emoasr_gemm_nt(dtype, M, d, d, io->z, d, L->pw2, d, io->cv_y, d, &eo, stream)) return 1;
  }
  // ---- feed-forward, final LayerNorm ----------------------------------------------------------------
  if (ffn_fwd(dtype, M, d, F, L->ff, io->cv_y, 0.5f, p_enc, io->seed[5], io->seed[6], io->ff, stream)) return 1;
  return emoasr_layernorm_fwd(dtype, M, d, io->ff.y, L->fin_ln_g, L->fin_ln_b, 1e-5f, io->y, io->fin_mean, io->fin_rstd, stream);
}
