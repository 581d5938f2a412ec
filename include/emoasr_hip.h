/* emoasr_hip.h -- C ABI of libemoasr_hip.so, the MI355X (gfx950) device library
 * behind the emoASR-compatible Python modules in emoasr_amd/.
 *
 * The reference (emonosuke/emoASR) is pure Python on PyTorch: its "FFI" for the
 * hot path is the set of ATen / cuDNN calls made by asr/modeling/**.  Every
 * entry point below replaces one such call site (cited per function) with a
 * hand-written HIP kernel.  Signatures use plain device pointers, sizes and a
 * hipStream_t passed as void*: no torch types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error;
 *     emoasr_last_error() returns a message for the calling thread's last error.
 *   - dtype: EMOASR_F32 (exact mode: f32 MFMA, a bit-exact k-ordered fmaf chain), EMOASR_BF16 (bf16 MFMA,
 *     f32 accumulate) or EMOASR_F32X3 (the PARITY mode at throughput: f32 buffers, statistics and epilogues
 *     exactly as EMOASR_F32, every matrix product as three bf16 MFMAs over (hi, lo) operand pairs -- 16
 *     significand bits per operand; meets north_star's 1e-3 / bit-exact-ids tolerances, tests/test_fullsize_gpu.py).
 *     The mode is an argument of every call: the library keeps no precision state (rounds 4-5 had a
 *     process-wide option "f32_split"; it is gone).  Entry points without a matrix product treat EMOASR_F32X3
 *     as EMOASR_F32.  "T*" below means a buffer of the dtype's element type (float for both f32 codes).
 *   - statistics, losses, lattices, parameter gradients and optimizer state are
 *     always f32; lengths / labels are int32.
 *   - all pointers are device pointers unless noted; kernels are enqueued on
 *     `stream` and never synchronise the host.
 *   - activations are row-major [rows, features] ("channels-last").
 */
#ifndef EMOASR_HIP_H
#define EMOASR_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { EMOASR_F32 = 0, EMOASR_BF16 = 1, EMOASR_F32X3 = 2 };
enum { EMOASR_ACT_NONE = 0, EMOASR_ACT_RELU = 1, EMOASR_ACT_SWISH = 2,
       /* emoasr_epilogue_t.dact only: the tensor behind dact_pre IS the factor to multiply by (saved by a forward epilogue whose
        * act carried EMOASR_ACT_SAVE_DACT); pass drop_p = 0 with it, the mask is part of the factor */
       EMOASR_DACT_MUL = 6 };
/* emoasr_epilogue_t.act | EMOASR_ACT_SAVE_DACT: pre_out receives act'(pre) * dropout_scale (the factor of the data gradient)
 * instead of the pre-activation: the backward of x -> dropout(act(x W^T + b)) then needs neither the activation's derivative nor
 * a second hashing of the dropout mask (transformer.py:117-118's feed-forward blocks in training). */
#define EMOASR_ACT_SAVE_DACT 0x100

const char* emoasr_last_error(void);
int emoasr_version(void);
/* options: "tr_read" (1 = ds_read_b64_tr_b16 operand reads, 0 = scalar fallback); tuning: "gemm_tile", "gemm_kb",
 * "gemm_xcd", "gemm_wholek", "tn_group_blocks", "tn_group_kb", "tn_place", "attn_lpt", "attn_xcd", "attn_fwd_waves", "attn_fwd_split", "attn_fw" (key tiles per workgroup of the single-pass attention backward: 2, 4, 0 = auto);
 * round 4: "attn_bwd_split", "attn_side", "ffn_save_dact", "big_n256" (long reductions onto one / two 256-column tiles on the
 * large-tile kernel: 0 off, 1 = N 512 from K 512 and N 256 from K 2048, 2 = N 256 from K 512 as well), "big_waves" (8 / 4),
 * "big_bm", "big_min_tiles", "gemm_wide128", "ln_fwd8", "ln_bwd_pf", "ln_bwd_blocks", "conv1_pair", "lstm_coop", "decode_coop",
 * "rnnt_greedy_coop" -- every default is what the training step (or the decode leg) measured fastest;
 * "timers" (see emoasr_timer_read) */
int emoasr_set_option(const char* name, int value);
/* Device time of selected kernels that sit behind composite entry points, measured with HIP events on the launch stream
 * while emoasr_set_option("timers", 1) is in effect.  name: "attn_bwd_fused_kernel", "attn_bwd_dpos_kernel",
 * "attn_fwd_kernel", "gemm_tn_grouped_kernel".  -> number of launches recorded and their summed milliseconds
 * (synchronises on the recorded events); reset != 0 clears the record.  bench.py's roofline object reads this. */
int emoasr_timer_read(const char* name, int* calls, double* ms, int reset);
/* The same with the ALGORITHMIC work of the recorded launches (sum of 2 M N K flops / operand + result bytes as the entry point
 * was called; 0 where the library cannot know it).  Families: "gemm_nt_nn" (emoasr_gemm_nt + emoasr_gemm_nn, forward and
 * data-gradient products), "gemm_tn" (emoasr_gemm_tn + _grouped, weight gradients), "layernorm" (forward + backward),
 * "conv_module" (emoasr_conv_module_fwd_seg / _bwd_seg), and the kernels named above.  "timers" = 1 records every family, a
 * bit mask 1 << (index + 1) only the chosen ones (index = position in this list: attn_bwd_fused_kernel 0, attn_bwd_dpos3_kernel
 * 1, attn_fwd_kernel 2, gemm_tn_grouped_kernel 3, gemm_nt_nn 4, gemm_tn 5, layernorm 6, conv_module 7).  Option
 * "timer_stride" = n records every n-th launch of a family only (an event pair per launch costs device time). */
int emoasr_timer_read_ex(const char* name, int* calls, double* ms, double* flops, double* bytes, int reset);

/* GEMM epilogue, applied in this order to v = alpha*acc + bias[col]:
 *   pre_out[row,col] = v            (if pre_out)
 *   v = act(v)
 *   v *= dact'(dact_pre[row,col])   (if dact_pre; derivative of activation `dact`)
 *   v *= dropout(seed, row*N+col)   (if drop_p > 0; 0 or 1/(1-p))
 *   v = residual[row,col] + res_scale*v   (if residual)
 * pre_out / dact_pre share C's leading dimension; residual uses ldr. */
typedef struct {
  const float* bias;
  const void* residual;
  void* pre_out;
  const void* dact_pre;
  float alpha;
  float res_scale;
  int act;
  int dact;
  int ldr;
  int out_f32; /* C is float* regardless of dtype */
  float drop_p;
  uint64_t seed;
} emoasr_epilogue_t;

/* C[M,N] = epilogue(A[M,K] . B[N,K]^T).  Replaces nn.Linear / Conv1d(k=1):
 * asr/modeling/transformer.py:62-71,94,110-118; conformer.py:62,103-105,115-117;
 * encoders/conv.py:16-19; decoders/ctc.py:34,103. */
int emoasr_gemm_nt(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                   void* C, long ldc, const emoasr_epilogue_t* ep, void* stream);
/* C[M,N] = epilogue(A[M,K] . B[K,N]) with B stored k-major (row stride ldb): the data
 * gradient dX = dY . W of the calls above, straight from the [out,in] weight. */
int emoasr_gemm_nn(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                   void* C, long ldc, const emoasr_epilogue_t* ep, void* stream);
/* batched form: C[b,h] = alpha * A[b,h] . B[b,h]; s*_b / s*_h are the outer / inner batch strides
 * (elements).  Attention backward: dV[b,h] = Pd[b,h] . dO[b,h], dK[b,h] = dS[b,h] . Q[b,h]. */
int emoasr_gemm_nn_batched(int dtype, int M, int N, int K, const void* A, long lda, long sa_b, long sa_h,
                           const void* B, long ldb, long sb_b, long sb_h, void* C, long ldc, long sc_b,
                           long sc_h, int nb, int nh, float alpha, int accumulate, void* stream);
/* C[N1,N2] (+)= alpha * A[K,N1]^T . B[K,N2], f32 output (weight gradients; the
 * autograd backward of the calls above).  If colsum != NULL the kernel also produces
 * colsum[N1] (+)= colsum_scale * sum_k A[k,:] (the bias gradient) from the staged A tiles. */
int emoasr_gemm_tn(int dtype, int N1, int N2, int K, const void* A, long lda, const void* B, long ldb,
                   float* C, long ldc, float alpha, int accumulate, float* colsum, float colsum_scale,
                   void* stream);
/* Grouped form of emoasr_gemm_tn: n <= EMOASR_TN_GROUP_MAX independent, always-accumulating
 * products C_i += alpha_i * A_i^T . B_i (+ colsum_i) in ONE launch.  One encoder layer's backward
 * has ~10 weight gradients of 16..64 output tiles each (autograd would run them as 10 separate
 * addmm kernels); grouped they fill the chip. */
#define EMOASR_TN_GROUP_MAX 16
typedef struct emoasr_tn_problem {
  int N1, N2, K;
  const void* A; long lda;
  const void* B; long ldb;
  float* C; long ldc;
  float alpha;
  float* colsum;        /* optional bias gradient, see emoasr_gemm_tn */
  float colsum_scale;
} emoasr_tn_problem_t;
int emoasr_gemm_tn_grouped(int dtype, int n, const emoasr_tn_problem_t* probs, void* stream);
/* out[N] (+)= scale * sum_rows X[M,N]   (bias gradients) */
int emoasr_colsum(int dtype, int M, int N, const void* X, long ldx, float* out, float scale,
                  int accumulate, void* stream);

/* ---- Conv2d subsampling front-end (asr/modeling/encoders/conv.py:5-28) ------
 * x f32 [B,T,F] -> y1 T [B,T1,F1,C] (channels-last), T1=(T-3)/2+1, F1=(F-3)/2+1
 * w1 f32 [C,9], b1 f32 [C]; ReLU fused. */
int emoasr_conv1_fwd(int dtype, int B, int T, int F, int C, const float* x, const float* w1,
                     const float* b1, void* y1, void* stream);
/* dw1[C,9], db1[C] (+)= from dy1 (gradient w.r.t. the pre-ReLU conv1 output);
 * scratch: emoasr_conv1_wgrad_scratch_floats(B, T, C) floats of per-block partial sums */
int emoasr_conv1_wgrad(int dtype, int B, int T, int F, int C, const float* x, const void* dy1,
                       float* dw1, float* db1, int accumulate, float* scratch, void* stream);
long emoasr_conv1_wgrad_scratch_floats(int B, int T, int C);
/* y2[(b,t2,f2), n] = epilogue(sum y1[b,2t2+kh,2f2+kw,c] * w[n,(kh,kw,c)])  (implicit GEMM) */
int emoasr_conv2_fwd(int dtype, int B, int T1, int F1, int C, const void* y1, const void* w, void* y2,
                     const emoasr_epilogue_t* ep, void* stream);
/* dw[n,(kh,kw,c)] (+)= dy2^T . im2col(y1);  dbias[n] (+)= sum dy2 (may be NULL) */
int emoasr_conv2_wgrad(int dtype, int B, int T1, int F1, int C, const void* dy2, const void* y1,
                       float* dw, float* dbias, int accumulate, void* stream);
/* dy1[b,t1,f1,c] = relu'(y1) * sum_{kh,kw} dcol[(b,t2,f2),(kh,kw,c)]  (col2im gather) */
/* Data gradient of the same convolution without an im2col buffer (four implicit GEMMs, one per parity class of
 * the output position): dy1[b,t1,f1,c] = relu'(y1[b,t1,f1,c]) * sum_{kh,kw,n} dy2[b,(t1-kh)/2,(f1-kw)/2,n]*w[n,(kh,kw,c)];
 * w = the [C, 9C] layout of emoasr_conv2_fwd.  Replaces gemm_nn (dcol) + emoasr_conv2_col2im. */
int emoasr_conv2_dgrad(int dtype, int B, int T1, int F1, int C, const void* dy2, const void* w, const void* y1,
                       void* dy1, void* stream);
int emoasr_conv2_col2im(int dtype, int B, int T1, int F1, int C, const void* dcol, const void* y1,
                        void* dy1, void* stream);
/* The same data gradient on the large-tile kernel (csrc/gemm_big.hip; bf16, C % 256 == 0): all four parity classes in ONE
 * launch.  wt = the weight laid out [c][kh][kw][n] (conv.2.weight.permute(1,2,3,0)), so that every tap's reduction over the
 * output channels n is contiguous. */
int emoasr_conv2_dgrad_kc(int dtype, int B, int T1, int F1, int C, const void* dy2, const void* wt, const void* y1,
                          void* dy1, void* stream);
/* Large-tile NT product for long, wide shapes (bf16; N % 8 == 0, K % 64 == 0): C = relu?(A . B^T + bias).  The kernel
 * behind emoasr_conv2_fwd / _dgrad_kc on a plain row-major A (nn.Linear with >= 256 outputs over >= 10^5 rows). */
int emoasr_gemm_nt_big(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                       const float* bias, int relu, void* stream);

/* ---- LayerNorm (nn.LayerNorm call sites: conformer.py:183-187,
 * encoders/transformer.py:73, transformer.py:140-141,178-180) ------------------ */
int emoasr_layernorm_fwd(int dtype, int M, int N, const void* x, const float* gamma, const float* beta,
                         float eps, void* y, float* mean, float* rstd, void* stream);
/* dx = dres + LN'(dy); dgamma/dbeta (+)= ; dres may be NULL.  scratch: f32 workspace of
 * emoasr_layernorm_bwd_scratch_floats(N) floats (per-block partial sums; NULL if no dgamma/dbeta) */
int emoasr_layernorm_bwd(int dtype, int M, int N, const void* dy, const void* x, const float* gamma,
                         const float* mean, const float* rstd, const void* dres, void* dx,
                         float* dgamma, float* dbeta, float* scratch, void* stream);
long emoasr_layernorm_bwd_scratch_floats(int N);
/* Extended form.  dy2 (optional) = dropout(dx * scale2; drop_p2, seed2), computed from the stored dx:
 * the gradient entering the next residual branch x + scale2 * dropout(f(x)) of the backward sweep, so
 * that branch needs no separate emoasr_scale_dropout pass (same mask index: row * N + col).
 * defer_finalize != 0 leaves the dgamma / dbeta partial sums in `scratch` (which must then stay
 * untouched) for one emoasr_layernorm_bwd_finalize call that folds up to EMOASR_LN_FINALIZE_MAX
 * LayerNorms in a single launch. */
typedef struct emoasr_ln_bwd_opts {
  void* dy2;
  float scale2, drop_p2;
  uint64_t seed2;
  int defer_finalize;
} emoasr_ln_bwd_opts_t;
int emoasr_layernorm_bwd_ex(int dtype, int M, int N, const void* dy, const void* x, const float* gamma,
                            const float* mean, const float* rstd, const void* dres, void* dx,
                            float* dgamma, float* dbeta, float* scratch, const emoasr_ln_bwd_opts_t* opts,
                            void* stream);
#define EMOASR_LN_FINALIZE_MAX 64
typedef struct emoasr_ln_finalize_item {
  int M, N;              /* the M, N of the deferred emoasr_layernorm_bwd_ex call */
  const float* part;     /* its scratch */
  float *dgamma, *dbeta; /* accumulated into (either may be NULL) */
} emoasr_ln_finalize_item_t;
int emoasr_layernorm_bwd_finalize(int n, const emoasr_ln_finalize_item_t* items, void* stream);

/* ---- attention (transformer.py:48-99, conformer.py:57-95) -------------------
 * q:[B,Tq,H*DK] k,v:[B,Tk,H*DK] with row strides ldq/ldk/ldv (elements), out [B,Tq,H*DK].
 * pos: projected relative position table T [2*Tq-1, H*DK] (row r <-> rel = Tq-1-r) or NULL;
 * bias_u / bias_v f32 [H*DK] or NULL.  klens int32 [B]: keys >= klens[b] are masked
 * (NULL: none).  causal != 0 adds the lower-triangular mask (decoder self-attn).
 * scores = ((q+u).k + (q+v).pos[i-j]) * scale; softmax; dropout(drop_p); .v
 * lse f32 [B,H,Tq] is saved for backward. */
#define EMOASR_MAX_SEGMENTS 8
typedef struct {
  int B, H, DK, Tq, Tk;
  long ldq, ldk, ldv, ldo, ldp;
  const void *q, *k, *v, *pos;
  const float *bias_u, *bias_v;
  const int* klens;
  int causal;
  float scale;
  float drop_p;
  uint64_t seed;
  void* out;
  float* lse;
  /* backward only */
  const void* dout;
  float* delta; /* scratch f32 [B,H,Tq] */
  void *dq, *dk, *dv; /* T, same strides as q/k/v */
  float *dpos;        /* f32 [2*Tq-1, H*DK], accumulated */
  float *dbias_u, *dbias_v; /* f32 [H*DK], accumulated */
  /* optional backward scratch ("materialised" mode): when pdT != NULL the dq kernel stores the
   * dropped probabilities P^T and dS^T (and the un-shifted dBD band) once, and dV / dK / dpos are
   * computed by batched GEMMs over them instead of recomputing the scores three more times.
   * All three must be zero-filled by the caller before the first use for a given
   * (B, Tq, Tk, klens) and may then be reused by later calls with the same masks (the set of
   * never-written entries does not change).  Row strides ldpd >= Tq, ldbd >= 2*Tq-1, both
   * multiples of 8 elements. */
  void *pdT, *dsT;    /* T [B,H,Tk,ldpd] */
  void *dbd;          /* T [H,B,Tq,ldbd]; NULL without relative positions */
  long ldpd, ldbd;
  float* cs;          /* f32 [H, ldbd] scratch */
  /* optional: scaled scores S^T, f32 [B,H,Tk,ldst] (ldst >= Tq).  The forward stores them when
   * st != NULL; a backward given the same buffer (with the scratch above) reads them back instead
   * of recomputing (q+u).k + (q+v).pos, and computes the (q+v) part of dq as a batched GEMM over
   * the stored dBD band. */
  float* st;
  long ldst;
  /* optional, materialised backward only: Q + pos_bias_u and Q + pos_bias_v as dense T [B,Tq,H*DK]
   * tensors (written by the backward's first pass; the dK and dpos GEMMs then need no bias fix-up
   * passes), and per-(batch, query tile) partial sums of dbias_u / dbias_v, f32
   * [B * ceil(Tq/32), H, 2, DK], folded by one small reduction instead of contended atomics. */
  void *qu, *qv;
  float* dbias_part;
  /* optional: several stacked micro-batches ("segments", see emoasr_segments_t below) in ONE launch -- emoasr_attn_fwd and
   * emoasr_attn_bwd_fused only, self-attention (Tq == Tk).  nseg > 1: segment s holds utterances seg_b0[s] .. seg_b0[s+1]-1,
   * each padded to seg_T[s] frames; its rows start at row seg_row[s] of q / k / v / out / dout / dq / dk / dv (seg_row[nseg] =
   * all rows), its relative-position table at row seg_prow[s] of pos / dpos; lse and delta are [H * rows]: segment s at
   * H * seg_row[s], laid out [utterances, H, seg_T[s]].  B = all utterances, Tq = Tk = the longest segment, klens [B] in
   * order.  The dropout mask index restarts in every segment (seed + s * 0x9E3779B97F4A7C15).  nseg <= 1: one dense batch. */
  int nseg;
  int seg_b0[EMOASR_MAX_SEGMENTS + 1], seg_T[EMOASR_MAX_SEGMENTS];
  long seg_row[EMOASR_MAX_SEGMENTS + 1], seg_prow[EMOASR_MAX_SEGMENTS + 1];
  /* filled by the library (callers leave it zero): the order in which a stacked launch hands out its segments' workgroups --
   * longest segment first, so that the short ones fill the tail of the launch */
  int seg_order[EMOASR_MAX_SEGMENTS];
  /* optional (round 6, bf16 training): the attention-dropout keep mask of this launch as BITS -- uint32 [rows, H, keep_nw], one word
   * per (query row, head, 32-key tile), bit k of word w = key 32 w + k kept; rows = B * Tq (stacked: all segments' rows, in order).
   * emoasr_attn_dropmask hashes it ONCE per layer and step (dropout(p=dropout_attn_rate) of transformer.py:88); emoasr_attn_fwd and
   * both passes of emoasr_attn_bwd_fused then test a bit (2-3 instructions per element) instead of hashing the counter-based mask
   * again (13-26 per element).  NULL: the kernels hash inline / the backward fills a mask of its own. */
  const unsigned* keep_mask;
  int keep_nw;
} emoasr_attn_t;
/* words per (row, head) of a keep mask for keys 0 .. Tk-1, and the mask of the launch described by `a` (q / k / v pointers are not
 * read: B, H, Tq, Tk, klens, drop_p, seed and the segment table are) written to mask [rows, H, nw] */
long emoasr_attn_dropmask_words(int Tk);
int emoasr_attn_dropmask(int dtype, const emoasr_attn_t* a, unsigned* mask, int nw, void* stream);
int emoasr_attn_fwd(int dtype, const emoasr_attn_t* a, void* stream);
int emoasr_attn_bwd(int dtype, const emoasr_attn_t* a, void* stream);
/* The materialised mode's optional scratch as TWO areas instead of seven pointers: emoasr_attn_bwd_mat_bytes(..., which) bytes each
 * -- which = 0: the images (P^T, dS^T, dBD band; zero-filled by the caller before the first call for a given (B, Tq, Tk, klens),
 * reusable by later calls with the same masks, e.g. every layer of a backward sweep); which = 1: plain scratch (cs, Q + bias copies,
 * dbias partials; no initialisation, reusable by any call).  emoasr_attn_bwd_mat_bind fills pdT / dsT / dbd / ldpd / ldbd / cs /
 * qu / qv / dbias_part of `a` (B, H, Tq, Tk, pos, bias_u, bias_v already set) from the two base addresses.  Replaces the
 * tensor-by-tensor allocation of emoasr_amd/ops.py: AttnScratch for C callers (csrc/layer.hip: the f32 layer backward). */
size_t emoasr_attn_bwd_mat_bytes(int dtype, int B, int H, int Tq, int Tk, int rel, int which);
int emoasr_attn_bwd_mat_bind(int dtype, emoasr_attn_t* a, void* images, void* scratch);
/* Single-pass backward (bf16, no causal mask): every score tile is recomputed once; dQ, dK, dV, dbias_u/v and dpos come
 * out of four launches (prologue: delta, Q+u, Q+v, zeroed dQ accumulator; main: one workgroup per (batch, head, 128 keys)
 * sweeping the query tiles; dpos: diagonals of the stored dS; finalize: dQ f32 -> T and dbias_v).  Replaces the
 * materialised mode of emoasr_attn_bwd (P^T / dS^T / dBD images + three GEMMs): same arguments, the pdT / dsT / dbd / cs /
 * st / qu / qv / dbias_part fields are ignored.  `ws`: emoasr_attn_bwd_fused_ws_bytes(...) bytes of scratch, no
 * initialisation needed.  Reference: transformer.py:73-94, conformer.py:77-95 (their autograd backward). */
size_t emoasr_attn_bwd_fused_ws_bytes(int dtype, int B, int H, int Tq, int Tk, int rel);
/* the same for `rows` rows in all (stacked micro-batches: the segments' rows together, Tk = the longest segment) */
size_t emoasr_attn_bwd_fused_ws_bytes_rows(int dtype, long rows, int H, int Tk, int rel);
int emoasr_attn_bwd_fused(int dtype, const emoasr_attn_t* a, void* ws, size_t ws_bytes, void* stream);

/* ---- Conformer convolution module (conformer.py:98-143) ---------------------- */
/* out[M,C] = in[M,:C] * sigmoid(in[M,C:2C]) */
int emoasr_glu_fwd(int dtype, int M, int C, const void* in, void* out, void* stream);
int emoasr_glu_bwd(int dtype, int M, int C, const void* in, const void* dout, void* din, void* stream);
/* depthwise Conv1d over time, channels-last x[B,T,C], w f32 [C,K], zero padding (K-1)/2 */
int emoasr_dwconv_fwd(int dtype, int B, int T, int C, int K, const void* x, const float* w,
                      const float* bias, void* y, void* stream);
int emoasr_dwconv_bwd_x(int dtype, int B, int T, int C, int K, const void* dy, const float* w, void* dx,
                        void* stream);
/* depthwise_conv -> batch_norm of conformer.py:129-131 with the batch statistics fused: the conv also
 * writes per-block (sum, centred sum of squares) partials of its stored output into `part`
 * (emoasr_dwconv_stats_floats(B,T,C) floats); emoasr_bn_stats_finalize merges them (Chan's parallel
 * variance) into mean / biased var [C], updates the running statistics (unbiased var, `momentum`) and
 * increments num_batches_tracked (int64 on the device; all three optional), as nn.BatchNorm1d does. */
long emoasr_dwconv_stats_floats(int B, int T, int C);
int emoasr_dwconv_fwd_stats(int dtype, int B, int T, int C, int K, const void* x, const float* w,
                            const float* bias, void* y, float* part, void* stream);
int emoasr_bn_stats_finalize(int B, int T, int C, const float* part, float* mean, float* var,
                             float* running_mean, float* running_var, float momentum,
                             long long* num_batches_tracked, void* stream);
/* scratch: emoasr_dwconv_bwd_w_scratch_floats(B,T,C,K) floats of per-block partial sums */
int emoasr_dwconv_bwd_w(int dtype, int B, int T, int C, int K, const void* dy, const void* x, float* dw,
                        float* dbias, int accumulate, float* scratch, void* stream);
long emoasr_dwconv_bwd_w_scratch_floats(int B, int T, int C, int K);
/* BatchNorm1d batch statistics over all M=B*T rows (padding included, as the reference):
 * mean[C], var[C] (biased); if running_* != NULL they are updated with `momentum`
 * (unbiased variance), exactly like nn.BatchNorm1d in training mode. */
int emoasr_bn_stats(int dtype, int M, int C, const void* y, float* mean, float* var,
                    float* running_mean, float* running_var, float momentum, void* stream);
/* z = swish(gamma*(y-mean)/sqrt(var+eps)+beta) */
int emoasr_bn_swish_fwd(int dtype, int M, int C, const void* y, const float* mean, const float* var,
                        const float* gamma, const float* beta, float eps, void* z, void* stream);
/* training-mode backward (batch statistics): dy from dz; dgamma/dbeta (+)=.  C % 8 == 0.
 * scratch: emoasr_bn_swish_bwd_scratch_floats(M, C) floats of per-block partial sums
 * (ceil(M/64) x 2 x C); needs no initialisation. */
long emoasr_bn_swish_bwd_scratch_floats(int M, int C);
int emoasr_bn_swish_bwd(int dtype, int M, int C, const void* dz, const void* y, const float* mean,
                        const float* var, const float* gamma, const float* beta, float eps, void* dy,
                        float* dgamma, float* dbeta, float* scratch, void* stream);
/* ---- fused convolution-module kernels (csrc/convfused.hip; bf16, K <= 31, C % 8 == 0): bit-identical to the separate
 * launches they replace (every intermediate is rounded to bf16 at the same point).
 * emoasr_glu_dwconv_fwd: c = depthwise_conv(GLU(g)), g [B*T, 2C]; part (may be NULL) receives the per-block BatchNorm
 *   partial statistics of emoasr_dwconv_fwd_stats.  conformer.py:126-131.
 * emoasr_bn_swish_bwd_sums: passes 1-2 of emoasr_bn_swish_bwd (dgamma / dbeta accumulated; *tot_out = the [2][C] means
 *   inside `scratch`).
 * emoasr_conv_bwd_fused: given those means, ds (gradient w.r.t. the Swish output), cv (the depthwise convolution's output)
 *   and g -> dg [B*T, 2C] (BatchNorm/Swish backward, depthwise data gradient, GLU backward) and dw [C,K] / dbias [C]
 *   accumulated (the GLU output is recomputed from g).  scratch: emoasr_dwconv_bwd_w_scratch_floats() floats. */
int emoasr_glu_dwconv_fwd(int dtype, int B, int Tn, int C, int K, const void* g, const float* w, const float* bias, void* c,
                          float* part, void* stream);
int emoasr_bn_swish_bwd_sums(int dtype, int M, int C, const void* dz, const void* y, const float* mean, const float* var,
                             const float* gamma, const float* beta, float eps, float* dgamma, float* dbeta, float* scratch,
                             float** tot_out, void* stream);
int emoasr_conv_bwd_fused(int dtype, int B, int Tn, int C, int K, const void* ds, const void* cv, const float* mean,
                          const float* var, const float* gamma, const float* beta, float eps, const float* tot, const void* g,
                          const float* w, void* dg, float* dw, float* dbias, float* scratch, void* stream);

/* ---- element-wise / layout helpers ------------------------------------------ */
/* out (contiguous [d0,d1,d2,d3], dtype_out) (+)= in[strides s0..s3 (elements), dtype_in] */
int emoasr_strided_copy(int dtype_in, int dtype_out, const void* in, void* out, int d0, int d1, int d2,
                        int d3, long s0, long s1, long s2, long s3, int accumulate, void* stream);
/* y = x*scale*dropout(seed, i)  (n elements) */
int emoasr_scale_dropout(int dtype, long n, const void* x, void* y, float scale, float drop_p,
                         uint64_t seed, void* stream);
/* y[b,t,:] = (x[b,t,:]*scale + pe[t,:]) * dropout   (absolute positional encoding,
 * transformer.py:43-45); pe f32 [>=T, N] or NULL (conformer.py:49: scale only) */
int emoasr_posenc(int dtype, int B, int T, int N, const void* x, const float* pe, float scale,
                  float drop_p, uint64_t seed, void* y, void* stream);
/* y[i] = a[i] + b[i] */
int emoasr_add(int dtype, long n, const void* a, const void* b, void* y, void* stream);

/* dst_i [cols_i, rows_i] (compute dtype `dtype_out`, row stride ld_dst_i) = transpose of src_i [rows_i, cols_i] (f32, dense), all
 * n items in ONE launch: the transposed weight copies of emoasr_conformer_layer_t, refreshed after every optimizer step. */
typedef struct emoasr_tc_item {
  const float* src; void* dst;
  int rows, cols; long ld_dst;
} emoasr_tc_item_t;
#define EMOASR_TC_MAX 64
int emoasr_transpose_cast_batched(int dtype_out, int n, const emoasr_tc_item_t* items, void* stream);

/* ---- CTC (decoders/ctc.py:36-38,103-115,176-201; torch.nn.CTCLoss semantics) -- */
/* lse[m] = logsumexp_v logits[m,:V] */
int emoasr_row_lse(int dtype, int M, int V, const void* logits, long ld, float* lse, void* stream);
/* The CTC head in one pass (csrc/gemm_big.hip; bf16, N % 8 == 0, K % 64 == 0): C = A . B^T + bias stored AND lse[m] = log sum_n
 * exp(C[m,n]) of the row as stored -- the soft-max partials leave the product's epilogue (part: scratch [ceil(N / 64), M, 2] f32)
 * instead of a second pass over the logits (emoasr_gemm_nt + emoasr_row_lse).  decoders/ctc.py:103-113. */
int emoasr_gemm_nt_lse(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                       const float* bias, float* part, float* lse, void* stream);
/* Forward-backward lattices.  S = 2*Lmax+1 states.  lp (scratch), alpha, beta: f32 [B,T,S];
 * nll[b] = -log p(labels_b | x_b) (+inf if infeasible).  alpha and beta both include the
 * emission at t, so occupancy(t,s) = exp(alpha+beta-lp+nll). */
int emoasr_ctc_forward(int dtype, int B, int T, int V, int Lmax, const void* logits, long ld,
                       const float* lse, const int* labels, const int* elens, const int* ylens, int blank,
                       float* lp, float* alpha, float* beta, float* nll, void* stream);
/* gscale_dev (device, 1 float, may be NULL) multiplies gscale without a host sync.
 * grad[b,t,v] = gscale * (softmax - occupancy) for t < elens[b]; 0 elsewhere and for
 * utterances with nll = inf (zero_infinity=True). */
int emoasr_ctc_grad(int dtype, int B, int T, int V, int Lmax, const void* logits, long ld,
                    const float* lse, const int* labels, const int* elens, const int* ylens, int blank,
                    const float* lp, const float* alpha, const float* beta, const float* nll,
                    float gscale, const float* gscale_dev, void* grad, long ldg, void* stream);
/* The same for the utterances of several stacked micro-batches in one set of launches (engine.ctc_train_stacked): utterance b's
 * logits / lse / gradient rows are row0[b] + t, t < tpad[b] (its micro-batch's padded length; rows t >= tpad[b] do not exist),
 * instead of b * Tn + t; the lattice tables lp / alpha / beta stay [B, Tn, S] with Tn = the longest padded length; uscale[b]
 * (optional) multiplies gscale per utterance (micro-batch weight / micro-batch size).  row0 == NULL: the dense layout. */
int emoasr_ctc_forward_rows(int dtype, int B, int Tn, int V, int Lmax, const void* logits, long ld, const float* lse,
                            const int* labels, const int* elens, const int* ylens, int blank, const long* row0, float* lp,
                            float* alpha, float* beta, float* nll, void* stream);
int emoasr_ctc_grad_rows(int dtype, int B, int Tn, int V, int Lmax, const void* logits, long ld, const float* lse,
                         const int* labels, const int* elens, const int* ylens, int blank, const float* lp, const float* alpha,
                         const float* beta, const float* nll, float gscale, const float* gscale_dev, const long* row0,
                         const int* tpad, const float* uscale, void* grad, long ldg, void* stream);
/* greedy: best[b,t] = argmax_v logits (first max wins); hyp[b,:hyplen[b]] = collapse
 * repeats then drop blank, over t < elens[b] */
int emoasr_ctc_greedy(int dtype, int B, int T, int V, const void* logits, long ld, const int* elens,
                      int blank, int* best, int* hyp, int* hyplen, void* stream);

/* ---- Transformer decoder side (decoders/transformer.py:82-146, criteria.py:5-46) ---- */
/* out[m,:] = (table[ids[m],:]*scale + pe[m % L,:]) * dropout   (embedding + PositionalEncoder) */
int emoasr_embed_fwd(int dtype, int M, int L, int d, const int* ids, const void* table, const float* pe,
                     float scale, float drop_p, uint64_t seed, void* out, void* stream);
/* dtable[ids[m],:] += dout[m,:]*scale*dropout  (f32 atomics) */
int emoasr_embed_bwd(int dtype, int M, int d, const int* ids, const void* dout, float scale, float drop_p,
                     uint64_t seed, float* dtable, void* stream);
/* LabelSmoothingLoss rows: w[m] = 0 for padded positions, else the row weight (1/B [/ylen]);
 * loss[m] = -w*sum_v q[v]*log_softmax(logits[m])[v], q = 1-eps on labels[m], eps/(V-1) elsewhere;
 * grad (may be NULL) = gscale*[gscale_dev]*w*(softmax - q) */
int emoasr_lsm_loss(int dtype, int M, int V, const void* logits, long ld, const int* labels, const float* w,
                    float lsm_prob, float* loss, float gscale, const float* gscale_dev, void* grad, long ldg,
                    void* stream);

/* ---- knowledge distillation (asr/criteria.py:49-288, decoders/ctc_aligner.py:96-221) ---- */
/* Soft-target cross-entropy rows (one kernel behind DistillLoss, CTCAlignDistillLoss, RNNTWordDistillLoss,
 * RNNTAlignDistillLoss).  Row r reads logits row lrow[r] (lrow NULL: r), the dense f32 soft target
 * soft[src[r], :V] (src NULL or src[r] < 0: no soft term) and the label-smoothed hard target of class hard[r]
 * (hard NULL or < 0: none):  loss[r] = -(w_soft[r]*sum_v q_s[v] log p[v] + w_hard[r]*sum_v q_h[v] log p[v]);
 * grad (may be NULL; rows indexed like logits) = gscale*[gscale_dev]*(w_soft*(p*sum(q_s) - q_s) + w_hard*(p - q_h)) */
int emoasr_soft_ce(int dtype, int R, int V, const void* logits, long ld, const int* lrow, const float* soft, long lds,
                   const int* src, const int* hard, const float* w_soft, const float* w_hard, float lsm_prob,
                   float* loss, float gscale, const float* gscale_dev, void* grad, long ldg, void* stream);
/* CTCForcedAligner.__call__ (ctc_aligner.py:139-221) on the lattices of emoasr_ctc_forward (same lp/alpha/beta
 * [B,Tn,2*Lmax+1]): aligns[b,t] (int32 [B,Tn]) = token of the best state among those reachable from frame t-1's
 * choice, 0 for t >= elens[b]. */
int emoasr_ctc_best_path(int B, int Tn, int Lmax, const float* lp, const float* alpha, const float* beta,
                         const int* labels, const int* elens, const int* ylens, int blank, int* aligns, void* stream);
/* CTCAlignDistillLoss._frame_to_label_mapping (criteria.py:170-215): label_map[b,t] = index of the label that
 * frame t is assigned to, or -1; position 0 "all", 1 "left", 2 "mid", 3 "right"; count[b] = frames mapped. */
int emoasr_ctc_label_map(int B, int Tn, const int* aligns, const int* xlens, int blank, int position, int* label_map,
                         int* count, void* stream);

/* RNNTForcedAligner.__call__ (decoders/rnnt_aligner.py:158-198) on the lattices of emoasr_rnnt_forward (alpha, beta
 * f32 [B,Tn,U]): aligns (int32 [B,U-1]) = frame at which each label is emitted along the greedy alpha+beta walk;
 * labels not reached before the last frame keep 0. */
int emoasr_rnnt_best_path(int B, int Tn, int U, const float* alpha, const float* beta, const int* elens,
                          const int* ylens, int* aligns, void* stream);

/* ---- joint CTC/attention beam search (decoders/transformer.py:161-294, ctc_score.py:13-85) ---- */
/* out[m,v] = log_softmax(x[m,:V])[v] + mu*add[m,v]   (add may be NULL; f32 out) */
int emoasr_log_softmax(int dtype, int M, int V, const void* x, long ldx, const float* add, long lda, float mu,
                       float* out, long ldo, void* stream);
/* k largest per row, descending, ties to the lowest index; aux (optional) is gathered at the same
 * indices into aux_out */
int emoasr_topk(int M, int V, int k, const float* x, long ldx, const float* aux, long ldaux, float* vals,
                int* idx, float* aux_out, void* stream);
/* log_softmax(dec row) + mu * log_softmax(lm row) and its k largest entries in one launch (the beam search needs nothing
 * else of the score rows): vals / idx as emoasr_topk, lm_at = log_softmax(lm)[idx].  dec: compute dtype [M, V]; lm: f32 raw
 * logits [M, V] or NULL. */
int emoasr_beam_scores_topk(int dtype, int M, int V, int k, const void* dec, long ldd, const float* lm, long ldl, float mu,
                            float* vals, int* idx, float* lm_at, void* stream);
/* CTCPrefixScorer.initial_state: r f32 [T,2] from the CTC log-probs x f32 [T,V] */
int emoasr_ctc_prefix_init(int T, int V, const float* x, int blank, float* r, void* stream);
/* CTCPrefixScorer.__call__ for nb beams x cw candidates.  Beam m's previous state is
 * prev_states[parent[m], pcand[m]] (prev_states f32 [*, cw_prev, T, 2]) or init_state when
 * prev_states == NULL; last[m] / out_len[m] = last label and number of labels of the prefix.
 * -> log_psi f32 [nb,cw], states f32 [nb,cw,T,2] */
int emoasr_ctc_prefix_score(int nb, int T, int V, int cw, const float* x, const float* prev_states, int cw_prev,
                            const int* parent, const int* pcand, const float* init_state, const int* last,
                            const int* out_len, const int* cands, int blank, int eos, float* log_psi,
                            float* states, void* stream);

/* ---- RNN-Transducer decoder (decoders/rnn_transducer.py:81-240) ----------------------
 * LSTM cell (nn.LSTM gate order i,f,g,o): gates_pre T [B,4H] = x.W_ih^T + h.W_hh^T + b;
 * c_prev / c f32 [B,H]; h T rows of stride ldh; gates_act T [B,4H] saved for the backward. */
int emoasr_lstm_cell_fwd(int dtype, int B, int H, const void* gates_pre, const float* c_prev, void* h, long ldh,
                         float* c, void* gates_act, void* stream);
/* dgates_pre from dh_out (row stride lddh) + dh_rec (may be NULL); dc f32 [B,H] is updated in place */
int emoasr_lstm_cell_bwd(int dtype, int B, int H, const void* dh_out, long lddh, const void* dh_rec, float* dc,
                         const void* gates_act, const float* c_prev, const float* c, void* dgates_pre,
                         void* stream);
/* The whole recurrence of one LSTM layer in one cooperative launch (csrc/lstm_coop.hip; bf16, B <= 64, H % 32 == 0, H <= 512):
 * pre [U][B][4H] = x . W_ih^T + b_ih + b_hh, w_hh [4H][H]; outputs hseq [U][B][H], cseq f32 [U][B][H], gact [U][B][4H] (activated
 * i | f | g | o).  h0 / c0 may be NULL (zeros).  emoasr_lstm_seq_supported() -> 1 if this shape runs here (option "lstm_coop").  Launches on DIFFERENT streams are ordered against each other by a per-device event chain (two partly resident cooperative
 * launches would wait for each other's workgroups): streams overlap everything else, not two recurrences. */
int emoasr_lstm_seq_supported(int dtype, int B, int H);
int emoasr_lstm_seq_fwd(int dtype, int U, int B, int H, const void* pre, const void* w_hh, const void* h0, const float* c0,
                        void* hseq, float* cseq, void* gact, void* stream);
/* ... and backward: dgp [U][B][4H] (gradient w.r.t. the gate pre-activations) from dh_seq [U][B][H] (gradient w.r.t. the outputs),
 * the stored gact / cseq and c0 (may be NULL); ws: emoasr_lstm_seq_bwd_ws_bytes(B, H) bytes of scratch.
 * emoasr_lstm_coop_status(): 0 unless a grid barrier of these kernels ever gave up waiting (synchronises the device). */
long emoasr_lstm_seq_bwd_ws_bytes(int B, int H);
int emoasr_lstm_seq_bwd(int dtype, int U, int B, int H, const void* dh_seq, const void* gact, const float* cseq, const float* c0,
                        const void* w_hh, void* dgp, void* ws, long ws_bytes, void* stream);
long emoasr_lstm_coop_status(void);
/* joint network: h[b,t,u,:] = tanh(e[b,t,:] + g[b,u,:]) ; reductions of d(pre-tanh) back to de / dg */
int emoasr_joint_tanh(int dtype, int B, int T, int U, int J, const void* e, const void* g, void* h, void* stream);
int emoasr_joint_reduce(int dtype, int B, int T, int U, int J, const void* d, void* de, void* dg, void* stream);
/* transducer lattice on joint logits T [B,T,U,V] (U = Lmax+1): per-cell lse / blank / label log-probs,
 * alpha, beta f32 [B,T,U], nll[b] = -log p(y_b|x_b) */
int emoasr_rnnt_forward(int dtype, int B, int T, int U, int V, int Lmax, const void* logits, const int* labels,
                        const int* elens, const int* ylens, int blank, float* lse, float* lpb, float* lpy,
                        float* alpha, float* beta, float* nll, void* stream);
/* dlogits = gscale*[gscale_dev]*(softmax*occ - [blank]gamma_b - [label]gamma_y); 0 outside (elens, ylens) */
int emoasr_rnnt_grad(int dtype, int B, int T, int U, int V, int Lmax, const void* logits, const float* lse,
                     const float* lpb, const float* lpy, const float* alpha, const float* beta, const int* labels,
                     const int* elens, const int* ylens, const float* nll, int blank, float gscale,
                     const float* gscale_dev, void* dlogits, void* stream);
/* The transducer's output layer WITHOUT the [B,T,U,V] logits (csrc/gemm_big.hip epilogues + csrc/rnnt.hip; bf16, V % 8 == 0,
 * J % 64 == 0).  Replaces `self.output(torch.tanh(...))` + `log_softmax` + warp_rnnt's gathers of rnn_transducer.py:101-115,
 * 147-156 for training: z = h . W^T + bias is formed tile by tile on the MFMA pipeline and reduced in the epilogue.
 *   emoasr_rnnt_head_fwd   cells row0 .. row0 + nrows (h: their joint activations [nrows, J]): per row and 64-column chunk the
 *                          soft-max partials part[c, row0 + n] = (max, sum exp(z - max)) -- part is the WHOLE chunk-major table
 *                          [ceil(V / 64), part_rows, 2] of all cells, every launch fills its rows -- and the two
 *                          logits the lattice reads, zb[n] = z[n, blank], zy[n] = z[n, labels[b, u]] (u < ylens[b]);
 *   emoasr_rnnt_forward_parts   lse from the partials, zb / zy turned into lpb / lpy IN PLACE, then the alpha / beta lattices and
 *                          nll exactly as emoasr_rnnt_forward;
 *   emoasr_rnnt_coef       the per-cell constants of the gradient (coef [cells, 4] = lse, occ, gamma_blank, gamma_label, the last
 *                          three times gscale [* gscale_dev]; ycol [cells] = label column or -1; zeros outside (elens, ylens));
 *   emoasr_rnnt_head_grad  dz[n, :] = exp(z - lse) occ - [blank] gamma_b - [label] gamma_y for the nrows cells of one row chunk,
 *                          z RECOMPUTED from h: the caller walks the cells in chunks (dz chunk -> emoasr_gemm_tn / _nn), so no
 *                          [cells, V] buffer exists in either direction. */
int emoasr_rnnt_head_fwd(int dtype, long row0, int nrows, int T, int U, int V, int J, int Lmax, const void* h, const void* w,
                         const float* bias, const int* labels, const int* ylens, int blank, float* part, long part_rows, float* zb,
                         float* zy, const int* ycol, void* stream);
/* ycol [B*T*U] for emoasr_rnnt_head_fwd: the label column of every lattice cell, labels[b, u] for u < ylens[b], else -1 */
int emoasr_rnnt_ycol(int B, int T, int U, int Lmax, const int* labels, const int* ylens, int* ycol, void* stream);
int emoasr_rnnt_forward_parts(int B, int T, int U, int V, const float* part, const int* elens, const int* ylens, float* lse,
                              float* zb_lpb, float* zy_lpy, float* alpha, float* beta, float* nll, void* stream);
int emoasr_rnnt_coef(int B, int T, int U, int Lmax, const float* lse, const float* lpb, const float* lpy, const float* alpha,
                     const float* beta, const int* labels, const int* elens, const int* ylens, const float* nll, float gscale,
                     const float* gscale_dev, float* coef, int* ycol, void* stream);
int emoasr_rnnt_head_grad(int dtype, int nrows, int V, int J, const void* h, const void* w, const float* bias, const float* coef,
                          const int* ycol, int blank, void* dz, long lddz, void* stream);
/* out[m] = argmax_v x[m,:V] (first maximum) */
int emoasr_argmax_rows(int dtype, int M, int V, const void* x, long ldx, int* out, void* stream);
/* out[0] = first i < n with x[i] != value (-1: none), out[1] = x[that i] (value: none) -- the windowed greedy
 * transducer search (rnn_transducer.py:194-240) finds the next non-blank frame with one 8-byte D2H copy */
int emoasr_first_not_equal(int n, const int* x, int value, int* out, void* stream);

/* ---- greedy transducer search on the device (csrc/rnnt_greedy.hip; RNNTDecoder._greedy, rnn_transducer.py:194-240) --------------
 * ONE launch per utterance: 32 or 64 co-resident workgroups (64 when H is a multiple of 64 and V >= 64) run the whole time-synchronous
 * search.  Each owns a slice of every matrix (rows of the output and w_dec projections, the LSTM rows of its H / G hidden units), kept
 * in LDS; per step a [V x J] GEMV + arg-max, per emitted label two LSTM layers + a [J x H] GEMV.  The hidden state, the joint input and
 * the per-workgroup arg-max candidates travel between the workgroups as DATA-TAGGED 8-byte words (value + 12-bit step tag in one relaxed
 * store, readers poll for the tag): no grid barrier, no host round trip per label; a wait that gives up sets the error flag.
 * e_all [T, J] = w_enc . eouts + bias in the compute dtype; b_lstm0 / b_lstm1 = bias_ih + bias_hh [4H] of the two layers;
 * ws: emoasr_rnnt_greedy_ws_bytes(H, J) bytes (cleared by the call).
 * Outputs (device): hyp int32 [max_len + 1], align int32 [T + max_len + 1] (the arg-max of every joint evaluation, in order),
 * lens int32 [2] = {len(hyp), len(align)}.  emoasr_rnnt_greedy_status: 0 unless a wait of the last launch in `ws` gave up
 * (synchronises the stream).
 * emoasr_rnnt_greedy_supported (the MODEL): bf16 or f32, two LSTM layers, E / H / J multiples of 8, H a multiple of the G workgroups
 * with at most 16 hidden units per workgroup, E <= 1024, H <= 1024, J <= 1024, V >= G, J >= G.
 * emoasr_rnnt_greedy_fits (ONE utterance): _supported, and the workgroup's LDS image within 160 KB, T + max_len + 2 < 4095 (the step
 * tag), V < 2^20.  An utterance that does not fit takes the launch chain (engine.rnnt_greedy falls back per utterance). */
long emoasr_rnnt_greedy_fits(int dtype, int E, int H, int J, int V, int nl, int T, int max_len);
long emoasr_rnnt_greedy_supported(int dtype, int E, int H, int J, int V, int nl);
long emoasr_rnnt_greedy_ws_bytes(int H, int J);
int emoasr_rnnt_greedy(int dtype, int T, int E, int H, int J, int V, int blank, int eos, int max_len, const void* e_all,
                       const void* emb, const void* w_ih0, const void* w_hh0, const float* b_lstm0, const void* w_ih1,
                       const void* w_hh1, const float* b_lstm1, const void* w_dec, const float* b_dec, const void* w_out,
                       const float* b_out, void* ws, long ws_bytes, int* hyp, int* align, int* lens, void* stream);
long emoasr_rnnt_greedy_status(const void* ws, void* stream);

/* ---- one expansion round of the transducer beam search in five launches (csrc/rnnt_beam.hip; RNNTDecoder._beam_search,
 * rnn_transducer.py:242-325: the per-round recurrency + joint + log_softmax + topk of the reference's alignment-length-synchronous
 * search).  nb <= 16 live hypotheses; LSTM states live in slot-addressed pools ph [slots, H] (compute dtype) / pc [slots, H] (f32).
 * All index lists are int64 device arrays (the search's uploaded control words), so the launches can be captured into a graph.
 *   emoasr_rnnt_beam_lstm : one LSTM layer step: x_i = xtab[xidx[i]] (embedding rows by label id / the layer below's new h by slot),
 *                           (h, c) = pools[src[i]] -> gates = W_ih x + b + W_hh h -> new (h, c) written to pools[dst[i]];
 *                           bias = bias_ih + bias_hh [4H]; dst must not alias any src of the same round; the index lists may live
 *                           in pinned host memory (each word is read once); copy_dst != NULL: the launch also copies copy_n <= 192
 *                           int64 words copy_src -> copy_dst (the host-resident control record to its device twin)
 *   emoasr_rnnt_beam_joint: hj[i] = tanh(e_all[*t] + w_dec . ph[dst[i]] + b_dec)   (rnn_transducer.py:147-156 without the output layer)
 *   emoasr_rnnt_beam_pick : out[i] = { log_softmax(logits[i])[blank], the k best of log_softmax(logits[i])[1:] (descending, ties ->
 *                           lowest index), their indices relative to column 1 (as floats) }, out row stride ldo >= 1 + 2 k */
int emoasr_rnnt_beam_lstm(int dtype, int nb, int nin, int H, const void* xtab, long ldx, const long long* xidx, const void* w_ih,
                          const void* w_hh, const float* bias, void* ph, float* pc, const long long* src, const long long* dst,
                          const long long* copy_src, long long* copy_dst, int copy_n, void* stream);
int emoasr_rnnt_beam_joint(int dtype, int nb, int H, int J, int Tmax, const void* ph, const long long* dst, const void* w_dec,
                           const float* b_dec, const void* e_all, const long long* t, void* hj, void* stream);
int emoasr_rnnt_beam_pick(int dtype, int nb, int V, int k, int blank, const void* logits, long ldl, float* out, long ldo,
                          void* stream);

/* ---- one Conformer encoder layer, forward, sequenced on the host in C++ ---------
 * ConformerEncoderLayer.forward (asr/modeling/conformer.py:146-225) with relative-position attention:
 *   x += 0.5 * drop(FFN_macaron(LN(x)));  x += drop(RelMHA(LN(x)));  x += drop(ConvModule(LN(x)));
 *   x += 0.5 * drop(FFN(LN(x)));  y = LN(x)
 * as ONE call that enqueues the ~24 kernels above on `stream` (the caller pays one FFI crossing per
 * layer instead of one per kernel).  Every intermediate the backward needs is written to
 * caller-provided buffers (emoasr_conformer_fwd_t); mean / rstd / u pointers may be NULL for inference.
 * Weights (w*, pw*, wqkv [3d,d], wpos, wout) are in the compute dtype, everything else f32.
 * seed[] = {ffm_in, ffm_out, att_probs, att_out, conv_out, ff_in, ff_out} dropout streams. */
typedef struct emoasr_ffn_params {
  const float *ln_g, *ln_b;
  const void* w1; const float* b1;   /* [F,d], [F] */
  const void* w2; const float* b2;   /* [d,F], [d] */
} emoasr_ffn_params_t;
typedef struct emoasr_conformer_layer {
  int d, H, F, K;                    /* model dim, heads, FFN inner dim, depthwise kernel size */
  emoasr_ffn_params_t ffm, ff;
  const float *att_ln_g, *att_ln_b;
  const void* wqkv; const float* bqkv;
  const void* wpos;
  const float *bias_u, *bias_v;
  const void* wout; const float* bout;
  const float *cv_ln_g, *cv_ln_b;
  const void* pw1; const float* pw1_b;   /* [2d,d] */
  const float *dw_w, *dw_b;              /* [d,K], [d] */
  const float *bn_g, *bn_b;
  float *bn_rm, *bn_rv; long long* bn_nbt;
  const void* pw2; const float* pw2_b;
  const float *fin_ln_g, *fin_ln_b;
  /* optional (bf16 backward, NULL = absent): TRANSPOSED copies of the weights whose data-gradient products are long reductions
   * onto one 256-column tile -- ffm / ff w1 as [d,F], wqkv as [d,3d], pw1 as [d,2d] (emoasr_transpose_cast_batched keeps them
   * current).  dX = dY . W is then the NT product dY . (W^T)^T, which the large-tile kernel takes (csrc/gemm_big.hip). */
  const void *ffm_w1t, *ff_w1t, *wqkv_t, *pw1_t;
  /* likewise optional (round 5): transposed copies of the K = d products' weights -- ffm / ff w2 as [F,d], wout and pw2 as [d,d] --
   * so that their data gradients run as NT products too (the k-major B tile of the NN form is read with transposing LDS loads) */
  const void *ffm_w2t, *ff_w2t, *wout_t, *pw2_t;
} emoasr_conformer_layer_t;
typedef struct emoasr_ffn_stash {
  void *h, *u, *a, *y;               /* LN out [M,d], pre-activation [M,F] (optional), activation [M,F], block output [M,d] */
  float *mean, *rstd;
} emoasr_ffn_stash_t;
/* Stacked micro-batches ("segments"): the reference accumulates gradients over `accum_grad` independent forward /
 * backward passes (asr/train_asr.py:106-128).  Here up to EMOASR_MAX_SEGMENTS of those micro-batches go through a layer
 * TOGETHER: their rows are concatenated (segment s = B[s] utterances padded to T[s] frames; M = sum B[s] T[s] rows), so every
 * row-wise kernel -- the projections, LayerNorms, the vocabulary head -- runs once over all of them, while attention stays per
 * utterance, the depthwise convolution sees each segment's own zero padding and BatchNorm takes its batch statistics PER
 * SEGMENT and updates the running statistics once per segment, in order -- the arithmetic of the separate passes.
 * n = 0 (or 1 with B[0], T[0]): one dense batch. */
typedef struct emoasr_segments {
  int n;
  int B[EMOASR_MAX_SEGMENTS], T[EMOASR_MAX_SEGMENTS];
} emoasr_segments_t;
typedef struct emoasr_conformer_fwd {
  int B, T;                          /* dense batch; with seg.n > 1: B = sum seg.B, T = max seg.T (informative) */
  const void* x;                     /* [M, d], M = B*T or the stacked row count */
  const void* pos_t;                 /* [2T-1, d] relative position table (after dropout), compute dtype; stacked: the
                                      * segments' tables one after the other, sum (2 T[s] - 1) rows */
  const int* klens;                  /* int32 [B] valid frames (stacked: all segments' utterances in order) */
  int training;                      /* batch statistics + running-stat update in BatchNorm */
  float p_enc, p_att;
  uint64_t seed[7];
  emoasr_ffn_stash_t ffm, ff;
  void *at_h, *qkv, *pp, *o, *at_y; float *lse, *at_mean, *at_rstd;
  void *cv_h, *g, *gl, *c, *z, *cv_y; float *bmean, *bvar, *bn_part, *cv_mean, *cv_rstd;
  void* y; float *fin_mean, *fin_rstd;
  emoasr_segments_t seg;             /* stacked micro-batches; lse is [H * M]: segment s at H * (its first row), laid out
                                      * [B[s], H, T[s]]; bmean / bvar are [n, d]; bn_part holds the segments' partial-sum
                                      * areas (emoasr_dwconv_stats_floats(B[s], T[s], d) floats each) back to back */
  /* optional (bf16 training with attention dropout): room for the layer's attention keep mask as bits, uint32
   * [M, H, att_mask_nw] with att_mask_nw >= emoasr_attn_dropmask_words(longest T).  The forward hashes the mask once (on the
   * attention's side stream, under the macaron feed-forward block) and the attention forward and both backward passes test bits
   * (emoasr_attn_t::keep_mask); it is part of the stash the backward takes.  NULL: every kernel hashes for itself. */
  unsigned* att_mask; int att_mask_nw;
  int att_mask_ready;                /* 1: att_mask already holds (or is being filled on the attention's side stream with) this layer's
                                      * bits -- emoasr_conformer_attn_masks hashed all layers' masks at the start of the pass */
} emoasr_conformer_fwd_t;
/* The attention keep masks of ALL nl layers of an encoder pass, hashed up front on the attention's side stream (forked from
 * `stream` here): called before the convolution front-end, the ~0.05 ms of integer hashing per layer run under its MFMA-bound
 * products instead of beside each layer's memory-bound kernels.  seeds[l] = the layer's att_probs dropout stream (seed[2] of its
 * emoasr_conformer_fwd_t); masks + l * layer_stride_words = its att_mask [M, H, nw]. */
int emoasr_conformer_attn_masks(int dtype, int nl, const emoasr_segments_t* seg, int B, int T, int H, int d, const int* klens,
                                float p_att, const uint64_t* seeds, unsigned* masks, long layer_stride_words, int nw, void* stream);
int emoasr_conformer_layer_fwd(int dtype, const emoasr_conformer_layer_t* layer,
                               const emoasr_conformer_fwd_t* io, void* stream);
/* Backward of the same layer (training-mode forward with stash) as one call: gradient kernels in the order the
 * reference's autograd runs them (conformer.py:146-225 backwards), the layer's nine weight-gradient products as one grouped
 * launch.  `grads`: the layer struct again with every parameter pointer replaced by the address of its f32 gradient
 * (accumulated into; d/H/F/K and the running-statistics fields are ignored).  `st`: the emoasr_conformer_fwd_t the forward
 * call was given (its buffers still intact).  dy / dx: gradient w.r.t. the layer's output / input, [B*T, d].
 * ws: emoasr_conformer_layer_bwd_ws_bytes(...) bytes, no initialisation, reusable by the next layer's call.
 * ln_part: 5 areas of emoasr_layernorm_bwd_scratch_floats(d) floats, ln_part_stride floats apart, that receive the
 * dgamma / dbeta partial sums of the layer's LayerNorms (order: final, feed-forward, convolution, attention, macaron);
 * the caller folds them with emoasr_layernorm_bwd_finalize (one launch for the whole backward sweep). */
typedef struct emoasr_conformer_bwd {
  const void* dy;
  void* dx;
  void* ws; size_t ws_bytes;
  float* ln_part; long ln_part_stride;
  /* f32 only (the attention backward of f32 layers is the materialised emoasr_attn_bwd, one call per stacked micro-batch):
   * emoasr_conformer_layer_bwd_img_bytes_seg(...) bytes holding every micro-batch's P^T / dS^T / dBD images, ZERO-filled by the
   * caller once per backward sweep (the masks of a sweep are the same in every layer) and passed to every layer's call. */
  void* attn_img; size_t attn_img_bytes;
} emoasr_conformer_bwd_t;
size_t emoasr_conformer_layer_bwd_ws_bytes(int dtype, int B, int T, int d, int H, int F, int K);
/* The convolution module's per-utterance part for stacked micro-batches, every kernel taking ALL segments in one launch (bf16;
 * csrc/convfused.hip).  Same arithmetic as emoasr_glu_dwconv_fwd + emoasr_bn_stats_finalize + emoasr_bn_swish_fwd (forward) and
 * emoasr_bn_swish_bwd_sums + emoasr_conv_bwd_fused (backward) called once per segment, in order: the depthwise convolution sees each
 * segment's own zero padding, BatchNorm takes its batch statistics per segment (bmean / bvar [n, C]) and moves the running
 * statistics once per segment.  g / dg [M, 2C], c / z / dz [M, C]; part: the segments' emoasr_dwconv_stats_floats areas back to back;
 * scratch sizes from emoasr_conv_module_bwd_seg_scratch_floats (which = 0: BatchNorm sums, 1: depthwise weight-gradient partials).
 * Reference: conformer.py:126-133 and its autograd. */
int emoasr_conv_module_fwd_seg(int dtype, const emoasr_segments_t* seg, int C, int K, const void* g, const float* w,
                               const float* bias, void* c, float* part, float* bmean, float* bvar, float* running_mean,
                               float* running_var, float momentum, long long* num_batches_tracked, const float* gamma,
                               const float* beta, float eps, void* z, int training, void* stream);
long emoasr_conv_module_bwd_seg_scratch_floats(const emoasr_segments_t* seg, int C, int K, int which);
int emoasr_conv_module_bwd_seg(int dtype, const emoasr_segments_t* seg, int C, int K, const void* dz, const void* c,
                               const float* bmean, const float* bvar, const float* gamma, const float* beta, float eps,
                               float* dgamma, float* dbeta, const void* g, const float* w, void* dg, float* dw, float* dbias,
                               float* bn_scratch, float* dw_scratch, void* stream);
size_t emoasr_conformer_layer_bwd_ws_bytes_seg(int dtype, const emoasr_segments_t* seg, int d, int H, int F, int K);
/* emoasr_set_option("wgrad_side", 1): emoasr_conformer_layer_bwd puts its grouped weight-gradient launch on a side stream (one
 * per (device, caller stream)) where it runs under the NEXT layer's gradient chain.  The launch reads the call's workspace: callers
 * then alternate between TWO workspaces from call to call (a call waits for the launch issued two calls earlier before it touches
 * its workspace) and call emoasr_wgrad_side_join(keep, stream) before anything reads the weight gradients or frees a workspace:
 * the caller's stream waits for the launches in flight -- all of them (keep = 0) or all but the latest (keep = 1: the gradients of
 * every layer but the one just finished are final; for a per-layer gradient hook one layer behind the sweep).  Captured streams run
 * the launch inline. */
int emoasr_wgrad_side_join(int keep, void* stream);
size_t emoasr_conformer_layer_bwd_img_bytes_seg(int dtype, const emoasr_segments_t* seg, int d, int H);   /* 0 for bf16 */
int emoasr_conformer_layer_bwd(int dtype, const emoasr_conformer_layer_t* layer, const emoasr_conformer_layer_t* grads,
                               const emoasr_conformer_fwd_t* st, const emoasr_conformer_bwd_t* io, void* stream);

/* ---- beam-search step runtime (inference): the Transformer decoder over the whole prefix and the
 * Transformer LM's next-token distribution, each as ONE call per output step ---------------------------
 * emoasr_transformer_decoder_infer: TransformerDecoder.forward_one_step (decoders/transformer.py:148-159;
 * layers: transformer.py:156-198, eps 1e-12, ReLU FFN) for nb live hypotheses of L tokens each.  Like the
 * reference it recomputes the prefix (no self-attention cache); unlike it, the cross-attention keys / values
 * of the encoder memory are projected once per utterance (kv[l]: [nb,T,2*dd], K then V) and only the LAST
 * position goes through the output projection: logits_last [nb,V].
 * emoasr_bert_lm_infer: TransformerLM.predict (lm/modeling/transformer.py:62-77 over the BERT-style stack,
 * modeling_bert.py:159-303,360-436): log-softmax of the next-token logits at position L-1, f32 [nb,V].
 * `ws`: scratch of at least emoasr_decode_ws_bytes(...) bytes. */
typedef struct emoasr_lin { const void* w; const float* b; } emoasr_lin_t;   /* weight (compute dtype), bias f32 */
typedef struct emoasr_lnp { const float *g, *b; } emoasr_lnp_t;
typedef struct emoasr_decoder_layer {
  emoasr_lnp_t ln1, ln2, ln3;
  emoasr_lin_t qkv, out;        /* self-attention: fused [3dd,dd] projection, output */
  emoasr_lin_t q2, out2;        /* source attention (K / V come from the cache) */
  emoasr_lin_t w1, w2;          /* feed-forward [F,dd], [dd,F] */
} emoasr_decoder_layer_t;
typedef struct emoasr_decoder_infer {
  int nb, L, T, dd, H, F, V;
  const int* ids;               /* int32 [nb,L] */
  const void* embed; const float* pe; float emb_scale;   /* token table [V,dd], abs. position table f32 [>=L,dd] */
  const int *kself, *kmem;      /* int32 [nb]: valid prefix length (= L), valid memory frames */
  const void* const* kv;        /* nl pointers */
  emoasr_lnp_t ln_out; emoasr_lin_t out;
  void* logits_last;
  void* ws; size_t ws_bytes;
} emoasr_decoder_infer_t;
typedef struct emoasr_bert_layer {
  emoasr_lin_t qkv, attn_out; emoasr_lnp_t ln_attn;
  emoasr_lin_t inter, out; emoasr_lnp_t ln_out;
} emoasr_bert_layer_t;
typedef struct emoasr_bert_infer {
  int nb, L, d, H, F, V;
  const int* ids; const void* word_emb; const float* pe;  /* pe = position + token-type-0 embeddings, f32 [>=L,d] */
  emoasr_lnp_t ln_emb;
  const int* klens;
  emoasr_lin_t transform; emoasr_lnp_t ln_transform; const float* out_bias;
  float* logp;
  void* ws; size_t ws_bytes;
} emoasr_bert_infer_t;
size_t emoasr_decode_ws_bytes(int dtype, int nb, int L, int d, int H, int F, int V);
int emoasr_transformer_decoder_infer(int dtype, int nl, const emoasr_decoder_layer_t* layers,
                                     const emoasr_decoder_infer_t* io, void* stream);
int emoasr_bert_lm_infer(int dtype, int nl, const emoasr_bert_layer_t* layers, const emoasr_bert_infer_t* io,
                         void* stream);

/* ---- the same two networks one POSITION at a time, with self-attention K / V caches, and the beam bookkeeping on the
 * device (csrc/decode_rt.hip): a step is a fixed launch sequence without host input, so it can be captured in a HIP graph.
 * Caches: compute dtype [nl][nb][Lmax][d] (K and V separately).  `pos` (device int) = position of the token in `ids`
 * (prefix length - 1); klens (device int32 [nb]) = pos + 1.  emoasr_beam_cache_gather re-orders the caches of a new step's
 * hypotheses by their parent: dst[l][b][t < pos] = src[l][parent[b]][t]. */
typedef struct emoasr_decoder_step {
  int nb, Lmax, T, dd, H, F, V;
  const int* ids; const int* pos; const int* klens;
  const void* embed; const float* pe; float emb_scale;
  const int* kmem; const void* const* kv;       /* as in emoasr_decoder_infer_t */
  void* kcache; void* vcache;
  emoasr_lnp_t ln_out; emoasr_lin_t out;
  void* logits_last;                             /* [nb, V] compute dtype */
  void* ws; size_t ws_bytes;                     /* emoasr_decode_step_ws_bytes() */
} emoasr_decoder_step_t;
typedef struct emoasr_bert_step {
  int nb, Lmax, d, H, F, V;
  const int* ids; const int* pos; const int* klens;
  const void* word_emb; const float* pe; emoasr_lnp_t ln_emb;
  void* kcache; void* vcache;
  emoasr_lin_t transform; emoasr_lnp_t ln_transform; const float* out_bias;
  float* logp;                                   /* f32 [nb, V] log-probabilities of the next token */
  int raw_logits;                                /* 1: leave the raw logits in logp (no log-softmax) */
  void* ws; size_t ws_bytes;
} emoasr_bert_step_t;
size_t emoasr_decode_step_ws_bytes(int dtype, int nb, int d, int H, int F, int V);
int emoasr_transformer_decoder_step(int dtype, int nl, const emoasr_decoder_layer_t* layers, const emoasr_decoder_step_t* io,
                                    void* stream);
int emoasr_bert_lm_step(int dtype, int nl, const emoasr_bert_layer_t* layers, const emoasr_bert_step_t* io, void* stream);
/* Small-M pieces of those steps (csrc/rowlin.hip; bf16): y[M <= 16, N] = act(LN?(x) . W^T + bias) (+ res | LayerNorm(res)) with the
 * LayerNorms computed inside the kernel, and single-query attention of row b of qkv [nb, 3d] against the caches, appending
 * the new key / value at *pos first. */
int emoasr_rowlin(int M, int N, int K, const void* x, long ldx, const float* lna_g, const float* lna_b, float lna_eps,
                  const void* w, const float* bias, int act, const void* res, long ldres, const float* lnr_g,
                  const float* lnr_b, float lnr_eps, void* y, int out_f32, long ldy, void* stream);
int emoasr_attn_step(int nb, int d, int H, int Lmax, const void* qkv, void* kcache, void* vcache, const int* pos, void* out,
                     void* stream);
/* The steps run as one cooperative launch per network (csrc/decode_coop.hip, option "decode_coop", bf16, <= 16 hypotheses): 0 when
 * none of its grid barriers ever gave up waiting, 1 otherwise (results of that step are then undefined), -1 on a runtime error.
 * Synchronises the device. */
long emoasr_decode_coop_status(void);
/* times the cooperative step kernel of a chain (0 decoder, 1 LM) was enqueued or captured since the library was loaded */
long emoasr_decode_coop_launches(int chain);
int emoasr_beam_cache_gather(int dtype, int nl, int nb, int Lmax, int d, const void* src_k, const void* src_v, void* dst_k,
                             void* dst_v, const int* parent, const int* pos, void* stream);
/* Beam bookkeeping of one output step (decoders/transformer.py:215-290) for up to 32 beams x 32 candidates, one workgroup:
 * candidate scores in numpy-float32 arithmetic, accumulated hypothesis scores in double (Python float), stable orders,
 * <eos> -> result (score + len_weight * len(hyp incl. sos/eos); empty hypotheses dropped), the rest -> the next step's
 * beams.  All arrays live on the device; `state` carries pos / n_alive / n_results / done between steps (done: the kernel
 * returns immediately, so replaying a captured step past the end is harmless).  hist_parent / hist_token [max_steps][bw]
 * record the surviving beams of every step (the host rebuilds the token sequences once, at the end). */
typedef struct emoasr_beam_state { int pos, n_alive, n_results, done; } emoasr_beam_state_t;
typedef struct emoasr_beam_update {
  int bw, cw, eos;
  float one_minus_lam, lam, mu;                  /* f32(1 - ctc_weight), f32(ctc_weight), f32(lm_weight) */
  double len_weight;
  const float* vals; const int* cands;           /* [bw, cw] top candidates of scores_att(+lm) per beam */
  const float* lm_at;                            /* [bw, cw] LM log-probs at the candidates, or NULL */
  const float* psi;                              /* [bw, cw] CTC prefix scores, or NULL (no CTC term) */
  double* score; float* score_ctc;               /* [bw] current beams; overwritten with the next step's */
  int *n_ids, *n_parent, *n_pcand, *n_last, *n_outlen, *n_klens;   /* [bw] inputs of the next step */
  int *hist_parent, *hist_token;
  double* res_score; int* res_step; int* res_parent;               /* [bw] finished hypotheses */
  emoasr_beam_state_t* state;
  int* host_mirror;                              /* optional: pinned host memory (device-visible) that receives a copy of `state`
                                                  * at the end of every live step -- the host polls it instead of copying */
} emoasr_beam_update_t;
int emoasr_beam_update(const emoasr_beam_update_t* u, void* stream);
/* One whole output step (cache gather, decoder step, LM step on `side_stream` when given, log-softmax + LM fusion, top-cw,
 * CTC prefix scores, beam update) from one call.  parent / pcand / last / out_len of the CURRENT beams are upd.n_parent /
 * n_pcand / n_last / n_outlen as the previous step's update left them; *_prev are the previous step's caches and CTC scorer
 * states [bw, cw, T, 2] (before the first step: the initial state at [0, 0]).  lm_nl == 0: no LM; upd.psi == NULL: no CTC. */
typedef struct emoasr_joint_step {
  int dec_nl; const emoasr_decoder_layer_t* dec_layers; emoasr_decoder_step_t dec;
  int lm_nl; const emoasr_bert_layer_t* lm_layers; emoasr_bert_step_t lm;
  const void *dec_k_prev, *dec_v_prev, *lm_k_prev, *lm_v_prev;
  const int* parent;
  float* scores_pre;                               /* f32 [bw, V] */
  const float* ctc_x; int T; int blank;            /* CTC log-probs f32 [T, V] */
  const float* states_prev; float* states_cur;
  emoasr_beam_update_t upd;
} emoasr_joint_step_t;
int emoasr_joint_beam_step(int dtype, const emoasr_joint_step_t* js, void* stream, void* side_stream);
/* Parts of the step on ONE stream (mask: 1 = decoder chain incl. its cache gather, 2 = LM chain, 4 = scoring tail), and the
 * parts as HIP graphs: build captures part 0 / 1 / 2 (decoder / LM / tail) of slot 0 / 1 (even / odd steps), or updates the
 * instantiated graph with a new utterance's pointers; launch replays it.  The caller orders the three launches of a step
 * with events: the two chains on two streams, the tail after both (~185 kernels per ~0.1 ms of host time, against
 * ~1.1 ms for launching them one by one). */
int emoasr_joint_beam_step_parts(int dtype, const emoasr_joint_step_t* js, int parts, void* stream);
int emoasr_joint_beam_graph_build(int dtype, const emoasr_joint_step_t* js, int slot, int part, void* stream);
int emoasr_joint_beam_graph_launch(int slot, int part, void* stream);

/* ---- optimizer (asr/train_asr.py:84-92, torch.optim.Adam semantics) ---------- */
/* out[0] += sum x^2 */
int emoasr_sqnorm(long n, const float* x, float* out, void* stream);
/* Adam with coupled L2 weight decay.  gnorm_sq (device, 1 float): gradients are
 * scaled by min(1, clip/(sqrt(gnorm_sq)+1e-6)) when clip > 0 (clip_grad_norm_);
 * the whole update is skipped on device if gnorm_sq is NaN/Inf (train_asr.py:88-89). */
int emoasr_adam_step(long n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, const float* gnorm_sq,
                     float clip, float grad_mult, void* stream);
/* the same; a skipped step also increments *skipped (device int, may be NULL): the host subtracts it from its step
 * counters at its next synchronisation point, so that the schedule position and the bias correction follow the number of
 * updates actually applied, as in the reference (which does not call optimizer.step() on a NaN norm) */
int emoasr_adam_step_ex(long n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, const float* gnorm_sq, float clip, float grad_mult, int* skipped,
                        void* stream);

/* ---- on-GPU features --------------------------------------------------------- */
/* SpecAugment (asr/spec_augment.py:39-95): zero (or fill) bands.  spans int32
 * [B, nf+nt, 2] = (start, end) per mask, first nf are frequency masks; built on the
 * host from the reference's sampling rule.  x f32 [B,T,F] in place. */
int emoasr_specaug_apply(int B, int T, int F, float* x, const int* spans, int nf, int nt,
                         const int* xlens, const float* fill, void* stream);
/* Kaldi-compatible log-mel filterbank (corpora/utils/wav_to_feats.py:26-33 defaults):
 * wav f32 [n_samples] (already scaled by 2^15) -> feats f32 [T, n_mel];
 * mel_fb f32 [n_mel, n_fft/2+1] from the host. */
int emoasr_fbank(const float* wav, long n_samples, int frame_len, int frame_shift, int n_fft, int n_mel,
                 float preemph, const float* window, const float* mel_fb, float* feats, int T,
                 void* stream);
/* (x - mean[f]) / std[f] in place, f32 [M,F] */
int emoasr_cmvn(int M, int F, float* x, const float* mean, const float* std, void* stream);

/* ---- CTC prefix beam search: per-frame bookkeeping (host) ------------------------------------------------- */
/* The prefix bookkeeping of CTCDecoder._beam_search / _merge_ctc_paths (asr/modeling/decoders/ctc.py:262-344, 372-397) as native
 * HOST code in IEEE doubles, bit-identical to the reference's python floats: extend every live prefix by blank / repeat and by the
 * frame's top-k labels, merge equal label sequences (path probabilities folded, the first path's LM / length scores kept), stable
 * sort by asr + lm + length bonus, keep beam_width.  The acoustic side (projection, log-soft-max, top-k) and the LM stay device
 * calls (emoasr_gemm_nt / emoasr_log_softmax / emoasr_topk; LM rows cached on the device, a frame brings only its k candidate
 * columns to the host).  emoasr_ctc_beam_step: row = the frame's log-probabilities f32 [V] (host), top = its k best labels (best
 * first; the blank is skipped), lm_lp = [live, k] LM log-probabilities of those labels after each live prefix or NULL;
 * parent / tok [beam_width] receive how each surviving prefix was formed (index into the previous live set; appended label or
 * -1 for an unchanged prefix).  -> number of live prefixes (< 0: error).  A search starts with the single prefix (<eos>). */
typedef struct emoasr_ctc_beam emoasr_ctc_beam_t;
emoasr_ctc_beam_t* emoasr_ctc_beam_new(int beam_width, int blank, int eos, double len_weight, double lm_weight);
void emoasr_ctc_beam_free(emoasr_ctc_beam_t* h);
int emoasr_ctc_beam_size(const emoasr_ctc_beam_t* h);
int emoasr_ctc_beam_step(emoasr_ctc_beam_t* h, const float* row, const int* top, int k, const double* lm_lp, int* parent, int* tok);
int emoasr_ctc_beam_prefix(const emoasr_ctc_beam_t* h, int i, int* toks, int cap);   /* -> length of live prefix i */
int emoasr_ctc_beam_scores(const emoasr_ctc_beam_t* h, double* total);              /* -> number of live prefixes */

#ifdef __cplusplus
}
#endif
#endif
