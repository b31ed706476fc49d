/*
 * mudiff_hip.h - C ABI of libmudiff_hip.so: the MI355X (gfx950) kernels behind MU-Diff's
 * dual-generator reverse-diffusion sampling path.
 *
 * The reference has no FFI for this path except two pybind functions (utils/op/upfirdn2d.cpp:20-31,
 * utils/op/fused_bias_act.cpp:18-28); everything else is torch ops called from Python.  The entry
 * points below are what a binding for the path binds instead; each one names the reference code it
 * replaces (paths relative to the reference checkout).  Plain pointers and sizes only - no torch
 * types.  All pointers are DEVICE pointers unless a parameter says "host".  Every call enqueues on
 * `stream` (a hipStream_t passed as void*; NULL = the default stream), allocates nothing and never
 * synchronises, so a caller may capture any sequence of calls into a hipGraph.
 *
 * Activation layout: NHWC fp32, described as a *view* (ptr, B, H, W, C, ld): element (b,y,x,c) lives
 * at ptr[((b*H + y)*W + x)*ld + c] with ld >= C, so a channel slice of a wider tensor (the
 * concatenations of the U-Net) is a view and never a copy.  With C == 1 (the generator inputs and
 * outputs) NHWC and the reference's NCHW coincide.
 *
 * Return value: 0 on success, a MUD_ERR_* code otherwise; mud_last_error() gives the text.
 */
#ifndef MUDIFF_HIP_H
#define MUDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUD_OK 0
#define MUD_ERR_ARG 1      /* bad shape / alignment / null pointer                 */
#define MUD_ERR_LAUNCH 2   /* hipGetLastError() after a launch reported a failure  */
#define MUD_ERR_UNSUPPORTED 3

#define MUD_ACT_NONE 0
#define MUD_ACT_SIGMOID 1
#define MUD_ACT_TANH 2
#define MUD_ACT_SILU 3
#define MUD_ACT_LRELU 4      /* LeakyReLU(0.2): the critic's activation (backbones/discriminator.py:178) */

#define MUD_PRO_NONE 0         /* A operand used as stored                                   */
#define MUD_PRO_AFFINE 1       /* a[b,c]*x + s[b,c]            (GroupNorm, AttnBlockpp)      */
#define MUD_PRO_AFFINE_SILU 2  /* silu(a[b,c]*x + s[b,c])      (AdaGN + SiLU of the ResBlock) */
#define MUD_PRO_LRELU 3        /* lrelu_0.2(x), no affine      (DownConvBlock of the critic)  */

/* arithmetic plan of one mud_conv2d_mfma launch (mud_conv_args.prec); the weights must have been packed for the same plan */
#define MUD_PREC_16X3 0        /* every product as hi*hi + hi*lo + lo*hi on the 16-bit MFMA (fp16 pieces): ~2^-22 per product */
#define MUD_PREC_FP8X 1        /* hi*hi on the fp16 MFMA + both cross terms on the block-scaled e4m3 MFMA: ~2^-15 per product, */
                               /* 0.78x the matrix cycles; 3x3 launches that fill the chip (mud_conv2d_mfma_prec_supported)     */

int mud_version(void);
const char* mud_last_error(void);
/* "" for the shipped build; an experiment build (scripts/build_variants.py) reports the -D flags it was compiled with, so that a
 * library can always be asked what it is (mudiff_hip.load() refuses anything else unless MUDIFF_ALLOW_VARIANT=1). */
const char* mud_build_flags(void);
/* bytes of device workspace the calls below need at most for a given problem: see each call */

/* ---- L3: Gaussian posterior / forward diffusion (engine/test.py:126-177, engine/train.py:256-281)
 * out = 0.5*((c1[t]*x01 + c2[t]*xt) + (c1[t]*x02 + c2[t]*xt)) + (t != 0) * std[t] * noise
 * x02 == NULL gives the single-predictor sample_posterior (engine/test.py:126-147).
 * std[t] = exp(0.5*posterior_log_variance_clipped[t]) is a host-made table (engine/test.py:143,173).
 * The arithmetic is un-contracted fp32 mul/add in the reference's order: results are bit-identical
 * to the PyTorch-CPU path.  t is int64 [B]; indices are clamped to [0, ntab). */
int mud_posterior_sample(const float* x01, const float* x02, const float* xt, const float* noise,
                         const int64_t* t, const float* coef1, const float* coef2, const float* std_tab,
                         int ntab, float* out, int B, int64_t per_sample, void* stream);
/* out = a_tab[t+toff]*x + s_tab[t+toff]*noise   (q_sample: a_s_cum/sigmas_cum, toff 0;
 * second half of q_sample_pairs: a_s/sigmas, toff 1).  engine/train.py:256-281. */
int mud_q_sample(const float* x, const float* noise, const int64_t* t, int toff, const float* a_tab,
                 const float* s_tab, int ntab, float* out, int B, int64_t per_sample, void* stream);

/* ---- embeddings and the small dense path (backbones/layers.py:465-479, dense_layer.py:67-71,
 *      ncsnpp_generator_adagn_feat.py:44-49,271-277,301-305; layerspp.py:42,277) */
int mud_timestep_embedding(const int64_t* t, float* out, int B, int dim, float max_positions, void* stream);
int mud_pixel_norm(const float* z, float* out, int B, int K, void* stream);
/* out[b, n] = act_out( sum_k W[n,k] * act_in(in[b,k]) + bias[n] ),  W row-major [N,K] (nn.Linear). */
int mud_dense(const float* in, int ldi, const float* W, const float* bias, float* out, int ldo,
              int B, int K, int N, int act_in, int act_out, void* stream);

/* A chain of dense layers in one launch (one workgroup per sample):  h = [pixel_norm](x);  for l: h = W[l] h + b[l], with
 * `act` applied between layers (and after the last one iff act_last).  The z-mapping network (PixelNorm, dense(nz, z_emb),
 * SiLU, n_mlp x [dense, SiLU]; ncsnpp_generator_adagn_feat.py:44-49,271-277) is {pixel_norm=1, act=SILU, act_last=1}; the
 * timestep MLP (:301-305: Linear, SiLU, Linear) is {act=SILU, act_last=0}.  W[l]: [dims[l+1], dims[l]] row-major (nn.Linear). */
#define MUD_MLP_MAX_LAYERS 6
typedef struct mud_mlp_args {
  const float* x; int ldx; int B;
  int nlayers; int dims[MUD_MLP_MAX_LAYERS + 1];
  const float* W[MUD_MLP_MAX_LAYERS]; const float* b[MUD_MLP_MAX_LAYERS];
  int pixel_norm; int act; int act_last;
  float* out; int ldo;
  int maxdim;                                    /* filled in by the library */
} mud_mlp_args;
int mud_mlp_chain(const mud_mlp_args* a, void* stream);
/* n (<= 4) independent chains a[0..n) in one launch (the z-mapping network and the timestep MLP of a generator do not depend on  */
/* each other).                                                                                                                 */
int mud_mlp_chains(const mud_mlp_args* a, int n, void* stream);

/* ---- GroupNorm statistics -> per-(sample, channel) scale/shift for a consumer's prologue
 *      (torch native_group_norm as used by backbones/layerspp.py:37-65,103, eps 1e-6, biased var).
 * scale[b,c] = gamma[b,c] * rstd[b,g(c)],  shift[b,c] = beta[b,c] - mean[b,g(c)] * scale[b,c]
 * gamma/beta: NULL (=1/0), per channel (g_bstride 0) or per sample (g_bstride = row stride).
 * ws: device workspace of mud_gn_ws_bytes(B, HW, C, G) bytes. */
int64_t mud_gn_ws_bytes(int B, int64_t HW, int C, int G);
int mud_gn_scale_shift(const float* x, int B, int64_t HW, int C, int ld, int G, float eps,
                       const float* gamma, const float* beta, int64_t g_bstride,
                       float* scale, float* shift, int ld_ss, float* mean_rstd /* [B,G,2] or NULL */,
                       void* ws, void* stream);
/* Same result from per-channel (sum, sumsq) already accumulated by the producer's epilogue
 * (mud_conv_args.stats, mud_gate_mix): sums[(b*sums_ld + c)*2 + {0,1}], `count` = pixels per channel. */
int mud_gn_scale_shift_from_sums(const double* sums, int sums_ld, int B, int C, int G, double count, float eps,
                                 const float* gamma, const float* beta, int64_t g_bstride,
                                 float* scale, float* shift, int ld_ss, void* stream);
/* out[b,c] = mean over pixels (nn.AdaptiveAvgPool2d(1), layerspp.py:473,491). ws as above with G=C. */
int mud_channel_mean(const float* x, int B, int64_t HW, int C, int ld, float* out, int ldo, void* ws, void* stream);

/* ---- convolutions (torch F.conv2d call sites: layers.py:104-128, layerspp.py:275-285,399-408,
 *      ncsnpp_generator_adagn_feat.py:267,620-631; NIN einsum layers.py:502-505; the attention
 *      contractions layerspp.py:118-122) */
typedef struct mud_conv_args {
  const float* x;  int B, H, W, Cin, ldx;        /* input view                                      */
  const void* w;   int64_t w_bstride;            /* weights (format depends on the call); bytes     */
                                                 /* between per-sample weight sets, 0 = shared       */
  int ks, stride, pad;                           /* square kernel                                    */
  const float* pro_scale; const float* pro_shift; int pro_ld; int pro_mode;   /* [B,Cin] each        */
                                                 /* (mud_conv2d_mfma keeps them in LDS: Cin <= 1024  */
                                                 /* with pro_mode AFFINE / AFFINE_SILU)               */
  const float* bias;                             /* [Cout] or NULL                                   */
  const float* bias2; int bias2_ld;              /* [B,Cout] or NULL  (Dense_0(act(temb)))           */
  const float* res; int ldr;                     /* residual view [B,Ho,Wo,Cout] or NULL             */
  float out_scale; int act;                      /* v = act((acc+bias+bias2+res)*out_scale)          */
  const float* emul; int ld_emul;                /* optional: v *= emul[pixel, co]                   */
  const float* egate; int ld_egate;              /* optional gated mix (G2 feature fusion, ...feat.py */
  const float* eother; int ld_eother;            /* :779-788): v = egate*v + (1-egate)*eother         */
  float* out; int Cout, ldo;                     /* output view [B,Ho,Wo,Cout]                       */
  int sub2;                                      /* mud_conv2d_mfma, ks 3 only: compute the stride-1  */
                                                 /* pad-1 result and keep only odd (y,x) positions as */
                                                 /* out[(y-1)/2,(x-1)/2] == stride-2 pad-0 convolution */
                                                 /* of the input (the FIR'd pyramid, odd H and W)      */
  double* stats; int stats_ld;                   /* optional: per-(b, channel) running (sum, sum of   */
                                                 /* squares) of the STORED outputs, stats[(b*stats_ld */
                                                 /* + co)*2 + {0,1}] += ... (fp64 atomics); lets the   */
                                                 /* next GroupNorm skip its pass over the tensor       */
  int emul_cout;                                 /* `emul` applies to output channels < emul_cout only */
                                                 /* (0 = all): lets two convs that share their input   */
                                                 /* but differ in this epilogue run as one launch       */
  /* mud_conv2d_mfma only - GroupNorm finalisation folded into the prologue (pro_mode AFFINE / AFFINE_SILU): when gn_sums  */
  /* is set, pro_scale / pro_shift are ignored and every workgroup forms scale = gamma*rstd, shift = beta - mean*scale of   */
  /* its sample from the producer-accumulated per-channel (sum, sumsq) itself - the arithmetic of                           */
  /* mud_gn_scale_shift_from_sums, without its launch (layerspp.py:37-54 AdaptiveGroupNorm, :56-65 GroupNorm_Conv).         */
  const double* gn_sums; int gn_sums_ld;         /* gn_sums[(b*gn_sums_ld + c)*2 + {0,1}], c < Cin                           */
  int gn_G; float gn_eps; double gn_count;       /* groups (Cin % gn_G == 0), eps, pixels per channel                        */
  const float* gn_gamma; const float* gn_beta;   /* [Cin] (gn_bstride 0) or [B, .] rows gn_bstride floats apart, or NULL      */
  int64_t gn_bstride;
  /* mud_conv2d_mfma, ks 3 only - optional split-K workspace: when a launch would fill less than the chip (one slice at a   */
  /* time), the K chunks of a tile are dealt to several workgroups that write raw partial tiles here, and a second small     */
  /* launch adds them in a fixed order and applies the epilogue.  mud_conv2d_mfma_splitk_bytes() sizes it; NULL = never split. */
  void* splitk_ws; int64_t splitk_ws_bytes;
  /* optional arrival counters (splitk_ncounters of them, ZERO when handed over; every launch leaves them zero): with them the   */
  /* last workgroup to finish an output tile adds the slabs (same fixed order) and applies the epilogue, so a split convolution   */
  /* is one launch instead of two.  One array per stream: launches that may run concurrently must not share it.                   */
  unsigned* splitk_counters; int splitk_ncounters;
  /* mud_conv2d_mfma, ks 3, pro_mode AFFINE_SILU, plain epilogue - the residual block's 1x1 skip convolution of the RAW input     */
  /* (layerspp.py:320-321, x = Conv_2(x)) produced by the same launch: skip_out[pixel, co] = sum_ci x[pixel, ci] * skip_w + bias. */
  /* skip_w: mud_pack_weights(ks = 1) of the [Cout, Cin] matrix; x is then read from HBM once instead of twice.  Cin <= 512.      */
  const void* skip_w; const float* skip_bias; float* skip_out; int skip_ldo;
  /* mud_conv2d_mfma - arithmetic plan of this launch (MUD_PREC_*; 0 = the default) and, for MUD_PREC_FP8X, the power-of-two         */
  /* exponent the weights' e4m3 image was packed with (mud_pack_weights_prec: the largest e with max|w| * 2^e <= 448).               */
  int prec; int w_exp;
} mud_conv_args;

/* Exact fp32 direct convolution (FMA chain per output), any ks/stride/pad/Cin/Cout.
 * w: fp32 [ks][ks][Cin][Cout].  Used for Cin==1 heads, Cout==1 tail and strided convs. */
int mud_conv2d_direct(const mud_conv_args* a, void* stream);

/* Implicit-GEMM convolution on the matrix cores, ks in {1,3}, stride 1, pad ks/2.
 * fp32 operands are split on the fly into fp16 hi+lo (saturating at +-65504) and multiplied as hi*hi + hi*lo + lo*hi with
 * fp32 accumulation (v_mfma_f32_32x32x16_f16 x3): ~2^-22 relative error per product; a.prec selects the cheaper plan.
 * w: packed by mud_pack_weights().  Requires Cin % 4 == 0, ldx % 4 == 0, 16-byte aligned x.
 * For ks == 1 the (H, W) plane is treated as one flat axis of H*W positions (plain GEMM). */
int64_t mud_packed_weight_bytes(int ks, int Cin, int Cout);
/* src element (tap, ci, co) = src[tap*s_tap + ci*s_ci + co*s_co] (+ b*src_bstride elements).
 *   OIHW conv weight:  s_tap=1, s_ci=ks*ks, s_co=Cin*ks*ks;   NIN W[in,out]: s_ci=Cout, s_co=1;
 *   K^T of attention: rows of K as "co": s_ci=1, s_co=ldk;   V: s_ci=ldv, s_co=1.               */
int mud_pack_weights(const float* src, int64_t s_tap, int64_t s_ci, int64_t s_co, int64_t src_bstride,
                     int ks, int Cin, int Cout, int nbatch, void* dst, void* stream);
/* The same for a given arithmetic plan: MUD_PREC_FP8X (ks == 3 only) writes fp16 hi planes and the two e4m3 images
 * (w * 2^w_exp, (w - fp16(w)) * 2^(w_exp + 11)) in place of the lo planes; same size. */
int mud_pack_weights_prec(const float* src, int64_t s_tap, int64_t s_ci, int64_t s_co, int64_t src_bstride,
                          int ks, int Cin, int Cout, int nbatch, int prec, int w_exp, void* dst, void* stream);
int mud_conv2d_mfma(const mud_conv_args* a, void* stream);
/* 1 when mud_conv2d_mfma has plan `prec` for this launch (sizes, prologue mode, skip_w, sub2 are looked at), else 0. */
int mud_conv2d_mfma_prec_supported(const mud_conv_args* a, int prec);
/* Bytes of split-K workspace mud_conv2d_mfma would use for this call (0: the launch is not split). */
int64_t mud_conv2d_mfma_splitk_bytes(const mud_conv_args* a);

/* ---- FIR resampling (utils/op/upfirdn2d.cpp:20-31 + upfirdn2d_kernel.cu:109-209; python front
 *      ends backbones/up_or_down_sampling.py:149-262).
 * Plane form = the reference's pybind op: input [planes, H, W] (its [major, in_h, in_w, minor=1]),
 * kernel fp32 [kh, kw] on the device, same argument meaning as upfirdn2d(...). */
int mud_upfirdn2d(const float* in, int64_t planes, int H, int W, const float* kernel, int kh, int kw,
                  int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                  float* out, void* stream);
/* NHWC form used inside the generators: same filter maths on a view, optionally producing in ONE
 * pass both FIR(prologue(x)) (out_h) and FIR(x) (out_x) as ResnetBlockBigGANpp_Adagn needs
 * (layerspp.py:293-308).  kernel: host pointer to kh*kw floats (<= 64). Either output may be NULL. */
int mud_fir_nhwc(const float* x, int B, int H, int W, int C, int ldx, const float* kernel_host, int kh, int kw,
                 int up, int down, int pad0, int pad1,
                 const float* pro_scale, const float* pro_shift, int pro_ld, int pro_mode,
                 float* out_h, int ldh, float* out_x, int ldxo, void* stream);

/* ---- minibatch standard deviation of the critic (backbones/discriminator.py:246-254, stddev_feat = 1):
 * x view [B, hw, C]; samples b = g*M + m (g < group, M = B/group); out[b] = mean_{c,p} sqrt(var_g(x[g*M+m,p,c]) + 1e-8) */
int mud_minibatch_stddev(const float* x, int B, int64_t hw, int C, int ld, int group, float* out, void* stream);

/* ---- fused attention (layerspp.py:118-122): out[b,i,:] = sum_j softmax_j(q_i.k_j * scale) v_j, single head.
 * qkv: [B, N, ld] with q at +0, k at +C, v at +2C (the fused NIN_0|1|2 output); out [B, N, ldo].
 * Flash-style (no N x N matrix in memory), split-fp16 MFMA.  Head dims: see mud_attention_supported().
 * When batch x ceil(N/128) workgroups would leave most CUs idle (single-slice latency case) the keys are split over
 * up to 16 workgroups per query block and merged by a second tiny kernel; that needs `ws` (mud_attention_ws_bytes(),
 * 16-byte aligned).  ws == NULL always runs unsplit. */
int mud_attention_supported(int C);
int64_t mud_attention_ws_bytes(int B, int N, int C);   /* 0 = no workspace needed for this problem */
int mud_attention(const float* qkv, int B, int N, int C, int ld, float scale, float* out, int ldo, void* ws, void* stream);

/* ---- unfused attention pieces (used when the head dim is not supported above) and the G2 feature fusion (…feat.py:769-788) */
int mud_softmax_rows(float* s, int64_t rows, int n, int ld, void* stream);          /* in place */
int mud_mul(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int64_t npix, int C, void* stream);
/* out = g*att + (1-g)*other; views are [B, hw, C]; optional per-channel stats of `out` as in mud_conv_args */
int mud_gate_mix(const float* g, int ldg, const float* att, int lda, const float* other, int ldb,
                 float* out, int ldo, int B, int64_t hw, int C, double* stats, int stats_ld, void* stream);

/* ---- rows f1 / f3: bilinear resize of planes [P,H,W] -> [P,Ho,Wo] with torch F.interpolate(mode='bilinear',
 *      align_corners=False) semantics (engine/test_volume.py:274 slice -> image_size; engine/train.py:959 uncertainty map),
 *      and out = clamp(x*scale + shift, lo, hi) (the [-1,1] -> [0,1] mapping, engine/test_volume.py:281). */
/* row f4: Gaussian Fourier features of log(t) (layerspp.py:68-77, ncsnpp_generator_adagn_feat.py:288-289):
 * out[b, k] = sin(2*pi*W[k]*log(t[b])), out[b, n+k] = cos(...); out is [B, 2n]. */
int mud_fourier_embedding(const float* t, const float* W, float* out, int B, int n, void* stream);
int mud_resize_bilinear(const float* in, int64_t planes, int H, int W, int Ho, int Wo, float* out, void* stream);
int mud_affine_clamp(const float* x, int64_t n, float scale, float shift, float lo, float hi, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
