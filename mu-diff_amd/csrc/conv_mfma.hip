// Implicit-GEMM convolution / GEMM on the CDNA4 matrix cores with fp32-class accuracy.
//
//   out[p, co] = sum_{tap, ci} f(x[p + off(tap), ci]) * w[tap, ci, co]         (ks = 3, pad 1 | ks = 1)
//
// gfx950 has no TF32; exact-fp32 MFMA runs at the VALU rate (157 TF), 1/16 of the 16-bit rate.  The
// path needs fp32-class results (<= 1e-3 after ~50 layers), so each fp32 operand is split into two
// fp16 terms (hi = rne(v), lo = rne(v - hi); mud_common.h) and every product is issued as
//      lo*hi + hi*lo + hi*hi          (3 x v_mfma_f32_32x32x16_f16, fp32 accumulate)
// i.e. ~2^-22 relative error per product at 16/3 = 5.3x the fp32-MFMA rate; a second plan (MUD_PREC_FP8X, below) issues the
// two cross terms on the block-scaled e4m3 MFMA instead: 2^-15 per product at 0.78x the matrix cycles.
//
// Structure (one 256-thread workgroup = 4 waves, template <KS, MT>):
//   * output tile: (4*MT) rows x 32 columns of pixels (ks=3) or 128*MT flat positions (ks=1) x 64
//     channels; wave w owns MT rows -> MT x 2 MFMA tiles of 32x32 (MT = 4: 128 accumulator regs).
//   * A operand (activations): the (rows+2) x 34 halo tile of one 16-channel K chunk lives in LDS as
//     80-byte pixel records [hi 16 x fp16 | lo 16 x fp16 | 16 B pad], so every MFMA A fragment is one
//     ds_read_b128 at base + (compile-time tap/row offset): the 80-B stride spreads the 16 lanes of a
//     b128 read group over 16 distinct 16-B bank slots (conflict-free) with NO per-tap address maths.  It is transformed ONCE on the way in (GroupNorm/AdaGN affine, SiLU, hi/lo split)
//     and reused by 9 taps x 64 output channels.  Double-buffered: the raw fp32 values of chunk k+1
//     are fetched into registers before the MFMAs of chunk k and written to the other LDS buffer
//     after them -> one barrier per chunk, global latency hidden under the matrix work.
//   * B operand (weights): pre-split, pre-packed on the host side of the ABI into exactly the MFMA
//     fragment order (16-B halves pre-swizzled).  Groups of 3 taps (12 KiB) are streamed global -> LDS by
//     direct-to-LDS loads (global_load_lds_dwordx4: no VGPRs, each wave moves a quarter) into a 2-slot
//     ring one group ahead of use and read back by all four waves with ds_read_b128: 4x less L1/TA
//     traffic than per-wave fragment loads (measured: the TA path, not L2 capacity, was the limiter).
//   * epilogue from the accumulators: + bias[co] + bias2[b,co] (time embedding) + residual, * scale,
//     activation; each half-wave stores 32 consecutive channels (128 B) of one pixel.
//   * workgroup ids are remapped so that the workgroups sharing an input tile (different output-channel
//     tiles) and spatial neighbours run on the same XCD (shared L2).
#include "mud_common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- arithmetic plans of a launch (mud_conv_args.prec); 16-bit pieces: mud_common.h (fp16 hi + lo)
//   MUD_PREC_16X3  every product as lo*hi + hi*lo + hi*hi on the fp16 MFMA (3 x v_mfma_f32_32x32x16_f16): ~2^-22 per product.
//   MUD_PREC_FP8X  hi*hi on the fp16 MFMA + the two cross terms of a 3-tap group as TWO block-scaled e4m3 MFMAs
//                  (v_mfma_scale_f32_32x32x64_f8f6f4, K = 64 = (3 taps + a zero tap) x 16 channels, 2x the 16-bit rate):
//                  288 -> 224 matrix cycles per 3-tap group and accumulator.  The cross terms carry 2^-11 of a product, so 4
//                  significant bits per operand keep the total at ~2^-15.  3x3 kernel only; same LDS images and weight steps:
//                  pixel record [f16 hi 32 B][e4m3 a*2^SA 16 B][e4m3 a_lo*2^SAL 16 B][16 B zero = the zero tap],
//                  weight step  [f16 hi 2 KiB][e4m3 w*2^w_exp 1 KiB][e4m3 w_lo*2^(w_exp+11) 1 KiB]  (w_exp: per layer, chosen at pack time).
typedef mud_h16x4 h16x4;
typedef mud_h16x8 h16x8;
#define CM_X_SA 2                // constant power-of-two pre-scales of the e4m3 activation images (undone by the MFMA's E8M0 scale operands):
#define CM_X_SAL 13              // a*2^2 covers |a| in [5e-4, 112]; a_lo <= 2^-11 |a| -> a_lo*2^13 <= 448 as well.  Out-of-range values only lose their cross term
__device__ __forceinline__ int cm_e4m3x4(f32x4 v, float scale) {    // 4 floats * scale -> 4 packed OCP e4m3 bytes (hardware converter)
  // v_cvt_pk_fp8_f32 does NOT saturate: |x| >= 480 comes out as NaN (scripts/mfma_f8_layout.hip), so out-of-range values are clamped to +-448 first
  v = v * scale;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -448.0f, 448.0f);
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], p, true);
}
#define CM_BN 64
#define CM_BPLANE (CM_BN * 32)   // bytes of one [64 co][16 ci] 16-bit plane
#define CM_BSTEP (2 * CM_BPLANE) // hi plane + lo plane of one (k16 chunk, tap)
#define CM_PIX 80                // LDS bytes per pixel record of the A tile
#define CM_GN_MAXC 1024         // most channels the folded GroupNorm finalisation takes (scale | shift arrays in LDS: up to 8 KiB)
#define CM_GN_BYTES (2 * CM_GN_MAXC * 4)

template <int KS, int MT, int WM, int WN, bool DUAL = false>
struct CmGeo {
  static constexpr int NT = 64 * WM * WN;               // threads per workgroup
  static constexpr int TAPS = KS * KS;
  static constexpr int CH = (KS == 3) ? 1 : 2;          // k16 steps per LDS chunk
  static constexpr int KCH = 16 * CH;                   // input channels per LDS chunk
  static constexpr int ROWS = WM * MT;
  static constexpr int PW = 32 + KS - 1;
  static constexpr int P = (KS == 3) ? (ROWS + 2) * PW : 32 * WM * MT;
  static constexpr int PLANE = P * CM_PIX;              // P pixel records [hi 32 B | lo 32 B | pad 16 B]
  static constexpr int BUF = CH * PLANE;                // [k16 s]
  static constexpr int Q = 4 * CH;                      // float4 per pixel per chunk
  static constexpr int ITEMS = P * Q;
  static constexpr int NLOAD = (ITEMS + NT - 1) / NT;
  static constexpr int STEPS = TAPS * CH;               // MFMA steps (tap, s) per chunk
  static constexpr int GS = (KS == 3) ? 3 : 2;          // steps per B group (one DMA batch, one barrier)
  static constexpr int NG = STEPS / GS;                 // B groups per chunk
  static constexpr int GB1 = GS * CM_BSTEP;             // bytes per B group of ONE 64-channel tile (contiguous in the packed weights)
  static constexpr int GB = WN * GB1;                   // bytes per B group of the workgroup (WN tiles)
  // DMA pieces of a group (1 KiB = 64 lanes x 16 B each): N4 per wave, and the first REM waves one more (12 KiB over 8 waves: 2 | 1.
  // Measured against equal counts with 256-byte pieces for the remainder - 1 KiB + 2 x 256 B per wave - the fewer instructions win by 2 %
  // on those tiles; the 12-byte form, 768 B per instruction, leaves a 4-byte hole behind every lane's 12 bytes: scripts/dma_x3_probe.hip)
  static constexpr int N4 = GB / (1024 * WM * WN), REM = GB / 1024 - N4 * WM * WN;
  static_assert(GB % 1024 == 0 && GB1 % 1024 == 0, "whole 1 KiB pieces, none straddling two tiles");
  static constexpr int A_BYTES = 2 * BUF;
  static constexpr int B_OFF = A_BYTES;                 // B ring: 2 groups
  static constexpr int EP_BYTES = WM * WN * (32 * 36 * 4 + 64 * 2 * 4);   // epilogue patches + statistics
  // DUAL (3x3 conv + the block's 1x1 skip conv of the RAW input from one staging pass, see k_conv_mfma): a second, single-
  // buffered A image of the tile's centre pixels [hi 32 B | lo 32 B] (64-B records, 16-B units XOR-swizzled by the column)
  // and one slot for the 1x1 weights of the current chunk.  The GroupNorm arrays shrink to 512 channels to make room.
  static constexpr int GN_MAXC = DUAL ? 512 : CM_GN_MAXC;
  static constexpr int A2_OFF = A_BYTES + 2 * GB;
  static_assert(!DUAL || A2_OFF % 64 == 0, "the raw image's hi / lo halves are addressed by flipping bit 5 of the offset");
  static constexpr int A2_BYTES = DUAL ? ROWS * 32 * 64 : 0;
  static constexpr int B2_OFF = A2_OFF + A2_BYTES;
  static constexpr int B2_BYTES = DUAL ? WN * CM_BSTEP : 0;
  // folded GroupNorm finalisation: scale | shift of the sample, 2 x roundup4(Cin) floats at the END of the image, requested
  // per launch (a fixed 8 KiB here would push the 8-row 4-wave tile over 80 KiB = from two workgroups per CU to one)
  static constexpr int GN_OFF = B2_OFF + B2_BYTES;
  static constexpr int LDS_MAX = (GN_OFF + 2 * GN_MAXC * 4) > EP_BYTES ? (GN_OFF + 2 * GN_MAXC * 4) : EP_BYTES;
  static constexpr int lds_bytes(int gn_channels) {
    const int m = GN_OFF + 8 * ((gn_channels + 3) & ~3);
    return m > EP_BYTES ? m : EP_BYTES;
  }
  static_assert(LDS_MAX <= 160 * 1024, "LDS budget");
  static_assert(NT % Q == 0, "staging split");
};



// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant in the body
template <class F, int... I>
__device__ __forceinline__ void cm_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void cm_static_for(F&& f) { cm_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ bool mud_dev_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }


// GroupNorm finalisation folded into the conv prologue (mud_conv_args.gn_*): scale / shift of sample b for all Cin channels
// into LDS, same arithmetic and summation order as k_gn_from_sums (groupnorm.hip) - one thread per channel, the group's
// (sum, sumsq) re-added by each of its channels (a few L1-resident fp64 pairs).
__device__ __forceinline__ void cm_gn_to_lds(const mud_conv_args& a, int b, int tid, int nthreads, float* sc_lds, int maxc) {
  float* sh_lds = sc_lds + maxc;
  const int cpg = a.Cin / a.gn_G;
  for (int c = tid; c < a.Cin; c += nthreads) {
    const int g0 = (c / cpg) * cpg;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < cpg; ++i) {
      const double* p = a.gn_sums + ((int64_t)b * a.gn_sums_ld + g0 + i) * 2;
      s += p[0];
      q += p[1];
    }
    const double n = a.gn_count * cpg, mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.gn_eps)), meanf = (float)mean;
    const float ga = a.gn_gamma ? a.gn_gamma[(int64_t)b * a.gn_bstride + c] : 1.0f;
    const float be = a.gn_beta ? a.gn_beta[(int64_t)b * a.gn_bstride + c] : 0.0f;
    const float sc = ga * rstd;
    sc_lds[c] = sc;
    sh_lds[c] = be - meanf * sc;
  }
}

__device__ __forceinline__ float cm_fast_silu(float v) {
  // v * sigmoid(v) with the hardware exp2 / rcp (each ~1 ulp): the result is rounded to fp16 hi+lo (2^-22) anyway
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}

// One float4 of raw activations -> its LDS pieces (prologue, hi / lo split, e4m3 images).  SCALAR f32 arithmetic on purpose: beside
// MFMAs a packed f32 instruction (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32) costs ~4x two plain ones (MI355X_MICROARCH.md, constants
// table, "price of one filler beside MFMAs"), and hipcc packs every float4 expression and every pair of adjacent scalar ones it can
// find - so the staging code works on elements and this file is compiled with -fno-slp-vectorize (csrc/Makefile).
// Saturating converters: with MODE.FP16_OVFL set, v_cvt_pk_f16_f32 clamps to +-65504 instead of producing inf and
// v_cvt_pk_fp8_f32 to +-448 instead of NaN (scripts/fp16_ovfl_probe.hip, profiles/r03_n_fp16_ovfl_probe.txt) - so the staging code
// needs no v_med3 clamp per element (12 of its ~62 vector instructions per float4).  Set once per wave at kernel entry; the mode
// touches only instructions that PRODUCE fp16 / fp8 values, and the kernels have no others.
__device__ __forceinline__ void cm_saturating_converters() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }

template <int PRO, bool X8>
__device__ __forceinline__ void cm_stage4(const f32x4& rw, const f32x4& sc, const f32x4& sh, float keep, h16x4& hi, h16x4& lo, int& a8, int& al8) {
  f32x4 v, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = rw[e];
    if (PRO == MUD_PRO_AFFINE || PRO == MUD_PRO_AFFINE_SILU) {
      x = __builtin_fmaf(x, sc[e], sh[e]);
      if (PRO == MUD_PRO_AFFINE_SILU) x = cm_fast_silu(x);
    } else if (PRO == MUD_PRO_LRELU) {
      x = x > 0.f ? x : 0.2f * x;
    }
    v[e] = x * keep;                           // zero padding stays zero (the fp16 pieces saturate in the converter: cm_saturating_converters)
  }
  hi = __builtin_convertvector(v, h16x4);      // (2 x v_cvt_pk_f16_f32: a conversion, not packed arithmetic)
#pragma unroll
  for (int e = 0; e < 4; ++e) l[e] = v[e] - (float)hi[e];
  if constexpr (X8) {
    // v_cvt_scalef32_pk_fp8_f32 divides by its power-of-two scale operand on the way (scripts/scalef32_probe.hip: cvt(x / scale)):
    // the pre-scales of the two images cost no multiply; out-of-range values saturate at +-448 (cm_saturating_converters)
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    constexpr float ia = 1.0f / (float)(1 << CM_X_SA), ial = 1.0f / (float)(1 << CM_X_SAL);
    const s16x2 z = {0, 0};
    const s16x2 av = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, v[0], v[1], ia, false), v[2], v[3], ia, true);
    const s16x2 lv = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, l[0], l[1], ial, false), l[2], l[3], ial, true);
    a8 = __builtin_bit_cast(int, av);
    al8 = __builtin_bit_cast(int, lv);
  } else {
    lo = __builtin_convertvector(l, h16x4);
  }
}
// v = hi + lo in fp16 pieces, element by element (the scalar twin of mud_split4, for the same reason)
__device__ __forceinline__ void cm_split4s(const f32x4& rw, float keep, h16x4& hi, h16x4& lo) {
  f32x4 v, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = rw[e] * keep;
  hi = __builtin_convertvector(v, h16x4);
#pragma unroll
  for (int e = 0; e < 4; ++e) l[e] = v[e] - (float)hi[e];
  lo = __builtin_convertvector(l, h16x4);
}

// DUAL: the launch also produces the residual block's 1x1 skip convolution of the RAW input (reference layerspp.py:320-321,
// `x = self.Conv_2(x)`) from the same staged tile - skip_out = skip_w * x + skip_bias - so that x is read from HBM once instead
// of twice (the skip convs are pure HBM streams: 3.3-4.8 TB/s, 8 % of a forward) at the price of one more tap's MFMAs.
// split-K reduced inside the launch (`a` is then the REAL convolution): slab workspace + per-output-tile arrival counters.
// counters == NULL with nsplit > 1: `a` describes raw NHWC partial slabs and k_splitk_epilogue reduces them (second launch).
struct CmFin {
  float* ws;
  unsigned* counters;
};

template <int KS, int MT, int WM, int WN, int PRO, bool DUAL = false, int PREC = MUD_PREC_16X3>
__global__ __launch_bounds__(64 * WM * WN, (MT == 1 && WM * WN == 8) ? 4 : 2) void k_conv_mfma(mud_conv_args a, int tiles_x, int tiles_per_img, int ntiles, int k16s,
                                                                unsigned nblocks, int nsplit, int64_t split_stride, CmFin fin) {
  using G = CmGeo<KS, MT, WM, WN, DUAL>;
  constexpr bool X8 = PREC == MUD_PREC_FP8X;           // fp16 hi.hi + e4m3 cross terms (3x3 only)
  static_assert(!X8 || KS == 3, "the fp8 cross-term plan is built for the 3x3 kernel");
  using HV4 = h16x4;
  using HV8 = h16x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cm_saturating_converters();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;     // wave grid: WM along pixel rows, WN along 64-channel tiles

  // ---- XCD-aware bijective remap: consecutive logical ids share an XCD (blocks i, i+8 are co-resident on one XCD)
  unsigned lid;
  {
    const unsigned orig = blockIdx.x, xcd = orig & 7u, q = nblocks >> 3, rem = nblocks & 7u;
    lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
  }
  const int nt = lid % ntiles;                  // output-channel tile fastest: sharers of one A tile are neighbours
  const unsigned rest = lid / ntiles;
  const int tile = rest % tiles_per_img;
  const unsigned rest2 = rest / tiles_per_img;
  const int b = rest2 % (unsigned)a.B;
  const int ksi = rest2 / (unsigned)a.B;        // split-K slice (small grids only: nsplit > 1, see cm_splits)

  const int64_t HW = (int64_t)a.H * a.W;
  int ty0 = 0, tx0 = 0;
  int64_t flat0 = 0;
  if (KS == 3) {
    ty0 = (tile / tiles_x) * G::ROWS;
    tx0 = (tile % tiles_x) * 32;
  } else {
    flat0 = (int64_t)tile * G::P;
  }
  const float* xb = a.x + (int64_t)b * HW * a.ldx;
  const int64_t tile_bytes = (int64_t)k16s * (G::TAPS * CM_BSTEP);          // packed bytes of one 64-channel tile
  const char* wb = (const char*)a.w + (int64_t)b * a.w_bstride + (int64_t)nt * WN * tile_bytes;
  const float* psc = a.pro_scale + (int64_t)b * a.pro_ld;
  const float* psh = a.pro_shift + (int64_t)b * a.pro_ld;
  const int nchunks_all = (k16s + G::CH - 1) / G::CH;
  const int per_split = (nchunks_all + nsplit - 1) / nsplit;
  const int kc0 = ksi * per_split;                                           // this workgroup reduces chunks [kc0, nchunks)
  const int nchunks = (kc0 + per_split < nchunks_all) ? kc0 + per_split : nchunks_all;
  a.out += (int64_t)ksi * split_stride;                                      // its slab of raw partial sums (nsplit > 1)

  // ---- per-thread staging slots: pixel -> global element offset and LDS byte offset, fixed for all chunks.
  // Everything below is branch-free: padding pixels load a clamped (valid) address and are zeroed by a 0/1
  // mask; slots past the tile wrap around and redo another thread's item (identical value, same address).
  const int q = tid % G::Q;                     // this thread's float4 (4 channels) inside a chunk
  int goff[G::NLOAD], loff[G::NLOAD];
  int loff2[DUAL ? G::NLOAD : 1];               // DUAL: byte offset of the slot's hi half in the raw centre image, or -1 (halo pixel)
  unsigned vmask = 0;                           // bit j: slot j is a real (non-padding) pixel
#pragma unroll
  for (int j = 0; j < G::NLOAD; ++j) {
    const int item = (tid + j * G::NT) % G::ITEMS;   // NT % Q == 0 and ITEMS % Q == 0: q is preserved
    const int p = item / G::Q;
    bool valid;
    int64_t g;
    if (KS == 3) {
      const int py = p / G::PW, px = p - py * G::PW;
      const int gy = ty0 + py - 1, gx = tx0 + px - 1;
      valid = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      g = ((int64_t)gy * a.W + gx) * a.ldx;
      if (DUAL) {
        const bool centre = py >= 1 && py <= G::ROWS && px >= 1 && px <= 32 && tid + j * G::NT < G::ITEMS;   // (wrapped slots redo an item: skip)
        const int col = px - 1, sw = (col >> 2) & 3;
        loff2[j] = centre ? ((py - 1) * 32 + col) * 64 + ((((q & 3) >> 1) ^ sw) << 4) + (q & 1) * 8 : -1;
      }
    } else {
      const int64_t fp = flat0 + p;
      valid = fp < HW;
      g = fp * a.ldx;
    }
    goff[j] = valid ? (int)g : 0;
    vmask |= (valid ? 1u : 0u) << j;
    loff[j] = (q >> 2) * G::PLANE + p * CM_PIX + (q & 3) * 8;
  }

  f32x4 raw[G::NLOAD];
  // CNT: the activation loads of chunk k+1 (issued at the head of chunk k's group 0, consumed from group 1 on) stay IN FLIGHT across
  // group 0's barrier.  hipcc cannot do that by itself: with LDS-DMA and ordinary loads both pending its wait-count pass assumes they
  // return out of order and waits for vmcnt(0) at the first use - i.e. also for the weight DMA it issued a moment ago - and
  // __syncthreads() drains everything anyway.  So these loads are issued from inline assembly (invisible to that pass) and waited for
  // by hand: memory operations of a wave retire in order, so "at most N outstanding" with N = the operations issued AFTER them is exact.
  // Needs a compile-time number of DMA pieces per wave and group (dma_b below) and no second DMA stream (the fused skip conv).
  constexpr bool CNT = KS == 3 && !DUAL;        // (the fused-skip kernels sit at 250+ registers: measured 2-5 % slower with it)
  // (CNT) wait until at most N younger operations are outstanding, i.e. every activation load has landed; uses of raw[] stay behind the wait
#define CM_RAW_WAIT(N)                                                          \
  do {                                                                          \
    if constexpr (CNT) {                                                        \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");                 \
      _Pragma("unroll") for (int j_ = 0; j_ < G::NLOAD; ++j_) asm volatile("" : "+v"(raw[j_])); \
    }                                                                           \
  } while (0)
  f32x4 psc_r = {1.f, 1.f, 1.f, 1.f}, psh_r = {0.f, 0.f, 0.f, 0.f};   // prologue scale/shift of the chunk in `raw`
  const bool gn_fold = (PRO == MUD_PRO_AFFINE || PRO == MUD_PRO_AFFINE_SILU) && a.gn_sums != nullptr;   // workgroup-uniform
  const float* const gn_sc = (const float*)(smem + G::GN_OFF);
  const int gn_c = (a.Cin + 3) & ~3;            // shift array follows the scale array
  auto fetch_raw = [&](int chunk) {
    int c = chunk * G::KCH + q * 4;
    c = c < a.Cin ? c : 0;                      // clamped; zeroed in store_a
#pragma unroll
    for (int j = 0; j < G::NLOAD; ++j) {
      if constexpr (CNT) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[j]) : "v"(xb + goff[j] + c) : "memory");
      else raw[j] = *(const f32x4*)(xb + goff[j] + c);
    }
  };
  auto fetch_ss = [&](int chunk) {
    if (PRO == MUD_PRO_AFFINE || PRO == MUD_PRO_AFFINE_SILU) {
      int c = chunk * G::KCH + q * 4;
      c = c < a.Cin ? c : 0;
      // always from LDS (the caller's arrays are copied there by the prologue when the GroupNorm finalisation is not folded in): as a
      // choice between an LDS and a global pointer this becomes a FLAT load, which returns out of order - every wait behind it is vmcnt(0)
      psc_r = *(const f32x4*)(gn_sc + c);
      psh_r = *(const f32x4*)(gn_sc + gn_c + c);
    }
  };
  auto fetch_a = [&](int chunk) {
    fetch_ss(chunk);
    fetch_raw(chunk);                           // LAST: the NLOAD youngest loads of the wave at group 0's barrier (counted wait there)
  };
  auto store_a_slots = [&](int chunk, char* buf, int j0, int j1) {
    const bool cvalid = chunk * G::KCH + q * 4 < a.Cin;
#pragma unroll
    for (int j = 0; j < G::NLOAD; ++j) {
      if (j < j0 || j >= j1) continue;
      const float keep = (cvalid && ((vmask >> j) & 1u)) ? 1.0f : 0.0f;   // zero padding stays zero
      HV4 hi, lo;
      int a8 = 0, al8 = 0;
      cm_stage4<PRO, X8>(raw[j], psc_r, psh_r, keep, hi, lo, a8, al8);
      *(HV4*)(buf + loff[j]) = hi;
      if constexpr (X8) {                       // (loff holds record + 8 q)
        *(int*)(buf + loff[j] + 32 - q * 4) = a8;          // record + 32 + 4 q
        *(int*)(buf + loff[j] + 48 - q * 4) = al8;         // record + 48 + 4 q
      } else {
        *(HV4*)(buf + loff[j] + 32) = lo;
      }
      if (DUAL) {                               // the RAW value of the tile's centre pixels: A operand of the 1x1 skip conv (always 16-bit x 3)
        if (loff2[j] >= 0) {
          h16x4 rhi, rlo;
          cm_split4s(raw[j], cvalid ? 1.0f : 0.0f, rhi, rlo);
          char* a2 = smem + G::A2_OFF;
          *(h16x4*)(a2 + loff2[j]) = rhi;
          *(h16x4*)(a2 + (loff2[j] ^ 32)) = rlo;          // unit u -> u ^ 2: the lo half sits two 16-B units away under the same swizzle
        }
      }
    }
  };
  auto store_a = [&](int chunk, char* buf) { store_a_slots(chunk, buf, 0, G::NLOAD); };

  // ---- B operand: groups of GS steps are copied global -> LDS by direct-to-LDS loads (no VGPRs) into a 2-slot ring, one group
  // ahead of its use.  The packed layout already is the LDS image (16-B halves pre-swizzled for conflict-free ds_read_b128).
  // The number of pieces a wave moves per group is a compile-time fact (CmGeo: N4, one more in the first REM waves) and there is no
  // branch on the group index: past the slice's last group the last one is fetched again, into the ring slot nobody reads any
  // more.  So a wave's count of outstanding operations is known at every point of the loop (CNT above).
  const int total_groups_all = (k16s * G::TAPS + G::GS - 1) / G::GS;
  const int total_groups = (nchunks * G::NG < total_groups_all) ? nchunks * G::NG : total_groups_all;   // nothing is fetched past this slice
  char* const bring = smem + G::B_OFF;
  auto dma_b = [&](int gg) {                    // group gg -> ring slot gg & 1
    char* dst = bring + (gg & 1) * G::GB;
    gg = gg < total_groups ? gg : total_groups - 1;
#pragma unroll
    for (int j = 0; j < G::N4; ++j) {
      const int byte = (wave + WM * WN * j) * 1024;     // wave-uniform piece of the group image [WN tiles][GS steps][hi|lo][2 KiB]; no piece straddles two tiles
      const char* src = wb + (byte / G::GB1) * tile_bytes + (int64_t)gg * G::GB1 + byte % G::GB1 + lane * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(dst + byte), 16, 0, 0);
    }
    if constexpr (G::REM > 0) {                 // the remaining REM KiB: one more piece for the first REM waves (a wave-uniform branch)
      const int byte = (wave + WM * WN * G::N4) * 1024;
      if (__builtin_amdgcn_readfirstlane(wave) < G::REM) {
        const char* src = wb + (byte / G::GB1) * tile_bytes + (int64_t)gg * G::GB1 + byte % G::GB1 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(dst + byte), 16, 0, 0);
      }
    }
  };
  const int lane_b = r * 32 + ((hh ^ ((r >> 3) & 1)) << 4);   // this lane's 16 B inside a [32 co][32 B] fragment image
  // DUAL: the 1x1 weights of one chunk ([WN tiles][hi|lo][64 co][16 ci], 4 KiB per tile) -> their single LDS slot
  auto dma_b2 = [&](int chunk) {
    if (DUAL) {
      if (wave < 4 * WN && chunk < nchunks) {   // wave-uniform: one 1 KiB piece per wave
        const int t64 = wave >> 2, within = wave & 3;
        const char* src = (const char*)a.skip_w + ((int64_t)(nt * WN + t64) * k16s + chunk) * CM_BSTEP + within * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + G::B2_OFF + wave * 1024), 16, 0, 0);
      }
    }
  };

  const int lane_a = ((KS == 3) ? (wm * MT * G::PW + r) : (wm * MT * 32 + r)) * CM_PIX + hh * 16;
  // fp8 cross terms: K = 64 = (tap 0, tap 1 | tap 2, zero) x 16 channels of a 3-tap group; lane half hh owns K [32 hh, 32 hh + 32)
  // (operand map checked by scripts/mfma_f8_layout.hip).  Every address is a per-lane base with the lane-half dependence folded in
  // + a compile-time offset, like the rest of the loop's LDS reads.
  const int xa_lane0 = lane_a - hh * 16 + hh * (2 * CM_PIX);                        // e4m3 operand, first 16 bytes: the record of tap 0 (hh = 0) / tap 2 (hh = 1)
  const int xa_lane1_t0 = lane_a - hh * 16 + (hh ? 2 * CM_PIX + 64 : CM_PIX + 48);  // second 16 bytes, term 0 (a_lo image at +48; term 1's a image sits 16 bytes below): tap 1 / the zero padding of tap 2's record
  const int xb_lane = r * 16 + hh * (2 * CM_BSTEP);
  // E8M0 scale operands (2^(byte - 127), the same in every byte) take the constant pre-scales of the e4m3 images back out
  const int s_a = (127 - CM_X_SA) * 0x01010101, s_al = (127 - CM_X_SAL) * 0x01010101;
  const int s_w = (127 - a.w_exp) * 0x01010101, s_wl = (127 - a.w_exp - 11) * 0x01010101;

  f32x16 acc[MT][2];
  f32x16 acc2[DUAL ? MT : 1][2];                // DUAL: the 1x1 skip conv of the raw input
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[m][n][i] = 0.f;
        if (DUAL) acc2[m][n][i] = 0.f;
      }
  const int lane_a2 = ((wm * MT) * 32 + r) * 64 + ((hh ^ ((r >> 2) & 3)) << 4);   // DUAL: this lane's hi unit in the raw centre image (row m adds 32 * 64)

  // ---- prologue: chunk 0 into A buffer 0, B group 0 into ring slot 0
  if constexpr (X8) {
    if (q == 0) {      // bytes 64..79 of every pixel record in both A buffers: the zero tap of the cross-term MFMAs (never written again)
#pragma unroll
      for (int j = 0; j < G::NLOAD; ++j) {
        *(i32x4*)(smem + loff[j] + 64) = i32x4{0, 0, 0, 0};
        *(i32x4*)(smem + G::BUF + loff[j] + 64) = i32x4{0, 0, 0, 0};
      }
    }
  }
  dma_b(kc0 * G::NG);
  dma_b2(kc0);
  fetch_raw(kc0);

  const int emul_lim = a.emul_cout > 0 ? a.emul_cout : a.Cout;       // emul covers channels [0, emul_lim)
  const bool vec = ((a.Cout | a.ldo | emul_lim | (a.res ? a.ldr : 0) | (a.emul ? a.ld_emul : 0) | (a.egate ? (a.ld_egate | a.ld_eother) : 0)) & 3) == 0 &&
                   mud_dev_aligned16(a.out) && (!a.res || mud_dev_aligned16(a.res)) && (!a.emul || mud_dev_aligned16(a.emul)) &&
                   (!a.egate || (mud_dev_aligned16(a.egate) && mud_dev_aligned16(a.eother)));
  const bool slab_mode = KS == 3 && nsplit > 1 && fin.counters != nullptr;    // block-uniform
  // ---- the residual starts in the accumulators.  The epilogue adds a [pixels x channels] tile of `res` the size of the
  // accumulators; loaded there its HBM time is paid after the MFMAs (a 128->128 layer at 256x256 ran 12 % longer with a
  // residual), and a register prefetch a few chunks before the end costs 64 VGPRs.  Instead every accumulator register is
  // LOADED with its residual element here (accumulator layout: lane = channel, register = pixel -> 128-byte runs per half
  // wave), behind the first staging loads: the sums come out as res + sum(products), no registers are held and nothing is
  // left for the epilogue.  (Not with sub2 / split-K slabs / the fused skip conv, which has no residual.)
  const bool res_in_acc = !DUAL && KS == 3 && a.res != nullptr && !a.sub2 && nsplit == 1;
  if (res_in_acc) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int co = (nt * WN + wn) * CM_BN + n * 32 + r;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int gy = ty0 + wm * MT + m;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int gx = tx0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
          const bool ok = gy < a.H && gx < a.W && co < a.Cout;
          acc[m][n][reg] = ok ? a.res[(((int64_t)b * a.H + gy) * a.W + gx) * a.ldr + co] : 0.f;
        }
      }
    }
  }

  if (gn_fold) {                                // scale / shift of this sample -> LDS while the first loads are in flight
    cm_gn_to_lds(a, b, tid, G::NT, (float*)(smem + G::GN_OFF), gn_c);
    __syncthreads();
  } else if (PRO == MUD_PRO_AFFINE || PRO == MUD_PRO_AFFINE_SILU) {
    float* sc_lds = (float*)(smem + G::GN_OFF);
    for (int c = tid; c < a.Cin; c += G::NT) {
      sc_lds[c] = psc[c];
      sc_lds[gn_c + c] = psh[c];
    }
    __syncthreads();
  }
  fetch_ss(kc0);
  CM_RAW_WAIT(0);
  store_a(kc0, smem + (kc0 & 1) * G::BUF);
  __syncthreads();

  // One chunk of the reduction.  `more` (a further chunk follows: its tile is fetched and staged under this one's MFMAs) is a
  // COMPILE-TIME flag and the last chunk its own copy of the body: as a run-time condition it put every staging slice into a basic
  // block of its own - ~45 dependent vector instructions with no MFMA among them, during which the wave feeds the matrix pipe
  // nothing - and kept hipcc's scheduler from spreading them over the MFMA gaps of the step (each gap hides ~5 vector issues).
  auto chunk_body = [&](const int kc, auto more_t) {
    constexpr bool more = decltype(more_t)::value;
    char* cur = smem + (kc & 1) * G::BUF;
    char* nxt = smem + ((kc + 1) & 1) * G::BUF;
    cm_static_for<G::NG>([&](auto g_t) {
      constexpr int g = decltype(g_t)::value;
      const int gg = kc * G::NG + g;
      dma_b(gg + 1);                            // next group's weights stream in under this group's MFMAs
      if (g == 0 && more) fetch_a(kc + 1);
      const char* bcur = bring + (gg & 1) * G::GB + wn * G::GB1;
      if (DUAL && g == 0) {
        // skip conv: centre tap on the RAW tile.  Read here, at the head of the chunk: the single-buffered raw image and
        // weight slot are rewritten for chunk k+1 from group 1 on, i.e. behind this group's barrier.
        const char* b2 = smem + G::B2_OFF + wn * CM_BSTEP;
        h16x8 bh[2], bl[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          bh[n] = *(const h16x8*)(b2 + n * 1024 + lane_b);
          bl[n] = *(const h16x8*)(b2 + CM_BPLANE + n * 1024 + lane_b);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          // (the lo half sits two 16-B units away under the swizzle: bit 5 of the OFFSET - A2_OFF is a multiple of 64.  Flipping the bit on
          // the pointer through uintptr_t loses the LDS address space: a FLAT load, and behind it hipcc waits vmcnt(0) lgkmcnt(0), i.e. for
          // the weight DMA and the activation loads issued a moment ago, at the head of every chunk)
          const int o2 = lane_a2 + m * (32 * 64);
          const h16x8 ah = *(const h16x8*)(smem + G::A2_OFF + o2);
          const h16x8 al = *(const h16x8*)(smem + G::A2_OFF + (o2 ^ 32));
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            acc2[m][n] = mud_mfma16(al, bh[n], acc2[m][n]);
            acc2[m][n] = mud_mfma16(ah, bl[n], acc2[m][n]);
            acc2[m][n] = mud_mfma16(ah, bh[n], acc2[m][n]);
          }
        }
      }
      if (DUAL && g == 1 && more) dma_b2(kc + 1);
#pragma unroll
      for (int sg = 0; sg < G::GS; ++sg) {
        const int st = g * G::GS + sg;          // step inside the chunk: KS=3: tap; KS=1: k16 half
        const int s = (KS == 3) ? 0 : st;
        const int tap = (KS == 3) ? st : 0;
        const int dy = tap / KS, dx = tap % KS;
        if (KS == 3 || kc * G::CH + s < k16s) {      // (3x3: one k16 step per chunk, always inside the slice)
          HV8 bh[2], bl[2];
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            bh[n] = *(const HV8*)(bcur + sg * CM_BSTEP + n * 1024 + lane_b);
            if constexpr (!X8) bl[n] = *(const HV8*)(bcur + sg * CM_BSTEP + CM_BPLANE + n * 1024 + lane_b);
          }
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            // lane base (wave row, column r, k half hh) + compile-time (m, tap, s) offset
            const int off = s * G::PLANE + ((KS == 3) ? ((m + dy) * G::PW + dx) : (m * 32)) * CM_PIX;
            const HV8 ah = *(const HV8*)(cur + lane_a + off);
            if constexpr (X8) {
              // hi.hi in fp16; the cross terms follow the group's last tap
              acc[m][0] = mud_mfma16(ah, bh[0], acc[m][0]);
              acc[m][1] = mud_mfma16(ah, bh[1], acc[m][1]);
            } else {
              // term by term, the two accumulators alternate: no back-to-back MFMAs on one accumulator (64->64 layers -2 %, others
              // +-0: profiles/r02_l_ab_mfma_order.txt); each accumulator adds its terms in the order lo.hi, hi.lo, hi.hi
              const HV8 al = *(const HV8*)(cur + lane_a + off + 32);
              acc[m][0] = mud_mfma16(al, bh[0], acc[m][0]);
              acc[m][1] = mud_mfma16(al, bh[1], acc[m][1]);
              acc[m][0] = mud_mfma16(ah, bl[0], acc[m][0]);
              acc[m][1] = mud_mfma16(ah, bl[1], acc[m][1]);
              acc[m][0] = mud_mfma16(ah, bh[0], acc[m][0]);
              acc[m][1] = mud_mfma16(ah, bh[1], acc[m][1]);
            }
          }
          if constexpr (X8) {
            if (sg == G::GS - 1) {
              // cross terms of this 3-tap group.  The zero tap lives on the A side: bytes 64..79 of every pixel record (its padding) are
              // zeroed once per workgroup, and lane half 1 reads them as its second 16 bytes, so no operand needs a select; its B bytes
              // there are tap 1's (finite e4m3, times 0).
              const char* brow = bcur + xb_lane;                                // this lane's 16 input channels of output channel r: tap 0 (hh = 0) / tap 2 (hh = 1)
              const char* brow1 = bcur + CM_BSTEP + r * 16;                     // tap 1
              const int o0 = ((g * G::GS) / KS * G::PW + (g * G::GS) % KS) * CM_PIX;      // the group's first tap (taps 3g, 3g+1, 3g+2 are one row: +80, +160 bytes)
              // term 0: a_lo . w_hi (a_lo image at +48 of a record, scales 2^-SAL 2^-w_exp), term 1: a_hi . w_lo (a image at +32)
              const int d1 = hh ? 0 : 16;
              auto cross = [&](int term) {
                const int wplane = CM_BPLANE + term * 1024, aoff = 48 - 16 * term;
                const char* pa0 = cur + xa_lane0 + o0 + aoff;                           // first 16 bytes: tap 0 / tap 2
                const char* pa1 = cur + xa_lane1_t0 + o0 - d1 * term;                   // second: tap 1 / the record's zero padding
                const int sa = term ? s_a : s_al, sb = term ? s_wl : s_w;
                i32x8 wq[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                  const i32x4 q0 = *(const i32x4*)(brow + wplane + n * 512);
                  const i32x4 q1 = *(const i32x4*)(brow1 + wplane + n * 512);
                  wq[n] = i32x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                  const i32x4 p0 = *(const i32x4*)(pa0 + m * G::PW * CM_PIX);
                  const i32x4 p1 = *(const i32x4*)(pa1 + m * G::PW * CM_PIX);
                  const i32x8 aq = i32x8{p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
#pragma unroll
                  for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq, wq[n], acc[m][n], 0, 0, 0, sa, 0, sb);
                }
              };
              if constexpr (DUAL || MT == 1) {
                // with the second accumulator set (256 registers) and in the one-row tile (128 registers, two workgroups per CU) the two
                // terms run as a REAL loop: unrolled, hipcc hoists all sixteen operand reads of the group ahead of its MFMAs and spills
                // 34-70 registers (scripts/kernel_resources.py); rolled, the second term reuses the first one's.  The plain two-row kernels
                // have the registers and are 1-3 % faster unrolled.
#pragma clang loop unroll(disable)
                for (int term = 0; term < 2; ++term) cross(term);
              } else {
                cross(0);
                cross(1);
              }
            }
          }
        }
        // chunk k+1's LDS image is written slot by slot behind the MFMAs of steps GS.. (its raw loads had group 0 to land),
        // so the conversion VALU work interleaves with matrix work instead of forming one long MFMA-free stretch
        if (more && G::NG > 1 && st >= G::GS) {
          if (st == G::GS) {                    // (CNT) only this group's DMA is younger: N4 pieces, one more in the first REM waves
            if constexpr (G::REM > 0) {
              if (__builtin_amdgcn_readfirstlane(wave) < G::REM) CM_RAW_WAIT(G::N4 + 1);
              else CM_RAW_WAIT(G::N4);
            } else CM_RAW_WAIT(G::N4);
          }
          constexpr int SPAN = G::STEPS - G::GS;                   // steps available for staging
          // slot boundaries of the staging steps.  Without prologue arithmetic (PRO_NONE: G2's gate / fusion convolutions) a slice is short, and
          // placing the slices one step LATER (3 slots: steps 5, 7, 8 instead of 4, 6, 8) gives the loads more time: 2-5 % on those launches;
          // with the AdaGN + SiLU prologue the later placement crowds the end of the group (+0.5-1.5 %: profiles/r03_q_staging_one_step_later.txt)
          auto jb = [](int i) {
            if (PRO != MUD_PRO_NONE) return (i * G::NLOAD) / SPAN;
            return i >= SPAN ? G::NLOAD : ((i > 0 ? i - 1 : 0) * G::NLOAD + G::NLOAD - 1) / SPAN;
          };
          const int j0 = jb(st - G::GS), j1 = jb(st - G::GS + 1);
          store_a_slots(kc + 1, nxt, j0, j1);
        }
      }
      if (G::NG == 1 && more) store_a(kc + 1, nxt);
      if constexpr (g == 0 && more && CNT) {
        // group 0 issued, in this order, the DMA of group gg+1 and then the NLOAD activation loads of chunk kc+1 (not needed before
        // group 1).  Memory operations of a wave retire in order: waiting until only NLOAD are outstanding guarantees the DMA has
        // landed and leaves the activation loads IN FLIGHT across the barrier - __syncthreads() would drain them (HBM latency, ~1-2 us
        // under load, inside a ~1 us group: with the loads removed the bench line gains 7.6 %).
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G::NLOAD) : "memory");
      } else
        __syncthreads();                          // DMA of group gg+1 landed (vmcnt drained by the fence) and is visible to all waves
    });
  };
  for (int kc = kc0; kc + 1 < nchunks; ++kc) chunk_body(kc, std::true_type{});
  if (kc0 < nchunks) chunk_body(nchunks - 1, std::false_type{});

  if constexpr (KS == 3) {
    if (slab_mode) {
      // ---- split-K, reduced in this launch.  Every workgroup dumps its accumulators to its slab in REGISTER order (word j of thread
      // t at [j * threads + t]: full coalesced lines, no transposition); the LAST workgroup to arrive for an output tile (arrival
      // counter, which it leaves at zero for the next launch) adds the nsplit slabs in slab order - the result does not depend on who
      // is last - and alone runs the epilogue below.
      // Visibility across the XCDs' private L2s WITHOUT device-scope fences (a release fence is a whole-L2 write-back, an acquire a
      // whole-L2 invalidate: measured +80 us per launch): the slab words are agent-scope relaxed atomics - write-through stores,
      // loads that do not hit stale lines - each wave drains its stores before the barrier, and only then is the arrival counted.
      constexpr int NT = 64 * WM * WN;
      const unsigned nslots = nblocks / (unsigned)nsplit;
      const unsigned slot = ((unsigned)b * (unsigned)tiles_per_img + (unsigned)tile) * (unsigned)ntiles + (unsigned)nt;
      constexpr int ACC_WORDS = 32 * MT;          // accumulator words per thread and output (DUAL: the skip conv's set follows)
      const int64_t tile_words = (int64_t)NT * ACC_WORDS * (DUAL ? 2 : 1);
      float* mine = fin.ws + ((int64_t)ksi * nslots + slot) * tile_words + tid;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            __hip_atomic_store(mine + ((m * 2 + n) * 16 + reg) * NT, acc[m][n][reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (DUAL) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
              __hip_atomic_store(mine + (ACC_WORDS + (m * 2 + n) * 16 + reg) * NT, acc2[m][n][reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int* flag = (int*)smem;                     // the staging buffers are dead
      if (tid == 0) {
        unsigned* cnt = fin.counters + slot;
        const int last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nsplit - 1u;
        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
      }
      __syncthreads();
      if (!*flag) return;
      __syncthreads();                            // (the epilogue reuses the flag's LDS word)
      const float* slab0 = fin.ws + (int64_t)slot * tile_words + tid;
      for (int k = 0; k < nsplit; ++k) {
        const float* src = slab0 + (int64_t)k * nslots * tile_words;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const float pv = __hip_atomic_load(src + ((m * 2 + n) * 16 + reg) * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              acc[m][n][reg] = k == 0 ? pv : acc[m][n][reg] + pv;
            }
        if (DUAL) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
              for (int reg = 0; reg < 16; ++reg) {
                const float pv = __hip_atomic_load(src + (ACC_WORDS + (m * 2 + n) * 16 + reg) * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc2[m][n][reg] = k == 0 ? pv : acc2[m][n][reg] + pv;
              }
        }
      }
    }
  }

  // ---- epilogue: D[row = pixel (reg&3)+8*(reg>>2)+4*hh][col = channel r]
  // Fast path: every 32x32 accumulator tile is transposed through a wave-private LDS patch so that a lane owns
  // 4 consecutive channels of one pixel: residual loads and output stores are 16 B per lane (4 instructions per
  // tile instead of 16 scalar ones); the same pass folds the per-channel GroupNorm statistics.
  constexpr int EP_LD = 36;                                     // floats per staged pixel row (16-B aligned, bank-spread)
  float* ep = (float*)smem + wave * (32 * EP_LD);               // the A buffers are dead after the last barrier
  float* st_lds = (float*)smem + WM * WN * (32 * EP_LD);        // [wave][64 ch][2]
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int co = (nt * WN + wn) * CM_BN + n * 32 + r;
    const bool cok = co < a.Cout;
    float badd = (a.bias && cok) ? a.bias[co] : 0.f;
    if (a.bias2 && cok) badd += a.bias2[(int64_t)b * a.bias2_ld + co];
    if (vec) {
      const int col = (lane & 7) * 4, co4 = (nt * WN + wn) * CM_BN + n * 32 + col;   // this lane's 4 channels in the read-back phase
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) ep[((reg & 3) + 8 * (reg >> 2) + 4 * hh) * EP_LD + r] = acc[m][n][reg] + badd;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int prow = pass * 8 + (lane >> 3);
          int64_t opix;
          bool valid;
          if (KS == 3) {
            const int gy = ty0 + wm * MT + m, gx = tx0 + prow;
            valid = gy < a.H && gx < a.W;
            opix = ((int64_t)b * a.H + gy) * a.W + gx;
            if (a.sub2) {
              valid = valid && (gy & 1) && (gx & 1);
              opix = ((int64_t)b * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1);
            }
          } else {
            const int64_t fp = flat0 + (wm * MT + m) * 32 + prow;
            valid = fp < HW;
            opix = (int64_t)b * HW + fp;
          }
          f32x4 v = *(const f32x4*)(ep + prow * EP_LD + col);
          if (valid && co4 < a.Cout) {
            if (a.res && !res_in_acc) v += *(const f32x4*)(a.res + opix * a.ldr + co4);
            v *= a.out_scale;
            if (a.act != MUD_ACT_NONE) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = mud_act_fast(v[e], a.act);
            }
            if (a.emul && co4 < emul_lim) v *= *(const f32x4*)(a.emul + opix * a.ld_emul + co4);
            if (a.egate) {
              const f32x4 gt = *(const f32x4*)(a.egate + opix * a.ld_egate + co4);
              v = gt * v + (1.0f - gt) * *(const f32x4*)(a.eother + opix * a.ld_eother + co4);
            }
            *(f32x4*)(a.out + opix * a.ldo + co4) = v;
            s4 += v;
            q4 += v * v;
          }
        }
      }
      if (a.stats) {                            // lanes with equal (lane & 7) hold the same 4 channels
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float ss = s4[e], qq = q4[e];
#pragma unroll
          for (int o = 8; o < 64; o <<= 1) { ss += __shfl_xor(ss, o, 64); qq += __shfl_xor(qq, o, 64); }
          if (lane < 8) {
            st_lds[(wave * 64 + n * 32 + col + e) * 2] = ss;
            st_lds[(wave * 64 + n * 32 + col + e) * 2 + 1] = qq;
          }
        }
      }
    } else {
      float ssum = 0.f, ssq = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int prow = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
          int64_t opix;
          bool valid;
          if (KS == 3) {
            const int gy = ty0 + wm * MT + m, gx = tx0 + prow;
            valid = gy < a.H && gx < a.W;
            opix = ((int64_t)b * a.H + gy) * a.W + gx;
            if (a.sub2) {
              valid = valid && (gy & 1) && (gx & 1);
              opix = ((int64_t)b * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1);
            }
          } else {
            const int64_t fp = flat0 + (wm * MT + m) * 32 + prow;
            valid = fp < HW;
            opix = (int64_t)b * HW + fp;
          }
          if (valid && cok) {
            float v = acc[m][n][reg] + badd;
            if (a.res && !res_in_acc) v += a.res[opix * a.ldr + co];
            v = mud_act_fast(v * a.out_scale, a.act);
            if (a.emul && co < emul_lim) v *= a.emul[opix * a.ld_emul + co];
            if (a.egate) {
              const float gt = a.egate[opix * a.ld_egate + co];
              v = gt * v + (1.0f - gt) * a.eother[opix * a.ld_eother + co];
            }
            a.out[opix * a.ldo + co] = v;
            ssum += v;
            ssq += v * v;
          }
        }
      }
      if (a.stats) {                            // wave-uniform
        ssum += __shfl_xor(ssum, 32, 64);       // the two half-waves hold the same channel
        ssq += __shfl_xor(ssq, 32, 64);
        if (hh == 0) {
          st_lds[(wave * 64 + n * 32 + r) * 2] = ssum;
          st_lds[(wave * 64 + n * 32 + r) * 2 + 1] = ssq;
        }
      }
    }
  }
  if (DUAL) {
    // second output: skip_out = skip_w * x + skip_bias (no residual / activation / statistics), same LDS-transposed 16-B stores
    // (host: Cout % 4 == 0, skip_ldo % 4 == 0, 16-byte aligned skip_out)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int co = (nt * WN + wn) * CM_BN + n * 32 + r;
      const float badd = (a.skip_bias && co < a.Cout) ? a.skip_bias[co] : 0.f;
      const int col = (lane & 7) * 4, co4 = (nt * WN + wn) * CM_BN + n * 32 + col;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) ep[((reg & 3) + 8 * (reg >> 2) + 4 * hh) * EP_LD + r] = acc2[m][n][reg] + badd;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int prow = pass * 8 + (lane >> 3);
          const int gy = ty0 + wm * MT + m, gx = tx0 + prow;
          const f32x4 v = *(const f32x4*)(ep + prow * EP_LD + col);
          if (gy < a.H && gx < a.W && co4 < a.Cout) *(f32x4*)(a.skip_out + (((int64_t)b * a.H + gy) * a.W + gx) * a.skip_ldo + co4) = v;
        }
      }
    }
  }
  if (a.stats) {
    __syncthreads();
    if (tid < 128 * WN) {
      const int tn = tid >> 7, ch = (tid & 127) >> 1, k = tid & 1, co = (nt * WN + tn) * CM_BN + ch;
      if (co < a.Cout) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) t += st_lds[((tn * WM + w) * 64 + ch) * 2 + k];   // the WM waves that share this channel tile
        atomicAdd(a.stats + ((int64_t)b * a.stats_ld + co) * 2 + k, (double)t);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Variant for ks == 1 (plain GEMMs: 1x1 skip convs, NIN, attention contractions).  Only 24 MFMAs separate two
// barriers there, too little to hide a DMA group, so B fragments go straight from L2 to registers (one coalesced
// 1 KiB load per fragment, fetched a step ahead) and LDS holds only the double-buffered A tile.
// ------------------------------------------------------------------------------------------------
template <int KS, int MT>
struct CmGeoRegB {
  static constexpr int TAPS = KS * KS;
  static constexpr int CH = (KS == 3) ? 1 : 2;          // k16 steps per LDS chunk
  static constexpr int KCH = 16 * CH;                   // input channels per LDS chunk
  static constexpr int ROWS = 4 * MT;
  static constexpr int PW = 32 + KS - 1;
  static constexpr int P = (KS == 3) ? (ROWS + 2) * PW : 128 * MT;
  static constexpr int PLANE = P * CM_PIX;              // P pixel records [hi 32 B | lo 32 B | pad 16 B]
  static constexpr int BUF = CH * PLANE;                // [k16 s]
  static constexpr int GN_OFF = 2 * BUF;                // folded GroupNorm finalisation: scale | shift of the sample (sized per launch)
  static constexpr int GN_MAXC = CM_GN_MAXC;
  static constexpr int LDS_MAX = GN_OFF + CM_GN_BYTES;
  static constexpr int lds_bytes(int gn_channels) { return GN_OFF + 8 * ((gn_channels + 3) & ~3); }
  static constexpr int Q = 4 * CH;                      // float4 per pixel per chunk
  static constexpr int ITEMS = P * Q;
  static constexpr int NLOAD = (ITEMS + 255) / 256;
  static constexpr int STEPS = TAPS * CH;               // MFMA steps (tap, s) per chunk
  static constexpr int RING = (KS == 3) ? 3 : 2;        // B fragment register ring (prefetch distance RING-1)
};
template <int KS, int MT, int PRO>
__global__ __launch_bounds__(256, (MT == 4 ? 1 : 2)) void k_conv_mfma_regb(mud_conv_args a, int tiles_x, int tiles_per_img, int ntiles, int k16s,
                                                       unsigned nblocks) {
  using G = CmGeoRegB<KS, MT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cm_saturating_converters();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;

  // ---- XCD-aware bijective remap: consecutive logical ids share an XCD (blocks i, i+8 are co-resident on one XCD)
  unsigned lid;
  {
    const unsigned orig = blockIdx.x, xcd = orig & 7u, q = nblocks >> 3, rem = nblocks & 7u;
    lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
  }
  const int nt = lid % ntiles;                  // output-channel tile fastest: sharers of one A tile are neighbours
  const unsigned rest = lid / ntiles;
  const int tile = rest % tiles_per_img;
  const int b = rest / tiles_per_img;

  const int64_t HW = (int64_t)a.H * a.W;
  int ty0 = 0, tx0 = 0;
  int64_t flat0 = 0;
  if (KS == 3) {
    ty0 = (tile / tiles_x) * G::ROWS;
    tx0 = (tile % tiles_x) * 32;
  } else {
    flat0 = (int64_t)tile * G::P;
  }
  const float* xb = a.x + (int64_t)b * HW * a.ldx;
  const char* wb = (const char*)a.w + (int64_t)b * a.w_bstride + (int64_t)nt * k16s * (G::TAPS * CM_BSTEP) + r * 32 + ((hh ^ ((r >> 3) & 1)) << 4);   // packed halves are pre-swizzled (LDS image of the DMA path)
  const float* psc = a.pro_scale + (int64_t)b * a.pro_ld;
  const float* psh = a.pro_shift + (int64_t)b * a.pro_ld;
  const int nchunks = (k16s + G::CH - 1) / G::CH;

  // ---- per-thread staging slots: pixel -> global element offset and LDS byte offset, fixed for all chunks.
  // Everything below is branch-free: padding pixels load a clamped (valid) address and are zeroed by a 0/1
  // mask; slots past the tile wrap around and redo another thread's item (identical value, same address).
  const int q = tid % G::Q;                     // this thread's float4 (4 channels) inside a chunk
  int goff[G::NLOAD], loff[G::NLOAD];
  unsigned vmask = 0;                           // bit j: slot j is a real (non-padding) pixel
#pragma unroll
  for (int j = 0; j < G::NLOAD; ++j) {
    const int item = (tid + j * 256) % G::ITEMS;   // 256 % Q == 0 and ITEMS % Q == 0: q is preserved
    const int p = item / G::Q;
    bool valid;
    int64_t g;
    if (KS == 3) {
      const int py = p / G::PW, px = p - py * G::PW;
      const int gy = ty0 + py - 1, gx = tx0 + px - 1;
      valid = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      g = ((int64_t)gy * a.W + gx) * a.ldx;
    } else {
      const int64_t fp = flat0 + p;
      valid = fp < HW;
      g = fp * a.ldx;
    }
    goff[j] = valid ? (int)g : 0;
    vmask |= (valid ? 1u : 0u) << j;
    loff[j] = (q >> 2) * G::PLANE + p * CM_PIX + (q & 3) * 8;
  }

  f32x4 raw[G::NLOAD];
  f32x4 psc_r = {1.f, 1.f, 1.f, 1.f}, psh_r = {0.f, 0.f, 0.f, 0.f};   // prologue scale/shift of the chunk in `raw`
  const bool gn_fold = (PRO == MUD_PRO_AFFINE || PRO == MUD_PRO_AFFINE_SILU) && a.gn_sums != nullptr;   // workgroup-uniform
  const float* const gn_sc = (const float*)(smem + G::GN_OFF);
  const int gn_c = (a.Cin + 3) & ~3;            // shift array follows the scale array
  auto fetch_raw = [&](int chunk) {
    int c = chunk * G::KCH + q * 4;
    c = c < a.Cin ? c : 0;                      // clamped; zeroed in store_a
#pragma unroll
    for (int j = 0; j < G::NLOAD; ++j) raw[j] = *(const f32x4*)(xb + goff[j] + c);
  };
  auto fetch_ss = [&](int chunk) {
    if (PRO == MUD_PRO_AFFINE || PRO == MUD_PRO_AFFINE_SILU) {
      int c = chunk * G::KCH + q * 4;
      c = c < a.Cin ? c : 0;
      psc_r = *(const f32x4*)(gn_sc + c);         // always from LDS (see k_conv_mfma: a choice of pointers here is a FLAT load)
      psh_r = *(const f32x4*)(gn_sc + gn_c + c);
    }
  };
  auto fetch_a = [&](int chunk) {
    fetch_raw(chunk);
    fetch_ss(chunk);
  };
  auto store_a = [&](int chunk, char* buf) {
    const bool cvalid = chunk * G::KCH + q * 4 < a.Cin;
#pragma unroll
    for (int j = 0; j < G::NLOAD; ++j) {
      const float keep = (cvalid && ((vmask >> j) & 1u)) ? 1.0f : 0.0f;   // zero padding stays zero
      h16x4 hi, lo;
      int a8 = 0, al8 = 0;
      cm_stage4<PRO, false>(raw[j], psc_r, psh_r, keep, hi, lo, a8, al8);
      *(h16x4*)(buf + loff[j]) = hi;
      *(h16x4*)(buf + loff[j] + 32) = lo;
    }
  };

  // ---- B fragments: [step ring][n tile][hi|lo]
  h16x8 bfr[G::RING][2][2];
  const int total_steps = k16s * G::TAPS;       // global step index = k16 * TAPS + tap
  auto fetch_b = [&](int gstep, int slot) {
    if (gstep >= total_steps) gstep = total_steps - 1;   // harmless re-read past the end
    const char* p = wb + (int64_t)gstep * CM_BSTEP;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      bfr[slot][n][0] = *(const h16x8*)(p + n * 1024);
      bfr[slot][n][1] = *(const h16x8*)(p + CM_BPLANE + n * 1024);
    }
  };

  const int lane_a = ((KS == 3) ? (wave * MT * G::PW + r) : (wave * MT * 32 + r)) * CM_PIX + hh * 16;
  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  // ---- prologue: chunk 0 into buffer 0, first B fragments
  fetch_raw(0);
#pragma unroll
  for (int s = 0; s < G::RING - 1; ++s) fetch_b(s, s);
  if (gn_fold) {                                // scale / shift of this sample -> LDS while the first loads are in flight
    cm_gn_to_lds(a, b, tid, 256, (float*)(smem + G::GN_OFF), gn_c);
    __syncthreads();
  } else if (PRO == MUD_PRO_AFFINE || PRO == MUD_PRO_AFFINE_SILU) {
    float* sc_lds = (float*)(smem + G::GN_OFF);
    for (int c = tid; c < a.Cin; c += 256) {
      sc_lds[c] = psc[c];
      sc_lds[gn_c + c] = psh[c];
    }
    __syncthreads();
  }
  fetch_ss(0);
  store_a(0, smem);
  __syncthreads();

  for (int kc = 0; kc < nchunks; ++kc) {
    char* cur = smem + (kc & 1) * G::BUF;
    char* nxt = smem + ((kc + 1) & 1) * G::BUF;
    const bool more = kc + 1 < nchunks;
    if (more) fetch_a(kc + 1);
#pragma unroll
    for (int st = 0; st < G::STEPS; ++st) {
      // step st of this chunk: (s, tap).  KS=3: CH=1 -> st = tap.  KS=1: TAPS=1 -> st = s.
      const int s = (KS == 3) ? 0 : st;
      const int tap = (KS == 3) ? st : 0;
      const int dy = tap / KS, dx = tap % KS;
      const int gstep = (kc * G::CH + s) * G::TAPS + tap;
      fetch_b(gstep + G::RING - 1, (st + G::RING - 1) % G::RING);
      const int slot = st % G::RING;
      if (kc * G::CH + s < k16s) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          // lane base (wave row, column r, k half hh) + compile-time (m, tap, s) offset
          const int off = s * G::PLANE + ((KS == 3) ? ((m + dy) * G::PW + dx) : (m * 32)) * CM_PIX;
          const h16x8 ah = *(const h16x8*)(cur + lane_a + off);
          const h16x8 al = *(const h16x8*)(cur + lane_a + off + 32);
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            acc[m][n] = mud_mfma16(al, bfr[slot][n][0], acc[m][n]);
            acc[m][n] = mud_mfma16(ah, bfr[slot][n][1], acc[m][n]);
            acc[m][n] = mud_mfma16(ah, bfr[slot][n][0], acc[m][n]);
          }
        }
      }
      if (st == G::STEPS / 2 && more) store_a(kc + 1, nxt);   // chunk k+1's LDS image, written under the MFMA shadow
    }
    __syncthreads();
  }

  // ---- epilogue: D[row = pixel (reg&3)+8*(reg>>2)+4*hh][col = channel r]
  // Fast path: every 32x32 accumulator tile is transposed through a wave-private LDS patch so that a lane owns
  // 4 consecutive channels of one pixel: residual loads and output stores are 16 B per lane (4 instructions per
  // tile instead of 16 scalar ones); the same pass folds the per-channel GroupNorm statistics.
  constexpr int EP_LD = 36;                                     // floats per staged pixel row (16-B aligned, bank-spread)
  float* ep = (float*)smem + wave * (32 * EP_LD);               // the A buffers are dead after the last barrier
  float* st_lds = (float*)smem + 4 * (32 * EP_LD);              // [wave][64 ch][2]
  const int emul_lim = a.emul_cout > 0 ? a.emul_cout : a.Cout;       // emul covers channels [0, emul_lim)
  const bool vec = ((a.Cout | a.ldo | emul_lim | (a.res ? a.ldr : 0) | (a.emul ? a.ld_emul : 0) | (a.egate ? (a.ld_egate | a.ld_eother) : 0)) & 3) == 0 &&
                   mud_dev_aligned16(a.out) && (!a.res || mud_dev_aligned16(a.res)) && (!a.emul || mud_dev_aligned16(a.emul)) &&
                   (!a.egate || (mud_dev_aligned16(a.egate) && mud_dev_aligned16(a.eother)));
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int co = nt * CM_BN + n * 32 + r;
    const bool cok = co < a.Cout;
    float badd = (a.bias && cok) ? a.bias[co] : 0.f;
    if (a.bias2 && cok) badd += a.bias2[(int64_t)b * a.bias2_ld + co];
    if (vec) {
      const int col = (lane & 7) * 4, co4 = nt * CM_BN + n * 32 + col;   // this lane's 4 channels in the read-back phase
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) ep[((reg & 3) + 8 * (reg >> 2) + 4 * hh) * EP_LD + r] = acc[m][n][reg] + badd;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int prow = pass * 8 + (lane >> 3);
          int64_t opix;
          bool valid;
          if (KS == 3) {
            const int gy = ty0 + wave * MT + m, gx = tx0 + prow;
            valid = gy < a.H && gx < a.W;
            opix = ((int64_t)b * a.H + gy) * a.W + gx;
            if (a.sub2) {
              valid = valid && (gy & 1) && (gx & 1);
              opix = ((int64_t)b * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1);
            }
          } else {
            const int64_t fp = flat0 + (wave * MT + m) * 32 + prow;
            valid = fp < HW;
            opix = (int64_t)b * HW + fp;
          }
          f32x4 v = *(const f32x4*)(ep + prow * EP_LD + col);
          if (valid && co4 < a.Cout) {
            if (a.res) v += *(const f32x4*)(a.res + opix * a.ldr + co4);
            v *= a.out_scale;
            if (a.act != MUD_ACT_NONE) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = mud_act_fast(v[e], a.act);
            }
            if (a.emul && co4 < emul_lim) v *= *(const f32x4*)(a.emul + opix * a.ld_emul + co4);
            if (a.egate) {
              const f32x4 gt = *(const f32x4*)(a.egate + opix * a.ld_egate + co4);
              v = gt * v + (1.0f - gt) * *(const f32x4*)(a.eother + opix * a.ld_eother + co4);
            }
            *(f32x4*)(a.out + opix * a.ldo + co4) = v;
            s4 += v;
            q4 += v * v;
          }
        }
      }
      if (a.stats) {                            // lanes with equal (lane & 7) hold the same 4 channels
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float ss = s4[e], qq = q4[e];
#pragma unroll
          for (int o = 8; o < 64; o <<= 1) { ss += __shfl_xor(ss, o, 64); qq += __shfl_xor(qq, o, 64); }
          if (lane < 8) {
            st_lds[(wave * 64 + n * 32 + col + e) * 2] = ss;
            st_lds[(wave * 64 + n * 32 + col + e) * 2 + 1] = qq;
          }
        }
      }
    } else {
      float ssum = 0.f, ssq = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int prow = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
          int64_t opix;
          bool valid;
          if (KS == 3) {
            const int gy = ty0 + wave * MT + m, gx = tx0 + prow;
            valid = gy < a.H && gx < a.W;
            opix = ((int64_t)b * a.H + gy) * a.W + gx;
            if (a.sub2) {
              valid = valid && (gy & 1) && (gx & 1);
              opix = ((int64_t)b * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1);
            }
          } else {
            const int64_t fp = flat0 + (wave * MT + m) * 32 + prow;
            valid = fp < HW;
            opix = (int64_t)b * HW + fp;
          }
          if (valid && cok) {
            float v = acc[m][n][reg] + badd;
            if (a.res) v += a.res[opix * a.ldr + co];
            v = mud_act_fast(v * a.out_scale, a.act);
            if (a.emul && co < emul_lim) v *= a.emul[opix * a.ld_emul + co];
            if (a.egate) {
              const float gt = a.egate[opix * a.ld_egate + co];
              v = gt * v + (1.0f - gt) * a.eother[opix * a.ld_eother + co];
            }
            a.out[opix * a.ldo + co] = v;
            ssum += v;
            ssq += v * v;
          }
        }
      }
      if (a.stats) {                            // wave-uniform
        ssum += __shfl_xor(ssum, 32, 64);       // the two half-waves hold the same channel
        ssq += __shfl_xor(ssq, 32, 64);
        if (hh == 0) {
          st_lds[(wave * 64 + n * 32 + r) * 2] = ssum;
          st_lds[(wave * 64 + n * 32 + r) * 2 + 1] = ssq;
        }
      }
    }
  }
  if (a.stats) {
    __syncthreads();
    if (tid < 128) {
      const int ch = tid >> 1, k = tid & 1, co = nt * CM_BN + ch;
      if (co < a.Cout) {
        const float t = (st_lds[(0 * 64 + ch) * 2 + k] + st_lds[(1 * 64 + ch) * 2 + k]) +
                        (st_lds[(2 * 64 + ch) * 2 + k] + st_lds[(3 * 64 + ch) * 2 + k]);
        atomicAdd(a.stats + ((int64_t)b * a.stats_ld + co) * 2 + k, (double)t);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight / B-operand packing: fp32 (arbitrary strides) -> [n tile][k16 chunk][tap][hi|lo][64 co][16 ci] 16-bit pieces
// (exactly the order in which a wave's lanes consume MFMA B fragments: lane (r, h) of n-tile n reads the 16 B
//  at co = 32n + r, ci = 8h..8h+7).  prec = MUD_PREC_FP8X (3x3 only): the lo plane of a step holds the two e4m3 images
//  instead: [w * 2^w_exp: [2 n][32 co][16 ci] bytes][w_lo * 2^(w_exp + 11): the same], the hi plane fp16.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_weights(const float* __restrict__ src, int64_t s_tap, int64_t s_ci, int64_t s_co,
                                                      int64_t src_bstride, int taps, int Cin, int Cout, int k16s,
                                                      int64_t units, char* __restrict__ dst, int64_t dst_bstride, int prec, int w_exp) {
  const int b = blockIdx.y;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (int64_t)gridDim.x * blockDim.x) {
    const int h2 = (int)(u & 1), co_l = (int)((u >> 1) & 63);
    int64_t rest = u >> 7;
    const int tap = (int)(rest % taps);
    rest /= taps;
    const int kc = (int)(rest % k16s);
    const int nt = (int)(rest / k16s);
    const int co = nt * CM_BN + co_l, ci0 = kc * 16 + h2 * 8;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      if (co < Cout && ci0 + j < Cin) v = src[(int64_t)b * src_bstride + tap * s_tap + (int64_t)(ci0 + j) * s_ci + (int64_t)co * s_co];
      if (j < 4) v0[j] = v;
      else v1[j - 4] = v;
    }
    char* step = dst + (int64_t)b * dst_bstride + (((int64_t)nt * k16s + kc) * taps + tap) * CM_BSTEP;
    char* base = step + co_l * 32 + ((h2 ^ ((co_l >> 3) & 1)) << 4);   // 16-B halves swapped on rows with bit 3 set: the image is copied verbatim to LDS
    if (prec == MUD_PREC_FP8X) {
      h16x4 h0, h1, l0, l1;
      mud_split4(v0, h0, l0);
      mud_split4(v1, h1, l1);
      *(h16x4*)base = h0;
      *(h16x4*)(base + 8) = h1;
      const f32x4 lf0 = v0 - __builtin_convertvector(h0, f32x4), lf1 = v1 - __builtin_convertvector(h1, f32x4);
      const float sw = exp2f((float)w_exp), swl = exp2f((float)(w_exp + 11));
      char* row8 = step + CM_BPLANE + (co_l >> 5) * 512 + (co_l & 31) * 16 + h2 * 8;
      *(int*)(row8) = cm_e4m3x4(v0, sw);
      *(int*)(row8 + 4) = cm_e4m3x4(v1, sw);
      *(int*)(row8 + 1024) = cm_e4m3x4(lf0, swl);
      *(int*)(row8 + 1024 + 4) = cm_e4m3x4(lf1, swl);
    } else {
      h16x4 h0, h1, l0, l1;
      mud_split4(v0, h0, l0);
      mud_split4(v1, h1, l1);
      *(h16x4*)base = h0;
      *(h16x4*)(base + 8) = h1;
      *(h16x4*)(base + CM_BPLANE) = l0;
      *(h16x4*)(base + CM_BPLANE + 8) = l1;
    }
  }
}

extern "C" int64_t mud_packed_weight_bytes(int ks, int Cin, int Cout) {
  if ((ks != 1 && ks != 3) || Cin <= 0 || Cout <= 0) return -1;
  return mud_cdiv(Cout, CM_BN) * mud_cdiv(Cin, 16) * (int64_t)ks * ks * CM_BSTEP + 2 * CM_BSTEP;   // + slack: the last DMA group of a 1x1 operand with an odd number of 16-channel chunks reads one step past the end
}

extern "C" int mud_pack_weights_prec(const float* src, int64_t s_tap, int64_t s_ci, int64_t s_co, int64_t src_bstride, int ks,
                                     int Cin, int Cout, int nbatch, int prec, int w_exp, void* dst, void* stream) {
  MUD_REQUIRE(src && dst, "mud_pack_weights: null pointer");
  MUD_REQUIRE((ks == 1 || ks == 3) && Cin > 0 && Cout > 0 && nbatch >= 1 && nbatch <= 65535, "mud_pack_weights: bad sizes");
  MUD_REQUIRE(mud_aligned16(dst), "mud_pack_weights: dst must be 16-byte aligned");
  MUD_REQUIRE(prec == MUD_PREC_16X3 || (prec == MUD_PREC_FP8X && ks == 3 && w_exp >= -100 && w_exp <= 100),
              "mud_pack_weights: prec must be MUD_PREC_16X3, or MUD_PREC_FP8X with ks == 3 and |w_exp| <= 100 (got prec=%d ks=%d w_exp=%d)", prec, ks, w_exp);
  const int k16s = (int)mud_cdiv(Cin, 16), ntiles = (int)mud_cdiv(Cout, CM_BN), taps = ks * ks;
  const int64_t units = (int64_t)ntiles * k16s * taps * 128;
  int64_t blocks = mud_cdiv(units, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_pack_weights, dim3((int)blocks, nbatch), dim3(256), 0, (hipStream_t)stream, src, s_tap, s_ci, s_co,
                     src_bstride, taps, Cin, Cout, k16s, units, (char*)dst, mud_packed_weight_bytes(ks, Cin, Cout), prec, w_exp);
  MUD_CHECK_LAUNCH("mud_pack_weights");
  return MUD_OK;
}

extern "C" int mud_pack_weights(const float* src, int64_t s_tap, int64_t s_ci, int64_t s_co, int64_t src_bstride, int ks,
                                int Cin, int Cout, int nbatch, void* dst, void* stream) {
  return mud_pack_weights_prec(src, s_tap, s_ci, s_co, src_bstride, ks, Cin, Cout, nbatch, MUD_PREC_16X3, 0, dst, stream);
}

// ------------------------------------------------------------------------------------------------
// split-K for small grids (one slice at a time: 64x64 maps give 128 workgroups of 16-32 serial K chunks on 256 CUs):
// the K chunks are dealt to nsplit workgroups per output tile, each writes its raw fp32 partial tile to a slab, and this
// kernel adds the slabs in a FIXED order and applies the conv epilogue (bias, time-embedding bias, residual, scale,
// activation, GroupNorm statistics).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_splitk_epilogue(mud_conv_args a, const float* __restrict__ part, int nsplit, int64_t split_stride,
                                                         int ldp, int64_t npix, int pix_per_block) {
  __shared__ float red[256 * 8];
  const int quads = a.Cout >> 2;                       // float4 columns (host: Cout % 4 == 0, quads <= 256)
  const int rows = 256 / quads;                        // pixels per pass
  const int tq = threadIdx.x % quads, tr = threadIdx.x / quads;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
  const int64_t HW = (int64_t)a.H * a.W;
  const int b = (int)(p0 / HW);                        // host: pix_per_block divides H*W -> one sample per block
  const int co = tq * 4;
  f32x4 badd = {0.f, 0.f, 0.f, 0.f};
  if (tr < rows) {
    if (a.bias) badd = *(const f32x4*)(a.bias + co);
    if (a.bias2) badd += *(const f32x4*)(a.bias2 + (int64_t)b * a.bias2_ld + co);
  }
  f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
  if (tr < rows) {
    for (int64_t p = p0 + tr; p < p0 + pix_per_block && p < npix; p += rows) {
      f32x4 v = *(const f32x4*)(part + p * ldp + co);
      for (int k = 1; k < nsplit; ++k) v += *(const f32x4*)(part + k * split_stride + p * ldp + co);
      v += badd;
      if (a.res) v += *(const f32x4*)(a.res + p * a.ldr + co);
      v *= a.out_scale;
      if (a.act != MUD_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = mud_act_fast(v[e], a.act);
      }
      *(f32x4*)(a.out + p * a.ldo + co) = v;
      s4 += v;
      q4 += v * v;
    }
  }
  if (a.stats) {                                       // block-uniform
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[threadIdx.x * 8 + e] = s4[e];
      red[threadIdx.x * 8 + 4 + e] = q4[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < quads * 8; i += 256) {   // one (quad, sum|sumsq component) at a time per thread
      const int qd = i >> 3, e = i & 7;
      float t = 0.f;
      for (int r = 0; r < rows; ++r) t += red[(r * quads + qd) * 8 + e];
      atomicAdd(a.stats + ((int64_t)b * a.stats_ld + qd * 4 + (e & 3)) * 2 + (e >> 2), (double)t);
    }
  }
}

// split-K needs a plain epilogue and whole, aligned float4 channel columns (the reduce kernel's access pattern)
static bool cm_split_eligible(const mud_conv_args& a) {
  return a.ks == 3 && !a.sub2 && !a.emul && !a.egate && a.Cout % 4 == 0 && a.Cout <= 1024 && a.ldo % 4 == 0 && (!a.res || a.ldr % 4 == 0) &&
         mud_aligned16(a.out) && (!a.res || mud_aligned16(a.res)) && (!a.bias || mud_aligned16(a.bias)) &&
         (!a.bias2 || (mud_aligned16(a.bias2) && a.bias2_ld % 4 == 0));
}

// bytes of nsplit slabs of whole output tiles (>= the NHWC slabs of the two-launch path, which cover only real pixels / channels)
static int64_t cm_slab_bytes(int ns, int64_t blocks, int tile_words) { return (int64_t)ns * blocks * tile_words * 4; }

// how many K slices a 3x3 launch of `blocks` workgroups over `nchunks` 16-channel chunks is cut into (1 = no split)
static int cm_splits(int64_t blocks, int nchunks) {
  static const int force = getenv("MUD_CONV_SPLITK") ? atoi(getenv("MUD_CONV_SPLITK")) : -1;   // A/B knob: 0/1 = never, n = force n
  static const int kMaxBlocks = getenv("MUD_SPLITK_MAXBLOCKS") ? atoi(getenv("MUD_SPLITK_MAXBLOCKS")) : 192;   // tuning knobs
  static const int kTarget = getenv("MUD_SPLITK_TARGET") ? atoi(getenv("MUD_SPLITK_TARGET")) : 512;
  static const int kMinChunks = getenv("MUD_SPLITK_MINCHUNKS") ? atoi(getenv("MUD_SPLITK_MINCHUNKS")) : 16;
  static const int kMinPer = getenv("MUD_SPLITK_MINPER") ? atoi(getenv("MUD_SPLITK_MINPER")) : 4;
  int ns = 1;
  if (force >= 0) ns = force < 1 ? 1 : force;
  else if (blocks <= kMaxBlocks && nchunks >= kMinChunks) {
    // measured at one slice (profiles/r02_layer_times_b1_*.txt): 64x64 maps with >= 256 input channels gain (256->256: 50 -> 43 us,
    // 512->256: 89 -> 59 us, 384->256: 70 -> 51 us); 128x128 / 256x256 maps and shorter reductions lose (every workgroup pays
    // ~8 us of first-fetch + epilogue latency, and the second launch ~8 us), so they are left alone
    ns = (int)((kTarget + blocks - 1) / blocks);
    if (ns > 4) ns = 4;
  }
  if (ns > nchunks / kMinPer) ns = nchunks / kMinPer;             // at least 4 chunks per slice: the prologue / epilogue must stay amortised
  if (ns < 1) ns = 1;
  while (ns > 1 && (nchunks + ns - 1) / ns * (ns - 1) >= nchunks) --ns;   // every slice gets at least one chunk
  return ns;
}

template <int KS, int MT, int WM, int WN, int PRO, bool DUAL = false, int PREC = MUD_PREC_16X3>
static int cm_launch_pro(const mud_conv_args& a, hipStream_t s) {
  using G = typename std::conditional<KS == 3, CmGeo<KS, MT, WM, WN, DUAL>, CmGeoRegB<KS, MT>>::type;
  const void* kfn;
  if constexpr (KS == 3) kfn = (const void*)k_conv_mfma<KS, MT, WM, WN, PRO, DUAL, PREC>;      // only the variant that is launched is instantiated
  else kfn = (const void*)k_conv_mfma_regb<KS, MT, PRO>;
  static mud_attr_once attr_once;
  if (attr_once.need()) {
    hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_MAX);
    if (e != hipSuccess) {
      mud_set_error("mud_conv2d_mfma: cannot reserve %d B of LDS: %s", G::LDS_MAX, hipGetErrorString(e));
      return MUD_ERR_LAUNCH;
    }
    attr_once.ok();
  }
  const int k16s = (int)mud_cdiv(a.Cin, 16), ntiles = (int)mud_cdiv(a.Cout, CM_BN * WN);
  const int lds = G::lds_bytes((PRO == MUD_PRO_AFFINE || PRO == MUD_PRO_AFFINE_SILU) ? a.Cin : 0);   // scale | shift arrays of the sample: always in LDS
  int tiles_x = 1;
  int64_t tiles;
  if (KS == 3) {
    tiles_x = (int)mud_cdiv(a.W, 32);
    tiles = (int64_t)tiles_x * mud_cdiv(a.H, WM * MT);
  } else {
    tiles = mud_cdiv((int64_t)a.H * a.W, G::P);
  }
  int64_t nblocks = tiles * ntiles * a.B;
  MUD_REQUIRE(nblocks <= 0x7fffffff, "mud_conv2d_mfma: grid too large");
  if constexpr (KS == 3) {
    // split-K (small grids): needs the caller's slab workspace, a plain epilogue and whole float4 channel columns
    const int64_t npix = (int64_t)a.B * a.H * a.W;
    int ns = 1;
    static const bool two_launches = getenv("MUD_CONV_SPLITK_2LAUNCH") != nullptr;   // A/B knob
    const bool in_launch = !two_launches && a.splitk_counters && nblocks <= a.splitk_ncounters;
    // (the fused skip conv has two accumulator sets: only the in-launch reduction handles it)
    if ((!DUAL || in_launch) && a.splitk_ws && mud_aligned16(a.splitk_ws) && cm_split_eligible(a)) ns = cm_splits(nblocks, k16s);
    if (ns > 1 && a.splitk_ws_bytes < cm_slab_bytes(ns, nblocks, 64 * WM * WN * 32 * MT * (DUAL ? 2 : 1))) ns = 1;      // workspace too small: run unsplit
    if (ns > 1) {
      mud_conv_args p = a;                       // raw partial sums: no epilogue terms, output = slab ksi of the workspace
      p.out = (float*)a.splitk_ws;
      p.ldo = a.Cout;
      p.bias = p.bias2 = p.res = nullptr;
      p.out_scale = 1.0f;
      p.act = MUD_ACT_NONE;
      p.stats = nullptr;
      const int64_t stride = npix * a.Cout;
      // with arrival counters (one per output tile, zero between launches) the last workgroup of each tile reduces the slabs itself
      if (in_launch) {
        hipLaunchKernelGGL((k_conv_mfma<KS, MT, WM, WN, PRO, DUAL, PREC>), dim3((unsigned)(nblocks * ns)), dim3(64 * WM * WN), lds, s, a, tiles_x,
                           (int)tiles, ntiles, k16s, (unsigned)(nblocks * ns), ns, (int64_t)0, CmFin{(float*)a.splitk_ws, a.splitk_counters});
        MUD_CHECK_LAUNCH("mud_conv2d_mfma(split-K, reduced in the launch)");
        return MUD_OK;
      }
      hipLaunchKernelGGL((k_conv_mfma<KS, MT, WM, WN, PRO, DUAL, PREC>), dim3((unsigned)(nblocks * ns)), dim3(64 * WM * WN), lds, s, p, tiles_x,
                         (int)tiles, ntiles, k16s, (unsigned)(nblocks * ns), ns, stride, CmFin{nullptr, nullptr});
      MUD_CHECK_LAUNCH("mud_conv2d_mfma(split-K)");
      const int64_t HW = (int64_t)a.H * a.W;
      int ppb = 32;                              // pixels per block of the reduce: a divisor of H*W (one sample per block)
      while (ppb > 1 && HW % ppb) ppb >>= 1;
      hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)mud_cdiv(npix, ppb)), dim3(256), 0, s, a, (const float*)a.splitk_ws, ns, stride, a.Cout,
                         npix, ppb);
      MUD_CHECK_LAUNCH("mud_conv2d_mfma(split-K epilogue)");
      return MUD_OK;
    }
    hipLaunchKernelGGL((k_conv_mfma<KS, MT, WM, WN, PRO, DUAL, PREC>), dim3((unsigned)nblocks), dim3(64 * WM * WN), lds, s, a, tiles_x, (int)tiles,
                       ntiles, k16s, (unsigned)nblocks, 1, (int64_t)0, CmFin{nullptr, nullptr});
  } else
    hipLaunchKernelGGL((k_conv_mfma_regb<KS, MT, PRO>), dim3((unsigned)nblocks), dim3(256), lds, s, a, tiles_x, (int)tiles, ntiles, k16s,
                       (unsigned)nblocks);
  MUD_CHECK_LAUNCH("mud_conv2d_mfma");
  return MUD_OK;
}

template <int KS, int MT, int WM = 4, int WN = 1>
static int cm_launch(const mud_conv_args& a, hipStream_t s) {
  switch (a.pro_mode) {
    case MUD_PRO_NONE: return cm_launch_pro<KS, MT, WM, WN, MUD_PRO_NONE>(a, s);
    case MUD_PRO_AFFINE: return cm_launch_pro<KS, MT, WM, WN, MUD_PRO_AFFINE>(a, s);
    case MUD_PRO_LRELU: return cm_launch_pro<KS, MT, WM, WN, MUD_PRO_LRELU>(a, s);
    default: return cm_launch_pro<KS, MT, WM, WN, MUD_PRO_AFFINE_SILU>(a, s);
  }
}

#ifdef MUD_EXPERIMENT_WS
#include "experiments/conv_ws.inc"
#endif

// 3x3 tile variant by problem size: big tiles (more MFMA work per weight byte) once they still fill the 256 CUs.
// Measured on MI355X (scripts/bench_conv.py): MT=2 (two workgroups co-resident per CU, one wave of each per SIMD)
// beats MT=4 (one workgroup per CU) by 15-40 % on every layer shape; MT=1 only when MT=2 cannot give 2 blocks/CU.
enum { CMV_8X2, CMV_16X1, CMV_MT2, CMV_MT1, CMV_8X1R };
static int cm_variant3(const mud_conv_args& a, int64_t* blocks) {
  const int64_t ntiles = mud_cdiv(a.Cout, CM_BN);
  int64_t nb;
  int v;
  // 8-wave variant (8 rows x 32 px x 128 channels per workgroup, 4 x 2 waves): the staged input tile (GroupNorm +
  // SiLU + split, the main non-MFMA cost) is shared by twice as many MFMAs; used when the layer has an even number
  // of 64-channel tiles and still fills the chip
  const int64_t blocks8 = mud_cdiv(a.W, 32) * mud_cdiv(a.H, 8) * (ntiles / 2) * a.B;
  // other layers (64 or an odd number of 64-channel tiles): 8 waves stacked along the rows (16 x 32 px x 64 channels):
  // less halo per staged pixel and the weight ring is shared by twice as many waves (+4-7 % measured)
  const int64_t blocks16 = mud_cdiv(a.W, 32) * mud_cdiv(a.H, 16) * ntiles * a.B;
  const int64_t blocks2 = mud_cdiv(a.W, 32) * mud_cdiv(a.H, 8) * ntiles * a.B;
  static const bool no16 = getenv("MUD_CONV_NO16") != nullptr;   // A/B knob
  // 64 -> 64 layers (four K chunks: a quarter of a workgroup's life is first fetch + epilogue): 8 waves x ONE row x 64 channels,
  // 79 KiB of LDS and 110 VGPRs, so two workgroups share a CU (four waves per SIMD) and one's residual / store phase runs
  // under the other's K loop.  Measured against the 16-row tile (profiles/r02_j_ab_8x1row.txt): with a residual 330 -> 305 us,
  // without 307 -> 303 us; deeper reductions (128 / 192 / 256 -> 64) are equal, so they keep the tile with less halo.
  static const bool no8x1r = getenv("MUD_CONV_NO8X1R") != nullptr;   // A/B knob
  if (!no8x1r && !a.skip_w && ntiles == 1 && a.Cin <= 64 && a.H >= 8 && blocks2 >= 512) v = CMV_8X1R, nb = blocks2;
  else if (ntiles % 2 == 0 && a.H >= 8 && blocks8 >= 256) v = CMV_8X2, nb = blocks8;
  else if (!no16 && a.H >= 16 && blocks16 >= 256) v = CMV_16X1, nb = blocks16;
  else if (blocks2 >= 512 && a.H >= 8) v = CMV_MT2, nb = blocks2;
  else v = CMV_MT1, nb = mud_cdiv(a.W, 32) * mud_cdiv(a.H, 4) * ntiles * a.B;
  if (blocks) *blocks = nb;
  return v;
}

extern "C" int64_t mud_conv2d_mfma_splitk_bytes(const mud_conv_args* ap) {
  if (!ap || ap->ks != 3 || ap->B <= 0 || ap->H <= 0 || ap->W <= 0 || ap->Cin <= 0 || ap->Cout <= 0 || !cm_split_eligible(*ap)) return 0;
  int64_t blocks = 0;
  int v = cm_variant3(*ap, &blocks);
  if (ap->skip_w) {                              // the fused skip conv: two accumulator sets, the in-launch reduction only, and
    if (!ap->splitk_counters) return 0;          // mud_conv2d_mfma's tile choice (small grids take the 4-row tile)
    if (v != CMV_8X2 && v != CMV_16X1) {
      v = CMV_MT1;
      blocks = mud_cdiv(ap->W, 32) * mud_cdiv(ap->H, 4) * mud_cdiv(ap->Cout, CM_BN) * ap->B;
    }
  }
  const int ns = cm_splits(blocks, (int)mud_cdiv(ap->Cin, 16));
  const int tile_words = ((v == CMV_8X2 || v == CMV_16X1) ? 512 * 64 : v == CMV_MT2 ? 256 * 64 : v == CMV_8X1R ? 512 * 32 : 256 * 32) * (ap->skip_w ? 2 : 1);   // threads x accumulators
  return ns > 1 ? cm_slab_bytes(ns, blocks, tile_words) : 0;
}

// MUD_PREC_FP8X is built for the 8-wave tiles (8 x 32 px x 128 ch, 16 x 32 px x 64 ch, and the one-row 8 x 32 px x 64 ch tile of the
// 64 -> 64 layers), i.e. for launches that fill the chip, with the prologues the generators use there (none: G2's gate / fusion
// convolutions; AdaGN + SiLU: the residual blocks, with or without the fused skip conv - whose own centre-tap products stay 16-bit x 3).
static bool cm_fp8x_built(const mud_conv_args& a) {
  if (a.ks != 3 || a.sub2 || (a.pro_mode != MUD_PRO_NONE && a.pro_mode != MUD_PRO_AFFINE_SILU)) return false;
  if (a.skip_w && a.pro_mode != MUD_PRO_AFFINE_SILU) return false;
  const int v = cm_variant3(a, nullptr);
  return v == CMV_8X2 || v == CMV_16X1 || v == CMV_8X1R;
}
extern "C" int mud_conv2d_mfma_prec_supported(const mud_conv_args* ap, int prec) {
  if (!ap || ap->B <= 0 || ap->H <= 0 || ap->W <= 0 || ap->Cin <= 0 || ap->Cout <= 0) return 0;
  if (prec == MUD_PREC_16X3) return ap->ks == 1 || ap->ks == 3;
  mud_conv_args a = *ap;
  a.prec = prec;                                 // (the tile choice looks at it)
  return prec == MUD_PREC_FP8X && cm_fp8x_built(a);
}

extern "C" int mud_conv2d_mfma(const mud_conv_args* ap, void* stream) {
  MUD_REQUIRE(ap, "mud_conv2d_mfma: null args");
  mud_conv_args a = *ap;
  MUD_REQUIRE(a.x && a.w && a.out, "mud_conv2d_mfma: null pointer");
  MUD_REQUIRE((a.ks == 1 || a.ks == 3) && a.stride == 1 && a.pad == a.ks / 2, "mud_conv2d_mfma: only ks in {1,3}, stride 1, pad ks/2 (got ks=%d stride=%d pad=%d)", a.ks, a.stride, a.pad);
  MUD_REQUIRE(a.B >= 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.Cout > 0, "mud_conv2d_mfma: bad sizes");
  MUD_REQUIRE(a.Cin % 4 == 0 && a.ldx % 4 == 0 && a.ldx >= a.Cin && mud_aligned16(a.x), "mud_conv2d_mfma: needs Cin%%4==0 (Cin=%d), ldx%%4==0 (ldx=%d), 16-byte aligned x", a.Cin, a.ldx);
  MUD_REQUIRE((int64_t)a.H * a.W * a.ldx < 0x7fffffffLL, "mud_conv2d_mfma: one image must stay below 2^31 elements");
  MUD_REQUIRE(mud_aligned16(a.w) && a.w_bstride % 16 == 0, "mud_conv2d_mfma: packed weights must be 16-byte aligned");
  MUD_REQUIRE(a.ldo >= a.Cout && (!a.res || a.ldr >= a.Cout), "mud_conv2d_mfma: bad output/residual view");
  MUD_REQUIRE(!a.stats || a.stats_ld >= a.Cout, "mud_conv2d_mfma: bad stats view");
  MUD_REQUIRE(!a.sub2 || (a.ks == 3 && (a.H & 1) && (a.W & 1)), "mud_conv2d_mfma: sub2 needs ks == 3 and odd H, W");
  MUD_REQUIRE(a.emul_cout >= 0 && a.emul_cout <= a.Cout, "mud_conv2d_mfma: emul_cout out of range");
  MUD_REQUIRE((!a.emul || a.ld_emul >= (a.emul_cout > 0 ? a.emul_cout : a.Cout)) && (!a.egate || (a.eother && a.ld_egate >= a.Cout && a.ld_eother >= a.Cout)), "mud_conv2d_mfma: bad emul / egate / eother view");
  MUD_REQUIRE(a.pro_mode >= MUD_PRO_NONE && a.pro_mode <= MUD_PRO_LRELU, "mud_conv2d_mfma: unknown prologue mode %d", a.pro_mode);
  if ((a.pro_mode == MUD_PRO_AFFINE || a.pro_mode == MUD_PRO_AFFINE_SILU) && a.gn_sums) {
    MUD_REQUIRE(a.Cin <= CM_GN_MAXC && a.gn_G > 0 && a.Cin % a.gn_G == 0 && a.gn_sums_ld >= a.Cin && a.gn_count > 0,
                "mud_conv2d_mfma: folded GroupNorm needs Cin <= %d, Cin %% gn_G == 0, gn_sums_ld >= Cin, gn_count > 0 (Cin=%d gn_G=%d)", CM_GN_MAXC, a.Cin, a.gn_G);
    MUD_REQUIRE((a.gn_gamma == nullptr) == (a.gn_beta == nullptr) && a.gn_bstride >= 0, "mud_conv2d_mfma: folded GroupNorm: gamma and beta come together");
    a.pro_scale = a.pro_shift = a.x;   // never dereferenced
    a.pro_ld = 0;
  } else if (a.pro_mode == MUD_PRO_AFFINE || a.pro_mode == MUD_PRO_AFFINE_SILU) {
    a.gn_sums = nullptr;
    MUD_REQUIRE(a.pro_scale && a.pro_shift && a.pro_ld >= a.Cin && a.pro_ld % 4 == 0 && mud_aligned16(a.pro_scale) && mud_aligned16(a.pro_shift),
                "mud_conv2d_mfma: prologue arrays missing or misaligned");
    MUD_REQUIRE(a.Cin <= CM_GN_MAXC, "mud_conv2d_mfma: a launch with prologue arrays takes Cin <= %d (the kernels keep them in LDS; Cin=%d)", CM_GN_MAXC, a.Cin);
  } else {
    a.pro_scale = a.pro_shift = a.x;   // never dereferenced
    a.pro_ld = 0;
    a.gn_sums = nullptr;
  }
  if (a.skip_w) {
    MUD_REQUIRE(a.ks == 3 && a.pro_mode == MUD_PRO_AFFINE_SILU && !a.res && !a.sub2 && !a.emul && !a.egate,
                "mud_conv2d_mfma: the fused skip conv needs ks == 3, the AFFINE_SILU prologue and a plain epilogue");
    MUD_REQUIRE(a.skip_out && a.Cin <= 512 && a.Cout % 4 == 0 && a.skip_ldo >= a.Cout && a.skip_ldo % 4 == 0 && a.ldo % 4 == 0 &&
                mud_aligned16(a.skip_w) && mud_aligned16(a.skip_out) && mud_aligned16(a.out),
                "mud_conv2d_mfma: fused skip conv needs Cin <= 512, Cout %% 4 == 0 and aligned float4 output rows (Cin=%d Cout=%d)", a.Cin, a.Cout);
  }
  MUD_REQUIRE(a.prec == MUD_PREC_16X3 || a.prec == MUD_PREC_FP8X, "mud_conv2d_mfma: unknown arithmetic plan prec=%d", a.prec);
  MUD_REQUIRE(a.prec != MUD_PREC_FP8X || (cm_fp8x_built(a) && a.w_exp >= -100 && a.w_exp <= 100),
              "mud_conv2d_mfma: MUD_PREC_FP8X is not built for this launch (ask mud_conv2d_mfma_prec_supported first; the weights were packed for it and cannot be read by another plan)");
  if (a.B == 0) return MUD_OK;
  hipStream_t s = (hipStream_t)stream;
#ifdef MUD_EXPERIMENT_WS      // scripts/build_variants.py ws:-DMUD_EXPERIMENT_WS (csrc/experiments/conv_ws.inc); never in the shipped build
  if (cm_ws_wanted(a)) return cm_ws_dispatch(a, s);
#endif
  if (a.prec == MUD_PREC_FP8X) {
    const int v8 = cm_variant3(a, nullptr);
    if (v8 == CMV_8X1R) return a.pro_mode == MUD_PRO_NONE ? cm_launch_pro<3, 1, 8, 1, MUD_PRO_NONE, false, MUD_PREC_FP8X>(a, s) : cm_launch_pro<3, 1, 8, 1, MUD_PRO_AFFINE_SILU, false, MUD_PREC_FP8X>(a, s);
    const bool x2 = v8 == CMV_8X2;
    if (a.skip_w) return x2 ? cm_launch_pro<3, 2, 4, 2, MUD_PRO_AFFINE_SILU, true, MUD_PREC_FP8X>(a, s) : cm_launch_pro<3, 2, 8, 1, MUD_PRO_AFFINE_SILU, true, MUD_PREC_FP8X>(a, s);
    if (a.pro_mode == MUD_PRO_NONE) return x2 ? cm_launch_pro<3, 2, 4, 2, MUD_PRO_NONE, false, MUD_PREC_FP8X>(a, s) : cm_launch_pro<3, 2, 8, 1, MUD_PRO_NONE, false, MUD_PREC_FP8X>(a, s);
    return x2 ? cm_launch_pro<3, 2, 4, 2, MUD_PRO_AFFINE_SILU, false, MUD_PREC_FP8X>(a, s) : cm_launch_pro<3, 2, 8, 1, MUD_PRO_AFFINE_SILU, false, MUD_PREC_FP8X>(a, s);
  }
  // tile height by problem size: big tiles (more MFMA work per weight byte) once they still fill the 256 CUs
  const int64_t ntiles = mud_cdiv(a.Cout, CM_BN);
  static const int force_mt = getenv("MUD_CONV_MT") ? atoi(getenv("MUD_CONV_MT")) : 0;   // tuning knob (plain launches only: the fused skip conv keeps its own tiles)
  if (force_mt && !a.skip_w) {
    if (a.ks == 3) return (force_mt == 8 && ntiles % 2 == 0) ? cm_launch<3, 2, 4, 2>(a, s) : force_mt == 16 ? cm_launch<3, 2, 8, 1>(a, s) : force_mt == 1 ? cm_launch<3, 1>(a, s) : cm_launch<3, 2>(a, s);
    return cm_launch<1, 1>(a, s);      // (256- and 512-pixel 1x1 tiles were measured 5-8 % slower and are no longer built)
  }
  if (a.ks == 3 && a.skip_w) {
    // fused 1x1 skip conv (DUAL): built for the AdaGN + SiLU prologue of the residual blocks only; the 8-row 4-wave tile would
    // drop to one workgroup per CU with the second image, so small grids take the 4-row tile
    switch (cm_variant3(a, nullptr)) {
      case CMV_8X2: return cm_launch_pro<3, 2, 4, 2, MUD_PRO_AFFINE_SILU, true>(a, s);
      case CMV_16X1: return cm_launch_pro<3, 2, 8, 1, MUD_PRO_AFFINE_SILU, true>(a, s);
      default: return cm_launch_pro<3, 1, 4, 1, MUD_PRO_AFFINE_SILU, true>(a, s);
    }
  }
  if (a.ks == 3) {
    switch (cm_variant3(a, nullptr)) {
      case CMV_8X2: return cm_launch<3, 2, 4, 2>(a, s);
      case CMV_16X1: return cm_launch<3, 2, 8, 1>(a, s);
      case CMV_MT2: return cm_launch<3, 2>(a, s);
      case CMV_8X1R: return cm_launch<3, 1, 8, 1>(a, s);
      default: return cm_launch<3, 1>(a, s);
    }
  }
  // 1x1 GEMMs: 128-pixel tiles (MT = 1) beat 256-pixel tiles (MT = 2) by 5-8 % on every skip-conv / NIN shape at batch 16
  // (scripts/bench_conv.py with MUD_CONV_MT=1/2) - these launches are HBM-bound and the smaller tile keeps more of them in flight
  return cm_launch<1, 1>(a, s);
}
