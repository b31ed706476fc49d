// Implicit-GEMM convolution / GEMM on the CDNA4 matrix cores with fp32-class accuracy.
//
//   out[p, co] = sum_{tap, ci} f(x[p + off(tap), ci]) * w[tap, ci, co]         (ks = 3, pad 1 | ks = 1)
//
// gfx950 has no TF32; exact-fp32 MFMA runs at the VALU rate (157 TF), 1/16 of the bf16 rate.  The
// path needs fp32-class results (<= 1e-3 after ~50 layers), so each fp32 operand is split into two
// bf16 terms (hi = rne(v), lo = rne(v - hi)) and every product is issued as
//      lo*hi + hi*lo + hi*hi          (3 x v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// i.e. ~2^-17 relative error per product at 16/3 = 5.3x the fp32-MFMA rate.
//
// Tiling (one 256-thread workgroup = 4 waves, one wave per SIMD):
//   * output tile: 8 rows x 32 columns of pixels (ks=3) or 256 flat positions (ks=1) x 64 channels;
//     wave w owns rows 2w, 2w+1 -> 2 (pixel) x 2 (channel) MFMA tiles of 32x32, 64 accumulator VGPRs.
//   * K loop over chunks of 32 input channels.  Per chunk the block stages in LDS
//       A: the (8+2)x(32+2) input halo tile, transformed ONCE on the way in (GroupNorm/AdaGN affine,
//          SiLU, bf16 hi/lo split) and then reused by all 9 taps and all 64 output channels;
//       B: the pre-packed, pre-split weight tile [tap][hi|lo][k16][64 co][16 ci] (linear copy).
//     Both images keep 16 contiguous K values (32 B) per pixel/channel so that every MFMA operand is
//     one ds_read_b128; the two 16-B halves of a row are swapped on rows with bit 3 set, which makes
//     the 4x16-lane ds_read_b128 groups hit 16 distinct 16-B bank slots (conflict-free).
//   * epilogue from the accumulators: + bias[co] + bias2[b,co] (time embedding) + residual, * scale,
//     activation; each half-wave stores 32 consecutive channels (128 B) of one pixel.
#include "mud_common.h"

#define CM_TH 8
#define CM_TW 32
#define CM_BN 64
#define CM_KC 32
#define CM_BPLANE (CM_BN * 32)   // bytes of one [64 co][16 ci] bf16 plane

template <int KS>
struct CmGeo {
  static constexpr int HALO = KS - 1;
  static constexpr int PW = CM_TW + HALO;
  static constexpr int P = (KS == 1) ? 256 : (CM_TH + HALO) * PW;
  static constexpr int A_PLANE = P * 32;
  static constexpr int A_BYTES = 4 * A_PLANE;               // [k16 half s][hi|lo]
  static constexpr int B_BYTES = KS * KS * 4 * CM_BPLANE;   // [tap][hi|lo][s]
  static constexpr int LDS_BYTES = A_BYTES + B_BYTES;
};

__device__ __forceinline__ int cm_row_off(int row, int half) { return row * 32 + ((half ^ ((row >> 3) & 1)) << 4); }

template <int KS>
__global__ __launch_bounds__(256) void k_conv_mfma(mud_conv_args a, int tiles_x, int kchunks) {
  using G = CmGeo<KS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + G::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, nt = blockIdx.y;
  const int64_t HW = (int64_t)a.H * a.W;
  int ty0 = 0, tx0 = 0;
  int64_t flat0 = 0;
  if (KS == 3) {
    ty0 = (blockIdx.x / tiles_x) * CM_TH;
    tx0 = (blockIdx.x % tiles_x) * CM_TW;
  } else {
    flat0 = (int64_t)blockIdx.x * 256;
  }
  const float* xb = a.x + (int64_t)b * HW * a.ldx;
  const char* wb = (const char*)a.w + (int64_t)b * a.w_bstride + (int64_t)nt * kchunks * G::B_BYTES;
  const float* psc = a.pro_scale + (int64_t)b * a.pro_ld;
  const float* psh = a.pro_shift + (int64_t)b * a.pro_ld;

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  for (int kc = 0; kc < kchunks; ++kc) {
    __syncthreads();
    // ---- stage A: global fp32 -> prologue -> bf16 hi/lo -> LDS
    for (int i = tid; i < G::P * 8; i += 256) {
      const int p = i >> 3, q = i & 7;
      const int c = kc * CM_KC + q * 4;
      bool valid = c < a.Cin;
      const float* src;
      if (KS == 3) {
        const int py = p / G::PW, px = p - py * G::PW;
        const int gy = ty0 + py - 1, gx = tx0 + px - 1;
        valid = valid && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        src = xb + ((int64_t)gy * a.W + gx) * a.ldx + c;
      } else {
        const int64_t fp = flat0 + p;
        valid = valid && fp < HW;
        src = xb + fp * a.ldx + c;
      }
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        v = *(const f32x4*)src;
        if (a.pro_mode != MUD_PRO_NONE) {
          const f32x4 sc = *(const f32x4*)(psc + c), sh = *(const f32x4*)(psh + c);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = mud_prologue(v[j], sc[j], sh[j], a.pro_mode);
        }
      }
      const bf16x4 hi = __builtin_convertvector(v, bf16x4);
      const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
      const int off = (q >> 2) * 2 * G::A_PLANE + cm_row_off(p, (q & 3) >> 1) + (q & 1) * 8;
      *(bf16x4*)(sA + off) = hi;
      *(bf16x4*)(sA + off + G::A_PLANE) = lo;
    }
    // ---- stage B: the packed weight tile is already in LDS order
    {
      const uint4* src = (const uint4*)(wb + (int64_t)kc * G::B_BYTES);
      for (int i = tid; i < G::B_BYTES / 16; i += 256) ((uint4*)sB)[i] = src[i];
    }
    __syncthreads();
    // ---- MFMA: 9 taps x 2 k16 halves x (2x2 tiles) x 3 split products
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
      const int dy = tap / KS, dx = tap % KS;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int p = (KS == 3) ? ((wave * 2 + m + dy) * G::PW + r + dx) : (wave * 64 + m * 32 + r);
          const int off = s * 2 * G::A_PLANE + cm_row_off(p, hh);
          ah[m] = *(const bf16x8*)(sA + off);
          al[m] = *(const bf16x8*)(sA + off + G::A_PLANE);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int off = (tap * 4 + s) * CM_BPLANE + cm_row_off(n * 32 + r, hh);
          bh[n] = *(const bf16x8*)(sB + off);
          bl[n] = *(const bf16x8*)(sB + off + 2 * CM_BPLANE);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
          }
      }
    }
  }

  // ---- epilogue: D[row = pixel (reg&3)+8*(reg>>2)+4*hh][col = channel r]
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int co = nt * CM_BN + n * 32 + r;
      if (co >= a.Cout) continue;
      float badd = a.bias ? a.bias[co] : 0.f;
      if (a.bias2) badd += a.bias2[(int64_t)b * a.bias2_ld + co];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int prow = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        int64_t opix;
        bool valid;
        if (KS == 3) {
          const int gy = ty0 + wave * 2 + m, gx = tx0 + prow;
          valid = gy < a.H && gx < a.W;
          opix = ((int64_t)b * a.H + gy) * a.W + gx;
        } else {
          const int64_t fp = flat0 + wave * 64 + m * 32 + prow;
          valid = fp < HW;
          opix = (int64_t)b * HW + fp;
        }
        if (!valid) continue;
        float v = acc[m][n][reg] + badd;
        if (a.res) v += a.res[opix * a.ldr + co];
        a.out[opix * a.ldo + co] = mud_act(v * a.out_scale, a.act);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight / B-operand packing: fp32 (arbitrary strides) -> [n tile][k chunk][tap][hi|lo][k16][64][16] bf16
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_weights(const float* __restrict__ src, int64_t s_tap, int64_t s_ci, int64_t s_co,
                                                      int64_t src_bstride, int taps, int Cin, int Cout, int kchunks,
                                                      int64_t units, char* __restrict__ dst, int64_t dst_bstride) {
  const int b = blockIdx.y;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (int64_t)gridDim.x * blockDim.x) {
    const int h2 = (int)(u & 1), co_l = (int)((u >> 1) & 63), s = (int)((u >> 7) & 1);
    int64_t rest = u >> 8;
    const int tap = (int)(rest % taps);
    rest /= taps;
    const int kc = (int)(rest % kchunks);
    const int nt = (int)(rest / kchunks);
    const int co = nt * CM_BN + co_l, ci0 = kc * CM_KC + s * 16 + h2 * 8;
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      if (co < Cout && ci0 + j < Cin) v = src[(int64_t)b * src_bstride + tap * s_tap + (int64_t)(ci0 + j) * s_ci + (int64_t)co * s_co];
      const __bf16 h = (__bf16)v;
      hi[j] = h;
      lo[j] = (__bf16)(v - (float)h);
    }
    char* base = dst + (int64_t)b * dst_bstride + (((int64_t)nt * kchunks + kc) * taps + tap) * (4 * CM_BPLANE);
    const int off = s * CM_BPLANE + cm_row_off(co_l, h2);
    *(bf16x8*)(base + off) = hi;
    *(bf16x8*)(base + off + 2 * CM_BPLANE) = lo;
  }
}

extern "C" int64_t mud_packed_weight_bytes(int ks, int Cin, int Cout) {
  if ((ks != 1 && ks != 3) || Cin <= 0 || Cout <= 0) return -1;
  return mud_cdiv(Cout, CM_BN) * mud_cdiv(Cin, CM_KC) * (int64_t)ks * ks * 4 * CM_BPLANE;
}

extern "C" int mud_pack_weights(const float* src, int64_t s_tap, int64_t s_ci, int64_t s_co, int64_t src_bstride, int ks,
                                int Cin, int Cout, int nbatch, void* dst, void* stream) {
  MUD_REQUIRE(src && dst, "mud_pack_weights: null pointer");
  MUD_REQUIRE((ks == 1 || ks == 3) && Cin > 0 && Cout > 0 && nbatch >= 1 && nbatch <= 65535, "mud_pack_weights: bad sizes");
  MUD_REQUIRE(mud_aligned16(dst), "mud_pack_weights: dst must be 16-byte aligned");
  const int kchunks = (int)mud_cdiv(Cin, CM_KC), ntiles = (int)mud_cdiv(Cout, CM_BN), taps = ks * ks;
  const int64_t units = (int64_t)ntiles * kchunks * taps * 256;
  int64_t blocks = mud_cdiv(units, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_pack_weights, dim3((int)blocks, nbatch), dim3(256), 0, (hipStream_t)stream, src, s_tap, s_ci, s_co,
                     src_bstride, taps, Cin, Cout, kchunks, units, (char*)dst, mud_packed_weight_bytes(ks, Cin, Cout));
  MUD_CHECK_LAUNCH("mud_pack_weights");
  return MUD_OK;
}

template <int KS>
static int cm_launch(const mud_conv_args& a, hipStream_t s) {
  using G = CmGeo<KS>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_conv_mfma<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      mud_set_error("mud_conv2d_mfma: cannot reserve %d B of LDS: %s", G::LDS_BYTES, hipGetErrorString(e));
      return MUD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int kchunks = (int)mud_cdiv(a.Cin, CM_KC), ntiles = (int)mud_cdiv(a.Cout, CM_BN);
  int tiles_x = 1;
  int64_t tiles;
  if (KS == 3) {
    tiles_x = (int)mud_cdiv(a.W, CM_TW);
    tiles = (int64_t)tiles_x * mud_cdiv(a.H, CM_TH);
  } else {
    tiles = mud_cdiv((int64_t)a.H * a.W, 256);
  }
  MUD_REQUIRE(tiles <= 0x7fffffff && ntiles <= 65535 && a.B <= 65535, "mud_conv2d_mfma: grid too large");
  hipLaunchKernelGGL((k_conv_mfma<KS>), dim3((unsigned)tiles, ntiles, a.B), dim3(256), G::LDS_BYTES, s, a, tiles_x, kchunks);
  MUD_CHECK_LAUNCH("mud_conv2d_mfma");
  return MUD_OK;
}

extern "C" int mud_conv2d_mfma(const mud_conv_args* ap, void* stream) {
  MUD_REQUIRE(ap, "mud_conv2d_mfma: null args");
  mud_conv_args a = *ap;
  MUD_REQUIRE(a.x && a.w && a.out, "mud_conv2d_mfma: null pointer");
  MUD_REQUIRE((a.ks == 1 || a.ks == 3) && a.stride == 1 && a.pad == a.ks / 2, "mud_conv2d_mfma: only ks in {1,3}, stride 1, pad ks/2 (got ks=%d stride=%d pad=%d)", a.ks, a.stride, a.pad);
  MUD_REQUIRE(a.B >= 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.Cout > 0, "mud_conv2d_mfma: bad sizes");
  MUD_REQUIRE(a.Cin % 4 == 0 && a.ldx % 4 == 0 && a.ldx >= a.Cin && mud_aligned16(a.x), "mud_conv2d_mfma: needs Cin%%4==0 (Cin=%d), ldx%%4==0 (ldx=%d), 16-byte aligned x", a.Cin, a.ldx);
  MUD_REQUIRE(mud_aligned16(a.w) && a.w_bstride % 16 == 0, "mud_conv2d_mfma: packed weights must be 16-byte aligned");
  MUD_REQUIRE(a.ldo >= a.Cout && (!a.res || a.ldr >= a.Cout), "mud_conv2d_mfma: bad output/residual view");
  if (a.pro_mode != MUD_PRO_NONE) {
    MUD_REQUIRE(a.pro_scale && a.pro_shift && a.pro_ld >= a.Cin && a.pro_ld % 4 == 0 && mud_aligned16(a.pro_scale) && mud_aligned16(a.pro_shift),
                "mud_conv2d_mfma: prologue arrays missing or misaligned");
  } else {
    a.pro_scale = a.pro_shift = a.x;   // never dereferenced
    a.pro_ld = 0;
  }
  if (a.B == 0) return MUD_OK;
  return a.ks == 3 ? cm_launch<3>(a, (hipStream_t)stream) : cm_launch<1>(a, (hipStream_t)stream);
}
