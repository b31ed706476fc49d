// Exact-fp32 direct convolution on NHWC views (one fmaf chain per output, like ATen's reference
// arithmetic up to summation order).  It serves the shapes that are bandwidth- rather than
// FLOP-bound and that the MFMA implicit GEMM does not cover:
//   * Cin == 1 head convolutions (1 -> nf at full resolution: a pure 256 B/pixel store stream),
//   * the Cout == 1 output convolution (GroupNorm+SiLU prologue, tanh epilogue fused),
//   * the stride-2 convolution of the input-pyramid Downsample (after the FIR),
// and acts as the exact fallback for any other shape.  One thread owns one output pixel and VO
// consecutive output channels; the VO-wide weight rows are wave-uniform loads served by L1/L2.
#include "mud_common.h"
#include <stdlib.h>

template <int VO, int VI>
__global__ __launch_bounds__(256) void k_conv_direct(mud_conv_args a, int Ho, int Wo, int co_groups) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* st_lds = (float*)smem_raw;             // [Cout][2] per-channel (sum, sumsq) of this block's outputs
  const int b = blockIdx.y;
  if (a.stats) {
    for (int i = threadIdx.x; i < 2 * a.Cout; i += 256) st_lds[i] = 0.f;
    __syncthreads();
  }
  // 32-bit index arithmetic (the launcher checks Ho*Wo*co_groups < 2^31): 64-bit div/mod are ~100-instruction routines
  const unsigned total = (unsigned)Ho * (unsigned)Wo * (unsigned)co_groups;
  // loop stride = a multiple of co_groups, so a thread keeps its output-channel group and can fold the
  // statistics of all its outputs in registers (threads past the last whole multiple sit out)
  const unsigned nthreads = gridDim.x * blockDim.x;
  const unsigned stride = (nthreads / (unsigned)co_groups) * (unsigned)co_groups;
  const unsigned first = blockIdx.x * blockDim.x + threadIdx.x;
  const int cg = (int)(first % (unsigned)co_groups);            // constant along the loop
  float ssum[VO], ssq[VO];
#pragma unroll
  for (int j = 0; j < VO; ++j) ssum[j] = ssq[j] = 0.f;
  for (uint64_t idx64 = first; first < stride && idx64 < total; idx64 += stride) {
    const unsigned p = (unsigned)idx64 / (unsigned)co_groups;
    const int oy = (int)(p / (unsigned)Wo);
    const int ox = (int)(p - (unsigned)oy * (unsigned)Wo);
    const int co = cg * VO;
    float acc[VO];
#pragma unroll
    for (int j = 0; j < VO; ++j) acc[j] = 0.f;
    const float* wb = (const float*)((const char*)a.w + (int64_t)b * a.w_bstride);
    for (int ky = 0; ky < a.ks; ++ky) {
      const int iy = oy * a.stride - a.pad + ky;
      if (iy < 0 || iy >= a.H) continue;
      for (int kx = 0; kx < a.ks; ++kx) {
        const int ix = ox * a.stride - a.pad + kx;
        if (ix < 0 || ix >= a.W) continue;
        const float* xp = a.x + (((int64_t)b * a.H + iy) * a.W + ix) * a.ldx;
        const float* wp = wb + ((int64_t)(ky * a.ks + kx) * a.Cin) * a.Cout + co;
        for (int ci = 0; ci < a.Cin; ci += VI) {
          float xv[VI];
          if (VI == 4) *(f32x4*)xv = *(const f32x4*)(xp + ci);
          else xv[0] = xp[ci];
#pragma unroll
          for (int u = 0; u < VI; ++u) {
            float v = xv[u];
            if (a.pro_mode == MUD_PRO_LRELU)
              v = v > 0.f ? v : 0.2f * v;
            else if (a.pro_mode != MUD_PRO_NONE)
              v = mud_prologue(v, a.pro_scale[(int64_t)b * a.pro_ld + ci + u], a.pro_shift[(int64_t)b * a.pro_ld + ci + u], a.pro_mode);
            const float* w = wp + (int64_t)(ci + u) * a.Cout;
            if (VO == 4) {
              const f32x4 wv = *(const f32x4*)w;
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[j] = fmaf(v, wv[j], acc[j]);
            } else {
              acc[0] = fmaf(v, w[0], acc[0]);
            }
          }
        }
      }
    }
    const int64_t opix = ((int64_t)b * Ho + oy) * Wo + ox;
#pragma unroll
    for (int j = 0; j < VO; ++j) {
      float v = acc[j];
      if (a.bias) v += a.bias[co + j];
      if (a.bias2) v += a.bias2[(int64_t)b * a.bias2_ld + co + j];
      if (a.res) v += a.res[opix * a.ldr + co + j];
      v = mud_act(v * a.out_scale, a.act);
      if (a.emul && (a.emul_cout <= 0 || co + j < a.emul_cout)) v *= a.emul[opix * a.ld_emul + co + j];
      if (a.egate) {
        const float gt = a.egate[opix * a.ld_egate + co + j];
        v = gt * v + (1.0f - gt) * a.eother[opix * a.ld_eother + co + j];
      }
      acc[j] = v;
    }
    float* op = a.out + opix * a.ldo + co;
    if (VO == 4) *(f32x4*)op = *(f32x4*)acc;
    else op[0] = acc[0];
#pragma unroll
    for (int j = 0; j < VO; ++j) {
      ssum[j] += acc[j];
      ssq[j] += acc[j] * acc[j];
    }
  }
  if (a.stats) {
    if (first < stride && first < total) {
      const int co = (int)(first % co_groups) * VO;
#pragma unroll
      for (int j = 0; j < VO; ++j) {
        atomicAdd(&st_lds[(co + j) * 2], ssum[j]);
        atomicAdd(&st_lds[(co + j) * 2 + 1], ssq[j]);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.Cout; i += 256)
      atomicAdd(a.stats + ((int64_t)b * a.stats_ld + (i >> 1)) * 2 + (i & 1), (double)st_lds[i]);
  }
}

// Cin == 1, 3x3, stride 1, pad 1 (the ConvFeatBlock / ConvBlock head convolutions): a 256 B/pixel store
// stream.  A thread owns 4 output channels (their 9x4 weights + bias live in registers) and a strip of
// HEAD_PIX consecutive pixels of one row: the 3 x (HEAD_PIX+2) input window is loaded once, branch-free
// (clamped address x 0/1 mask), and reused by the 9 taps.  co_groups lanes cover one pixel's channels, so
// stores are 16 B per lane in runs of 4*co_groups*4 B (256 B for 64 channels).
#define HEAD_PIX 8
template <int S>                               // stride: 1 (pad 1: head convs) or 2 (pad 0: the input-pyramid conv after its FIR)
__global__ __launch_bounds__(256) void k_conv_head3x3(mud_conv_args a, int co_groups, int strips_per_row, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* st_lds = (float*)smem_raw;
  const int b = blockIdx.y;
  if (a.stats) {
    for (int i = threadIdx.x; i < 2 * a.Cout; i += 256) st_lds[i] = 0.f;
    __syncthreads();
  }
  const int cg = threadIdx.x % co_groups, co = cg * 4;
  const int spb = 256 / co_groups;                      // strips handled concurrently by the block
  const int ls = threadIdx.x / co_groups;
  f32x4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = *(const f32x4*)((const float*)a.w + (int64_t)t * a.Cout + co);
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias = *(const f32x4*)(a.bias + co);
  if (a.bias2) bias += *(const f32x4*)(a.bias2 + (int64_t)b * a.bias2_ld + co);
  const int64_t HW = (int64_t)a.H * a.W;
  const float* xb = a.x + (int64_t)b * HW * a.ldx;
  f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t nstrips = (int64_t)Ho * strips_per_row;
  constexpr int WIN = S * (HEAD_PIX - 1) + 3;       // input columns under a strip
  const bool active = ls < spb;
  for (int64_t st = (int64_t)blockIdx.x * spb + ls; active && st < nstrips; st += (int64_t)gridDim.x * spb) {
    const int y = (int)(st / strips_per_row), x0 = (int)(st - (int64_t)y * strips_per_row) * HEAD_PIX;
    float win[3][WIN];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = y * S + dy - a.pad;
      const bool yok = iy >= 0 && iy < a.H;
      const int64_t rowoff = (int64_t)(yok ? iy : 0) * a.W;
#pragma unroll
      for (int j = 0; j < WIN; ++j) {
        const int ix = x0 * S + j - a.pad;
        const bool ok = yok && ix >= 0 && ix < a.W;
        const float v = xb[(rowoff + (ok ? ix : 0)) * a.ldx];
        win[dy][j] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < HEAD_PIX; ++k) {
      if (x0 + k >= Wo) break;
      f32x4 acc = bias;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) acc += win[dy][k * S + dx] * w[dy * 3 + dx];
      const int64_t opix = ((int64_t)b * Ho + y) * Wo + x0 + k;
      if (a.res) acc += *(const f32x4*)(a.res + opix * a.ldr + co);
      acc *= a.out_scale;
      if (a.act != MUD_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = mud_act(acc[e], a.act);
      }
      *(f32x4*)(a.out + opix * a.ldo + co) = acc;
      s4 += acc;
      q4 += acc * acc;
    }
  }
  if (a.stats) {
    if (active) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(&st_lds[(co + e) * 2], s4[e]);
        atomicAdd(&st_lds[(co + e) * 2 + 1], q4[e]);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.Cout; i += 256)
      atomicAdd(a.stats + ((int64_t)b * a.stats_ld + (i >> 1)) * 2 + (i & 1), (double)st_lds[i]);
  }
}

// Cout <= 4, 3x3, stride 1, pad 1, Cin % 16 == 0 (the generators' output convolution 64 -> 1 with its GroupNorm+SiLU
// prologue and tanh epilogue, and the progressive='output_skip' pyramid convolutions): HBM-bound (reads Cin*4 B per
// pixel, writes 4 B).  A 256-thread block owns an 8x32 output tile; the (8+2)x(32+2) halo tile is staged 16 input
// channels at a time through LDS - loaded once as 16-B vectors, activated ONCE per element (the generic kernel applies
// the prologue 9 times per element, the matrix-core kernel wastes 63 of its 64 output columns) - as 80-byte pixel
// records (conflict-free ds_read_b128 at one pixel per lane); each thread then accumulates its pixel's 9 x 16 taps per
// chunk from LDS with wave-uniform (scalar) weight loads.  Exact fp32 FMA chain.
#define TAIL_ROWS 8
#define TAIL_REC 20                               // floats per pixel record: 16 channels + 4 padding
template <int COUT>
__global__ __launch_bounds__(256) void k_conv_tail3x3(mud_conv_args a, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float tile[(TAIL_ROWS + 2) * 34 * TAIL_REC];
  const int b = blockIdx.y;
  const int ty0 = (blockIdx.x / tiles_x) * TAIL_ROWS, tx0 = (blockIdx.x % tiles_x) * 32;
  const int tid = threadIdx.x, py = tid >> 5, px = tid & 31;      // this thread's output pixel inside the tile
  const float* xb = a.x + (int64_t)b * a.H * a.W * a.ldx;
  const float* wb = (const float*)((const char*)a.w + (int64_t)b * a.w_bstride);
  float acc[COUT];
#pragma unroll
  for (int j = 0; j < COUT; ++j) acc[j] = 0.f;
  constexpr int NPIX = (TAIL_ROWS + 2) * 34, ITEMS = NPIX * 4;
  for (int c0 = 0; c0 < a.Cin; c0 += 16) {
    // ---- stage 16 channels of the halo tile (prologue applied once per element; padding pixels are zeros)
    for (int it = tid; it < ITEMS; it += 256) {
      const int p = it >> 2, q = it & 3;
      const int gy = ty0 + p / 34 - 1, gx = tx0 + p % 34 - 1;
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        v = *(const f32x4*)(xb + ((int64_t)gy * a.W + gx) * a.ldx + c0 + q * 4);
        if (a.pro_mode == MUD_PRO_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
        } else if (a.pro_mode != MUD_PRO_NONE) {
          const f32x4 sc = *(const f32x4*)(a.pro_scale + (int64_t)b * a.pro_ld + c0 + q * 4);
          const f32x4 sh = *(const f32x4*)(a.pro_shift + (int64_t)b * a.pro_ld + c0 + q * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = mud_prologue(v[e], sc[e], sh[e], a.pro_mode);
        }
      }
      *(f32x4*)(tile + p * TAIL_REC + q * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float* rec = tile + ((py + t / 3) * 34 + px + t % 3) * TAIL_REC;
      const float* wt = wb + ((int64_t)t * a.Cin + c0) * a.Cout;      // wave-uniform: [tap][ci][co]
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *(const f32x4*)(rec + q * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < COUT; ++j) acc[j] = fmaf(v[e], wt[(q * 4 + e) * COUT + j], acc[j]);
      }
    }
    __syncthreads();
  }
  const int gy = ty0 + py, gx = tx0 + px;
  if (gy < a.H && gx < a.W) {
    const int64_t opix = ((int64_t)b * a.H + gy) * a.W + gx;
#pragma unroll
    for (int j = 0; j < COUT; ++j) {
      float v = acc[j];
      if (a.bias) v += a.bias[j];
      if (a.res) v += a.res[opix * a.ldr + j];
      a.out[opix * a.ldo + j] = mud_act(v * a.out_scale, a.act);
    }
  }
}

extern "C" int mud_conv2d_direct(const mud_conv_args* ap, void* stream) {
  MUD_REQUIRE(ap, "mud_conv2d_direct: null args");
  const mud_conv_args a = *ap;
  MUD_REQUIRE(a.x && a.w && a.out, "mud_conv2d_direct: null pointer");
  MUD_REQUIRE(a.B >= 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.Cout > 0 && a.ks >= 1 && a.stride >= 1 && a.pad >= 0,
              "mud_conv2d_direct: bad sizes");
  MUD_REQUIRE(a.ldx >= a.Cin && a.ldo >= a.Cout, "mud_conv2d_direct: ld smaller than C");
  MUD_REQUIRE(a.pro_mode == MUD_PRO_NONE || a.pro_mode == MUD_PRO_LRELU || (a.pro_scale && a.pro_shift && a.pro_ld >= a.Cin), "mud_conv2d_direct: prologue arrays missing");
  MUD_REQUIRE(!a.res || a.ldr >= a.Cout, "mud_conv2d_direct: bad residual view");
  const int Ho = (a.H + 2 * a.pad - a.ks) / a.stride + 1, Wo = (a.W + 2 * a.pad - a.ks) / a.stride + 1;
  MUD_REQUIRE(Ho > 0 && Wo > 0, "mud_conv2d_direct: empty output");
  if (a.B == 0) return MUD_OK;
  const bool vo4 = a.Cout % 4 == 0 && a.ldo % 4 == 0 && mud_aligned16(a.out) && mud_aligned16(a.w) && (a.w_bstride % 16 == 0);
  const bool vi4 = a.Cin % 4 == 0 && a.ldx % 4 == 0 && mud_aligned16(a.x);
  const int co_groups = vo4 ? a.Cout / 4 : a.Cout;
  if (a.Cin == 1 && a.ks == 3 && ((a.stride == 1 && a.pad == 1) || (a.stride == 2 && a.pad == 0)) && vo4 && !a.emul && !a.egate && co_groups <= 256 && a.pro_mode == MUD_PRO_NONE && a.w_bstride == 0 &&
      (!a.res || (a.ldr % 4 == 0 && mud_aligned16(a.res))) && (!a.bias || mud_aligned16(a.bias)) &&
      (!a.bias2 || (a.bias2_ld % 4 == 0 && mud_aligned16(a.bias2)))) {
    MUD_REQUIRE(a.B <= 65535 && (!a.stats || a.stats_ld >= a.Cout), "mud_conv2d_direct: bad batch / stats view");
    const int spb = 256 / co_groups, strips_per_row = (int)mud_cdiv(Wo, HEAD_PIX);
    int64_t blocks = mud_cdiv((int64_t)Ho * strips_per_row, spb);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (a.stats) {
      // every block ends with one fp64 atomic per channel on the SAME [b, channel] cells, and same-address atomics retire
      // one at a time (~150 ns each in L2): 512 blocks per image cost 79 us of a 133 us launch.  Keep about 2048 blocks
      // in total (8 per CU) and let each block walk more strips instead.
      // At small batches the same holds per image: measured at 256x256 -> 64 channels (profiles/r02_e_head_blocks.txt), B = 1 / 2 / 4:
      // 512 blocks per image 22.7 / 28.3 / 45.3 us, 256: 17.4 / 20.6 / 26.7 us, 128: 18.8 / 19.3 / 23.4 us, 64: 25.6 / 26.0 / 27.0 us.
      static const int cap = getenv("MUD_HEAD_CAP") ? atoi(getenv("MUD_HEAD_CAP")) : 0;     // tuning knob
      int64_t per_image = 2048 / a.B > 16 ? 2048 / a.B : 16;
      const int64_t lim = cap > 0 ? cap : (a.B == 1 ? 256 : 128);
      if (per_image > lim) per_image = lim;
      if (blocks > per_image) blocks = per_image;
    }
    if (a.stride == 1)
      hipLaunchKernelGGL(k_conv_head3x3<1>, dim3((int)blocks, a.B), dim3(256), a.stats ? 2 * a.Cout * sizeof(float) : 0, (hipStream_t)stream, a, co_groups,
                         strips_per_row, Ho, Wo);
    else
      hipLaunchKernelGGL(k_conv_head3x3<2>, dim3((int)blocks, a.B), dim3(256), a.stats ? 2 * a.Cout * sizeof(float) : 0, (hipStream_t)stream, a, co_groups,
                         strips_per_row, Ho, Wo);
    MUD_CHECK_LAUNCH("mud_conv2d_direct(head)");
    return MUD_OK;
  }
  if (a.ks == 3 && a.stride == 1 && a.pad == 1 && a.Cout <= 4 && a.Cin % 16 == 0 && vi4 && !a.emul && !a.egate && !a.stats && !a.bias2 && !a.sub2 &&
      (a.pro_mode == MUD_PRO_NONE || a.pro_mode == MUD_PRO_LRELU || (a.pro_ld % 4 == 0 && mud_aligned16(a.pro_scale) && mud_aligned16(a.pro_shift)))) {
    MUD_REQUIRE(a.B <= 65535, "mud_conv2d_direct: B too large");
    const int tiles_x = (int)mud_cdiv(a.W, 32), tiles_y = (int)mud_cdiv(a.H, TAIL_ROWS);
    dim3 grid(tiles_x * tiles_y, a.B), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (a.Cout) {
      case 1: hipLaunchKernelGGL((k_conv_tail3x3<1>), grid, block, 0, s, a, tiles_x, tiles_y); break;
      case 2: hipLaunchKernelGGL((k_conv_tail3x3<2>), grid, block, 0, s, a, tiles_x, tiles_y); break;
      case 3: hipLaunchKernelGGL((k_conv_tail3x3<3>), grid, block, 0, s, a, tiles_x, tiles_y); break;
      default: hipLaunchKernelGGL((k_conv_tail3x3<4>), grid, block, 0, s, a, tiles_x, tiles_y); break;
    }
    MUD_CHECK_LAUNCH("mud_conv2d_direct(tail)");
    return MUD_OK;
  }
  const int64_t total = (int64_t)Ho * Wo * co_groups;
  int64_t blocks = mud_cdiv(total, 256 * (a.stats ? 8 : 1));   // several outputs per thread when the block folds statistics
  if (blocks > 256 * 32) blocks = 256 * 32;
  while (blocks * 256 < co_groups) ++blocks;                   // the loop stride (a multiple of co_groups) must be positive
  MUD_REQUIRE(a.B <= 65535 && total < (1ll << 31) - (1ll << 24), "mud_conv2d_direct: B or image too large");
  MUD_REQUIRE(!a.stats || (a.stats_ld >= a.Cout && a.Cout <= 8192), "mud_conv2d_direct: bad stats view");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((int)blocks, a.B), block(256);
  const size_t lds = a.stats ? 2 * a.Cout * sizeof(float) : 0;
  if (vo4 && vi4) hipLaunchKernelGGL((k_conv_direct<4, 4>), grid, block, lds, s, a, Ho, Wo, co_groups);
  else if (vo4) hipLaunchKernelGGL((k_conv_direct<4, 1>), grid, block, lds, s, a, Ho, Wo, co_groups);
  else if (vi4) hipLaunchKernelGGL((k_conv_direct<1, 4>), grid, block, lds, s, a, Ho, Wo, co_groups);
  else hipLaunchKernelGGL((k_conv_direct<1, 1>), grid, block, lds, s, a, Ho, Wo, co_groups);
  MUD_CHECK_LAUNCH("mud_conv2d_direct");
  return MUD_OK;
}
