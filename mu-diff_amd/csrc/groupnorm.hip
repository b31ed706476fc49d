// GroupNorm statistics for NHWC views -> per-(sample, channel) scale/shift that the consuming
// convolution applies while it loads its A operand (so the normalised tensor is never written).
//
// Pass 1 (HBM-bound, one coalesced read of the activation): grid (nsplit, B); a block sweeps a
//   contiguous pixel range with 16 B per lane, every lane keeps fp32 sum / sum-of-squares of ITS 4
//   channels over <= ~64 pixels, the rows of the block are folded in fp64 through LDS and each
//   group's (sum, sumsq) partial is written to the workspace (no global atomics; the LDS fold is fp64, so order effects vanish in fp32).
// Pass 2 (tiny): one block per sample folds the partials in fp64 (wave shuffles), forms mean / rstd
//   (biased variance, like torch native_group_norm) and writes scale = gamma*rstd,
//   shift = beta - mean*scale.
#include "mud_common.h"

#define GN_THREADS 256
#define GN_MAX_SPLIT 1024
#define GN_PIX_PER_LANE 64

struct GnGeom { int cols, rows, nsplit; int64_t chunk; };

static GnGeom gn_geom(int64_t HW, int C) {
  GnGeom g;
  g.cols = C / 4;
  g.rows = GN_THREADS / g.cols;
  if (g.rows < 1) g.rows = 1;
  int64_t ns = mud_cdiv(HW, (int64_t)g.rows * GN_PIX_PER_LANE);
  if (ns > GN_MAX_SPLIT) ns = GN_MAX_SPLIT;
  if (ns < 1) ns = 1;
  g.chunk = mud_cdiv(HW, ns);
  g.nsplit = (int)mud_cdiv(HW, g.chunk);
  return g;
}

extern "C" int64_t mud_gn_ws_bytes(int B, int64_t HW, int C, int G) {
  if (C <= 0 || C % 4) return -1;
  const GnGeom g = gn_geom(HW, C);
  return (int64_t)B * g.nsplit * (int64_t)(G > C ? G : C) * 2 * (int64_t)sizeof(double);
}

// PER_CHANNEL: partials per channel (channel mean) instead of per group.
template <bool PER_CHANNEL>
__global__ __launch_bounds__(GN_THREADS) void k_gn_partial(const float* __restrict__ x, int64_t HW, int C, int ld, int G,
                                                           int cols, int rows, int64_t chunk, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_sum = (double*)smem_raw;       // [C]
  double* s_sq = s_sum + C;                // [C]
  const int b = blockIdx.y, split = blockIdx.x, nsplit = gridDim.x;
  const int tid = threadIdx.x;
  for (int c = tid; c < 2 * C; c += GN_THREADS) s_sum[c] = 0.0;
  __syncthreads();
  const int64_t p0 = (int64_t)split * chunk;
  int64_t p1 = p0 + chunk;
  if (p1 > HW) p1 = HW;
  if (tid < cols * rows) {
    const int col = tid % cols, row = tid / cols;
    const float* base = x + (int64_t)b * HW * ld + col * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    for (int64_t p = p0 + row; p < p1; p += rows) {
      const f32x4 v = *(const f32x4*)(base + p * ld);
      s += v;
      q += v * v;
    }
    // fold the rows of this block in fp64 (LDS atomics on doubles are native on gfx950)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      atomicAdd(&s_sum[col * 4 + j], (double)s[j]);
      atomicAdd(&s_sq[col * 4 + j], (double)q[j]);
    }
  }
  __syncthreads();
  if (PER_CHANNEL) {
    for (int c = tid; c < C; c += GN_THREADS) {
      double* o = part + (((int64_t)b * nsplit + split) * C + c) * 2;
      o[0] = s_sum[c];
      o[1] = s_sq[c];
    }
  } else {
    const int cpg = C / G;
    for (int g = tid; g < G; g += GN_THREADS) {
      double a = 0.0, bq = 0.0;
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += s_sum[c]; bq += s_sq[c]; }
      double* o = part + (((int64_t)b * nsplit + split) * G + g) * 2;
      o[0] = a;
      o[1] = bq;
    }
  }
}

__global__ __launch_bounds__(256) void k_gn_finalize(const double* __restrict__ part, int nsplit, int C, int G, double count,
                                                     float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int64_t g_bstride, float* __restrict__ scale, float* __restrict__ shift,
                                                     int ld_ss, float* __restrict__ mean_rstd) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cpg = C / G;
  for (int g = wave; g < G; g += 4) {
    double s = 0.0, q = 0.0;
    for (int i = lane; i < nsplit; i += 64) {
      const double* o = part + (((int64_t)b * nsplit + i) * G + g) * 2;
      s += o[0];
      q += o[1];
    }
    s = mud_wave_sum(s);
    q = mud_wave_sum(q);
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    if (mean_rstd && lane == 0) { mean_rstd[((int64_t)b * G + g) * 2] = meanf; mean_rstd[((int64_t)b * G + g) * 2 + 1] = rstd; }
    for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
      const float ga = gamma ? gamma[(int64_t)b * g_bstride + c] : 1.0f;
      const float be = beta ? beta[(int64_t)b * g_bstride + c] : 0.0f;
      const float sc = ga * rstd;
      scale[(int64_t)b * ld_ss + c] = sc;
      shift[(int64_t)b * ld_ss + c] = be - meanf * sc;
    }
  }
}

__global__ __launch_bounds__(256) void k_mean_finalize(const double* __restrict__ part, int nsplit, int C, double count,
                                                       float* __restrict__ out, int ldo) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    double s = 0.0;
    for (int i = 0; i < nsplit; ++i) s += part[(((int64_t)b * nsplit + i) * C + c) * 2];
    out[(int64_t)b * ldo + c] = (float)(s / count);
  }
}

static int gn_common_checks(const char* name, const float* x, int B, int64_t HW, int C, int ld, const void* ws) {
  MUD_REQUIRE(x && ws, "%s: null pointer", name);
  MUD_REQUIRE(B >= 0 && B <= 65535 && HW > 0, "%s: bad B=%d HW=%lld", name, B, (long long)HW);
  MUD_REQUIRE(C > 0 && C % 4 == 0 && C <= 1024 && ld >= C && ld % 4 == 0 && mud_aligned16(x),
              "%s: needs C%%4==0 (C=%d), ld%%4==0 (ld=%d), 16-byte aligned x", name, C, ld);
  return MUD_OK;
}

extern "C" int mud_gn_scale_shift(const float* x, int B, int64_t HW, int C, int ld, int G, float eps, const float* gamma,
                                  const float* beta, int64_t g_bstride, float* scale, float* shift, int ld_ss,
                                  float* mean_rstd, void* ws, void* stream) {
  if (int e = gn_common_checks("mud_gn_scale_shift", x, B, HW, C, ld, ws)) return e;
  MUD_REQUIRE(G > 0 && C % G == 0, "mud_gn_scale_shift: C=%d not divisible by G=%d", C, G);
  MUD_REQUIRE(scale && shift && ld_ss >= C, "mud_gn_scale_shift: bad scale/shift outputs");
  if (B == 0) return MUD_OK;
  const GnGeom g = gn_geom(HW, C);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((k_gn_partial<false>), dim3(g.nsplit, B), dim3(GN_THREADS), 2 * C * sizeof(double), s, x, HW, C, ld, G,
                     g.cols, g.rows, g.chunk, (double*)ws);
  MUD_CHECK_LAUNCH("mud_gn_scale_shift(partial)");
  hipLaunchKernelGGL(k_gn_finalize, dim3(B), dim3(256), 0, s, (const double*)ws, g.nsplit, C, G,
                     (double)HW * (double)(C / G), eps, gamma, beta, g_bstride, scale, shift, ld_ss, mean_rstd);
  MUD_CHECK_LAUNCH("mud_gn_scale_shift(finalize)");
  return MUD_OK;
}

__global__ __launch_bounds__(256) void k_gn_from_sums(const double* __restrict__ sums, int sums_ld, int C, int G, double count, float eps,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, int64_t g_bstride,
                                                      float* __restrict__ scale, float* __restrict__ shift, int ld_ss) {
  const int b = blockIdx.x, cpg = C / G;
  // one thread per group (G <= 32 in this model family; loop for generality)
  for (int g = threadIdx.x; g < G; g += 256) {
    double s = 0.0, q = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      s += sums[((int64_t)b * sums_ld + c) * 2];
      q += sums[((int64_t)b * sums_ld + c) * 2 + 1];
    }
    const double n = count * cpg, mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), meanf = (float)mean;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const float ga = gamma ? gamma[(int64_t)b * g_bstride + c] : 1.0f;
      const float be = beta ? beta[(int64_t)b * g_bstride + c] : 0.0f;
      const float sc = ga * rstd;
      scale[(int64_t)b * ld_ss + c] = sc;
      shift[(int64_t)b * ld_ss + c] = be - meanf * sc;
    }
  }
}

extern "C" int mud_gn_scale_shift_from_sums(const double* sums, int sums_ld, int B, int C, int G, double count, float eps,
                                            const float* gamma, const float* beta, int64_t g_bstride, float* scale, float* shift,
                                            int ld_ss, void* stream) {
  MUD_REQUIRE(sums && scale && shift, "mud_gn_scale_shift_from_sums: null pointer");
  MUD_REQUIRE(B >= 0 && C > 0 && G > 0 && C % G == 0 && sums_ld >= C && ld_ss >= C && count > 0, "mud_gn_scale_shift_from_sums: bad sizes");
  if (B == 0) return MUD_OK;
  hipLaunchKernelGGL(k_gn_from_sums, dim3(B), dim3(G <= 64 ? 64 : 256), 0, (hipStream_t)stream, sums, sums_ld, C, G, count, eps, gamma, beta,
                     g_bstride, scale, shift, ld_ss);
  MUD_CHECK_LAUNCH("mud_gn_scale_shift_from_sums");
  return MUD_OK;
}

extern "C" int mud_channel_mean(const float* x, int B, int64_t HW, int C, int ld, float* out, int ldo, void* ws, void* stream) {
  if (int e = gn_common_checks("mud_channel_mean", x, B, HW, C, ld, ws)) return e;
  MUD_REQUIRE(out && ldo >= C, "mud_channel_mean: bad output");
  if (B == 0) return MUD_OK;
  const GnGeom g = gn_geom(HW, C);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((k_gn_partial<true>), dim3(g.nsplit, B), dim3(GN_THREADS), 2 * C * sizeof(double), s, x, HW, C, ld, C,
                     g.cols, g.rows, g.chunk, (double*)ws);
  MUD_CHECK_LAUNCH("mud_channel_mean(partial)");
  hipLaunchKernelGGL(k_mean_finalize, dim3(B), dim3(256), 0, s, (const double*)ws, g.nsplit, C, (double)HW, out, ldo);
  MUD_CHECK_LAUNCH("mud_channel_mean(finalize)");
  return MUD_OK;
}
