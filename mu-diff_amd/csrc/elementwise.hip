// HBM-bound elementwise kernels of the sampling path: Gaussian posterior step, q_sample,
// embeddings, row softmax and the G2 gated feature fusion.  16 B per lane everywhere
// (coalesced 1 KiB per wave instruction); grids capped at ~8 blocks/CU and grid-strided.
#include "mud_common.h"

static inline int mud_grid_1d(int64_t work_items, int block) {
  int64_t g = mud_cdiv(work_items, block);
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------
// posterior: reference order of operations, no FMA contraction (engine/test.py:150-177)
// ------------------------------------------------------------------------------------------------
template <bool DUAL, bool VEC>
__global__ __launch_bounds__(256) void k_posterior(const float* __restrict__ x01, const float* __restrict__ x02,
                                                   const float* __restrict__ xt, const float* __restrict__ noise,
                                                   const int64_t* __restrict__ t, const float* __restrict__ c1t,
                                                   const float* __restrict__ c2t, const float* __restrict__ sdt,
                                                   int ntab, float* __restrict__ out, int64_t per_sample) {
#pragma clang fp contract(off)   // every * and + below rounds separately, like the reference's chain of torch ops
  const int b = blockIdx.y;
  int64_t ti = t[b];
  ti = ti < 0 ? 0 : (ti >= ntab ? ntab - 1 : ti);
  const float c1 = c1t[ti], c2 = c2t[ti];
  const float sd = (ti == 0) ? 0.0f : sdt[ti];   // nonzero_mask * exp(0.5*log_var)
  const int64_t base = (int64_t)b * per_sample;
  constexpr int V = VEC ? 4 : 1;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < per_sample;
       i += (int64_t)gridDim.x * blockDim.x * V) {
    float a[V], bb[V], x[V], n[V], o[V];
    if (VEC) {
      *(f32x4*)a = *(const f32x4*)(x01 + base + i);
      if (DUAL) *(f32x4*)bb = *(const f32x4*)(x02 + base + i);
      *(f32x4*)x = *(const f32x4*)(xt + base + i);
      *(f32x4*)n = *(const f32x4*)(noise + base + i);
    } else {
      a[0] = x01[base + i];
      if (DUAL) bb[0] = x02[base + i];
      x[0] = xt[base + i];
      n[0] = noise[base + i];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float cx = c2 * x[j];
      float mean = c1 * a[j] + cx;
      if (DUAL) {
        const float m2 = c1 * bb[j] + cx;
        mean = (mean + m2) * 0.5f;   // (mean1 + mean2) / 2, exact halving
      }
      o[j] = mean + sd * n[j];
    }
    if (VEC) *(f32x4*)(out + base + i) = *(f32x4*)o;
    else out[base + i] = o[0];
  }
}

extern "C" int mud_posterior_sample(const float* x01, const float* x02, const float* xt, const float* noise,
                                    const int64_t* t, const float* coef1, const float* coef2, const float* std_tab,
                                    int ntab, float* out, int B, int64_t per_sample, void* stream) {
  MUD_REQUIRE(B >= 0 && per_sample >= 0 && ntab > 0, "mud_posterior_sample: bad sizes B=%d per_sample=%lld ntab=%d", B,
              (long long)per_sample, ntab);
  if (B == 0 || per_sample == 0) return MUD_OK;   // empty batch: nothing to do (pointers may be null)
  MUD_REQUIRE(x01 && xt && noise && t && coef1 && coef2 && std_tab && out, "mud_posterior_sample: null pointer");
  MUD_REQUIRE(B <= 65535, "mud_posterior_sample: B=%d exceeds 65535", B);
  const bool vec = (per_sample % 4 == 0) && mud_aligned16(x01) && mud_aligned16(xt) && mud_aligned16(noise) &&
                   mud_aligned16(out) && (!x02 || mud_aligned16(x02));
  dim3 grid(mud_grid_1d(per_sample / (vec ? 4 : 1), 256), B), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(D, V) hipLaunchKernelGGL((k_posterior<D, V>), grid, block, 0, s, x01, x02, xt, noise, t, coef1, coef2, std_tab, ntab, out, per_sample)
  if (x02) { if (vec) LAUNCH(true, true); else LAUNCH(true, false); }
  else     { if (vec) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  MUD_CHECK_LAUNCH("mud_posterior_sample");
  return MUD_OK;
}

__global__ __launch_bounds__(256) void k_q_sample(const float* __restrict__ x, const float* __restrict__ noise,
                                                  const int64_t* __restrict__ t, int toff, const float* __restrict__ at,
                                                  const float* __restrict__ st, int ntab, float* __restrict__ out,
                                                  int64_t per_sample) {
#pragma clang fp contract(off)
  const int b = blockIdx.y;
  int64_t ti = t[b] + toff;
  ti = ti < 0 ? 0 : (ti >= ntab ? ntab - 1 : ti);
  const float a = at[ti], sg = st[ti];
  const int64_t base = (int64_t)b * per_sample;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * blockDim.x)
    out[base + i] = a * x[base + i] + sg * noise[base + i];
}

extern "C" int mud_q_sample(const float* x, const float* noise, const int64_t* t, int toff, const float* a_tab,
                            const float* s_tab, int ntab, float* out, int B, int64_t per_sample, void* stream) {
  MUD_REQUIRE(B >= 0 && per_sample >= 0 && ntab > 0 && B <= 65535, "mud_q_sample: bad sizes");
  if (B == 0 || per_sample == 0) return MUD_OK;
  MUD_REQUIRE(x && noise && t && a_tab && s_tab && out, "mud_q_sample: null pointer");
  dim3 grid(mud_grid_1d(per_sample, 256), B), block(256);
  hipLaunchKernelGGL(k_q_sample, grid, block, 0, (hipStream_t)stream, x, noise, t, toff, a_tab, s_tab, ntab, out, per_sample);
  MUD_CHECK_LAUNCH("mud_q_sample");
  return MUD_OK;
}

// ------------------------------------------------------------------------------------------------
// embeddings
// ------------------------------------------------------------------------------------------------
__global__ void k_timestep_embedding(const int64_t* __restrict__ t, float* __restrict__ out, int B, int dim, float neg_log_over) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * dim) return;
  const int b = i / dim, j = i % dim;
  float v = 0.0f;
  if (j < 2 * half) {
    const int k = j < half ? j : j - half;
    const float freq = expf((float)k * neg_log_over);
    const float arg = (float)t[b] * freq;
    v = j < half ? sinf(arg) : cosf(arg);
  }
  out[i] = v;   // odd dim: last column zero-padded (layers.py:476-477)
}

extern "C" int mud_timestep_embedding(const int64_t* t, float* out, int B, int dim, float max_positions, void* stream) {
  MUD_REQUIRE(t && out && B >= 0 && dim >= 4, "mud_timestep_embedding: bad args");
  if (B == 0) return MUD_OK;
  const int half = dim / 2;
  const float neg = -(float)(log((double)max_positions) / (double)(half - 1));
  hipLaunchKernelGGL(k_timestep_embedding, dim3(mud_cdiv((int64_t)B * dim, 256)), dim3(256), 0, (hipStream_t)stream, t, out, B, dim, neg);
  MUD_CHECK_LAUNCH("mud_timestep_embedding");
  return MUD_OK;
}

// Gaussian Fourier features of log(sigma) (reference layerspp.py:68-77 applied to torch.log(time_cond),
// ncsnpp_generator_adagn_feat.py:288-289): x_proj = ((log(t) * W[k]) * 2) * pi in fp32, out = [sin | cos].
__global__ void k_fourier_embedding(const float* __restrict__ t, const float* __restrict__ W, float* __restrict__ out, int B, int n) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  const int b = i / n, k = i % n;
  const float arg = ((logf(t[b]) * W[k]) * 2.0f) * 3.14159265358979323846f;
  out[(int64_t)b * 2 * n + k] = sinf(arg);
  out[(int64_t)b * 2 * n + n + k] = cosf(arg);
}

extern "C" int mud_fourier_embedding(const float* t, const float* W, float* out, int B, int n, void* stream) {
  MUD_REQUIRE(B >= 0 && n > 0, "mud_fourier_embedding: bad sizes");
  if (B == 0) return MUD_OK;
  MUD_REQUIRE(t && W && out, "mud_fourier_embedding: null pointer");
  hipLaunchKernelGGL(k_fourier_embedding, dim3(mud_cdiv((int64_t)B * n, 256)), dim3(256), 0, (hipStream_t)stream, t, W, out, B, n);
  MUD_CHECK_LAUNCH("mud_fourier_embedding");
  return MUD_OK;
}

__global__ __launch_bounds__(64) void k_pixel_norm(const float* __restrict__ z, float* __restrict__ out, int K) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float ss = 0.f;
  for (int k = lane; k < K; k += 64) { const float v = z[(int64_t)b * K + k]; ss += v * v; }
  ss = mud_wave_sum(ss);
  const float d = sqrtf(ss / (float)K + 1e-8f);
  for (int k = lane; k < K; k += 64) out[(int64_t)b * K + k] = z[(int64_t)b * K + k] / d;
}

extern "C" int mud_pixel_norm(const float* z, float* out, int B, int K, void* stream) {
  MUD_REQUIRE(z && out && B >= 0 && K > 0, "mud_pixel_norm: bad args");
  if (B == 0) return MUD_OK;
  hipLaunchKernelGGL(k_pixel_norm, dim3(B), dim3(64), 0, (hipStream_t)stream, z, out, K);
  MUD_CHECK_LAUNCH("mud_pixel_norm");
  return MUD_OK;
}

// ------------------------------------------------------------------------------------------------
// row softmax, in place: one 256-thread block per row, three L2-resident sweeps
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_softmax_rows(float* __restrict__ s, int64_t rows, int n, int ld) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    float* row = s + r * (int64_t)ld;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, row[i]);
    m = mud_wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float e = expf(row[i] - m); row[i] = e; sum += e; }
    sum = mud_wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    const float inv = 1.0f / sum;
    for (int i = threadIdx.x; i < n; i += 256) row[i] *= inv;
  }
}

// rows of up to 256*4*SM_V floats (n % 4 == 0) held in registers: one read, one write
#define SM_V 4
__global__ __launch_bounds__(256) void k_softmax_rows_reg(float* __restrict__ s, int64_t rows, int n, int ld) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    float* row = s + r * (int64_t)ld;
    f32x4 v[SM_V];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < SM_V; ++k) {
      const int i = (k * 256 + threadIdx.x) * 4;
      if (i < n) {
        v[k] = *(const f32x4*)(row + i);
        m = fmaxf(fmaxf(fmaxf(m, v[k][0]), fmaxf(v[k][1], v[k][2])), v[k][3]);
      }
    }
    m = mud_wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < SM_V; ++k) {
      if ((k * 256 + threadIdx.x) * 4 < n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[k][e] = expf(v[k][e] - m); sum += v[k][e]; }
      }
    }
    sum = mud_wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    const float inv = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < SM_V; ++k) {
      const int i = (k * 256 + threadIdx.x) * 4;
      if (i < n) *(f32x4*)(row + i) = v[k] * inv;
    }
  }
}

extern "C" int mud_softmax_rows(float* s, int64_t rows, int n, int ld, void* stream) {
  MUD_REQUIRE(s && rows >= 0 && n > 0 && ld >= n, "mud_softmax_rows: bad args");
  if (rows == 0) return MUD_OK;
  if (n % 4 == 0 && ld % 4 == 0 && n <= 256 * 4 * SM_V && mud_aligned16(s)) {
    const int grid = (int)(rows < 256 * 32 ? rows : 256 * 32);
    hipLaunchKernelGGL(k_softmax_rows_reg, dim3(grid), dim3(256), 0, (hipStream_t)stream, s, rows, n, ld);
    MUD_CHECK_LAUNCH("mud_softmax_rows");
    return MUD_OK;
  }
  const int grid = (int)(rows < 256 * 32 ? rows : 256 * 32);
  hipLaunchKernelGGL(k_softmax_rows, dim3(grid), dim3(256), 0, (hipStream_t)stream, s, rows, n, ld);
  MUD_CHECK_LAUNCH("mud_softmax_rows");
  return MUD_OK;
}

// ------------------------------------------------------------------------------------------------
// G2 feature fusion elementwise pieces (ncsnpp_generator_adagn_feat.py:778-788)
// ------------------------------------------------------------------------------------------------
template <int MODE>  // 0: a*b   1: g*att + (1-g)*other (+ optional per-channel statistics of the result)
__global__ __launch_bounds__(256) void k_ew3(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                             const float* __restrict__ c, int ldc, float* __restrict__ out, int ldo,
                                             int64_t hw, int C4, double* __restrict__ stats, int stats_ld) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* st_lds = (float*)smem_raw;             // [C][2]
  const int bi = blockIdx.y;
  if (stats) {
    for (int i = threadIdx.x; i < 8 * C4; i += 256) st_lds[i] = 0.f;
    __syncthreads();
  }
  const int64_t total = hw * C4, base = (int64_t)bi * hw;
  // a thread keeps the same 4 channels over its strided loop when 256 % C4 == 0 or the stride is a multiple of C4
  f32x4 ls = {0.f, 0.f, 0.f, 0.f}, lq = {0.f, 0.f, 0.f, 0.f};
  const int64_t stride = ((int64_t)gridDim.x * blockDim.x / C4) * C4;   // multiple of C4: channel quad is loop-invariant
  const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(first % C4) * 4;
  const bool active = first < stride;            // threads past the last whole multiple of C4 would duplicate work
  for (int64_t i = first; active && i < total; i += stride) {
    const int64_t p = base + i / C4;
    const f32x4 va = *(const f32x4*)(a + p * lda + q);
    const f32x4 vb = *(const f32x4*)(b + p * ldb + q);
    f32x4 o;
    if (MODE == 0) o = va * vb;
    else {
      const f32x4 vc = *(const f32x4*)(c + p * ldc + q);
      o = va * vb + (1.0f - va) * vc;
    }
    *(f32x4*)(out + p * ldo + q) = o;
    ls += o;
    lq += o * o;
  }
  if (stats) {
    if (active && first < total) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        atomicAdd(&st_lds[(q + j) * 2], ls[j]);
        atomicAdd(&st_lds[(q + j) * 2 + 1], lq[j]);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * C4; i += 256)
      atomicAdd(stats + ((int64_t)bi * stats_ld + (i >> 1)) * 2 + (i & 1), (double)st_lds[i]);
  }
}

static int ew_check(const char* name, const void* a, int lda, const void* b, int ldb, const void* o, int ldo, int C) {
  MUD_REQUIRE(a && b && o, "%s: null pointer", name);
  MUD_REQUIRE(C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0 && mud_aligned16(a) && mud_aligned16(b) && mud_aligned16(o),
              "%s: needs C, ld multiples of 4 and 16-byte aligned views", name);
  return MUD_OK;
}

static dim3 ew_grid(int64_t hw, int C4, int B) {
  int64_t blocks = mud_cdiv(hw * C4, 256 * 4);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  // grid.x * 256 must be >= C4 so that the loop stride (a multiple of C4) is positive
  while (blocks * 256 < C4) ++blocks;
  return dim3((unsigned)blocks, B);
}

extern "C" int mud_mul(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int64_t npix, int C, void* stream) {
  if (int e = ew_check("mud_mul", a, lda, b, ldb, out, ldo, C)) return e;
  if (npix == 0) return MUD_OK;
  hipLaunchKernelGGL((k_ew3<0>), ew_grid(npix, C / 4, 1), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
                     (const float*)nullptr, 0, out, ldo, npix, C / 4, (double*)nullptr, 0);
  MUD_CHECK_LAUNCH("mud_mul");
  return MUD_OK;
}

extern "C" int mud_gate_mix(const float* g, int ldg, const float* att, int lda, const float* other, int ldb, float* out,
                            int ldo, int B, int64_t hw, int C, double* stats, int stats_ld, void* stream) {
  if (int e = ew_check("mud_gate_mix", g, ldg, att, lda, out, ldo, C)) return e;
  MUD_REQUIRE(other && ldb % 4 == 0 && mud_aligned16(other), "mud_gate_mix: bad `other` view");
  MUD_REQUIRE(B >= 0 && B <= 65535 && hw >= 0 && (!stats || stats_ld >= C), "mud_gate_mix: bad sizes");
  if (B == 0 || hw == 0) return MUD_OK;
  hipLaunchKernelGGL((k_ew3<1>), ew_grid(hw, C / 4, B), dim3(256), stats ? 2 * C * sizeof(float) : 0, (hipStream_t)stream, g, ldg, att,
                     lda, other, ldb, out, ldo, hw, C / 4, stats, stats_ld);
  MUD_CHECK_LAUNCH("mud_gate_mix");
  return MUD_OK;
}

// ------------------------------------------------------------------------------------------------
// minibatch standard deviation (backbones/discriminator.py:246-254): tiny (the critic's last feature map is 4x4)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minibatch_stddev(const float* __restrict__ x, int M, int64_t hw, int C, int ld, int group,
                                                          float* __restrict__ out) {
  __shared__ double red[4];
  const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = hw * C;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int64_t p = i / C;
    const int c = (int)(i % C);
    float mean = 0.f;
    for (int g = 0; g < group; ++g) mean += x[(((int64_t)g * M + m) * hw + p) * ld + c];
    mean /= (float)group;
    float var = 0.f;
    for (int g = 0; g < group; ++g) {
      const float d = x[(((int64_t)g * M + m) * hw + p) * ld + c] - mean;
      var += d * d;
    }
    acc += (double)sqrtf(var / (float)group + 1e-8f);
  }
  acc = mud_wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)n);
    for (int g = 0; g < group; ++g) out[g * M + m] = s;
  }
}

extern "C" int mud_minibatch_stddev(const float* x, int B, int64_t hw, int C, int ld, int group, float* out, void* stream) {
  MUD_REQUIRE(x && out, "mud_minibatch_stddev: null pointer");
  MUD_REQUIRE(B > 0 && group > 0 && B % group == 0 && hw > 0 && C > 0 && ld >= C, "mud_minibatch_stddev: B=%d must be a multiple of group=%d", B, group);
  hipLaunchKernelGGL(k_minibatch_stddev, dim3(B / group), dim3(256), 0, (hipStream_t)stream, x, B / group, hw, C, ld, group, out);
  MUD_CHECK_LAUNCH("mud_minibatch_stddev");
  return MUD_OK;
}

// ------------------------------------------------------------------------------------------------
// Bilinear resize of planes, torch `F.interpolate(mode='bilinear', align_corners=False)` semantics
// (the reference's calls: engine/test_volume.py:274 slices -> image_size, engine/train.py:959
// uncertainty map -> image size).  Half-pixel centres: src = (dst + 0.5) * in/out - 0.5 clamped at 0;
// neighbours i0 = min(int(src), in-1), i1 = i0 + (i0 < in-1); weight of i1 = clamp(src - i0, 0, 1).
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__device__ __forceinline__ void mud_bilinear_src(int dst, float scale, int in, int& i0, int& i1, float& w1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = min((int)src, in - 1);
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  w1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
}

__global__ __launch_bounds__(256) void k_resize_bilinear(const float* __restrict__ in, int64_t planes, int H, int W, int Ho, int Wo,
                                                         float sy, float sx, float* __restrict__ out) {
  const int64_t total = planes * Ho * Wo;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % Wo);
    const int64_t r = idx / Wo;
    const int oy = (int)(r % Ho);
    const float* ip = in + (r / Ho) * H * W;
    int y0, y1, x0, x1;
    float wy, wx;
    mud_bilinear_src(oy, sy, H, y0, y1, wy);
    mud_bilinear_src(ox, sx, W, x0, x1, wx);
    const float top = (1.f - wx) * ip[(int64_t)y0 * W + x0] + wx * ip[(int64_t)y0 * W + x1];
    const float bot = (1.f - wx) * ip[(int64_t)y1 * W + x0] + wx * ip[(int64_t)y1 * W + x1];
    out[idx] = (1.f - wy) * top + wy * bot;
  }
}

extern "C" int mud_resize_bilinear(const float* in, int64_t planes, int H, int W, int Ho, int Wo, float* out, void* stream) {
  MUD_REQUIRE(planes >= 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "mud_resize_bilinear: bad sizes");
  if (planes == 0) return MUD_OK;
  MUD_REQUIRE(in && out, "mud_resize_bilinear: null pointer");
  int64_t blocks = mud_cdiv(planes * Ho * Wo, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(k_resize_bilinear, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, in, planes, H, W, Ho, Wo,
                     (float)H / (float)Ho, (float)W / (float)Wo, out);
  MUD_CHECK_LAUNCH("mud_resize_bilinear");
  return MUD_OK;
}

// out = clamp(x * scale + shift, lo, hi): the sampler's [-1,1] -> [0,1] image mapping (engine/test_volume.py:281,
// engine/test.py to_range_0_1) with scale = shift = 0.5, lo = 0, hi = 1.
__global__ __launch_bounds__(256) void k_affine_clamp(const float* __restrict__ x, int64_t n, float scale, float shift, float lo, float hi,
                                                      float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = fminf(fmaxf(x[i] * scale + shift, lo), hi);
}

extern "C" int mud_affine_clamp(const float* x, int64_t n, float scale, float shift, float lo, float hi, float* out, void* stream) {
  MUD_REQUIRE(n >= 0, "mud_affine_clamp: bad size");
  if (n == 0) return MUD_OK;
  MUD_REQUIRE(x && out, "mud_affine_clamp: null pointer");
  int64_t blocks = mud_cdiv(n, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(k_affine_clamp, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, n, scale, shift, lo, hi, out);
  MUD_CHECK_LAUNCH("mud_affine_clamp");
  return MUD_OK;
}
