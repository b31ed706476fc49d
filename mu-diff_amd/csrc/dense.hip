// The small-M dense path: z-mapping MLP, timestep MLP, and every per-block Dense_0 / AdaGN style
// Linear (batched by the host into one weight matrix per source vector).  M = batch is tiny, so
// this is a weight-streaming GEMV: one wave per output feature, lanes stride K, weights are read
// once per 8 samples straight into registers (no LDS), butterfly reduction at the end.
#include "mud_common.h"

#define DENSE_BCHUNK 8

__device__ __forceinline__ bool mud_dev_aligned16f(const float* p) { return (((uintptr_t)p) & 15u) == 0; }

__global__ __launch_bounds__(256) void k_dense(const float* __restrict__ in, int ldi, const float* __restrict__ W,
                                               const float* __restrict__ bias, float* __restrict__ out, int ldo, int B, int K,
                                               int N, int act_in, int act_out) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  for (int n = wave_global; n < N; n += nwaves) {
    const float* w = W + (int64_t)n * K;
    for (int b0 = 0; b0 < B; b0 += DENSE_BCHUNK) {
      float acc[DENSE_BCHUNK];
#pragma unroll
      for (int j = 0; j < DENSE_BCHUNK; ++j) acc[j] = 0.f;
      for (int k = lane; k < K; k += 64) {
        const float wv = w[k];
#pragma unroll
        for (int j = 0; j < DENSE_BCHUNK; ++j) {
          if (b0 + j < B) {
            float v = in[(int64_t)(b0 + j) * ldi + k];
            if (act_in != MUD_ACT_NONE) v = mud_act(v, act_in);
            acc[j] = fmaf(wv, v, acc[j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < DENSE_BCHUNK; ++j) {
        const float s = mud_wave_sum(acc[j]);
        if (lane == 0 && b0 + j < B) out[(int64_t)(b0 + j) * ldo + n] = mud_act(s + (bias ? bias[n] : 0.f), act_out);
      }
    }
  }
}

// Lane-per-output form (K % 4 == 0): the wave-per-output kernel above spends most of its instructions on the 8 butterfly
// reductions per output and re-activates the inputs for every output.  Here a 256-thread block owns 64 outputs: lane = output
// (it streams its own weight row as float4s), the 4 waves split K, the activated inputs sit in LDS and are read as
// broadcasts, every lane keeps one accumulator per sample, and the 4 partial sums meet in LDS: no cross-lane reduction.
#define DENSE_TB 16                               // samples per pass
__global__ __launch_bounds__(256) void k_dense_lane(const float* __restrict__ in, int ldi, const float* __restrict__ W,
                                                    const float* __restrict__ bias, float* __restrict__ out, int ldo, int B, int K,
                                                    int N, int act_in, int act_out) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float* xin = dsm;                               // [DENSE_TB][K] activated inputs
  float* part = dsm + DENSE_TB * K;               // [4][DENSE_TB][64] partial sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int K4 = K >> 2, per = (K4 + 3) >> 2;     // float4 steps per wave
  const int k0 = wave * per, k1 = min(K4, k0 + per);
  const float* wrow = W + (int64_t)min(n, N - 1) * K;
  for (int b0 = 0; b0 < B; b0 += DENSE_TB) {
    const int nb = min(DENSE_TB, B - b0);
    __syncthreads();
    for (int i = tid; i < nb * K; i += 256) {
      const int bb = i / K, k = i - bb * K;
      float v = in[(int64_t)(b0 + bb) * ldi + k];
      if (act_in != MUD_ACT_NONE) v = mud_act(v, act_in);
      xin[bb * K + k] = v;
    }
    __syncthreads();
    float acc[DENSE_TB];
#pragma unroll
    for (int j = 0; j < DENSE_TB; ++j) acc[j] = 0.f;
    for (int k4 = k0; k4 < k1; ++k4) {
      const f32x4 w = *(const f32x4*)(wrow + 4 * k4);
#pragma unroll
      for (int j = 0; j < DENSE_TB; ++j) {
        if (j < nb) {
          const f32x4 x = *(const f32x4*)(xin + j * K + 4 * k4);       // same address in every lane: LDS broadcast
          acc[j] = fmaf(w[0], x[0], fmaf(w[1], x[1], fmaf(w[2], x[2], fmaf(w[3], x[3], acc[j]))));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < DENSE_TB; ++j) part[(wave * DENSE_TB + j) * 64 + lane] = acc[j];
    __syncthreads();
    for (int i = tid; i < nb * 64; i += 256) {
      const int bb = i >> 6, l = i & 63, nn = blockIdx.x * 64 + l;
      if (nn < N) {
        const float sacc = part[(0 * DENSE_TB + bb) * 64 + l] + part[(1 * DENSE_TB + bb) * 64 + l] + part[(2 * DENSE_TB + bb) * 64 + l] +
                           part[(3 * DENSE_TB + bb) * 64 + l];
        out[(int64_t)(b0 + bb) * ldo + nn] = mud_act(sacc + (bias ? bias[nn] : 0.f), act_out);
      }
    }
  }
}

extern "C" int mud_dense(const float* in, int ldi, const float* W, const float* bias, float* out, int ldo, int B, int K, int N,
                         int act_in, int act_out, void* stream) {
  MUD_REQUIRE(in && W && out, "mud_dense: null pointer");
  MUD_REQUIRE(B >= 0 && K > 0 && N > 0 && ldi >= K && ldo >= N, "mud_dense: bad sizes B=%d K=%d N=%d ldi=%d ldo=%d", B, K, N, ldi, ldo);
  if (B == 0) return MUD_OK;
  if (K % 4 == 0 && K <= 512 && mud_aligned16(W)) {      // LDS: 16*K + 4096 floats <= 48 KiB
    const size_t lds = (size_t)(DENSE_TB * K + 4 * DENSE_TB * 64) * sizeof(float);
    hipLaunchKernelGGL(k_dense_lane, dim3((unsigned)mud_cdiv(N, 64)), dim3(256), lds, (hipStream_t)stream, in, ldi, W, bias, out, ldo, B, K, N,
                       act_in, act_out);
    MUD_CHECK_LAUNCH("mud_dense");
    return MUD_OK;
  }
  int64_t blocks = mud_cdiv(N, 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_dense, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, in, ldi, W, bias, out, ldo, B, K, N, act_in, act_out);
  MUD_CHECK_LAUNCH("mud_dense");
  return MUD_OK;
}


// ------------------------------------------------------------------------------------------------
// A whole small MLP in ONE launch: the z-mapping network (PixelNorm -> dense(nz, 256) -> SiLU -> 3 x [dense(256, 256) -> SiLU],
// reference ncsnpp_generator_adagn_feat.py:271-277) and the timestep MLP (dense -> SiLU -> dense, :301-305) were chains of
// 5 and 2 dependent launches of a few microseconds each - at one slice per step the chain's launch boundaries cost more
// than its arithmetic.  One workgroup per sample, activations ping-pong in LDS, a thread owns an output feature and streams
// its weight row (the inputs are LDS broadcasts), same multiply-add order along K as k_dense_lane's per-lane sums is NOT kept
// (one accumulator chain per output here): results agree with the separate launches to fp32 rounding.
// ------------------------------------------------------------------------------------------------
#define MUD_MLP_MAX_CHAINS 4
struct MlpPack {
  mud_mlp_args c[MUD_MLP_MAX_CHAINS];
};
__global__ __launch_bounds__(256) void k_mlp_chain(MlpPack pack) {
  extern __shared__ __attribute__((aligned(16))) float msm[];
  const mud_mlp_args& a = pack.c[blockIdx.y];     // independent chains side by side (blockIdx.y), one workgroup per sample
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;                           // block-uniform
  float* cur = msm;
  float* nxt = msm + a.maxdim;
  const int K0 = a.dims[0];
  float ssq = 0.f;
  for (int k = tid; k < K0; k += 256) {
    const float v = a.x[(int64_t)b * a.ldx + k];
    cur[k] = v;
    ssq += v * v;
  }
  if (a.pixel_norm) {                                  // x * rsqrt(mean(x^2) + 1e-8)   (PixelNorm, :44-49)
    __shared__ float red[4];
    ssq = mud_wave_sum(ssq);
    if ((tid & 63) == 0) red[tid >> 6] = ssq;
    __syncthreads();
    const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K0 + 1e-8f);
    for (int k = tid; k < K0; k += 256) cur[k] *= inv;
  }
  __syncthreads();
  for (int l = 0; l < a.nlayers; ++l) {
    const int K = a.dims[l], N = a.dims[l + 1];
    const bool last = l + 1 == a.nlayers;
    const float* W = a.W[l];
    const float* bias = a.b[l];
    for (int n = tid; n < N; n += 256) {
      const float* w = W + (int64_t)n * K;
      float acc = 0.f;
      int k = 0;
      if ((K & 3) == 0 && mud_dev_aligned16f(w)) {
        for (; k < K; k += 4) {
          const f32x4 wv = *(const f32x4*)(w + k);
          const f32x4 xv = *(const f32x4*)(cur + k);
          acc = fmaf(wv[0], xv[0], fmaf(wv[1], xv[1], fmaf(wv[2], xv[2], fmaf(wv[3], xv[3], acc))));
        }
      }
      for (; k < K; ++k) acc = fmaf(w[k], cur[k], acc);
      acc += bias ? bias[n] : 0.f;
      if (!last || a.act_last) acc = mud_act(acc, a.act);
      if (last) a.out[(int64_t)b * a.ldo + n] = acc;
      else nxt[n] = acc;
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
}

static int mlp_check(mud_mlp_args& a) {
  MUD_REQUIRE(a.x && a.out && a.nlayers >= 1 && a.nlayers <= MUD_MLP_MAX_LAYERS && a.B >= 0, "mud_mlp_chain: bad arguments");
  int maxdim = 0;
  for (int l = 0; l <= a.nlayers; ++l) {
    MUD_REQUIRE(a.dims[l] > 0 && a.dims[l] <= 4096, "mud_mlp_chain: layer width %d out of range", a.dims[l]);
    if (a.dims[l] > maxdim) maxdim = a.dims[l];
    if (l < a.nlayers) MUD_REQUIRE(a.W[l] != nullptr, "mud_mlp_chain: null weight matrix");
  }
  MUD_REQUIRE(a.ldx >= a.dims[0] && a.ldo >= a.dims[a.nlayers], "mud_mlp_chain: bad row pitches");
  a.maxdim = (maxdim + 3) & ~3;
  return MUD_OK;
}

extern "C" int mud_mlp_chains(const mud_mlp_args* ap, int n, void* stream) {
  MUD_REQUIRE(ap && n >= 1 && n <= MUD_MLP_MAX_CHAINS, "mud_mlp_chains: 1..%d chains per launch", MUD_MLP_MAX_CHAINS);
  MlpPack pack;
  int maxB = 0, maxdim = 0;
  for (int i = 0; i < n; ++i) {
    pack.c[i] = ap[i];
    const int rc = mlp_check(pack.c[i]);
    if (rc != MUD_OK) return rc;
    if (pack.c[i].B > maxB) maxB = pack.c[i].B;
    if (pack.c[i].maxdim > maxdim) maxdim = pack.c[i].maxdim;
  }
  for (int i = n; i < MUD_MLP_MAX_CHAINS; ++i) pack.c[i] = pack.c[0];
  if (maxB == 0) return MUD_OK;
  hipLaunchKernelGGL(k_mlp_chain, dim3(maxB, n), dim3(256), 2 * maxdim * sizeof(float), (hipStream_t)stream, pack);
  MUD_CHECK_LAUNCH("mud_mlp_chain");
  return MUD_OK;
}

extern "C" int mud_mlp_chain(const mud_mlp_args* ap, void* stream) {
  MUD_REQUIRE(ap, "mud_mlp_chain: null args");
  return mud_mlp_chains(ap, 1, stream);
}
