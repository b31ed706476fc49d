// The small-M dense path: z-mapping MLP, timestep MLP, and every per-block Dense_0 / AdaGN style
// Linear (batched by the host into one weight matrix per source vector).  M = batch is tiny, so
// this is a weight-streaming GEMV: one wave per output feature, lanes stride K, weights are read
// once per 8 samples straight into registers (no LDS), butterfly reduction at the end.
#include "mud_common.h"

#define DENSE_BCHUNK 8

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
