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

extern "C" int mud_dense(const float* in, int ldi, const float* W, const float* bias, float* out, int ldo, int B, int K, int N,
                         int act_in, int act_out, void* stream) {
  MUD_REQUIRE(in && W && out, "mud_dense: null pointer");
  MUD_REQUIRE(B >= 0 && K > 0 && N > 0 && ldi >= K && ldo >= N, "mud_dense: bad sizes B=%d K=%d N=%d ldi=%d ldo=%d", B, K, N, ldi, ldo);
  if (B == 0) return MUD_OK;
  int64_t blocks = mud_cdiv(N, 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_dense, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, in, ldi, W, bias, out, ldo, B, K, N, act_in, act_out);
  MUD_CHECK_LAUNCH("mud_dense");
  return MUD_OK;
}
