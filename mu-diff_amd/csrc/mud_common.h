// Shared device/host helpers for libmudiff_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mudiff_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

extern "C" void mud_set_error(const char* fmt, ...);

#define MUD_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) {                                               \
      mud_set_error(__VA_ARGS__);                                \
      return MUD_ERR_ARG;                                        \
    }                                                            \
  } while (0)

#define MUD_CHECK_LAUNCH(name)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      mud_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return MUD_ERR_LAUNCH;                                                     \
    }                                                                            \
  } while (0)

// Per-device "dynamic LDS size already raised for this kernel" flags: hipFuncSetAttribute acts on the calling thread's
// current device, so a process that drives several GPUs must repeat it on each.
struct mud_attr_once {
  bool done[64] = {};
  bool need() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
    if (done[d]) return false;
    done[d] = true;      // benign race: two threads may both set the attribute
    return true;
  }
};

static inline bool mud_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int64_t mud_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float mud_silu(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float mud_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ float mud_act(float v, int act) {
  switch (act) {
    case MUD_ACT_SIGMOID: return mud_sigmoid(v);
    case MUD_ACT_TANH: return tanhf(v);
    case MUD_ACT_SILU: return mud_silu(v);
    case MUD_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    default: return v;
  }
}

// sigmoid / silu with the hardware exp2 / rcp (each ~1 ulp): for epilogues whose values were produced by split-bf16 MFMAs
// (2^-17 per product) - the G2 gate convolution evaluates 4e8 sigmoids per batch-16 launch, and expf + a true divide there
// cost ~25 instructions per element against 5
__device__ __forceinline__ float mud_act_fast(float v, int act) {
  switch (act) {
    case MUD_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
    case MUD_ACT_SILU: return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
    case MUD_ACT_TANH: return tanhf(v);
    case MUD_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    default: return v;
  }
}

// v * sigmoid(v) with the hardware exp2 / rcp (each ~1 ulp, ~3e-7 relative overall): used where the value
// feeds a bf16 hi+lo split (2^-17) or a 16-tap filter, and a full-precision expf + divide would dominate
__device__ __forceinline__ float mud_fast_silu(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}

__device__ __forceinline__ float mud_prologue(float v, float sc, float sh, int mode) {
  if (mode == MUD_PRO_NONE) return v;
  if (mode == MUD_PRO_LRELU) return v > 0.f ? v : 0.2f * v;
  v = fmaf(v, sc, sh);
  return mode == MUD_PRO_AFFINE_SILU ? mud_silu(v) : v;
}
__device__ __forceinline__ float mud_prologue_fast(float v, float sc, float sh, int mode) {
  if (mode == MUD_PRO_NONE) return v;
  if (mode == MUD_PRO_LRELU) return v > 0.f ? v : 0.2f * v;
  v = fmaf(v, sc, sh);
  return mode == MUD_PRO_AFFINE_SILU ? mud_fast_silu(v) : v;
}

// 64-lane butterfly sums
__device__ __forceinline__ float mud_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double mud_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float mud_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
