// Shared device/host helpers for libmudiff_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mudiff_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// The 16-bit pieces of the split-precision matrix products (convolutions, attention): every fp32 operand v is carried as
// hi = fp16(v), lo = fp16(v - hi) and every product is issued as lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16 with fp32
// accumulation: ~2^-22 per product.  fp16, not bf16 (8-bit pieces, 2^-17 per product: what rounds 1 and 2 shipped): both take
// the same MFMA cycles, gfx950's MFMA keeps fp16 SUBNORMAL inputs (scripts/mfma_denorm_probe.hip), so values below 2^-14 lose
// only absolute precision (<= 2^-25), and the 5 extra mantissa bits cut the path's deviation from the fp32 reference 5-8x
// (profiles/r03_a_parity_plans.txt).  The one thing fp16 lacks is range: pieces SATURATE at +-65504 instead of overflowing.
typedef _Float16 mud_h16;
typedef mud_h16 mud_h16x4 __attribute__((ext_vector_type(4)));
typedef mud_h16 mud_h16x8 __attribute__((ext_vector_type(8)));

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
  int dev = -1;
  bool need() {          // true: the caller must (re)set the attribute on the current device, then call ok() once that succeeded
    dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
      dev = -1;
      return true;
    }
    return !done[dev];   // benign race: two threads may both set the attribute
  }
  void ok() {            // a FAILED hipFuncSetAttribute is retried by the next call instead of being remembered as done
    if (dev >= 0) done[dev] = true;
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

// sigmoid / silu with the hardware exp2 / rcp (each ~1 ulp): for epilogues whose values were produced by split-precision MFMAs
// (2^-15 .. 2^-22 per product) - the G2 gate convolution evaluates 4e8 sigmoids per batch-16 launch, and expf + a true divide there
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
// feeds an fp16 hi+lo split or a 16-tap filter, and a full-precision expf + divide would dominate
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

__device__ __forceinline__ f32x16 mud_mfma16(mud_h16x8 a, mud_h16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float mud_sat_h16(float v) { return __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f); }
__device__ __forceinline__ void mud_split1(float v, mud_h16& hi, mud_h16& lo) {
  v = mud_sat_h16(v);
  hi = (mud_h16)v;
  lo = (mud_h16)(v - (float)hi);
}
__device__ __forceinline__ void mud_split4(f32x4 v, mud_h16x4& hi, mud_h16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = mud_sat_h16(v[e]);
  hi = __builtin_convertvector(v, mud_h16x4);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), mud_h16x4);
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
