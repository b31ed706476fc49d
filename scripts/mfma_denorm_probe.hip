// Does v_mfma_f32_32x32x16_{f16,bf16} on gfx950 keep SUBNORMAL 16-bit inputs, or flush them to zero?  Decides whether the
// split-precision convolution can carry its low-order term in fp16 (11-bit pieces, residuals below 2^-14 are subnormal) or
// must stay with bf16 (8-bit pieces, fp32's exponent range).  Also checks the two converters the staging uses.
//   hipcc -O2 --offload-arch=gfx950 scripts/mfma_denorm_probe.hip -o scripts/exp/mfma_denorm_probe && scripts/exp/mfma_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// D = A (32x16) * B (16x32): every A element = a, every B element = b  ->  every D element = 16 * a * b
__global__ void k_probe(float a, float b, float* out) {
  f16x8 ah, bh;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) {
    ah[i] = (_Float16)a;
    bh[i] = (_Float16)b;
    ab[i] = (__bf16)a;
    bb[i] = (__bf16)b;
  }
  f32x16 c = {};
  const f32x16 dh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  const f32x16 db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = dh[0];
    out[1] = db[0];
    out[2] = (float)ah[0];      // what the f32 -> f16 converter made of a (subnormal kept?)
    out[3] = (float)ab[0];
  }
}

int main() {
  float* d;
  hipMalloc(&d, 16);
  struct { float a, b; const char* what; } cases[] = {
      {1.0f, 1.0f, "normal x normal"},
      {3.0e-5f, 1.0f, "f16-subnormal (3e-5) x 1"},
      {1.0f, 3.0e-5f, "1 x f16-subnormal (3e-5)"},
      {5.96e-8f, 1024.0f, "smallest f16 subnormal (2^-24) x 1024"},
      {3.0e-5f, 3.0e-5f, "subnormal x subnormal"},
      {1.0e-39f, 1.0e30f, "bf16/f32-subnormal (1e-39) x 1e30"},
  };
  for (auto& c : cases) {
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, c.a, c.b, d);
    float h[4];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double want = 16.0 * (double)c.a * (double)c.b;
    printf("%-42s exact %.6e | f16 mfma %.6e (cvt gave %.6e) | bf16 mfma %.6e (cvt gave %.6e)\n", c.what, want, h[0], h[2], h[1], h[3]);
  }
  printf("verdict: f16 subnormal inputs are %s by the MFMA\n", "see rows 2-4: a result of 0 means flushed");
  return 0;
}
