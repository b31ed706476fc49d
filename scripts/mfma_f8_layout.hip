// Operand lane maps of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3), its E8M0 scale operands, and what v_cvt_pk_fp8_f32 does
// with out-of-range inputs - checked with exact small-integer data, as cdna_hip_programming.md section 3 asks before relying on a map.
//   hipcc -O2 --offload-arch=gfx950 scripts/mfma_f8_layout.hip -o scripts/exp/mfma_f8_layout && scripts/exp/mfma_f8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned char to_e4m3(float v) {        // through the hardware converter
  const int p = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false);
  return (unsigned char)(p & 0xff);
}

// hyp 0: lane (r, h) byte j holds k = 32h + j.   hyp 1: k = 16h + (j & 15) + 32 (j >> 4)
__device__ int kmap(int hyp, int h, int j) { return hyp == 0 ? 32 * h + j : 16 * h + (j & 15) + 32 * (j >> 4); }

__global__ void k_test(const float* A, const float* B, float* D, int hyp, int sa, int sb) {   // A [32][64], B [64][32], D [32][32]
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  union { i32x8 v; unsigned char b[32]; } a, b;
  for (int j = 0; j < 32; ++j) {
    const int k = kmap(hyp, h, j);
    a.b[j] = to_e4m3(A[r * 64 + k]);
    b.b[j] = to_e4m3(B[k * 32 + r]);
  }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 0, sa, 0, sb);
  for (int reg = 0; reg < 16; ++reg) D[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = c[reg];
}

__global__ void k_cvt(const float* in, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(in[i], -in[i], 0, false) & 0xffff;
}

int main() {
  float hA[32 * 64], hB[64 * 32], hD[32 * 32], ref[32 * 32];
  srand(7);
  for (auto& v : hA) v = (float)(rand() % 9 - 4);
  for (auto& v : hB) v = (float)(rand() % 9 - 4);
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      float s = 0;
      for (int k = 0; k < 64; ++k) s += hA[m * 64 + k] * hB[k * 32 + n];
      ref[m * 32 + n] = s;
    }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int hyp = 0; hyp < 2; ++hyp) {
    hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp, 0x7f7f7f7f, 0x7f7f7f7f);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double e = 0;
    for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(hD[i] - ref[i]));
    printf("operand map hypothesis %d (%s): max |D - ref| = %g  %s\n", hyp, hyp == 0 ? "k = 32h + j" : "k = 16h + (j&15) + 32(j>>4)", e, e == 0 ? "EXACT" : "");
  }
  // scales: A x 2^3 (E8M0 127 + 3 in every byte), B x 2^-1
  hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, 0x82828282, 0x7e7e7e7e);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  double e4 = 0, e1 = 0;
  for (int i = 0; i < 1024; ++i) { e4 = fmax(e4, fabs(hD[i] - 4 * ref[i])); e1 = fmax(e1, fabs(hD[i] - ref[i])); }
  printf("scale_a = 2^3, scale_b = 2^-1 (hypothesis 0 operands): max |D - 4 ref| = %g, max |D - ref| = %g\n", e4, e1);
  // only byte 0 of the scale words set: which byte does opsel 0 read?
  hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, 0x7f7f7f82, 0x7f7f7f7f);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  double e8 = 0; e1 = 0;
  for (int i = 0; i < 1024; ++i) { e8 = fmax(e8, fabs(hD[i] - 8 * ref[i])); e1 = fmax(e1, fabs(hD[i] - ref[i])); }
  printf("scale_a byte 0 = 2^3, other bytes 1: max |D - 8 ref| = %g, max |D - ref| = %g\n", e8, e1);
  // converter
  float hin[8] = {448.f, 449.f, 480.f, 1000.f, 1e30f, 0.001f, 0.0009765625f, 3.3f};
  unsigned hout[8];
  float* din; unsigned* dout;
  hipMalloc(&din, sizeof hin); hipMalloc(&dout, sizeof hout);
  hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, din, dout, 8);
  hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("cvt_pk_fp8_f32(%g, %g) -> 0x%02x 0x%02x\n", hin[i], -hin[i], hout[i] & 0xff, hout[i] >> 8);
  return 0;
}
