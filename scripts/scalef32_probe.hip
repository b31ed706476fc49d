#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  const float a = in[0], b = in[1], sc = in[2];
  s16x2 old = {0, 0};
  s16x2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, sc, false);
  out[0] = (unsigned)(unsigned short)r[0];
  out[1] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffff;
}
int main() {
  float h[3]; unsigned o[2]; float* din; unsigned* dout;
  (void)hipMalloc(&din, 12); (void)hipMalloc(&dout, 8);
  const float cases[][3] = {{1.0f, 8.0f, 1.0f}, {1.0f, 8.0f, 0.125f}, {1.0f, 8.0f, 8.0f}, {1.0f, 8.0f, 0.1f}, {3e-5f, 1e-4f, 1.0f / 8192.0f}, {3e-5f, 1e-4f, 8192.0f}, {1000.f, -1000.f, 1.0f}};
  for (auto& c : cases) {
    h[0] = c[0]; h[1] = c[1]; h[2] = c[2];
    (void)hipMemcpy(din, h, 12, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, din, dout);
    (void)hipMemcpy(o, dout, 8, hipMemcpyDeviceToHost);
    printf("a=%g b=%g scale=%g: scalef32 bytes 0x%04x | plain cvt bytes 0x%04x\n", c[0], c[1], c[2], o[0], o[1]);
  }
  printf("(e4m3: 1.0 = 0x38, 8.0 = 0x50, 0.125 = 0x20, 64 = 0x68, 0.25 = 0x28, 448 = 0x7e)\n");
  return 0;
}
