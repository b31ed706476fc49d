// Does MODE.FP16_OVFL (bit 23 of the MODE register) make gfx950's f32 -> f16 and f32 -> fp8 converters SATURATE instead of producing
// inf / NaN?  If so the staging code of the split-precision convolutions can drop its v_med3 clamps (12 of ~62 vector instructions
// per staged float4).    hipcc -O2 --offload-arch=gfx950 scripts/fp16_ovfl_probe.hip -o scripts/exp/fp16_ovfl_probe && scripts/exp/fp16_ovfl_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void k_probe(const float* in, float* out, int set) {
  if (set) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  const float v = in[threadIdx.x];
  f32x2 p = {v, -v};
  const f16x2 h = __builtin_convertvector(p, f16x2);                     // v_cvt_pk_f16_f32 (RNE)
  const int f8 = __builtin_amdgcn_cvt_pk_fp8_f32(v, -v, 0, false);      // OCP e4m3
  const int b8 = __builtin_amdgcn_cvt_pk_bf8_f32(v, -v, 0, false);      // OCP e5m2
  out[threadIdx.x * 4 + 0] = (float)h[0];
  out[threadIdx.x * 4 + 1] = (float)h[1];
  out[threadIdx.x * 4 + 2] = (float)(f8 & 0xffff);
  out[threadIdx.x * 4 + 3] = (float)(b8 & 0xffff);
}
int main() {
  const float h[8] = {1.0f, 65504.0f, 65520.0f, 1e6f, 448.0f, 480.0f, 1000.0f, 1e9f};
  float *din, *dout, o[32];
  hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  for (int set = 0; set < 2; ++set) {
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(8), 0, 0, din, dout, set);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("MODE.FP16_OVFL = %d\n", set);
    for (int i = 0; i < 8; ++i)
      printf("   v = %-10g -> f16 (%g, %g)   e4m3 bytes 0x%04x   e5m2 bytes 0x%04x\n", h[i], o[i * 4], o[i * 4 + 1], (unsigned)o[i * 4 + 2], (unsigned)o[i * 4 + 3]);
  }
  return 0;
}
