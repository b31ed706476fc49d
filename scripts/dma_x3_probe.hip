// Where does global_load_lds_dwordx3 put each lane's 12 bytes?  (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const int* src, int* out) {
  __shared__ int lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1;
  __syncthreads();
  const char* p = (const char*)src + threadIdx.x * 12;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)lds, 12, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  int *s, *o, h[1024];
  hipMalloc(&s, 4096); hipMalloc(&o, 4096);
  for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMemcpy(s, h, 4096, hipMemcpyHostToDevice);
  k<<<1, 64>>>(s, o);
  hipMemcpy(h, o, 4096, hipMemcpyDeviceToHost);
  for (int i = 0; i < 272; ++i) printf("%d%c", h[i], (i % 16 == 15) ? '\n' : ' ');
  return 0;
}
