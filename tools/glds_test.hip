// feasibility: __builtin_amdgcn_global_load_lds on gfx950 (4 and 16 bytes per lane), 4-byte aligned source
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* src, float* out, int off) {
  extern __shared__ float lds[];
  const int t = threadIdx.x;
  // 256 threads x 16 B = 4 KiB per instruction; LDS dest = lds base (uniform) + lane*16 within each wave's 1 KiB
  const float* g = src + off + t * 4;
  float* l = lds + (t >> 6) * 256;  // wave-uniform LDS base; hardware adds lane*16
  __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(src + off + 1024 + t, (__attribute__((address_space(3))) void*)(lds + 1024 + (t >> 6) * 64), 4, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = t; i < 1280; i += 256) out[i] = lds[i];
}
int main() {
  float *d, *o; hipMalloc(&d, 8192 * 4); hipMalloc(&o, 2048 * 4);
  float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = i; hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  for (int off : {0, 1, 3}) {
    k<<<1, 256, 8192>>>(d, o, off); float r[1280]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += (r[i] != off + i); for (int i = 0; i < 256; ++i) bad += (r[1024 + i] != off + 1024 + i);
    printf("off %d: mismatches %d (r[0..3]=%g %g %g %g, r[1024]=%g)\n", off, bad, r[0], r[1], r[2], r[3], r[1024]);
  }
  return 0;
}
