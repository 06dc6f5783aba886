// Does the order in which a phase writes its rows change what the NEXT phase's reads cost (Infinity Cache / L2 reuse)?
// 1024 workgroups x 256 threads, each workgroup owns NF fields of n x m floats (the step's layout).  Kernel W writes all
// rows of its fields (8 bands of rows at a time, like project_tail_v2) ascending or descending inside a band; kernel R
// reads them four quarters at a time ascending (like advdif_v3).  Times R after W(asc), W(desc) and after a cache flush.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int n = 258, m = 130, NS = 33600;  // floats per field slot (>= n*m, multiple of 64)
__global__ __launch_bounds__(256, 4) void kw(float *base, int nf, int desc, int bands, float val) {
  float *f = base + (size_t)blockIdx.x * nf * NS;
  const int c4 = threadIdx.x & 31, rg = threadIdx.x >> 5;  // 32 chunks of 4 columns, 8 row groups
  const int rpg = 256 / bands;                             // rows per band
  for (int t = 0; t < rpg * bands / 8; ++t) {
    const int band = (rg + 8 * (t / rpg)) % bands, k = t % rpg;
    const int g = 1 + band * rpg + (desc ? rpg - 1 - k : k);
    for (int q = 0; q < nf; ++q) {
      float4 v = make_float4(val + g, val + c4, val, val + q);
      __builtin_memcpy(f + (size_t)q * NS + g * m + 1 + 4 * c4, &v, 16);
    }
  }
}
__global__ __launch_bounds__(256, 4) void kr(const float *base, int nf, float *out) {
  const float *f = base + (size_t)blockIdx.x * nf * NS;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  float acc = 0.f;
  for (int i = 0; i < 64; ++i) {
    const int g = 1 + 64 * w + i;
    for (int q = 0; q < nf; ++q) {
      float2 v;
      __builtin_memcpy(&v, f + (size_t)q * NS + g * m + 1 + 2 * l, 8);
      acc += v.x + v.y;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void kflush(float4 *p, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1, 2, 3, 4);
}
int main(int argc, char **argv) {
  const int nf = argc > 1 ? atoi(argv[1]) : 2, WG = argc > 2 ? atoi(argv[2]) : 1024;
  float *base, *out;
  float4 *fl;
  const size_t bytes = (size_t)WG * nf * NS * 4, flb = (size_t)2 << 30;
  hipMalloc(&base, bytes), hipMalloc(&out, 64), hipMalloc(&fl, flb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  printf("%d workgroups x %d fields: %.1f MB written, then read\n", WG, nf, bytes / 1e6);
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 5; ++mode) {
      // 0: W asc 8 bands | 1: W desc 8 bands | 2: W asc, flush between | 3: W asc 4 bands (= the reader's quarters) | 4: W desc 4 bands
      const int desc = (mode == 1 || mode == 4), bands = (mode >= 3) ? 4 : 8;
      kflush<<<2048, 256>>>(fl, flb / 16);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      kw<<<WG, 256>>>(base, nf, desc, bands, (float)rep);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float msw;
      hipEventElapsedTime(&msw, e0, e1);
      if (mode == 2) {
        kflush<<<2048, 256>>>(fl, flb / 16);
        hipDeviceSynchronize();
      }
      hipEventRecord(e0);
      kr<<<WG, 256>>>(base, nf, out);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float msr;
      hipEventElapsedTime(&msr, e0, e1);
      printf("mode %d (%s, %d bands%s): write %.3f ms (%.2f TB/s)  read %.3f ms (%.2f TB/s)\n", mode, desc ? "desc" : "asc ", bands,
             mode == 2 ? ", flushed" : "", msw, bytes / msw / 1e9, msr, bytes / msr / 1e9);
    }
  return 0;
}
