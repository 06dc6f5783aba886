// Streaming pass shaped like the projection tail (5 reads + 2 writes per cell, one workgroup per env),
// with the per-env field slices laid out field-major ([field][env][cell], what the library does) or
// env-major ([env][field][cell]).  Tells whether address-translation reach matters for this access shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct F4 { float v[4]; };
__device__ inline F4 ld(const float* p) { F4 r; __builtin_memcpy(&r, p, 16); return r; }
__device__ inline void st(float* p, F4 v) { __builtin_memcpy(p, &v, 16); }
__global__ __launch_bounds__(256, 4) void k(float* base, size_t env_stride, size_t field_stride, int ncell, int reps, int off) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = 0.f;
  float* b = base + blockIdx.x * env_stride + off;
  for (int r = 0; r < reps; ++r)
    for (int q = threadIdx.x * 4; q + 3 < ncell; q += 1024) {
      F4 a0 = ld(b + q), a1 = ld(b + field_stride + q), a2 = ld(b + 2 * field_stride + q), a3 = ld(b + 3 * field_stride + q),
         a4 = ld(b + 4 * field_stride + q);
      F4 o0, o1;
      for (int e = 0; e < 4; ++e) {
        o0.v[e] = a0.v[e] + a1.v[e] * a2.v[e];
        o1.v[e] = a3.v[e] + a4.v[e] * a2.v[e];
      }
      st(b + 5 * field_stride + q, o0);
      st(b + 6 * field_stride + q, o1);
    }
}
int main(int argc, char** argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 1024, ncell = 258 * 130, NF = 7;
  const size_t Ns = 33792;  // padded slice
  float* d;
  hipMalloc(&d, (size_t)E * NF * Ns * 4);
  hipMemset(d, 0, (size_t)E * NF * Ns * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 38 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int layout = 0; layout < 3; ++layout) {
    const size_t es = (layout == 1) ? NF * Ns : Ns, fs = (layout == 1) ? Ns : (size_t)E * Ns;
    const int off = (layout == 2) ? 1 : 0;  // 2: field-major, every 16-byte access 4 bytes off alignment
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(e0);
      k<<<E, 256, 38 * 1024>>>(d, es, fs, ncell - 4, 8, off);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)E * (ncell - 4) * 4 * NF * 8;
      printf("E=%d layout=%s: %.3f ms  %.2f TB/s  (%.1f us per pass per env)\n", E, layout == 1 ? "env-major" : (layout == 2 ? "field-major+4B" : "field-major"), ms,
             bytes / ms / 1e9, ms * 1e3 / 8);
    }
  }
  return 0;
}
