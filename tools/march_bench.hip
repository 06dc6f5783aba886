// Cycles per step of the ordered Gauss-Seidel march (gs_march3 of bdim_v2.h) in isolation:
// one wave per workgroup marches NS strips of a 256x128 level out of LDS; WG workgroups per launch
// (256 = one per CU, 1024 = four per CU).  Prints ns and core cycles (2.4 GHz nominal) per step.
#include "../rlfluidcontrol_amd/csrc/bdim_v2.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace afc;
using namespace afc::prod;
#ifndef NW
#define NW 1  // marching waves per workgroup (all on the same rings: timing only)
#endif
#ifndef MARCH_CALL
#define MARCH_CALL                                           \
  if (D.zero_bnd)                                            \
    gs_march3<true, true>(L, rr, dr, zrow, dummy, I0);       \
  else                                                       \
    gs_march3<true, false>(L, rr, dr, zrow, dummy, I0)
#endif
__global__ __launch_bounds__(256, 4) void k(LevelDesc D, int n, int m, int nstrips, float *out, long long *ticks,
                                             const float *coef, int generic) {
  extern __shared__ float lds[];
  const int PW = ring_pitch(m);
  float *rr = lds + kPad, *dr = rr + kRing * PW + kPad, *zrow = dr + kRing * PW + kPad, *dummy = zrow + PW + kPad;
  for (int q = threadIdx.x; q < gs_lds_floats(m); q += blockDim.x) lds[q] = 0.f;
  __syncthreads();
  for (int q = threadIdx.x; q < kRing * PW; q += blockDim.x) {
    const int j = q % PW;
    if (j >= 1 && j <= m - 2) {
      rr[q] = 0.001f * (float)((q * 7 + blockIdx.x) % 97) - 0.04f;
      dr[q] = rr[q] * D.invtab[0];
    }
  }
  __syncthreads();
  Lv L;
  L.n = n;
  L.m = m;
  L.lx = coef;
  L.ly = coef + n * m + 256;
  L.inv = coef + 2 * (n * m + 256);
  L.r = L.x = L.d = nullptr;
  L.desc = &D;
  long long t0 = 0;
  if (threadIdx.x < 64 * NW) {
    t0 = wall_clock64();
    for (int s = 0; s < nstrips; ++s) {
      const int I0 = 17 + 16 * (s % 13);  // interior strips
      if (generic)
        gs_march3<false>(L, rr, dr, zrow, dummy, I0);
      else
        MARCH_CALL;
    }
    t0 = wall_clock64() - t0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ticks[blockIdx.x] = t0;
    float acc = 0.f;
    for (int q = 0; q < kRing * PW; q += 37) acc += dr[q];
    out[blockIdx.x] = acc;
  }
}
int main(int argc, char **argv) {
  const int n = 258, m = 130, nstrips = 200;
  LevelDesc D{};
  D.n = n;
  D.m = m;
  D.cu = 0.25f;
  D.zero_bnd = argc > 1 ? atoi(argv[1]) : 0;
  for (int i = 0; i < 9; ++i) D.invtab[i] = -1.f;
  D.sp_i0 = 1 << 30;
  D.sp_i1 = -1;
  D.sp_j0 = 1;  // generic = 1: every step takes the coefficient loads (special columns = all)
  D.sp_j1 = m - 2;
  const size_t lds_bytes = (gs_lds_floats(m) + 64) * sizeof(float);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  float *out;
  long long *ticks;
  hipMalloc(&out, 1024 * 4);
  hipMalloc(&ticks, 1024 * 8);
  const int steps_per_strip = (m - 2 + 15 + 3 * kLag + 3) & ~3;
  const int generic = argc > 2 ? atoi(argv[2]) : 0;
  float *coef;
  {
    std::vector<float> hc(3 * (n * m + 256), 0.25f);
    for (int q = 2 * (n * m + 256); q < 3 * (n * m + 256); ++q) hc[q] = -1.f;
    hipMalloc(&coef, hc.size() * 4);
    hipMemcpy(coef, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
  }
  for (int wg : {1, 256, 1024}) {
    for (int rep = 0; rep < 2; ++rep) {
      k<<<wg, 256, lds_bytes>>>(D, n, m, nstrips, out, ticks, coef, generic);
      hipDeviceSynchronize();
    }
    std::vector<long long> h(wg);
    std::vector<float> ho(wg);
    hipMemcpy(h.data(), ticks, wg * 8, hipMemcpyDeviceToHost);
    hipMemcpy(ho.data(), out, wg * 4, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= wg;
    const double ns = avg * 10.0 / ((double)nstrips * steps_per_strip);
    printf("generic=%d zero_bnd=%d workgroups=%4d: %.1f ns per step (%.0f cycles at 2.4 GHz), checksum %.6f\n", generic, D.zero_bnd, wg, ns,
           ns * 2.4, ho[0]);
  }
  return 0;
}
