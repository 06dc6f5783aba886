// One level's up-leg (prolongation increment + 4 ordered sweeps + increment) in isolation, on synthetic data:
// mg_up_smooth (fill | barrier | march | barrier | exit per strip) against mg_up_pipe (the march back to back, waves 1-3
// exiting / filling behind it), WG workgroups per launch (256 = one per CU, 1024 = four per CU), every workgroup on its
// own arrays.  Prints us per call and whether x, r and the three sums agree bit for bit.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -o tools/bin/upleg_bench tools/upleg_bench.hip
#include "../rlfluidcontrol_amd/csrc/bdim_v2.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace afc;
using namespace afc::prod;
#ifndef NREP
#define NREP 8
#endif
struct Args {
  LevelDesc D, Dc;
  int n, m, nc, mc;
  float *r, *x, *xc;       // per workgroup: n*m, n*m, nc*mc floats
  const float *coef;       // lx | ly | inv, shared
  float *sums;             // 3 per workgroup
  long long *ticks;
  int mode, need_r, rmode;
};
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(Args A) {
  extern __shared__ float lds[];
  const size_t b = blockIdx.x, N = (size_t)A.n * A.m, Nc = (size_t)A.nc * A.mc;
  Lv F, C;
  F.n = A.n, F.m = A.m;
  F.lx = A.coef, F.ly = A.coef + N + 256, F.inv = A.coef + 2 * (N + 256);
  F.r = A.r + b * N, F.x = A.x + b * N, F.d = nullptr, F.desc = &A.D;
  C = F;
  C.n = A.nc, C.m = A.mc, C.x = A.xc + b * Nc, C.r = nullptr, C.desc = &A.Dc;
  const Lv Fu = uni(F), Cu = uni(C);
  ProfT pf;
  float rr = 0.f, linf = 0.f, xs = 0.f;
  long long t0 = wall_clock64();
  for (int rep = 0; rep < NREP; ++rep) {
    if (MODE == 0)
      mg_up_smooth<16>(true, Fu, Cu, A.need_r != 0, lds, nullptr, &rr, &linf, pf, &xs, A.rmode);
    else if (MODE == 1)
      mg_up_pipe<16, false>(true, Fu, Cu, A.need_r != 0, lds, nullptr, &rr, &linf, pf, &xs, A.rmode);
    else
      mg_up_pipe<16, true>(true, Fu, Cu, A.need_r != 0, lds, nullptr, &rr, &linf, pf, &xs, A.rmode);
  }
  t0 = wall_clock64() - t0;
  if (threadIdx.x == 0) {
    A.ticks[b] = t0;
    A.sums[3 * b] = rr, A.sums[3 * b + 1] = linf, A.sums[3 * b + 2] = xs;
  }
}
static unsigned rng = 12345u;
static float frand() {
  rng = rng * 1664525u + 1013904223u;
  return (float)((rng >> 8) & 0xFFFF) / 65536.f - 0.5f;
}
int main(int argc, char **argv) {
  const int m_ = argc > 1 ? atoi(argv[1]) : 128, n_ = 2 * m_, zb = argc > 2 ? atoi(argv[2]) : 0;
  const int need_r = argc > 3 ? atoi(argv[3]) : 1, rmode = argc > 4 ? atoi(argv[4]) : 1;
  const int n = n_ + 2, m = m_ + 2, nc = n_ / 2 + 2, mc = m_ / 2 + 2;
  const size_t N = (size_t)n * m, Nc = (size_t)nc * mc;
  Args A{};
  A.n = n, A.m = m, A.nc = nc, A.mc = mc, A.need_r = need_r, A.rmode = rmode;
  LevelDesc D{};
  D.n = n, D.m = m, D.cu = 0.25f, D.zero_bnd = zb;
  for (int i = 0; i < 9; ++i) D.invtab[i] = -1.f - 0.125f * (zb ? i : 0);
  D.sp_i0 = 1 << 30, D.sp_i1 = -1, D.sp_j0 = 1, D.sp_j1 = 0;
  A.D = D;
  A.Dc = D;
  A.Dc.n = nc, A.Dc.m = mc;
  const int WGMAX = 1024;
  std::vector<float> hr(N * WGMAX), hx(N * WGMAX), hxc(Nc * WGMAX);
  for (auto &v : hr) v = 0.01f * frand();
  for (auto &v : hx) v = frand();
  for (auto &v : hxc) v = 0.1f * frand();
  std::vector<float> hc(3 * (N + 256), 0.25f);
  for (size_t q = 2 * (N + 256); q < hc.size(); ++q) hc[q] = -1.f;
  float *coef, *r, *x, *xc, *sums;
  long long *ticks;
  hipMalloc(&coef, hc.size() * 4);
  hipMemcpy(coef, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&r, hr.size() * 4), hipMalloc(&x, hx.size() * 4), hipMalloc(&xc, hxc.size() * 4);
  hipMalloc(&sums, WGMAX * 12), hipMalloc(&ticks, WGMAX * 8);
  A.coef = coef, A.r = r, A.x = x, A.xc = xc, A.sums = sums, A.ticks = ticks;
  const size_t lds_bytes = (gs_lds_floats(m) + 64) * sizeof(float) > 37376 ? (gs_lds_floats(m) + 64) * sizeof(float) : 37376;
  hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  const bool big_ok = (pipe_big_lds_floats(m) + 16) * sizeof(float) <= lds_bytes;
  printf("level %dx%d zero_bnd=%d need_r=%d rmode=%d, %d calls per launch, LDS %zu B\n", n_, m_, zb, need_r, rmode, NREP, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int wg : {1, 256, 1024}) {
    std::vector<float> outx[3], outr[3], outs[3];
    double us[3] = {0, 0, 0}, usw[3] = {0, 0, 0};
    for (int mode = 0; mode < (big_ok ? 3 : 2); ++mode) {
      A.mode = mode;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(r, hr.data(), N * wg * 4, hipMemcpyHostToDevice);
        hipMemcpy(x, hx.data(), N * wg * 4, hipMemcpyHostToDevice);
        hipMemcpy(xc, hxc.data(), Nc * wg * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        if (mode == 0)
          k<0><<<wg, 256, lds_bytes>>>(A);
        else if (mode == 1)
          k<1><<<wg, 256, lds_bytes>>>(A);
        else
          k<2><<<wg, 256, lds_bytes>>>(A);
        hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) {
          printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
          return 1;
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      us[mode] = best * 1000.0 / NREP;
      std::vector<long long> ht(wg);
      hipMemcpy(ht.data(), ticks, wg * 8, hipMemcpyDeviceToHost);
      double avg = 0;
      for (auto v : ht) avg += (double)v;
      usw[mode] = avg / wg * 0.01 / NREP;
      outx[mode].resize(N * wg), outr[mode].resize(N * wg), outs[mode].resize(3 * wg);
      hipMemcpy(outx[mode].data(), x, N * wg * 4, hipMemcpyDeviceToHost);
      hipMemcpy(outr[mode].data(), r, N * wg * 4, hipMemcpyDeviceToHost);
      hipMemcpy(outs[mode].data(), sums, 3 * wg * 4, hipMemcpyDeviceToHost);
    }
    const bool okx = memcmp(outx[0].data(), outx[1].data(), N * wg * 4) == 0;
    const bool okr = memcmp(outr[0].data(), outr[1].data(), N * wg * 4) == 0;
    const bool oks = memcmp(outs[0].data(), outs[1].data(), 3 * wg * 4) == 0;
    if (big_ok) {
      const bool ok2 = memcmp(outx[0].data(), outx[2].data(), N * wg * 4) == 0 && memcmp(outr[0].data(), outr[2].data(), N * wg * 4) == 0 &&
                       memcmp(outs[0].data(), outs[2].data(), 3 * wg * 4) == 0;
      printf("workgroups=%4d: big rings %8.1f us per call (%8.1f) ratio %.3f %s\n", wg, us[2], usw[2], us[2] / us[0], ok2 ? "same" : "DIFFER");
    }
    printf("workgroups=%4d: mg_up_smooth %8.1f us per call (per-workgroup clock %8.1f) | mg_up_pipe %8.1f (%8.1f) | ratio %.3f | x %s r %s sums %s (rr %.9g / %.9g)\n",
           wg, us[0], usw[0], us[1], usw[1], us[1] / us[0], okx ? "same" : "DIFFER", okr ? "same" : "DIFFER",
           oks ? "same" : "DIFFER", outs[0][0], outs[1][0]);
  }
  return 0;
}
