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
      mg_up_pipe(true, Fu, Cu, A.need_r != 0, lds, nullptr, &rr, &linf, pf, &xs, A.rmode);
    else {
      afc::lat::ProfT pfl;
      afc::lat::up_tr(true, Fu, Cu, A.need_r != 0, lds, &rr, &linf, pfl, &xs, A.rmode);
    }
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
  const int box = argc > 5 ? atoi(argv[5]) : 0;  // 1: a box of perturbed coefficients around the middle (rows that cross a body)
  LevelDesc D{};
  D.n = n, D.m = m, D.cu = 0.25f, D.zero_bnd = zb;
  for (int rc = 0; rc < 3; ++rc)
    for (int cc = 0; cc < 3; ++cc) {
      const float lw = (rc == 1 && zb) ? 0.f : D.cu, le = (rc == 2 && zb) ? 0.f : D.cu;
      const float ls = (cc == 1 && zb) ? 0.f : D.cu, ln = (cc == 2 && zb) ? 0.f : D.cu;
      D.invtab[3 * rc + cc] = -1.f / (lw + le + ls + ln);
    }
  // the real arrays: analytic values (boundary faces zeroed on a zero_bnd level), perturbed inside the box
  std::vector<float> hc(3 * (N + 256), 0.f);
  float *hlx = hc.data(), *hly = hc.data() + N + 256, *hiv = hc.data() + 2 * (N + 256);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      const size_t k = (size_t)i * m + j;
      hlx[k] = (zb && (i <= 1 || i >= n - 1)) ? 0.f : D.cu;
      hly[k] = (zb && (j <= 1 || j >= m - 1)) ? 0.f : D.cu;
    }
  if (box) {
    const int bi0 = n_ / 4 - n_ / 24, bi1 = n_ / 4 + n_ / 24, bj0 = m_ / 2 - m_ / 9, bj1 = m_ / 2 + m_ / 9;
    for (int i = bi0; i <= bi1; ++i)
      for (int j = bj0; j <= bj1; ++j) {
        const size_t k = (size_t)i * m + j;
        hlx[k] = D.cu * (0.25f + 0.5f * (frand() + 0.5f));
        hly[k] = D.cu * (0.25f + 0.5f * (frand() + 0.5f));
      }
  }
  for (int i = 1; i < n - 1; ++i)
    for (int j = 1; j < m - 1; ++j) {
      const size_t k = (size_t)i * m + j;
      hiv[k] = -1.f / (hlx[k] + hlx[k + m] + hly[k] + hly[k + 1]);
    }
  D.sp_i0 = D.sp_j0 = 1 << 30, D.sp_i1 = D.sp_j1 = -1;
  for (int i = 1; i < n - 1; ++i)
    for (int j = 1; j < m - 1; ++j) {
      const size_t k = (size_t)i * m + j;
      const int rc = (i == 1) ? 1 : ((i == n - 2) ? 2 : 0), cc = (j == 1) ? 1 : ((j == m - 2) ? 2 : 0);
      const float lw = (rc == 1 && zb) ? 0.f : D.cu, le = (rc == 2 && zb) ? 0.f : D.cu;
      const float ls = (cc == 1 && zb) ? 0.f : D.cu, ln = (cc == 2 && zb) ? 0.f : D.cu;
      if (hlx[k] != lw || hlx[k + m] != le || hly[k] != ls || hly[k + 1] != ln || hiv[k] != D.invtab[3 * rc + cc]) {
        if (i < D.sp_i0) D.sp_i0 = i;
        if (i > D.sp_i1) D.sp_i1 = i;
        if (j < D.sp_j0) D.sp_j0 = j;
        if (j > D.sp_j1) D.sp_j1 = j;
      }
    }
  if (D.sp_i1 < 0) D.sp_i0 = 1 << 30, D.sp_i1 = -1, D.sp_j0 = 1, D.sp_j1 = 0;
  printf("special rows [%d, %d] columns [%d, %d]\n", D.sp_i0, D.sp_i1, D.sp_j0, D.sp_j1);
  A.D = D;
  A.Dc = D;
  A.Dc.n = nc, A.Dc.m = mc;
  const int WGMAX = 1024;
  std::vector<float> hr(N * WGMAX), hx(N * WGMAX), hxc(Nc * WGMAX);
  for (auto &v : hr) v = 0.01f * frand();
  for (auto &v : hx) v = frand();
  for (auto &v : hxc) v = 0.1f * frand();
  float *coef, *r, *x, *xc, *sums;
  long long *ticks;
  hipMalloc(&coef, hc.size() * 4);
  hipMemcpy(coef, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&r, hr.size() * 4), hipMalloc(&x, hx.size() * 4), hipMalloc(&xc, hxc.size() * 4);
  hipMalloc(&sums, WGMAX * 12), hipMalloc(&ticks, WGMAX * 8);
  A.coef = coef, A.r = r, A.x = x, A.xc = xc, A.sums = sums, A.ticks = ticks;
  size_t lds_bytes = (gs_lds_floats(m) + 64) * sizeof(float) > 37376 ? (gs_lds_floats(m) + 64) * sizeof(float) : 37376;
  const size_t tmb = (tm_lds_floats(m, D.sp_i0, D.sp_i1, D.sp_j0, D.sp_j1) + 64) * sizeof(float);
  if (tmb > lds_bytes) lds_bytes = tmb;
  printf("mg_up_tr needs %zu B of LDS\n", tmb);
  hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  printf("level %dx%d zero_bnd=%d need_r=%d rmode=%d, %d calls per launch, LDS %zu B\n", n_, m_, zb, need_r, rmode, NREP, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int only_wg = argc > 6 ? atoi(argv[6]) : 0;
  for (int wg : {1, 256, 1024}) {
    if (only_wg && wg != only_wg) continue;
    std::vector<float> outx[3], outr[3], outs[3];
    double us[3] = {0, 0, 0}, usw[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; ++mode) {
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
        else {
#ifdef AFC_TM_PROF
          long long z[64] = {0};
          hipMemcpyToSymbol(HIP_SYMBOL(afc::lat::g_tm_prof), z, sizeof(z));
#endif
          k<2><<<wg, 256, lds_bytes>>>(A);
        }
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
    for (int mode = 1; mode < 3; ++mode) {
      const bool okx = memcmp(outx[0].data(), outx[mode].data(), N * wg * 4) == 0;
      const bool okr = memcmp(outr[0].data(), outr[mode].data(), N * wg * 4) == 0;
      const bool oks = memcmp(outs[0].data(), outs[mode].data(), 3 * wg * 4) == 0;
      printf("workgroups=%4d: mg_up_smooth %8.1f us per call (per-workgroup clock %8.1f) | %s %8.1f (%8.1f) | ratio %.3f | x %s r %s sums %s (rr %.9g / %.9g)\n",
             wg, us[0], usw[0], mode == 1 ? "mg_up_pipe" : "mg_up_tr  ", us[mode], usw[mode], us[mode] / us[0], okx ? "same" : "DIFFER", okr ? "same" : "DIFFER",
             oks ? "same" : "DIFFER", outs[0][0], outs[mode][0]);
      if (!okx && wg == 1) {  // where: first differing cells
        int shown = 0;
        for (size_t q = 0; q < N && shown < 12; ++q)
          if (memcmp(&outx[0][q], &outx[mode][q], 4)) {
            printf("   x differs at row %zu col %zu: %.9g vs %.9g\n", q / m, q % m, outx[0][q], outx[mode][q]);
            ++shown;
          }
      }
    }
  }
#ifdef AFC_TM_PROF
  {
    long long hp[64];
    hipMemcpyFromSymbol(hp, HIP_SYMBOL(afc::lat::g_tm_prof), sizeof(hp));
    printf("prof (cycles, last launch, workgroup 0, last call): init %lld | to march loop %lld | march loop %lld | march end -> barrier %lld | fill wave end %lld | exit wave end %lld (after march start) | waited %lld cycles in %lld waits (all calls)\n",
           hp[1] - hp[0], hp[2] - hp[1], hp[3] - hp[2], hp[4] - hp[3], hp[5] - hp[2], hp[6] - hp[2], hp[8], hp[9]);
  }
#endif
  return 0;
}
