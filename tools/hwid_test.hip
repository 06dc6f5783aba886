// where do the waves of 1024 co-resident 256-thread / 40 KB-LDS workgroups land? (gfx950)
// prints, per workgroup, XCC / SE / CU and the SIMD of each of its four waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 4) void k(unsigned* o) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = 0.f;
  const long long t_end = wall_clock64() + 200000;  // 2 ms: everybody resident at once
  while (wall_clock64() < t_end) __builtin_amdgcn_s_sleep(64);
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
    o[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    o[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
}
int main() {
  const int nb = 1024;
  unsigned* d;
  hipMalloc(&d, nb * 8 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 38 * 1024);
  k<<<nb, 256, 38 * 1024>>>(d);
  std::vector<unsigned> h(nb * 8);
  hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu_wgs;
  int simd_hist[4][4] = {};
  for (int b = 0; b < nb; ++b) {
    unsigned key = 0;
    for (int w = 0; w < 4; ++w) {
      const unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      simd_hist[w][simd]++;
      if (b < 24 || (b % 256) < 2) printf("wg %4d wave %d: xcc %u se %u sh %u cu %2u simd %u waveslot %u\n", b, w, xcc, se, sh, cu, simd, hw & 15);
    }
    cu_wgs[key].push_back(b);
  }
  printf("distinct CUs: %zu\n", cu_wgs.size());
  for (int w = 0; w < 4; ++w) printf("wave %d -> simd histogram %d %d %d %d\n", w, simd_hist[w][0], simd_hist[w][1], simd_hist[w][2], simd_hist[w][3]);
  int shown = 0;
  for (auto& kv : cu_wgs) {
    if (shown++ >= 12) break;
    printf("cu %05x:", kv.first);
    for (int b : kv.second) printf(" %d(s%u)", b, (h[(b * 4) * 2] >> 4) & 3);
    printf("\n");
  }
  std::map<size_t, int> cnt;
  for (auto& kv : cu_wgs) cnt[kv.second.size()]++;
  for (auto& c : cnt) printf("%d CUs host %zu workgroups\n", c.second, c.first);
  return 0;
}
