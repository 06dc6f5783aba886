// Which allocations stream fast?  afc_create's placement search picks the best of up to 32 hipMalloc candidates by a copy
// probe (two kinds, 4-5 % apart in the probe, 6-7 % in the step kernel).  This tool looks for a DETERMINISTIC recipe:
// the same probe over (a) plain hipMalloc, (b) one hipMemCreate handle mapped into a reserved range, (c) the range
// built from chunks of 2 MB / 64 MB / 256 MB (hipMemCreate + hipMemMap), each several times in one process.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy(const f4v *__restrict__ a, f4v *__restrict__ b, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256 * 8;
  for (size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; i < n4; i += stride) {
    f4v v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (i + q * 256 < n4) ? a[i + q * 256] : f4v{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (i + q * 256 < n4) b[i + q * 256] = v[q];
  }
}
static hipEvent_t e0, e1;
static float probe(void *q, size_t bytes) {
  const size_t n4 = bytes / 32;
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0, nullptr);
    k_copy<<<2048, 256>>>((const f4v *)q, (f4v *)q + n4, n4);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  return (float)(2.0 * 16.0 * (double)n4 / (best * 1e-3) / 1e9);
}
#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                   \
      return 1;                                                               \
    }                                                                         \
  } while (0)
int main(int argc, char **argv) {
  const size_t want = argc > 1 ? (size_t)atoll(argv[1]) << 20 : (size_t)1740 << 20;
  const int reps = argc > 2 ? atoi(argv[2]) : 8;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  size_t gmin = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  printf("slab %zu MB, allocation granularity: recommended %zu KB, minimum %zu KB\n", want >> 20, gran >> 10, gmin >> 10);
  {  // (a) hipMalloc, all held at once
    std::vector<void *> held;
    printf("hipMalloc         :");
    for (int r = 0; r < reps; ++r) {
      void *q;
      CK(hipMalloc(&q, want));
      hipMemset(q, 0, want);
      held.push_back(q);
      if (r == 0) probe(q, want);
      printf(" %5.0f", probe(q, want));
    }
    printf("  GB/s\n");
    for (void *q : held) hipFree(q);
  }
  for (size_t chunk : {(size_t)0, (size_t)2 << 20, (size_t)64 << 20, (size_t)256 << 20}) {
    printf("VMM chunk %4zu MB :", chunk >> 20);
    std::vector<std::pair<hipDeviceptr_t, size_t>> ranges;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    for (int r = 0; r < reps; ++r) {
      const size_t c = chunk ? chunk : ((want + gran - 1) / gran) * gran;
      const size_t total = ((want + c - 1) / c) * c;
      hipDeviceptr_t base;
      CK(hipMemAddressReserve(&base, total, 0, 0, 0));
      for (size_t off = 0; off < total; off += c) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, c, &prop, 0));
        CK(hipMemMap((hipDeviceptr_t)((char *)base + off), c, 0, h, 0));
        handles.push_back(h);
      }
      hipMemAccessDesc ad{};
      ad.location = prop.location;
      ad.flags = hipMemAccessFlagsProtReadWrite;
      CK(hipMemSetAccess(base, total, &ad, 1));
      hipMemset(base, 0, want);
      ranges.push_back({base, total});
      printf(" %5.0f", probe(base, want));
      fflush(stdout);
    }
    printf("  GB/s\n");
    for (auto &rg : ranges) {
      hipMemUnmap(rg.first, rg.second);
      hipMemAddressFree(rg.first, rg.second);
    }
    for (auto h : handles) hipMemRelease(h);
  }
  return 0;
}
