// prints, for each lane, which source lane each cross-lane primitive delivers (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned shift16(unsigned prev, int lane) {
  auto a = __builtin_amdgcn_permlane16_swap(prev, prev, false, false);   // a[0]=[r0,r0,r2,r2] a[1]=[r1,r1,r3,r3] ?
  auto b = __builtin_amdgcn_permlane32_swap(a[0], a[1], false, false);   // b[0]=[a0.r0,a0.r1,a1.r0,a1.r1] ?
  return ((lane >> 4) == 3) ? a[0] : b[0];
}
__global__ void k(unsigned* o) {
  const int lane = threadIdx.x;
  unsigned v = lane;
  auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  o[lane] = a[0]; o[64 + lane] = a[1]; o[128 + lane] = b[0]; o[192 + lane] = b[1];
  o[256 + lane] = shift16(v, lane);
  o[320 + lane] = __builtin_amdgcn_update_dpp(1000u + lane, v, 0x111, 0xF, 0xF, false);
  o[384 + lane] = __builtin_amdgcn_update_dpp(1000u + lane, v, 0x138, 0xF, 0xF, false);  // wave_shr:1
}
int main() {
  unsigned* d; hipMalloc(&d, 448 * 4); k<<<1, 64>>>(d); unsigned h[448]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[] = {"pl16.0", "pl16.1", "pl32.0", "pl32.1", "shift16", "row_shr1", "wave_shr1"};
  for (int t = 0; t < 7; ++t) { printf("%-9s", names[t]); for (int l = 0; l < 64; ++l) printf(" %u", h[t * 64 + l]); printf("\n"); }
  return 0;
}
