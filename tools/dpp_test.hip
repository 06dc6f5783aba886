// verifies on hardware: DPP wave_shr:1 / wave_shl:1 / row_bcast15 semantics on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
  int v = threadIdx.x + 100;
  int r = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false); // wave_shr:1: lane L <- L-1, lane 0 keeps old
  int s = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, false); // wave_shl:1: lane L <- L+1, lane 63 keeps old
  int b = __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xE, 0x1, false); // row_bcast15, rows 1-3, bank 0
  int c = __builtin_amdgcn_update_dpp(b, v, 0x111, 0xF, 0xF, false);  // then row_shr:1 on top
  o[threadIdx.x] = r; o[64 + threadIdx.x] = s; o[128 + threadIdx.x] = b; o[192 + threadIdx.x] = c;
}
int main() {
  int* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); int h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad_r = 0, bad_s = 0, bad_c = 0;
  for (int l = 0; l < 64; ++l) { bad_r += h[l] != (l ? l - 1 + 100 : -1); bad_s += h[64 + l] != (l < 63 ? l + 1 + 100 : -1); bad_c += h[192 + l] != (l ? l - 1 + 100 : -1); }
  printf("wave_shr:1 mismatches %d, wave_shl:1 mismatches %d, bcast15+row_shr mismatches %d\n", bad_r, bad_s, bad_c);
  printf("shr: %d %d %d | %d %d | %d %d\n", h[0], h[1], h[2], h[15], h[16], h[31], h[32]);
  printf("bcast: %d %d %d %d %d | %d %d\n", h[128+15], h[128+16], h[128+17], h[128+19], h[128+20], h[128+32], h[128+48]);
  return 0;
}
