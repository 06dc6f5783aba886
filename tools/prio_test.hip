// Where s_setprio's value can be read back: USER_PRIO = bits 4:3 of the STATUS register (hwreg 2) on gfx950?
//   hipcc -O2 --offload-arch=gfx950 -o tools/bin/prio_test tools/prio_test.hip && tools/bin/prio_test
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  unsigned a[4], s[4];
  __builtin_amdgcn_s_setprio(0);
  a[0] = __builtin_amdgcn_s_getreg((1 << 11) | (3 << 6) | 2);
  s[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 2);
  __builtin_amdgcn_s_setprio(1);
  a[1] = __builtin_amdgcn_s_getreg((1 << 11) | (3 << 6) | 2);
  s[1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 2);
  __builtin_amdgcn_s_setprio(2);
  a[2] = __builtin_amdgcn_s_getreg((1 << 11) | (3 << 6) | 2);
  s[2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 2);
  __builtin_amdgcn_s_setprio(3);
  a[3] = __builtin_amdgcn_s_getreg((1 << 11) | (3 << 6) | 2);
  s[3] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 2);
  __builtin_amdgcn_s_setprio(0);
  if (threadIdx.x == 0)
    for (int i = 0; i < 4; ++i) out[i] = a[i], out[4 + i] = s[i];
}
int main() {
  unsigned *d, h[8];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i) printf("s_setprio %d: field(3,2) = %u   STATUS = %#010x\n", i, h[i], h[4 + i]);
  return 0;
}
