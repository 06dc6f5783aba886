// Dependent-issue cost of VALU instructions for ONE wave on gfx950 (cycles per instruction, shader clock):
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_lat tools/valu_lat.hip && tools/bin/valu_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__device__ __forceinline__ unsigned long long clk() { return __builtin_readcyclecounter(); }
template <int MODE>
__global__ void k(float *out, unsigned long long *t, int iters) {
  float a = threadIdx.x * 1e-3f, b = a + 1.f, c = b + 1.f, d = c + 1.f, e2 = d + 1.f;
  const float s = 1.0000001f;
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = clk();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // one dependent chain of adds
      REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(s));)
    } else if (MODE == 1) {  // two independent chains, interleaved
      REP16(asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(s));)
    } else if (MODE == 2) {  // four independent chains
      REP16(asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));)
    } else if (MODE == 3) {  // dependent chain: dpp mov + mul (the march's first two links)
      REP16(asm volatile("s_nop 1\n\tv_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mul_f32 %0, %2, %1"
                         : "+v"(a), "+v"(b) : "v"(s));)
    } else if (MODE == 4) {  // dependent chain: fused dpp mul
      REP16(asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(s));)
    } else if (MODE == 5) {  // the march's step: 10 instructions, 8 of them one chain
      REP16(asm volatile("v_mul_f32 %1, %3, %2\n\tv_mul_f32 %2, %3, %1\n\t"
                         "v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mul_f32 %1, %3, %1\n\tv_add_f32 %1, %1, %2\n\tv_mul_f32 %2, %3, %0\n\t"
                         "v_add_f32 %1, %1, %2\n\tv_add_f32 %1, %1, %2\n\tv_sub_f32 %1, %1, %3\n\tv_mul_f32 %0, %3, %1"
                         : "+v"(a), "+v"(b), "+v"(c) : "v"(s));)
    } else if (MODE == 7) {  // the real march core: every link as in gs_core (a = res of the previous step)
      REP16(asm volatile("v_mul_f32 %1, %4, %1\n\tv_mul_f32 %2, %4, %3\n\t"
                         "v_mul_f32_dpp %1, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mul_f32 %3, %4, %0\n\tv_add_f32 %1, %1, %2\n\tv_mul_f32 %2, %4, %4\n\t"
                         "v_add_f32 %1, %1, %3\n\tv_add_f32 %1, %1, %2\n\tv_sub_f32 %1, %1, %4\n\tv_mul_f32 %0, %4, %1"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));)
    } else if (MODE == 8) {  // lag-2 variant: the DPP reads the result of TWO steps ago (b2), own-lane chain only
      REP16(asm volatile("v_mul_f32 %2, %5, %2\n\tv_mul_f32 %3, %5, %5\n\t"
                         "v_mul_f32_dpp %2, %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32 %1, %0\n\t"
                         "v_mul_f32 %4, %5, %0\n\tv_add_f32 %2, %2, %3\n\tv_mul_f32 %3, %5, %5\n\t"
                         "v_add_f32 %2, %2, %4\n\tv_add_f32 %2, %2, %3\n\tv_sub_f32 %2, %2, %5\n\tv_mul_f32 %0, %5, %2"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e2) : "v"(s));)
    } else if (MODE == 6) {  // dependent chain of packed adds
      REP16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&a) : "v"(*(const double *)&c));)
    }
  }
  const unsigned long long t1 = clk();
  const unsigned long long w1 = wall_clock64();
  out[threadIdx.x] = a + b + c + d + e2;
  if (threadIdx.x == 0) {
    t[0] = t1 - t0;
    t[1] = w1 - w0;  // 100 MHz
  }
}
template <int MODE>
void run(const char *name, int per_iter) {
  float *o; unsigned long long *t, h[2];
  hipMalloc(&o, 256 * 4); hipMalloc(&t, 16);
  const int iters = 2000;
  k<MODE><<<1, 64>>>(o, t, iters); k<MODE><<<1, 64>>>(o, t, iters);
  hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
  printf("%-56s %6.2f clk/instr  (counter runs at %.0f MHz)\n", name, (double)h[0] / ((double)iters * 16 * per_iter), 100.0 * (double)h[0] / (double)h[1]);
}
int main() {
  run<0>("dependent v_add_f32 chain", 1);
  run<1>("2 independent chains", 2);
  run<2>("4 independent chains", 4);
  run<3>("s_nop 1 + dpp mov + mul (per 3 instr)", 3);
  run<4>("s_nop 1 + fused dpp mul (per 2 instr)", 2);
  run<5>("march step, 10 instr", 10);
  run<6>("dependent v_pk_add_f32 chain", 1);
  run<7>("march core as shipped (per step of 10)", 10);
  run<8>("march core, DPP source two steps old (per step of 11)", 11);
  return 0;
}
