// pitch_test.hip -- what would row-exclusive cache lines be worth to the row pipelines?
// 1024 workgroups x 4 waves, each wave walking down a quarter of the rows of its env's two fields the way advdif_v3 /
// project_tail_v2 do: per row and field a 132-float read (33 lanes x 16 bytes at a 4-byte-aligned address) and a
// 128-float write (64 lanes x 8 bytes) plus the two wall columns by the first / last lane.  Three layouts of the same
// 258 x 130 fields:
//   0  pitch 130 floats (520 B: the reference layout -- rows share their first and last cache line with their neighbours)
//   1  pitch 144 floats, column 0 at the start of a 64-byte line (rows own 9 lines, the interior starts 4 bytes in)
//   2  pitch 144 floats, column 1 at the start of a 64-byte line (the interior owns 8 whole lines, the ghosts sit in
//      the lines before / after)
// Prints the time per pass and the useful bytes per second; run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for the
// traffic.   hipcc -O3 --offload-arch=gfx950 pitch_test.hip -o pitch_test && ./pitch_test
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

struct F4 {
  float v[4];
};

template <int MODE>
__global__ __launch_bounds__(256, 4) void k_rows(const float *__restrict__ sx, const float *__restrict__ sy,
                                                 float *__restrict__ dx, float *__restrict__ dy, size_t env_stride) {
  constexpr int n = 258, m = 130;
  constexpr int pitch = (MODE == 0) ? 130 : 144, c0 = (MODE == 2) ? 15 : 0;  // floats from the row's start to column 0
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t eo = (size_t)blockIdx.x * env_stride;
  const float *ax = sx + eo, *ay = sy + eo;
  float *bx = dx + eo, *by = dy + eo;
  const int i0 = 1 + wave * 64, i1 = i0 + 63;
  for (int i = i0; i <= i1; ++i) {
    const size_t ro = (size_t)i * pitch + c0;
    float ox[2] = {0.f, 0.f}, oy[2] = {0.f, 0.f};
    if (lane < 33) {  // the staged row: 132 floats from column 0
      F4 a, b;
      __builtin_memcpy(&a, ax + ro + 4 * lane, 16);
      __builtin_memcpy(&b, ay + ro + 4 * lane, 16);
      ox[0] = a.v[0] + a.v[3];
      ox[1] = a.v[1] + a.v[2];
      oy[0] = b.v[0] + b.v[3];
      oy[1] = b.v[1] + b.v[2];
    }
    // every lane ends up with something to store (values do not matter)
    ox[0] += __shfl(ox[0], lane >> 1), ox[1] += __shfl(ox[1], lane >> 1);
    oy[0] += __shfl(oy[0], lane >> 1), oy[1] += __shfl(oy[1], lane >> 1);
    __builtin_memcpy(bx + ro + 1 + 2 * lane, ox, 8);
    __builtin_memcpy(by + ro + 1 + 2 * lane, oy, 8);
    if (lane == 0 || lane == 63) {
      const int j = lane == 0 ? 0 : m - 1;
      bx[ro + j] = ox[0];
      by[ro + j] = oy[0];
    }
  }
  (void)n;
}

template <int MODE>
float run(const float *sx, const float *sy, float *dx, float *dy, size_t env_stride) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 12; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rows<MODE>, dim3(1024), dim3(256), 0, nullptr, sx, sy, dx, dy, env_stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 1 && ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t env_stride = ((size_t)258 * 144 + 64 + 63) / 64 * 64;
  const size_t total = 1024 * env_stride + 256;
  float *buf[4];
  for (auto &b : buf) {
    hipMalloc((void **)&b, total * 4);
    hipMemset(b, 0, total * 4);
  }
  const double useful = 1024.0 * 256 * (2 * 132 + 2 * 130) * 4;  // bytes read + written per pass
  const float t0 = run<0>(buf[0], buf[1], buf[2], buf[3], env_stride);
  const float t1 = run<1>(buf[0], buf[1], buf[2], buf[3], env_stride);
  const float t2 = run<2>(buf[0], buf[1], buf[2], buf[3], env_stride);
  printf("pitch 130 (reference layout)          %.3f ms  %.0f GB/s useful\n", t0, useful / t0 / 1e6);
  printf("pitch 144, column 0 on a line start   %.3f ms  %.0f GB/s useful\n", t1, useful / t1 / 1e6);
  printf("pitch 144, column 1 on a line start   %.3f ms  %.0f GB/s useful\n", t2, useful / t2 / 1e6);
  return 0;
}
