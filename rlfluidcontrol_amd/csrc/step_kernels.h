// step_kernels.h -- the step kernels and their launchers for ONE instantiation of the optimised phases.  Included by
// step_prod.hip / step_diag.hip / step_perf.hip, which define AFC_MODE (0 / 1 / 2) and AFC_MODE_NS (prod / diag / perf)
// first: three translation units that build in parallel, each with its own device code object.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "bdim_device.h"
#define AFC_ONLY_MODE AFC_MODE
#include "bdim_v2.h"
#include "step_launch.h"

#ifdef AFC_NO_HARD_HINT  // experiment: every solve with the lazy residual (tools/ab_inproc.py)
#define AFC_HARD_HINT(x) false
#else
#define AFC_HARD_HINT(x) (x)
#endif

namespace {

using afc::EnvScalars;
using afc::G;
using afc::kThreads;
using afc::StepParams;

// ---------------- kernels ----------------
// CPL = columns per lane of the optimised kernels (grid height m-2 <= 64*CPL); CPL = 0 selects the
// straightforward reference kernels.  One instantiation per CPL keeps each kernel's register
// budget at 4 workgroups (16 waves) per CU.
// MODE names the instantiation of the optimised phases this translation unit carries (AFC_MODE / AFC_MODE_NS, see
// bdim_v2.h): 0 = shipped, 1 = diagnostics compiled in, 2 = perf mode (red-black smoother; cfg.gs_mode = 1).
template <int CPL, int MODE>
__device__ __forceinline__ void step_dispatch(const StepParams &P, size_t e, float a, float b, float *pshift,
                                              double *sm, afc::ShBcast *sh, float *lds, bool first_after_action = false,
                                              bool heun_in = false, bool defer_out = false) {
  if constexpr (CPL == 0)
    afc::sim_step(P, e, a, b, sm, sh);
  else
    afc::AFC_MODE_NS::sim_step_v2<CPL>(P, e, a, b, pshift, sm, sh, lds, first_after_action, heun_in, defer_out);
}

template <int CPL, int MODE>
__global__ __launch_bounds__(afc::kThreads, 4) void k_sim_step(const StepParams *__restrict__ Pp,
                                                               const float *__restrict__ xi_m,
                                                               float *__restrict__ force) {
  constexpr bool DIAG = (MODE == 1);
  const StepParams &P = *Pp;  // parameters live in device memory: phase functions take them by reference
  __shared__ double sm[afc::kThreads];
  __shared__ afc::ShBcast sh;
  extern __shared__ float lds_dyn[];
  const size_t e = blockIdx.x;
  const float a = xi_m[2 * e], b = xi_m[2 * e + 1];
  float pshift = G(P.sc)[e].p_shift;
  if (DIAG) afc::gate_init(P);
  __syncthreads();
  if (DIAG && CPL != 0 && !((P.dbg >> 12) & 15)) afc::gate_acquire(P);
  step_dispatch<CPL, MODE>(P, e, a, b, &pshift, sm, &sh, lds_dyn);
  if (DIAG && CPL != 0 && !((P.dbg >> 12) & 15)) afc::gate_release(P, true);
  if (threadIdx.x == 0) {
    EnvScalars &s = G(P.sc)[e];
    s.p_shift = pshift;
    s.xi1_m = a;
    s.xi2_m = b;
    s.t += P.dt / P.resolution;  // AFCCylinder.pde:56
    s.force_x = sh.fx;
    s.force_y = sh.fy;
    s.iters[0] = sh.it0;
    s.iters[1] = sh.it1;
    s.resid[0] = sh.r0;
    s.resid[1] = sh.r1;
    s.sim_steps += 1;
    s.it_hist[afc::clampi(sh.it0, 1, 8) - 1] += 1;  // clamped: diagnostic launches that leave the step early
    s.it_hist[afc::clampi(sh.it1, 1, 8) - 1] += 1;  // do not fill sh
    if (!(isfinite(sh.fx) && isfinite(sh.fy))) s.diverged = 1;
    if (force) {
      force[2 * e] = sh.fx;
      force[2 * e + 1] = sh.fy;
    }
  }
}

// The body of draw() (clientCFD.pde:35-56) repeated until the next action request.
template <int CPL, int MODE>
__global__ __launch_bounds__(afc::kThreads, 4) void k_agent_step(const StepParams *__restrict__ Pp,
                                                                 const float *__restrict__ action,
                                                                 float *__restrict__ obs, float *__restrict__ reward,
                                                                 uint8_t *__restrict__ done, int *__restrict__ nsteps,
                                                                 int max_steps) {
  constexpr bool DIAG = (MODE == 1);
  const StepParams &P = *Pp;
  __shared__ double sm[afc::kThreads];
  __shared__ afc::ShBcast sh;
  extern __shared__ float lds_dyn[];
  const size_t e = blockIdx.x;
  // placement experiment (gate value 14): the workgroups of every second group of 128 CU slots leave at once, so that
  // half the CUs carry four envs each and the other half idles -- against the same number of envs two per CU on all
  // CUs this separates contention inside a CU from contention for the chip's memory system
  if (DIAG && P.gate_limit == 14 && (blockIdx.x & 255) >= 128) return;
  if (DIAG && P.stagger_ticks > 0) {  // workgroups b, b+256, b+512, ... tend to share a CU: offset their phases
    const long long t_end = wall_clock64() + (long long)P.stagger_ticks * (long long)((blockIdx.x >> (P.stagger_shift & 15)) % (unsigned)((P.stagger_shift >> 4) + 2));
    while (wall_clock64() < t_end) __builtin_amdgcn_s_sleep(64);
  }
  // the env's episode scalars live in LDS for the whole launch (thread 0 does the bookkeeping of
  // clientCFD.pde:35-56, everybody reads what the control flow needs): no per-thread register copies
  __shared__ EnvScalars s;
  __shared__ int s_stop;
  const float a1 = action[2 * e], a2 = action[2 * e + 1];
  if (threadIdx.x == 0) {
    s = G(P.sc)[e];
    s.xi1 = a1;  // clientCFD.pde:51-54
    s.xi2 = a2;
    s.xi1_m = P.act_scale * s.xi1;
    s.xi2_m = P.act_scale * s.xi2;
    s.window_done = 0;
    s_stop = 0;
  }
  if (DIAG) afc::gate_init(P);
  __syncthreads();
  if (DIAG && CPL != 0) afc::gate_acquire(P);
  const float xi1_m = afc::AFC_MODE_NS::uni(s.xi1_m), xi2_m = afc::AFC_MODE_NS::uni(s.xi2_m);
  float pshift = s.p_shift;
  int steps = 0;
  long long dbg_t0 = 0;
  // A sim-step that is followed by another one in this launch may leave the tail of its corrector's projection and the
  // Heun average to the next step's advection pass (sim_step_v2: heun_in / defer_out).  Whether another one follows is
  // known before the step: the loop below ends on the step count, on the time after the step, or on the action window
  // completing with it (clientCFD.pde:36-47) -- the same expressions, evaluated in advance.  Should the loop end with
  // a tail pending all the same, the tail runs on its own after it.
  bool heun_pending = false;
  bool can_defer = false;
  if constexpr (CPL >= 1 && (CPL <= 2 || AFC_FUSE_T2) && !DIAG) can_defer = afc::AFC_MODE_NS::heun_fusable<CPL>(P);
  while (steps < max_steps) {
    if (!(s.t < P.end_time)) break;  // clientCFD.pde:36
#if AFC_EXP_STAMPS
    if ((P.dbg & 8) && steps == 8) dbg_t0 = wall_clock64();
#else
    if (DIAG && (P.dbg & 8) && steps == 8) dbg_t0 = wall_clock64();
#endif
#if (AFC_MODE == 0 || AFC_MODE == 1) && AFC_PRIO_ROT >= 0  // (afc::prod and afc::diag: AFC_PRIO in bdim_v2.h)
    if constexpr (CPL == 1 || CPL == 2) afc::AFC_MODE_NS::prio_rotate(steps);  // (not for the wide grids: see prio_rotate)
    if constexpr (CPL == 3) afc::AFC_MODE_NS::prio_by_mate();  // 384x192: static priorities by mate
#endif
    bool defer = false;
    if (can_defer) {
      float t_after = s.t;
      t_after += P.dt / P.resolution;
      const bool window_ends = (t_after > P.init_time) && (s.callLearn - 1 <= 0);
      defer = (steps + 1 < max_steps) && (t_after < P.end_time) && !window_ends;
      defer = afc::AFC_MODE_NS::uni((int)defer) != 0;
    }
    step_dispatch<CPL, MODE>(P, e, xi1_m, xi2_m, &pshift, sm, &sh, lds_dyn, AFC_HARD_HINT(steps == 0), heun_pending, defer);  // the action is new in the first step
    heun_pending = defer;
    ++steps;
    if (threadIdx.x == 0) {
      s.t += P.dt / P.resolution;
      s.force_x = sh.fx;
      s.force_y = sh.fy;
      s.iters[0] = sh.it0;
      s.iters[1] = sh.it1;
      s.resid[0] = sh.r0;
      s.resid[1] = sh.r1;
      s.sim_steps += 1;
      s.it_hist[afc::clampi(sh.it0, 1, 8) - 1] += 1;
      s.it_hist[afc::clampi(sh.it1, 1, 8) - 1] += 1;
      if (!(isfinite(sh.fx) && isfinite(sh.fy))) s.diverged = 1;
      if (s.t > P.init_time) {  // clientCFD.pde:39-47
        s.callLearn--;
        s.Cd += sh.fx;
        s.Cl += sh.fy;
        if (s.callLearn <= 0) {
          s.callLearn = P.substeps;
          s.Cd = s.Cd / s.callLearn * 2 / P.resolution;
          s.Cl = s.Cl / s.callLearn * 2 / P.resolution;
          s.obs_cl = s.Cl;
          s.obs_cd = s.Cd;
          if (!P.carry_quirk) {
            s.Cd = 0.f;
            s.Cl = 0.f;
          }
          s.window_done = 1;
          s_stop = 1;
        }
      }
    }
    __syncthreads();
    if (s_stop) break;
  }
  if constexpr (CPL >= 1 && (CPL <= 2 || AFC_FUSE_T2) && !DIAG) {
    if (heun_pending) afc::AFC_MODE_NS::heun_tail_only(P, e, pshift, lds_dyn);
  }
  if (DIAG && CPL != 0) afc::gate_release(P, true);
  if (threadIdx.x == 0) {
    s.p_shift = pshift;
    G(P.sc)[e] = s;
    obs[2 * e] = s.obs_cl;  // wire order is Cl then Cd (clientCFD.pde:120)
    obs[2 * e + 1] = s.obs_cd;
    const float aa1 = afc::jabs(a1), aa2 = afc::jabs(a2);
    reward[e] = -s.obs_cd - P.reward_penalty * (aa1 * aa1 * aa1 + aa2 * aa2 * aa2);  // server.py:61-65
    done[e] = (!(s.t < P.end_time)) ? 1 : 0;
    if (nsteps) nsteps[e] = steps;
#if AFC_EXP_STAMPS
    if (P.dbg & 8) {
#else
    if (DIAG && (P.dbg & 8)) {  // diagnostic: wall-clock ticks (100 MHz, low 24 bits) at the start of step 8 and at the end
#endif
      obs[2 * e] = (float)(dbg_t0 & 0xFFFFFF);
      obs[2 * e + 1] = (float)(wall_clock64() & 0xFFFFFF);
    }
  }
}

// unit-level operators on env 0 (same device functions as the step, chosen the way sim_step_v2 chooses them)
#if AFC_MODE != 3  // (afc::lat carries the step kernels only)
#if AFC_MODE != 2
template <int MODE>
__global__ __launch_bounds__(afc::kThreads, 4) void k_op_advdif(StepParams P, int same) {
  extern __shared__ float lds_dyn[];
  if (P.impl == 0) {
    afc::phase_advdif(P, 0, true);  // F = (adv+nu*dif)(u)*dt + u0 ; result in fx,fy
    return;
  }
  for (int k = threadIdx.x; k < P.N; k += afc::kThreads) {  // F starts as a copy of u (VectorField.pde:171)
    G(P.fx)[k] = G(P.ux)[k];
    G(P.fy)[k] = G(P.uy)[k];
  }
  __syncthreads();
  const int m_ = P.m - 2;
  // same != 0: the base field IS the advected field (BDIM.update, BDIM.pde:82-84: u0 = u) -- the predictor's form,
  // which takes its base rows from the staged source rows
  const float *bx = same ? G(P.ux) : G(P.u0x), *by = same ? G(P.uy) : G(P.u0y);
  const bool v3 = P.adv_ns == afc::AFC_MODE_NS::kAdvNS && !(MODE == 1 && (P.dbg & 1024));
#define ADV(c)                                                                                    \
  {                                                                                               \
    if (v3)                                                                                       \
      afc::AFC_MODE_NS::advdif_v3<(c > 2 ? 2 : c), 10 + c, (c == 3 || c == 4)>(P, G(P.ux), G(P.uy), bx, by, G(P.fx), G(P.fy), lds_dyn, false); \
    else                                                                                          \
      afc::AFC_MODE_NS::advdif_v2<c, 10 + c>(P, G(P.ux), G(P.uy), bx, by, G(P.fx), G(P.fy), lds_dyn);    \
  }
  if (m_ <= 64) ADV(1)
  else if (m_ <= 128) ADV(2)
  else if (m_ <= 192) ADV(3)
  else ADV(4)
#undef ADV
}
#endif
// VectorField.project (VectorField.pde:130-143) on env 0: u in (fx, fy), p in P.p; the solution is left in P.d with its
// pending mean shift in *shift_out (project_v2: fused head, coarse legs, up-legs, projection tail -- what a half-step runs)
template <int MODE>
__global__ __launch_bounds__(afc::kThreads, 4) void k_op_project(StepParams P, int *iters, float *linf, float *shift_out) {
  __shared__ double sm[afc::kThreads];
  extern __shared__ float lds_dyn[];
  float shift = 0.f, l = 0.f;
  afc::AFC_MODE_NS::ProfT pf;
  pf.start(P);
  const int m_ = P.m - 2;
  int it;
  // TAG 1: instances of their own (see advdif_v2's TAG); the column count selects what the step kernel of this grid runs
  // (C only enters project_v2 / mg_solve_v2 through up_strip_h(C), the strip height of the finest up-leg, which is 16 for
  // C = 1 and C = 2 alike: grids of up to 64 columns run the same code under either)
#define PROJ(c)                                                                                                       \
  it = afc::AFC_MODE_NS::project_v2<c, 1>(false, P, 0, G(P.fx), G(P.fy), G(P.p), G(P.d), &shift, nullptr, nullptr, \
                                           nullptr, nullptr, sm, &l, pf, lds_dyn)
  if (m_ <= 128) PROJ(2);
  else if (m_ <= 192) PROJ(3);
  else PROJ(4);
#undef PROJ
  if (threadIdx.x == 0) {
    *iters = it;
    *linf = l;
    *shift_out = shift;
  }
}
template <int MODE>
__global__ __launch_bounds__(afc::kThreads, 4) void k_op_mg(StepParams P, int *iters, float *linf) {
  __shared__ double sm[afc::kThreads];
  // b is staged in fx; x in p.  r = b - A x  (MG.pde:51)
  const afc::Lv L0 = afc::make_level(P, 0, 0);
  const int n = P.n, m = P.m;
  for (int k = threadIdx.x; k < P.N; k += afc::kThreads) {
    const int i = k / m, j = k - i * m;
    float rv = G(P.fx)[k];  // b.minus(A.times(x)): ghosts of A x are 0 -> r ghost = b ghost
    if (i >= 1 && i <= n - 2 && j >= 1 && j <= m - 2) {
      const float lw = L0.lx[k], le = L0.lx[k + m], ls = L0.ly[k], ln = L0.ly[k + 1];
      const float diag = -(lw + le + ls + ln);
      const float *p = G(P.p);
      rv = G(P.fx)[k] - (p[k] * diag + p[k - m] * lw + p[k + m] * le + p[k - 1] * ls + p[k + 1] * ln);
    }
    G(P.r)[k] = rv;
  }
  __syncthreads();
  float l;
  extern __shared__ float lds_dyn[];
  int it;
  if (P.impl == 0) {
    afc::Prof pf;
    pf.start(P);
    it = afc::mg_solve(P, 0, sm, &l, pf);
  } else {
    afc::Lv L0v = afc::make_level(P, 0, 0);
    L0v.d = G(P.u0y);
    float xs;
    afc::AFC_MODE_NS::ProfT pf;
    pf.start(P);
    const int m_ = P.m - 2;
    if (m_ <= 128)
      it = afc::AFC_MODE_NS::mg_solve_v2<16>(P, 0, L0v, nullptr, nullptr, nullptr, 0.f, sm, &l, &xs, pf, lds_dyn, false);
    else if (m_ <= 192)
      it = afc::AFC_MODE_NS::mg_solve_v2<afc::AFC_MODE_NS::up_strip_h(3)>(P, 0, L0v, nullptr, nullptr, nullptr, 0.f, sm, &l, &xs, pf, lds_dyn, false);
    else
      it = afc::AFC_MODE_NS::mg_solve_v2<afc::AFC_MODE_NS::up_strip_h(4)>(P, 0, L0v, nullptr, nullptr, nullptr, 0.f, sm, &l, &xs, pf, lds_dyn, false);
  }
  if (threadIdx.x == 0) {
    *iters = it;
    *linf = l;
  }
}
#endif  // AFC_MODE != 3


}  // namespace

namespace afc {

#define AFC_CAT2(a, b) a##b
#define AFC_CAT(a, b) AFC_CAT2(a, b)
#define AFC_FN(name) AFC_CAT(name, AFC_MODE_NS)

// CPL = 0 (the straightforward reference kernels) exists in the diagnostic translation unit only
void AFC_FN(launch_sim_step_)(int cpl, int E, size_t lds, hipStream_t st, const StepParams *dP, const float *dxi,
                              float *dforce) {
  switch (cpl) {
#define LAUNCH(c) case c: hipLaunchKernelGGL((k_sim_step<c, AFC_MODE>), dim3(E), dim3(kThreads), lds, st, dP, dxi, dforce); break
#if AFC_MODE == 1
    LAUNCH(0);
#endif
    LAUNCH(1); LAUNCH(2);
#if AFC_MODE != 3  // (afc::lat: grids of up to 128 columns only -- the kernels whose levels mg_up_tr serves)
    LAUNCH(3); LAUNCH(4);
#endif
#undef LAUNCH
  }
}
void AFC_FN(launch_agent_step_)(int cpl, int E, size_t lds, hipStream_t st, const StepParams *dP, const float *dact,
                                float *dobs, float *drew, uint8_t *ddone, int *dns, int max_steps) {
  switch (cpl) {
#define LAUNCH(c) case c: hipLaunchKernelGGL((k_agent_step<c, AFC_MODE>), dim3(E), dim3(kThreads), lds, st, dP, dact, dobs, drew, ddone, dns, max_steps); break
#if AFC_MODE == 1
    LAUNCH(0);
#endif
    LAUNCH(1); LAUNCH(2);
#if AFC_MODE != 3
    LAUNCH(3); LAUNCH(4);
#endif
#undef LAUNCH
  }
}
#if AFC_MODE != 3
#if AFC_MODE != 2  // (the advection pass has no smoother in it: a perf-mode env runs the shipped operator kernel)
void AFC_FN(launch_op_advdif_)(size_t lds, hipStream_t st, const StepParams &P, int same) {
  hipLaunchKernelGGL(k_op_advdif<AFC_MODE>, dim3(1), dim3(kThreads), lds, st, P, same);
}
#endif
void AFC_FN(launch_op_project_)(size_t lds, hipStream_t st, const StepParams &P, int *dit, float *dl, float *dshift) {
  hipLaunchKernelGGL(k_op_project<AFC_MODE>, dim3(1), dim3(kThreads), lds, st, P, dit, dl, dshift);
}
void AFC_FN(launch_op_mg_)(size_t lds, hipStream_t st, const StepParams &P, int *dit, float *dl) {
  hipLaunchKernelGGL(k_op_mg<AFC_MODE>, dim3(1), dim3(kThreads), lds, st, P, dit, dl);
}
#endif  // AFC_MODE != 3
#if AFC_MODE == 1
int occupancy_agent_step_diag(int cpl, size_t lds, int *per_cu) {
  switch (cpl) {
#define OCC(c) case c: return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_agent_step<c, 1>, kThreads, lds) == hipSuccess ? 0 : -1
    OCC(1); OCC(2); OCC(3); OCC(4);
#undef OCC
  }
  return -1;
}
#endif

}  // namespace afc
