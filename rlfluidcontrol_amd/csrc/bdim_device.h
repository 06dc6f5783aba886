// bdim_device.h -- device side of the batched BDIM step (gfx950).
//
// Execution model: ONE WORKGROUP PER ENV, persistent over a whole agent step.  Envs are
// independent simulations, so no inter-workgroup communication exists anywhere: every
// reduction (outflow mean, mean pressure, r.r) is workgroup-local and every data-dependent
// loop (multigrid iterations, episode bookkeeping) is a per-workgroup loop.  With E >= 4 x 256
// workgroups the latency-bound phases of one env (ordered Gauss-Seidel) overlap the streaming
// phases of the other envs resident on the same CU.
//
// Arithmetic: float32, -ffp-contract=off, the reference's evaluation order (file:line cited at
// each function).  The only deviations from the reference's bits are two order-sensitive
// reductions (Field.sum for the mean pressure, Field.inner for the stopping test), which are tree
// reductions by default and sequential (bit-identical step) when cfg.exact_sums is set.
#pragma once
// experiment builds (tools/build_variant.sh -DAFC_EXP_STAMPS=1): dbg = 8 (start-of-step-8 / end stamps in obs, tools/desync_probe.py)
// is honoured by the SHIPPED instantiation instead of switching to the diagnostic one
#ifndef AFC_EXP_STAMPS
#define AFC_EXP_STAMPS 0
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afc {

constexpr int kThreads = 256;    // workgroup size (4 waves)
constexpr int kMaxLevels = 12;

#define AFC_AS1 __attribute__((address_space(1)))
// Global-memory pointers inside StepParams carry their address space: the struct itself is read
// from device memory, so without it every access would be a flat_* instruction.  G() converts to an
// ordinary pointer; the compiler still knows where it came from and emits global_* loads/stores.
template <class T>
__host__ __device__ __forceinline__ T *G(AFC_AS1 T *p) {
  return (T *)p;
}
template <class T>
__host__ __device__ __forceinline__ const T *G(const AFC_AS1 T *p) {
  return (const T *)p;
}

struct LevelDesc {
  int n, m;          // array size incl. ghosts
  int coef_off;      // offset (floats) of this level in mg_lx/mg_ly/mg_inv
  int scratch_off;   // offset (floats) of this level in the per-env coarse scratch (levels >= 1)
  // analytic description of the coefficients away from the bodies (verified on the host against
  // the real arrays): every face coefficient equals `cu`, except the four boundary face lines
  // which are 0 when zero_bnd (coarse levels, MG.pde:120); inv = invtab[3*rowclass+colclass]
  // (class 0 interior, 1 first, 2 last).  Rows [sp_i0, sp_i1] contain cells that deviate.
  float cu;
  int zero_bnd;
  float invtab[9];
  int sp_i0, sp_i1;
  int sp_j0, sp_j1;  // ... and columns [sp_j0, sp_j1] (all rows taken together)
  int bot_off;       // mg_bottom: byte offset of this level's planes from the first bottom level's (levels >= bottom_lb)
  unsigned bot_rcp;  // mg_bottom: ceil(2^16 / m), the multiplier of its index -> (row, column) split
  int dbg_copy;      // StepParams.dbg, reachable from an Lv (diagnostic switches)
  int mw_H;          // strip height of the multi-wave smoother on this level (rows per sweep per strip, <= 63)
  int rbw;           // perf mode: 1 = the level's up-leg runs as register-resident row pipelines (mg_up_rbw)
  int tm;            // the up-leg with the sweeps marching down the rows (mg_up_tr) fits: bit 0 the shipped kernels' LDS region, bit 1 the diagnostic ones'
};

struct SampleDev {
  int base;  // i*m + j
  float s, t;
  int exact;
};

struct EnvScalars {  // per-env episode state (clientCFD.pde globals + AFCCylinder fields)
  float t;           // AFCCylinder.t
  float xi1, xi2;    // test.xi1, test.xi2
  float xi1_m, xi2_m;
  int callLearn;     // clientCFD.pde:12
  float Cd, Cl;      // clientCFD.pde:13 running sums
  float force_x, force_y;
  float obs_cl, obs_cd;  // last observation sent
  int iters[2];
  float resid[2];
  int diverged;
  int sim_steps;     // total since reset
  float p_shift;     // deferred mean shift of the last projection (bdim_v2.h)
  int window_done;   // 1: the last agent step ended because an averaging window completed (obs is fresh)
  int it_hist[8];    // pressure solves since the last clear that took 1,2,...,7,>=8 MG iterations
};

struct StepParams {
  int n, m, N;       // field dims, N = n*m
  int Ns;            // per-env stride of a fine field (floats)
  int Cs;            // per-env stride of the coarse scratch (floats), 4 arrays of Cs/4
  int nlev;
  LevelDesc lev[kMaxLevels];
  float dt, nu, D, dR, mg_tol;
  int mg_max_iters, resolution;
  // env-shared static fields (L2/MALL resident)
  const AFC_AS1 float *delx, *dely, *del1x, *del1y, *wnxx, *wnxy, *wnyx, *wnyy, *cx, *cy;
  const AFC_AS1 float *w1x, *w2x, *ry1x, *ry2x, *w1y, *w2y, *rx1y, *rx2y;
  const AFC_AS1 float *mg_lx, *mg_ly, *mg_inv;
  const AFC_AS1 SampleDev *fs;
  const AFC_AS1 float *seg_l, *seg_nx, *seg_ny;
  int nseg;
  // per-env state [E][Ns]
  AFC_AS1 float *ux, *uy, *p, *u0x, *u0y, *usx, *usy, *fx, *fy, *r, *d, *qx, *qy;
  AFC_AS1 float *coarse;  // [E][Cs]: r_l | x_l | d_l for l >= 1
  AFC_AS1 EnvScalars *sc;
  AFC_AS1 long long *stamps;  // optional [32] accumulated wall-clock ticks per phase (env 0 only; diagnostics)
  AFC_AS1 int *gate;          // [256] streaming-phase tokens per group of workgroups that share a CU (experiment)
  int gate_limit;             // 0 = off; 1..14 = per-CU token count (experiment); 15 = two populations of CUs take turns
  int gate_n[2];              // workgroups per population (gate_limit 15)
  // episode constants
  float act_scale, init_time, end_time, reward_penalty;
  int substeps, carry_quirk, exact_sums;
  // v2 path
  int dbg;             // experiment switches (bit flags), 0 in production
  int stagger_shift;
  int stagger_ticks;   // start delay (100 MHz ticks) per workgroup phase: de-synchronises the envs of a CU
  int impl;            // 0 = v1 reference kernels, 2 = optimised kernels
  int adv_rows;        // rows per LDS tile block in advdif_v2
  int adv_ns;          // ring depth (rows) of advdif_v3's per-wave row pipelines; 0 = use advdif_v2
  int head_v3;         // 1: the barrier-free head of the pressure solve (grids of 128 interior columns)
  int gs_mode;         // 0: MG.smooth in the reference's lexicographic order; 1: red-black perf mode
  int fuse_tail;       // bit 0: the predictor's projection tail runs inside the corrector's advection (advdif_v3f), its result in qx / qy; bit 1: the corrector's tail + Heun average inside the next sim-step's predictor advection
  int bottom_lb;       // first multigrid level of the LDS-resident bottom of the V-cycle (mg_bottom); 0 = none
  int head_v4;         // 1: the barrier-free head on grids of 132 .. 256 interior columns (head_v4)
  int down_v3;         // 1: coarse down-legs as barrier-free per-wave row pipelines (down_v3c) where the level's shape allows
  int bc_fold;         // 1: the row-streaming passes store the wall columns of u themselves, setBC is row work only (setbc_rows)
  int upleg;           // afc_config.upleg: 0 = mg_up_tr where it applies, 1 = round 5's strip pipeline (mg_up_pipe), 2 = strip by strip with barriers (mg_up_smooth); 1 and 2 run the diagnostic instantiation
  const AFC_AS1 float *bottom_rec;  // its table: per level the planes {lw, le, ls, ln} / {inv, 0} / 0 / 0 per cell, levels bottom_lb .. nlev-1 back to back
  const AFC_AS1 int *band_x, *band_y;  // interior cells touched by the immersed-boundary forcing
  int nband_x, nband_y;
  // compact band tables (forcing_cell_tab): 24 arrays of band_stride floats per component, slot q = band cell q:
  // del, del1, wnx, wny, then for the cell and its neighbours +m, -m, +1, -1: w1, w2, lever arm 1, lever arm 2
  const AFC_AS1 float *band_tab_x, *band_tab_y;
  int band_stride;       // 0: no tables (the gather form)
  int nflag_x, nflag_y;  // band cells within reach of a control cylinder come first in band_x / band_y
};

// ---- PApplet float helpers (same semantics as Java's) ----
__device__ __forceinline__ float jmin(float a, float b) { return (a < b) ? a : b; }
__device__ __forceinline__ float jmax(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ float jabs(float a) { return (a < 0) ? -a : a; }
__device__ __forceinline__ float med3(float a, float b, float c) {  // VectorField.pde:221-223
  return jmax(jmin(a, b), jmin(jmax(a, b), c));
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- optional phase profiling (env 0, thread 0; ticks of wall_clock64 = 100 MHz) ----
struct Prof {
  long long *acc;  // same in every lane (null unless this is workgroup 0 of a profiled launch): SGPRs
  int stop = 0;    // diagnostic: leave the step after phase `stop` (per-phase HBM traffic by differencing)
  __device__ __forceinline__ long long &last_stamp() {
    __shared__ long long s_last;  // time of the previous mark (thread 0 only)
    return s_last;
  }
  __device__ void start(const StepParams &P) {
    long long *a = (blockIdx.x == 0) ? G(P.stamps) : nullptr;
    const unsigned long long v = reinterpret_cast<unsigned long long>(a);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    acc = reinterpret_cast<long long *>(((unsigned long long)hi << 32) | lo);
    if (acc && threadIdx.x == 0) last_stamp() = wall_clock64();
  }
  __device__ __forceinline__ void mark(int id) {
    if (acc && threadIdx.x == 0) {
      const long long now = wall_clock64();
      acc[id] += now - last_stamp();
      last_stamp() = now;
    }
  }
};
enum { PF_ADVDIF = 0, PF_FORCING, PF_SETBC, PF_RESID, PF_DOWN0, PF_COARSE_DOWN, PF_COARSE_GS, PF_COARSE_UP, PF_UP0,
       PF_GS0, PF_INC0, PF_RR, PF_PROJECT, PF_HEUN, PF_COUNT };

// ---- streaming-phase gate: workgroups b, b+256, b+512, ... share a CU (tools/hwid_test.hip).
// gate_limit 1..14 (experiment): at most gate_limit of the workgroups of a CU may be in the HBM-heavy part of the step
// (projection tail .. solve head) at a time.
// gate_limit 15 = two POPULATIONS of CUs taking turns: population 0 (CUs with bit 3 of the block index clear) streams
// while population 1 is in its multigrid part, and vice versa -- the streaming half of the chip then has the HBM to
// itself and no marching wave has a streaming neighbour on its CU.  cnt[p] counts the streaming parts population p has
// completed (all of its workgroups together); the k-th streaming part of a workgroup of population 0 starts when
// population 1 has completed k of them, the k-th of population 1 when population 0 has completed k+1.  A workgroup that
// leaves the kernel adds 2^20 (more than any threshold of a launch, and 512 of them still fit an int), so nobody ever
// waits for a workgroup that is gone; the host clears the
// counters before every launch and switches the mode on only when every workgroup of the launch is resident at once.
// Speed only, never correctness.
__device__ __forceinline__ int &gate_round() {
  __shared__ int k;
  return k;
}
__device__ inline void gate_init(const StepParams &P) {
  if (P.gate_limit == 15 && threadIdx.x == 0) gate_round() = 0;
}
__device__ inline void gate_acquire(const StepParams &P) {
  if (P.gate_limit <= 0) return;
  if (threadIdx.x == 0) {
    if (P.gate_limit == 15) {
      const int pop = (blockIdx.x >> 3) & 1, k = gate_round();
      gate_round() = k + 1;
      const int need = pop ? P.gate_n[0] * (k + 1) : P.gate_n[1] * k;
      int *other = G(P.gate) + 32 * (1 - pop);
      // bounded: a turn that does not come within ~50 ms is taken anyway (speed only, never a hang)
      for (int spin = 0; spin < (1 << 15) && __hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need; ++spin)
        __builtin_amdgcn_s_sleep(32);
    } else {
      int *g = G(P.gate) + (blockIdx.x & 255);
      while (true) {
        const int v = __hip_atomic_fetch_add(g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < P.gate_limit) break;
        __hip_atomic_fetch_add(g, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_sleep(64);
      }
    }
  }
  __syncthreads();
}
__device__ inline void gate_release(const StepParams &P, bool leaving = false) {
  if (P.gate_limit <= 0) return;
  if (threadIdx.x == 0) {
    if (P.gate_limit == 15)
      __hip_atomic_fetch_add(G(P.gate) + 32 * ((blockIdx.x >> 3) & 1), leaving ? (1 << 20) : 1, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    else
      __hip_atomic_fetch_add(G(P.gate) + (blockIdx.x & 255), -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- workgroup reductions (fixed tree order => deterministic) ----
__device__ inline double block_sum_d(double v, double *sm) {
  const int t = threadIdx.x;
  sm[t] = v;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (t < s) sm[t] += sm[t + s];
    __syncthreads();
  }
  double r = sm[0];
  __syncthreads();
  return r;
}
__device__ inline float block_max_f(float v, double *smd) {
  float *sm = reinterpret_cast<float *>(smd);
  const int t = threadIdx.x;
  sm[t] = v;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (t < s) sm[t] = jmax(sm[t], sm[t + s]);
    __syncthreads();
  }
  float r = sm[0];
  __syncthreads();
  return r;
}

// ---- QUICK advection + diffusion (VectorField.pde:170-223) ----
// b: the advected component, (bx,by) = velocity pair, all in reference layout with pitch m
__device__ __forceinline__ float bho(const float *__restrict__ b, int n, int m, int i, int j, int d1, int d2,
                                     float uf) {  // VectorField.pde:202-219
  float bf = 0.5f * (b[(i + d1) * m + (j + d2)] + b[i * m + j]);
  if (d1 * uf < 0) {
    i += d1;
    d1 = -d1;
  }
  if (d2 * uf < 0) {
    j += d2;
    d2 = -d2;
  }
  if (i > n - 2 || i < 2 || j > m - 2 || j < 2) return bf;
  const float bc = b[i * m + j];
  const float bd = b[(i + d1) * m + (j + d2)];
  const float bu = b[(i - d1) * m + (j - d2)];
  bf -= (1.f / 6.f) * (bd - 2 * bc + bu);
  const float b1 = bu + 10.f * (bc - bu);
  return med3(bf, bc, med3(bc, bd, b1));
}
__device__ __forceinline__ float advdif_x(const float *__restrict__ x, const float *__restrict__ y, int n, int m,
                                          int i, int j, float nu) {
  const int k = i * m + j;  // VectorField.pde:183-188 (btype 1)
  const float uo = 0.5f * (x[k - m] + x[k]);
  const float ue = 0.5f * (x[k + m] + x[k]);
  const float vs = 0.5f * (y[k] + y[k - m]);
  const float vn = 0.5f * (y[k + 1] + y[k - m + 1]);
  const float adv = ((uo * bho(x, n, m, i, j, -1, 0, uo) - ue * bho(x, n, m, i, j, 1, 0, ue)) +
                     (vs * bho(x, n, m, i, j, 0, -1, vs) - vn * bho(x, n, m, i, j, 0, 1, vn)));
  const float dif = x[k + m] + x[k + 1] - 4 * x[k] + x[k - m] + x[k - 1];  // :198-200
  return adv + nu * dif;
}
__device__ __forceinline__ float advdif_y(const float *__restrict__ x, const float *__restrict__ y, int n, int m,
                                          int i, int j, float nu) {
  const int k = i * m + j;  // VectorField.pde:189-194 (btype 2)
  const float uo = 0.5f * (x[k - 1] + x[k]);
  const float ue = 0.5f * (x[k + m - 1] + x[k + m]);
  const float vs = 0.5f * (y[k - 1] + y[k]);
  const float vn = 0.5f * (y[k] + y[k + 1]);
  const float adv = ((uo * bho(y, n, m, i, j, -1, 0, uo) - ue * bho(y, n, m, i, j, 1, 0, ue)) +
                     (vs * bho(y, n, m, i, j, 0, -1, vs) - vn * bho(y, n, m, i, j, 0, 1, vn)));
  const float dif = y[k + m] + y[k + 1] - 4 * y[k] + y[k - m] + y[k - 1];
  return adv + nu * dif;
}

// Phase A.  predictor (BDIM.update, BDIM.pde:79-87): u0 = u; F = u; F.AdvDif(u0)
//           corrector (BDIM.update2, BDIM.pde:89-97): us = u; F = u; F.AdvDif(u0)
__device__ inline void phase_advdif(const StepParams &P, size_t eo, bool corrector) {
  const int n = P.n, m = P.m;
  const float *ux = G(P.ux) + eo, *uy = G(P.uy) + eo;
  float *savx = (corrector ? G(P.usx) : G(P.u0x)) + eo, *savy = (corrector ? G(P.usy) : G(P.u0y)) + eo;
  const float *u0x = G(P.u0x) + eo, *u0y = G(P.u0y) + eo;
  float *fx = G(P.fx) + eo, *fy = G(P.fy) + eo;
  for (int k = threadIdx.x; k < P.N; k += kThreads) {
    const int i = k / m, j = k - i * m;
    const float cxv = ux[k], cyv = uy[k];
    float ox = cxv, oy = cyv;
    if (i >= 1 && i <= n - 2 && j >= 1 && j <= m - 2) {
      const float bx = corrector ? u0x[k] : cxv, by = corrector ? u0y[k] : cyv;
      ox = advdif_x(ux, uy, n, m, i, j, P.nu) * P.dt + bx;  // VectorField.pde:174-175
      oy = advdif_y(ux, uy, n, m, i, j, P.nu) * P.dt + by;
    }
    savx[k] = cxv;
    savy[k] = cyv;
    fx[k] = ox;
    fy[k] = oy;
  }
  __syncthreads();
}

// body velocity at a u / v point (BDIM.pde:140-149, BodyUnion.pde:84-92, Body.pde:234-240)
__device__ __forceinline__ float ub_x_at(const StepParams &P, int k, float dphi1, float dphi2) {
  float v = 0.f;
  v += 0.f;  // main cylinder: dphi == 0 -> (0 - ry*0)/dt * w0 == 0
  v += ((0.f - G(P.ry1x)[k] * dphi1) / P.dt) * G(P.w1x)[k];
  v += ((0.f - G(P.ry2x)[k] * dphi2) / P.dt) * G(P.w2x)[k];
  return v;
}
__device__ __forceinline__ float ub_y_at(const StepParams &P, int k, float dphi1, float dphi2) {
  float v = 0.f;
  v += 0.f;
  v += ((0.f + G(P.rx1y)[k] * dphi1) / P.dt) * G(P.w1y)[k];
  v += ((0.f + G(P.rx2y)[k] * dphi2) / P.dt) * G(P.w2y)[k];
  return v;
}

// Phase B.  BDIM.updateUP lines 118-120 (BDIM.pde): u = del*F - ub*(del-1) + del1*ddn(F-ub)
__device__ inline void phase_forcing(const StepParams &P, size_t eo, float dphi1, float dphi2) {
  const int n = P.n, m = P.m;
  const float *fx = G(P.fx) + eo, *fy = G(P.fy) + eo;
  float *ux = G(P.ux) + eo, *uy = G(P.uy) + eo;
  for (int k = threadIdx.x; k < P.N; k += kThreads) {
    const int i = k / m, j = k - i * m;
    const bool interior = (i >= 1 && i <= n - 2 && j >= 1 && j <= m - 2);
    {
      const float del = G(P.delx)[k], R = fx[k], ub = ub_x_at(P, k, dphi1, dphi2);
      float un = del * R - ub * (del + (-1.f));
      if (interior) {  // VectorField.normalGrad, VectorField.pde:46-54 (ghosts of g are 0)
        const float dE = fx[k + m] - ub_x_at(P, k + m, dphi1, dphi2);
        const float dW = fx[k - m] - ub_x_at(P, k - m, dphi1, dphi2);
        const float dN = fx[k + 1] - ub_x_at(P, k + 1, dphi1, dphi2);
        const float dS = fx[k - 1] - ub_x_at(P, k - 1, dphi1, dphi2);
        const float g = 0.5f * (G(P.wnxx)[k] * (dE - dW) + G(P.wnyx)[k] * (dN - dS));
        un = un + G(P.del1x)[k] * g;
      }
      ux[k] = un;
    }
    {
      const float del = G(P.dely)[k], R = fy[k], ub = ub_y_at(P, k, dphi1, dphi2);
      float un = del * R - ub * (del + (-1.f));
      if (interior) {
        const float dE = fy[k + m] - ub_y_at(P, k + m, dphi1, dphi2);
        const float dW = fy[k - m] - ub_y_at(P, k - m, dphi1, dphi2);
        const float dN = fy[k + 1] - ub_y_at(P, k + 1, dphi1, dphi2);
        const float dS = fy[k - 1] - ub_y_at(P, k - 1, dphi1, dphi2);
        const float g = 0.5f * (G(P.wnxy)[k] * (dE - dW) + G(P.wnyy)[k] * (dN - dS));
        un = un + G(P.del1y)[k] * g;
      }
      uy[k] = un;
    }
  }
  __syncthreads();
}

// Phase C.  u.setBC() (Field.pde:209-234): u.x is btype 1 with gradientExit (bval 1),
// u.y is btype 2 (bval 0).  Statement order of the reference is preserved: x-loop, y-loop,
// then the outflow-mean correction.  The outflow sum is accumulated by ONE thread in the
// reference's sequential float order (m-2 <= 1022 values staged in LDS): a tree sum differs in
// the last bits, and because the Neumann pressure problem is singular that flux rounding
// shows up as a 1e-4 low-mode drift of p within a few steps.
constexpr int kRowCap = 1024;
__device__ inline void setbc_vel(const StepParams &P, float *ux, float *uy, float *rowbuf) {
  // rowbuf: kRowCap floats of LDS scratch supplied by the caller
  __shared__ float s_mean;
  const int n = P.n, m = P.m;
  for (int j = threadIdx.x; j < m; j += kThreads) {
    ux[j] = ux[m + j];
    const float o = ux[(n - 2) * m + j];
    ux[(n - 1) * m + j] = o;
    ux[m + j] = 1.f;
    if (j < kRowCap) rowbuf[j] = o;
    uy[j] = uy[m + j];
    uy[(n - 1) * m + j] = uy[(n - 2) * m + j];
  }
  __syncthreads();  // also orders the x-loop before the y-loop
  if (threadIdx.x == 0) {
    float s = 0;
    if (m <= kRowCap) {
      for (int j = 1; j < m - 1; ++j) s += rowbuf[j];
    } else {
      for (int j = 1; j < m - 1; ++j) s += ux[(n - 2) * m + j];
    }
    s /= (float)(m - 2);
    s_mean = s;
  }
  for (int i = threadIdx.x; i < n; i += kThreads) {
    float *rx = ux + i * m, *ry = uy + i * m;
    rx[0] = rx[1];
    rx[m - 1] = rx[m - 2];
    ry[0] = ry[1];
    ry[m - 1] = ry[m - 2];
    ry[1] = 0.f;
    ry[m - 1] = 0.f;
  }
  __syncthreads();
  const float corr = 1.f - s_mean;
  for (int j = 1 + threadIdx.x; j < m - 1; j += kThreads) ux[(n - 1) * m + j] += corr;
  __syncthreads();
}
__device__ inline void phase_setbc_u(const StepParams &P, size_t eo, double *sm) {
  (void)sm;
  __shared__ float rowbuf_v1[kRowCap];
  setbc_vel(P, G(P.ux) + eo, G(P.uy) + eo, rowbuf_v1);
}

// ---- multigrid (MG.pde, PoissonMatrix.pde) ----
struct Lv {
  int n, m;
  const float *lx, *ly, *inv;
  float *r, *x, *d;
  const LevelDesc *desc;
};

__device__ __forceinline__ Lv make_level(const StepParams &P, size_t e, int l) {
  Lv L;
  L.n = P.lev[l].n;
  L.m = P.lev[l].m;
  L.lx = G(P.mg_lx) + P.lev[l].coef_off;
  L.ly = G(P.mg_ly) + P.lev[l].coef_off;
  L.inv = G(P.mg_inv) + P.lev[l].coef_off;
  L.desc = &P.lev[l];
  if (l == 0) {
    const size_t eo = e * (size_t)P.Ns;
    L.r = G(P.r) + eo;
    L.x = G(P.p) + eo;
    L.d = G(P.d) + eo;
  } else {
    const int q = P.Cs / 4;
    float *base = G(P.coarse) + e * (size_t)P.Cs + P.lev[l].scratch_off;
    L.r = base;
    L.x = base + q;
    L.d = base + 2 * q;
  }
  return L;
}

// A.times(x) at an interior cell with ghosts of x read through `clamp` (== x after setBC for a
// btype-0 field): PoissonMatrix.pde:53-64
__device__ __forceinline__ float apply_A_clamped(const Lv &L, const float *__restrict__ x, int i, int j) {
  const int m = L.m, k = i * m + j;
  const float lw = L.lx[k], le = L.lx[k + m], ls = L.ly[k], ln = L.ly[k + 1];
  const float diag = -(lw + le + ls + ln);  // PoissonMatrix.pde:45-48
  const int iw = (i - 1 < 1) ? 1 : i - 1, ie = (i + 1 > L.n - 2) ? L.n - 2 : i + 1;
  const int js = (j - 1 < 1) ? 1 : j - 1, jn = (j + 1 > m - 2) ? m - 2 : j + 1;
  return x[k] * diag + x[iw * m + j] * lw + x[ie * m + j] * le + x[i * m + js] * ls + x[i * m + jn] * ln;
}

// d = r.times(A.inv) on the whole array (MG.pde:80)
__device__ inline void mg_dinv(const Lv &L) {
  const int N = L.n * L.m;
  for (int k = threadIdx.x; k < N; k += kThreads) L.d[k] = L.r[k] * L.inv[k];
  __syncthreads();
}

// itmx in-place lexicographic Gauss-Seidel sweeps (MG.pde:81-89), executed in the reference's
// dependency order as pipelined hyperplanes: cell (i,j) of sweep k runs at step (i-1)+(j-1)+2k,
// when d[i-1][j], d[i][j-1] already hold sweep-k values and d[i+1][j], d[i][j+1] still hold
// sweep-(k-1) values -- exactly what the sequential loops see.  Bit-identical to the reference.
__device__ inline void mg_gs(const Lv &L, int itmx) {
  const int n_ = L.n - 2, m_ = L.m - 2, m = L.m;
  const int H = n_ + m_ - 1;
  const int W = (n_ < m_) ? n_ : m_;
  const int steps = H + 2 * (itmx - 1);
  float *d = L.d;
  for (int tau = 0; tau < steps; ++tau) {
    for (int w = threadIdx.x; w < itmx * W; w += kThreads) {
      const int kk = w / W, q = w - kk * W;
      const int h = tau - 2 * kk;
      if (h < 0 || h >= H) continue;
      const int lo = (h - (m_ - 1) > 0) ? h - (m_ - 1) : 0;
      const int hi = (h < n_ - 1) ? h : n_ - 1;
      const int ii = lo + q;
      if (ii > hi) continue;
      const int i = ii + 1, j = h - ii + 1;
      const int k = i * m + j;
      d[k] = -(d[k - m] * L.lx[k] + d[k + m] * L.lx[k + m] + d[k - 1] * L.ly[k] + d[k + 1] * L.ly[k + 1] - L.r[k]) *
             L.inv[k];
    }
    __syncthreads();
  }
}

// d.setBC(); increment(): x += d (whole array), r -= A d (interior)   (MG.pde:90-97)
__device__ inline void mg_increment(const Lv &L) {
  const int n = L.n, m = L.m, N = n * m;
  for (int k = threadIdx.x; k < N; k += kThreads) {
    const int i = k / m, j = k - i * m;
    const int ci = clampi(i, 1, n - 2), cj = clampi(j, 1, m - 2);
    L.x[k] += L.d[ci * m + cj];
    if (ci == i && cj == j) L.r[k] -= apply_A_clamped(L, L.d, i, j);
  }
  __syncthreads();
}

// restrict(r) + setBC, x_coarse = 0  (MG.pde:124-137, 53-59)
__device__ inline void mg_restrict(const Lv &F, const Lv &Cc) {
  const int n = Cc.n, m = Cc.m, N = n * m, mf = F.m;
  for (int k = threadIdx.x; k < N; k += kThreads) {
    const int i = k / m, j = k - i * m;
    const int ci = clampi(i, 1, n - 2), cj = clampi(j, 1, m - 2);
    const int ii = (ci - 1) * 2 + 1, jj = (cj - 1) * 2 + 1;
    const float *a = F.r + ii * mf + jj;
    Cc.r[k] = a[0] + a[1] + a[mf] + a[mf + 1];
    Cc.x[k] = 0.f;
  }
  __syncthreads();
}

// d = prolongate(coarse.x) (+setBC)  (MG.pde:139-152)
__device__ inline void mg_prolong(const Lv &F, const Lv &Cc) {
  const int n = F.n, m = F.m, N = n * m, mc = Cc.m;
  for (int k = threadIdx.x; k < N; k += kThreads) {
    const int i = k / m, j = k - i * m;
    const int ci = clampi(i, 1, n - 2), cj = clampi(j, 1, m - 2);
    F.d[k] = Cc.x[((ci - 1) / 2 + 1) * mc + ((cj - 1) / 2 + 1)];
  }
  __syncthreads();
}

// parity mode: the reference's sequential accumulations (Field.pde:302-318), one thread
__device__ inline float seq_inner(const Lv &L, double *sm) {
  if (threadIdx.x == 0) {
    double s = 0;
    for (int i = 1; i < L.n - 1; ++i)
      for (int j = 1; j < L.m - 1; ++j) {
        const float pr = L.r[i * L.m + j] * L.r[i * L.m + j];
        s += pr;
      }
    reinterpret_cast<float *>(sm)[0] = (float)s;
  }
  __syncthreads();
  const float v = reinterpret_cast<float *>(sm)[0];
  __syncthreads();
  return v;
}
__device__ inline float seq_sum(const float *a, int n, int m, double *sm) {
  if (threadIdx.x == 0) {
    float s = 0;
    for (int i = 1; i < n - 1; ++i)
      for (int j = 1; j < m - 1; ++j) s += a[i * m + j];
    reinterpret_cast<float *>(sm)[0] = s;
  }
  __syncthreads();
  const float v = reinterpret_cast<float *>(sm)[0];
  __syncthreads();
  return v;
}

// MGsolver(20, PoissonMatrix(c), p, div u)  (MG.pde:30-38) -- r must hold b - A x on entry
__device__ inline int mg_solve_v2(const StepParams &P, size_t e, double *sm, float *linf_out, Prof &pf, float *lds);
__device__ inline void mg_smooth4(const StepParams &P, const Lv &L, float *lds) {
  (void)P;
  (void)lds;
  mg_dinv(L);
  mg_gs(L, 4);
}

__device__ inline int mg_solve(const StepParams &P, size_t e, double *sm, float *linf_out, Prof &pf,
                               float *lds = nullptr) {
  if (P.impl != 0 && lds != nullptr) return mg_solve_v2(P, e, sm, linf_out, pf, lds);
  const int kc = P.nlev - 1;
  int iter = 0;
  const Lv L0 = make_level(P, e, 0);
  while (iter < P.mg_max_iters) {
    ++iter;  // MG.update(), MG.pde:61-66: vCycle(); smooth(its)
    for (int l = 0; l < kc; ++l) {  // descending leg of vCycle (MG.pde:69-70)
      const Lv F = make_level(P, e, l), Cc = make_level(P, e, l + 1);
      mg_dinv(F);  // smooth(0)
      mg_increment(F);
      mg_restrict(F, Cc);
      pf.mark(l == 0 ? PF_DOWN0 : PF_COARSE_DOWN);
    }
    {
      const Lv Lk = make_level(P, e, kc);  // coarsest: coarse.smooth(its), MG.pde:73
      mg_smooth4(P, Lk, lds);
      mg_increment(Lk);
      pf.mark(PF_COARSE_GS);
    }
    for (int l = kc - 1; l >= 0; --l) {  // ascending leg (MG.pde:75-76) + the caller's smooth(its)
      const Lv F = make_level(P, e, l), Cc = make_level(P, e, l + 1);
      mg_prolong(F, Cc);
      mg_increment(F);
      pf.mark(l == 0 ? PF_UP0 : PF_COARSE_UP);
      mg_smooth4(P, F, lds);
      pf.mark(l == 0 ? PF_GS0 : PF_COARSE_GS);
      mg_increment(F);
      pf.mark(l == 0 ? PF_INC0 : PF_COARSE_UP);
    }
    double part = 0;  // r.inner(r): float products, double accumulation (Field.pde:302-310)
    const int n = L0.n, m = L0.m;
    for (int k = threadIdx.x; k < n * m; k += kThreads) {
      const int i = k / m, j = k - i * m;
      if (i >= 1 && i <= n - 2 && j >= 1 && j <= m - 2) {
        const float pr = L0.r[k] * L0.r[k];
        part += (double)pr;
      }
    }
    float rr = (float)block_sum_d(part, sm);
    if (P.exact_sums) rr = seq_inner(L0, sm);
    pf.mark(PF_RR);
    if (rr < P.mg_tol) break;
  }
  float mx = 0;  // solver.r.L_inf(), MG.pde:36
  for (int k = threadIdx.x; k < L0.n * L0.m; k += kThreads) mx = jmax(mx, jabs(L0.r[k]));
  *linf_out = block_max_f(mx, sm);
  return iter;
}

// Phase D+E.  u.project(c, p)  (VectorField.pde:130-143)
__device__ inline int project_vel(const StepParams &P, size_t e, float *ux, float *uy, double *sm, float *linf_out,
                                  Prof &pf, float *lds = nullptr) {
  const int n = P.n, m = P.m;
  const size_t eo = e * (size_t)P.Ns;
  float *p = G(P.p) + eo, *r = G(P.r) + eo;
  const Lv L0 = make_level(P, e, 0);
  // r = b - A x with b = divergence (VectorField.pde:56-65, MG.pde:51)
  for (int k = threadIdx.x; k < P.N; k += kThreads) {
    const int i = k / m, j = k - i * m;
    float rv = 0.f;
    if (i >= 1 && i <= n - 2 && j >= 1 && j <= m - 2) {
      const float div = ux[k + m] - ux[k] + uy[k + 1] - uy[k];
      const float lw = L0.lx[k], le = L0.lx[k + m], ls = L0.ly[k], ln = L0.ly[k + 1];
      const float diag = -(lw + le + ls + ln);
      const float ap = p[k] * diag + p[k - m] * lw + p[k + m] * le + p[k - 1] * ls + p[k + 1] * ln;
      rv = div - ap;
    }
    r[k] = rv;
  }
  __syncthreads();
  pf.mark(PF_RESID);
  const int iters = mg_solve(P, e, sm, linf_out, pf, lds);
  // p.plusEq(-1*p.sum()/(float)((n-2)*(m-2)))  (VectorField.pde:136)
  double part = 0;
  for (int k = threadIdx.x; k < P.N; k += kThreads) {
    const int i = k / m, j = k - i * m;
    if (i >= 1 && i <= n - 2 && j >= 1 && j <= m - 2) part += (double)p[k];
  }
  float psum = (float)block_sum_d(part, sm);
  if (P.exact_sums) psum = seq_sum(p, n, m, sm);
  const float shift = -1 * psum / (float)((n - 2) * (m - 2));
  for (int k = threadIdx.x; k < P.N; k += kThreads) p[k] += shift;
  __syncthreads();
  // u -= c * grad p on the cells that survive the following setBC (VectorField.pde:137-139,
  // Field.pde:71-81)
  for (int k = threadIdx.x; k < P.N; k += kThreads) {
    const int i = k / m, j = k - i * m;
    if (i >= 1 && i <= n - 2 && j >= 1 && j <= m - 2) {
      const float pc = p[k];
      const float gx = pc - p[k - m], gy = pc - p[k - 1];
      ux[k] = ux[k] + G(P.cx)[k] * (gx * -1.f);
      uy[k] = uy[k] + G(P.cy)[k] * (gy * -1.f);
    }
  }
  __syncthreads();
  __shared__ float rowbuf_v1p[kRowCap];
  setbc_vel(P, ux, uy, rowbuf_v1p);
  pf.mark(PF_PROJECT);
  return iters;
}
__device__ inline int phase_project(const StepParams &P, size_t e, double *sm, float *linf_out, Prof &pf) {
  const size_t eo = e * (size_t)P.Ns;
  return project_vel(P, e, G(P.ux) + eo, G(P.uy) + eo, sm, linf_out, pf);
}

// Body.pressForce(flow.p).mult(-1) of the main cylinder (Body.pde:296-303, AFCCylinder.pde:58)
__device__ inline void eval_force(const StepParams &P, const float *p, float shift, float *fxo, float *fyo) {
  float px = 0, py = 0;
  const int m = P.m;
  for (int s = 0; s < P.nseg; ++s) {
    const SampleDev sd = G(P.fs)[s];
    const float *a = p + sd.base;
    float val;
    if (sd.exact) {
      val = a[0] + shift;
    } else {
      val = sd.s * (sd.t * (a[m + 1] + shift) + (1 - sd.t) * (a[m] + shift)) +
            (1 - sd.s) * (sd.t * (a[1] + shift) + (1 - sd.t) * (a[0] + shift));
    }
    const float pdl = val * G(P.seg_l)[s];
    px += pdl * G(P.seg_nx)[s];
    py += pdl * G(P.seg_ny)[s];
  }
  *fxo = px * -1.f;
  *fyo = py * -1.f;
}

struct ShBcast {
  float fx, fy;
  int it0, it1;
  float r0, r1;
};

// AFCCylinder.update2()  (AFCCylinder.pde:45-61) for env e
__device__ inline void sim_step(const StepParams &P, size_t e, float xi1_m, float xi2_m, double *sm,
                                ShBcast *sh) {
  const size_t eo = e * (size_t)P.Ns;
  const float dphi1 = (2 * xi1_m * P.dt) / (P.dR * P.D);  // AFCCylinder.pde:48-49
  const float dphi2 = (2 * xi2_m * P.dt) / (P.dR * P.D);
  float l0, l1;
  Prof pf;
  pf.start(P);
  // flow.update(body): BDIM.pde:126-129, 79-87
  phase_advdif(P, eo, false);
  pf.mark(PF_ADVDIF);
  phase_forcing(P, eo, dphi1, dphi2);
  pf.mark(PF_FORCING);
  phase_setbc_u(P, eo, sm);
  pf.mark(PF_SETBC);
  const int it0 = phase_project(P, e, sm, &l0, pf);
  // flow.update2(body): BDIM.pde:89-107
  phase_advdif(P, eo, true);
  pf.mark(PF_ADVDIF);
  phase_forcing(P, eo, dphi1, dphi2);
  pf.mark(PF_FORCING);
  phase_setbc_u(P, eo, sm);
  pf.mark(PF_SETBC);
  const int it1 = phase_project(P, e, sm, &l1, pf);
  {  // u = (u + us) * 0.5 on the whole array, no setBC (BDIM.pde:95-96)
    float *ux = G(P.ux) + eo, *uy = G(P.uy) + eo;
    const float *usx = G(P.usx) + eo, *usy = G(P.usy) + eo;
    for (int k = threadIdx.x; k < P.N; k += kThreads) {
      ux[k] = (ux[k] + usx[k]) * 0.5f;
      uy[k] = (uy[k] + usy[k]) * 0.5f;
    }
  }
  if (threadIdx.x == 0) {
    float fx, fy;
    eval_force(P, G(P.p) + eo, 0.f, &fx, &fy);
    sh->fx = fx;
    sh->fy = fy;
    sh->it0 = it0;
    sh->it1 = it1;
    sh->r0 = l0;
    sh->r1 = l1;
  }
  __syncthreads();
  pf.mark(PF_HEUN);
}

}  // namespace afc
