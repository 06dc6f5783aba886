// afc_env.hip -- libafc_env.so: C ABI (include/afc_env.h) + kernels for the batched BDIM
// cylinder-wake environment on MI355X (gfx950).  No CPU fallback: every compute entry point
// fails with AFC_ENODEV when no HIP device is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/afc_env.h"
#include "bdim_device.h"
#define AFC_ONLY_MODE 0  // host-side helpers of the optimised phases only (LDS sizes)
#include "bdim_v2.h"
#include "step_launch.h"
#include "geometry.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(AFC_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));                    \
  } while (0)

using afc::EnvScalars;
using afc::G;
using afc::kThreads;
using afc::StepParams;

// ---------------- kernels (the step kernels live in step_kernels.h: three translation units) ----------------
__global__ void k_reset(StepParams P, const uint8_t *__restrict__ mask, const float *__restrict__ ru,
                        const float *__restrict__ rv, const float *__restrict__ rp, int keep_window) {
  const size_t e = blockIdx.x;
  if (mask && !mask[e]) return;
  const size_t eo = e * (size_t)P.Ns;
  for (int k = threadIdx.x; k < P.N; k += blockDim.x) {
    G(P.ux)[eo + k] = ru[k];
    G(P.uy)[eo + k] = rv[k];
    G(P.p)[eo + k] = rp[k];
    G(P.u0x)[eo + k] = 0.f;  // BDIM.pde:53
    G(P.u0y)[eo + k] = 0.f;
  }
  if (threadIdx.x == 0) {
    EnvScalars s = G(P.sc)[e];
    const int cl = s.callLearn;
    const float cd = s.Cd, cll = s.Cl;
    int hist[8];
    for (int q = 0; q < 8; ++q) hist[q] = s.it_hist[q];  // diagnostics survive an episode roll-over
    s = EnvScalars{};
    for (int q = 0; q < 8; ++q) s.it_hist[q] = hist[q];
    s.callLearn = P.substeps;
    if (keep_window) {  // clientCFD.pde:12-13 are sketch globals that setUpNewSim does not reset
      s.callLearn = cl;
      s.Cd = cd;
      s.Cl = cll;
    }
    G(P.sc)[e] = s;
  }
}

__global__ void k_vorticity(StepParams P, int env_idx, float *__restrict__ out) {  // VectorField.pde:78-87
  const size_t eo = (size_t)env_idx * P.Ns;
  const float *x = G(P.ux) + eo, *y = G(P.uy) + eo;
  const int n = P.n, m = P.m;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < P.N; k += gridDim.x * blockDim.x) {
    const int i = k / m, j = k - i * m;
    float v = 0.f;
    if (i >= 1 && i <= n - 2 && j >= 1 && j <= m - 2) v = x[k - 1] - x[k] + y[k] - y[k - m];
    out[k] = v;
  }
}

__global__ void k_probes(StepParams P, const afc::SampleDev *__restrict__ ps, int num, float *__restrict__ out) {
  const int e = blockIdx.x;
  const float *p = G(P.p) + (size_t)e * P.Ns;
  const float sh = G(P.sc)[e].p_shift;  // pending mean shift of the last projection (bdim_v2.h)
  const int m = P.m;
  for (int q = threadIdx.x; q < num; q += blockDim.x) {
    const afc::SampleDev sd = ps[q];
    const float *a = p + sd.base;
    float val;
    if (sd.exact)
      val = a[0] + sh;
    else
      val = sd.s * (sd.t * (a[m + 1] + sh) + (1 - sd.t) * (a[m] + sh)) +
            (1 - sd.s) * (sd.t * (a[1] + sh) + (1 - sd.t) * (a[0] + sh));
    out[(size_t)e * num + q] = val;
  }
}

// VectorField.CFL (VectorField.pde:225-235) / BDIM.checkCFL (BDIM.pde:217): 1/(max(|u|+|v|) + 3 nu), the max taken
// over the interior starting from cell [0][0]; max is order-independent, so a tree reduction gives the same bits.
__global__ void k_cfl(StepParams P, float *__restrict__ out) {
  __shared__ float sm[256];
  const int e = blockIdx.x;
  const float *x = G(P.ux) + (size_t)e * P.Ns, *y = G(P.uy) + (size_t)e * P.Ns;
  const int n = P.n, m = P.m;
  float b = afc::jabs(x[0]) + afc::jabs(y[0]);
  for (int k = threadIdx.x; k < P.N; k += blockDim.x) {
    const int i = k / m, j = k - i * m;
    if (i >= 1 && i <= n - 2 && j >= 1 && j <= m - 2) {
      const float c = afc::jabs(x[k]) + afc::jabs(y[k]);
      if (c > b) b = c;
    }
  }
  sm[threadIdx.x] = b;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s && sm[threadIdx.x + s] > sm[threadIdx.x]) sm[threadIdx.x] = sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[e] = 1.f / (sm[0] + 3.f * P.nu);
}

// unit-level operators on env 0 (same device functions as the step; advdif / mg: step_kernels.h)
__global__ __launch_bounds__(afc::kThreads) void k_op_setbc(StepParams P) {
  __shared__ double sm[afc::kThreads];
  afc::phase_setbc_u(P, 0, sm);
}

// Measurement helper (bench.py's copy ceiling): a plain 16-byte-per-lane device-to-device copy -- 2048 workgroups of 256
// threads, every thread with eight independent 16-byte loads in flight before its first store, tiles of 32 KB dealt
// to the workgroups with a grid stride -- the hard ceiling a streaming pass can reach on the box.
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_copy_f4(const f4v *__restrict__ src, f4v *__restrict__ dst, size_t n4) {
  constexpr int U = 8;
  const size_t tile = (size_t)256 * U;
  for (size_t base = (size_t)blockIdx.x * tile; base < n4; base += (size_t)gridDim.x * tile) {
    f4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t k = base + (size_t)u * 256 + threadIdx.x;
      if (k < n4) v[u] = NT ? __builtin_nontemporal_load(src + k) : src[k];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t k = base + (size_t)u * 256 + threadIdx.x;
      if (k < n4) {
        if (NT)
          __builtin_nontemporal_store(v[u], dst + k);
        else
          dst[k] = v[u];
      }
    }
  }
}
}  // namespace

// ---------------- host side ----------------
struct afc_env {
  afc_config cfg;
  afc::CaseParams cp;
  afc::StaticFields sf;
  StepParams P;
  hipStream_t stream = nullptr;
  int io_space = AFC_MEM_HOST;
  int E = 0;
  std::vector<void *> allocs;
  bool slab_shared = false;  // development (AFC_DEVHOOKS builds): the state slab belongs to another instance and is not freed here
  int place_trials = 0;      // placement search of afc_create: candidates probed (0: no search), best / median probe rate
  float place_best = 0.f, place_median = 0.f;
  char *arena = nullptr;  // one allocation for all the small device buffers (tables shared by the envs, scalars, staging)
  size_t arena_size = 0, arena_off = 0;
  float *rs_u = nullptr, *rs_v = nullptr, *rs_p = nullptr;  // reset state (device)
  float reset_flow_t = 0, reset_flow_dt = 0;
  // device staging for host-space I/O
  float *d_in = nullptr, *d_f2 = nullptr, *d_f1 = nullptr;
  uint8_t *d_u8 = nullptr;
  int32_t *d_i32 = nullptr;
  long long *d_stamps = nullptr;
  size_t lds_bytes = 0;
  size_t lds_bytes_step = 0;  // dynamic LDS of the shipped / perf step kernels (>= lds_bytes: + the fused advection's three row slots)
  size_t lds_bytes_lat = 0;   // ... of afc::lat's (>= lds_bytes_step: the rings + special-cell table of mg_up_tr)
  StepParams *d_P = nullptr;  // device copy of P for the step kernels
  bool p_dirty = true;
  int cpl = 0;  // kernel instantiation: columns per lane (0 = reference kernels)
  std::vector<int> level_off;
  afc::SampleDev *d_probes = nullptr;  // probe tables are cached per `num` (freed with the env)
  int n_probes = 0;
  std::vector<std::pair<int, afc::SampleDev *>> probe_tables;
  float *d_probe_out = nullptr;
  size_t probe_out_cap = 0;
  float *d_cfl = nullptr;
};

namespace {

// Buffers of up to 16 MB come out of ONE arena per env (created on first use): the coefficient tables, the multigrid
// hierarchy and the record table of mg_bottom -- what every workgroup keeps re-reading out of L2 -- then sit in one
// contiguous range instead of wherever a dozen separate hipMalloc calls put them.  AFC_ARENA=0: separate allocations.
// Sized from the case: ~45 arrays of one grid's size (coefficients, body tables, the multigrid hierarchy, reset state)
// plus per-env staging; what does not fit falls back to a separate allocation.
constexpr size_t kArenaMaxBytes = (size_t)96 << 20, kArenaMaxItem = (size_t)16 << 20;
inline size_t arena_bytes_for(size_t cells, size_t envs) {
  size_t b = 64 * cells * sizeof(float) + envs * 64 + ((size_t)2 << 20);
  b = (b + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  return b > kArenaMaxBytes ? kArenaMaxBytes : b;
}
template <typename T>
int dev_alloc(afc_env *env, T **p, size_t count) {
  const size_t bytes = count * sizeof(T);
  const char *ev = getenv("AFC_ARENA");
  const bool use_arena = !(ev && atoi(ev) == 0);
  if (use_arena && bytes <= kArenaMaxItem) {
    if (!env->arena) {
      void *a = nullptr;
      const size_t ab = arena_bytes_for((size_t)env->sf.n * (size_t)env->sf.m, (size_t)env->E);
      HIP_TRY(hipMalloc(&a, ab));
      env->allocs.push_back(a);
      env->arena = static_cast<char *>(a);
      env->arena_size = ab;
      env->arena_off = 0;
    }
    const size_t off = (env->arena_off + 255) & ~(size_t)255;
    if (off + bytes <= env->arena_size) {
      *p = reinterpret_cast<T *>(env->arena + off);
      env->arena_off = off + bytes;
      return AFC_OK;
    }
  }
  void *q = nullptr;
  HIP_TRY(hipMalloc(&q, bytes));
  env->allocs.push_back(q);
  *p = static_cast<T *>(q);
  return AFC_OK;
}
template <typename T>
int dev_upload(afc_env *env, const T **p, const std::vector<T> &h) {
  T *q = nullptr;
  int rc = dev_alloc(env, &q, h.size());
  if (rc) return rc;
  HIP_TRY(hipMemcpy(q, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *p = q;
  return AFC_OK;
}

// StepParams members are address_space(1) pointers (see bdim_device.h): same allocations, typed
template <typename T>
int dev_alloc(afc_env *env, AFC_AS1 T **p, size_t count) {
  T *q = nullptr;
  int rc = dev_alloc(env, &q, count);
  *p = (AFC_AS1 T *)q;
  return rc;
}
template <typename T>
int dev_upload(afc_env *env, const AFC_AS1 T **p, const std::vector<T> &h) {
  const T *q = nullptr;
  int rc = dev_upload(env, &q, h);
  *p = (const AFC_AS1 T *)q;
  return rc;
}

int check_env(const afc_env *env) {
  if (!env) return fail(AFC_EINVAL, "afc: null env handle");
  return AFC_OK;
}

afc::SampleDev to_dev(const afc::PressSample &s, int m) {
  afc::SampleDev d;
  d.base = s.i * m + s.j;
  d.s = s.s;
  d.t = s.t;
  d.exact = s.exact;
  return d;
}

int sync_params(afc_env *env) {
  if (!env->p_dirty) return AFC_OK;
  HIP_TRY(hipMemcpyAsync(env->d_P, &env->P, sizeof(StepParams), hipMemcpyHostToDevice, env->stream));
  HIP_TRY(hipStreamSynchronize(env->stream));
  env->p_dirty = false;
  return AFC_OK;
}

// the instantiation with the diagnostics compiled in runs only when one of them is switched on
bool need_diag(const afc_env *env) {
  const StepParams &P = env->P;
  return (AFC_EXP_STAMPS ? (P.dbg & ~8) != 0 : P.dbg != 0) || P.upleg == 2 || P.stagger_ticks > 0 || P.gate_limit != 0 || P.stamps != nullptr;
}

// 0 = shipped kernels, 1 = diagnostics compiled in, 2 = perf mode (red-black smoother; diagnostics are not available in it),
// 3 = the shipped code with the up-legs marching down the rows (afc::lat: launches of few envs per CU)
int launch_mode(const afc_env *env) {
  if (env->P.gs_mode == 1) return 2;
  if (need_diag(env)) return 1;
  return (env->P.upleg == 3 && env->cpl >= 1 && env->cpl <= 2) ? 3 : 0;
}

int launch_check(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AFC_EHIP, std::string(what) + ": " + hipGetErrorString(e));
  return AFC_OK;
}

}  // namespace

extern "C" {

const char *afc_last_error(void) { return g_err.c_str(); }

int afc_default_config(afc_config *cfg) {
  if (!cfg) return fail(AFC_EINVAL, "afc_default_config: null");
  memset(cfg, 0, sizeof(*cfg));
  cfg->resolution = 24;
  cfg->x_lengths = 16;
  cfg->y_lengths = 8;
  cfg->reynolds = 500;
  cfg->t_step = 0.0075f;
  cfg->dR = 0.125f;
  cfg->gR = 0.2f;
  cfg->theta = 3.14159265358979323846f / 3;  // PI/3 in float (clientCFD.pde:96)
  cfg->act_scale = 5.f;
  cfg->substeps = 16;
  cfg->init_time = 1.f;
  cfg->end_time = 50.f;
  cfg->num_envs = 1;
  cfg->device = 0;
  cfg->mg_max_iters = 20;
  cfg->carry_quirk = 0;
  cfg->exact_sums = 0;
  cfg->impl = 2;
  cfg->reward_penalty = (float)(3.14159265358979323846 * (1.0 / 8) * 0.0097 * (3.66 * 3.66 * 3.66));
  return AFC_OK;
}

int afc_create(const afc_config *cfg, afc_env **out) {
  if (!cfg || !out) return fail(AFC_EINVAL, "afc_create: null argument");
  *out = nullptr;
  if (cfg->num_envs < 1) return fail(AFC_EINVAL, "afc_create: num_envs must be >= 1");
  if (cfg->substeps < 1 || cfg->mg_max_iters < 1) return fail(AFC_EINVAL, "afc_create: substeps/mg_max_iters");
  if (cfg->impl != 0 && cfg->impl != 2) return fail(AFC_EINVAL, "afc_create: impl must be 0 (reference kernels) or 2");
  if (cfg->gs_mode != 0 && cfg->gs_mode != 1) return fail(AFC_EINVAL, "afc_create: gs_mode must be 0 (lexicographic) or 1 (red-black)");
  if (cfg->gs_mode == 1 && cfg->impl != 2) return fail(AFC_EINVAL, "afc_create: gs_mode 1 needs impl 2");
  if (cfg->gs_mode == 1 && cfg->exact_sums) return fail(AFC_EINVAL, "afc_create: gs_mode 1 (perf mode) and exact_sums (parity mode) exclude each other");
  // the perf-mode instantiation of the kernels carries none of the diagnostics: refuse what it would silently ignore
  if (cfg->gs_mode == 1 && (cfg->stagger_us != 0 || cfg->reserved[0] != 0))
    return fail(AFC_EINVAL, "afc_create: gs_mode 1 (perf mode) has no diagnostics: stagger_us and reserved[0] (dbg) must be 0");
  if (cfg->y_lengths * cfg->resolution > 256 && cfg->impl != 0) return fail(AFC_EINVAL, "afc_create: impl 2 supports up to 256 cells in y");
  if (cfg->fuse_tail < -1 || cfg->fuse_tail > 1) return fail(AFC_EINVAL, "afc_create: fuse_tail must be -1 (never), 0 (where it is faster) or 1 (wherever the grid allows)");
  if (cfg->upleg < 0 || cfg->upleg > 3)
    return fail(AFC_EINVAL, "afc_create: upleg must be 0 (by occupancy), 1 (strip pipeline), 2 (barrier form) or 3 (marching down the rows)");
  if (cfg->gs_mode == 1 && cfg->upleg != 0) return fail(AFC_EINVAL, "afc_create: gs_mode 1 (perf mode) has its own up-leg: upleg must be 0");
  int ndev = 0;
  hipError_t he = hipGetDeviceCount(&ndev);
  if (he != hipSuccess || ndev < 1)
    return fail(AFC_ENODEV, "afc_create: no HIP device available (this library has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(AFC_EINVAL, "afc_create: device ordinal out of range");
  HIP_TRY(hipSetDevice(cfg->device));

  afc_env *env = new afc_env();
  env->cfg = *cfg;
  env->E = cfg->num_envs;
  env->cp.resolution = cfg->resolution;
  env->cp.x_lengths = cfg->x_lengths;
  env->cp.y_lengths = cfg->y_lengths;
  env->cp.reynolds = cfg->reynolds;
  env->cp.t_step = cfg->t_step;
  env->cp.dR = cfg->dR;
  env->cp.gR = cfg->gR;
  env->cp.theta = cfg->theta;
  try {
    env->sf = afc::build_static(env->cp);
  } catch (const std::exception &ex) {
    std::string msg = ex.what();
    delete env;
    return fail(AFC_EINVAL, msg);
  }
  const afc::StaticFields &sf = env->sf;
  if ((int)sf.levels.size() > afc::kMaxLevels) {
    delete env;
    return fail(AFC_EINVAL, "afc_create: too many multigrid levels");
  }
  StepParams &P = env->P;
  memset(&P, 0, sizeof(P));
  P.n = sf.n;
  P.m = sf.m;
  P.N = sf.n * sf.m;
  P.Ns = (P.N + 63) / 64 * 64;
  P.nlev = (int)sf.levels.size();
  P.dt = sf.dt;
  P.nu = sf.nu;
  P.D = sf.D;
  P.dR = cfg->dR;
  P.mg_tol = sf.mg_tol;
  P.mg_max_iters = cfg->mg_max_iters;
  P.resolution = cfg->resolution;
  P.act_scale = cfg->act_scale;
  P.init_time = cfg->init_time;
  P.end_time = cfg->end_time;
  P.reward_penalty = cfg->reward_penalty;
  P.substeps = cfg->substeps;
  P.carry_quirk = cfg->carry_quirk;
  P.exact_sums = cfg->exact_sums;
  P.impl = cfg->impl;
  P.gs_mode = cfg->gs_mode;
  P.dbg = cfg->reserved[0];
  P.upleg = cfg->upleg;  // (0 is resolved below, once the levels are known)
  P.stagger_ticks = (cfg->stagger_us & 0xFFFFF) * 100;
  P.stagger_shift = (cfg->stagger_us >> 20) & 63;  // experimental: bits 0-3 = block-index shift, bits 4-5 = phases-2
  P.adv_rows = (sf.m <= 130) ? 16 : 8;
  env->cpl = (P.impl == 0) ? 0 : ((sf.m - 2) + 63) / 64;
  {
    const size_t tile = 2 * ((size_t)(P.adv_rows + 4) * sf.m + 4) + 8;  // advdif_v2 (16-byte aligned tiles)
    // mg_up_smooth's rings on the finest level, with the strip height of this grid's kernel instantiation (coarser
    // levels have at most half the columns and always fit 16-row strips)
    // (perf mode keeps its 12-row strips of eight shifted stages in 24-row rings on every grid: a wide grid then runs
    // fewer workgroups per CU, as in round 3)
    const size_t rings = (size_t)afc::prod::gs_lds_floats(sf.m, cfg->gs_mode == 1 ? 16 : afc::prod::up_strip_h(((sf.m - 2) + 63) / 64)) + 64;
    size_t fl = tile > rings ? tile : rings;
    // multi-wave smoother: both rings of every level live in the same region; with four workgroups per CU a
    // workgroup may use 40 KB of LDS in all (static part ~2.3 KB)
    const size_t mw_budget = (36 * 1024 + 512) / sizeof(float);
    if (fl < mw_budget) fl = mw_budget;
    // advdif_v3's rings (4 waves x (2 x 7 + 4) rows of one column tile): may take the region up to what still leaves
    // four workgroups per CU (160 KB / 4 = 40960 B per workgroup, ~2.3 KB of it static).  Grids of 64 or 128 interior
    // columns are one tile wide (C = 1, 2 cells per lane); 192 and 256 columns are two tiles of 96 / 128 (C = 2)
    const int m_ = sf.m - 2, n_ = sf.n - 2, cplv = (m_ + 63) / 64, ctile = cplv > 2 ? 2 : cplv;
    const bool one_tile = (m_ == 64 * ctile), two_tiles = (cplv > 2 && (m_ & 3) == 0 && m_ / 2 <= 64 * ctile);
    const size_t a3 = (size_t)afc::prod::adv3_lds_floats(ctile);
    if ((one_tile || two_tiles) && a3 > fl && a3 * sizeof(float) + 2560 <= 40960) fl = a3;
    env->lds_bytes = (P.impl == 0) ? 0 : sizeof(float) * fl;
    // advdif_v3 (one barrier-free row pipeline per wave): the rings fit the LDS region, >= 8 rows per wave
    P.adv_ns = 0;
    const int rows_per_wave = one_tile ? (n_ + 3) / 4 : (n_ + 1) / 2;
    if (P.impl != 0 && (one_tile || two_tiles) && a3 <= fl && rows_per_wave >= 8) P.adv_ns = afc::prod::kAdvNS;
  }

  // advdif_v3f (a projection tail inside the advection pass that follows it): one more ring row per wave.  The
  // diagnostic instantiation never runs it and keeps the smaller region (its static part is 2 KB larger); every other
  // "does it fit" above and below is answered for the smaller region.
  env->lds_bytes_step = env->lds_bytes;
  P.fuse_tail = 0;
  {
    const int m_ = sf.m - 2, cplv = (m_ + 63) / 64;
    const size_t af = (size_t)afc::prod::advf_lds_floats(cplv > 2 ? 2 : cplv) * sizeof(float);
    // Measured on the same memory (tools/ab_inproc.py --share-slab, 256x128, per launch of 16 sim-steps; first figure:
    // only the predictor's tail fused, second: the corrector's too):
    //   1024 envs -1.3 / -4.3 %   768: -1.4 / -6.2 %   512: 0.0 / -4.1 %   256: +1.5 / -0.8 %   64: +2.6 / +2.6 %   1: +3.0 / +4.3 %
    // -- a wave alone on its SIMD has nobody to hide the projection's LDS round trips behind.  So by default: both, in
    // launches of at least one env per CU (afc_config.fuse_tail: 1 = wherever the shape allows, -1 = never).
    int cus = 256;
    {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    int level = (cfg->fuse_tail > 0 || (cfg->fuse_tail == 0 && (long long)env->E >= (long long)cus)) ? 3 : 0;
#if AFC_DEVHOOKS
    // tools/ab_inproc.py lib.so:AFC_FUSE_TAIL=0 lib.so:AFC_FUSE_TAIL=1 ... (bits: 1 the predictor's tail, 2 the corrector's; 3 both)
    if (const char *ft = getenv("AFC_FUSE_TAIL")) level = atoi(ft) & 3;
#endif
    if (level & 2) level |= 1;  // (the second needs the first: `us` must not live in the field the predictor pass writes)
    if (P.impl != 0 && P.adv_ns && cplv <= 4 && afc::prod::advf_ok_for(sf.n, sf.m, cplv) && af + 512 <= 40960 && level > 0) {
      // bit 0: the predictor's tail in the corrector's advection; bit 1: the corrector's tail + Heun average in the next
      // sim-step's predictor advection, inside a launch of several sim-steps
      P.fuse_tail = level;
      if (af > env->lds_bytes_step) env->lds_bytes_step = af;
    }
  }
  P.head_v3 = (P.impl != 0 && sf.m == 130 && ((sf.n - 2) & 7) == 0 &&
               (size_t)afc::prod::kHeadLdsFloats * sizeof(float) <= env->lds_bytes) ? 1 : 0;
  P.head_v4 = (P.impl != 0 && afc::prod::head4_shape_ok(sf.n, sf.m) &&
               (size_t)afc::prod::head4_lds_floats(sf.m) * sizeof(float) <= env->lds_bytes) ? 1 : 0;
  P.down_v3 = 0;
  if (P.impl != 0) {  // down_v3c: the rings of every level it would serve must fit the region
    P.down_v3 = 1;
    for (size_t l = 1; l < sf.levels.size(); ++l)
      if (afc::prod::down3_shape_ok(sf.levels[l].n, sf.levels[l].m) &&
          (size_t)afc::prod::down3_lds_floats(sf.levels[l].m) * sizeof(float) > env->lds_bytes)
        P.down_v3 = 0;
  }
  int rc = AFC_OK;
#define UP(field) if (!rc) rc = dev_upload(env, &P.field, sf.field)
  UP(delx); UP(dely); UP(del1x); UP(del1y); UP(wnxx); UP(wnxy); UP(wnyx); UP(wnyy); UP(cx); UP(cy);
  UP(w1x); UP(w2x); UP(ry1x); UP(ry2x); UP(w1y); UP(w2y); UP(rx1y); UP(rx2y);
  UP(seg_l); UP(seg_nx); UP(seg_ny);
  // band cells within reach of a control cylinder (a non-zero weight at the cell or at one of its four neighbours)
  // first: the others skip the body-velocity part of the forcing (ub == +0 there, exactly)
  std::vector<int> bxs = sf.band_x, bys = sf.band_y;
  auto reach_x = [&](int k) {
    for (int o : {0, sf.m, -sf.m, 1, -1})
      if (sf.w1x[k + o] != 0.f || sf.w2x[k + o] != 0.f) return true;
    return false;
  };
  auto reach_y = [&](int k) {
    for (int o : {0, sf.m, -sf.m, 1, -1})
      if (sf.w1y[k + o] != 0.f || sf.w2y[k + o] != 0.f) return true;
    return false;
  };
  P.nflag_x = (int)(std::stable_partition(bxs.begin(), bxs.end(), reach_x) - bxs.begin());
  P.nflag_y = (int)(std::stable_partition(bys.begin(), bys.end(), reach_y) - bys.begin());
  {
    const std::vector<int> *bx_ = &bxs, *by_ = &bys;
    if (!rc) rc = dev_upload(env, &P.band_x, *bx_);
    if (!rc) rc = dev_upload(env, &P.band_y, *by_);
  }
  P.nband_x = (int)sf.band_x.size();
  P.nband_y = (int)sf.band_y.size();
  {  // the compact tables
    const size_t nb = bxs.size() > bys.size() ? bxs.size() : bys.size();
    const int stride = (int)((nb + 63) / 64 * 64) + 64;
    std::vector<float> tx((size_t)24 * stride, 0.f), ty((size_t)24 * stride, 0.f);
    const int offs[5] = {0, sf.m, -sf.m, 1, -1};
    for (size_t q = 0; q < bxs.size(); ++q) {
      const int k = bxs[q];
      tx[q] = sf.delx[k], tx[stride + q] = sf.del1x[k], tx[2 * (size_t)stride + q] = sf.wnxx[k], tx[3 * (size_t)stride + q] = sf.wnyx[k];
      for (int p_ = 0; p_ < 5; ++p_) {
        const int kk = k + offs[p_];
        float *t4 = tx.data() + (size_t)(4 + 4 * p_) * stride + q;
        t4[0] = sf.w1x[kk], t4[stride] = sf.w2x[kk], t4[2 * (size_t)stride] = sf.ry1x[kk], t4[3 * (size_t)stride] = sf.ry2x[kk];
      }
    }
    for (size_t q = 0; q < bys.size(); ++q) {
      const int k = bys[q];
      ty[q] = sf.dely[k], ty[stride + q] = sf.del1y[k], ty[2 * (size_t)stride + q] = sf.wnxy[k], ty[3 * (size_t)stride + q] = sf.wnyy[k];
      for (int p_ = 0; p_ < 5; ++p_) {
        const int kk = k + offs[p_];
        float *t4 = ty.data() + (size_t)(4 + 4 * p_) * stride + q;
        t4[0] = sf.w1y[kk], t4[stride] = sf.w2y[kk], t4[2 * (size_t)stride] = sf.rx1y[kk], t4[3 * (size_t)stride] = sf.rx2y[kk];
      }
    }
    P.band_stride = (P.impl != 0) ? stride : 0;
    if (!rc) rc = dev_upload(env, &P.band_tab_x, tx);
    if (!rc) rc = dev_upload(env, &P.band_tab_y, ty);
  }
  {  // wall columns folded into the row stores: the forcing band (and the cells it reads) must stay clear of the walls
    bool clear = true;
    for (const std::vector<int> *b : {&sf.band_x, &sf.band_y})
      for (int k : *b) {
        const int i = k / sf.m, j = k % sf.m;
        if (i < 3 || i > sf.n - 4 || j < 3 || j > sf.m - 4) clear = false;
      }
    P.bc_fold = (P.impl != 0 && clear && sf.m <= afc::kRowCap) ? 1 : 0;
    // the fused advection passes stage rows WITH their wall columns: no fusion without the folded walls (the device gates
    // on bc_fold too; decided here so that afc_fused_tail() and the slab's q fields follow what really runs).  Perf mode
    // (validation only) keeps the two-pass form: no test drives its kernels through the fused passes.
    if (!P.bc_fold || cfg->gs_mode == 1) {
      P.fuse_tail = 0;
      env->lds_bytes_step = env->lds_bytes;
    }
  }
#undef UP
  // multigrid hierarchy, all levels concatenated
  std::vector<float> lx, ly, inv;
  int coff = 0, soff = 0;
  for (int l = 0; l < P.nlev; ++l) {
    const afc::MGLevel &L = sf.levels[l];
    P.lev[l].n = L.n;
    P.lev[l].m = L.m;
    P.lev[l].coef_off = coff;
    P.lev[l].scratch_off = (l == 0) ? 0 : soff;
    {  // analytic coefficient model of the level + rows where the real arrays deviate from it
      afc::LevelDesc &D = P.lev[l];
      D.cu = sf.dt;
      D.zero_bnd = (l > 0) ? 1 : 0;
      for (int rc = 0; rc < 3; ++rc)
        for (int cc = 0; cc < 3; ++cc) {
          const float lw = (rc == 1 && D.zero_bnd) ? 0.f : D.cu, le = (rc == 2 && D.zero_bnd) ? 0.f : D.cu;
          const float ls = (cc == 1 && D.zero_bnd) ? 0.f : D.cu, ln = (cc == 2 && D.zero_bnd) ? 0.f : D.cu;
          const float sumd = lw + le + ls + ln;
          D.invtab[3 * rc + cc] = (sumd > 1e-5f) ? -1.f / sumd : 1.f;
        }
      {  // strip height of the multi-wave smoother: the largest H <= 63 whose two rings fit the LDS region
        const int budget = (int)(env->lds_bytes / sizeof(float));
        int H = L.n - 2 + 3;  // never more lanes than rows (+ the three shifted ones)
        if (H > 62) H = 62;  // lane 0 re-plays the row above the strip, lane H+1 holds the coefficients of the row below
        while (H > 4 && afc::prod::mw_lds_floats(L.m, H) > budget) --H;
        D.mw_H = H;
        D.dbg_copy = cfg->reserved[0];
      }
      D.sp_i0 = D.sp_j0 = 1 << 30;
      D.sp_i1 = D.sp_j1 = -1;
      for (int i = 1; i < L.n - 1; ++i)
        for (int j = 1; j < L.m - 1; ++j) {
          const size_t k = (size_t)i * L.m + j;
          const int rc = (i == 1) ? 1 : ((i == L.n - 2) ? 2 : 0), cc = (j == 1) ? 1 : ((j == L.m - 2) ? 2 : 0);
          const float lw = (rc == 1 && D.zero_bnd) ? 0.f : D.cu, le = (rc == 2 && D.zero_bnd) ? 0.f : D.cu;
          const float ls = (cc == 1 && D.zero_bnd) ? 0.f : D.cu, ln = (cc == 2 && D.zero_bnd) ? 0.f : D.cu;
          if (L.lx[k] != lw || L.lx[k + L.m] != le || L.ly[k] != ls || L.ly[k + 1] != ln ||
              L.inv[k] != D.invtab[3 * rc + cc]) {
            if (i < D.sp_i0) D.sp_i0 = i;
            if (i > D.sp_i1) D.sp_i1 = i;
            if (j < D.sp_j0) D.sp_j0 = j;
            if (j > D.sp_j1) D.sp_j1 = j;
          }
        }
    }
    // perf mode: up-leg as register-resident row pipelines where the level has the shape and the rows that cross a body
    // (their coefficient rows are staged in LDS) fit the region; AFC_RBW=<mask of levels>, 0: the barrier form everywhere (same bits)
    P.lev[l].rbw = 0;
    {
      const char *ev = getenv("AFC_RBW");
      // default: the finest level only -- measured per level (1024 envs, 256x128, ms per launch against 33.7 for the barrier
      // form everywhere): level 0 alone 32.6, level 1 alone 33.9, level 2 alone 35.7, levels 3-6 36.6, all 37.1: on the small
      // levels a wave's serial walk of n_ + 19 steps (almost all of them boundary / body steps) loses to 256 threads x few barriers
      const int lmask = ev ? atoi(ev) : 1;  // (bit l: level l)
      if (cfg->gs_mode == 1 && ((lmask >> l) & 1) && afc::prod::up_rbw_shape_ok(L.n, L.m) &&
          (size_t)afc::prod::up_rbw_lds_floats(L.m, P.lev[l].sp_i0, P.lev[l].sp_i1) * sizeof(float) <= env->lds_bytes)
        P.lev[l].rbw = 1;
    }
    if (getenv("AFC_PRINT_ALLOC"))
      fprintf(stderr, "afc_create: level %d (%d x %d): special rows [%d, %d], columns [%d, %d]\n", l, L.n - 2, L.m - 2,
              P.lev[l].sp_i0, P.lev[l].sp_i1, P.lev[l].sp_j0, P.lev[l].sp_j1);
    const int cnt = L.n * L.m;
    lx.insert(lx.end(), L.lx.begin(), L.lx.end());
    ly.insert(ly.end(), L.ly.begin(), L.ly.end());
    inv.insert(inv.end(), L.inv.begin(), L.inv.end());
    const int pad = (64 - cnt % 64) % 64;
    lx.insert(lx.end(), pad, 0.f);
    ly.insert(ly.end(), pad, 0.f);
    inv.insert(inv.end(), pad, 1.f);
    coff += cnt + pad;
    if (l >= 1) soff += cnt + pad;
  }
  P.Cs = 4 * ((soff + 63) / 64 * 64);
  {  // the bottom of the V-cycle out of LDS: needs at least two levels there, one level above it, and room
    const int lb = afc::prod::bottom_first_level(P.lev, P.nlev);
    P.bottom_lb = (P.impl != 0 && lb >= 2 && lb <= P.nlev - 2 && afc::prod::bottom_shape_ok(P.lev, P.nlev) &&
                   (size_t)afc::prod::bottom_lds_floats(P.lev, P.nlev) * sizeof(float) <= env->lds_bytes) ? lb : 0;
  }

  // mg_up_tr (the sweeps marching down the rows): the levels above the bottom of the kernels for grids of up to 128
  // columns whose shape it has and whose rings + special-cell table fit the LDS region -- bit 0 that of afc::lat (sized
  // for them: still four workgroups per CU), bit 1 the diagnostic instantiation's
  env->lds_bytes_lat = env->lds_bytes_step;
  bool any_tm = false;
  for (int l = 0; l < P.nlev; ++l) {
    afc::LevelDesc &D = P.lev[l];
    D.tm = 0;
    const bool above_bottom = P.bottom_lb ? (l < P.bottom_lb) : (l < P.nlev - 1);
    if (P.impl == 0 || cfg->gs_mode != 0 || env->cpl > 2 || !above_bottom || !afc::prod::tm_shape_ok(D.n, D.m)) continue;
    const size_t need_lat = (size_t)afc::prod::tm_lds_floats(D.m, D.sp_i0, D.sp_i1, D.sp_j0, D.sp_j1) * sizeof(float);
    const size_t need_diag = (size_t)afc::prod::tm_lds_floats(D.m, D.sp_i0, D.sp_i1, D.sp_j0, D.sp_j1, 24) * sizeof(float);
    if (need_lat + 512 <= 40960) {
      D.tm |= 1;
      any_tm = true;
      if (need_lat > env->lds_bytes_lat) env->lds_bytes_lat = need_lat;
    }
    if (need_diag <= env->lds_bytes) D.tm |= 2;
#if AFC_DEVHOOKS
    if (const char *tmk = getenv("AFC_TR_MASK")) {  // tools/ab_inproc.py lib.so:AFC_TR_MASK=<bit l: level l marches down the rows>
      if (!((atoi(tmk) >> l) & 1)) D.tm = 0;
    }
#endif
  }
  // afc_config.upleg = 0: by occupancy.  Measured on the same memory (tools/ab_inproc.py --share-slab, 256x128, per launch
  // of 16 sim-steps, mg_up_tr against the strip pipeline): 1 env -13.8 %, 256 envs (one per CU) -8.8 %, 512 -0.8 %,
  // 768 +5.4 %, 1024 +3.9 % -- the shorter latency chain pays while a CU has idle issue slots to spend on it
  if (P.upleg == 0) {
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    P.upleg = (any_tm && 2 * (long long)env->E < 5 * (long long)cus) ? 3 : 1;
  }
  if (P.upleg == 3 && !any_tm) P.upleg = 1;

  for (int l = 0, off = 0; l < P.nlev; ++l) {  // what bottom_level() needs per level
    P.lev[l].bot_off = 0;
    P.lev[l].bot_rcp = (65536u + (unsigned)P.lev[l].m - 1u) / (unsigned)P.lev[l].m;
    if (P.bottom_lb && l >= P.bottom_lb) {
      P.lev[l].bot_off = off;
      off += 32 * afc::prod::bottom_cells4(P.lev[l]);
    }
  }
  std::vector<float> brec;  // mg_bottom's table: per bottom level the planes CF {lw, le, ls, ln} | IR {inv, r = 0} | X = 0 | D = 0
  if (P.bottom_lb)
    for (int l = P.bottom_lb; l < P.nlev; ++l) {
      const int n = P.lev[l].n, m = P.lev[l].m, off = P.lev[l].coef_off;
      const int c4 = afc::prod::bottom_cells4(P.lev[l]);
      std::vector<float> lvl((size_t)8 * c4, 0.f);
      for (int k = 0; k < n * m; ++k) {
        const int i = k / m, j = k % m;
        if (i >= 1 && i <= n - 2 && j >= 1 && j <= m - 2) {
          lvl[4 * k + 0] = lx[off + k];
          lvl[4 * k + 1] = lx[off + k + m];
          lvl[4 * k + 2] = ly[off + k];
          lvl[4 * k + 3] = ly[off + k + 1];
        }
        lvl[(size_t)4 * c4 + 2 * k] = inv[off + k];
      }
      brec.insert(brec.end(), lvl.begin(), lvl.end());
    }
  brec.insert(brec.end(), 64, 0.f);
  if (!rc) rc = dev_upload(env, &P.bottom_rec, brec);
  lx.insert(lx.end(), 128, 0.f);  // tail padding: the strip Gauss-Seidel prefetches a few cells past a row
  ly.insert(ly.end(), 128, 0.f);
  inv.insert(inv.end(), 128, 1.f);
  if (!rc) rc = dev_upload(env, &P.mg_lx, lx);
  if (!rc) rc = dev_upload(env, &P.mg_ly, ly);
  if (!rc) rc = dev_upload(env, &P.mg_inv, inv);
  std::vector<afc::SampleDev> fs;
  for (const auto &s : sf.force_samples) fs.push_back(to_dev(s, sf.m));
  P.nseg = (int)fs.size();
  if (!rc) rc = dev_upload(env, &P.fs, fs);

  const size_t E = (size_t)env->E;
  // (qx, qy: the projected predictor field of the fused advection passes -- only envs that run them carry it, so that
  // the slab of every other env is what it was: the placement search works on the slab's size)
  std::vector<AFC_AS1 float **> fields = {&P.ux, &P.uy, &P.p, &P.u0x, &P.u0y, &P.usx, &P.usy, &P.fx, &P.fy, &P.r, &P.d};
#if AFC_DEVHOOKS
  const bool with_q = true;  // tools/ab_inproc.py --share-slab: instances with and without the fused passes share one layout
#else
  const bool with_q = (P.fuse_tail & 1) != 0;
#endif
  if (with_q) {
    fields.push_back(&P.qx);
    fields.push_back(&P.qy);
  }
  const size_t nfields = fields.size();
  {
    // The per-env state lives in ONE allocation (the eleven or thirteen fine fields back to back, then the coarse scratch): the
    // same library ran 7 % apart between env instances whose arrays came from separate hipMalloc calls (same box, same
    // process, launches interleaved: tools/ab_inproc.py) -- where the driver happens to place a dozen 140 MB buffers
    // decides how the streams of a phase fall on the memory channels / translation fragments.
    // + 64 floats per field: the row staging copies of advdif_v3 read up to 3 floats past the end of a row -- and ONE float
    // in front of row 0 (a left-hand tile's ring index 0 is "column -1"): kFront floats of padding precede the first field.
    // Development: AFC_ALLOC_MODE=0 restores separate allocations.
#if AFC_DEVHOOKS
    const char *am = getenv("AFC_ALLOC_MODE");
    const int mode = am ? atoi(am) : 1;
#else
    const int mode = 1;
#endif
    const size_t fsz = ((E * P.Ns + 64 + 63) / 64) * 64, csz = E * (size_t)P.Cs;
    constexpr size_t kFront = 64;
    if (mode == 0) {
      for (AFC_AS1 float **f : fields) {
        if (!rc) rc = dev_alloc(env, f, fsz + kFront);
        if (!rc && hipMemset(G(*f), 0, (fsz + kFront) * sizeof(float)) != hipSuccess) rc = fail(AFC_EHIP, "hipMemset");
        if (!rc) *f += kFront;
      }
      if (!rc) rc = dev_alloc(env, &P.coarse, csz);
      if (!rc && hipMemset(G(P.coarse), 0, csz * sizeof(float)) != hipSuccess) rc = fail(AFC_EHIP, "hipMemset");
    } else {
      float *slab = nullptr;
      const size_t scz = ((E * sizeof(EnvScalars) + 255) / 256) * 64;  // the per-env scalars, in floats, at the end of the slab
      const size_t total = kFront + nfields * fsz + csz + scz;
      // development (tools/ab_inproc.py --share-slab): an instance adopts the slab of another instance of the same shape,
      // so that builds are compared on the SAME memory (placements of the fast kind still differ by +-0.5 %)
      // (compiled in with -DAFC_DEVHOOKS=1 only -- tools/build_variant.sh; the shipped library takes no pointers from
      // its environment)
#if AFC_DEVHOOKS
      const char *sp = getenv("AFC_SLAB_PTR");
      if (sp && *sp) {
        slab = reinterpret_cast<float *>(strtoull(sp, nullptr, 0));
        env->slab_shared = true;
      }
#endif
      // PLACEMENT.  Where the driver puts a buffer of this size decides what it streams at: the same copy kernel runs
      // 4-5 % apart between 1.7 GB allocations of one process, the step kernel 6-7 % (bimodal, stable to 0.1 % per
      // allocation; translation misses are nil either way: tools/ab_inproc.py, tools/alloc_probe.py).  Large
      // batches therefore take the best of a few candidate allocations by a streaming probe (a copy of one half of
      // the buffer onto the other, ~0.5 ms each) and give the others back.  AFC_PLACEMENT_TRIALS=1 switches it off.
      // The kind of a placement follows the PHYSICAL region the driver hands out (tools/vmm_probe.hip: plain hipMalloc,
      // one hipMemCreate handle and ranges mapped from 2 / 64 / 256 MB chunks show the same fast / slow pattern by
      // position in the sequence of allocations, whatever the granularity) -- nothing user space can ask for, so the
      // search stays, BOUNDED: candidates are held at once (they must be different memory) but never more than a
      // quarter of the device's free memory at the time of the call, at most 32; a failed probe ends the search with
      // what has been measured so far.  afc_placement_info() reports what happened.
      const char *pt = getenv("AFC_PLACEMENT_TRIALS");
      int trials = pt ? atoi(pt) : 32;  // at most; the search stops early once a fast candidate stands out (below)
      if (slab) trials = 0;
      if (!slab && (total * sizeof(float) < ((size_t)256 << 20) || trials < 1)) trials = 1;
      if (trials > 32) trials = 32;
      if (trials > 1) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
          (void)hipGetLastError();
          trials = 1;
        } else {
          const size_t cap = (fr / 4) / (total * sizeof(float));  // candidates that fit a quarter of the free memory
          if ((size_t)trials > cap) trials = (int)(cap < 1 ? 1 : cap);
        }
      }
      env->place_trials = 0;
      env->place_best = env->place_median = 0.f;
      if (!rc && trials > 1) {
        std::vector<float *> cand;
        std::vector<float> rate;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        bool probe_ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
        const size_t n4 = total / 8;  // 16-byte pieces of one half
        auto probe = [&](void *q, int reps) {
          float best = 1e30f;
          for (int r = 0; r < reps && probe_ok; ++r) {
            probe_ok = hipEventRecord(e0, nullptr) == hipSuccess;
            hipLaunchKernelGGL(k_copy_f4<false>, dim3(2048), dim3(256), 0, nullptr, (const f4v *)q, (f4v *)q + n4, n4);
            probe_ok = probe_ok && hipGetLastError() == hipSuccess && hipEventRecord(e1, nullptr) == hipSuccess &&
                       hipEventSynchronize(e1) == hipSuccess;
            float ms = 0;
            probe_ok = probe_ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f;
            if (probe_ok && r > 0 && ms < best) best = ms;
          }
          return probe_ok ? (float)(2.0 * 16.0 * (double)n4 / (best * 1e-3) / 1e9) : 0.f;
        };
        for (int t = 0; t < trials && probe_ok; ++t) {
          void *q = nullptr;
          if (hipMalloc(&q, total * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            break;
          }
          cand.push_back(static_cast<float *>(q));
          if (hipMemsetAsync(q, 0, total * sizeof(float), nullptr) != hipSuccess) probe_ok = false;
          if (t == 0) (void)probe(q, 8);  // the first readings of a process come from a device that is still waking up
          rate.push_back(probe(q, 4));
          if (!probe_ok) break;  // (the candidate stays in the list with rate 0: it is a usable allocation)
          // enough: at least SIXTEEN candidates seen and the best stands 3.5 % above their MEDIAN (the two kinds of placement
          // are 4-5 % apart in this probe; the median is of the slow kind, four in five are).  (Six was too few: the slow
          // kind alone spreads over 3 % in this probe, and four of twenty-one development runs stopped early on the best
          // of the slow ones; a candidate costs ~3 ms.)  The number of candidates was capped above at a quarter of the
          // free memory AT THAT TIME; somebody else (a learner, another rank's afc_create) may be allocating on the same
          // device meanwhile, so every fourth candidate looks again and stops once less is free than the candidates
          // already hold -- afc_create is still best not raced with other large allocations on its device.
          std::vector<float> srt(rate);
          std::sort(srt.begin(), srt.end());
          const float med = srt[srt.size() / 2], hi = srt.back();
          env->place_best = hi;
          env->place_median = med;
          if (cand.size() >= 16 && hi >= 1.035f * med) break;
          if ((cand.size() & 3) == 0) {
            size_t fr2 = 0, tot2 = 0;
            if (hipMemGetInfo(&fr2, &tot2) != hipSuccess) {
              (void)hipGetLastError();
              break;
            }
            if (fr2 < cand.size() * total * sizeof(float)) break;
          }
        }
        env->place_trials = (int)cand.size();
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipGetLastError();
        int pick = -1;
        for (size_t t = 0; t < cand.size(); ++t)
          if (pick < 0 || rate[t] > rate[pick]) pick = (int)t;
        for (size_t t = 0; t < cand.size(); ++t) {
          if (getenv("AFC_PRINT_ALLOC"))
            fprintf(stderr, "afc_create: candidate %zu at %p streams at %.0f GB/s%s\n", t, (void *)cand[t], rate[t], (int)t == pick ? "  <- kept" : "");
          if ((int)t != pick) (void)hipFree(cand[t]);
        }
        if (pick >= 0) {
          slab = cand[pick];
          env->allocs.push_back(slab);
          if (hipMemset(slab, 0, total * sizeof(float)) != hipSuccess) rc = fail(AFC_EHIP, "hipMemset");
        }
      }
      if (!rc && !slab) {
        rc = dev_alloc(env, &slab, total);
        if (!rc && hipMemset(slab, 0, total * sizeof(float)) != hipSuccess) rc = fail(AFC_EHIP, "hipMemset");
      }
      if (!rc) {
        if (getenv("AFC_PRINT_ALLOC")) fprintf(stderr, "afc_create: state slab %p, %zu bytes\n", (void *)slab, total * sizeof(float));
#if AFC_DEVHOOKS
        if (getenv("AFC_EXPORT_SLAB")) {  // development: tools/ab_inproc.py --share-slab reads the address back
          char buf[32];
          snprintf(buf, sizeof buf, "%p", (void *)slab);
          setenv("AFC_LAST_SLAB", buf, 1);
        }
#endif
        P.sc = (AFC_AS1 EnvScalars *)(slab + kFront + nfields * fsz + csz);
        size_t k = 0;
        for (AFC_AS1 float **f : fields) *f = (AFC_AS1 float *)(slab + kFront + (k++) * fsz);
        P.coarse = (AFC_AS1 float *)(slab + kFront + nfields * fsz);
      }
    }
  }
  if (!rc && !P.sc) {  // (separate allocations: AFC_ALLOC_MODE=0)
    rc = dev_alloc(env, &P.sc, E);
    if (!rc && hipMemset(G(P.sc), 0, E * sizeof(EnvScalars)) != hipSuccess) rc = fail(AFC_EHIP, "hipMemset");
  }
  if (!rc) rc = dev_alloc(env, &env->rs_u, (size_t)P.Ns);
  if (!rc) rc = dev_alloc(env, &env->rs_v, (size_t)P.Ns);
  if (!rc) rc = dev_alloc(env, &env->rs_p, (size_t)P.Ns);
  if (!rc) rc = dev_alloc(env, &env->d_in, E * 2);
  if (!rc) rc = dev_alloc(env, &env->d_f2, E * 2);
  if (!rc) rc = dev_alloc(env, &env->d_f1, E);
  if (!rc) rc = dev_alloc(env, &env->d_u8, E);
  if (!rc) rc = dev_alloc(env, &env->d_i32, E * 2);
  if (!rc) rc = dev_alloc(env, &env->d_P, (size_t)1);
  if (!rc) rc = dev_alloc(env, &P.gate, (size_t)256);
  if (!rc && hipMemset(G(P.gate), 0, 256 * sizeof(int)) != hipSuccess) rc = fail(AFC_EHIP, "hipMemset");
  P.gate_limit = (cfg->reserved[0] >> 16) & 15;  // dbg bits 16-19: 1..14 per-CU tokens (experiment), 15 = populations
  P.gate_n[0] = P.gate_n[1] = 0;
  for (int b = 0; b < (int)E; ++b) P.gate_n[(b >> 3) & 1]++;
  if (P.gate_limit == 15) {  // taking turns needs every workgroup of a launch resident at once (a waiting workgroup
                             // would otherwise hold the slot of the one it waits for) and two non-empty populations
    int per_cu = 0, cus = 0;
    hipDeviceProp_t prop;
    bool ok = (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess);
    if (ok) cus = prop.multiProcessorCount;
    ok = ok && env->cpl >= 1 && afc::occupancy_agent_step_diag(env->cpl, env->lds_bytes, &per_cu) == 0;
    if (!ok || (long long)per_cu * cus < (long long)E || P.gate_n[0] == 0 || P.gate_n[1] == 0) P.gate_limit = 0;
  }
  if (!rc) rc = dev_alloc(env, &env->d_stamps, (size_t)32);
  if (!rc && hipMemset(env->d_stamps, 0, 32 * sizeof(long long)) != hipSuccess) rc = fail(AFC_EHIP, "hipMemset");
  if (rc) {
    afc_destroy(env);
    return rc;
  }
  // free-stream reset state: new VectorField(n+2,m+2,1,0), p = 0  (BDIM.pde:70-71,54)
  std::vector<float> ones(P.N, 1.f), zeros(P.N, 0.f);
  *out = env;
  rc = afc_set_reset_state(env, ones.data(), zeros.data(), zeros.data());
  if (rc) {
    afc_destroy(env);
    *out = nullptr;
    return rc;
  }
  return AFC_OK;
}

int afc_destroy(afc_env *env) {
  if (!env) return AFC_OK;
  (void)hipSetDevice(env->cfg.device);
  (void)hipDeviceSynchronize();
  for (void *p : env->allocs) (void)hipFree(p);
  delete env;
  return AFC_OK;
}

int afc_dims(const afc_env *env, int32_t *n, int32_t *m) {
  if (check_env(env)) return AFC_EINVAL;
  if (n) *n = env->P.n;
  if (m) *m = env->P.m;
  return AFC_OK;
}
int afc_num_envs(const afc_env *env) { return env ? env->E : AFC_EINVAL; }
int afc_num_levels(const afc_env *env) { return env ? env->P.nlev : AFC_EINVAL; }
float afc_dt(const afc_env *env) { return env ? env->P.dt : 0.f; }
int afc_level_dims(const afc_env *env, int level, int32_t *n, int32_t *m) {
  if (check_env(env)) return AFC_EINVAL;
  if (level < 0 || level >= env->P.nlev) return fail(AFC_EINVAL, "afc_level_dims: bad level");
  if (n) *n = env->P.lev[level].n;
  if (m) *m = env->P.lev[level].m;
  return AFC_OK;
}

int afc_set_io_space(afc_env *env, int space) {
  if (check_env(env)) return AFC_EINVAL;
  if (space != AFC_MEM_HOST && space != AFC_MEM_DEVICE) return fail(AFC_EINVAL, "afc_set_io_space: bad space");
  env->io_space = space;
  return AFC_OK;
}
int afc_set_stream(afc_env *env, void *hip_stream) {
  if (check_env(env)) return AFC_EINVAL;
  env->stream = static_cast<hipStream_t>(hip_stream);
  return AFC_OK;
}
int afc_synchronize(afc_env *env) {
  if (check_env(env)) return AFC_EINVAL;
  HIP_TRY(hipSetDevice(env->cfg.device));
  HIP_TRY(hipStreamSynchronize(env->stream));
  return AFC_OK;
}

int afc_set_reset_state(afc_env *env, const float *u, const float *v, const float *p) {
  if (check_env(env)) return AFC_EINVAL;
  if (!u || !v || !p) return fail(AFC_EINVAL, "afc_set_reset_state: null array");
  HIP_TRY(hipSetDevice(env->cfg.device));
  const size_t bytes = (size_t)env->P.N * sizeof(float);
  HIP_TRY(hipMemcpy(env->rs_u, u, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(env->rs_v, v, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(env->rs_p, p, bytes, hipMemcpyHostToDevice));
  // a fresh AFCCylinder: callLearn etc. are initialised like the sketch globals
  hipLaunchKernelGGL(k_reset, dim3(env->E), dim3(256), 0, env->stream, env->P, (const uint8_t *)nullptr, env->rs_u,
                     env->rs_v, env->rs_p, 0);
  int rc = launch_check("k_reset");
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(env->stream));
  return AFC_OK;
}

int afc_load_state_bdim(afc_env *env, const char *path) {  // BDIM.resume, BDIM.pde:239-251
  if (check_env(env)) return AFC_EINVAL;
  if (!path) return fail(AFC_EINVAL, "afc_load_state_bdim: null path");
  FILE *f = fopen(path, "r");
  if (!f) return fail(AFC_EIO, std::string("afc_load_state_bdim: cannot open ") + path);
  const int n = env->P.n, m = env->P.m;
  std::vector<float> u((size_t)n * m), v((size_t)n * m), p((size_t)n * m);
  char line[512];
  bool ok = fgets(line, sizeof line, f) != nullptr;
  if (ok) env->reset_flow_t = strtof(line, nullptr);
  ok = ok && fgets(line, sizeof line, f) != nullptr;
  if (ok) env->reset_flow_dt = strtof(line, nullptr);
  for (size_t k = 0; ok && k < (size_t)n * m; ++k) {
    if (!fgets(line, sizeof line, f)) {
      ok = false;
      break;
    }
    char *q = line;  // Java: float(split(line, ','))
    u[k] = strtof(q, &q);
    while (*q == ',' || *q == ' ') ++q;
    v[k] = strtof(q, &q);
    while (*q == ',' || *q == ' ') ++q;
    p[k] = strtof(q, &q);
  }
  fclose(f);
  if (!ok) return fail(AFC_EIO, std::string("afc_load_state_bdim: truncated or malformed file ") + path);
  return afc_set_reset_state(env, u.data(), v.data(), p.data());
}

// Java's Float.toString / String.valueOf(float) (what `""+u.x.a[i][j]` in BDIM.write and clientCFD.pde:120 produce):
// the shortest decimal that uniquely identifies the float32, at least one digit after the point, plain notation
// for 1e-3 <= |x| < 1e7 and d.dddE[-]n otherwise.
int afc_format_float_java(float x, char *buf, int cap) {
  if (!buf || cap < 16) return fail(AFC_EINVAL, "afc_format_float_java: buffer too small (need >= 16)");
  if (std::isnan(x)) return snprintf(buf, cap, "NaN"), AFC_OK;
  if (std::isinf(x)) return snprintf(buf, cap, x > 0 ? "Infinity" : "-Infinity"), AFC_OK;
  if (x == 0.f) return snprintf(buf, cap, std::signbit(x) ? "-0.0" : "0.0"), AFC_OK;
  // shortest digit string that parses back to x; among the candidates of that length the one nearest to x
  char best[32] = "";
  for (int p = 1; p <= 9 && !best[0]; ++p) {
    char t[32];
    snprintf(t, sizeof t, "%.*e", p - 1, (double)x);
    // candidates: the correctly rounded p-digit decimal and its two neighbours in the last place
    char mant[16];
    int nd = 0;
    for (const char *q = t; *q && *q != 'e'; ++q)
      if (*q >= '0' && *q <= '9') mant[nd++] = *q;
    mant[nd] = 0;
    const int ex = atoi(strchr(t, 'e') + 1);
    const long long base = atoll(mant);
    double bestd = 1e300;
    for (int dlt = -1; dlt <= 1; ++dlt) {
      long long v = base + dlt;
      int e2 = ex;
      long long lim = 1;
      for (int q = 0; q < p; ++q) lim *= 10;
      if (v >= lim) {  // 9.99 + 0.01 -> 1.00e+1: one digit too many, drop the trailing zero
        v /= 10;
        ++e2;
      }
      if (v < lim / 10 && p > 1) continue;  // lost a leading digit: not a p-digit candidate
      if (v <= 0) continue;
      char c[40];
      snprintf(c, sizeof c, "%s%llde%d", x < 0 ? "-" : "", v, e2 - (p - 1));
      if (strtof(c, nullptr) != x) continue;
      const double dd = fabs(strtod(c, nullptr) - (double)x);
      // a tie between two candidates goes to the even last digit (FloatingDecimal.dtoa's "it's a tie" branch)
      if (dd < bestd || (dd == bestd && (v & 1) == 0)) {
        bestd = dd;
        snprintf(best, sizeof best, "%lld %d", v, e2);
      }
    }
  }
  long long dig = 0;
  int ex = 0;
  sscanf(best, "%lld %d", &dig, &ex);
  char ds[24];
  snprintf(ds, sizeof ds, "%lld", dig);
  int nd = (int)strlen(ds);
  while (nd > 1 && ds[nd - 1] == '0') ds[--nd] = 0;  // trailing zeros carry no information
  std::string out = (x < 0) ? "-" : "";
  const double ax = fabs((double)x);
  if (ax >= 1e-3 && ax < 1e7) {
    if (ex >= 0) {
      std::string ip(ds, ds + (nd < ex + 1 ? nd : ex + 1));
      ip.append((size_t)(ex + 1 - (int)ip.size()), '0');
      std::string fp = (nd > ex + 1) ? std::string(ds + ex + 1) : std::string("0");
      out += ip + "." + fp;
    } else {
      out += "0." + std::string((size_t)(-ex - 1), '0') + ds;
    }
  } else {
    out += std::string(1, ds[0]) + "." + (nd > 1 ? std::string(ds + 1) : std::string("0")) + "E" + std::to_string(ex);
  }
  if ((int)out.size() + 1 > cap) return fail(AFC_EINVAL, "afc_format_float_java: buffer too small");
  memcpy(buf, out.c_str(), out.size() + 1);
  return AFC_OK;
}

int afc_save_state_bdim(afc_env *env, int env_idx, const char *path) {  // BDIM.write, BDIM.pde:226-237
  if (check_env(env)) return AFC_EINVAL;
  if (!path || env_idx < 0 || env_idx >= env->E) return fail(AFC_EINVAL, "afc_save_state_bdim: bad argument");
  const int N = env->P.N;
  std::vector<float> u(N), v(N), p(N);
  int rc = afc_get_field(env, env_idx, AFC_FIELD_U, u.data());
  if (!rc) rc = afc_get_field(env, env_idx, AFC_FIELD_V, v.data());
  if (!rc) rc = afc_get_field(env, env_idx, AFC_FIELD_P, p.data());
  if (rc) return rc;
  EnvScalars s;
  HIP_TRY(hipMemcpy(&s, G(env->P.sc) + env_idx, sizeof(s), hipMemcpyDeviceToHost));
  FILE *f = fopen(path, "w");
  if (!f) return fail(AFC_EIO, std::string("afc_save_state_bdim: cannot open ") + path);
  // BDIM.t advances by dt per step (BDIM.pde:106), accumulated in float like the reference's `t += dt`
  float flow_t = env->reset_flow_t;
  for (int q = 0; q < s.sim_steps; ++q) flow_t += env->P.dt;
  // output.println(t); println(dt); println(""+u+", "+v+", "+p): Java float text (BDIM.pde:229-233)
  char a[32], b[32], c[32];
  afc_format_float_java(flow_t, a, sizeof a);
  afc_format_float_java(env->P.dt, b, sizeof b);
  fprintf(f, "%s\n%s\n", a, b);
  for (int k = 0; k < N; ++k) {
    afc_format_float_java(u[k], a, sizeof a);
    afc_format_float_java(v[k], b, sizeof b);
    afc_format_float_java(p[k], c, sizeof c);
    fprintf(f, "%s, %s, %s\n", a, b, c);
  }
  fclose(f);
  return AFC_OK;
}

int afc_reset(afc_env *env, const uint8_t *mask) {
  if (check_env(env)) return AFC_EINVAL;
  HIP_TRY(hipSetDevice(env->cfg.device));
  const uint8_t *dmask = mask;
  if (mask && env->io_space == AFC_MEM_HOST) {
    HIP_TRY(hipMemcpyAsync(env->d_u8, mask, env->E, hipMemcpyHostToDevice, env->stream));
    dmask = env->d_u8;
  }
  hipLaunchKernelGGL(k_reset, dim3(env->E), dim3(256), 0, env->stream, env->P, dmask, env->rs_u, env->rs_v,
                     env->rs_p, env->cfg.carry_quirk);
  int rc = launch_check("k_reset");
  if (rc) return rc;
  if (env->io_space == AFC_MEM_HOST) HIP_TRY(hipStreamSynchronize(env->stream));
  return AFC_OK;
}

int afc_sim_step(afc_env *env, const float *xi_m, float *force) {
  if (check_env(env)) return AFC_EINVAL;
  if (!xi_m) return fail(AFC_EINVAL, "afc_sim_step: null xi_m");
  HIP_TRY(hipSetDevice(env->cfg.device));
  const size_t E = env->E;
  const float *dxi = xi_m;
  float *dforce = force;
  if (env->io_space == AFC_MEM_HOST) {
    HIP_TRY(hipMemcpyAsync(env->d_in, xi_m, E * 2 * sizeof(float), hipMemcpyHostToDevice, env->stream));
    dxi = env->d_in;
    dforce = env->d_f2;
  }
  if (int rcp = sync_params(env)) return rcp;
  if (env->P.gate_limit == 15) HIP_TRY(hipMemsetAsync(G(env->P.gate), 0, 64 * sizeof(int), env->stream));
  switch (env->cpl == 0 ? 1 : launch_mode(env)) {  // the reference kernels (cpl 0) live in the diagnostic unit
    case 1: afc::launch_sim_step_diag(env->cpl, env->E, env->lds_bytes, env->stream, env->d_P, dxi, dforce); break;
    case 2: afc::launch_sim_step_perf(env->cpl, env->E, env->lds_bytes_step, env->stream, env->d_P, dxi, dforce); break;
    case 3: afc::launch_sim_step_lat(env->cpl, env->E, env->lds_bytes_lat, env->stream, env->d_P, dxi, dforce); break;
    default: afc::launch_sim_step_prod(env->cpl, env->E, env->lds_bytes_step, env->stream, env->d_P, dxi, dforce);
  }
  int rc = launch_check("k_sim_step");
  if (rc) return rc;
  if (env->io_space == AFC_MEM_HOST) {
    if (force)
      HIP_TRY(hipMemcpyAsync(force, env->d_f2, E * 2 * sizeof(float), hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(hipStreamSynchronize(env->stream));
  }
  return AFC_OK;
}

int afc_agent_step(afc_env *env, const float *action, float *obs, float *reward, uint8_t *done,
                   int32_t *nsteps_out) {
  if (check_env(env)) return AFC_EINVAL;
  if (!action || !obs || !reward || !done) return fail(AFC_EINVAL, "afc_agent_step: null argument");
  HIP_TRY(hipSetDevice(env->cfg.device));
  const size_t E = env->E;
  const bool host = env->io_space == AFC_MEM_HOST;
  const float *dact = action;
  float *dobs = obs, *drew = reward;
  uint8_t *ddone = done;
  int32_t *dns = nsteps_out;
  if (host) {
    HIP_TRY(hipMemcpyAsync(env->d_in, action, E * 2 * sizeof(float), hipMemcpyHostToDevice, env->stream));
    dact = env->d_in;
    dobs = env->d_f2;
    drew = env->d_f1;
    ddone = env->d_u8;
    dns = env->d_i32;
  }
  // upper bound on sim-steps per call: warm-up (t <= init_time) + one averaging window
  const float per = env->P.dt / env->cfg.resolution;
  const int warm = (int)(env->cfg.init_time / per);
  const int max_steps = (warm > 0 ? warm : 0) + 2 * env->cfg.substeps + 8;
  if (int rcp = sync_params(env)) return rcp;
  if (env->P.gate_limit == 15) HIP_TRY(hipMemsetAsync(G(env->P.gate), 0, 64 * sizeof(int), env->stream));
  switch (env->cpl == 0 ? 1 : launch_mode(env)) {
    case 1: afc::launch_agent_step_diag(env->cpl, env->E, env->lds_bytes, env->stream, env->d_P, dact, dobs, drew, ddone, dns, max_steps); break;
    case 2: afc::launch_agent_step_perf(env->cpl, env->E, env->lds_bytes_step, env->stream, env->d_P, dact, dobs, drew, ddone, dns, max_steps); break;
    case 3: afc::launch_agent_step_lat(env->cpl, env->E, env->lds_bytes_lat, env->stream, env->d_P, dact, dobs, drew, ddone, dns, max_steps); break;
    default: afc::launch_agent_step_prod(env->cpl, env->E, env->lds_bytes_step, env->stream, env->d_P, dact, dobs, drew, ddone, dns, max_steps);
  }
  int rc = launch_check("k_agent_step");
  if (rc) return rc;
  if (host) {
    HIP_TRY(hipMemcpyAsync(obs, dobs, E * 2 * sizeof(float), hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(hipMemcpyAsync(reward, drew, E * sizeof(float), hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(hipMemcpyAsync(done, ddone, E, hipMemcpyDeviceToHost, env->stream));
    if (nsteps_out)
      HIP_TRY(hipMemcpyAsync(nsteps_out, dns, E * sizeof(int32_t), hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(hipStreamSynchronize(env->stream));
  }
  return AFC_OK;
}

static int fetch_scalars(afc_env *env, std::vector<EnvScalars> &h) {
  HIP_TRY(hipSetDevice(env->cfg.device));
  HIP_TRY(hipStreamSynchronize(env->stream));
  h.resize(env->E);
  HIP_TRY(hipMemcpy(h.data(), G(env->P.sc), env->E * sizeof(EnvScalars), hipMemcpyDeviceToHost));
  return AFC_OK;
}

int afc_get_time(afc_env *env, float *t) {
  if (check_env(env) || !t) return fail(AFC_EINVAL, "afc_get_time: bad argument");
  std::vector<EnvScalars> h;
  int rc = fetch_scalars(env, h);
  if (rc) return rc;
  for (int e = 0; e < env->E; ++e) t[e] = h[e].t;
  return AFC_OK;
}
int afc_get_solver_stats(afc_env *env, int32_t *iters, float *resid) {
  if (check_env(env)) return AFC_EINVAL;
  std::vector<EnvScalars> h;
  int rc = fetch_scalars(env, h);
  if (rc) return rc;
  for (int e = 0; e < env->E; ++e) {
    if (iters) {
      iters[2 * e] = h[e].iters[0];
      iters[2 * e + 1] = h[e].iters[1];
    }
    if (resid) {
      resid[2 * e] = h[e].resid[0];
      resid[2 * e + 1] = h[e].resid[1];
    }
  }
  return AFC_OK;
}
int afc_get_diverged(afc_env *env, uint8_t *flags) {
  if (check_env(env) || !flags) return fail(AFC_EINVAL, "afc_get_diverged: bad argument");
  std::vector<EnvScalars> h;
  int rc = fetch_scalars(env, h);
  if (rc) return rc;
  for (int e = 0; e < env->E; ++e) flags[e] = h[e].diverged ? 1 : 0;
  return AFC_OK;
}

int afc_get_field(afc_env *env, int env_idx, int which, float *out) {
  if (check_env(env)) return AFC_EINVAL;
  if (!out || env_idx < 0 || env_idx >= env->E) return fail(AFC_EINVAL, "afc_get_field: bad argument");
  HIP_TRY(hipSetDevice(env->cfg.device));
  HIP_TRY(hipStreamSynchronize(env->stream));
  const StepParams &P = env->P;
  const size_t eo = (size_t)env_idx * P.Ns, bytes = (size_t)P.N * sizeof(float);
  const float *src = nullptr;
  switch (which) {
    case AFC_FIELD_U: src = G(P.ux) + eo; break;
    case AFC_FIELD_V: src = G(P.uy) + eo; break;
    case AFC_FIELD_P: src = G(P.p) + eo; break;
    case AFC_FIELD_VORT: {
      float *tmp = G(P.fx) + eo;  // fx is scratch between steps
      hipLaunchKernelGGL(k_vorticity, dim3(64), dim3(256), 0, env->stream, P, env_idx, tmp);
      int rc = launch_check("k_vorticity");
      if (rc) return rc;
      HIP_TRY(hipStreamSynchronize(env->stream));
      src = tmp;
      break;
    }
    // development only (tools/path_diff.py): raw work buffers of the pressure solve, valid after a launch that left the
    // step early (dbg stop codes): 16 = second pressure buffer, 17 = finest residual, 18 = coarse scratch (first N floats)
    case 16: src = G(P.d) + eo; break;
    case 17: src = G(P.r) + eo; break;
    case 18: src = G(P.coarse) + (size_t)env_idx * P.Cs; break;
    case 19: src = G(P.usx) + eo; break;
    case 20: src = G(P.usy) + eo; break;
    default: return fail(AFC_EINVAL, "afc_get_field: unknown field");
  }
  HIP_TRY(hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
  if (which == AFC_FIELD_P) {  // apply the pending mean shift (p.plusEq(-mean), VectorField.pde:136)
    EnvScalars s;
    HIP_TRY(hipMemcpy(&s, G(P.sc) + env_idx, sizeof(s), hipMemcpyDeviceToHost));
    if (s.p_shift != 0.f)
      for (int k = 0; k < P.N; ++k) out[k] = out[k] + s.p_shift;
  }
  return AFC_OK;
}
int afc_set_field(afc_env *env, int env_idx, int which, const float *in) {
  if (check_env(env)) return AFC_EINVAL;
  if (!in || env_idx < 0 || env_idx >= env->E) return fail(AFC_EINVAL, "afc_set_field: bad argument");
  HIP_TRY(hipSetDevice(env->cfg.device));
  HIP_TRY(hipStreamSynchronize(env->stream));
  const StepParams &P = env->P;
  const size_t eo = (size_t)env_idx * P.Ns;
  float *dst = nullptr;
  switch (which) {
    case AFC_FIELD_U: dst = G(P.ux) + eo; break;
    case AFC_FIELD_V: dst = G(P.uy) + eo; break;
    case AFC_FIELD_P: dst = G(P.p) + eo; break;
    default: return fail(AFC_EINVAL, "afc_set_field: field is read-only or unknown");
  }
  HIP_TRY(hipMemcpy(dst, in, (size_t)P.N * sizeof(float), hipMemcpyHostToDevice));
  if (which == AFC_FIELD_P) {
    const float zero = 0.f;
    HIP_TRY(hipMemcpy(reinterpret_cast<char *>(G(P.sc) + env_idx) + offsetof(EnvScalars, p_shift), &zero, sizeof(float),
                      hipMemcpyHostToDevice));
  }
  return AFC_OK;
}

int afc_get_probes(afc_env *env, int num, float *out) {
  if (check_env(env)) return AFC_EINVAL;
  if (num < 1 || num > 4096 || !out) return fail(AFC_EINVAL, "afc_get_probes: bad argument");
  HIP_TRY(hipSetDevice(env->cfg.device));
  if (env->n_probes != num) {
    afc::SampleDev *d = nullptr;
    for (const auto &t : env->probe_tables)
      if (t.first == num) d = t.second;
    if (!d) {
      std::vector<afc::SampleDev> ps;
      for (const auto &s : afc::build_probes(env->cp, num)) ps.push_back(to_dev(s, env->P.m));
      int rc = dev_alloc(env, &d, ps.size());
      if (rc) return rc;
      HIP_TRY(hipMemcpy(d, ps.data(), ps.size() * sizeof(afc::SampleDev), hipMemcpyHostToDevice));
      env->probe_tables.emplace_back(num, d);
    }
    env->d_probes = d;
    env->n_probes = num;
  }
  const size_t need = (size_t)env->E * num;
  if (need > env->probe_out_cap) {  // persistent output staging, grown when a larger `num` is asked for
    float *q = nullptr;
    int rc = dev_alloc(env, &q, need);
    if (rc) return rc;
    env->d_probe_out = q;
    env->probe_out_cap = need;
  }
  hipLaunchKernelGGL(k_probes, dim3(env->E), dim3(64), 0, env->stream, env->P, env->d_probes, num, env->d_probe_out);
  int rc = launch_check("k_probes");
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, env->d_probe_out, need * sizeof(float), hipMemcpyDeviceToHost, env->stream));
  HIP_TRY(hipStreamSynchronize(env->stream));
  return AFC_OK;
}

int afc_get_cfl(afc_env *env, float *out) {
  if (check_env(env)) return AFC_EINVAL;
  if (!out) return fail(AFC_EINVAL, "afc_get_cfl: null out");
  HIP_TRY(hipSetDevice(env->cfg.device));
  if (!env->d_cfl) {
    int rc = dev_alloc(env, &env->d_cfl, (size_t)env->E);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_cfl, dim3(env->E), dim3(256), 0, env->stream, env->P, env->d_cfl);
  int rc = launch_check("k_cfl");
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, env->d_cfl, (size_t)env->E * sizeof(float), hipMemcpyDeviceToHost, env->stream));
  HIP_TRY(hipStreamSynchronize(env->stream));
  return AFC_OK;
}

int afc_get_window_complete(afc_env *env, uint8_t *flags) {
  if (check_env(env) || !flags) return fail(AFC_EINVAL, "afc_get_window_complete: bad argument");
  std::vector<EnvScalars> h;
  int rc = fetch_scalars(env, h);
  if (rc) return rc;
  for (int e = 0; e < env->E; ++e) flags[e] = h[e].window_done ? 1 : 0;
  return AFC_OK;
}

int afc_get_mg_histogram(afc_env *env, int64_t *hist8, int clear) {
  if (check_env(env) || !hist8) return fail(AFC_EINVAL, "afc_get_mg_histogram: bad argument");
  std::vector<EnvScalars> h;
  int rc = fetch_scalars(env, h);
  if (rc) return rc;
  for (int q = 0; q < 8; ++q) hist8[q] = 0;
  for (int e = 0; e < env->E; ++e)
    for (int q = 0; q < 8; ++q) hist8[q] += h[e].it_hist[q];
  if (clear) {
    for (auto &s : h)
      for (int q = 0; q < 8; ++q) s.it_hist[q] = 0;
    HIP_TRY(hipMemcpy(G(env->P.sc), h.data(), env->E * sizeof(EnvScalars), hipMemcpyHostToDevice));
  }
  return AFC_OK;
}

int afc_get_static(afc_env *env, int which, int level, float *out) {
  if (check_env(env)) return AFC_EINVAL;
  if (!out) return fail(AFC_EINVAL, "afc_get_static: null out");
  const afc::StaticFields &sf = env->sf;
  if (level < 0 || level >= (int)sf.levels.size()) return fail(AFC_EINVAL, "afc_get_static: bad level");
  const std::vector<float> *src = nullptr;
  if (level > 0) {
    if (which == AFC_STATIC_CX) src = &sf.levels[level].lx;
    else if (which == AFC_STATIC_CY) src = &sf.levels[level].ly;
    else return fail(AFC_EINVAL, "afc_get_static: only CX/CY exist on coarse levels");
  } else {
    switch (which) {
      case AFC_STATIC_DELX: src = &sf.delx; break;
      case AFC_STATIC_DELY: src = &sf.dely; break;
      case AFC_STATIC_DEL1X: src = &sf.del1x; break;
      case AFC_STATIC_DEL1Y: src = &sf.del1y; break;
      case AFC_STATIC_WNXX: src = &sf.wnxx; break;
      case AFC_STATIC_WNXY: src = &sf.wnxy; break;
      case AFC_STATIC_WNYX: src = &sf.wnyx; break;
      case AFC_STATIC_WNYY: src = &sf.wnyy; break;
      case AFC_STATIC_CX: src = &sf.cx; break;
      case AFC_STATIC_CY: src = &sf.cy; break;
      case AFC_STATIC_W1X: src = &sf.w1x; break;
      case AFC_STATIC_W2X: src = &sf.w2x; break;
      case AFC_STATIC_W1Y: src = &sf.w1y; break;
      case AFC_STATIC_W2Y: src = &sf.w2y; break;
      default: return fail(AFC_EINVAL, "afc_get_static: unknown array");
    }
  }
  memcpy(out, src->data(), src->size() * sizeof(float));
  return AFC_OK;
}

int afc_profile_phases(afc_env *env, int enable, int64_t *ticks_out, int count) {
  if (check_env(env)) return AFC_EINVAL;
  if (env->P.gs_mode == 1 && enable) return fail(AFC_EINVAL, "afc_profile_phases: not available in perf mode (gs_mode 1): its kernels carry no phase stamps");
  HIP_TRY(hipSetDevice(env->cfg.device));
  HIP_TRY(hipStreamSynchronize(env->stream));
  if (ticks_out && count > 0) {
    if (count > 32) count = 32;
    HIP_TRY(hipMemcpy(ticks_out, env->d_stamps, count * sizeof(long long), hipMemcpyDeviceToHost));
  }
  HIP_TRY(hipMemset(env->d_stamps, 0, 32 * sizeof(long long)));
  env->P.stamps = (AFC_AS1 long long *)(enable ? env->d_stamps : nullptr);
  env->p_dirty = true;
  return AFC_OK;
}

// ---- unit-level operators (env 0's buffers are used as scratch: state of env 0 is clobbered) ----
int afc_op_advdif(afc_env *env, float *fx, float *fy, const float *u0x, const float *u0y) {
  if (check_env(env)) return AFC_EINVAL;
  if (!fx || !fy || !u0x || !u0y) return fail(AFC_EINVAL, "afc_op_advdif: null argument");
  HIP_TRY(hipSetDevice(env->cfg.device));
  HIP_TRY(hipStreamSynchronize(env->stream));
  const StepParams &P = env->P;
  const size_t bytes = (size_t)P.N * sizeof(float);
  HIP_TRY(hipMemcpy(G(P.ux), fx, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(G(P.uy), fy, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(G(P.u0x), u0x, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(G(P.u0y), u0y, bytes, hipMemcpyHostToDevice));
  // u0 passed as the very arrays of F: BDIM.update's u0 = u (BDIM.pde:82-84), the predictor's form of the pass
  const int same = (u0x == fx && u0y == fy) ? 1 : 0;
  if (need_diag(env) || P.impl == 0)
    afc::launch_op_advdif_diag(env->lds_bytes, env->stream, P, same);
  else
    afc::launch_op_advdif_prod(env->lds_bytes, env->stream, P, same);
  int rc = launch_check("k_op_advdif");
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(env->stream));
  HIP_TRY(hipMemcpy(fx, G(P.fx), bytes, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(fy, G(P.fy), bytes, hipMemcpyDeviceToHost));
  return AFC_OK;
}

int afc_op_setbc_u(afc_env *env, float *ux, float *uy) {
  if (check_env(env)) return AFC_EINVAL;
  if (!ux || !uy) return fail(AFC_EINVAL, "afc_op_setbc_u: null argument");
  HIP_TRY(hipSetDevice(env->cfg.device));
  HIP_TRY(hipStreamSynchronize(env->stream));
  const StepParams &P = env->P;
  const size_t bytes = (size_t)P.N * sizeof(float);
  HIP_TRY(hipMemcpy(G(P.ux), ux, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(G(P.uy), uy, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_op_setbc, dim3(1), dim3(kThreads), 0, env->stream, P);
  int rc = launch_check("k_op_setbc");
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(env->stream));
  HIP_TRY(hipMemcpy(ux, G(P.ux), bytes, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(uy, G(P.uy), bytes, hipMemcpyDeviceToHost));
  return AFC_OK;
}

int afc_op_project(afc_env *env, float *ux, float *uy, float *p, int32_t *iters, float *resid_linf) {
  if (check_env(env)) return AFC_EINVAL;
  if (!ux || !uy || !p) return fail(AFC_EINVAL, "afc_op_project: null argument");
  if (env->P.impl == 0) return fail(AFC_EINVAL, "afc_op_project: impl 2 only (the fused projection of the optimised kernels)");
  HIP_TRY(hipSetDevice(env->cfg.device));
  HIP_TRY(hipStreamSynchronize(env->stream));
  const StepParams &P = env->P;
  const size_t bytes = (size_t)P.N * sizeof(float);
  HIP_TRY(hipMemcpy(G(P.fx), ux, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(G(P.fy), uy, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(G(P.p), p, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(G(P.d), p, bytes, hipMemcpyHostToDevice));  // cells the solver never writes keep p's values
  int *dit = reinterpret_cast<int *>(env->d_i32);
  float *dl = env->d_f1, *dsh = env->d_f2;
  StepParams Pq = P;
  Pq.gate_limit = 0;  // a single workgroup: nobody to take turns with
  Pq.stamps = nullptr;
  if (P.gs_mode == 1)  // perf mode: the red-black smoother the env's step kernels use
    afc::launch_op_project_perf(env->lds_bytes, env->stream, Pq, dit, dl, dsh);
  else if (need_diag(env))
    afc::launch_op_project_diag(env->lds_bytes, env->stream, Pq, dit, dl, dsh);
  else
    afc::launch_op_project_prod(env->lds_bytes_step, env->stream, Pq, dit, dl, dsh);
  int rc = launch_check("k_op_project");
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(env->stream));
  std::vector<float> ph((size_t)P.N);
  HIP_TRY(hipMemcpy(ux, G(P.fx), bytes, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(uy, G(P.fy), bytes, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ph.data(), G(P.d), bytes, hipMemcpyDeviceToHost));
  int it = 0;
  float l = 0, sh = 0;
  HIP_TRY(hipMemcpy(&it, dit, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&l, dl, sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&sh, dsh, sizeof(float), hipMemcpyDeviceToHost));
  // the kernels keep the mean shift pending (project_tail_v2); p.plusEq(s) is this float addition (VectorField.pde:136)
  for (int k = 0; k < P.N; ++k) p[k] = ph[(size_t)k] + sh;
  if (iters) *iters = it;
  if (resid_linf) *resid_linf = l;
  return AFC_OK;
}

int afc_op_mg_solve(afc_env *env, float *x, const float *b, int32_t *iters, float *resid_linf) {
  if (check_env(env)) return AFC_EINVAL;
  if (!x || !b) return fail(AFC_EINVAL, "afc_op_mg_solve: null argument");
  HIP_TRY(hipSetDevice(env->cfg.device));
  HIP_TRY(hipStreamSynchronize(env->stream));
  const StepParams &P = env->P;
  const size_t bytes = (size_t)P.N * sizeof(float);
  HIP_TRY(hipMemcpy(G(P.p), x, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(G(P.fx), b, bytes, hipMemcpyHostToDevice));
  int *dit = reinterpret_cast<int *>(env->d_i32);
  float *dl = env->d_f1;
  StepParams Pq = P;
  Pq.gate_limit = 0;  // a single workgroup: nobody to take turns with
  if (P.gs_mode == 1)  // perf mode: the red-black smoother the env's step kernels use
    afc::launch_op_mg_perf(env->lds_bytes, env->stream, Pq, dit, dl);
  else if (need_diag(env) || P.impl == 0)
    afc::launch_op_mg_diag(env->lds_bytes, env->stream, Pq, dit, dl);
  else
    afc::launch_op_mg_prod(env->lds_bytes_step, env->stream, Pq, dit, dl);
  int rc = launch_check("k_op_mg");
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(env->stream));
  HIP_TRY(hipMemcpy(x, G(P.p), bytes, hipMemcpyDeviceToHost));
  int it = 0;
  float l = 0;
  HIP_TRY(hipMemcpy(&it, dit, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&l, dl, sizeof(float), hipMemcpyDeviceToHost));
  if (iters) *iters = it;
  if (resid_linf) *resid_linf = l;
  return AFC_OK;
}

int afc_fused_tail(const afc_env *env) {
  if (!env) return fail(AFC_EINVAL, "afc_fused_tail: null");
  // (the diagnostic instantiation -- dbg switches, a start stagger, phase stamps -- keeps the tails as passes of their own)
  return (env->P.fuse_tail != 0 && launch_mode(env) != 1) ? 1 : 0;
}

int afc_upleg(const afc_env *env) {
  if (!env) return fail(AFC_EINVAL, "afc_upleg: null");
  if (env->P.gs_mode == 1 || env->P.impl == 0) return 2;  // (perf mode / the reference kernels: their own strip-by-strip forms)
  return env->P.upleg;
}

int afc_placement_info(afc_env *env, int32_t *trials, float *best_gbs, float *median_gbs) {
  if (!env) return fail(AFC_EINVAL, "afc_placement_info: null env");
  if (trials) *trials = env->place_trials;
  if (best_gbs) *best_gbs = env->place_best;
  if (median_gbs) *median_gbs = env->place_median;
  return AFC_OK;
}

int afc_measure_copy_rate(int device, uint64_t bytes, int reps, float *gb_per_s) {
  if (!gb_per_s || bytes < (1u << 20) || reps < 1) return fail(AFC_EINVAL, "afc_measure_copy_rate: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(AFC_ENODEV, "afc_measure_copy_rate: no HIP device available");
  if (device < 0 || device >= ndev) return fail(AFC_EINVAL, "afc_measure_copy_rate: device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  const size_t n4 = (size_t)(bytes / 16);
  // the rate depends on where the driver places the buffers (4-5 % between allocations of one process, see afc_create):
  // a ceiling is the best of several placements, all held at once so that each is a different piece of memory
  constexpr int kPlacements = 4;
  f4v *buf[2 * kPlacements];
  int got = 0;
  for (; got < 2 * kPlacements; ++got)
    if (hipMalloc((void **)&buf[got], n4 * 16) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
  if (got < 2) {
    for (int q = 0; q < got; ++q) (void)hipFree(buf[q]);
    return fail(AFC_EHIP, "afc_measure_copy_rate: hipMalloc");
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best_ms = 1e30f;
  for (int pl = 0; pl + 1 < got; pl += 2) {
    f4v *a = buf[pl], *b = buf[pl + 1];
    (void)hipMemset(a, 0x3c, n4 * 16);
    // default-policy and non-temporal accesses, 2048 and 8192 workgroups: the best of the four forms is the ceiling
    for (int form = 0; form < 4; ++form)
      for (int r = 0; r < reps + 1; ++r) {  // the first pass of a form warms the TLB
        const dim3 grid((form & 2) ? 8192 : 2048);
        (void)hipEventRecord(e0, nullptr);
        if (form & 1)
          hipLaunchKernelGGL(k_copy_f4<true>, grid, dim3(256), 0, nullptr, a, b, n4);
        else
          hipLaunchKernelGGL(k_copy_f4<false>, grid, dim3(256), 0, nullptr, a, b, n4);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best_ms) best_ms = ms;
      }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (int q = 0; q < got; ++q) (void)hipFree(buf[q]);
  int rc = launch_check("k_copy_f4");
  if (rc) return rc;
  *gb_per_s = (float)(2.0 * 16.0 * (double)n4 / (best_ms * 1e-3) / 1e9);  // bytes read + bytes written
  return AFC_OK;
}

}  // extern "C"
