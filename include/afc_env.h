/*
 * afc_env.h -- C ABI of libafc_env.so: the MI355X-native, batched replacement for the
 * reference's Lilypad 2-D BDIM cylinder-wake environment (one `AFCCylinder` per env).
 *
 * Drop-in boundary.  The reference has no FFI on this path: the Processing sketch
 * (clientLilypad/clientCFD.pde) owns an `AFCCylinder test` object and calls
 *     test = new AFCCylinder(...)            clientCFD.pde:101, AFCCylinder.pde:11-42
 *     test.update2()                         clientCFD.pde:37,  AFCCylinder.pde:45-61
 *     test.force / test.t / test.flow.p      clientCFD.pde:38-42
 *     flow.resume / flow.write               BDIM.pde:226-251
 *     dat.addData03(... 32 probes ...)       SaveScalar.pde:61-72
 *     test.display() -> flow.u.curl()        AFCCylinder.pde:86-90, VectorField.pde:78-87
 * and talks to the agent by XML-RPC strings "Cl_Cd" -> "a1_a2" (clientCFD.pde:117-133,
 * server/server.py:114-143).  Each entry point below names the reference member it
 * replaces; a host program (our Python driver, or a JNI stub in the Processing sketch,
 * see INTEGRATION.md) binds exactly these symbols.
 *
 * Conventions: plain C, no torch types.  All functions return 0 on success or a negative
 * AFC_E* code; afc_last_error() gives the message for the calling thread's last failure.
 * Handles are not thread-safe.  Buffers are caller-owned.  By default buffers are HOST
 * pointers; after afc_set_io_space(env, AFC_MEM_DEVICE) the per-env I/O arrays of
 * afc_sim_step / afc_agent_step / afc_get_obs ... are DEVICE pointers (e.g. torch tensors'
 * data_ptr()) and the calls are asynchronous on the handle's stream.
 * Field layout everywhere is the reference's: a[i*m + j], i = x index (slow), j = y index,
 * one ghost layer per side, n = 16*res+2, m = 8*res+2 (Field.pde:25,33).
 */
#ifndef AFC_ENV_H
#define AFC_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFC_OK 0
#define AFC_EINVAL (-1)   /* bad argument / bad config */
#define AFC_ENODEV (-2)   /* no usable HIP device (the product has NO CPU fallback) */
#define AFC_EHIP (-3)     /* HIP runtime error */
#define AFC_EIO (-4)      /* file error */
#define AFC_ESTATE (-5)   /* call not valid in this state */
#define AFC_EDIVERGED (-6) /* NaN/Inf detected in an env (reported, env is flagged) */

#define AFC_MEM_HOST 0
#define AFC_MEM_DEVICE 1

/* which field for afc_get_field / afc_set_field */
#define AFC_FIELD_U 0    /* u.x, BDIM.u.x */
#define AFC_FIELD_V 1    /* u.y */
#define AFC_FIELD_P 2    /* p */
#define AFC_FIELD_VORT 3 /* VectorField.curl(), VectorField.pde:78-87 (read only) */

/* static coefficient arrays for afc_get_static (parity tests compare them with the oracle) */
#define AFC_STATIC_DELX 0
#define AFC_STATIC_DELY 1
#define AFC_STATIC_DEL1X 2
#define AFC_STATIC_DEL1Y 3
#define AFC_STATIC_WNXX 4 /* wnx.x */
#define AFC_STATIC_WNXY 5 /* wnx.y */
#define AFC_STATIC_WNYX 6 /* wny.x */
#define AFC_STATIC_WNYY 7 /* wny.y */
#define AFC_STATIC_CX 8   /* c.x = del.x*dt */
#define AFC_STATIC_CY 9
#define AFC_STATIC_W1X 10 /* BodyUnion weight of control cylinder 1 at u points */
#define AFC_STATIC_W2X 11
#define AFC_STATIC_W1Y 12
#define AFC_STATIC_W2Y 13

typedef struct afc_env afc_env;

/* Mirrors the constants the reference hard-codes in clientCFD.pde:5-14,94-96 and
 * AFCCylinder.pde:11-34.  afc_default_config() fills the reference-native values. */
typedef struct afc_config {
  int32_t resolution;   /* clientCFD.pde:14  (24); grid = 16*res x 8*res, D = res */
  int32_t x_lengths;    /* clientCFD.pde:94  (16) */
  int32_t y_lengths;    /* clientCFD.pde:94  (8)  */
  int32_t reynolds;     /* clientCFD.pde:94  (500); nu = D/Re */
  float t_step;         /* clientCFD.pde:9   tStep = .0075; dt = tStep*res */
  float dR;             /* clientCFD.pde:95  .125 */
  float gR;             /* clientCFD.pde:95  .2 */
  float theta;          /* clientCFD.pde:96  PI/3 */
  float act_scale;      /* clientCFD.pde:53-54 xi_m = 5*xi */
  int32_t substeps;     /* clientCFD.pde:12  callLearn = 16 */
  float init_time;      /* clientCFD.pde:6   initTime = 1 */
  float end_time;       /* clientCFD.pde:5   Time = 50 */
  int32_t num_envs;     /* batch of independent AFCCylinder instances */
  int32_t device;       /* HIP device ordinal */
  int32_t mg_max_iters; /* VectorField.pde:135  MGsolver(20, ...) */
  int32_t carry_quirk;  /* 1: keep the Cd/Cl carry of clientCFD.pde:41-47 (reference behaviour) */
  float reward_penalty; /* server.py:64  0.186756 */
  int32_t exact_sums;   /* 1: parity mode -- Field.sum (VectorField.pde:136) and Field.inner (MG.pde:34)
                           are accumulated by one thread in the reference's sequential order, making the
                           whole step bit-identical to the reference arithmetic; 0 (default): tree sums */
  int32_t impl;         /* 2 (default): optimised kernels; 0: the straightforward reference kernels
                           (same arithmetic; kept for A/B timing and bisecting).  impl 2 supports grids up to
                           256 cells in y (y_lengths*resolution <= 256: four columns per lane); afc_create
                           rejects taller grids with AFC_EINVAL -- create those with impl = 0 */
  int32_t stagger_us;   /* start offset between the (up to 4) envs sharing a CU, microseconds (0 = none) */
  int32_t reserved[1];  /* [0]: development switches (StepParams.dbg); 0 in production */
  int32_t gs_mode;      /* MG.smooth (MG.pde:79-92): 0 (default) = the reference's in-place lexicographic Gauss-Seidel,
                           iterate for iterate (every parity test runs in this mode; the headline benchmark too);
                           1 = red-black Gauss-Seidel (4 sweeps = 8 half-sweeps per smooth), a different but equally
                           convergent smoother -- VALIDATION ONLY: results agree with the reference to the tolerance
                           enforced in tests/test_gpu_perfmode.py (|du|, |dv| <= 1e-3, |dp| <= 2e-2, |dC| <= 3e-3 over 16
                           steps from init.bdim; 1.5 x that for u, v, p at 64 steps), not bit for bit, and on this machine
                           it is SLOWER than mode 0 (round 5: 509 k against 685 k env-sim-steps/s).
                           The finest level's up-leg runs as register-resident row pipelines (mg_up_rbw; AFC_RBW=<mask
                           of levels> in the environment selects others, 0 none: same bits).  No diagnostics:
                           afc_create refuses it together with stagger_us != 0 or reserved[0] != 0, afc_profile_phases
                           refuses to enable stamps; afc_op_mg_solve / afc_op_project run the red-black smoother too */
  int32_t fuse_tail;    /* the tail of u.project (VectorField.pde:136-140: u -= c grad p, setBC) -- and, for the corrector, the
                           Heun average of BDIM.pde:95-96 -- inside the advection pass that follows it (advdif_v3f: the
                           projected field is not re-read from memory; inside a launch of several sim-steps the corrector's
                           tail runs in the next sim-step's predictor pass; same bits either way):
                           0 (default) = where it is faster -- one-tile grids (<= 128 interior columns) in launches of at
                           least one env per CU; 1 = wherever the grid allows it; -1 = never */
  int32_t upleg;        /* the way back up a multigrid level (MG.pde:73-97: prolongation increment, smooth(4), increment; the
                           reference's lexicographic order and the same bits in every case):
                           1 = round 5's strip pipeline (mg_up_pipe: one wave marches 16-row strips along the rows, three
                               waves exit and fill behind it);
                           3 = the four ordered sweeps by one wave marching DOWN THE ROWS (mg_up_tr, round 6) on the levels
                               of 128 / 64 / 32 interior columns of grids up to 128 columns wide -- half the latency chain,
                               the same instruction count: -14 % per launch for one env, -9 % at one env per CU, +4 % at
                               four per CU (a kernel instantiation of its own, afc::lat);
                           0 (default) = by occupancy: 3 for launches of fewer than 2.5 envs per CU, else 1;
                           2 = strip by strip with workgroup barriers (mg_up_smooth; diagnostic instantiation) */
} afc_config;

/* Environment variables the SHIPPED library reads (all at afc_create; none of them changes a result):
 *   AFC_PLACEMENT_TRIALS=n  candidates of the placement search (default up to 32; 1 = no search), see afc_placement_info
 *   AFC_ARENA=<MiB>         size of the env's arena for its small device buffers (default: sized from the case)
 *   AFC_RBW=<level mask>    perf mode (gs_mode 1) only: the levels whose up-leg runs as register-resident row pipelines
 *   AFC_PRINT_ALLOC=1       prints the placement search, the slab and the multigrid levels' special cells to stderr
 * (the Python binding additionally honours AFC_ENV_LIB, and AFC_DBG together with AFC_DBG_ENABLE=1).  Builds with
 * -DAFC_DEVHOOKS=1 (tools/build_variant.sh; never the shipped library) also read AFC_SLAB_PTR / AFC_EXPORT_SLAB /
 * AFC_ALLOC_MODE / AFC_FUSE_TAIL / AFC_TR_MASK for tools/ab_inproc.py. */

/* fills *cfg with the reference-native case (res 24, Re 500, 1 env, device 0) */
int afc_default_config(afc_config *cfg);

/* new AFCCylinder(...) x num_envs (AFCCylinder.pde:11-42 with isResume=false): free stream
 * u=(1,0), p=0, t=0.  Builds the static geometry/coefficient fields on the host with Java
 * float semantics, uploads them, allocates per-env state in HBM. */
int afc_create(const afc_config *cfg, afc_env **out);
int afc_destroy(afc_env *env);
const char *afc_last_error(void);

/* array sizes incl. ghosts (BDIM.n, BDIM.m) and derived constants */
int afc_dims(const afc_env *env, int32_t *n, int32_t *m);
int afc_num_envs(const afc_env *env);
int afc_num_levels(const afc_env *env); /* multigrid levels (MG.pde:99-106) */
float afc_dt(const afc_env *env);

int afc_set_io_space(afc_env *env, int space);  /* AFC_MEM_HOST | AFC_MEM_DEVICE */
int afc_set_stream(afc_env *env, void *hip_stream); /* hipStream_t; NULL = default stream */
int afc_synchronize(afc_env *env);

/* BDIM.resume (BDIM.pde:239-251): parse a reference .bdim text checkpoint and make it the
 * RESET STATE of every env (AFCCylinder.pde:35-37); also loads it into all envs now. */
int afc_load_state_bdim(afc_env *env, const char *path);
/* same, from arrays in reference layout [n*m] (u, v, p) */
int afc_set_reset_state(afc_env *env, const float *u, const float *v, const float *p);
/* BDIM.write (BDIM.pde:226-237) for one env: `t`, `dt`, then one line `u, v, p` per cell, every number in
 * Java's Float.toString text (what `""+u.x.a[i][j]` prints), so the file is byte-compatible with the
 * reference's own checkpoints (clientLilypad/saved/init/init.bdim) */
int afc_save_state_bdim(afc_env *env, int env_idx, const char *path);
/* Java `String.valueOf(float)` / `""+f` (BDIM.pde:233, clientCFD.pde:120): shortest decimal that identifies the
 * float32, plain notation for 1e-3 <= |x| < 1e7, otherwise d.dddE-n.  buf must hold >= 16 chars.  Host only. */
int afc_format_float_java(float x, char *buf, int cap);

/* setUpNewSim (clientCFD.pde:93-114): envs with mask[e]!=0 (all if mask==NULL) restart from
 * the reset state with t=0, xi=0.  The averaging window (callLearn, Cd, Cl) is cleared too
 * unless carry_quirk is set (the reference leaks it across episodes). */
int afc_reset(afc_env *env, const uint8_t *mask);

/* AFCCylinder.update2 (AFCCylinder.pde:45-61) for every env:
 * xi_m [E,2] = (xi1_m, xi2_m) rotation rates (already scaled), force [E,2] = test.force. */
int afc_sim_step(afc_env *env, const float *xi_m, float *force);

/* One agent interaction = the body of draw() (clientCFD.pde:35-56) repeated until the next
 * action request: action [E,2] in [-1,1] -> xi, xi_m = act_scale*xi; sim-steps with force
 * averaging after t>init_time; returns obs [E,2] = (Cl, Cd) exactly as sent over XML-RPC,
 * reward [E] = -Cd - penalty*sum|a|^3 (server.py:61-65), done [E] = (t >= end_time).
 * nsteps_out (may be NULL) [E] = sim-steps executed for each env. */
int afc_agent_step(afc_env *env, const float *action, float *obs, float *reward,
                   uint8_t *done, int32_t *nsteps_out);

/* per-env scalars: t = AFCCylinder.t [E] */
int afc_get_time(afc_env *env, float *t);
/* MG iterations of the last two pressure solves [E,2] and their L_inf residual [E,2]
 * (the reference prints these, MG.pde:36) */
int afc_get_solver_stats(afc_env *env, int32_t *iters, float *resid);
/* diverged flag per env [E] (NaN/Inf guard; absent in the reference) */
int afc_get_diverged(afc_env *env, uint8_t *flags);
/* BDIM.checkCFL before the min with 1 (BDIM.pde:217 -> VectorField.CFL, VectorField.pde:225-235):
 * out [E] = 1/(max_interior(|u|+|v|) + 3 nu).  The update2 path runs at fixed dt and never calls it
 * (AFCCylinder.pde:46 is commented out); kept as the diagnostic the reference has. */
int afc_get_cfl(afc_env *env, float *out);
/* flags [E]: 1 if the env's last afc_agent_step ended because an averaging window completed (its obs is a
 * fresh "Cl_Cd" the reference would send, clientCFD.pde:44-49), 0 if it ended only because t reached
 * end_time or the step cap.  A window can complete on the very step that ends the episode; the reference
 * still sends that request (draw() asks inside the step, the episode ends on the next frame). */
int afc_get_window_complete(afc_env *env, uint8_t *flags);
/* hist8[k] = number of pressure solves, summed over envs since the last clear, that took k+1 multigrid
 * iterations (hist8[7]: 8 or more); the reference prints the count per solve (MG.pde:36). */
int afc_get_mg_histogram(afc_env *env, int64_t *hist8, int clear);

/* field access for one env, reference layout [n*m]; host pointers always */
int afc_get_field(afc_env *env, int env_idx, int which, float *out);
int afc_set_field(afc_env *env, int env_idx, int which, const float *in);
/* SaveScalar.addData03 probes (SaveScalar.pde:61-72): out [E,num] pressure at num points on
 * the main cylinder surface */
int afc_get_probes(afc_env *env, int num, float *out);
/* static coefficient arrays (host pointer, [n*m]); level>0 only for CX/CY (= MG lower.x/.y)
 * and returns the coarse arrays [n_l*m_l] */
int afc_get_static(afc_env *env, int which, int level, float *out);
int afc_level_dims(const afc_env *env, int level, int32_t *n, int32_t *m);

/* diagnostics: per-phase wall-clock ticks (100 MHz) accumulated by env 0 since the last call;
 * enable!=0 turns accumulation on.  Phase ids: see afc::PF_* in csrc/bdim_device.h. */
int afc_profile_phases(afc_env *env, int enable, int64_t *ticks_out, int count);

/* --- unit-level device entry points (parity tests drive single operators through the same
 * kernels the step uses; all pointers HOST, reference layout) --- */
/* VectorField.AdvDif (VectorField.pde:170-179) on env 0's geometry: (fx,fy) in/out.  Passing u0x == fx and u0y == fy
 * (the same arrays) is BDIM.update's u0 = u (BDIM.pde:82-84) and runs the predictor's form of the pass. */
int afc_op_advdif(afc_env *env, float *fx, float *fy, const float *u0x, const float *u0y);
/* MGsolver(itmx, PoissonMatrix(c), x, b) (MG.pde:30-38) with the env's own coefficient
 * hierarchy: x in/out, returns iterations in *iters */
int afc_op_mg_solve(afc_env *env, float *x, const float *b, int32_t *iters, float *resid_linf);
/* VectorField.project(c, p) (VectorField.pde:130-143) with the env's own c = del*dt: p = MGsolver(20, PoissonMatrix(c),
 * p, div u); p -= mean(p); u -= c*grad p; u.setBC().  ux, uy, p in/out.  Runs the fused projection of a half-step
 * (solve head, coarse legs, up-legs, projection tail) exactly as the step dispatches it; impl 2 only. */
int afc_op_project(afc_env *env, float *ux, float *uy, float *p, int32_t *iters, float *resid_linf);
/* Field.setBC of the velocity pair (Field.pde:209-234; u.x has gradientExit) */
int afc_op_setbc_u(afc_env *env, float *ux, float *uy);

/* Measurement helper, no counterpart in the reference: the device-to-device copy rate of the box, bench.py's
 * "copy ceiling" -- a hand-written 16-byte-per-lane copy of `bytes` bytes (256-thread workgroups, eight loads in flight
 * per thread; default-policy and non-temporal accesses on 2048 and 8192 workgroups; four placements of the buffer pair,
 * because the rate depends on where the driver puts them), best pass of `reps` per form and placement; *gb_per_s = (bytes read + bytes written) / time / 1e9. */
int afc_measure_copy_rate(int device, uint64_t bytes, int reps, float *gb_per_s);

/* What afc_create's placement search did for this env (no counterpart in the reference).  Batches above 256 MB take the
 * best of several candidate allocations of the whole state by a streaming probe: where the driver puts a buffer of that
 * size decides what it streams at (two kinds of placement, 6-7 % apart in the step kernel; the kind follows the physical
 * region, not the allocation call or its granularity -- tools/vmm_probe.hip).  The candidates are held at once, NEVER
 * more than a quarter of the device's free memory at the time of the call and at most 32 (AFC_PLACEMENT_TRIALS=n lowers
 * that, 1 switches the search off); the losers are freed before afc_create returns.  *trials = candidates probed (0: no
 * search -- small batch, switched off, or no room), *best_gbs / *median_gbs = probe rates of the kept candidate and of
 * the median one.  A kept candidate less than 3.5 % above the median means the search ended without a fast placement. */
int afc_placement_info(afc_env *env, int32_t *trials, float *best_gbs, float *median_gbs);

/* 1 if this env's step kernels run the projection tails inside the advection passes that follow them
 * (afc_config.fuse_tail, advdif_v3f), 0 if they are separate passes; negative on error */
int afc_fused_tail(const afc_env *env);

/* The up-leg this env's step kernels run (afc_config.upleg resolved: 1 = strip pipeline, 2 = barrier form, 3 = the sweeps
 * marching down the rows where the level has their shape); negative on error */
int afc_upleg(const afc_env *env);

#ifdef __cplusplus
}
#endif
#endif /* AFC_ENV_H */
