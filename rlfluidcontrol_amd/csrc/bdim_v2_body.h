// bdim_v2.h -- optimised phases of the per-env BDIM step (same arithmetic, same bits as
// bdim_device.h's reference kernels; selected with StepParams.impl == 2).
//
// Velocity buffers rotate instead of being copied (BDIM.pde:82,91 copy u0/us):
//   A = (ux,uy)   state at the start of the step (= u0 of both half-steps)
//   B = (usx,usy) predictor result  (= `us` of the corrector)
//   C = (fx,fy)   corrector result; Heun average (A = (C+B)/2) lands back in A.
// (included twice by bdim_v2.h: AFC_NS = prod with AFC_DIAG 0, AFC_NS = diag with AFC_DIAG 1)

// Inlining of the phases (experiment hooks, build with AFC_EXTRA_FLAGS=-DAFC_ADV_ATTR=__noinline__ ...): `inline` leaves
// the decision to the compiler, which inlines every phase into the step kernels (one register allocation for all).
// Measured on the same memory (tools/ab_inproc.py --share-slab, 1024 envs 256x128, ms per launch): everything inline
// 26.240; projection tail outlined 25.999 (-0.9 %); solve head outlined 26.151; both 26.081; advection + head + tail
// 26.020; up-leg / bottom / coarse down-legs outlined: +18 %.  The projection tail is the one
// phase that ships as a function of its own.
#ifndef AFC_ADV_ATTR
#define AFC_ADV_ATTR inline
#endif
#ifndef AFC_HEAD_ATTR
#define AFC_HEAD_ATTR inline
#endif
#ifndef AFC_TAIL_ATTR
#define AFC_TAIL_ATTR __noinline__
#endif
#ifndef AFC_UP_ATTR
#define AFC_UP_ATTR inline
#endif
#ifndef AFC_DOWN_ATTR
#define AFC_DOWN_ATTR inline
#endif
#ifndef AFC_BOT_ATTR
#define AFC_BOT_ATTR inline
#endif

namespace afc {
namespace AFC_NS {

#if AFC_DIAG
typedef Prof ProfT;
#define AFC_DBG_OF(P) ((P).dbg)
#else
struct ProfT {  // the shipped build: no stamps, no stop codes
  static constexpr int stop = 0;
  __device__ __forceinline__ void start(const StepParams &) {}
  __device__ __forceinline__ void mark(int) {}
};
#define AFC_DBG_OF(P) 0
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding
// global store of the wave (s_waitcnt vmcnt(0)), which puts a store round trip on the critical path
// of each tile of a streaming loop; inside those loops nothing read from global memory was written
// by this workgroup in the same phase, so only the LDS hand-over needs ordering.  Phase boundaries
// keep __syncthreads().
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __attribute__((address_space(3))) float lds_f;  // explicit LDS pointers: ds_* accesses, 32-bit addresses
typedef __attribute__((address_space(3))) int lds_i;
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));  // two consecutive floats, 4-byte aligned
typedef float f4a __attribute__((ext_vector_type(4), aligned(16)));
typedef __attribute__((address_space(3))) f4a lds_f4;
typedef __attribute__((address_space(3))) f2u lds_f2;
typedef float f2a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef __attribute__((address_space(3))) f2a8 lds_f2a;
// One Gauss-Seidel update of the march up to the subtraction of r, as the nine instructions it needs, in the
// reference's order (MG.pde:84-88): ((dW*lw + dE*le) + dS*ls) + dN*ln - r with dW = lane-1's previous result
// (DPP row_shr:1 folded INTO the multiply; lanes 0,16,32,48 have no DPP source and keep `w*lw`, the value from the
// ring times the coefficient, formed one instruction earlier) and dS = this lane's previous result.
// The march is bound by the LATENCY of this dependent chain (a dependent VALU instruction issues ~9.5 cycles after
// its producer, an independent one after ~5: tools/valu_lat.hip), so what counts is its depth: mul_dpp, three adds,
// sub, mul = 6.  The two leading products do not depend on `prev` and fill the two wait states the hardware needs
// between the VALU write of `prev` (the last instruction before this block) and its DPP read.
// UNI: the four coefficients are wave-uniform (scalar registers).
#define GS_CORE_ASM                                                          \
  "v_mul_f32 %0, %6, %0\n\t"                                                 \
  "v_mul_f32 %1, %7, %4\n\t"                                                 \
  "v_mul_f32_dpp %0, %3, %11 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"       \
  "v_mul_f32 %2, %8, %3\n\t"                                                 \
  "v_add_f32 %0, %0, %1\n\t"                                                 \
  "v_mul_f32 %1, %9, %5\n\t"                                                 \
  "v_add_f32 %0, %0, %2\n\t"                                                 \
  "v_add_f32 %0, %0, %1\n\t"                                                 \
  "v_sub_f32 %0, %0, %10"
// UNI: the four coefficients are wave-uniform (scalar registers).  FIN: also the final d = -(...)*inv.
// `w` is read-write and overwritten by the first instruction while the other inputs are read later: it is marked
// early-clobber so that it can never share a register with an input the compiler proves equal.  The two leading
// multiplies are the wait states between the producer of `prev` (the instruction right before the block: the block is
// asm volatile and `prev` is the previous block's output) and its DPP read; tests/test_build_resources.py checks the
// mul / mul / mul_dpp pattern in the shipped disassembly.
template <bool UNI, bool FIN>
__device__ __forceinline__ float gs_core(float prev, float w, float dE, float dN, float r, float lw, float le, float ls,
                                         float ln, float iv) {
  float a, b;
  if (UNI && FIN)
    asm volatile(GS_CORE_ASM "\n\tv_mul_f32_e64 %0, %12, -%0"
                 : "+&v"(w), "=&v"(a), "=&v"(b)
                 : "v"(prev), "v"(dE), "v"(dN), "s"(lw), "s"(le), "s"(ls), "s"(ln), "v"(r), "v"(lw), "v"(iv));
  else if (UNI)
    asm volatile(GS_CORE_ASM
                 : "+&v"(w), "=&v"(a), "=&v"(b)
                 : "v"(prev), "v"(dE), "v"(dN), "s"(lw), "s"(le), "s"(ls), "s"(ln), "v"(r), "v"(lw));
  else
    asm volatile(GS_CORE_ASM
                 : "+&v"(w), "=&v"(a), "=&v"(b)
                 : "v"(prev), "v"(dE), "v"(dN), "v"(lw), "v"(le), "v"(ls), "v"(ln), "v"(r), "v"(lw));
  return w;
}
#undef GS_CORE_ASM
__device__ __forceinline__ unsigned lds_addr(const float *p) { return (unsigned)(uintptr_t)(const lds_f *)p; }

// Values that are the same in every lane but reached the wave through vector loads (the level
// table sits in global memory) occupy a VGPR each and get spilled around the register-hungry
// loops; readfirstlane moves them to SGPRs.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Wave priorities.  Marches (serial chains: the critical path of their workgroup) run at 3.  Everything else ran at 0
// through round 5 -- and among equal priorities the hardware issues the OLDEST wave first, so that the four workgroups of a
// CU (b, b+256, b+512, b+768: dispatched in that order) drift 0.1 ms apart each, the first ahead (0.35 ms in the diagnostic
// instantiation), and the last of them ends 0.7 ms (1.3 ms) after the first while the CU empties (tools/desync_probe.py).  AFC_PRIO >= 0 (bdim_v2.h: the shipped and the
// diagnostic instantiation): the mates'
// base priorities (0 .. 2) rotate every 2^AFC_PRIO sim-steps of a launch, so that every mate leads for a quarter of
// it; a march saves the base (USER_PRIO of the STATUS register) and returns to it.  Only in the kernels of the grids of up
// to 128 columns (k_agent_step<1 | 2, *>): the wide grids' launches lose 6 % to this rotation (512x256, 1024 envs: 91.6 ->
// 97.1 ms; a static pattern instead gave the one-tile grids -0.1 .. -0.5 % where the rotation gives -0.7 .. -1.5 %).
#undef AFC_PRIO_RAISE
#undef AFC_PRIO_DROP
#if AFC_PRIO >= 0
__device__ __forceinline__ void prio_set(unsigned q) {
  if (q == 0u) __builtin_amdgcn_s_setprio(0);
  else if (q == 1u) __builtin_amdgcn_s_setprio(1);
  else if (q == 2u) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
}
#define AFC_PRIO_RAISE() const unsigned prio0 = __builtin_amdgcn_s_getreg((1 << 11) | (3 << 6) | 2); /* hwreg(HW_REG_STATUS, 3, 2) = USER_PRIO */ \
  __builtin_amdgcn_s_setprio(3)
#define AFC_PRIO_DROP() prio_set(prio0)
__device__ __forceinline__ void prio_rotate(int steps) {  // all waves, at the start of a sim-step of a launch
  // only where four workgroups share a CU: with three (768 envs) the same rotation measured +0.2 .. +0.5 % per launch
  if (gridDim.x > 768u) {
    const unsigned k = (blockIdx.x >> 8) & 3u, s = (unsigned)uni(steps) >> AFC_PRIO;
    const unsigned q = (k + s) & 3u;
    prio_set(q > 2u ? 2u : q);  // base priorities 0 .. 2 (two mates tie): 3 stays the marches' alone
  }
}
// The kernels of the 192-column grids (k_agent_step<3, *>: the reference's own 384x192 case): STATIC base priorities by mate,
// 0 / 0 / 1 / 3 in dispatch order -- the youngest on top.  Same memory, 1024 envs, per launch: -2.9 % (two boxes; 0/0/2/3
// -2.8 %, 0/0/1/2 -2.5 %, 0/1/2/3 -2.0 %, 0/1/1/3 -2.0 %, 0/0/0/3 -1.8 %, 0/0/0/1 -1.6 %, 0/0/1/1 -0.8 %, the oldest on top
// +-0.0 %, rotating them as above +6 %); 768 / 512 envs -0.2 %.  The 256-column kernels (<4, *>) answer none of these
// patterns by more than +-0.3 % and keep 0 for all.
__device__ __forceinline__ void prio_by_mate() { prio_set((0x3100u >> (4u * ((blockIdx.x >> 8) & 3u))) & 3u); }
#else  // (the statements of round 5, verbatim: afc::lat's code answers any edit with +-0.5 % per launch)
#define AFC_PRIO_RAISE() __builtin_amdgcn_s_setprio(3)
#define AFC_PRIO_DROP() __builtin_amdgcn_s_setprio(0)
#endif

__device__ __forceinline__ float uni(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
template <class T>
__device__ __forceinline__ T *uni(T *p) {  // p points to GLOBAL memory (all our field / table pointers do)
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  // rebuilt as an address-space-1 pointer so that accesses stay global_* (SGPR base) instead of flat_*
  return (T *)reinterpret_cast<AFC_AS1 T *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ Lv uni(const Lv &L) {
  Lv U;
  U.n = uni(L.n);
  U.m = uni(L.m);
  U.lx = uni(L.lx);
  U.ly = uni(L.ly);
  U.inv = uni(L.inv);
  U.r = uni(L.r);
  U.x = uni(L.x);
  U.d = uni(L.d);
  {  // the descriptor may sit in a by-value copy of the parameters (kernel arguments / private memory in
     // the operator kernels): keep that pointer generic
    const unsigned long long a = reinterpret_cast<unsigned long long>(L.desc);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    U.desc = reinterpret_cast<const LevelDesc *>(((unsigned long long)hi << 32) | lo);
  }
  return U;
}
struct DescU {  // the scalar fields of a LevelDesc, in SGPRs
  float cu, iv0;
  int zb, sp_i0, sp_i1;
};
__device__ __forceinline__ DescU desc_u(const LevelDesc &D) {
  DescU u;
  u.cu = uni(D.cu);
  u.iv0 = uni(D.invtab[0]);
  u.zb = uni(D.zero_bnd);
  u.sp_i0 = uni(D.sp_i0);
  u.sp_i1 = uni(D.sp_i1);
  return u;
}

struct F4 {
  float v[4];
};
__device__ __forceinline__ F4 load4u(const float *p) {  // 4-byte-aligned 16-byte load (global_load_dwordx4)
  F4 r;
  __builtin_memcpy(&r, p, 16);
  return r;
}
__device__ __forceinline__ void store4u(float *p, const F4 &v) { __builtin_memcpy(p, &v, 16); }

// 16-byte access at a 32-bit byte offset from a wave-uniform base: global_* with the base in SGPRs, no 64-bit lane address
__device__ __forceinline__ F4 ld4b(const float *base, unsigned boff) {
  F4 r;
  __builtin_memcpy(&r, (const char *)base + boff, 16);
  return r;
}
__device__ __forceinline__ float ld1b(const float *base, unsigned boff) {
  float r;
  __builtin_memcpy(&r, (const char *)base + boff, 4);
  return r;
}
__device__ __forceinline__ void st4b(float *base, unsigned boff, const F4 &v) { __builtin_memcpy((char *)base + boff, &v, 16); }
__device__ __forceinline__ void st1b(float *base, unsigned boff, float v) { __builtin_memcpy((char *)base + boff, &v, 4); }

// One QUICK face value (VectorField.bho, VectorField.pde:202-219) for the face between cell L
// and cell H = L+1 along one axis.  bLm = b[L-1], bHp = b[H+1]; uf = face velocity (positive:
// flow from L to H); okL/okH = "upwind cell is not in the fallback zone" (VectorField.pde:212).
// The reference evaluates every interior face twice (as the E face of L and the W face of H, or
// N/S) with identical operands -- addition is commutative -- so one evaluation serves both
// cells bit for bit; for uf == 0 the two evaluations may pick different upwind cells but the
// flux uf*value is +-0 either way.
__device__ __forceinline__ float quick_face(float bLm, float bL, float bH, float bHp, float uf, bool okL,
                                            bool okH) {
  const float bf = 0.5f * (bL + bH);
  const bool up = uf > 0.f;
  const float bc = up ? bL : bH, bd = up ? bH : bL, bu = up ? bLm : bHp;
  const bool ok = up ? okL : okH;
  const float q = bf - (1.f / 6.f) * (bd - 2 * bc + bu);
  const float b1 = bu + 10.f * (bc - bu);
  const float lim = __builtin_amdgcn_fmed3f(q, bc, __builtin_amdgcn_fmed3f(bc, bd, b1));  // :218,221-223
  return ok ? lim : bf;
}

// The same for the x- and the y-component at once: both faces have the same geometry (same cells,
// same fallback flags), so the arithmetic can use packed-f32 instructions (v_pk_add/mul_f32 --
// per-component IEEE operations, nothing fused, bits unchanged).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned long long lmask;  // one bit per lane, in a scalar register pair
__device__ __forceinline__ float lane_prev_keep(float own, float v) {  // lane-1's v (whole wave); lane 0 keeps `own`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, own), __builtin_bit_cast(int, v), 0x138,
                                                                0xF, 0xF, false));
}
// (callers pass ONE comparison -- a range test as a single unsigned compare: `a && b` reaches the ballot as a 0/1 VGPR and
// a second compare, four vector instructions instead of one)
__device__ __forceinline__ lmask lanes_if(bool c) { return __builtin_amdgcn_ballot_w64(c); }
// (lane in up ? okL : okH) ? a : b -- the mask logic on the scalar unit, one vector select
__device__ __forceinline__ float sel_upwind_ok(lmask up, lmask okL, lmask okH, float a, float b) {
  float r;
  lmask t1, t2;
  asm("s_and_b64 %1, %3, %4\n\t"
      "s_andn2_b64 %2, %5, %3\n\t"
      "s_or_b64 %1, %1, %2\n\t"
      "v_cndmask_b32_e64 %0, %7, %6, %1"
      : "=v"(r), "=&s"(t1), "=&s"(t2)
      : "s"(up), "s"(okL), "s"(okH), "v"(a), "v"(b)
      : "scc");
  return r;
}
// okL / okH arrive as LANE MASKS: "upwind ? okL : okH" is then three scalar instructions on the masks instead of a
// vector select between two booleans (five vector instructions per face and component)
__device__ __forceinline__ f2 quick_face2(f2 bLm, f2 bL, f2 bH, f2 bHp, f2 uf, lmask okL, lmask okH) {
  const f2 bf = 0.5f * (bL + bH);
  const bool upx = uf.x > 0.f, upy = uf.y > 0.f;
  f2 bc, bd, bu;
  bc.x = upx ? bL.x : bH.x;
  bd.x = upx ? bH.x : bL.x;
  bu.x = upx ? bLm.x : bHp.x;
  bc.y = upy ? bL.y : bH.y;
  bd.y = upy ? bH.y : bL.y;
  bu.y = upy ? bLm.y : bHp.y;
  const f2 q = bf - (1.f / 6.f) * (bd - 2.f * bc + bu);
  const f2 b1 = bu + 10.f * (bc - bu);
  f2 r;
  const float lx = __builtin_amdgcn_fmed3f(q.x, bc.x, __builtin_amdgcn_fmed3f(bc.x, bd.x, b1.x));
  const float ly = __builtin_amdgcn_fmed3f(q.y, bc.y, __builtin_amdgcn_fmed3f(bc.y, bd.y, b1.y));
  const lmask mx = lanes_if(upx), my = lanes_if(upy);
  r.x = sel_upwind_ok(mx, okL, okH, lx, bf.x);
  r.y = sel_upwind_ok(my, okL, okH, ly, bf.y);
  return r;
}

// F = (advection + nu*diffusion)(src)*dt + base on interior cells (VectorField.AdvDif,
// VectorField.pde:170-179), streamed through an LDS tile of R+4 rows.  Wave w owns R/4
// consecutive rows of the block and carries the east-face flux down the rows; lane l owns C
// adjacent columns and carries the north-face flux across them.  Non-interior cells of dst are
// NOT written (setbc_vel defines them).
// (TAG: the unit-operator kernels instantiate their own copies -- a second call site of the same instance would make
// the compiler outline it and change the step kernels' code)
template <int C, int TAG = 0>
__device__ inline void advdif_v2(const StepParams &P, const float *__restrict__ sx, const float *__restrict__ sy,
                                 const float *__restrict__ bx, const float *__restrict__ by, float *__restrict__ dx,
                                 float *__restrict__ dy, float *lds, const bool fold = false) {
  // fold: rows are stored with the wall columns setBC gives them (see advdif_v3)
  const int n = uni(P.n), m = uni(P.m), N = uni(P.N);
  const int R = uni(P.adv_rows), TR = R + 4;
  // diagnostics: 64 = no tile loads, 512 = tiles staged through registers
  const int xdbg = uni(AFC_DBG_OF(P));
  // tiles on 16-byte boundaries (the LDS-DMA copies below write 16 bytes per lane); TR*m + 4 floats apart, a multiple
  // of 16 bytes whenever TR*m is a multiple of 4
  float *tx = lds + 4 + (((16u - (lds_addr(lds) & 15u)) & 15u) >> 2), *ty = tx + TR * m + 4;
  const bool dma_ok = (((TR * m) & 3) == 0) && !(xdbg & 512);
  const int wave = uni((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // wave index in a scalar register: row indices and the fallback masks stay on the scalar unit
  const int RW = R >> 2;
  const int j0 = 1 + C * lane;
  const bool lane_on = (j0 <= m - 2);
  const float nu = uni(P.nu), dt = uni(P.dt);
  const bool same = (bx == sx) && (by == sy);
  for (int a = 1; a <= n - 2; a += R) {
    // ---- stage rows a-2 .. a+R+1 of src into LDS (flat, coalesced float2) ----
    const int t0 = (a - 2) * m;                    // flat index of tile row 0
    const int lo = t0 < 0 ? 0 : t0;
    const int hi = ((a + R + 2) * m < N) ? (a + R + 2) * m : N;
    lds_barrier();  // previous block's readers are done
    if (dma_ok && t0 >= 0 && (a + R + 2) * m <= N && !(xdbg & 64)) {
      // whole blocks (all but the first and the last): the tile is a flat copy of TR*m contiguous floats per field,
      // moved by the LDS-DMA path (global_load_lds_dwordx4: 1 KB per wave instruction straight into LDS -- no
      // registers, no ds_write, no address arithmetic; source 4-byte aligned, destination = wave-uniform base +
      // 16*lane, tools/glds_test.hip)
      const int nfl = TR * m;
      for (int p0 = wave * 256; p0 < nfl; p0 += 4 * 256) {
        const int q = p0 + 4 * lane;
        if (q < nfl) {
          __builtin_amdgcn_global_load_lds(sx + t0 + q, (__attribute__((address_space(3))) void *)(tx + p0), 16, 0, 0);
          __builtin_amdgcn_global_load_lds(sy + t0 + q, (__attribute__((address_space(3))) void *)(ty + p0), 16, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (!(xdbg & (64 | 512))) {
      // first / last block (the tile hangs over the array): the same path 4 bytes per lane, 256 B per instruction
      const int nfl = hi - lo, o0 = lo - t0;
      for (int p0 = wave * 64; p0 < nfl; p0 += 4 * 64) {
        if (p0 + lane < nfl) {
          __builtin_amdgcn_global_load_lds(sx + lo + p0 + lane, (__attribute__((address_space(3))) void *)(tx + o0 + p0), 4, 0, 0);
          __builtin_amdgcn_global_load_lds(sy + lo + p0 + lane, (__attribute__((address_space(3))) void *)(ty + o0 + p0), 4, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (!(xdbg & 64))
    for (int q = lo + 2 * threadIdx.x; q < hi; q += 2 * kThreads) {
      const float2 vx = *reinterpret_cast<const float2 *>(sx + q);
      const float2 vy = *reinterpret_cast<const float2 *>(sy + q);
      tx[q - t0] = vx.x;
      tx[q - t0 + 1] = vx.y;
      ty[q - t0] = vy.x;
      ty[q - t0 + 1] = vy.y;
    }
    lds_barrier();
#define TX(gi, gj) tx[((gi) - (a - 2)) * m + (gj)]
#define TY(gi, gj) ty[((gi) - (a - 2)) * m + (gj)]
    if (lane_on) {
      const int i_beg = a + wave * RW;
      f2 fE2[C];
      for (int ii = 0; ii < RW; ++ii) {
        const int i = i_beg + ii;
        if (i > n - 2) break;
        const bool rok = (i >= 2) && (i <= n - 2);       // row part of the fallback test for cells of row i
        const bool rokE = (i + 1 >= 2) && (i + 1 <= n - 2);
        const bool rokW = (i - 1 >= 2) && (i - 1 <= n - 2);
        // the lane's C cells are contiguous: one (4-byte aligned) C-float load/store per array
        const bool full = (j0 + C - 1 <= m - 2);
        float bxv[C], byv[C], oxv[C], oyv[C];
        if (same) {  // predictor: the base field is the advected field, already in the tile
#pragma unroll
          for (int e = 0; e < C; ++e) {
            const int jj = (j0 + e <= m - 2) ? j0 + e : m - 2;
            bxv[e] = TX(i, jj);
            byv[e] = TY(i, jj);
          }
        } else if (full) {
          // the lane offset is made opaque per row: otherwise `base + lane offset` is hoisted out of the
          // row loop as two 64-bit lane addresses that do not fit the register file and come back from
          // scratch, one dependent round trip each, in front of every base-field load
          int off = i * m + j0;
          asm volatile("" : "+v"(off));
          __builtin_memcpy(bxv, bx + off, sizeof(float) * C);
          __builtin_memcpy(byv, by + off, sizeof(float) * C);
        } else {
#pragma unroll
          for (int e = 0; e < C; ++e) {
            const bool in = (j0 + e <= m - 2);
            bxv[e] = in ? bx[i * m + j0 + e] : 0.f;
            byv[e] = in ? by[i * m + j0 + e] : 0.f;
          }
        }
        // pass 1: the east and north faces of the lane's cells; pass 2 combines.  The south face of the lane's FIRST cell
        // is the north face of lane-1's LAST cell -- same cells, same fallback flags, evaluated once (DPP wave_shr:1);
        // lane 0 sits at the wall, where the face is in the fallback zone: its value is the plain average.
        f2 fEn[C], fNn[C], ccv[C], difv[C];
        f2 fSw = {0.f, 0.f};  // lane 0: vs * 0.5*(cs + cc)
#pragma unroll
        for (int e = 0; e < C; ++e) {
          const int j = j0 + e;
          if (j > m - 2) break;
          const lmask mC = lanes_if((unsigned)(j - 2) <= (unsigned)(m - 4));  // column part of the fallback test, as lane masks
          const lmask mCN = lanes_if((unsigned)(j - 1) <= (unsigned)(m - 4));
          // both components together (x: btype 1, VectorField.pde:183-188; y: btype 2, :189-194)
#define XY(gi, gj) f2{TX(gi, gj), TY(gi, gj)}
          const f2 cc = XY(i, j), ce = XY(i + 1, j), cw = XY(i - 1, j), cn = XY(i, j + 1), cs = XY(i, j - 1);
          ccv[e] = cc;
          if (ii == 0) {
            const f2 uo = 0.5f * (f2{cw.x, cs.x} + f2{cc.x, cc.x});  // x: x[i-1][j]+x[i][j]; y: x[i][j-1]+x[i][j]
            fE2[e] = uo * quick_face2(XY(i - 2, j), cw, cc, ce, uo, rokW ? mC : 0, rok ? mC : 0);  // = this row's west face
          }
          const f2 ue = 0.5f * (f2{ce.x, TX(i + 1, j - 1)} + f2{cc.x, ce.x});
          fEn[e] = ue * quick_face2(cw, cc, ce, XY(i + 2, j), ue, rok ? mC : 0, rokE ? mC : 0);
          if (e == 0) {
            const f2 vs = 0.5f * (f2{cc.y, cs.y} + f2{cw.y, cc.y});  // x: y[i][j]+y[i-1][j]; y: y[i][j-1]+y[i][j]
            fSw = vs * (0.5f * (cs + cc));
          }
          const f2 vn = 0.5f * (f2{cn.y, cc.y} + f2{TY(i - 1, j + 1), cn.y});
          fNn[e] = vn * quick_face2(cs, cc, cn, XY(i, j + 2), vn, rok ? mC : 0, rok ? mCN : 0);
          difv[e] = ce + cn - 4.f * cc + cw + cs;                           // :198-200
#undef XY
        }
        {  // the last cell the lane owns in this row (a partial last lane owns fewer than C)
          f2 last = fNn[0];
#pragma unroll
          for (int e = 1; e < C; ++e)
            if (j0 + e <= m - 2) last = fNn[e];
          fSw.x = lane_prev_keep(fSw.x, last.x);
          fSw.y = lane_prev_keep(fSw.y, last.y);
        }
#pragma unroll
        for (int e = 0; e < C; ++e) {
          const int j = j0 + e;
          if (j > m - 2) break;
          const f2 fW = fE2[e], fS = (e == 0) ? fSw : fNn[e > 0 ? e - 1 : 0];
          const f2 adv = ((fW - fEn[e]) + (fS - fNn[e]));
          const f2 o = (adv + nu * difv[e]) * dt + f2{bxv[e], byv[e]};              // :174-175
          oxv[e] = o.x;
          oyv[e] = o.y;
          fE2[e] = fEn[e];
        }
        if (fold) {  // Field.pde:217-224: u.x mirrors the adjacent interior cell; u.y[.][0] = old u.y[.][1], u.y[.][1] = u.y[.][m-1] = 0
          if (j0 == 1) {
            dx[i * m] = oxv[0];
            dy[i * m] = oyv[0];
            oyv[0] = 0.f;
          }
          const int el = m - 2 - j0;  // the lane that owns column m-2
          if (el >= 0 && el < C) {
            float last = oxv[0];
#pragma unroll
            for (int e = 1; e < C; ++e)
              if (e <= el) last = oxv[e];
            dx[i * m + m - 1] = last;
            dy[i * m + m - 1] = 0.f;
          }
        }
        if (full) {
          __builtin_memcpy(dx + i * m + j0, oxv, sizeof(float) * C);
          __builtin_memcpy(dy + i * m + j0, oyv, sizeof(float) * C);
        } else {
#pragma unroll
          for (int e = 0; e < C; ++e)
            if (j0 + e <= m - 2) {
              dx[i * m + j0 + e] = oxv[e];
              dy[i * m + j0 + e] = oyv[e];
            }
        }
      }
    }
#undef TX
#undef TY
  }
  __syncthreads();
}

// s_waitcnt vmcnt(N) with the largest even immediate N <= K (K wave-uniform): at most N of the wave's youngest
// vector-memory operations may still be outstanding afterwards.
__device__ __forceinline__ void wait_vm_le(int K) {
  if (K >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (K >= 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
  else if (K >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (K >= 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if (K >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (K >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (K >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (K >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (K >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (K >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (K >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (K >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One LDS-DMA copy: lane l moves 16 bytes from sbase + voff (its own byte offset) to LDS byte address lds + 16*l.
// Written as inline assembly so that the compiler's wait-count pass does not see an LDS write it has to drain before
// the next LDS read: the waits are placed by hand (wait_vm_le).
// m0: the block sets it and names it as clobbered, but LLVM ignores clobbers of RESERVED registers (the build log says
// so once per instantiation), and the compiler does use m0 itself (for the LDS-DMA builtin of mg_bottom's table load).
// What guarantees that neither overwrites the other's value before its use is
// tests/test_build_resources.py::test_every_lds_dma_sets_m0_right_in_front_of_itself, which reads the shipped disassembly:
// every mention of m0 in every kernel is an s_mov into it, every global_load_lds finds one within the eight instructions
// in front of it with no branch, branch target or other copy in between, and no other implicit reader of m0 exists.
// (Saving and restoring m0 around each copy instead would cost two scalar instructions per row and field in the
// issue-bound advection.)
// (Cache policy, round 6: the same copies with the `nt` bit -- and non-temporal stores of the advection's results, of the
// head's r, non-temporal last reads in the head and the up-leg -- measured on the same memory at 1024 envs: +0.3 ... +2.4 %
// per launch in every combination; what is streamed once here is re-read a phase later, from the Infinity Cache if at all.)
__device__ __forceinline__ void lds_dma16(const float *sbase, unsigned voff, unsigned ldsaddr) {
  const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)ldsaddr);  // uniform by construction; pins it to an SGPR
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(uni(sbase)), "s"(la) : "memory", "m0");
}

// advdif_v3: per wave a ring of kAdvNS rows per field plus two rows per base field, row pitch 64*C + 4 floats
constexpr int kAdvNS = 7;
__device__ __host__ __forceinline__ int adv3_lds_floats(int C) { return 4 * (2 * kAdvNS + 4) * (64 * C + 4) + 4; }

// The same pass as advdif_v2 as FOUR INDEPENDENT ROW PIPELINES, one per wave, without a single workgroup barrier:
// wave w owns a contiguous quarter of the rows and walks down it.  The rows of src it needs (i-1 .. i+2; i-2 only for
// the west face of its first row, afterwards the east-face flux is carried down the quarter in registers) live in a
// private ring of 7 row slots per field that the wave itself fills by the LDS-DMA path (global_load_lds_dwordx4, one
// instruction per row and field, ceil(m/4) lanes x 16 bytes) THREE ROWS AHEAD of the row it computes; the corrector's
// base field comes the same way, one row ahead, through a double row buffer.  The loads of the rows to come are thus
// in flight while the current row's faces are evaluated, and a wave only ever waits for its own counted vmcnt (all
// vector-memory operations of a wave complete in order).  Issue order of row iteration i:
//   copies of base row i+1 | copies of src row i+5 | wait: src row i+2 | faces | wait: base row i | last add | stores
// For grids of 64*C interior columns (one column tile) or of two column tiles of at most 64*C (see TILES below), >= 8
// rows per wave; the caller uses advdif_v2 otherwise.
// The lane index, recomputed where it is used (two VALU instructions): values derived from threadIdx.x that are live
// across the whole kernel get spilled around the register-hungry loops and every reload inside such a loop is a
// scratch load -- a vector-memory operation whose wait drains the staging copies in flight.
__device__ __forceinline__ unsigned lane_fresh() {
  unsigned l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// KSTP: the steady-state wait counts as immediates (see KST below) -- by default what the step kernel of the tag ships;
// the unit-operator kernels pass it explicitly so that tests/test_gpu_parity.py::test_advdif_bit_exact runs the wait path
// of the step kernel it stands for
template <int C, int TAG = 0, bool KSTP = (TAG == 3 || TAG == 4)>
__device__ AFC_ADV_ATTR void advdif_v3(const StepParams &P, const float *__restrict__ sx, const float *__restrict__ sy,
                                 const float *__restrict__ bx, const float *__restrict__ by, float *__restrict__ dx,
                                 float *__restrict__ dy, float *lds, const bool fold) {
  // fold: every row is stored WITH the wall columns setBC would give it (Field.pde:217-224: u.x mirrors the adjacent
  // interior cell; u.y[.][0] = old u.y[.][1], then u.y[.][1] = u.y[.][m-1] = 0) by the lane at the wall -- two
  // more store instructions per row -- so that what is left of setBC is whole-row work (setbc_rows)
  //
  // TILES.  A wave's pipeline covers a contiguous run of rows of ONE column tile of W = lanes x C interior columns; its
  // ring rows hold the W + 4 columns c0 .. c0 + W + 3 the tile's QUICK stencils reach (two columns either side).  Grids
  // of 64*C interior columns are one tile wide and the four waves split the rows in quarters (256x128, 128x64).  Wider
  // grids (384x192: W = 96, 48 lanes; 512x256: W = 128) are TWO tiles wide -- their rings would not fit the workgroup's
  // 40 KB as whole rows -- and the waves are a 2 x 2 arrangement of row halves and column halves.  The only thing a
  // right-hand tile cannot take from its neighbour is the south-face flux of its first column (lane 0 has no lane - 1):
  // it evaluates that face itself, from the two halo columns, with the expression the left tile's last lane uses for
  // its north face.
  constexpr int PR = 64 * C + 4, NS = kAdvNS;    // floats per ring row (16-byte rows); ring rows per field
  // steady-state wait counts as immediates instead of formed per row (below): measured per kernel instantiation on the same
  // memory, -0.8 % / -0.5 % per launch in the kernels of the two-tile grids (512x256, 384x192) and +0.2 % in the one-tile
  // kernel of 256x128 -- each gets the form that is faster for it
  constexpr bool KST = KSTP;
  const int ST = fold ? 4 : 2;  // store instructions per row iteration
  constexpr unsigned PRB = PR * 4u;              // bytes per ring row
  constexpr int YO = NS * PR;                    // floats from a row of the x ring to the same row of the y ring
  const int n = uni(P.n), n_ = n - 2, m = uni(P.m), m_ = m - 2;
  const int wave = uni((int)(threadIdx.x >> 6));
  const int ncol = (m_ > 64 * C) ? 2 : 1, nrow = 4 / ncol;
  const int tcol = (ncol == 2) ? (wave & 1) : 0, trow = (ncol == 2) ? (wave >> 1) : wave;
  const int W = (ncol == 2) ? (m_ >> 1) : m_, lanes_on = W / C;  // (no runtime division at a phase entry: every wave waits for it)
  // first interior column of the tile; column at ring index 0: always jlo - 2, so that the lane's first cell sits at
  // ring index 2 + C*lane -- for C = 2 every pair of columns (j0 + 2k, j0 + 2k + 1) is one aligned 8-byte LDS read (as
  // dword pairs at odd indices the reads of a wave hit every bank twice).  A left-hand tile's index 0 is "column -1":
  // the float in front of the row (the previous row's wall / the previous env / the pad in front of the state slab),
  // staged and never used.
  const int jlo = 1 + tcol * W, c0 = jlo - 2;
  const int LQ = (W + 7) >> 2;                            // 16-byte pieces of a staged row (W + 4 columns)
  const int RPW = (ncol == 2) ? ((n_ + 1) >> 1) : ((n_ + 3) >> 2);  // ceil(n_ / nrow), nrow = 4 / ncol
  const int i_first = 1 + trow * RPW, i_last = (i_first + RPW - 1 < n_) ? i_first + RPW - 1 : n_;
  const unsigned ring = uni((int)(((lds_addr(lds) + 15u) & ~15u) + (unsigned)wave * ((2 * NS + 4) * PRB)));  // x ring
  const float nu = uni(P.nu), dt = uni(P.dt);
  const bool same = (bx == sx) && (by == sy);
  const int nb = same ? 0 : 2;  // base-row copies per iteration
  const float *sxc = sx + c0, *syc = sy + c0, *bxc = bx + c0, *byc = by + c0;  // staging sources: column c0 of row 0
  if (i_first <= i_last) {
    const int r_hi = (i_last + 2 < n - 1) ? i_last + 2 : n - 1;  // last row this wave stages
    int r_next = i_first - 2, s_next = 0;  // next row to stage and its ring slot
#define ADV3_STAGE()                                                                   \
  {                                                                                    \
    const unsigned ln_ = lane_fresh();                                                 \
    if (r_next >= 0 && r_next <= r_hi && ln_ < (unsigned)LQ) {                         \
      const unsigned so_ = 4u * (unsigned)(r_next * m) + 16u * ln_;                    \
      lds_dma16(sxc, so_, ring + (unsigned)s_next * PRB);                              \
      lds_dma16(syc, so_, ring + (unsigned)(s_next + NS) * PRB);                       \
    }                                                                                  \
    ++r_next;                                                                          \
    s_next = (s_next + 1 == NS) ? 0 : s_next + 1;                                      \
  }
#define ADV3_BASE(row)                                                                 \
  {                                                                                    \
    const unsigned ln_ = lane_fresh();                                                 \
    if (ln_ < (unsigned)LQ) {                                                          \
      const unsigned so_ = 4u * (unsigned)((row) * m) + 16u * ln_;                     \
      const unsigned lb_ = ring + (unsigned)(2 * NS + 2 * ((row) & 1)) * PRB;          \
      lds_dma16(bxc, so_, lb_);                                                        \
      lds_dma16(byc, so_, lb_ + PRB);                                                  \
    }                                                                                  \
  }
    for (int q = 0; q < NS; ++q) ADV3_STAGE()  // rows i_first-2 .. i_first+4
    if (nb) ADV3_BASE(i_first)
    f2 fE2[C];
    int s0 = 0;  // ring slot of row i-2
    for (int i = i_first; i <= i_last; ++i) {
      const int t = i - i_first;
      if (nb && i < i_last) ADV3_BASE(i + 1)
      if (t > 0) ADV3_STAGE()  // row i+5 replaces row i-2 (dead: only the first row of the run reads its row i-2)
      {  // how many of the wave's youngest vector-memory operations may still be in flight once row i+2 has landed.
        // Per iteration j, in order: nb base copies (row j+1, if j < i_last), 2 src copies (row j+5, if staged), ST
        // stores (row j); the first iteration issues its src copies last (row i-2 is still read by its west faces).
        if (KST && t > 3 && i + 5 <= r_hi) {
          // steady state (all but the first four and the last three rows of a run): the count is a constant of the call,
          // 3 ST + 3 nb + 6 -- formed per row it was ~40 scalar instructions plus wait_vm_le's chain of up to twelve
          // compare-and-branch pairs
          if (nb) {
            if (ST == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
          } else {
            if (ST == 4) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          }
        } else {
          int K;
          if (t >= 3) {  // the copies of row i+2 were issued in iteration i-3: followed by its stores unless that was the first
            K = (t > 3) ? ST : 0;
            for (int jj = i - 2; jj <= i; ++jj) K += ((jj < i_last) ? nb : 0) + ((jj + 5 <= r_hi) ? 2 : 0) + ((jj < i) ? ST : 0);
          } else {  // ... or by the prologue: rows first-2 .. first+4, then base row first; whole iterations follow
            // (>= 8 rows per wave, so every base / src copy of the first three iterations exists)
            K = 2 * (2 - t) + nb + nb + (t > 0 ? 2 : 0);  // rest of the prologue, this iteration's copies
            K += t * (nb + 2 + ST);                       // earlier iterations
          }
          wait_vm_le(K);
        }
      }
      const int j0 = jlo + C * (int)lane_fresh();
      const unsigned lo = ring + 4u * (unsigned)(j0 - c0);
      const int s1 = (s0 + 1 >= NS) ? s0 + 1 - NS : s0 + 1, s2 = (s0 + 2 >= NS) ? s0 + 2 - NS : s0 + 2,
                s3 = (s0 + 3 >= NS) ? s0 + 3 - NS : s0 + 3, s4 = (s0 + 4 >= NS) ? s0 + 4 - NS : s0 + 4;
      const lds_f *r1 = (const lds_f *)(uintptr_t)(lo + (unsigned)s1 * PRB);
      const lds_f *r2 = (const lds_f *)(uintptr_t)(lo + (unsigned)s2 * PRB);
      const lds_f *r3 = (const lds_f *)(uintptr_t)(lo + (unsigned)s3 * PRB);
      const lds_f *r4 = (const lds_f *)(uintptr_t)(lo + (unsigned)s4 * PRB);
      // The values the row's stencils touch, by (ring row, column relative to the lane's first cell): row i columns
      // -2 .. C+1 (-2 only for a right-hand tile's own south face), row i-1 columns 0 .. C (x: .. C-1), row i+1 columns
      // -1 .. C-1 (y: from 0), row i+2 columns 0 .. C-1.  C = 2: aligned 8-byte reads (the lane's first cell is at an even
      // ring index), single dwords for the odd ones out.
      float X_r2[C + 4], Y_r2[C + 4], X_r1[C + 1], Y_r1[C + 1], X_r3[C + 1], Y_r3[C + 1], X_r4[C], Y_r4[C];
      constexpr int O_r2 = 2, O_r1 = 0, O_r3 = 1, O_r4 = 0, O_r0 = 0;
#define LDP(r, dj) (*(const lds_f2a *)((r) + (dj)))
      if constexpr (C == 2) {
#pragma unroll
        for (int d = -2; d < C + 2; d += 2) {
          const f2 a = LDP(r2, d), b = LDP(r2 + YO, d);
          X_r2[d + 2] = a.x, X_r2[d + 3] = a.y, Y_r2[d + 2] = b.x, Y_r2[d + 3] = b.y;
        }
        const f2 a1 = LDP(r1, 0), b1 = LDP(r1 + YO, 0), a3 = LDP(r3, 0), b3 = LDP(r3 + YO, 0), a4 = LDP(r4, 0), b4 = LDP(r4 + YO, 0);
        X_r1[0] = a1.x, X_r1[1] = a1.y, X_r1[C] = 0.f, Y_r1[0] = b1.x, Y_r1[1] = b1.y, Y_r1[C] = r1[YO + C];
        X_r3[0] = r3[-1], X_r3[1] = a3.x, X_r3[2] = a3.y, Y_r3[0] = 0.f, Y_r3[1] = b3.x, Y_r3[2] = b3.y;
        X_r4[0] = a4.x, X_r4[1] = a4.y, Y_r4[0] = b4.x, Y_r4[1] = b4.y;
      } else {
#pragma unroll
        for (int d = -2; d < C + 2; ++d) X_r2[d + 2] = (d == -2 && !tcol) ? 0.f : r2[d], Y_r2[d + 2] = (d == -2 && !tcol) ? 0.f : r2[YO + d];
#pragma unroll
        for (int d = 0; d <= C; ++d) X_r1[d] = (d < C) ? r1[d] : 0.f, Y_r1[d] = r1[YO + d];
#pragma unroll
        for (int d = -1; d < C; ++d) X_r3[d + 1] = r3[d], Y_r3[d + 1] = (d >= 0) ? r3[YO + d] : 0.f;
#pragma unroll
        for (int d = 0; d < C; ++d) X_r4[d] = r4[d], Y_r4[d] = r4[YO + d];
      }
      // (ring row, column relative to the lane's first cell) -> both components
#define XY(r, dj) f2{X_##r[(dj) + O_##r], Y_##r[(dj) + O_##r]}
      const bool rok = (i >= 2) && (i <= n - 2);  // row part of the fallback test for cells of row i
      const bool rokE = (i + 1 >= 2) && (i + 1 <= n - 2);
      float oxv[C], oyv[C];
      f2 fEn[C], fNn[C], difv[C], ccv[C];
      f2 fSw = {0.f, 0.f};  // lane 0 at the wall: vs * 0.5*(cs + cc)
#pragma unroll
      for (int e = 0; e < C; ++e) {
        const int j = j0 + e;
        const lmask mC = lanes_if((unsigned)(j - 2) <= (unsigned)(m - 4));  // column part of the fallback test, as lane masks
        const lmask mCN = lanes_if((unsigned)(j - 1) <= (unsigned)(m - 4));
        const f2 cc = XY(r2, e), ce = XY(r3, e), cw = XY(r1, e), cn = XY(r2, e + 1), cs = XY(r2, e - 1);
        ccv[e] = cc;
        if (t == 0) {  // the run's first row: its west face (afterwards = the previous row's east face)
          const lds_f *r0p = (const lds_f *)(uintptr_t)(lo + (unsigned)s0 * PRB);
          float X_r0[C], Y_r0[C];
          if constexpr (C == 2) {
            const f2 a0 = LDP(r0p, 0), b0 = LDP(r0p + YO, 0);
            X_r0[0] = a0.x, X_r0[1] = a0.y, Y_r0[0] = b0.x, Y_r0[1] = b0.y;
          } else {
#pragma unroll
            for (int d = 0; d < C; ++d) X_r0[d] = r0p[d], Y_r0[d] = r0p[YO + d];
          }
          const bool rokW = (i - 1 >= 2) && (i - 1 <= n - 2);
          const f2 uo = 0.5f * (f2{cw.x, cs.x} + f2{cc.x, cc.x});
          fE2[e] = uo * quick_face2(XY(r0, e), cw, cc, ce, uo, rokW ? mC : 0, rok ? mC : 0);
        }
        const f2 ue = 0.5f * (f2{ce.x, X_r3[e - 1 + O_r3]} + f2{cc.x, ce.x});
        fEn[e] = ue * quick_face2(cw, cc, ce, XY(r4, e), ue, rok ? mC : 0, rokE ? mC : 0);
        if (e == 0) {
          const f2 vs = 0.5f * (f2{cc.y, cs.y} + f2{cw.y, cc.y});
          fSw = vs * (0.5f * (cs + cc));
        }
        const f2 vn = 0.5f * (f2{cn.y, cc.y} + f2{Y_r1[e + 1], cn.y});
        fNn[e] = vn * quick_face2(cs, cc, cn, XY(r2, e + 2), vn, rok ? mC : 0, rok ? mCN : 0);
        difv[e] = ce + cn - 4.f * cc + cw + cs;  // VectorField.pde:198-200
      }
      if (tcol) {  // right-hand tile: lane 0's south face = the north face of column j0-1, evaluated here (e = -1 above)
        const lmask mC = lanes_if((unsigned)(j0 - 3) <= (unsigned)(m - 4));
        const lmask mCN = lanes_if((unsigned)(j0 - 2) <= (unsigned)(m - 4));
        const f2 cc = XY(r2, -1), cn = XY(r2, 0), cs = XY(r2, -2);
        const f2 vn = 0.5f * (f2{cn.y, cc.y} + f2{Y_r1[0], cn.y});
        fSw = vn * quick_face2(cs, cc, cn, XY(r2, 1), vn, rok ? mC : 0, rok ? mCN : 0);
      }
#undef XY
      fSw.x = lane_prev_keep(fSw.x, fNn[C - 1].x);  // south face of the lane's first cell = north face of lane-1's last
      fSw.y = lane_prev_keep(fSw.y, fNn[C - 1].y);
      f2 basev[C];
      if (nb) {
        // base row i: copied in iteration i-1 (in front of that iteration's src copies and stores, and of this
        // iteration's copies) or by the prologue
        if (KST && t > 0 && i + 5 <= r_hi) {  // steady state: 2 + ST + nb + 2 with nb = 2
          if (ST == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else
          wait_vm_le((t > 0 ? ((i + 4 <= r_hi) ? 2 : 0) + ST : 0) + ((i < i_last) ? nb : 0) + ((t > 0 && i + 5 <= r_hi) ? 2 : 0));
        const lds_f *bb = (const lds_f *)(uintptr_t)(lo + (unsigned)(2 * NS + 2 * (i & 1)) * PRB);
        if constexpr (C == 2) {
          const f2 ab = LDP(bb, 0), bb2 = LDP(bb + PR, 0);
          basev[0] = f2{ab.x, bb2.x}, basev[1] = f2{ab.y, bb2.y};
        } else {
#pragma unroll
          for (int e = 0; e < C; ++e) basev[e] = f2{bb[e], bb[PR + e]};
        }
      } else {
#pragma unroll
        for (int e = 0; e < C; ++e) basev[e] = ccv[e];
      }
#pragma unroll
      for (int e = 0; e < C; ++e) {
        const f2 fW = fE2[e], fS = (e == 0) ? fSw : fNn[e > 0 ? e - 1 : 0];
        const f2 adv = ((fW - fEn[e]) + (fS - fNn[e]));
        const f2 o = (adv + nu * difv[e]) * dt + basev[e];  // VectorField.pde:174-175
        oxv[e] = o.x;
        oyv[e] = o.y;
        fE2[e] = fEn[e];
      }
      const unsigned ob = 4u * (unsigned)(i * m + j0);  // byte offset of the lane's cells of row i (32-bit: SGPR base + offset)
      const unsigned ln_ = lane_fresh();
      if (fold) {
        // the lane at the lower wall (lane 0 of a left-hand tile) and the lane at the upper wall (the last lane of a
        // right-hand tile): at least one of them exists in every wave, so these are always two store instructions
        const bool first = (ln_ == 0u) && (tcol == 0), last = (ln_ == (unsigned)(lanes_on - 1)) && (tcol == ncol - 1);
        if (first || last) {
          const unsigned gb = 4u * (unsigned)(i * m + (first ? 0 : m - 1));
          st1b(dx, gb, first ? oxv[0] : oxv[C - 1]);
          st1b(dy, gb, first ? oyv[0] : 0.f);
        }
        if (first) oyv[0] = 0.f;
      }
      if (ln_ < (unsigned)lanes_on) {  // (a tile of 96 columns leaves the upper 16 lanes idle)
        __builtin_memcpy((char *)dx + ob, oxv, sizeof(float) * C);
        __builtin_memcpy((char *)dy + ob, oyv, sizeof(float) * C);
      }
      asm volatile("" ::: "memory");
      if (t == 0) ADV3_STAGE()  // row i+5 replaces row i-2, whose last reader was this row's west face
      s0 = s1;
    }
#undef ADV3_STAGE
#undef LDP
#undef ADV3_BASE
  }
  __syncthreads();
}

#include "bdim_v2_advfuse.h"

// BDIM.updateUP lines 118-120 restricted to the cells where it is not the identity:
// del != 1 or del1 != 0 (|d| < eps, plus the solid interior).  Everywhere else
// u = 1*F - ub*0 + 0*g == F.  Every band cell must see the un-forced F of its neighbours, as the reference's
// whole-array expression does: all results are staged in LDS, one barrier, then scattered.
// The phase moves almost no data but runs while other workgroups of the chip stream, when a dependent global-memory
// round trip costs microseconds: a thread therefore fetches the indices of ALL its cells first (two x- and two y-cells
// per pass), then issues every gather of every cell before the first result is needed -- two round trips per pass
// instead of two per cell and component.
__device__ __forceinline__ float forcing_x_cell(const StepParams &P, const float *__restrict__ fx, int k, int m,
                                                float dphi1, float dphi2) {
  const float del = G(P.delx)[k], R = fx[k], ub = ub_x_at(P, k, dphi1, dphi2);
  float un = del * R - ub * (del + (-1.f));
  const float dE = fx[k + m] - ub_x_at(P, k + m, dphi1, dphi2);
  const float dW = fx[k - m] - ub_x_at(P, k - m, dphi1, dphi2);
  const float dN = fx[k + 1] - ub_x_at(P, k + 1, dphi1, dphi2);
  const float dS = fx[k - 1] - ub_x_at(P, k - 1, dphi1, dphi2);
  const float g = 0.5f * (G(P.wnxx)[k] * (dE - dW) + G(P.wnyx)[k] * (dN - dS));
  return un + G(P.del1x)[k] * g;
}
__device__ __forceinline__ float forcing_y_cell(const StepParams &P, const float *__restrict__ fy, int k, int m,
                                                float dphi1, float dphi2) {
  const float del = G(P.dely)[k], R = fy[k], ub = ub_y_at(P, k, dphi1, dphi2);
  float un = del * R - ub * (del + (-1.f));
  const float dE = fy[k + m] - ub_y_at(P, k + m, dphi1, dphi2);
  const float dW = fy[k - m] - ub_y_at(P, k - m, dphi1, dphi2);
  const float dN = fy[k + 1] - ub_y_at(P, k + 1, dphi1, dphi2);
  const float dS = fy[k - 1] - ub_y_at(P, k - 1, dphi1, dphi2);
  const float g = 0.5f * (G(P.wnxy)[k] * (dE - dW) + G(P.wnyy)[k] * (dN - dS));
  return un + G(P.del1y)[k] * g;
}
// The same cell from the COMPACT BAND TABLE (StepParams::band_tab_x / _y, built by the host): everything the cell
// needs from the env-shared coefficient arrays -- del, del1, the two wall normals, and for the cell and its four
// neighbours the two control cylinders' weights and lever arms -- sits at the cell's own slot q of 24 arrays of `stride`
// floats, so that a wave reads it with coalesced loads (64 slots = 256 contiguous bytes per instruction) instead of 25
// gathers of 64 different cache lines each; only the five values of F itself are still gathered.  As gathers the phase
// was bound by the texture addresser: 375 fully divergent load instructions per env and half-step, 50 us per half-step
// with four envs per CU against 9 us for one env alone.  Cells out of reach of the control cylinders (weights exactly
// +0 at all five positions: ub == +0 there) come after the `nflag` cells within reach and skip the 20 loads.
// Same operands, same order, same bits as forcing_x_cell / forcing_y_cell.
template <bool YC>
__device__ __forceinline__ float forcing_cell_tab(const float *__restrict__ tab, int stride, int q, bool reach,
                                                  const float *__restrict__ f, int k, int m, float dphi1, float dphi2,
                                                  float dt) {
  const float del = tab[q], del1 = tab[stride + q], wnx = tab[2 * stride + q], wny = tab[3 * stride + q];
  float ub[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // centre, +m, -m, +1, -1
  if (reach) {
#pragma unroll
    for (int p_ = 0; p_ < 5; ++p_) {
      const float *t4 = tab + (4 + 4 * p_) * stride + q;
      const float w1 = t4[0], w2 = t4[stride], r1 = t4[2 * stride], r2 = t4[3 * stride];
      float v = 0.f;
      v += 0.f;  // main cylinder (ub_x_at / ub_y_at)
      v += ((YC ? (0.f + r1 * dphi1) : (0.f - r1 * dphi1)) / dt) * w1;
      v += ((YC ? (0.f + r2 * dphi2) : (0.f - r2 * dphi2)) / dt) * w2;
      ub[p_] = v;
    }
  }
  // (the five values of F: staging the bodies' neighbourhood of F in LDS first, coalesced, and gathering from there is
  // bit-exact and changes nothing -- 25.340 against 25.322 ms per launch on the same memory)
  const float R = f[k];
  const float un = del * R - ub[0] * (del + (-1.f));
  const float dE = f[k + m] - ub[1];
  const float dW = f[k - m] - ub[2];
  const float dN = f[k + 1] - ub[3];
  const float dS = f[k - 1] - ub[4];
  const float g = 0.5f * (wnx * (dE - dW) + wny * (dN - dS));
  return un + del1 * g;
}
__device__ inline void forcing_band(const StepParams &P, float *fx, float *fy, float *stage, float dphi1,
                                    float dphi2) {
  // stage: LDS, nband_x + nband_y floats (the host checks that the region holds them)
  const int m = uni(P.m), nbx = uni(P.nband_x), nby = uni(P.nband_y);
  const int *bandx = G(P.band_x), *bandy = G(P.band_y);
  const int tid = threadIdx.x;
  const int nmax = nbx > nby ? nbx : nby;
  const bool tabs = uni(P.band_stride) != 0 && !(AFC_DBG_OF(P) & 32);  // dbg bit 5: the gather form
  const float *tx = uni(G(P.band_tab_x)), *ty = uni(G(P.band_tab_y));
  const int stride = uni(P.band_stride), nfx = uni(P.nflag_x), nfy = uni(P.nflag_y);
  const float dt = uni(P.dt);
  lds_barrier();  // the region's previous user (the advection rings) is done
  for (int base = 0; base < nmax; base += 2 * kThreads) {
    const int q0 = base + tid, q1 = q0 + kThreads;
    const bool x0 = q0 < nbx, x1 = q1 < nbx, y0 = q0 < nby, y1 = q1 < nby;
    const int kx0 = x0 ? bandx[q0] : m + 1, kx1 = x1 ? bandx[q1] : m + 1;  // an idle slot reads (and ignores) cell [1][1]
    const int ky0 = y0 ? bandy[q0] : m + 1, ky1 = y1 ? bandy[q1] : m + 1;
    float ux0, ux1, uy0, uy1;
    if (tabs) {  // (idle slots read slot 0 of the table)
      ux0 = forcing_cell_tab<false>(tx, stride, x0 ? q0 : 0, x0 && q0 < nfx, fx, kx0, m, dphi1, dphi2, dt);
      ux1 = forcing_cell_tab<false>(tx, stride, x1 ? q1 : 0, x1 && q1 < nfx, fx, kx1, m, dphi1, dphi2, dt);
      uy0 = forcing_cell_tab<true>(ty, stride, y0 ? q0 : 0, y0 && q0 < nfy, fy, ky0, m, dphi1, dphi2, dt);
      uy1 = forcing_cell_tab<true>(ty, stride, y1 ? q1 : 0, y1 && q1 < nfy, fy, ky1, m, dphi1, dphi2, dt);
    } else {
      ux0 = forcing_x_cell(P, fx, kx0, m, dphi1, dphi2), ux1 = forcing_x_cell(P, fx, kx1, m, dphi1, dphi2);
      uy0 = forcing_y_cell(P, fy, ky0, m, dphi1, dphi2), uy1 = forcing_y_cell(P, fy, ky1, m, dphi1, dphi2);
    }
    if (x0) stage[q0] = ux0;
    if (x1) stage[q1] = ux1;
    if (y0) stage[nbx + q0] = uy0;
    if (y1) stage[nbx + q1] = uy1;
  }
  __syncthreads();  // every gather has been consumed (the scatter below overwrites what neighbours gathered)
  for (int q = tid; q < nbx; q += kThreads) fx[bandx[q]] = stage[q];
  for (int q = tid; q < nby; q += kThreads) fy[bandy[q]] = stage[nbx + q];
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// MG.smooth(4) without the trailing increment (MG.pde:80-89): d = r*inv, then four in-place
// lexicographic Gauss-Seidel sweeps, in the reference's exact dependency order.
//
// One wave marches a strip of 16 rows along j.  Lane = 16*k + c: sweep k (0..3), row
// i = I0 - k + c (the strip of sweep k is shifted up by k rows so that sweep k never overtakes
// sweep k-1); at step tau the lane updates column j = tau - c - kLag*k + 1.  The four neighbours of
// cell (i,j) in sweep k:
//   d[i-1][j] (sweep k)   lane (k, c-1), one step earlier -> DPP row_shr:1   (c == 0: the row above, finished
//                                                                             by the previous strip, in the ring)
//   d[i][j-1] (sweep k)   this lane, one step earlier     -> register
//   d[i+1][j] (sweep k-1) | stored IN PLACE in the d ring by the previous sweep kLag steps earlier (sweep 0: d0 =
//   d[i][j+1] (sweep k-1) | r*inv from the fill) and overwritten by this very sweep one step later
// which is what the sequential loops see, so every cell gets the reference's bits.  r and d live in 24-row LDS rings
// that all four waves fill/drain with coalesced accesses; out-of-domain neighbours read as 0 (ghost d is 0 on the
// finest level and multiplies a zero face coefficient on the coarse ones).
// Coefficients: on strips that do not touch the bodies they are synthesised from the level's
// analytic description (LevelDesc), otherwise loaded per cell.
// (Round 1 exchanged the sweeps through registers -- lane-16 by two gfx950 row swaps, 8 of its 22 vector
// instructions per step, one step of lag -- and kept the boundary rows of a strip in separate carry rows; the
// in-place form below issues 10 vector + 2.5 LDS instructions per step and runs the finest level's march at 0.55 of
// that time.  DESIGN.md section 3 has the stages and what bounds it now.)
// ---------------------------------------------------------------------------------------------
constexpr int kRing = 24;

// LDS layout used by the smoother (floats, PW = ring_pitch(m)):
//   [kPad][r ring 24*PW][kPad][d ring 24*PW][kPad][zero row PW][kPad][dummy row PW]
#ifndef AFC_KPAD
#define AFC_KPAD 36
#endif
// Ring row pitch: at least m + 34 floats (the last lanes of a strip run up to column m_+35) and == 12 (mod 32): a lane
// is one row down and one column back from its neighbour, sweep k+1 one row up and four columns back from sweep k, so
// with pitch == 12 the 32 lanes of a half-wave hit 32 different banks (pitch 166 = m + 36 on the 256x128 grid made the
// two sweeps of a half-wave collide on 14 of 16 banks: every LDS access of the march took twice its cycles, which is
// what kept a second marching wave per workgroup from paying -- tools/march_bench.hip).
__device__ __host__ __forceinline__ int ring_pitch(int m) {
#ifdef AFC_KPAD_FORCE  // tools/march_bench.hip only: a fixed pad (36 = the pitch of rounds 1-2) for comparison
  return m + AFC_KPAD_FORCE;
#endif
  const int lo = m + 34;
  return lo + ((12 - lo) & 31);
}
// Strip height of the FINEST level's up-leg by kernel instantiation (columns per lane): both rings of a 16-row strip
// (24 ring rows of ring_pitch(m) floats each) fit a workgroup's 40 KB share of the LDS up to 128 interior columns; wider
// grids take shorter strips -- 12 rows at 192 columns, 8 at 256, lanes c >= H of every sweep idle -- so that four
// workgroups per CU stay resident (as 16-row strips the 512x256 grid ran two workgroups per CU, 384x192 three).
__device__ __host__ constexpr int up_strip_h(int cpl) { return cpl <= 2 ? 16 : (cpl == 3 ? 12 : 8); }
// ring rows for strips of H rows: the march has rows I0-4 .. I0+H in flight and the exit reads one row further up
__device__ __host__ constexpr int up_ring_rows(int H) { return H == 16 ? 24 : H + 6; }
__device__ __host__ __forceinline__ int gs_lds_floats(int m, int H = 16) {
  return AFC_KPAD + (2 * up_ring_rows(H) + 2) * ring_pitch(m) + 3 * AFC_KPAD + 8 + 96 /* kPipeWords */;
}

// analytic coefficients of an interior cell (LevelDesc) -- used on rows without special cells
__device__ __forceinline__ void coef_analytic(const LevelDesc &D, const DescU &du, int i, int j, int n_, int m_,
                                              float &lw, float &le, float &ls, float &ln, float &iv) {
  if (!du.zb) {  // finest level: boundary faces keep c = dt, inv is the same everywhere
    lw = le = ls = ln = du.cu;
    iv = du.iv0;
    return;
  }
  const float b = 0.f;
  lw = (i == 1) ? b : du.cu;
  le = (i == n_) ? b : du.cu;
  ls = (j == 1) ? b : du.cu;
  ln = (j == m_) ? b : du.cu;
  iv = D.invtab[3 * ((i == 1) ? 1 : ((i == n_) ? 2 : 0)) + ((j == 1) ? 1 : ((j == m_) ? 2 : 0))];
}

// Ring rows are stored with pitch PW = ring_pitch(m) >= m + 34; the trailing floats (and a kPad front pad of
// each array) only ever receive zeros.  Running pointers are never clamped: a lane that is before
// the first / past the last column of its row reads a pad, a lane whose row lies outside the grid
// reads the all-zero row `zrow` and writes to `dummy`.  One unsigned compare per step forces the
// result of such a lane to 0 (which is also what its neighbours must see: ghost d = 0); its writes
// are unconditional -- zeros into a pad or the dummy row.
constexpr int kPad = AFC_KPAD;  // >= 34: the last lanes run (and write zeros) up to column m_+35

// The march.  EVERY sweep writes its result in place (as the sequential loops do) and reads d[i+1][j], d[i][j+1] of the
// previous sweep from LDS; consecutive sweeps follow each other at kLag = 4 steps.  Right after steps e, e+1 of a
// 4-step chunk have stored their values (one ds_write2_b32), the operands of the same two steps of the NEXT chunk are
// fetched (ds_read2_b32): what they need was stored exactly kLag steps before their use, i.e. by the store just issued
// (same wave: LDS operations execute in order), and the fetch has two to three steps to land.
constexpr int kLag = 4;
// du: the descriptor's scalars where the caller already holds them in scalar registers (mg_up_smooth) -- used by the
// instantiation for levels without zeroed boundary faces (the finest level), whose inverse diagonal is the same in every
// cell: its march then starts without a read of the descriptor.  (Fetching the whole invtab once per level call for the
// other levels as well measured +0.6 %: eleven more live scalars across the strip loop cost more than the L1 hits at the
// head of each strip.)
// Hand-over words of the pipelined up-leg (mg_up_pipe): the marching wave publishes the steps it has completed (one
// word per lane, so that the store needs no lane mask; readers look at lane 0's), the helper waves publish per column
// group that a strip's rows have been filled (ff) / have left (xf).  A word stored after data by the same wave is a valid
// flag: the LDS operations of a wave execute in order.
struct PipeSync {
  unsigned mp;       // LDS byte address of THIS lane's progress word
  const lds_i *ff;   // fill flags, one per column group: >= s once the rows of strip s are in the rings
  int sbase, s, Q;   // sbase = s << 10 (progress is published as sbase + steps), Q = column groups per row
};
// the marching wave waits until the fill flags lo .. hi-1 have reached `need` (lane q looks at flag q)
__device__ __forceinline__ void pipe_wait_fill(const lds_i *ff, int lo, int hi, int need) {
  const int l = threadIdx.x & 63;
  const bool mine = (l >= lo) && (l < hi);
  for (;;) {
    const int v = *(volatile const lds_i *)(ff + (l & 15));
    if (__builtin_amdgcn_ballot_w64(mine && v < need) == 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");  // nothing that reads the rings moves in front of the wait
}
// a helper wave waits until a (wave-uniform) word has reached `need`
__device__ __forceinline__ void pipe_wait_word(const lds_i *w, int need) {
#ifndef AFC_PIPE_SLEEP
#define AFC_PIPE_SLEEP 4  // x 64 clocks between two looks at a flag: 1 / 4 / 8 / 16 measured 24.08 / 24.00 / 24.00 / 24.01 ms per launch at
#endif                    // 1024 envs, 16.43 / 16.38 / 16.50 / 16.58 at 512, 11.52 / 11.50 / 11.54 / 11.69 for one env (same memory)
  while (uni(*(volatile const lds_i *)w) < need) __builtin_amdgcn_s_sleep(AFC_PIPE_SLEEP);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pipe_post_word(lds_i *w, int v) {
  asm volatile("" ::: "memory");  // after everything this wave has read / written in the rings
  if ((threadIdx.x & 63) == 0) *(volatile lds_i *)w = v;
}
constexpr int kPipeHeadGroups = 5;  // column groups (of 16 columns) the first 64 steps of a march can touch
template <bool FAST, bool ZB = true, int H = 16, bool PIPE = false>
__device__ __forceinline__ void gs_march3(const Lv &L, const float *rr, float *dr, float *zrow, float *dummy, int I0,
                                          const DescU *du = nullptr, const PipeSync *ps = nullptr) {
  constexpr int NR = up_ring_rows(H);  // (= kRing for H = 16)
  const int m = L.m, n_ = L.n - 2, m_ = L.m - 2, PW = ring_pitch(m);
  const int lane = threadIdx.x & 63, k = lane >> 4, c = lane & 15;
  const int i = I0 - k + c;
  const bool row_ok = (i >= 1) && (i <= n_) && (H == 16 || c < H);  // strips of H < 16 rows: lanes c >= H have no row
  const int ic = i < 0 ? 0 : (i > n_ ? n_ : i);  // clamped for addressing only
  const int slot = (ic % NR) * PW, slotE = ((ic + 1) % NR) * PW;
  const LevelDesc &D = *L.desc;
  float iv0 = 0.f, iv1 = 0.f, iv2 = 0.f;
  const bool know = FAST && !ZB && du != nullptr;  // no zeroed boundary faces: inv = invtab[0] everywhere
  {  // row-dependent part of the analytic inverse diagonal (FAST = false: for the steps outside the special columns)
    if (know) {
      iv0 = du->iv0;
    } else {
      const int rc = (i == 1) ? 1 : ((i == n_) ? 2 : 0);
      iv0 = D.invtab[3 * rc + 0];
      iv1 = D.invtab[3 * rc + 1];
      iv2 = D.invtab[3 * rc + 2];
    }
  }
  const float cu = du ? du->cu : uni(D.cu);
  const bool zb = ZB && (du ? du->zb : uni(D.zero_bnd)) != 0;  // ZB = false: the caller knows the level has no zeroed boundary faces
  // the last lane (sweep 3, row 15) reaches column m_ at step m_ + 15 + 3*kLag - 1: whole 4-step chunks, two per loop
  // iteration and possibly one more at the end
  const int steps = (m_ + (H - 1) + 3 * kLag + 3) & ~3, steps8 = steps & ~7;
  const unsigned m_eff = row_ok ? (unsigned)m_ : 0u;
  // Running LDS pointers: at step tau this lane is at column j = tau + jo.  Lanes before the first / past the last
  // column of their row read a pad and store zeros into it (kPad >= 34 covers the furthest one), lanes whose row lies
  // outside the grid read the zero row and write to the dummy row.
  const int jo = 1 - c - kLag * k;
  const float *p_rv = (row_ok ? rr + slot : zrow) + jo;
  const bool rowW_ok = (i - 1 >= 1) && (i - 1 <= n_);
  const float *p_cW = (rowW_ok ? dr + ((i - 1) % NR) * PW : zrow) + jo;  // d[i-1][j]: only lanes c == 0 use it
  const float *p_cN = (row_ok ? dr + slot : zrow) + 1 + jo;                 // d[i][j+1] of the previous sweep
  const float *p_dE = ((row_ok && i + 1 <= n_) ? dr + slotE : zrow) + jo;   // d[i+1][j] of the previous sweep
  float *w_d = (row_ok ? dr + slot : dummy) + jo;
  // The five streams of a lane as LDS byte addresses that advance by one opaque add per loop iteration (8 steps): every
  // access below is `base + immediate`, two consecutive steps per instruction (ds_read2_b32 / ds_write2_b32).
  unsigned q_rv = lds_addr(p_rv), q_cW = lds_addr(p_cW), q_cN = lds_addr(p_cN), q_dE = lds_addr(p_dE), q_wd = lds_addr(w_d);
#define LD2(q, o) (*(const lds_f2 *)(uintptr_t)((q) + 4u * (unsigned)(o)))
#define ST2(q, o, v) (*(lds_f2 *)(uintptr_t)((q) + 4u * (unsigned)(o)) = (v))
  float aR[4], aW[4], aN[4], aE[4], bR[4], bW[4], bN[4], bE[4];
#pragma unroll
  for (int h = 0; h < 4; h += 2) {
    const f2u r2 = LD2(q_rv, h), w2 = LD2(q_cW, h), n2 = LD2(q_cN, h), e2 = LD2(q_dE, h);
    aR[h] = r2.x, aR[h + 1] = r2.y, aW[h] = w2.x, aW[h + 1] = w2.y;
    aN[h] = n2.x, aN[h + 1] = n2.y, aE[h] = e2.x, aE[h + 1] = e2.y;
  }
  F4 a_lw, a_le, a_ls, a_iv, b_lw, b_le, b_ls, b_iv;
  float a_ln4 = 0.f, b_ln4 = 0.f;
  const int cbase = ic * m + jo;  // + tau = flat index of this lane's cell at step tau
#define GS_CHUNK(TAU, lw_, le_, ls_, iv_, ln4_) \
  {                                             \
    int kk_ = cbase + (TAU);                    \
    kk_ = kk_ < 0 ? 0 : kk_;                    \
    lw_ = load4u(L.lx + kk_);                   \
    le_ = load4u(L.lx + kk_ + m);               \
    ls_ = load4u(L.ly + kk_);                   \
    iv_ = load4u(L.inv + kk_);                  \
    ln4_ = L.ly[kk_ + 4];                       \
  }
  // FAST = false (a strip with special rows): only the steps at which some lane stands on one of the level's special
  // COLUMNS [sp_j0, sp_j1] take the coefficient loads -- the lanes of a step cover columns tau - 26 .. tau + 1, an 8-step
  // iteration tau0 - 26 .. tau0 + 8 -- and everything before and after runs the FAST code (same bits: outside the special
  // cells the arrays hold exactly the analytic values, which is what lets whole strips take that path).  Every step of
  // such a strip through the loads cost 140 ns against 45: the loads are issued one chunk (~150 ns) ahead and come from
  // L2.  The window's first chunk is fetched here, at the head of the march.
  int g_lo = 0, g_end = 0;
  if (!FAST) {
    const int sj0 = uni(D.sp_j0), sj1 = uni(D.sp_j1);
    g_lo = (sj0 - 1) & ~7;
    g_lo = g_lo < 0 ? 0 : (g_lo > steps8 ? steps8 : g_lo);  // (a level without special cells: an empty window at the end)
    g_end = ((sj1 + 26) & ~7) + 8;
    GS_CHUNK(g_lo, a_lw, a_le, a_ls, a_iv, a_ln4)
  }
  int pv = 0;
  unsigned q_mp = 0;
  if (PIPE) {
    pv = ps->sbase;
    q_mp = ps->mp;
  }
  float res = 0.f;
  // one 4-step chunk on operand set (cR,cW,cN,cE / c_*), fetching set (nR,nW,nN,nE / n_*) for the next one, two steps
  // per LDS instruction: after steps e, e+1 have been stored, the operands of the same two steps of the NEXT chunk are
  // fetched (what they need was stored kLag steps before its use, i.e. by the store just issued).  O = offset of the
  // chunk from the iteration's base addresses; STEADY: no activity mask, interior inverse diagonal
#define GS3_STEPS(GEN, STEADY, T0, O, cR, cW, cN, cE, nR, nW, nN, nE, c_lw, c_le, c_ls, c_iv, c_ln4)                         \
  _Pragma("unroll") for (int h = 0; h < 4; h += 2) {                                                                   \
    f2u out;                                                                                                           \
    _Pragma("unroll") for (int e = h; e < h + 2; ++e) {                                                                \
      float lw, le, ls, ln, iv;                                                                                        \
      if (!(GEN)) {                                                                                                    \
        /* every out-of-domain neighbour reads as d = 0, so a boundary face may keep the uniform coefficient:       */ \
        /* cu*0 == 0*ghost.  Only inv knows about the boundary (coarse levels).                                     */ \
        lw = le = ls = ln = cu;                                                                                        \
        iv = iv0;                                                                                                      \
        if (zb && !(STEADY)) {                                                                                         \
          const int j = (T0) + jo + e;                                                                                 \
          iv = (j == 1) ? iv1 : ((j == m_) ? iv2 : iv0);                                                               \
        }                                                                                                              \
      } else {                                                                                                         \
        lw = c_lw.v[e];                                                                                                \
        le = c_le.v[e];                                                                                                \
        ls = c_ls.v[e];                                                                                                \
        ln = (e < 3) ? c_ls.v[(e + 1) & 3] : c_ln4;                                                                    \
        iv = c_iv.v[e];                                                                                                \
      }                                                                                                                \
      /* (k, c-1) holds sweep k at (i-1, j) one step earlier: DPP row_shr:1; c == 0 keeps cW = the row above, in the ring */ \
      const float t = gs_core<!(GEN), !(GEN) && (STEADY)>(res, cW[e], cE[e], cN[e], cR[e], lw, le, ls, ln, iv);        \
      const float v = (!(GEN) && (STEADY)) ? t : -t * iv; /* MG.pde:84-88 */                                           \
      res = ((STEADY) || (unsigned)((T0) + jo + e - 1) < m_eff) ? v : 0.f;                                             \
      if (e == h) out.x = res; else out.y = res;                                                                       \
    }                                                                                                                  \
    ST2(q_wd, (O) + h, out);                                                                                           \
    const f2u r2 = LD2(q_rv, (O) + 4 + h), w2 = LD2(q_cW, (O) + 4 + h);                                                \
    const f2u n2 = LD2(q_cN, (O) + 4 + h), e2 = LD2(q_dE, (O) + 4 + h);                                                \
    nR[h] = r2.x, nR[h + 1] = r2.y, nW[h] = w2.x, nW[h + 1] = w2.y;                                                    \
    nN[h] = n2.x, nN[h + 1] = n2.y, nE[h] = e2.x, nE[h + 1] = e2.y;                                                    \
  }
#define GS3_CHUNK2(GEN, STEADY, T0, O, cR, cW, cN, cE, nR, nW, nN, nE, c_lw, c_le, c_ls, c_iv, c_ln4, n_lw, n_le, n_ls, \
                   n_iv, n_ln4)                                                                                        \
  {                                                                                                                    \
    if (GEN) GS_CHUNK((T0) + 4, n_lw, n_le, n_ls, n_iv, n_ln4)                                                         \
    GS3_STEPS(GEN, STEADY, T0, O, cR, cW, cN, cE, nR, nW, nN, nE, c_lw, c_le, c_ls, c_iv, c_ln4)                       \
  }
#define GS3_ITER(GEN, STEADY)                                                                                          \
  {                                                                                                                    \
    GS3_CHUNK2(GEN, STEADY, tau0, 0, aR, aW, aN, aE, bR, bW, bN, bE, a_lw, a_le, a_ls, a_iv, a_ln4, b_lw, b_le, b_ls,  \
               b_iv, b_ln4)                                                                                            \
    GS3_CHUNK2(GEN, STEADY, tau0 + 4, 4, bR, bW, bN, bE, aR, aW, aN, aE, b_lw, b_le, b_ls, b_iv, b_ln4, a_lw, a_le,    \
               a_ls, a_iv, a_ln4)                                                                                      \
    /* opaque to the optimiser: the addresses are NOT re-derived from tau0 (one VALU add per access otherwise) */     \
    asm volatile("v_add_u32 %0, 32, %0\n\tv_add_u32 %1, 32, %1\n\tv_add_u32 %2, 32, %2\n\tv_add_u32 %3, 32, %3\n\t"  \
                 "v_add_u32 %4, 32, %4"                                                                                \
                 : "+v"(q_rv), "+v"(q_cW), "+v"(q_cN), "+v"(q_dE), "+v"(q_wd));                                        \
    if (PIPE) { /* steps < tau0 + 8 are stored: tell the helper waves (after the stores: LDS executes in order) */   \
      pv += 8;                                                                                                         \
      *(volatile lds_i *)(uintptr_t)q_mp = pv;                                                                         \
    }                                                                                                                  \
  }
  // three loops, each one basic block per 8 steps: ramp-up, the steady stretch (strips whose 19 rows all lie inside
  // the grid: every lane at an interior column 2..m_-1, no activity mask, interior inverse diagonal), ramp-down
  // (lanes without a row -- H < 16 -- run the steady stretch on the zero row: they stay at zero, their stores go to the dummy row)
  const bool all_rows = (I0 - 3 >= 1) && (I0 + H - 1 <= n_);
  const int s_end = (all_rows && m_ >= 9) ? (((m_ - 9) & ~7) + 8) : 0;  // steady: 32 <= tau0 <= m_ - 9
  int tau0 = 0;
  // straight-line loops, one basic block per 8 steps each (a single loop that picks its body per iteration keeps every
  // body's operands live everywhere: +100 B of scratch per lane, reloads inside the march, +9.6 % per launch)
#define GS3_FAST_RANGE(T_END)                                                  \
  for (; tau0 < (T_END) && tau0 < 32; tau0 += 8) GS3_ITER(false, false)        \
  for (; tau0 < (T_END) && tau0 < s_end; tau0 += 8) GS3_ITER(false, true)      \
  for (; tau0 < (T_END); tau0 += 8) GS3_ITER(false, false)
  if (FAST) {
    if (PIPE && ps->Q > kPipeHeadGroups) {  // the columns past the head groups: their fill may still be on its way
      const int t64 = steps8 < 64 ? steps8 : 64;
      GS3_FAST_RANGE(t64)
      pipe_wait_fill(ps->ff, kPipeHeadGroups, ps->Q, ps->s);
    }
    GS3_FAST_RANGE(steps8)
    if (tau0 < steps)
      GS3_CHUNK2(false, false, tau0, 0, aR, aW, aN, aE, bR, bW, bN, bE, a_lw, a_le, a_ls, a_iv, a_ln4, b_lw, b_le, b_ls, b_iv,
                 b_ln4)
  } else {
    const int w0 = g_lo < steps8 ? g_lo : steps8, w1 = g_end < steps8 ? g_end : steps8;
    GS3_FAST_RANGE(w0)
    for (; tau0 < w1; tau0 += 8) GS3_ITER(true, false)
    GS3_FAST_RANGE(steps8)
    if (tau0 < steps) {
      if (tau0 >= g_lo && tau0 < g_end)
        GS3_CHUNK2(true, false, tau0, 0, aR, aW, aN, aE, bR, bW, bN, bE, a_lw, a_le, a_ls, a_iv, a_ln4, b_lw, b_le, b_ls, b_iv,
                   b_ln4)
      else
        GS3_CHUNK2(false, false, tau0, 0, aR, aW, aN, aE, bR, bW, bN, bE, a_lw, a_le, a_ls, a_iv, a_ln4, b_lw, b_le, b_ls, b_iv,
                   b_ln4)
    }
  }
#undef GS3_FAST_RANGE
#undef GS3_ITER
#undef LD2
#undef ST2
#undef GS3_CHUNK2
#undef GS3_STEPS
#undef GS_CHUNK
}

// prolongate(coarse.x) at a (possibly ghost) cell of the fine level (MG.pde:139-152 incl. setBC)
__device__ __forceinline__ float prolong_at(const float *__restrict__ xc, int mc, int i, int j, int n_, int m_) {
  const int ci = i < 1 ? 1 : (i > n_ ? n_ : i), cj = j < 1 ? 1 : (j > m_ ? m_ : j);
  return xc[((ci - 1) / 2 + 1) * mc + ((cj - 1) / 2 + 1)];
}

// ---- the two per-chunk bodies of the up-leg (4 columns of one row), split into "issue the global
// loads" and "compute" so that the finest level can issue the loads while the march is running ----
struct UpGeom {
  const Lv *F, *Cc;
  const LevelDesc *D;
  int n, m, n_, m_, mc, PW, CPR;
  float *rr, *dr;
  DescU du;
  int NR = kRing;        // rows in the LDS rings (24 for the strip march, H + 6 for the multi-wave smoother)
  bool store_r = true;   // false: r -= A d is formed (for the norms) but not written back
  bool update_x = true;  // false: replay that only materialises r
  bool al16 = false;       // ring column 1 sits on a 16-byte boundary: the chunk accesses of fill / exit are 128-bit
};
struct FillIn {  // r of the chunk + the coarse x values its prolongation stencil touches
  F4 r;
  float a[4], w1, w2, e1, e2;
};
struct ExitIn {  // x of the chunk, the coarse x under it, the two x ghosts a boundary chunk mirrors into
  F4 x;
  float a1, a2, gl, gh;
};

__device__ __forceinline__ FillIn fill_load(const UpGeom &U, const bool HASC, int g, int c4) {
  FillIn in;
#pragma unroll
  for (int e = 0; e < 4; ++e) in.r.v[e] = in.a[e] = 0.f;
  in.w1 = in.w2 = in.e1 = in.e2 = 0.f;
  if (g <= U.n_) {
    in.r = load4u(U.F->r + g * U.m + 1 + 4 * c4);
    if (HASC) {
      const int gw = (g - 1 < 1) ? 1 : g - 1, ge = (g + 1 > U.n_) ? U.n_ : g + 1;
      const float *xa = U.Cc->x + ((g - 1) / 2 + 1) * U.mc + 2 * c4;  // coarse cols 2c4 .. 2c4+3
      const float *xw = U.Cc->x + ((gw - 1) / 2 + 1) * U.mc + 2 * c4;
      const float *xe = U.Cc->x + ((ge - 1) / 2 + 1) * U.mc + 2 * c4;
      // three 16-byte loads instead of eight dwords
      const F4 va = load4u(xa), vw = load4u(xw), ve = load4u(xe);
#pragma unroll
      for (int e = 0; e < 4; ++e) in.a[e] = va.v[e];
      in.w1 = vw.v[1];
      in.w2 = vw.v[2];
      in.e1 = ve.v[1];
      in.e2 = ve.v[2];
    }
  }
  return in;
}

// r (after the prolongation increment) and d0 = r*inv of one chunk -> rings
__device__ __forceinline__ void fill_apply(const UpGeom &U, const bool HASC, int g, int c4, const FillIn &in, bool sp) {
  const int m = U.m, n_ = U.n_, m_ = U.m_, CPR = U.CPR;
  const int j = 1 + 4 * c4;
  const int sl = (g % U.NR) * U.PW + j;
  F4 rv = in.r, dv;
#pragma unroll
  for (int e = 0; e < 4; ++e) dv.v[e] = 0.f;
  if (g <= n_) {
    const int kk = g * m + j;
    F4 lw, le, ls, iv;
    float ln4;
    if (sp) {
      lw = load4u(U.F->lx + kk);
      le = load4u(U.F->lx + kk + m);
      ls = load4u(U.F->ly + kk);
      ln4 = U.F->ly[kk + 4];
      iv = load4u(U.F->inv + kk);
    } else {
      float a0, a1, a2, a3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        coef_analytic(*U.D, U.du, g, j + e, n_, m_, a0, a1, a2, a3, iv.v[e]);
        lw.v[e] = a0;
        le.v[e] = a1;
        ls.v[e] = a2;
        if (e == 3) ln4 = a3;
      }
    }
    if (HASC) {  // r -= A * prolongate(coarse.x)   (PoissonMatrix.pde:53-64 order)
      const float a1 = in.a[1], a2 = in.a[2];
      const float a0 = (c4 == 0) ? a1 : in.a[0], a3 = (c4 == CPR - 1) ? a2 : in.a[3];
      const float w1 = in.w1, w2 = in.w2, e1 = in.e1, e2 = in.e2;
      const float dc[4] = {a1, a1, a2, a2}, dw[4] = {w1, w1, w2, w2}, de[4] = {e1, e1, e2, e2};
      const float ds[4] = {a0, a1, a1, a2}, dn[4] = {a1, a2, a2, a3};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lnv = (e < 3) ? ls.v[(e + 1) & 3] : ln4;
        const float diag = -(lw.v[e] + le.v[e] + ls.v[e] + lnv);
        rv.v[e] = rv.v[e] - (dc[e] * diag + dw[e] * lw.v[e] + de[e] * le.v[e] + ds[e] * ls.v[e] + dn[e] * lnv);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) dv.v[e] = rv.v[e] * iv.v[e];  // MG.pde:80
  }
  if (U.al16) {  // one ds_write_b128 per ring instead of four stride-4 dword stores (4-way bank conflicts)
    f4a r4, d4;
#pragma unroll
    for (int e = 0; e < 4; ++e) r4[e] = rv.v[e], d4[e] = dv.v[e];
    *(lds_f4 *)(lds_f *)(U.rr + sl) = r4;
    *(lds_f4 *)(lds_f *)(U.dr + sl) = d4;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      U.rr[sl + e] = rv.v[e];
      U.dr[sl + e] = dv.v[e];
    }
  }
  if (c4 == 0) U.dr[sl - 1] = 0.f;  // ghost columns of d0 read as zero
  if (c4 == CPR - 1) U.dr[sl + 4] = 0.f;
}

__device__ __forceinline__ ExitIn exit_load(const UpGeom &U, const bool HASC, int g, int c4) {
  ExitIn in;
  const int kk = g * U.m + 1 + 4 * c4;
  in.a1 = in.a2 = in.gl = in.gh = 0.f;
  if (!U.update_x) {
#pragma unroll
    for (int e = 0; e < 4; ++e) in.x.v[e] = 0.f;
    return in;
  }
  in.x = load4u(U.F->x + kk);
  if (HASC) {
    const float *xa = U.Cc->x + ((g - 1) / 2 + 1) * U.mc + 2 * c4;
    float a12[2];
    __builtin_memcpy(a12, xa + 1, 8);  // one 8-byte load
    in.a1 = a12[0];
    in.a2 = a12[1];
  }
  if (c4 == 0) in.gl = U.F->x[kk - 1];
  if (c4 == U.CPR - 1) in.gh = U.F->x[kk + 4];
  return in;
}

// d.setBC(); x += d (+ the ghosts that mirror the chunk); r -= A d   (MG.pde:90-97)
__device__ __forceinline__ void exit_apply(const UpGeom &U, const bool HASC, const bool need_r, int g, int c4,
                                           const ExitIn &in, bool sp, double &rr_part, double &xs_part, float &linf) {
  const int n = U.n, m = U.m, n_ = U.n_, m_ = U.m_, CPR = U.CPR, PW = U.PW;
  const Lv &F = *U.F;
  float *rr = U.rr, *dr = U.dr;
  const int j = 1 + 4 * c4, kk = g * m + j;
  const int gw = (g - 1 < 1) ? 1 : g - 1, ge = (g + 1 > n_) ? n_ : g + 1;
  const int sl = (g % U.NR) * PW + j;
  float dp[4] = {0.f, 0.f, 0.f, 0.f};
  if (HASC) {
    dp[0] = in.a1;
    dp[1] = in.a1;
    dp[2] = in.a2;
    dp[3] = in.a2;
  }
  float d6[6];  // d at columns j-1 .. j+4 with the setBC mirror
  if (U.al16) {
    const f4a dc4 = *(const lds_f4 *)(lds_f *)(dr + sl);
    d6[0] = dr[sl - 1];
#pragma unroll
    for (int e = 0; e < 4; ++e) d6[1 + e] = dc4[e];
    d6[5] = dr[sl + 4];
  } else {
#pragma unroll
    for (int b = 0; b < 6; ++b) d6[b] = dr[sl - 1 + b];
  }
  if (c4 == 0) d6[0] = d6[1];
  if (c4 == CPR - 1) d6[5] = d6[4];
  F4 xn;
  if (need_r) {
    const float *dwp = dr + (gw % U.NR) * PW + j, *dep = dr + (ge % U.NR) * PW + j;
    F4 lw, le, ls;
    float ln4;
    if (sp) {
      lw = load4u(F.lx + kk);
      le = load4u(F.lx + kk + m);
      ls = load4u(F.ly + kk);
      ln4 = F.ly[kk + 4];
    } else {
      float a0, a1, a2, a3, a4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        coef_analytic(*U.D, U.du, g, j + e, n_, m_, a0, a1, a2, a3, a4);
        lw.v[e] = a0;
        le.v[e] = a1;
        ls.v[e] = a2;
        if (e == 3) ln4 = a3;
      }
    }
    F4 rn;
    float dwv[4], dev[4], rov[4];
    if (U.al16) {
      const f4a w4 = *(const lds_f4 *)(lds_f *)dwp, e4 = *(const lds_f4 *)(lds_f *)dep;
      const f4a o4 = *(const lds_f4 *)(lds_f *)(rr + sl);
#pragma unroll
      for (int e = 0; e < 4; ++e) dwv[e] = w4[e], dev[e] = e4[e], rov[e] = o4[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) dwv[e] = dwp[e], dev[e] = dep[e], rov[e] = rr[sl + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lnv = (e < 3) ? ls.v[(e + 1) & 3] : ln4;
      const float diag = -(lw.v[e] + le.v[e] + ls.v[e] + lnv);
      const float d = d6[1 + e];
      const float rold = rov[e];
      rn.v[e] = rold - (d * diag + dwv[e] * lw.v[e] + dev[e] * le.v[e] + d6[e] * ls.v[e] + d6[2 + e] * lnv);
      const float pr = rn.v[e] * rn.v[e];
      rr_part += (double)pr;
      linf = jmax(linf, jabs(rn.v[e]));
    }
    if (U.store_r) store4u(F.r + kk, rn);
  }
  if (!U.update_x) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float xo = in.x.v[e];
    if (HASC) xo = xo + dp[e];
    xn.v[e] = xo + d6[1 + e];
    if (need_r) xs_part += (double)xn.v[e];
  }
  store4u(F.x + kk, xn);
  // ghosts mirror the adjacent interior cell (d.setBC(), Field.pde:211-224)
  if (c4 == 0) {
    float xg = in.gl;
    if (HASC) xg = xg + dp[0];
    F.x[kk - 1] = xg + d6[1];
  }
  if (c4 == CPR - 1) {
    float xg = in.gh;
    if (HASC) xg = xg + dp[3];
    F.x[kk + 4] = xg + d6[4];
  }
  if (g == 1 || g == n_) {
    const int kg = ((g == 1) ? 0 : (n - 1)) * m + j;
    const F4 xg4 = load4u(F.x + kg);
    F4 xgn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xg = xg4.v[e];
      if (HASC) xg = xg + dp[e];
      xgn.v[e] = xg + d6[1 + e];
    }
    store4u(F.x + kg, xgn);
    if (c4 == 0) {
      float xg = F.x[kg - 1];
      if (HASC) xg = xg + dp[0];
      F.x[kg - 1] = xg + d6[1];
    }
    if (c4 == CPR - 1) {
      float xg = F.x[kg + 4];
      if (HASC) xg = xg + dp[3];
      F.x[kg + 4] = xg + d6[4];
    }
  }
}

#include "bdim_v2_uplean.h"

// Strip geometry of the up-leg: a strip brings kStripH new rows into the rings; the smoother runs in stages whose row
// ranges are shifted up by one row each (stage s covers rows I0-s .. I0-s+kStripH-1), kStageShift = stages - 1.
//   reference order: 4 stages = the 4 sweeps, run by the march as one fused pass
//   perf mode:       8 stages = the 8 colour half-sweeps of 4 red-black sweeps
#if AFC_RB
constexpr int kStripH = 12, kStageShift = 7;
#else
constexpr int kStripH = 16, kStageShift = 3;
#endif

#if AFC_RB
// PERF MODE smoother: half-sweep h (0..7) of four red-black Gauss-Seidel sweeps on rows I0-h .. I0-h+kStripH-1 of the
// rings, by all threads: the cells with (i + j) & 1 == (h & 1) take
//   d = -(d[i-1][j]*lw + d[i+1][j]*le + d[i][j-1]*ls + d[i][j+1]*ln - r) * inv        (MG.pde:84-88)
// from their four neighbours of the other colour, which hold the values of half-sweep h-1 (rows below the strip of
// stage h were covered by stage h-1 one row further down; rows above it have not seen stage h+1 yet).  Out-of-domain
// neighbours read as 0, like in the march.  One LDS barrier per stage (the caller's).
__device__ inline void gs_rb_stage(const Lv &L, const float *rr, float *dr, int PW, int I0, int h, bool special,
                                   const DescU &du) {
  const int m = L.m, n_ = L.n - 2, m_ = L.m - 2;
  // slots of one colour per row: ceil(m_ / 2) -- on a level of ODD width (the coarsest levels of the 384x192 grid:
  // 3 and 1 columns) one colour has a cell more than the other; the slot past the last column is skipped below
  const int half = (m_ + 1) >> 1;
  const int c = h & 1;
  const LevelDesc &D = *L.desc;
  const bool pow2 = (half & (half - 1)) == 0;  // half >= 1: m_ >= 1 on every level
  const int sh = 31 - __builtin_clz(half);
  const float cu = du.cu;
  for (int q = threadIdx.x; q < kStripH * half; q += kThreads) {
    const int rw = pow2 ? (q >> sh) : (q / half), qq = q - rw * half;
    const int i = I0 - h + rw;
    if (i < 1 || i > n_) continue;
    const int j = 1 + 2 * qq + ((c + i + 1) & 1);
    if (j > m_) continue;
    const int so = (i % kRing) * PW + j;
    const float dW = (i > 1) ? dr[((i - 1) % kRing) * PW + j] : 0.f;
    const float dE = (i < n_) ? dr[((i + 1) % kRing) * PW + j] : 0.f;
    const float dS = dr[so - 1], dN = dr[so + 1];  // ghost columns of the ring hold 0
    float v;
    if (special) {
      const int k = i * m + j;
      v = -(dW * L.lx[k] + dE * L.lx[k + m] + dS * L.ly[k] + dN * L.ly[k + 1] - rr[so]) * L.inv[k];
    } else {
      // away from the bodies every face coefficient is cu (a zeroed boundary face multiplies a neighbour that reads
      // as 0 anyway); only the inverse diagonal knows about the boundary (coarse levels)
      float iv = du.iv0;
      if (du.zb) iv = D.invtab[3 * ((i == 1) ? 1 : ((i == n_) ? 2 : 0)) + ((j == 1) ? 1 : ((j == m_) ? 2 : 0))];
      v = -(dW * cu + dE * cu + dS * cu + dN * cu - rr[so]) * iv;
    }
    dr[so] = v;
  }
}
#endif

// The three workgroup reductions that end the finest up-leg (r.r and sum(x) in double, max|r|) in ONE pass: the tree of
// block_sum_d / block_max_f (t += t + s for s = 128 ... 1) with the same operands in the same order -- so the same
// bits --, its two levels that cross waves through LDS and the six inside wave 0 through lane shuffles: 5 workgroup
// barriers instead of 30.  stage: >= 2.6 KB of LDS nobody else is using.
__device__ __forceinline__ void block_reduce3(double &a, double &b, float &c, float *stage) {
  const int t = threadIdx.x;
  double *sa = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(stage) + 7) & ~(uintptr_t)7), *sb = sa + 128;
  float *sc = reinterpret_cast<float *>(sb + 128);
  if (t >= 128) sa[t - 128] = a, sb[t - 128] = b, sc[t - 128] = c;
  __syncthreads();
  if (t < 128) a += sa[t], b += sb[t], c = jmax(c, sc[t]);
  __syncthreads();
  if (t >= 64 && t < 128) sa[t - 64] = a, sb[t - 64] = b, sc[t - 64] = c;
  __syncthreads();
  if (t < 64) {
    a += sa[t], b += sb[t], c = jmax(c, sc[t]);
#pragma unroll
    for (int s_ = 32; s_ > 0; s_ >>= 1) {
      a += __shfl_down(a, s_);
      b += __shfl_down(b, s_);
      c = jmax(c, __shfl_down(c, s_));
    }
    if (t == 0) sa[0] = a, sb[0] = b, sc[0] = c;
  }
  __syncthreads();
  a = sa[0], b = sb[0], c = sc[0];
  __syncthreads();
}

// One call = everything a level does on the way up (MG.pde:73-76 seen from the level itself):
//   HASC:  d = prolongate(coarse.x); increment()            [x += d, r -= A d]
//   then   smooth(4): d = r*inv; 4 Gauss-Seidel sweeps; d.setBC(); increment()
// streamed strip by strip: r (and p on the finest level) cross HBM once, d never leaves LDS.
// need_r: also form/store r -= A d of the last increment and return r.r / L_inf (finest level).
// HS: rows per strip (16; 12 / 8 on the finest level of the grids wider than 128 columns: up_strip_h)
template <int HS = 16>
__device__ AFC_UP_ATTR void mg_up_smooth(const bool HASC, const Lv &F, const Lv &Cc, bool need_r, float *lds, double *sm,
                                    float *rr_out,
                                    float *linf_out, ProfT &pf, float *xsum_out = nullptr, const int rmode = 0) {
  // rmode (finest level, chunked path): 0 = r written back; 1 = r not written back (the caller replays
  // with rmode 2 in the rare case a second V-cycle needs it); 2 = replay: same d, only r is stored
  const int n = F.n, m = F.m, n_ = n - 2, m_ = m - 2;
  const int PW = ring_pitch(m);
#if AFC_RB
  constexpr int H = kStripH, NR = kRing;
#else
  constexpr int H = HS, NR = up_ring_rows(HS);
#endif
  const bool mk = need_r || (AFC_DIAG && ((F.desc->dbg_copy >> 20) == n));  // diagnostics: phase marks for the level with n rows
  // the whole layout sits 3 floats in: ring column 1 (the first interior one) falls on a 16-byte boundary, so a thread's
  // 4-column chunk is one 128-bit LDS access in fill / exit (PW, kPad and NR * PW are multiples of 4 floats)
  float *rr = lds + kPad + 3 + (((16u - (lds_addr(lds) & 15u)) & 15u) >> 2), *dr = rr + NR * PW + kPad, *zrow = dr + NR * PW + kPad, *dummy = zrow + PW + kPad;
  const int tid = threadIdx.x;
  const LevelDesc &D = *F.desc;
  const int mc = Cc.m;
  const int CPR = m_ >> 2;
  const bool chunked = ((m_ & 3) == 0);
  // (integer divisions cost ~30 dependent instructions each, here in front of a barrier every wave waits at: the flat
  // stepping is only formed for the levels that use it, the chunk split by a shift where the chunk count is a power of two)
  const int rstep = chunked ? 0 : kThreads / m, cstep = kThreads - rstep * m;  // flat (row, col) stepping by kThreads
  const int r0 = chunked ? 0 : tid / m, c0 = tid - r0 * m;
  double rr_part = 0, xs_part = 0;
  float linf = 0.f;
  const DescU du = desc_u(D);
  UpGeom U{&F, &Cc, &D, n, m, n_, m_, mc, PW, CPR, rr, dr, du};
  U.NR = NR;
  U.al16 = true;
  U.store_r = (rmode != 1);
  U.update_x = (rmode != 2);
  // a thread keeps its 4-column chunk and walks down the rows of its group: no integer division in the strip loops
  const bool cpr_p2 = chunked && (CPR & (CPR - 1)) == 0;
  const int cpr_sh = cpr_p2 ? __builtin_ctz((unsigned)CPR) : 0;
  const int ngrp = chunked ? (cpr_p2 ? (kThreads >> cpr_sh) : kThreads / CPR) : 1;
  const int rg = chunked ? (cpr_p2 ? (tid >> cpr_sh) : tid / CPR) : 0, c4 = chunked ? (tid - rg * CPR) : 0;
  const bool grp_on = chunked && (rg < ngrp);
  __syncthreads();
  // Only the zero row has to hold zeros: ring interiors and the ghost columns of d are written by the fills before
  // they are read, and what a lane reads from a pad (before the first / past the last column of its row) never
  // reaches a result -- such a lane's value is forced to 0 by the activity test.  (The region is shared with the
  // other phases' tiles, so this is redone on every call.)
  for (int q = tid; q < PW + 2 * kPad; q += kThreads) zrow[q - kPad] = 0.f;
  lds_barrier();
  int loaded = 0;  // rows 1..loaded are (or were) in the rings
  for (int I0 = 1; I0 - kStageShift <= n_; I0 += H) {
    // ---- fill rows loaded+1 .. I0+H: r (after the prolongation increment) and d0 = r*inv ----
    const int hi = I0 + H;
    const bool sp_fill = (hi + 1 >= du.sp_i0) && (loaded <= du.sp_i1);
    if (chunked) {
      // four columns per thread: one 16-byte load of r (+ the coarse x it needs), four LDS stores
      if (grp_on)
        for (int g = loaded + 1 + rg; g <= hi; g += ngrp) {
          const FillIn in = fill_load(U, HASC, g, c4);
          fill_apply(U, HASC, g, c4, in, sp_fill);
        }
    } else
    {
      int g = loaded + 1 + r0, j = c0;
      while (g <= hi) {
        float rv = 0.f, dv = 0.f;
        if (g <= n_ && j >= 1 && j <= m_) {
          const int kk = g * m + j;
          rv = F.r[kk];
          float lw, le, ls, ln, iv;
          if (sp_fill) {
            lw = F.lx[kk];
            le = F.lx[kk + m];
            ls = F.ly[kk];
            ln = F.ly[kk + 1];
            iv = F.inv[kk];
          } else {
            coef_analytic(D, du, g, j, n_, m_, lw, le, ls, ln, iv);
          }
          if (HASC) {  // r -= A * prolongate(coarse.x)   (PoissonMatrix.pde:53-64 order)
            const float dc = prolong_at(Cc.x, mc, g, j, n_, m_);
            const float dw = prolong_at(Cc.x, mc, g - 1, j, n_, m_), de = prolong_at(Cc.x, mc, g + 1, j, n_, m_);
            const float ds = prolong_at(Cc.x, mc, g, j - 1, n_, m_), dn = prolong_at(Cc.x, mc, g, j + 1, n_, m_);
            const float diag = -(lw + le + ls + ln);
            rv = rv - (dc * diag + dw * lw + de * le + ds * ls + dn * ln);
          }
          dv = rv * iv;  // MG.pde:80
        }
        const int sl = (g % NR) * PW + j;
        rr[sl] = rv;
        dr[sl] = dv;
        g += rstep;
        j += cstep;
        if (j >= m) {
          j -= m;
          ++g;
        }
      }
    }
    loaded = hi;
    lds_barrier();
    if (mk) pf.mark(PF_UP0);
#if AFC_RB
    {
      const bool special = (I0 + H >= du.sp_i0) && (I0 - kStageShift - 1 <= du.sp_i1);
      for (int h = 0; h <= kStageShift; ++h) {
        gs_rb_stage(F, rr, dr, PW, I0, h, special, du);
        lds_barrier();
      }
    }
#else
    if (tid < 64) {
#ifdef AFC_FORCE_SPECIAL  // development: every strip marches as a strip with special rows (same bits; prices that path)
      const bool special = true;
#else
      const bool special = (I0 + H >= du.sp_i0) && (I0 - 3 <= du.sp_i1);
#endif
      AFC_PRIO_RAISE();  // the serial march is the critical path of its workgroup
      if (special)
        gs_march3<false, true, H>(F, rr, dr, zrow, dummy, I0, &du);
      else if (du.zb)
        gs_march3<true, true, H>(F, rr, dr, zrow, dummy, I0, &du);
      else
        gs_march3<true, false, H>(F, rr, dr, zrow, dummy, I0, &du);
      AFC_PRIO_DROP();
    }
    lds_barrier();
#endif
    if (mk) pf.mark(PF_GS0);
    // ---- exit: rows whose d (and d of both row neighbours) is final: d.setBC(); increment() ----
    const int e0 = (I0 - kStageShift - 1 < 1) ? 1 : I0 - kStageShift - 1,
              e1 = (I0 + H - kStageShift - 1 >= n_) ? n_ : I0 + H - kStageShift - 2;
    const bool sp_exit = (e1 + 1 >= du.sp_i0) && (e0 - 1 <= du.sp_i1);
    if (chunked) {
      if (grp_on)
        for (int g = e0 + rg; g <= e1; g += ngrp) {
          const ExitIn in = exit_load(U, HASC, g, c4);
          exit_apply(U, HASC, need_r, g, c4, in, sp_exit, rr_part, xs_part, linf);
        }
    } else
    {
      int g = e0 + r0, j = c0;
      while (g <= e1) {
        if (j >= 1 && j <= m_) {
          const int kk = g * m + j;
          const int gw = (g - 1 < 1) ? 1 : g - 1, ge = (g + 1 > n_) ? n_ : g + 1;
          const int js = (j - 1 < 1) ? 1 : j - 1, jn = (j + 1 > m_) ? m_ : j + 1;
          const int sl = (g % NR) * PW;
          const float d = dr[sl + j];
          if (need_r) {
            const float dW = dr[(gw % NR) * PW + j], dE = dr[(ge % NR) * PW + j];
            const float dS = dr[sl + js], dN = dr[sl + jn];
            float lw, le, ls, ln, iv;
            if (sp_exit) {
              lw = F.lx[kk];
              le = F.lx[kk + m];
              ls = F.ly[kk];
              ln = F.ly[kk + 1];
            } else {
              coef_analytic(D, du, g, j, n_, m_, lw, le, ls, ln, iv);
            }
            const float diag = -(lw + le + ls + ln);
            const float rn = rr[sl + j] - (d * diag + dW * lw + dE * le + dS * ls + dN * ln);  // MG.pde:96
            F.r[kk] = rn;
            const float pr = rn * rn;
            rr_part += (double)pr;
            linf = jmax(linf, jabs(rn));
          }
          // x += d on the cell and on the ghosts that mirror it (MG.pde:95 after d.setBC())
          float xo = F.x[kk];
          if (HASC) xo = xo + prolong_at(Cc.x, mc, g, j, n_, m_);
          const float xnew = xo + d;
          F.x[kk] = xnew;
          if (need_r) xs_part += (double)xnew;
          const bool gi0 = (g == 1), gi1 = (g == n_), gj0 = (j == 1), gj1 = (j == m_);
          if (gi0 | gi1 | gj0 | gj1) {
            const float dp = HASC ? prolong_at(Cc.x, mc, g, j, n_, m_) : 0.f;
#define GHOST_UPD(GI, GJ)                        \
  {                                              \
    float xg = F.x[(GI) * m + (GJ)];             \
    if (HASC) xg = xg + dp;                      \
    F.x[(GI) * m + (GJ)] = xg + d;               \
  }
            if (gi0) GHOST_UPD(0, j)
            if (gi1) GHOST_UPD(n - 1, j)
            if (gj0) GHOST_UPD(g, 0)
            if (gj1) GHOST_UPD(g, m - 1)
            if (gi0 && gj0) GHOST_UPD(0, 0)
            if (gi0 && gj1) GHOST_UPD(0, m - 1)
            if (gi1 && gj0) GHOST_UPD(n - 1, 0)
            if (gi1 && gj1) GHOST_UPD(n - 1, m - 1)
#undef GHOST_UPD
          }
        }
        g += rstep;
        j += cstep;
        if (j >= m) {
          j -= m;
          ++g;
        }
      }
    }
    lds_barrier();  // the next strip's fill re-uses ring slots this exit has just read
    if (mk) pf.mark(PF_INC0);
  }
  __syncthreads();  // phase boundary: the x (and r) this level stored must be visible to what follows
  if (need_r && rmode != 2) {
    block_reduce3(rr_part, xs_part, linf, lds);  // (the rings are dead: the phase-boundary barrier above); -0.22 % on the same memory
    *rr_out = uni((float)rr_part);
    *linf_out = uni(linf);
    if (xsum_out) *xsum_out = uni((float)xs_part);
  }
}

// mg_up_smooth as a SOFTWARE PIPELINE over the strips (round 5): wave 0 marches strip after strip without a workgroup
// barrier in between; waves 1-3 follow it column group by column group (16 columns x the 16 rows of a strip per wave
// and task) -- the rows whose last sweep the march has finished leave (exit of strip s: d.setBC(), increment, MG.pde:90-97)
// as soon as the march has passed their columns, and the next strip's rows are filled into the ring slots they free
// (rows I0+17 .. I0+32 take the slots of rows I0-7 .. I0+8: the ring keeps its 24 rows).  Hand-over through progress words
// in LDS (PipeSync): the march publishes its step count every 8 steps; a helper task waits for the step at which sweep 3
// of the strip's last row has passed its columns (16 q + 44), loads having been issued before the wait; the march waits
// for the fill of the next strip's first five column groups at its head and for the rest at step 64.  Same operands in
// the same order as mg_up_smooth: same bits (the per-thread partial sums of r.r / sum(x) visit the cells in another
// order -- they are doubles of float terms, re-rounded to float).
//   task q (0 .. Q) of strip s:  exit of column group q  |  fill of column group q-1 of strip s+1
// The fill of group q-1 overwrites ring slots that the exits of groups q-2, q-1 (another wave's tasks, earlier ones: xf
// flags) and q (this task) read -- an exit reads one column to either side of its chunk.
// For levels of 16-column groups and 16-row strips (m_ % 16 == 0, n_ % 16 == 0); the caller keeps mg_up_smooth otherwise.
// The protocol on its own -- who may touch which ring cell when -- is checked by a discrete-event model under random and
// adversarial schedules, with weakened variants as counter-examples: tests/test_pipe_protocol_model.py.
__device__ __host__ __forceinline__ bool up_pipe_shape_ok(int n, int m) {
  return ((m - 2) & 15) == 0 && ((n - 2) & 15) == 0 && (m - 2) >= 32 && (m - 2) <= 256 && (n - 2) >= 32;
}
constexpr int kPipeWords = 64 + 16 + 16;  // progress (one word per lane of the marching wave), xf, ff
__device__ AFC_UP_ATTR void mg_up_pipe(const bool HASC, const Lv &F, const Lv &Cc, bool need_r, float *lds, double *sm,
                                       float *rr_out, float *linf_out, ProfT &pf, float *xsum_out = nullptr,
                                       const int rmode = 0) {
  const int n = F.n, m = F.m, n_ = n - 2, m_ = m - 2;
  const int PW = ring_pitch(m);
  constexpr int H = 16, NR = kRing;
  float *rr = lds + kPad + 3 + (((16u - (lds_addr(lds) & 15u)) & 15u) >> 2), *dr = rr + NR * PW + kPad,
        *zrow = dr + NR * PW + kPad, *dummy = zrow + PW + kPad;
  lds_i *flags = (lds_i *)(lds_f *)(dummy + PW + kPad);  // [64 progress words][16 xf][16 ff]
  lds_i *xf = flags + 64, *ff = flags + 80;
  const int tid = threadIdx.x;
  const LevelDesc &D = *F.desc;
  const int mc = Cc.m;
  const int CPR = m_ >> 2, Q = m_ >> 4;
  double rr_part = 0, xs_part = 0;
  float linf = 0.f;
  const DescU du = desc_u(D);
  UpGeom U{&F, &Cc, &D, n, m, n_, m_, mc, PW, CPR, rr, dr, du};
  U.NR = NR;
  U.al16 = true;
  U.store_r = (rmode != 1);
  U.update_x = (rmode != 2);
  __syncthreads();
  for (int q = tid; q < PW + 2 * kPad; q += kThreads) zrow[q - kPad] = 0.f;
  if (tid < kPipeWords) flags[tid] = 0;
  {  // prologue: rows 1 .. H+1 by all threads (mg_up_smooth's fill of the first strip)
    const int cpr_sh = __builtin_ctz((unsigned)CPR);
    const bool cpr_p2 = (CPR & (CPR - 1)) == 0;
    const int ngrp = cpr_p2 ? (kThreads >> cpr_sh) : kThreads / CPR;
    const int rg = cpr_p2 ? (tid >> cpr_sh) : tid / CPR, c4 = tid - rg * CPR;
    const bool sp_fill = (H + 2 >= du.sp_i0) && (0 <= du.sp_i1);
    if (rg < ngrp)
      for (int g = 1 + rg; g <= H + 1; g += ngrp) {
        const FillIn in = fill_load(U, HASC, g, c4);
        fill_apply(U, HASC, g, c4, in, sp_fill);
      }
  }
  lds_barrier();
  if (tid < 64) {
    PipeSync ps;
    ps.mp = lds_addr((const float *)(lds_f *)(flags + tid));
    ps.ff = ff;
    ps.Q = Q;
    AFC_PRIO_RAISE();  // the serial march is the critical path of its workgroup
    int s = 0;
    for (int I0 = 1; I0 - 3 <= n_; I0 += H, ++s) {
      const bool special = (I0 + H >= du.sp_i0) && (I0 - 3 <= du.sp_i1);
      ps.s = s;
      ps.sbase = s << 10;
      // rows of this strip filled: all column groups for a strip that takes coefficient loads (no wait inside its
      // march), the groups the first 64 steps can touch otherwise
      pipe_wait_fill(ff, 0, (special || Q < kPipeHeadGroups) ? Q : kPipeHeadGroups, s);
      if (special)
        gs_march3<false, true, H, true>(F, rr, dr, zrow, dummy, I0, &du, &ps);
      else if (du.zb)
        gs_march3<true, true, H, true>(F, rr, dr, zrow, dummy, I0, &du, &ps);
      else
        gs_march3<true, false, H, true>(F, rr, dr, zrow, dummy, I0, &du, &ps);
      asm volatile("" ::: "memory");
      *(volatile lds_i *)(flags + tid) = (s + 1) << 10;  // strip s marched
    }
    AFC_PRIO_DROP();
  } else {
    // (lane -> (row, chunk) with the chunk running fastest: four lanes read 64 contiguous bytes of a row.  With the ROW
    // running fastest the 128-bit LDS accesses would be free of bank conflicts -- rows are 12 (mod 32) banks apart -- and the
    // launch is 0.9 % slower: the global accesses of a lane quad then fall into four different rows)
    const int hw = (tid >> 6) - 1, l = tid & 63, rI = l >> 2, cq = l & 3;
    int s = 0;
    for (int I0 = 1; I0 - 3 <= n_; I0 += H, ++s) {
      const int e0 = (I0 - 4 < 1) ? 1 : I0 - 4, e1 = (I0 + H - 4 >= n_) ? n_ : I0 + H - 5;
      const bool last = (I0 + H - 3 > n_);  // no further strip: nothing to fill
      const bool sp_exit = (e1 + 1 >= du.sp_i0) && (e0 - 1 <= du.sp_i1);
      const bool sp_fill = (I0 + 2 * H + 1 >= du.sp_i0) && (I0 + H <= du.sp_i1);
      const int ge = I0 - 4 + rI, gf = I0 + H + 1 + rI;
      for (int q = hw; q <= Q; q += 3) {
        const bool ex_on = (q < Q) && (ge >= e0) && (ge <= e1);
        const bool fi_on = (q >= 1) && !last && (gf <= n_);  // (rows past the grid are never read from the rings)
        const int cx = 4 * q + cq, cf = cx - 4;
        ExitIn ein;
        FillIn fin;
        if (ex_on) ein = exit_load(U, HASC, ge, cx);
        if (fi_on) fin = fill_load(U, HASC, gf, cf);
        if (q < Q) {
          pipe_wait_word(flags, (s << 10) + 16 * q + 44);  // sweep 3 of the strip's last row has passed column 16 q + 17
          if (ex_on) exit_apply(U, HASC, need_r, ge, cx, ein, sp_exit, rr_part, xs_part, linf);
          pipe_post_word(xf + q, s + 1);
        }
        if (q >= 1 && !last) {
          pipe_wait_word(xf + q - 1, s + 1);
          if (q >= 2) pipe_wait_word(xf + q - 2, s + 1);
          if (fi_on) fill_apply(U, HASC, gf, cf, fin, sp_fill);
          pipe_post_word(ff + q - 1, s + 1);
        }
      }
    }
  }
  __syncthreads();  // phase boundary: the x (and r) this level stored must be visible to what follows
  if (need_r || (AFC_DIAG && ((F.desc->dbg_copy >> 20) == n))) pf.mark(PF_GS0);
  if (need_r && rmode != 2) {
    block_reduce3(rr_part, xs_part, linf, lds);
    *rr_out = uni((float)rr_part);
    *linf_out = uni(linf);
    if (xsum_out) *xsum_out = uni((float)xs_part);
  }
}


#include "bdim_v2_tmarch.h"

// (the all-waves hyperplane smoother lives in bdim_v2_multiwave.h, diagnostic instantiation only; its LDS footprint is
// sized by the host in every instantiation)
constexpr int kMwS = 8;
constexpr int kMwGuardLo = 96, kMwGuardMid = 160;  // lanes outside their row read (and ignore) whatever lies there

__device__ __host__ __forceinline__ int mw_lds_floats(int m, int H) {
  return kMwGuardLo + 2 * ((H + 6) * m + kMwGuardMid) + 64 + 8;
}

__device__ __forceinline__ float lane_prev(float v) {  // the value lane-1 holds (whole wave); lane 0 gets 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_next(float v) {  // the value lane+1 holds (whole wave); lane 63 gets 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}


#include "bdim_v2_rbw.h"

#if AFC_DIAG
#include "bdim_v2_multiwave.h"
#endif

// ---------------------------------------------------------------------------------------------
// Fused head of the pressure solve on the finest level (MG.pde:51 + first smooth(0) + restrict):
//   r0 = div(u) - A (p+s)      s = pending mean shift of the previous projection (see project_v2)
//   d  = r0*inv; d.setBC();    pout = (p+s) + d   (all cells incl. ghosts);   r = r0 - A d
//   r1 = restrict(r)           (interior; coarse ghosts are never read)
// Streamed in blocks of 14 rows: r0 of 16 rows goes through an LDS tile, every thread then owns
// a 2x4 patch (one restriction pair of rows, two coarse cells).  u, v, p are read once, p and r
// written once.  pin and pout are different buffers (rows a-2.. of pin are still needed when rows
// ..a-1 of pout are written).
// ---------------------------------------------------------------------------------------------
// RESID = true : finest level, first iteration: r0 = div(u) - A (pin+s), x_out = (pin+s) + d
// RESID = false: any level with an existing residual: r0 = F.r, x_out = F.x + d  (F.x is 0 on
//                coarse levels, MG.pde:56; on the finest level it is the current iterate)
// LDS: two tiles of TR = R+2 rows (global rows a-1 .. a+R), pitch PT = m+2 floats, a cell of column j at index j+3 so
// that a thread's four columns 1+4c..4+4c are one aligned 16-byte access: `tr` holds r0, `td` holds d = r0*inv WITH the
// setBC mirror written into its ghost columns and rows, so that the patches read their 4x6 window of d without any
// clamping, computed once per cell.  Threads keep their 4-column chunk c4 and walk down rows / row pairs: no division
// inside the loops.
template <bool RESID, bool XZ>
__device__ inline void mg_down_fused(const StepParams &P, const Lv &L0, const Lv &L1, const float *__restrict__ ux,
                                     const float *__restrict__ uy, const float *__restrict__ pin, float shift,
                                     float *lds) {
  const int n = L0.n, m = L0.m, n_ = n - 2, m_ = m - 2, CPR = m_ >> 2;
  const LevelDesc &D = *L0.desc;
  float *pout = L0.x, *r = L0.r;
  if (!RESID) {
    pin = L0.x;
    shift = 0.f;
  }
  shift = uni(shift);
  const int tid = threadIdx.x;
  const int m1 = L1.m;
  const DescU du = desc_u(D);
  const bool unif = (du.zb == 0);  // finest level: all coefficients uniform outside the special rows
  // rows per block: 14 on the finest grids; on narrow (coarse) levels enough rows to give every
  // thread a patch (7*k*CPR patches per block), bounded by the 2080-float tile
  const int PT = m + 2;
  int kblk = 1;
  if (!RESID && !(AFC_DBG_OF(P) & 1)) {
    kblk = 32 / (CPR > 0 ? CPR : 1);
    if (kblk < 1) kblk = 1;
    while (kblk > 1 && (14 * kblk + 2) * PT > 2112) --kblk;
  }
  const int R = 14 * kblk, TR = R + 2, NPAIR = 7 * kblk;
  // 16-byte aligned tiles; index of (tile row t, column j) = t*PT + j + 3
  const unsigned lbase = (lds_addr(lds) + 15u) & ~15u;
  const unsigned a_tr = lbase, a_td = lbase + 4u * (unsigned)(TR * PT + 4);
#define T_R4(t, c) (*(lds_f4 *)(uintptr_t)(a_tr + 4u * (unsigned)((t) * PT + 4 * (c) + 4)))
#define T_D4(t, c) (*(lds_f4 *)(uintptr_t)(a_td + 4u * (unsigned)((t) * PT + 4 * (c) + 4)))
#define T_D1(t, j) (*(lds_f *)(uintptr_t)(a_td + 4u * (unsigned)((t) * PT + (j) + 3)))
  const int ngrp = kThreads / CPR, rg = tid / CPR, c4 = tid - rg * CPR;  // the only divisions of the call
  const bool on = rg < ngrp;
  const int j = 1 + 4 * c4;
  for (int a = 1; a <= n_; a += R) {
    const bool sp = (a + R + 1 >= du.sp_i0) && (a - 2 <= du.sp_i1);
    lds_barrier();
    // ---- step 1: r0 and d = r0*inv for rows a-1 .. a+R ----
    if (on)
      for (int t = rg; t < TR; t += ngrp) {
        const int g = a - 1 + t;
        if (g < 1 || g > n_) continue;
        if (!RESID && t == 0) continue;  // row a-1 was overwritten by the previous block: kept in the tiles
        const int k = g * m + j;
        F4 r0;
        if (RESID) {
          const F4 uxc = load4u(ux + k), uxe = load4u(ux + k + m), uyc = load4u(uy + k);
          const float uyn = uy[k + 4];
          const F4 pc = load4u(pin + k), pw = load4u(pin + k - m), pe = load4u(pin + k + m);
          const float psl = pin[k - 1], pnr = pin[k + 4];
          F4 lw, le, ls;
          float ln4;
          if (sp) {
            lw = load4u(L0.lx + k);
            le = load4u(L0.lx + k + m);
            ls = load4u(L0.ly + k);
            ln4 = L0.ly[k + 4];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) lw.v[e] = le.v[e] = ls.v[e] = du.cu;
            ln4 = du.cu;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float uyN = (e < 3) ? uyc.v[(e + 1) & 3] : uyn;
            const float div = uxe.v[e] - uxc.v[e] + uyN - uyc.v[e];  // VectorField.pde:61-62
            const float lwv = lw.v[e], lev = le.v[e], lsv = ls.v[e], lnv = (e < 3) ? ls.v[(e + 1) & 3] : ln4;
            const float diag = -(lwv + lev + lsv + lnv);
            const float pcv = pc.v[e] + shift, pwv = pw.v[e] + shift, pev = pe.v[e] + shift;
            const float psv = ((e > 0) ? pc.v[(e + 3) & 3] : psl) + shift,
                        pnv = ((e < 3) ? pc.v[(e + 1) & 3] : pnr) + shift;
            const float ap = pcv * diag + pwv * lwv + pev * lev + psv * lsv + pnv * lnv;  // PoissonMatrix.pde:57-61
            r0.v[e] = div - ap;
          }
        } else {
          r0 = load4u(r + k);
        }
        F4 iv;
        if (sp) {
          iv = load4u(L0.inv + k);
        } else if (unif) {
#pragma unroll
          for (int e = 0; e < 4; ++e) iv.v[e] = du.iv0;
        } else {
          const int rc = 3 * ((g == 1) ? 1 : ((g == n_) ? 2 : 0));
          const float i0 = D.invtab[rc];
          iv.v[0] = (c4 == 0) ? D.invtab[rc + 1] : i0;
          iv.v[1] = i0;
          iv.v[2] = i0;
          iv.v[3] = (c4 == CPR - 1) ? D.invtab[rc + 2] : i0;
        }
        f4a r4, d4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          r4[e] = r0.v[e];
          d4[e] = r0.v[e] * iv.v[e];  // MG.pde:80
        }
        T_R4(t, c4) = r4;
        T_D4(t, c4) = d4;
        // d.setBC(): ghosts mirror the adjacent interior cell (Field.pde:211-224)
        if (c4 == 0) T_D1(t, 0) = d4[0];
        if (c4 == CPR - 1) T_D1(t, m_ + 1) = d4[3];
        if (g == 1 || g == n_) {
          const int tg = (g == 1) ? t - 1 : t + 1;
          if (tg >= 0 && tg < TR) {
            T_D4(tg, c4) = d4;
            if (c4 == 0) T_D1(tg, 0) = d4[0];
            if (c4 == CPR - 1) T_D1(tg, m_ + 1) = d4[3];
          }
        }
      }
    lds_barrier();
    // ---- step 2: one 2x4 patch per thread ----
    if (on)
      for (int pr = rg; pr < NPAIR; pr += ngrp) {
        const int g = a + 2 * pr;
        if (g > n_) continue;
        const int tq = g - (a - 1);  // tile row of g
        // d on rows g-1..g+2, cols j-1..j+4
        float dd[4][6];
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
          const f4a v = T_D4(tq - 1 + a4, c4);
          dd[a4][0] = T_D1(tq - 1 + a4, j - 1);
          dd[a4][1] = v[0], dd[a4][2] = v[1], dd[a4][3] = v[2], dd[a4][4] = v[3];
          dd[a4][5] = T_D1(tq - 1 + a4, j + 4);
        }
        float rn[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int gg = g + h, k = gg * m + j;
          F4 lw, le, ls;
          float ln4;
          if (sp) {
            lw = load4u(L0.lx + k);
            le = load4u(L0.lx + k + m);
            ls = load4u(L0.ly + k);
            ln4 = L0.ly[k + 4];
          } else if (unif) {
#pragma unroll
            for (int e = 0; e < 4; ++e) lw.v[e] = le.v[e] = ls.v[e] = du.cu;
            ln4 = du.cu;
          } else {
            float a0, a1, a2, a3, a4v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              coef_analytic(D, du, gg, j + e, n_, m_, a0, a1, a2, a3, a4v);
              lw.v[e] = a0;
              le.v[e] = a1;
              ls.v[e] = a2;
              if (e == 3) ln4 = a3;
            }
          }
          F4 pc;  // x before the increment: the iterate on the finest level, exactly zero on coarse ones (MG.pde:56)
          if (!XZ) {
            pc = load4u(pin + k);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) pc.v[e] = 0.f;
          }
          const f4a r0row = T_R4(tq + h, c4);
          F4 xo, ro;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lwv = lw.v[e], lev = le.v[e], lsv = ls.v[e], lnv = (e < 3) ? ls.v[(e + 1) & 3] : ln4;
            const float diag = -(lwv + lev + lsv + lnv);
            const float d = dd[1 + h][1 + e];
            const float ad = d * diag + dd[h][1 + e] * lwv + dd[2 + h][1 + e] * lev + dd[1 + h][e] * lsv +
                             dd[1 + h][2 + e] * lnv;  // PoissonMatrix.pde:57-61
            rn[h][e] = r0row[e] - ad;                  // MG.pde:96
            ro.v[e] = rn[h][e];
            xo.v[e] = (pc.v[e] + shift) + d;           // MG.pde:95
          }
          store4u(pout + k, xo);
          store4u(r + k, ro);
          // ghosts mirror the adjacent interior d (d.setBC(), Field.pde:211-224)
          if (c4 == 0) pout[k - 1] = ((XZ ? 0.f : pin[k - 1]) + shift) + dd[1 + h][1];
          if (c4 == CPR - 1) pout[k + 4] = ((XZ ? 0.f : pin[k + 4]) + shift) + dd[1 + h][4];
          const bool top = (gg == 1), bot = (gg == n_);
          if (top || bot) {
            const int kg = (top ? 0 : (n - 1)) * m + j;
            F4 pg;
            if (!XZ) {
              pg = load4u(pin + kg);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) pg.v[e] = 0.f;
            }
            F4 xg;
#pragma unroll
            for (int e = 0; e < 4; ++e) xg.v[e] = (pg.v[e] + shift) + dd[1 + h][1 + e];
            store4u(pout + kg, xg);
            if (c4 == 0) pout[kg - 1] = ((XZ ? 0.f : pin[kg - 1]) + shift) + dd[1 + h][1];
            if (c4 == CPR - 1) pout[kg + 4] = ((XZ ? 0.f : pin[kg + 4]) + shift) + dd[1 + h][4];
          }
        }
        // restrict (MG.pde:124-137): sum of the 2x2 children, reference order
        const int ci = (g - 1) / 2 + 1, cj = (j - 1) / 2 + 1;
        L1.r[ci * m1 + cj] = rn[0][0] + rn[0][1] + rn[1][0] + rn[1][1];
        L1.r[ci * m1 + cj + 1] = rn[0][2] + rn[0][3] + rn[1][2] + rn[1][3];
      }
    if (!RESID) {  // in-place r: the next block's row a'-1 (= tile row R) must keep its ORIGINAL r (and its d)
      lds_barrier();
      for (int q = tid; q <= PT; q += kThreads) {  // indices 3 .. PT of the row are in use
        lds_f *tr1 = (lds_f *)(uintptr_t)a_tr, *td1 = (lds_f *)(uintptr_t)a_td;
        tr1[q] = tr1[R * PT + q];
        td1[q] = td1[R * PT + q];
      }
    }
  }
#undef T_R4
#undef T_D4
#undef T_D1
  __syncthreads();
}

constexpr int kHeadNS = 8;                                       // rows in a wave's d ring
constexpr int kHeadLdsFloats = 4 * (kHeadNS * 132 + 8) + 8;      // head_v3: four rings (m = 130: pitch 132)

// The fused head (mg_down_fused<true,false>: r0 = div u - A(p+s); d = r0*inv; d.setBC; pout = (p+s) + d; r = r0 - A d;
// r1 = restrict(r)) on grids of 128 interior columns as FOUR INDEPENDENT ROW PIPELINES without workgroup barriers:
// wave w owns a contiguous quarter of the rows (an even number, starting on an odd row, so restriction pairs stay
// inside a wave); lanes 0-31 / 32-63 hold the 32 four-column chunks of an odd row / the even row below it.  Per
// iteration a wave (1) forms r0 and d of rows g+2, g+3 (one row pair ahead: d of the rows above and below a cell must
// exist before its residual can), d into a private 8-row LDS ring with the setBC mirror in its ghost columns and
// rows, r0 and the centre p staying in registers for one iteration; (2) finishes rows g, g+1: p, r and the
// restricted r1 (the even row's four residuals come down to the odd row's lanes with v_permlane32_swap).  The rows
// just outside a quarter (one above, one below) are evaluated by both neighbours.  Same expressions, same order, same
// bits as mg_down_fused.
__device__ AFC_HEAD_ATTR void head_v3(const StepParams &P, const Lv &L0, const Lv &L1, const float *__restrict__ ux,
                               const float *__restrict__ uy, const float *__restrict__ pin, float shift, float *lds) {
  constexpr int m = 130, m_ = 128, CPR = 32, PT = 132, NS = kHeadNS;
  const int n = uni(L0.n), n_ = n - 2;
  const LevelDesc &D = *L0.desc;
  float *pout = L0.x, *r = L0.r, *r1 = L1.r;
  const float *lxp = L0.lx, *lyp = L0.ly, *invp = L0.inv;
  const int m1 = uni(L1.m);
  shift = uni(shift);
  const float cu = uni(D.cu), iv0 = uni(D.invtab[0]);
  const int sp_i0 = uni(D.sp_i0), sp_i1 = uni(D.sp_i1);
  const int wave = uni((int)(threadIdx.x >> 6));
  const int RPW = n_ >> 2;
  const int i_first = 1 + wave * RPW, i_last = i_first + RPW - 1;
  const unsigned ring = uni((int)(((lds_addr(lds) + 15u) & ~15u) + (unsigned)wave * ((NS * PT + 8) * 4u)));
  // ring element (row R, column jj) at float index (R & 7)*PT + jj + 3: a lane's four columns are one aligned 16 bytes
#define HD_D4(R, c) (*(lds_f4 *)(uintptr_t)(ring + 4u * (unsigned)(((R) & (NS - 1)) * PT + 4 * (c) + 4)))
#define HD_D1(R, jj) (*(lds_f *)(uintptr_t)(ring + 4u * (unsigned)(((R) & (NS - 1)) * PT + (jj) + 3)))
  F4 c_r0, c_pc;        // carried from step 1 to the next iteration's step 2: r0 and p of the lane's chunk
  float c_psl, c_pnr;   // ... and the p values left / right of the chunk (ghost columns of the first / last chunk)
#pragma unroll
  for (int e = 0; e < 4; ++e) c_r0.v[e] = c_pc.v[e] = 0.f;
  c_psl = c_pnr = 0.f;
  // ---- step 1 on the row pair (R0, R0+1): lanes of rows outside [lo_ok, hi_ok] idle ----
#define HD_STEP1(R0, lo_ok, hi_ok, n_r0, n_pc, n_psl, n_pnr)                                                       \
  {                                                                                                                \
    const unsigned ln_ = lane_fresh();                                                                             \
    const int h_ = (int)(ln_ >> 5), c4_ = (int)(ln_ & 31u);                                                        \
    const int R_ = (R0) + h_;                                                                                      \
    if (R_ >= (lo_ok) && R_ <= (hi_ok)) {                                                                          \
      const unsigned kb_ = 4u * (unsigned)(R_ * m + 1 + 4 * c4_);                                                  \
      const bool sp_ = ((R0) + 1 >= sp_i0) && ((R0) <= sp_i1);                                                     \
      const F4 uxc = ld4b(ux, kb_), uxe = ld4b(ux, kb_ + 4u * m), uyc = ld4b(uy, kb_);                             \
      const float uyn = ld1b(uy, kb_ + 16u);                                                                       \
      const F4 pc = ld4b(pin, kb_), pw = ld4b(pin, kb_ - 4u * m), pe = ld4b(pin, kb_ + 4u * m);                    \
      const float psl = ld1b(pin, kb_ - 4u), pnr = ld1b(pin, kb_ + 16u);                                           \
      F4 lw, le, ls, iv;                                                                                           \
      float ln4;                                                                                                   \
      if (sp_) {                                                                                                   \
        lw = ld4b(lxp, kb_);                                                                                       \
        le = ld4b(lxp, kb_ + 4u * m);                                                                              \
        ls = ld4b(lyp, kb_);                                                                                       \
        ln4 = ld1b(lyp, kb_ + 16u);                                                                                \
        iv = ld4b(invp, kb_);                                                                                      \
      } else {                                                                                                     \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) lw.v[e] = le.v[e] = ls.v[e] = cu, iv.v[e] = iv0;             \
        ln4 = cu;                                                                                                  \
      }                                                                                                            \
      f4a d4;                                                                                                      \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                              \
        const float uyN = (e < 3) ? uyc.v[(e + 1) & 3] : uyn;                                                      \
        const float div = uxe.v[e] - uxc.v[e] + uyN - uyc.v[e]; /* VectorField.pde:61-62 */                        \
        const float lwv = lw.v[e], lev = le.v[e], lsv = ls.v[e], lnv = (e < 3) ? ls.v[(e + 1) & 3] : ln4;          \
        const float diag = -(lwv + lev + lsv + lnv);                                                               \
        const float pcv = pc.v[e] + shift, pwv = pw.v[e] + shift, pev = pe.v[e] + shift;                           \
        const float psv = ((e > 0) ? pc.v[(e + 3) & 3] : psl) + shift,                                             \
                    pnv = ((e < 3) ? pc.v[(e + 1) & 3] : pnr) + shift;                                             \
        const float ap = pcv * diag + pwv * lwv + pev * lev + psv * lsv + pnv * lnv; /* PoissonMatrix.pde:57-61 */ \
        n_r0.v[e] = div - ap;                                                                                      \
        d4[e] = n_r0.v[e] * iv.v[e]; /* MG.pde:80 */                                                               \
      }                                                                                                            \
      n_pc = pc;                                                                                                   \
      n_psl = psl;                                                                                                 \
      n_pnr = pnr;                                                                                                 \
      HD_D4(R_, c4_) = d4;                                                                                         \
      if (c4_ == 0) HD_D1(R_, 0) = d4[0]; /* d.setBC(): ghosts mirror the adjacent interior cell */                \
      if (c4_ == CPR - 1) HD_D1(R_, m_ + 1) = d4[3];                                                               \
      if (R_ == 1 || R_ == n_) {                                                                                   \
        const int Rg_ = (R_ == 1) ? 0 : n_ + 1;                                                                    \
        HD_D4(Rg_, c4_) = d4;                                                                                      \
        if (c4_ == 0) HD_D1(Rg_, 0) = d4[0];                                                                       \
        if (c4_ == CPR - 1) HD_D1(Rg_, m_ + 1) = d4[3];                                                            \
      }                                                                                                            \
    }                                                                                                              \
  }
  const int d_hi = (i_last + 1 < n_) ? i_last + 1 : n_;  // last row whose d this wave forms itself
  {
    F4 t_r0, t_pc;
    float t_a, t_b;
    HD_STEP1(i_first - 2, (i_first - 1 > 1 ? i_first - 1 : 1), i_first - 1, t_r0, t_pc, t_a, t_b)  // the row above the quarter
    (void)t_r0, (void)t_pc, (void)t_a, (void)t_b;
  }
  HD_STEP1(i_first, i_first, i_first + 1, c_r0, c_pc, c_psl, c_pnr)
  asm volatile("" ::: "memory");
  for (int g = i_first; g <= i_last; g += 2) {
    F4 n_r0, n_pc;
    float n_psl = 0.f, n_pnr = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) n_r0.v[e] = n_pc.v[e] = 0.f;
    HD_STEP1(g + 2, g + 2, d_hi, n_r0, n_pc, n_psl, n_pnr)
    // lanes read below what OTHER lanes of the wave have just stored: program order must survive the optimiser (for one
    // thread the addresses are provably different); in hardware the LDS operations of a wave execute in order
    asm volatile("" ::: "memory");
    {  // ---- step 2: rows g (lanes 0-31), g+1 (lanes 32-63) ----
      const unsigned ln = lane_fresh();
      const int h = (int)(ln >> 5), c4 = (int)(ln & 31u);
      const int G = g + h, j = 1 + 4 * c4;
      const unsigned kb = 4u * (unsigned)(G * m + j);
      const bool sp = (g + 1 >= sp_i0) && (g <= sp_i1);
      F4 lw, le, ls;
      float ln4;
      if (sp) {
        lw = ld4b(lxp, kb);
        le = ld4b(lxp, kb + 4u * m);
        ls = ld4b(lyp, kb);
        ln4 = ld1b(lyp, kb + 16u);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) lw.v[e] = le.v[e] = ls.v[e] = cu;
        ln4 = cu;
      }
      const f4a dc = HD_D4(G, c4), dw = HD_D4(G - 1, c4), de = HD_D4(G + 1, c4);
      const float dl = HD_D1(G, j - 1), dr = HD_D1(G, j + 4);
      const float d6[6] = {dl, dc[0], dc[1], dc[2], dc[3], dr};
      F4 xo, ro;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lwv = lw.v[e], lev = le.v[e], lsv = ls.v[e], lnv = (e < 3) ? ls.v[(e + 1) & 3] : ln4;
        const float diag = -(lwv + lev + lsv + lnv);
        const float d = d6[1 + e];
        const float ad = d * diag + dw[e] * lwv + de[e] * lev + d6[e] * lsv + d6[2 + e] * lnv;  // PoissonMatrix.pde:57-61
        ro.v[e] = c_r0.v[e] - ad;                                                              // MG.pde:96
        xo.v[e] = (c_pc.v[e] + shift) + d;                                                     // MG.pde:95
      }
      st4b(pout, kb, xo);
      st4b(r, kb, ro);
      // ghosts mirror the adjacent interior d (d.setBC(), Field.pde:211-224)
      if (c4 == 0) st1b(pout, kb - 4u, (c_psl + shift) + d6[1]);
      if (c4 == CPR - 1) st1b(pout, kb + 16u, (c_pnr + shift) + d6[4]);
      if (G == 1 || G == n_) {
        const unsigned kg = 4u * (unsigned)(((G == 1) ? 0 : (n - 1)) * m + j);
        const F4 pg = ld4b(pin, kg);
        F4 xg;
#pragma unroll
        for (int e = 0; e < 4; ++e) xg.v[e] = (pg.v[e] + shift) + d6[1 + e];
        st4b(pout, kg, xg);
        if (c4 == 0) st1b(pout, kg - 4u, (ld1b(pin, kg - 4u) + shift) + d6[1]);
        if (c4 == CPR - 1) st1b(pout, kg + 16u, (ld1b(pin, kg + 16u) + shift) + d6[4]);
      }
      // restrict (MG.pde:124-137): sum of the 2x2 children in the reference order; the even row's residuals come down
      // from lanes 32-63
      float up[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // v_permlane32_swap a, b: lanes 32-63 of a <-> lanes 0-31 of b; afterwards b's lanes 0-31 hold what a's lanes
        // 32-63 held.  (Written out: with the builtin on two copies of one value and only the second result used, hipcc
        // 7.2 hands back the first -- the lane's own value; s_nop 1 = the two wait states between a VALU write of an
        // operand and the swap.)
        float a_ = ro.v[e], b_ = ro.v[e];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a_), "+v"(b_));
        up[e] = b_;
      }
      if (h == 0) {
        const int ci = (g - 1) / 2 + 1, cj = 2 * c4 + 1;
        float two[2];
        two[0] = ro.v[0] + ro.v[1] + up[0] + up[1];
        two[1] = ro.v[2] + ro.v[3] + up[2] + up[3];
        __builtin_memcpy((char *)r1 + 4u * (unsigned)(ci * m1 + cj), two, 8);
      }
    }
    asm volatile("" ::: "memory");
    c_r0 = n_r0;
    c_pc = n_pc;
    c_psl = n_psl;
    c_pnr = n_pnr;
  }
#undef HD_STEP1
#undef HD_D4
#undef HD_D1
  __syncthreads();
}

// The fused head on grids of 132 .. 256 interior columns (33 .. 64 chunks of four columns per row: 384x192, 512x256) as
// four barrier-free row pipelines: head_v3's scheme with ONE ROW PER PASS of the wave (lane = chunk) -- a row pair is two
// passes, so both rows of a restriction pair end up in the same lane and the restricted residual needs no lane
// exchange.  Per iteration a wave forms r0 and d of rows g+2, g+3 (d into its private 8-row ring, r0 and p carried in
// registers) and finishes rows g, g+1.  Same expressions, same order, same bits as mg_down_fused<true, false>.
constexpr int kHead4NS = 8;
__device__ __host__ __forceinline__ bool head4_shape_ok(int n, int m) {
  const int n_ = n - 2, m_ = m - 2;
  return (m_ & 3) == 0 && (m_ >> 2) > 32 && (m_ >> 2) <= 64 && (n_ & 7) == 0;
}
__device__ __host__ __forceinline__ int head4_lds_floats(int m) { return 4 * (kHead4NS * (m + 2) + 8) + 8; }
__device__ AFC_HEAD_ATTR void head_v4(const StepParams &P, const Lv &L0, const Lv &L1, const float *__restrict__ ux,
                                      const float *__restrict__ uy, const float *__restrict__ pin, float shift, float *lds) {
  constexpr int NS = kHead4NS;
  const int n = uni(L0.n), m = uni(L0.m), n_ = n - 2, m_ = m - 2, CPR = m_ >> 2, PT = m_ + 4;
  const unsigned mb = 4u * (unsigned)m;  // bytes per grid row
  const LevelDesc &D = *L0.desc;
  float *pout = L0.x, *r = L0.r, *r1 = L1.r;
  const float *lxp = L0.lx, *lyp = L0.ly, *invp = L0.inv;
  const int m1 = uni(L1.m);
  shift = uni(shift);
  const float cu = uni(D.cu), iv0 = uni(D.invtab[0]);
  const int sp_i0 = uni(D.sp_i0), sp_i1 = uni(D.sp_i1);
  const int wave = uni((int)(threadIdx.x >> 6));
  const int RPW = n_ >> 2;
  const int i_first = 1 + wave * RPW, i_last = i_first + RPW - 1;
  const unsigned ring = uni((int)(((lds_addr(lds) + 15u) & ~15u) + (unsigned)wave * ((unsigned)(NS * PT + 8) * 4u)));
#define H4_D4(R, c) (*(lds_f4 *)(uintptr_t)(ring + 4u * (unsigned)(((R) & (NS - 1)) * PT + 4 * (c) + 4)))
#define H4_D1(R, jj) (*(lds_f *)(uintptr_t)(ring + 4u * (unsigned)(((R) & (NS - 1)) * PT + (jj) + 3)))
  struct Row {  // carried from step 1 to the next iteration's step 2: r0 and p of the lane's chunk, p left / right of it
    F4 r0, pc;
    float psl, pnr;
  };
  const int c4 = (int)lane_fresh();
  const bool on = c4 < CPR;
  const int j = 1 + 4 * c4;
  const int d_hi = (i_last + 1 < n_) ? i_last + 1 : n_;  // last row whose d this wave forms itself
  // ---- step 1 on row R_ (rows outside [lo_ok, hi_ok]: nothing) ----
  auto step1 = [&](int R_, int lo_ok, int hi_ok) {
    Row o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.r0.v[e] = o.pc.v[e] = 0.f;
    o.psl = o.pnr = 0.f;
    if (on && R_ >= lo_ok && R_ <= hi_ok) {
      const unsigned kb_ = 4u * (unsigned)(R_ * m + j);
      const bool sp_ = (R_ >= sp_i0) && (R_ <= sp_i1);
      const F4 uxc = ld4b(ux, kb_), uxe = ld4b(ux, kb_ + mb), uyc = ld4b(uy, kb_);
      const float uyn = ld1b(uy, kb_ + 16u);
      const F4 pc = ld4b(pin, kb_), pw = ld4b(pin, kb_ - mb), pe = ld4b(pin, kb_ + mb);
      const float psl = ld1b(pin, kb_ - 4u), pnr = ld1b(pin, kb_ + 16u);
      F4 lw, le, ls, iv;
      float ln4;
      if (sp_) {
        lw = ld4b(lxp, kb_);
        le = ld4b(lxp, kb_ + mb);
        ls = ld4b(lyp, kb_);
        ln4 = ld1b(lyp, kb_ + 16u);
        iv = ld4b(invp, kb_);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) lw.v[e] = le.v[e] = ls.v[e] = cu, iv.v[e] = iv0;
        ln4 = cu;
      }
      f4a d4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float uyN = (e < 3) ? uyc.v[(e + 1) & 3] : uyn;
        const float div = uxe.v[e] - uxc.v[e] + uyN - uyc.v[e];  // VectorField.pde:61-62
        const float lwv = lw.v[e], lev = le.v[e], lsv = ls.v[e], lnv = (e < 3) ? ls.v[(e + 1) & 3] : ln4;
        const float diag = -(lwv + lev + lsv + lnv);
        const float pcv = pc.v[e] + shift, pwv = pw.v[e] + shift, pev = pe.v[e] + shift;
        const float psv = ((e > 0) ? pc.v[(e + 3) & 3] : psl) + shift, pnv = ((e < 3) ? pc.v[(e + 1) & 3] : pnr) + shift;
        const float ap = pcv * diag + pwv * lwv + pev * lev + psv * lsv + pnv * lnv;  // PoissonMatrix.pde:57-61
        o.r0.v[e] = div - ap;
        d4[e] = o.r0.v[e] * iv.v[e];  // MG.pde:80
      }
      o.pc = pc;
      o.psl = psl;
      o.pnr = pnr;
      H4_D4(R_, c4) = d4;
      if (c4 == 0) H4_D1(R_, 0) = d4[0];  // d.setBC(): ghosts mirror the adjacent interior cell
      if (c4 == CPR - 1) H4_D1(R_, m_ + 1) = d4[3];
      if (R_ == 1 || R_ == n_) {
        const int Rg_ = (R_ == 1) ? 0 : n_ + 1;
        H4_D4(Rg_, c4) = d4;
        if (c4 == 0) H4_D1(Rg_, 0) = d4[0];
        if (c4 == CPR - 1) H4_D1(Rg_, m_ + 1) = d4[3];
      }
    }
    return o;
  };
  // ---- step 2 on row G_: p, r of the row; returns the row's residuals for the restriction ----
  auto step2 = [&](int G_, const Row &c) {
    F4 ro;
#pragma unroll
    for (int e = 0; e < 4; ++e) ro.v[e] = 0.f;
    if (on) {
      const unsigned kb = 4u * (unsigned)(G_ * m + j);
      const bool sp = (G_ >= sp_i0) && (G_ <= sp_i1);
      F4 lw, le, ls;
      float ln4;
      if (sp) {
        lw = ld4b(lxp, kb);
        le = ld4b(lxp, kb + mb);
        ls = ld4b(lyp, kb);
        ln4 = ld1b(lyp, kb + 16u);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) lw.v[e] = le.v[e] = ls.v[e] = cu;
        ln4 = cu;
      }
      const f4a dc = H4_D4(G_, c4), dw = H4_D4(G_ - 1, c4), de = H4_D4(G_ + 1, c4);
      const float dl = H4_D1(G_, j - 1), dr = H4_D1(G_, j + 4);
      const float d6[6] = {dl, dc[0], dc[1], dc[2], dc[3], dr};
      F4 xo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lwv = lw.v[e], lev = le.v[e], lsv = ls.v[e], lnv = (e < 3) ? ls.v[(e + 1) & 3] : ln4;
        const float diag = -(lwv + lev + lsv + lnv);
        const float d = d6[1 + e];
        const float ad = d * diag + dw[e] * lwv + de[e] * lev + d6[e] * lsv + d6[2 + e] * lnv;  // PoissonMatrix.pde:57-61
        ro.v[e] = c.r0.v[e] - ad;                                                              // MG.pde:96
        xo.v[e] = (c.pc.v[e] + shift) + d;                                                     // MG.pde:95
      }
      st4b(pout, kb, xo);
      st4b(r, kb, ro);
      // ghosts mirror the adjacent interior d (d.setBC(), Field.pde:211-224)
      if (c4 == 0) st1b(pout, kb - 4u, (c.psl + shift) + d6[1]);
      if (c4 == CPR - 1) st1b(pout, kb + 16u, (c.pnr + shift) + d6[4]);
      if (G_ == 1 || G_ == n_) {
        const unsigned kg = 4u * (unsigned)(((G_ == 1) ? 0 : (n - 1)) * m + j);
        const F4 pg = ld4b(pin, kg);
        F4 xg;
#pragma unroll
        for (int e = 0; e < 4; ++e) xg.v[e] = (pg.v[e] + shift) + d6[1 + e];
        st4b(pout, kg, xg);
        if (c4 == 0) st1b(pout, kg - 4u, (ld1b(pin, kg - 4u) + shift) + d6[1]);
        if (c4 == CPR - 1) st1b(pout, kg + 16u, (ld1b(pin, kg + 16u) + shift) + d6[4]);
      }
    }
    return ro;
  };
  (void)step1(i_first - 1, (i_first - 1 > 1 ? i_first - 1 : 1), i_first - 1);  // the row above the run: its d only
  Row cA = step1(i_first, i_first, d_hi), cB = step1(i_first + 1, i_first, d_hi);
  asm volatile("" ::: "memory");
  for (int g = i_first; g <= i_last; g += 2) {
    const Row nA = step1(g + 2, g + 2, d_hi), nB = step1(g + 3, g + 2, d_hi);
    // lanes read below what OTHER lanes of the wave have just stored: program order must survive the optimiser; in
    // hardware the LDS operations of a wave execute in order
    asm volatile("" ::: "memory");
    const F4 ra = step2(g, cA), rb = step2(g + 1, cB);
    if (on) {  // restrict (MG.pde:124-137): sum of the 2x2 children in the reference order
      const int ci = (g - 1) / 2 + 1, cj = 2 * c4 + 1;
      float two[2];
      two[0] = ra.v[0] + ra.v[1] + rb.v[0] + rb.v[1];
      two[1] = ra.v[2] + ra.v[3] + rb.v[2] + rb.v[3];
      __builtin_memcpy((char *)r1 + 4u * (unsigned)(ci * m1 + cj), two, 8);
    }
    asm volatile("" ::: "memory");
    cA = nA;
    cB = nB;
  }
#undef H4_D4
#undef H4_D1
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// The descending leg of a COARSE level (mg_down_fused<false, true>: d = r*inv; d.setBC; x = 0 + d; r -= A d;
// r_c = restrict(r), MG.pde:69-70 with MG.pde:56,80,90-97,124-137) as FOUR INDEPENDENT ROW PIPELINES without a workgroup
// barrier -- head_v3's shape for any level whose rows are 6 .. 32 chunks of four columns.  As blocks of 14*k rows
// (mg_down_fused) the leg is twenty barrier-separated segments with a global round trip each on level 1 of the
// 256x128 grid: 24 us for one env alone, 119 us with 1024 resident (profiles/r3_level_profile_*.txt).
// Lanes: h = lane / CPR rows of a block of RPI = (64 / CPR) & ~1 rows (restriction pairs stay inside a block), c4 = lane
// % CPR the chunk.  Wave w owns a contiguous run of row pairs.  Per iteration a wave (0) fetches r of the block two
// ahead, (1) forms d = r*inv of the block one ahead into its private LDS ring (with the setBC mirror in the ghost
// columns and rows), r staying in registers for one iteration, (2) finishes its block: x, r - A d and the restricted
// residual (the even row's four residuals reach the odd row's lanes through ds_bpermute).  The rows just outside a
// wave's run (one above, one below) are evaluated by both neighbours, from the level's r array, which this leg does
// not modify: the new residual goes to the level's `d` scratch array (the up-leg reads it from there).
// Same expressions, same order, same bits as mg_down_fused<false, true>.
// ---------------------------------------------------------------------------------------------
__device__ __host__ __forceinline__ int down3_rpi(int m_) { return (64 / (m_ >> 2)) & ~1; }
__device__ __host__ __forceinline__ int down3_ns(int m_) {
  int ns = 8;
  while (ns < 2 * down3_rpi(m_) + 2) ns <<= 1;
  return ns;
}
__device__ __host__ __forceinline__ bool down3_shape_ok(int n, int m) {
  const int n_ = n - 2, m_ = m - 2;
  return (m_ & 3) == 0 && (n_ & 1) == 0 && (m_ >> 2) >= 6 && (m_ >> 2) <= 32 && n_ >= 8;
}
__device__ __host__ __forceinline__ int down3_lds_floats(int m) {
  const int m_ = m - 2;
  return 4 * (down3_ns(m_) * (m_ + 4) + 8) + 8;
}
__device__ AFC_DOWN_ATTR void down_v3c(const StepParams &P, const Lv &L0, const Lv &L1, float *lds) {
  const int n = uni(L0.n), m = uni(L0.m), n_ = n - 2, m_ = m - 2;
  const int CPR = m_ >> 2, PT = m_ + 4, RPI = down3_rpi(m_), NS = down3_ns(m_);
  const LevelDesc &D = *L0.desc;
  const float *rin = L0.r;
  float *xout = L0.x, *rout = L0.d, *r1 = L1.r;
  const float *lxp = L0.lx, *lyp = L0.ly, *invp = L0.inv;
  const int m1 = uni(L1.m);
  const float cu = uni(D.cu);
  float it[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) it[q] = uni(D.invtab[q]);
  const int sp_i0 = uni(D.sp_i0), sp_i1 = uni(D.sp_i1);
  const int wave = uni((int)(threadIdx.x >> 6));
  const int ppw = ((n_ >> 1) + 3) >> 2;  // row pairs per wave
  const int i_first = 1 + 2 * wave * ppw, i_last = (i_first + 2 * ppw - 1 < n_) ? i_first + 2 * ppw - 1 : n_;
  const unsigned ring = uni((int)(((lds_addr(lds) + 15u) & ~15u) + (unsigned)wave * ((unsigned)(NS * PT + 8) * 4u)));
  // ring element (row R, column jj) at float index (R & (NS-1))*PT + jj + 3: a lane's four columns are one aligned 16 bytes
#define DN_D4(R, c) (*(lds_f4 *)(uintptr_t)(ring + 4u * (unsigned)(((R) & (NS - 1)) * PT + 4 * (c) + 4)))
#define DN_D1(R, jj) (*(lds_f *)(uintptr_t)(ring + 4u * (unsigned)(((R) & (NS - 1)) * PT + (jj) + 3)))
  if (i_first <= n_) {
    const unsigned ln = lane_fresh();
    const int h = (int)((ln * ((65536u + (unsigned)CPR - 1u) / (unsigned)CPR)) >> 16), c4 = (int)ln - h * CPR;
    const bool lane_on = h < RPI;
    const int d_hi = (i_last + 1 < n_) ? i_last + 1 : n_;  // last row whose d this wave forms itself
    // r of the block starting at row R0 (rows outside [lo, hi] and idle lanes: zeros, never used)
    auto fetch = [&](int R0, int lo, int hi) {
      F4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v.v[e] = 0.f;
      const int R_ = R0 + h;
      if (lane_on && R_ >= lo && R_ <= hi) v = ld4b(rin, 4u * (unsigned)(R_ * m + 1 + 4 * c4));
      return v;
    };
    // step 1 on the block starting at row R0: d = r*inv into the ring
    auto step1 = [&](int R0, int lo, int hi, const F4 &rv) {
      const int R_ = R0 + h;
      if (lane_on && R_ >= lo && R_ <= hi) {
        const bool sp_ = (R0 + RPI - 1 >= sp_i0) && (R0 <= sp_i1);
        F4 iv;
        if (sp_) {
          iv = ld4b(invp, 4u * (unsigned)(R_ * m + 1 + 4 * c4));
        } else {
          const bool rf = (R_ == 1), rl = (R_ == n_);  // inv = invtab[3*rowclass + colclass] (LevelDesc), from scalar registers
          const float i0 = rf ? it[3] : (rl ? it[6] : it[0]);
          iv.v[0] = (c4 == 0) ? (rf ? it[4] : (rl ? it[7] : it[1])) : i0;
          iv.v[1] = i0;
          iv.v[2] = i0;
          iv.v[3] = (c4 == CPR - 1) ? (rf ? it[5] : (rl ? it[8] : it[2])) : i0;
        }
        f4a d4;
#pragma unroll
        for (int e = 0; e < 4; ++e) d4[e] = rv.v[e] * iv.v[e];  // MG.pde:80
        DN_D4(R_, c4) = d4;
        if (c4 == 0) DN_D1(R_, 0) = d4[0];  // d.setBC(): ghosts mirror the adjacent interior cell (Field.pde:211-224)
        if (c4 == CPR - 1) DN_D1(R_, m_ + 1) = d4[3];
        if (R_ == 1 || R_ == n_) {
          const int Rg_ = (R_ == 1) ? 0 : n_ + 1;
          DN_D4(Rg_, c4) = d4;
          if (c4 == 0) DN_D1(Rg_, 0) = d4[0];
          if (c4 == CPR - 1) DN_D1(Rg_, m_ + 1) = d4[3];
        }
      }
    };
    F4 c_r0, nx_r;
    {
      const int ra = i_first - RPI;  // the block whose last row is the row above the run
      const F4 ta = fetch(ra, (i_first - 1 > 1 ? i_first - 1 : 1), i_first - 1);
      c_r0 = fetch(i_first, i_first, d_hi);
      nx_r = fetch(i_first + RPI, i_first + RPI, d_hi);
      step1(ra, (i_first - 1 > 1 ? i_first - 1 : 1), i_first - 1, ta);
      step1(i_first, i_first, d_hi, c_r0);
    }
    asm volatile("" ::: "memory");
    for (int g = i_first; g <= i_last; g += RPI) {
      const F4 pf_r = fetch(g + 2 * RPI, g + 2 * RPI, d_hi);
      step1(g + RPI, g + RPI, d_hi, nx_r);
      // lanes read below what OTHER lanes of the wave have just stored: program order must survive the optimiser; in
      // hardware the LDS operations of a wave execute in order
      asm volatile("" ::: "memory");
      const int G_ = g + h, j = 1 + 4 * c4;
      const bool on2 = lane_on && G_ <= i_last;
      F4 ro;
#pragma unroll
      for (int e = 0; e < 4; ++e) ro.v[e] = 0.f;
      if (on2) {  // ---- step 2: rows g .. g+RPI-1 ----
        const unsigned kb = 4u * (unsigned)(G_ * m + j);
        const bool sp = (g + RPI - 1 >= sp_i0) && (g <= sp_i1);
        F4 lw, le, ls;
        float ln4;
        if (sp) {
          lw = ld4b(lxp, kb);
          le = ld4b(lxp, kb + 4u * (unsigned)m);
          ls = ld4b(lyp, kb);
          ln4 = ld1b(lyp, kb + 16u);
        } else {  // coef_analytic on a level with zeroed boundary faces (MG.pde:120)
          const float lwr = (G_ == 1) ? 0.f : cu, ler = (G_ == n_) ? 0.f : cu;
#pragma unroll
          for (int e = 0; e < 4; ++e) lw.v[e] = lwr, le.v[e] = ler, ls.v[e] = cu;
          if (c4 == 0) ls.v[0] = 0.f;
          ln4 = (c4 == CPR - 1) ? 0.f : cu;
        }
        const f4a dc = DN_D4(G_, c4), dw = DN_D4(G_ - 1, c4), de = DN_D4(G_ + 1, c4);
        const float dl = DN_D1(G_, j - 1), dr = DN_D1(G_, j + 4);
        const float d6[6] = {dl, dc[0], dc[1], dc[2], dc[3], dr};
        F4 xo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lwv = lw.v[e], lev = le.v[e], lsv = ls.v[e], lnv = (e < 3) ? ls.v[(e + 1) & 3] : ln4;
          const float diag = -(lwv + lev + lsv + lnv);
          const float d = d6[1 + e];
          const float ad = d * diag + dw[e] * lwv + de[e] * lev + d6[e] * lsv + d6[2 + e] * lnv;  // PoissonMatrix.pde:57-61
          ro.v[e] = c_r0.v[e] - ad;                                                              // MG.pde:96
          xo.v[e] = (0.f + 0.f) + d;                                                             // MG.pde:56,95
        }
        st4b(xout, kb, xo);
        st4b(rout, kb, ro);
        // ghosts mirror the adjacent interior d (d.setBC(), Field.pde:211-224)
        if (c4 == 0) st1b(xout, kb - 4u, (0.f + 0.f) + d6[1]);
        if (c4 == CPR - 1) st1b(xout, kb + 16u, (0.f + 0.f) + d6[4]);
        if (G_ == 1 || G_ == n_) {
          const unsigned kg = 4u * (unsigned)(((G_ == 1) ? 0 : (n - 1)) * m + j);
          F4 xg;
#pragma unroll
          for (int e = 0; e < 4; ++e) xg.v[e] = (0.f + 0.f) + d6[1 + e];
          st4b(xout, kg, xg);
          if (c4 == 0) st1b(xout, kg - 4u, (0.f + 0.f) + d6[1]);
          if (c4 == CPR - 1) st1b(xout, kg + 16u, (0.f + 0.f) + d6[4]);
        }
      }
      // restrict (MG.pde:124-137): sum of the 2x2 children in the reference order; the even row's residuals come from
      // the lanes one row further down (all lanes take part in the exchange)
      float up[4];
      {
        const int src = (int)(4u * (ln + (unsigned)CPR));
#pragma unroll
        for (int e = 0; e < 4; ++e)
          up[e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, ro.v[e])));
      }
      if (on2 && !(h & 1)) {
        const int ci = (G_ - 1) / 2 + 1, cj = 2 * c4 + 1;
        float two[2];
        two[0] = ro.v[0] + ro.v[1] + up[0] + up[1];
        two[1] = ro.v[2] + ro.v[3] + up[2] + up[3];
        __builtin_memcpy((char *)r1 + 4u * (unsigned)(ci * m1 + cj), two, 8);
      }
      asm volatile("" ::: "memory");
      c_r0 = nx_r;
      nx_r = pf_r;
    }
  }
#undef DN_D4
#undef DN_D1
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// The bottom of the V-cycle out of LDS.  Levels from LB on (the first level whose shorter side has at most 16 cells:
// 32x16 and below on the 16:8 grids, ~700 cells in all) cost next to nothing in arithmetic but, done level by level through
// global memory, a dozen barrier-separated passes with a round trip each per level -- 200 us of the 1 750 us sim-step
// at 1024 envs (profiles/r3_level_profile_E1024.txt).  Here everything those levels touch lives in LDS for the whole
// sub-cycle, eight floats per cell (ghost cells included), per level in four planes (see BLv):
//     { lw, le, ls, ln } | { inv, r } | x | d      lw..ln = the four face coefficients of the cell, gathered by the host
// The planes of all bottom levels are one contiguous table in global memory (StepParams::bottom_rec, shared by all
// envs, r = x = d = 0) that the workgroup copies into LDS with the LDS-DMA path: one round trip, no registers.
//
// The reference's whole-array operations (MG.pde:53-97, 124-152; the straightforward kernels of bdim_device.h are
// their transcription) are regrouped into passes that form every value with the same operands in the same order:
//   load    the record table; r of level LB with the setBC clamp the restriction ends with; d = r*inv
//   down l  B:  x = 0 + d;  r -= A d  (d of out-of-domain neighbours through the setBC mirror)        [smooth(0)]
//           CA: r_c = restrict(r) (+ setBC clamp), x_c = 0, d_c = r_c*inv_c on the whole coarse array
//   kc      the four ordered Gauss-Seidel sweeps (bottom_gs_wave)
//   up l    E:  with X(c) = x_c + d_c formed on the fly (the coarse level's last increment, never stored):
//               x += X(parent); r -= A prolongate(X); d = r*inv; then the four sweeps
//   store   x + d of level LB (interior; nothing reads a coarse level's ghost x) for the up-leg of the level above
// The ghost cells of d keep r_ghost*inv_ghost from the pass that formed r (nothing changes a ghost r afterwards), which
// is what the reference's d = r.times(inv) puts there before every run of sweeps.  r -= A d of a level's LAST
// increment is never formed: nothing reads that residual.
// ---------------------------------------------------------------------------------------------
__device__ __host__ __forceinline__ int bottom_first_level(const LevelDesc *lev, int nlev) {
  for (int l = 1; l < nlev; ++l) {
    const int a = lev[l].n - 2, b = lev[l].m - 2;
    if ((a < b ? a : b) <= 16) return l;
  }
  return nlev;  // none
}
// the sweeps march down the columns with one lane per column: at most 16 columns, no more columns than rows, and
// few enough cells (three per thread; the multiply-shift division of the passes is exact below 2048)
__device__ __host__ __forceinline__ bool bottom_shape_ok(const LevelDesc *lev, int nlev) {
  for (int l = bottom_first_level(lev, nlev); l < nlev; ++l) {
    const int a = lev[l].n - 2, b = lev[l].m - 2;
    if (b > 16 || b > a || b < 1 || lev[l].n * lev[l].m > 3 * kThreads) return false;  // <= 3 cells per thread
  }
  return true;
}
__device__ __host__ __forceinline__ int bottom_cells4(const LevelDesc &D) { return (D.n * D.m + 3) & ~3; }
__device__ __host__ __forceinline__ int bottom_lds_floats(const LevelDesc *lev, int nlev) {
  int f = 16;
  for (int l = bottom_first_level(lev, nlev); l < nlev; ++l) f += 8 * bottom_cells4(lev[l]);
  return f;
}

struct BLv {      // a bottom level in LDS (everything wave-uniform)
  int n, m, n_, m_, cnt;
  unsigned rcp;   // ceil(2^16 / m): k / m == (k * rcp) >> 16 for k < 2048, m <= 18
  unsigned rec;   // LDS byte address of the level's planes: CF (16 bytes per cell) | IR (8) | X (4) | D (4)
  unsigned ir, x, d;  // byte addresses of the IR, X and D planes
};
// The record's eight floats live in four PLANES per level (cells4 = the cell count rounded up to 4):
//   CF  {lw, le, ls, ln} per cell | IR  {inv, r} per cell | X | D
// so that what the lanes of a wave touch together lies together: as 32-byte records (round 3) every access of the passes
// (consecutive cells) and of the sweeps (cells 1 - m apart) had a stride of a multiple of 8 dwords -- 8 lanes per bank,
// 16 cycles for a dword access instead of 2, the LDS pipe as busy as the chain of dependent steps it was meant to feed.
enum { BR_INV = 4, BR_R = 5, BR_X = 6, BR_D = 7 };  // record fields after the four coefficients
template <int F>
__device__ __forceinline__ unsigned b_addr(const BLv &L, unsigned k) {
  static_assert(F >= BR_INV && F <= BR_D, "field");
  return F == BR_INV ? L.ir + 8u * k : (F == BR_R ? L.ir + 8u * k + 4u : (F == BR_X ? L.x + 4u * k : L.d + 4u * k));
}
#define B_REC4(L, k) (*(lds_f4 *)(uintptr_t)((L).rec + 16u * (unsigned)(k)))
#define B_F(L, k, f) (*(lds_f *)(uintptr_t)(b_addr<f>((L), (unsigned)(k))))
__device__ __forceinline__ BLv bottom_level(const StepParams &P, int LB, int l, unsigned base) {
  // offset and multiplier come ready-made from the host (LevelDesc::bot_off / bot_rcp): summing the levels above and
  // dividing here was a chain of dependent scalar loads plus an integer division in front of every pass
  (void)LB;
  BLv L;
  L.n = uni(P.lev[l].n);
  L.m = uni(P.lev[l].m);
  L.n_ = L.n - 2;
  L.m_ = L.m - 2;
  L.cnt = L.n * L.m;
  L.rcp = (unsigned)uni((int)P.lev[l].bot_rcp);
  L.rec = base + (unsigned)uni(P.lev[l].bot_off);
  const unsigned c4 = (unsigned)((L.cnt + 3) & ~3);
  L.ir = L.rec + 16u * c4;
  L.x = L.rec + 24u * c4;
  L.d = L.rec + 28u * c4;
  return L;
}

// Four in-place lexicographic Gauss-Seidel sweeps (MG.pde:81-89) by the calling wave alone, as pipelined hyperplanes:
// lane (s, c) = sweep s, column c+1 walks DOWN its column, one row per step, sweep s+1 kLagB = 3 steps behind sweep s.
//   d[i-1][j]  this lane's previous result (the ghost row's d for i = 1)
//   d[i][j-1]  the previous result of the lane to the left (DPP row_shr:1; column 1 keeps the ghost column's d)
//   d[i+1][j], d[i][j+1]  what the previous sweep stored two steps ago (the initial d for sweep 0): fetched, with the
//   cell's record, right after the stores of step t for step t+2 (same wave: LDS operations execute in order), so
//   no LDS round trip sits on the chain of dependent steps.
// One wave issues a dependent instruction every ~10 cycles, so the step is written for few instructions: ONE running
// address per lane (the record left of its cell; the row below is the next step's address), no clamping -- a lane
// above or below the grid reads whatever is there (reads outside the workgroup's LDS return 0) and stores to a
// record nobody reads (the ghost corner).
constexpr int kLagB = 3;
__device__ inline void bottom_gs_wave(const BLv &L) {
  const int lane = threadIdx.x & 63, s = lane >> 4, c = lane & 15;
  const unsigned pitch = 4u * (unsigned)L.m;  // bytes per row of the D plane
  const unsigned n_ = (unsigned)L.n_;
  const int steps = L.n_ + L.m_ - 1 + 3 * kLagB;
  const int i0 = 1 - kLagB * s - c;                                 // the row of this lane at step 0
  unsigned im1 = (c < L.m_) ? (unsigned)(i0 - 1) : 0x40000000u;     // row - 1: rows 1 .. n_ are the active ones
  const unsigned dummy = L.d;                                       // d of the ghost corner (0, 0): never read
  const float g0 = *(lds_f *)(uintptr_t)(L.d + 4u * (unsigned)(c + 1));  // d of the ghost row above the column
#define BG_F(a, o) (*(lds_f *)(uintptr_t)((a) + (o)))
  struct Ops {
    f4a cf;
    float iv, r, dE, dN, dS0;
  };
  // The lane's ONE running address: d of the cell LEFT of its cell (kL = i*m + c; the cell itself is kL + 1), in the D
  // plane.  The cell's coefficients and (inv, r) sit at 16 and 8 bytes per cell in their planes:
  //   &CF[kL + 1] = 4 * a + kcf,   &IR[kL + 1] = 2 * a + kir     (a = L.d + 4 kL; unsigned wrap-around is fine)
  const unsigned kcf = L.rec + 16u - 4u * L.d, kir = L.ir + 8u - 2u * L.d;
  // operands of the step whose cell is right of address a; an = the address one row further down
  auto fetch = [&](unsigned a, unsigned an) {
    Ops o;
    o.cf = *(lds_f4 *)(uintptr_t)(4u * a + kcf);
    const f2u sv = *(const lds_f2 *)(uintptr_t)(2u * a + kir);  // (inv, r)
    o.iv = sv.x;
    o.r = sv.y;
    o.dS0 = BG_F(a, 0u);
    o.dN = BG_F(a, 8u);
    o.dE = BG_F(an, 4u);
    return o;
  };
  unsigned wa = L.d + 4u * (unsigned)(i0 * L.m + c);  // may wrap below the plane (see above)
  unsigned wb = wa + pitch, nx = wb + pitch;
  Ops oa = fetch(wa, wb), ob = fetch(wb, nx);
  // both sets have landed before the loop is entered: its waits are then the steady state's (a fetch two steps old)
  asm volatile("" : "+v"(oa.cf), "+v"(oa.iv), "+v"(oa.r), "+v"(oa.dE), "+v"(oa.dN), "+v"(oa.dS0));
  asm volatile("" : "+v"(ob.cf), "+v"(ob.iv), "+v"(ob.r), "+v"(ob.dE), "+v"(ob.dN), "+v"(ob.dS0));
  float res = 0.f;
#define BGS_STEP(O, W)                                                                                                \
  {                                                                                                                   \
    const float dW = (im1 == 0u) ? g0 : res;                                                                          \
    const float dS = __builtin_bit_cast(                                                                              \
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, O.dS0), __builtin_bit_cast(int, res), 0x111, 0xf,  \
                                           0xf, false));                                                              \
    res = -(dW * O.cf[0] + O.dE * O.cf[1] + dS * O.cf[2] + O.dN * O.cf[3] - O.r) * O.iv; /* MG.pde:84-88 */           \
    BG_F((im1 < n_) ? W + 4u : dummy, 0u) = res;                                                                      \
    asm volatile("" ::: "memory"); /* the fetch below reads what OTHER lanes have just stored */                      \
    const unsigned ny = nx + pitch;                                                                                   \
    O = fetch(nx, ny);                                                                                                \
    __builtin_amdgcn_sched_barrier(0); /* keep the fetch here, two steps ahead of its use */                          \
    W = nx;                                                                                                           \
    nx = ny;                                                                                                          \
    ++im1;                                                                                                            \
  }
  for (int t = 0; t < steps; t += 2) {
    BGS_STEP(oa, wa)
    BGS_STEP(ob, wb)
  }
#undef BGS_STEP
#undef BG_F
}

__device__ AFC_BOT_ATTR void mg_bottom(const StepParams &P, size_t e, int LB, float *lds, ProfT &pf) {
  const int kc = P.nlev - 1;
  const int tid = threadIdx.x;
  const unsigned base = (lds_addr(lds) + 15u) & ~15u;
  __syncthreads();  // the region's previous user is done; r of level LB (written by the level above) is visible
  // ---- load: the record table by LDS-DMA (16 bytes per lane and copy); r of level LB; d = r*inv ----
  {
    const BLv S = bottom_level(P, LB, LB, base);
    const Lv G_ = uni(make_level(P, e, LB));
    const int per = (S.cnt + kThreads - 1) / kThreads;
    float rv[3];  // bottom_shape_ok: level LB has at most 34 x 18 cells
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int k = tid + q * kThreads;
      rv[q] = 0.f;
      if (q < per && k < S.cnt) {  // the level above stored the interior of its restricted residual; restrict() ends with setBC
        const int i = (int)(((unsigned)k * S.rcp) >> 16), j = k - i * S.m;
        rv[q] = G_.r[clampi(i, 1, S.n_) * S.m + clampi(j, 1, S.m_)];
      }
    }
    unsigned chunks = 0;  // 16-byte pieces of the table
    for (int l = LB; l <= kc; ++l) chunks += 2u * (unsigned)((uni(P.lev[l].n) * uni(P.lev[l].m) + 3) & ~3);
    const float *tab = uni(G(P.bottom_rec));
    const unsigned lane = threadIdx.x & 63u, wv = (unsigned)uni((int)(threadIdx.x >> 6));
    for (unsigned c0 = 64u * wv; c0 < chunks; c0 += 64u * (kThreads / 64)) {
      if (c0 + lane < chunks) lds_dma16(tab, 16u * (c0 + lane), base + 16u * c0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int k = tid + q * kThreads;
      if (q < per && k < S.cnt) {
        const int i = (int)(((unsigned)k * S.rcp) >> 16), j = k - i * S.m;
        B_F(S, k, BR_D) = rv[q] * B_F(S, k, BR_INV);  // MG.pde:80
        if ((unsigned)(i - 1) < (unsigned)S.n_ && (unsigned)(j - 1) < (unsigned)S.m_) B_F(S, k, BR_R) = rv[q];
      }
    }
  }
  lds_barrier();
  pf.mark(24 + LB);  // diagnostics: load | down passes | up passes | sweeps -> down_L<LB>, down_L<LB+1>, up_L<LB>, up_L<LB+1>
  // ---- descending leg (MG.pde:69-70) ----
  for (int l = LB; l < kc; ++l) {
    const BLv F = bottom_level(P, LB, l, base), C = bottom_level(P, LB, l + 1, base);
    for (int k = tid; k < F.cnt; k += kThreads) {  // B: d.setBC(); increment()  (MG.pde:90-97)
      const int i = (int)(((unsigned)k * F.rcp) >> 16), j = k - i * F.m;
      if ((unsigned)(i - 1) < (unsigned)F.n_ && (unsigned)(j - 1) < (unsigned)F.m_) {
        const f4a cf = B_REC4(F, k);
        const int iw = (i - 1 < 1) ? 1 : i - 1, ie = (i + 1 > F.n_) ? F.n_ : i + 1;
        const int js = (j - 1 < 1) ? 1 : j - 1, jn = (j + 1 > F.m_) ? F.m_ : j + 1;
        const float dc = B_F(F, k, BR_D), dw = B_F(F, iw * F.m + j, BR_D), de = B_F(F, ie * F.m + j, BR_D),
                    ds = B_F(F, i * F.m + js, BR_D), dn = B_F(F, i * F.m + jn, BR_D);
        const float diag = -(cf[0] + cf[1] + cf[2] + cf[3]);  // PoissonMatrix.pde:45-48
        const float ad = dc * diag + dw * cf[0] + de * cf[1] + ds * cf[2] + dn * cf[3];  // PoissonMatrix.pde:57-61
        B_F(F, k, BR_R) = B_F(F, k, BR_R) - ad;
        B_F(F, k, BR_X) = 0.f + dc;
      }
    }
    lds_barrier();
    for (int k = tid; k < C.cnt; k += kThreads) {  // CA: restrict + setBC (MG.pde:124-137); d = r*inv (MG.pde:80)
      const int i = (int)(((unsigned)k * C.rcp) >> 16), j = k - i * C.m;
      const int ci = clampi(i, 1, C.n_), cj = clampi(j, 1, C.m_);
      const int kf = ((ci - 1) * 2 + 1) * F.m + (cj - 1) * 2 + 1;
      const float rc = B_F(F, kf, BR_R) + B_F(F, kf + 1, BR_R) + B_F(F, kf + F.m, BR_R) + B_F(F, kf + F.m + 1, BR_R);
      B_F(C, k, BR_D) = rc * B_F(C, k, BR_INV);
      if (ci == i && cj == j) {
        B_F(C, k, BR_R) = rc;
        B_F(C, k, BR_X) = 0.f;
      }
    }
    lds_barrier();
  }
  pf.mark(24 + LB + 1);
  // ---- coarsest level: smooth(its) (MG.pde:73); ascending leg (MG.pde:75-76) + the caller's smooth(its) ----
  for (int l = kc; l >= LB; --l) {
    const BLv F = bottom_level(P, LB, l, base);
    if (l < kc) {
      const BLv C = bottom_level(P, LB, l + 1, base);
      for (int k = tid; k < F.cnt; k += kThreads) {  // E
        const int i = (int)(((unsigned)k * F.rcp) >> 16), j = k - i * F.m;
        if ((unsigned)(i - 1) < (unsigned)F.n_ && (unsigned)(j - 1) < (unsigned)F.m_) {
          const int iw = (i - 1 < 1) ? 1 : i - 1, ie = (i + 1 > F.n_) ? F.n_ : i + 1;
          const int js = (j - 1 < 1) ? 1 : j - 1, jn = (j + 1 > F.m_) ? F.m_ : j + 1;
          // X at the parents of the cell and of its four (clamped) neighbours: MG.pde:139-152 after MG.pde:95
          const int pi = ((i - 1) >> 1) + 1, pj = ((j - 1) >> 1) + 1;
          const int pw = ((iw - 1) >> 1) + 1, pe = ((ie - 1) >> 1) + 1, ps = ((js - 1) >> 1) + 1, pn = ((jn - 1) >> 1) + 1;
          const int kcC = pi * C.m + pj, kw = pw * C.m + pj, ke = pe * C.m + pj, ks = pi * C.m + ps, kn = pi * C.m + pn;
          const float xc = B_F(C, kcC, BR_X) + B_F(C, kcC, BR_D), xw = B_F(C, kw, BR_X) + B_F(C, kw, BR_D),
                      xe = B_F(C, ke, BR_X) + B_F(C, ke, BR_D), xs = B_F(C, ks, BR_X) + B_F(C, ks, BR_D),
                      xn = B_F(C, kn, BR_X) + B_F(C, kn, BR_D);
          const f4a cf = B_REC4(F, k);
          const float iv = B_F(F, k, BR_INV), ro = B_F(F, k, BR_R), xo = B_F(F, k, BR_X);
          const float diag = -(cf[0] + cf[1] + cf[2] + cf[3]);
          const float ad = xc * diag + xw * cf[0] + xe * cf[1] + xs * cf[2] + xn * cf[3];
          const float rn = ro - ad;
          B_F(F, k, BR_R) = rn;
          B_F(F, k, BR_X) = xo + xc;
          B_F(F, k, BR_D) = rn * iv;  // MG.pde:80
        }
      }
      lds_barrier();
      pf.mark(16 + LB);
    }
    if (tid < 64) {
      AFC_PRIO_RAISE();  // a serial chain, like the march
      bottom_gs_wave(F);
      AFC_PRIO_DROP();
    }
    lds_barrier();
    pf.mark(16 + LB + 1);
  }
  {  // x + d of the first bottom level to global memory: the up-leg of the level above prolongates from it
    const Lv G_ = uni(make_level(P, e, LB));
    const BLv S = bottom_level(P, LB, LB, base);
    for (int k = tid; k < S.cnt; k += kThreads) {
      const int i = (int)(((unsigned)k * S.rcp) >> 16), j = k - i * S.m;
      if ((unsigned)(i - 1) < (unsigned)S.n_ && (unsigned)(j - 1) < (unsigned)S.m_)
        G_.x[k] = B_F(S, k, BR_X) + B_F(S, k, BR_D);
    }
  }
  __syncthreads();
}
#undef B_REC4
#undef B_F

// MGsolver (MG.pde:30-38): fused head (first iteration), straightforward descending leg on the
// coarse levels, fused ascending leg.  L0.x is the buffer the solution is left in.
#if AFC_TR
// mg_up_tr by the level's width (16 CB interior columns)
__device__ __forceinline__ void up_tr(const bool HASC, const Lv &F, const Lv &Cc, bool need_r, float *lds, float *rr_out,
                                      float *linf_out, ProfT &pf, float *xsum_out, const int rmode) {
  const unsigned la = lds_addr(lds);
  TrOut o;
  if (F.m == 130)
    o = mg_up_tr<8>(HASC, F, Cc, need_r, la, rmode);
  else if (F.m == 66)
    o = mg_up_tr<4>(HASC, F, Cc, need_r, la, rmode);
  else
    o = mg_up_tr<2>(HASC, F, Cc, need_r, la, rmode);
  if (need_r || (AFC_DIAG && ((F.desc->dbg_copy >> 20) == F.n))) pf.mark(PF_GS0);
  if (need_r && rmode != 2) {
    *rr_out = o.rr;
    *linf_out = o.linf;
    if (xsum_out) *xsum_out = o.xs;
  }
}
#endif

// (forced inline: as a function of its own it would not inherit the kernels' register cap -- launch bounds apply to
// kernels only -- and its 248 VGPRs would halve the occupancy of every kernel that calls it)
// H0: strip height of the finest level's up-leg (up_strip_h of the kernel instantiation)
template <int H0 = 16>
__device__ __forceinline__ int mg_solve_v2(const StepParams &P, size_t e, const Lv &L0, const float *ux, const float *uy,
                                  const float *pin, float shift, double *sm, float *linf_out, float *xsum_out,
                                  ProfT &pf, float *lds, bool fused_head = true, bool hard = false) {
  const int kc = P.nlev - 1;
  // levels LB .. kc run out of LDS (mg_bottom) where the host found room for them; LB > kc: level by level as before
  const int LB = (!AFC_RB && uni(P.bottom_lb) >= 2 && !(AFC_DBG_OF(P) & 128)) ? uni(P.bottom_lb) : kc + 1;
  // coarse down-legs as per-wave row pipelines (down_v3c) where the host found room for their rings; the experiments
  // that use a level's `d` scratch array themselves (dbg bit 2) and dbg bit 4 keep the blocked form
  const bool down3 = uni(P.down_v3) != 0 && !(AFC_DBG_OF(P) & (4 | 16));
  int iter = 0;
  float linf = 0.f, xsum = 0.f;
  while (iter < P.mg_max_iters) {
    ++iter;
    for (int l = 0; l < kc && l < LB; ++l) {  // descending leg of vCycle (MG.pde:69-70)
      const Lv F = uni((l == 0) ? L0 : make_level(P, e, l)), Cc = uni(make_level(P, e, l + 1));
      if (l == 0 && iter == 1 && fused_head) {
        if (F.m == 130 && ((F.n - 2) & 7) == 0 && uni(P.head_v3) && !(AFC_DBG_OF(P) & 2048))
          head_v3(P, L0, Cc, ux, uy, pin, shift, lds);
        else if (H0 != 16 && uni(P.head_v4) && head4_shape_ok(F.n, F.m) && !(AFC_DBG_OF(P) & 2048))
          head_v4(P, L0, Cc, ux, uy, pin, shift, lds);  // (H0 != 16: only the instantiations for grids wider than 128 columns carry it)
        else
          mg_down_fused<true, false>(P, L0, Cc, ux, uy, pin, shift, lds);
      } else if (((F.m - 2) & 3) == 0 && ((F.n - 2) & 1) == 0) {
        if (l == 0)
          mg_down_fused<false, false>(P, F, Cc, nullptr, nullptr, nullptr, 0.f, lds);
        else if (down3 && down3_shape_ok(F.n, F.m))  // barrier-free row pipelines; the new residual lands in F.d
          down_v3c(P, F, Cc, lds);
        else  // coarse levels start from x = 0 (MG.pde:56): folded into the leg, never stored or re-read
          mg_down_fused<false, true>(P, F, Cc, nullptr, nullptr, nullptr, 0.f, lds);
      } else {
        if (l > 0) {  // x = 0 on entry (MG.pde:56); the fused legs fold this into their arithmetic
          for (int q = threadIdx.x; q < F.n * F.m; q += kThreads) F.x[q] = 0.f;
          __syncthreads();
        }
        mg_dinv(F);  // smooth(0)
        mg_increment(F);
        mg_restrict(F, Cc);
      }
      pf.mark(l == 0 ? PF_DOWN0 : (24 + l));
      if (AFC_DIAG && l == 0 && iter == 1 && fused_head) gate_release(P);  // the streaming part of this half-step ends here
      if (l == 0 && pf.stop == 3) return iter;
    }
    float rr = 0.f;
    if (LB <= kc) {
      mg_bottom(P, e, LB, lds, pf);
      pf.mark(16 + kc);
    } else if (!AFC_RB) {
      const Lv Lk = uni(make_level(P, e, kc));  // coarsest: coarse.smooth(its), MG.pde:73
      for (int q = threadIdx.x; q < Lk.n * Lk.m; q += kThreads) Lk.x[q] = 0.f;  // MG.pde:56
#if AFC_DIAG
      if (AFC_DBG_OF(P) & 4)
        mg_up_smooth_mw(false, Lk, Lk, false, lds, sm, &rr, &linf, pf);
      else
#endif
        mg_up_smooth<16>(false, Lk, Lk, false, lds, sm, &rr, &linf, pf);
      pf.mark(16 + kc);
    }
    // (perf mode: the loop starts at the coarsest level itself -- coarse.smooth(its) from x = 0, the same call with
    // hasc = false -- so that the up-leg has ONE call site there: it is inlined, and it is large)
    for (int l = (LB <= kc ? LB - 1 : (AFC_RB ? kc : kc - 1)); l >= 0; --l) {
      if (l == 0 && pf.stop == 4) return iter;
      const bool hasc = (l < kc);
      Lv F = uni((l == 0) ? L0 : make_level(P, e, l));
      const Lv Cc = hasc ? uni(make_level(P, e, l + 1)) : F;
      if (!hasc)
        for (int q = threadIdx.x; q < F.n * F.m; q += kThreads) F.x[q] = 0.f;  // MG.pde:56 (the up-leg opens with a barrier)
      if (hasc && l > 0 && down3 && down3_shape_ok(F.n, F.m) && ((F.m - 2) & 3) == 0 && ((F.n - 2) & 1) == 0) F.r = F.d;  // where down_v3c left r
      // the finest level's new r is only read again by a further V-cycle (or by the sequential
      // inner product of the parity mode): normally it is not written back at all
      // hard: the caller expects this solve to need a second V-cycle (the predictor's solve of the first sim-step after an
      // action change -- where ALL second cycles of the bench workload occur, 4.7 % of the envs per agent step): its
      // residual is written back right away, so that the stragglers of a launch are spared the replay of the level
      const bool lazy_r = (l == 0) && !P.exact_sums && (((F.m - 2) & 3) == 0) && !(AFC_DBG_OF(P) & 2) && !hard;
      // the up-leg as a software pipeline (mg_up_pipe: the march back to back, waves 1-3 exiting / filling behind it)
      // where the level has its shape; dbg bit 8: strip by strip with workgroup barriers (mg_up_smooth) everywhere
      // (only in the kernels of the grids up to 128 columns, H0 == 16: in the two wide-grid instantiations the same code
      // on the coarse levels measured +0.3 % (512x256) and +23.7 % (384x192) per launch -- their register allocation
      // answers the extra branch with reloads inside the loops; they keep round 4's up-leg unchanged)
      // round 6: the sweeps marching down the rows (mg_up_tr) where the host found the level's shape and room for its rings
      // (LevelDesc::tm) -- in afc::lat, the instantiation for launches of few envs per CU, and selectable in afc::diag
      // (afc_config.upleg: 1 = the strip pipeline, 2 = the barrier form, else mg_up_tr)
      const bool tr = AFC_TR && !AFC_RB && H0 == 16 && ((uni(F.desc->tm) >> (AFC_DIAG ? 1 : 0)) & 1) != 0 &&
                      !(AFC_DBG_OF(P) & (256 | 4)) && !(AFC_DIAG && (uni(P.upleg) == 1 || uni(P.upleg) == 2));
      const bool pipe = !(AFC_TR && !AFC_DIAG) && !AFC_RB && H0 == 16 && up_pipe_shape_ok(F.n, F.m) && !(AFC_DBG_OF(P) & 256) &&
                        !(AFC_DIAG && uni(P.upleg) == 2);
#if AFC_RB
      // perf mode: the register-resident row pipelines where the level has their shape; on the finest level only when
      // the new residual stays unwritten (a wave reads the old r of its neighbours' rows)
      if (uni(F.desc->rbw) != 0 && (l > 0 || lazy_r))
        mg_up_rbw(hasc, F, Cc, l == 0, lds, &rr, &linf, &xsum, lazy_r ? 1 : 0);
      else
#endif
#if AFC_DIAG
      if (AFC_DBG_OF(P) & 4)
        mg_up_smooth_mw(hasc, F, Cc, l == 0, lds, sm, &rr, &linf, pf, &xsum, lazy_r ? 1 : 0);
      else
#endif
#if AFC_TR
      if (tr)
        up_tr(true, F, Cc, l == 0, lds, &rr, &linf, pf, &xsum, lazy_r ? 1 : 0);
      else
#endif
      if (pipe)
        mg_up_pipe(true, F, Cc, l == 0, lds, sm, &rr, &linf, pf, &xsum, lazy_r ? 1 : 0);
      else if (H0 != 16 && l == 0)
        mg_up_smooth<H0>(true, F, Cc, true, lds, sm, &rr, &linf, pf, &xsum, lazy_r ? 1 : 0);
      else
        mg_up_smooth<16>(hasc, F, Cc, l == 0, lds, sm, &rr, &linf, pf, &xsum, lazy_r ? 1 : 0);
      if (lazy_r && !(rr < P.mg_tol) && iter < P.mg_max_iters) {
        float rr2, linf2;  // replay the level: identical d, r -= A d stored this time
#if AFC_DIAG
        if (AFC_DBG_OF(P) & 4)
          mg_up_smooth_mw(true, F, Cc, true, lds, sm, &rr2, &linf2, pf, nullptr, 2);
        else
#endif
#if AFC_TR
        if (tr)
          up_tr(true, F, Cc, true, lds, &rr2, &linf2, pf, nullptr, 2);
        else
#endif
        if (pipe)
          mg_up_pipe(true, F, Cc, true, lds, sm, &rr2, &linf2, pf, nullptr, 2);
        else
          mg_up_smooth<H0>(true, F, Cc, true, lds, sm, &rr2, &linf2, pf, nullptr, 2);
      }
      pf.mark(l == 0 ? PF_RR : (16 + l));
    }
    if (P.exact_sums) rr = seq_inner(L0, sm);
    if (rr < P.mg_tol) break;
  }
  if (AFC_DIAG && fused_head) gate_acquire(P);  // projection tail and the next advection follow
  *linf_out = linf;
  *xsum_out = xsum;
  return iter;
}

// What is left of u.setBC() (Field.pde:204-224 as transcribed by setbc_vel) once every row 1 .. n-2 carries its wall
// columns (the folded stores of advdif_v3 / project_tail_v2): the inflow / outflow rows.  Row 0 and row n-1 become
// copies of the folded rows 1 and n-2 -- the same values setbc_vel's column pass leaves in them --, u.x of row 1
// becomes the inflow value 1, and the interior of the last row of u.x gets the flux correction 1 - mean(outflow).
// No strided column access is left: the column pass of setbc_vel touched one 64-byte line per row, side and
// component to move four floats.
__device__ inline void setbc_rows(const StepParams &P, float *ux, float *uy, float *rowbuf) {
  __shared__ float s_mean2;
  const int n = uni(P.n), m = uni(P.m);
  for (int j = threadIdx.x; j < m; j += kThreads) {
    ux[j] = ux[m + j];
    const float o = ux[(n - 2) * m + j];
    rowbuf[j] = o;
    ux[m + j] = 1.f;
    uy[j] = uy[m + j];
    uy[(n - 1) * m + j] = uy[(n - 2) * m + j];
  }
  lds_barrier();
  if (threadIdx.x == 0) {  // Field.pde:302-318 order (one wave adding the row through v_readlane: 0.25 % SLOWER on the same memory)
    // eight values per LDS round trip, then eight adds in the same order: as `s += rowbuf[j]` the loop was one LDS read,
    // a full wait and a branch per element -- ~7 us per call with the other three waves at the barrier
    float s = 0;
    int j = 1;
    for (; j + 8 <= m - 1; j += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = rowbuf[j + q];
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; j < m - 1; ++j) s += rowbuf[j];
    s /= (float)(m - 2);
    s_mean2 = s;
  }
  lds_barrier();
  const float corr = 1.f - s_mean2;
  for (int j = threadIdx.x; j < m; j += kThreads) {
    const float o = rowbuf[j];
    ux[(n - 1) * m + j] = (j >= 1 && j <= m - 2) ? o + corr : o;
  }
  __syncthreads();
}

// Tail of u.project (VectorField.pde:136-140) [+ the Heun average of BDIM.update2, BDIM.pde:95-96]:
//   p' = p + s (s = -sum(p)/N);  u -= c * grad p';  u.setBC();  [A = (u + us)/2]
// p' is NOT materialised: p stays as the solver left it and s is returned as the env's pending
// shift; every reader of p (next solve's head, force, probes, field export) forms p+s itself --
// the same float32 addition the reference's plusEq performs, so the bits are identical.
__device__ AFC_TAIL_ATTR void project_tail_v2(const bool HEUN, const StepParams &P, float *lds, const float *__restrict__ p,
                                       float shift, float *vx,
                                       float *vy, const float *__restrict__ ox, const float *__restrict__ oy,
                                       float *ax, float *ay, const bool fold) {
  const int n = uni(P.n), m = uni(P.m), n_ = n - 2, m_ = m - 2, CPR = m_ >> 2;
  const int sp0 = uni(P.lev[0].sp_i0), sp1 = uni(P.lev[0].sp_i1);
  const float cu = uni(P.lev[0].cu);
  const float *pcx = uni(G(P.cx)), *pcy = uni(G(P.cy));
  // a thread walks down one 4-column chunk through its group's rows, so the row above (p west of
  // the x-faces) is the chunk it read one iteration earlier and p crosses HBM once
  // (chunk counts that are powers of two -- every grid but 384x192 -- split by shifts: three integer divisions less at the
  // head of the phase)
  const bool cpr_p2 = (CPR & (CPR - 1)) == 0 && CPR <= kThreads;
  const int cpr_sh = cpr_p2 ? __builtin_ctz((unsigned)CPR) : 0;
  const int ngrp = cpr_p2 ? (kThreads >> cpr_sh) : kThreads / CPR;
  const int rg = cpr_p2 ? ((int)threadIdx.x >> cpr_sh) : (int)threadIdx.x / CPR, c4 = threadIdx.x - rg * CPR;
  const int rpg = cpr_p2 ? ((n_ + ngrp - 1) >> (8 - cpr_sh)) : (n_ + ngrp - 1) / ngrp;
  const int g_beg = 1 + rg * rpg, g_end = (rg >= ngrp) ? 0 : ((g_beg + rpg - 1 > n_) ? n_ : g_beg + rpg - 1);
  F4 pprev;
  for (int g = g_beg; g <= g_end; ++g) {
    const int k = g * m + 1 + 4 * c4;
    const F4 pc = load4u(p + k);
    if (g == g_beg) pprev = load4u(p + k - m);
    const F4 pw = pprev;
    pprev = pc;
    const float psl = p[k - 1];
    const F4 u4 = load4u(vx + k), v4 = load4u(vy + k);
    F4 cx4, cy4;
    if (g >= sp0 && g <= sp1) {  // rows that cross a body: the real coefficient arrays
      cx4 = load4u(pcx + k);
      cy4 = load4u(pcy + k);
    } else {  // verified on the host: every face coefficient of the row equals cu (LevelDesc)
#pragma unroll
      for (int e = 0; e < 4; ++e) cx4.v[e] = cy4.v[e] = cu;
    }
    F4 un, vn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pcs = pc.v[e] + shift, pws = pw.v[e] + shift;
      const float pss = ((e > 0) ? pc.v[(e + 3) & 3] : psl) + shift;
      const float gx = pcs - pws, gy = pcs - pss;                // Field.pde:76-77
      un.v[e] = u4.v[e] + cx4.v[e] * (gx * -1.f);                // VectorField.pde:138
      vn.v[e] = v4.v[e] + cy4.v[e] * (gy * -1.f);
    }
    // The corrector's projected field C is only an intermediate of the Heun average: after this pass nothing reads
    // its interior again (the next step's advection overwrites every interior cell before use; the wall columns of
    // A are formed from registers right here).  Only rows 1 and n-2 are still read: by setBC's row copies and outflow
    // mean and by the 4-row fix-up below.
    if (!HEUN || g == 1 || g == n_) {
      if (fold) {  // the row with the wall columns setBC gives it (see advdif_v3); setbc_rows does the rest
        if (c4 == 0) {
          vx[k - 1] = un.v[0];
          vy[k - 1] = vn.v[0];
        }
        if (c4 == CPR - 1) {
          vx[k + 4] = un.v[3];
          vy[k + 4] = 0.f;
        }
        F4 vs = vn;
        if (c4 == 0) vs.v[0] = 0.f;
        store4u(vx + k, un);
        store4u(vy + k, vs);
      } else {
        store4u(vx + k, un);
        store4u(vy + k, vn);
      }
    }
    if (HEUN) {  // cells on the rows setBC rewrites (0, 1, n-2, n-1) are redone below
      const F4 b4 = load4u(ox + k), d4 = load4u(oy + k);
      F4 a4, e4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a4.v[e] = (un.v[e] + b4.v[e]) * 0.5f;
        e4.v[e] = (vn.v[e] + d4.v[e]) * 0.5f;
      }
      // wall columns: the average uses what setBC is about to put there (Field.pde:217-224):
      // u.x mirrors the adjacent interior cell, u.y[.][0] = old u.y[.][1], u.y[.][1] = u.y[.][m-1] = 0
      if (c4 == 0) {
        ax[k - 1] = (un.v[0] + ox[k - 1]) * 0.5f;
        ay[k - 1] = (vn.v[0] + oy[k - 1]) * 0.5f;
        e4.v[0] = (0.f + d4.v[0]) * 0.5f;
      }
      if (c4 == CPR - 1) {
        ax[k + 4] = (un.v[3] + ox[k + 4]) * 0.5f;
        ay[k + 4] = (0.f + oy[k + 4]) * 0.5f;
      }
      store4u(ax + k, a4);
      store4u(ay + k, e4);
    }
  }
  __syncthreads();
  if (fold)
    setbc_rows(P, vx, vy, lds);
  else
    setbc_vel(P, vx, vy, lds);
  if (HEUN) {  // whole rows touched by setBC's row part or by its mean correction: 0, 1, n-2, n-1
    for (int q = threadIdx.x; q < 4 * m; q += kThreads) {
      const int rsel = (q >= m) + (q >= 2 * m) + (q >= 3 * m), j = q - rsel * m;  // q / m for q < 4 m
      const int k = ((rsel < 2) ? rsel : (n - 4 + rsel)) * m + j;
      ax[k] = (vx[k] + ox[k]) * 0.5f;
      ay[k] = (vy[k] + oy[k]) * 0.5f;
    }
    __syncthreads();
  }
}

// u.project(c, p) (VectorField.pde:130-143).  pin holds the previous pressure (+ pending shift),
// the result is left in pout (a different buffer) with a new pending shift *shift_io.
template <int C, int TAG = 0>
__device__ inline int project_v2(const bool HEUN, const StepParams &P, size_t e, float *vx, float *vy,
                                 const float *pin, float *pout,
                                 float *shift_io, const float *ox, const float *oy, float *ax, float *ay, double *sm,
                                 float *linf_out, ProfT &pf, float *lds, bool hard = false, bool skip_tail = false) {
  const size_t eo = e * (size_t)P.Ns;
  Lv L0 = make_level(P, e, 0);
  L0.x = pout;
  L0.d = G(P.u0y) + eo;  // scratch of the straightforward kernels (rare 2nd MG iteration)
  float xsum = 0.f;
  const int iters = mg_solve_v2<up_strip_h(C)>(P, e, L0, vx, vy, pin, *shift_io, sm, linf_out, &xsum, pf, lds, true, hard);
  if (pf.stop >= 3 && pf.stop <= 5) return iters;
  float psum = xsum;
  if (P.exact_sums) psum = seq_sum(pout, P.n, P.m, sm);  // Field.sum in the reference's order
  const float shift = uni(-1 * psum / (float)((P.n - 2) * (P.m - 2)));  // VectorField.pde:136
  // (skip_tail: the caller's next advection applies the tail to the rows it stages -- advdif_v3f)
  if (!skip_tail)
    project_tail_v2(HEUN, P, lds, pout, shift, vx, vy, ox, oy, ax, ay, uni(P.bc_fold) != 0 && !(AFC_DBG_OF(P) & 0x80000000u));
  *shift_io = shift;
  pf.mark(PF_PROJECT);
  return iters;
}

// Body.pressForce(flow.p).mult(-1) of the main cylinder (Body.pde:296-303, AFCCylinder.pde:58) by ONE WAVE: lane s
// samples segment s (p.linear at its centre) and forms the two products pdl*nx, pdl*ny; the 40 products are then added
// in segment order, exactly as the reference's loop accumulates them (same operands, same order, same bits as
// eval_force), through v_readlane -- no LDS, no serial chain of dependent global loads on one thread.
__device__ __forceinline__ void eval_force_wave(const StepParams &P, const float *p, float shift, float *fxo,
                                                float *fyo) {
  const int nseg = uni(P.nseg);
  if (nseg > 64) {  // more segments than lanes: the serial form
    eval_force(P, p, shift, fxo, fyo);
    return;
  }
  const int lane = threadIdx.x & 63, m = uni(P.m);
  float tx = 0.f, ty = 0.f;
  if (lane < nseg) {
    const SampleDev sd = G(P.fs)[lane];
    const float *a = p + sd.base;
    float val;
    if (sd.exact) {
      val = a[0] + shift;
    } else {
      val = sd.s * (sd.t * (a[m + 1] + shift) + (1 - sd.t) * (a[m] + shift)) +
            (1 - sd.s) * (sd.t * (a[1] + shift) + (1 - sd.t) * (a[0] + shift));
    }
    const float pdl = val * G(P.seg_l)[lane];
    tx = pdl * G(P.seg_nx)[lane];
    ty = pdl * G(P.seg_ny)[lane];
  }
  float px = 0, py = 0;
  for (int s = 0; s < nseg; ++s) {
    px += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tx), s));
    py += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ty), s));
  }
  *fxo = px * -1.f;
  *fyo = py * -1.f;
}

// (forced inline, like mg_solve_v2: outlined, the diagnostic instantiation's copy would be shared by k_sim_step and
// k_agent_step and allocated without their register cap)
template <int C>
__device__ __forceinline__ void sim_step_v2(const StepParams &P, size_t e, float xi1_m, float xi2_m, float *pshift, double *sm,
                                   ShBcast *sh, float *lds, const bool first_after_action = false, const bool heun_in = false,
                                   const bool defer_out = false) {
  // heun_in: the sim-step before this one left out the tail of its corrector's projection and the Heun average
  // (BDIM.pde:95-96): u in memory is stale, this step's predictor advection forms it from F2, p and us as it goes
  // (advdif_v3f<C, true>).  defer_out: this step leaves them out in turn -- the caller promises that another sim-step
  // of the same launch follows (heun_fusable() says whether this env's kernels can do either).
  const size_t eo = e * (size_t)P.Ns;
  const float dphi1 = uni((2 * xi1_m * P.dt) / (P.dR * P.D));  // AFCCylinder.pde:48-49
  const float dphi2 = uni((2 * xi2_m * P.dt) / (P.dR * P.D));
  float *Ax = uni(G(P.ux) + eo), *Ay = uni(G(P.uy) + eo), *Bx = uni(G(P.usx) + eo), *By = uni(G(P.usy) + eo),
        *Cx = uni(G(P.fx) + eo), *Cy = uni(G(P.fy) + eo), *Qx = uni(G(P.qx) + eo), *Qy = uni(G(P.qy) + eo);
  float *p0 = uni(G(P.p) + eo), *p1 = uni(G(P.d) + eo);  // pressure ping-pong: p0 -> p1 (predictor) -> p0 (corrector)
  float l0, l1;
  float shift = uni(*pshift);
  ProfT pf;
  pf.start(P);
#if AFC_DIAG
  pf.stop = uni((P.dbg >> 12) & 15);
#endif
  // half 0 = predictor, BDIM.update(): F = AdvDif(u; u0 = u); updateUP                    -> B
  // half 1 = corrector, BDIM.update2(): us = u (= B); F = AdvDif(u; u0); updateUP; (u+us)/2 -> C, A
  int its[2] = {0, 0};
  float lin[2] = {0.f, 0.f};
  // the barrier-free row pipelines where they apply (adv_ns = ring depth chosen by the host; dbg bit 10 forces the tiled form), else the tiled form
  const int adv3_ns = uni(P.adv_ns);
  const bool adv3_ok = adv3_ns == kAdvNS && !(AFC_DBG_OF(P) & 1024);
  // the predictor's projection tail inside the corrector's advection (advdif_v3f): the projected predictor field is then
  // Q, not B
  const int fuse_bits = (!AFC_DIAG && adv3_ok && uni(P.bc_fold) != 0 && advf_ok_for(uni(P.n), uni(P.m), C)) ? uni(P.fuse_tail) : 0;
  const bool fuse = (fuse_bits & 1) != 0;    // the predictor's tail inside the corrector's advection: the projected predictor field is Q
  // the corrector's tail + Heun average inside the next predictor's advection -- only together with the first: with `us` in B
  // the pass would read the rows of us from the field it writes its result to
  const bool fuse_h = fuse && (fuse_bits & 2) != 0;
  float *const Ux = fuse ? Qx : Bx, *const Uy = fuse ? Qy : By;  // `us` of the Heun average
#pragma nounroll
  for (int half = 0; half < 2; ++half) {
    const float *sx = uni(half ? Bx : Ax), *sy = uni(half ? By : Ay);
    float *dx = uni(half ? Cx : Bx), *dy = uni(half ? Cy : By);
    const float *pin = uni(half ? p1 : p0);
    float *pout = uni(half ? p0 : p1);
    const bool fold = uni(P.bc_fold) != 0 && !(AFC_DBG_OF(P) & 0x80000000u);
    if (fuse && half) {
      if constexpr (C <= 2 || AFC_FUSE_T2) advdif_v3f<(C > 2 ? 2 : C), false, (C > 2)>(P, Bx, By, Ax, Ay, Cx, Cy, lds, p1, shift, Qx, Qy);
    } else if (fuse_h && heun_in) {
      if constexpr (C <= 2 || AFC_FUSE_T2) advdif_v3f<(C > 2 ? 2 : C), true, (C > 2)>(P, Cx, Cy, Ux, Uy, Bx, By, lds, p0, shift, Ax, Ay);
    } else if (adv3_ok)
      advdif_v3<(C > 2 ? 2 : C), C>(P, sx, sy, Ax, Ay, dx, dy, lds, fold);  // TAG = C: one instance per kernel instantiation
    else
      advdif_v2<C>(P, sx, sy, Ax, Ay, dx, dy, lds, fold);
    pf.mark(PF_ADVDIF);
    if (pf.stop == 1 + 8 * half) return;
    forcing_band(P, dx, dy, lds, dphi1, dphi2);
    pf.mark(PF_FORCING);
    if (fold)
      setbc_rows(P, dx, dy, lds);
    else
      setbc_vel(P, dx, dy, lds);
    pf.mark(PF_SETBC);
    if (pf.stop == 2 + 8 * half) return;
    float l;
#if AFC_DIAG
    if (pf.stop) pf.stop -= 8 * half;  // phases 3..6 of this half
#endif
    const int it = project_v2<C>(half != 0, P, e, dx, dy, pin, pout, &shift, Ux, Uy, Ax, Ay, sm, &l, pf,
                                 lds, first_after_action && half == 0, half == 0 ? fuse : (fuse_h && defer_out));
    if (pf.stop >= 3 && pf.stop <= 6) return;
#if AFC_DIAG
    if (pf.stop) pf.stop += 8 * half;
#endif
    if (half) {
      its[1] = it;
      lin[1] = l;
    } else {
      its[0] = it;
      lin[0] = l;
    }
  }
  const int it0 = its[0], it1 = its[1];
  l0 = lin[0];
  l1 = lin[1];
  *pshift = shift;
  if (threadIdx.x < 64) {
    float fx, fy;
    eval_force_wave(P, p0, shift, &fx, &fy);
  if (threadIdx.x == 0) {
    sh->fx = fx;
    sh->fy = fy;
    sh->it0 = it0;
    sh->it1 = it1;
    sh->r0 = l0;
    sh->r1 = l1;
  }
  }
  __syncthreads();
  pf.mark(PF_HEUN);
}

// can this env's step kernels leave the Heun tail of a sim-step to the next one (sim_step_v2's heun_in / defer_out)?
template <int C>
__device__ __forceinline__ bool heun_fusable(const StepParams &P) {
  const int adv3_ns = uni(P.adv_ns);
  return !AFC_DIAG && adv3_ns == kAdvNS && !(AFC_DBG_OF(P) & 1024) && (uni(P.fuse_tail) & 3) == 3 && uni(P.bc_fold) != 0 &&
         advf_ok_for(uni(P.n), uni(P.m), C);
}
// the tail a sim-step left out, on its own (a launch that ends with one pending: the state in memory must be complete)
__device__ __forceinline__ void heun_tail_only(const StepParams &P, size_t e, float shift, float *lds) {
  const size_t eo = e * (size_t)P.Ns;
  const bool q = (uni(P.fuse_tail) & 1) != 0;  // where `us` is: Q if the predictor's tail ran inside the corrector's advection, else B
  project_tail_v2(true, P, lds, uni(G(P.p) + eo), shift, uni(G(P.fx) + eo), uni(G(P.fy) + eo), uni((q ? G(P.qx) : G(P.usx)) + eo),
                  uni((q ? G(P.qy) : G(P.usy)) + eo), uni(G(P.ux) + eo), uni(G(P.uy) + eo), true);
}

#undef AFC_DBG_OF
}  // namespace AFC_NS
}  // namespace afc
