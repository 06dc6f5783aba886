// THE UP-LEG WITH THE ORDERED SWEEPS MARCHING DOWN THE ROWS (round 6; included by bdim_v2_body.h inside the namespace of
// the instantiation).
//
// MG.smooth's lexicographic Gauss-Seidel sweeps (MG.pde:82-89) leave exactly one freedom: cell (i, j) of sweep k needs
// the NEW values of (i-1, j) and (i, j-1) and the values sweep k-1 left in (i+1, j) and (i, j+1).  gs_march3 walks a strip
// of 16 rows along j, one cell per lane and step -- a chain of ~80 cycles per cell, 2720 steps on the 256x128 level, the
// other three waves parked or busy shuffling strips in and out of the rings.  Here ONE wave takes the whole level in one
// pass DOWN THE ROWS: lane 16 k + c owns the CB = m_/16 columns 1 + CB c .. CB (c + 1) of sweep k and at step T stands on
// row i = T - c - 3 k; within a step it updates its CB cells left to right.  Of the four neighbours of a cell
//   (i-1, j)  is the lane's own result of the previous step                                 -> register
//   (i, j-1)  is the cell it has just updated; for its first column lane c-1's last result
//             of the previous step (that lane is one row ahead)                             -> register / DPP row_shr:1
//   (i+1, j)  was left in the d ring by sweep k-1 (lane 16 (k-1) + c stood there two steps ago) -> LDS, fetched a step ahead
//   (i, j+1)  is what the lane fetched as "(i+1, j)" one step earlier; for its last column the first
//             column of lane c+1, which sweep k-1 left there two steps ago and sweep k has not reached yet -> LDS, a step ahead
// Every sweep stores its row in place, as the sequential loops do: same operands, same order, same bits.  A step is CB
// updates of 9 vector instructions, the chain through them five deep per cell (mul, add, add, sub, mul) instead of
// seven, no activity mask and no boundary test in the 200-odd steady steps, 5 LDS reads and 2 writes of 16 bytes per
// lane and step instead of 5 LDS instructions per CELL: 281 steps for the 256x128 level.
//
// Rows enter and leave the rings along DIAGONALS: cell block (row g, column block c) is on diagonal D = g + c; the march
// needs diagonal T + 2 at the start of step T and has finished with diagonal D (the cells around it included) after step
// D + 10.  The helper waves 1-3 work in passes of 64 four-column chunks = 2 / 4 / 8 diagonals: fill (prolongation
// increment, d0 = r inv: fill_load / fill_apply, the same functions as mg_up_smooth) and exit (d.setBC, x += d, r -= A d,
// norms: exit_load / exit_apply).  Because a column block only ever holds the rows of ITS window, the rings are 28 / 32
// rows deep whatever the level's size, and the rows that cross a body keep their real coefficients in an LDS table
// (rows sp_i0 .. sp_i1 + 1 of the column blocks that contain special cells, copied once per call).
// Hand-over: three counters in LDS -- diagonals filled (F), march steps done (M), diagonals exited (X) -- each advanced
// in order by whoever finishes the next piece; a word stored after data by the same wave is a valid flag (the LDS
// operations of a wave execute in order).  Every wait points to an earlier piece of the same total order: no cycle.
template <int CB>
struct TMC {
  static constexpr int PW = 16 * CB + 4;                         // floats per ring row (columns 0 .. 16 CB + 1)
  static constexpr int NR = (CB == 8) ? (AFC_DIAG ? 24 : 28) : (CB == 4 ? 32 : 64);  // ring rows (the diagnostic instantiation's LDS region is 2 KB smaller: tm_lds_floats' nr8)
  static constexpr int CPD = 4 * CB;                             // four-column chunks per row = helper lanes per diagonal
  static constexpr int ND = 64 / CPD;                            // diagonals per helper pass
  static constexpr int KX = 11 + (CB == 2 ? 1 : 0);              // diagonal D may leave once the march has done D + KX steps
  static constexpr int GD = (CB == 8) ? 8 : (CB == 4 ? 5 : 3);   // a helper's exit pass trails its fill pass by GD passes
  static constexpr int LG = 3;                                   // sweep k + 1 follows sweep k at three rows
  static constexpr int SPAN = 15 + 3 * LG;                       // rows between the first and the last lane
};
// special-cell table of a level: rows sp_i0 .. sp_i1 + 1, the column blocks that contain special columns (+ 4 columns)
__device__ __host__ __forceinline__ int tm_tab_w(int CB, int sp_j0, int sp_j1) {
  return (sp_j1 < sp_j0) ? 0 : CB * ((sp_j1 - 1) / CB - (sp_j0 - 1) / CB + 1) + 4;
}
__device__ __host__ __forceinline__ int tm_tab_floats(int CB, int sp_i0, int sp_i1, int sp_j0, int sp_j1) {
  return (sp_i1 < sp_i0) ? 0 : 3 * (sp_i1 - sp_i0 + 2) * tm_tab_w(CB, sp_j0, sp_j1);
}
__device__ __host__ __forceinline__ int tm_cb(int m_) { return m_ >> 4; }
__device__ __host__ __forceinline__ bool tm_shape_ok(int n, int m) {
  const int m_ = m - 2, n_ = n - 2;
  return (m_ == 128 || m_ == 64 || m_ == 32) && n_ >= 32 && (n_ & 7) == 0;
}
// nr8: ring rows of a 128-column level (TMC<8>::NR of the instantiation asked about: 28 shipped, 24 diagnostic)
__device__ __host__ __forceinline__ int tm_lds_floats(int m, int sp_i0, int sp_i1, int sp_j0, int sp_j1, int nr8 = TMC<8>::NR) {
  const int CB = tm_cb(m - 2);
  const int PW = 16 * CB + 4, NR = (CB == 8) ? nr8 : (CB == 4 ? TMC<4>::NR : TMC<2>::NR);
  return 8 + 2 * NR * PW + PW + tm_tab_floats(CB, sp_i0, sp_i1, sp_j0, sp_j1) + 2 * CB + 16 + 2 + 64 + 1 + 5;
}

#ifdef AFC_TM_PROF  // development (tools/upleg_bench.hip): time stamps of workgroup 0
__device__ long long g_tm_prof[64];
#define TM_STAMP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_tm_prof[k] = (long long)__builtin_readcyclecounter(); } while (0)
#define TM_ACC(k, v) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_tm_prof[k] += (v); } while (0)
#else
#define TM_STAMP(k) do { } while (0)
#define TM_ACC(k, v) do { } while (0)
#endif
// ---- the march: per-lane state, constants, one step ----
template <int CB>
struct TmState {
  float up[CB];        // the lane's results of the previous step (row i-1)
  float Rr[CB - 1];    // what sweep k-1 left in (i, j0+1 .. j0+CB-1): fetched as "(i+1, .)" a step earlier
  float Dn[CB], Xn, Rn[CB];  // fetched a step ahead: d(i+1, own columns), d(i, j0+CB), r(i, own columns)
  unsigned base, zb;   // LDS byte address of the lane's first column in slot 0 of the r ring / in the zero row
  unsigned oa, ob, oc; // slot offsets (bytes) of rows i, i+1, i+2
  int i;               // the row the lane stands on in the step to come
};
template <int CB>
struct TmConst {
  int n_, c, sp_i0, sp_n;
  float cu, it0, ivF0, ivL0;
  bool zb, sblk;
  unsigned a_tab, a_apk, TWB, tabA;
};
// slot offset of the next row: (o + PWB) mod RINGB without a compare (o < RINGB: the subtraction wraps to a huge value
// unless the sum has reached RINGB)
template <int CB>
__device__ __forceinline__ unsigned tm_next(unsigned o) {
  constexpr unsigned PWB = TMC<CB>::PW * 4u, RINGB = TMC<CB>::NR * PWB;
  const unsigned t = o + PWB, u = t - RINGB;
  return t < u ? t : u;
}
// issue the fetches for the step on row i (slot offsets oi = row i, oi1 = row i + 1): d(i+1, own), d(i, j0+CB), r(i, own)
template <int CB, bool GEN>
__device__ __forceinline__ void tm_fetch(TmState<CB> &S, const TmConst<CB> &Q, const int i, const unsigned oi,
                                          const unsigned oi1, const bool) {
  constexpr unsigned PWB = TMC<CB>::PW * 4u, RINGB = TMC<CB>::NR * PWB;
  // (lanes read what OTHER lanes of the wave stored a step earlier: program order must survive the optimiser -- for one
  // thread the addresses are provably different; in hardware the LDS operations of a wave execute in order)
  asm volatile("" ::: "memory");
  unsigned aD = S.base + oi1 + RINGB, aX = S.base + oi + RINGB + 4u * CB, aR = S.base + oi;
  if (GEN) {
    const bool okD = (unsigned)i < (unsigned)Q.n_;        // 1 <= i + 1 <= n_
    const bool okI = (unsigned)(i - 1) < (unsigned)Q.n_;  // 1 <= i <= n_
    aD = okD ? aD : S.zb;
    aX = okI ? aX : S.zb;
    aR = okI ? aR : S.zb;
  }
  if constexpr (CB >= 4) {
#pragma unroll
    for (int q = 0; q < CB / 4; ++q) {
      const f4a d4 = *(const lds_f4 *)(uintptr_t)(aD + 16u * q), r4 = *(const lds_f4 *)(uintptr_t)(aR + 16u * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) S.Dn[4 * q + e] = d4[e], S.Rn[4 * q + e] = r4[e];
    }
  } else {
    const f2 d2 = *(const lds_f2a *)(uintptr_t)aD, r2 = *(const lds_f2a *)(uintptr_t)aR;
    S.Dn[0] = d2.x, S.Dn[1] = d2.y, S.Rn[0] = r2.x, S.Rn[1] = r2.y;
  }
  S.Xn = *(const lds_f *)(uintptr_t)aX;
}
// One step: the lane's CB cells of row S.i, left to right, from (up, Rr, Dn, Xn, Rn); the result goes to the d ring in
// place and into `up`; the fetches for the next step are issued first (into the same buffers: their old contents are
// copied out -- the compiler renames).  GEN: lanes may stand outside the grid or on its first / last row (activity,
// boundary classes of the inverse diagonal); SP: coefficients through LDS (special-cell table / analytic packet).
template <int CB, bool GEN, bool SP, bool ZB>
__device__ __forceinline__ void tm_step(TmState<CB> &S, const TmConst<CB> &Q) {
  constexpr unsigned PWB = TMC<CB>::PW * 4u, RINGB = TMC<CB>::NR * PWB;
  const int i = S.i;
  float Dd[CB], Rv[CB];
  const float Xx = S.Xn;
#pragma unroll
  for (int e = 0; e < CB; ++e) Dd[e] = S.Dn[e], Rv[e] = S.Rn[e];
  const unsigned oa = S.oa;
  // next step: row i + 1 (slots ob, oc).  Issued FIRST, and kept there: left to itself the scheduler sinks these reads to the
  // end of the step (nothing needs them before the next one) and every step then opens with an exposed LDS round trip
#ifdef AFC_TM_EXP_NOFETCH
  if (Q.n_ < 0)
#endif
  tm_fetch<CB, GEN>(S, Q, i + 1, S.ob, S.oc, true);
  __builtin_amdgcn_sched_barrier(0);
  float lw[CB], le[CB], ls[CB + 1], iv[CB];
  bool act = true;
  if (GEN) {
    act = (unsigned)(i - 1) < (unsigned)Q.n_;
    bool use_tab = false;
    if (SP) {  // coefficients through LDS: the table for lanes on special cells, the analytic packet otherwise
      use_tab = Q.sblk && (unsigned)(i - Q.sp_i0) <= (unsigned)Q.sp_n;
      const unsigned ro = Q.a_tab + (unsigned)(i - Q.sp_i0) * Q.TWB;
      const unsigned aLx = use_tab ? ro : Q.a_apk, aLxn = use_tab ? ro + Q.TWB : Q.a_apk, aLy = use_tab ? ro + Q.tabA : Q.a_apk,
                     aIv = use_tab ? ro + 2u * Q.tabA : Q.a_apk + 4u * (CB + 4);
      if constexpr (CB >= 4) {
#pragma unroll
        for (int q = 0; q < CB / 4; ++q) {
          const f4a a4 = *(const lds_f4 *)(uintptr_t)(aLx + 16u * q), b4 = *(const lds_f4 *)(uintptr_t)(aLxn + 16u * q),
                    c4 = *(const lds_f4 *)(uintptr_t)(aLy + 16u * q), d4 = *(const lds_f4 *)(uintptr_t)(aIv + 16u * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) lw[4 * q + e] = a4[e], le[4 * q + e] = b4[e], ls[4 * q + e] = c4[e], iv[4 * q + e] = d4[e];
        }
      } else {
        const f2 a2 = *(const lds_f2a *)(uintptr_t)aLx, b2 = *(const lds_f2a *)(uintptr_t)aLxn,
                 c2 = *(const lds_f2a *)(uintptr_t)aLy, d2 = *(const lds_f2a *)(uintptr_t)aIv;
        lw[0] = a2.x, lw[1] = a2.y, le[0] = b2.x, le[1] = b2.y, ls[0] = c2.x, ls[1] = c2.y, iv[0] = d2.x, iv[1] = d2.y;
      }
      ls[CB] = *(const lds_f *)(uintptr_t)(aLy + 4u * CB);
    } else {
#pragma unroll
      for (int e = 0; e < CB; ++e) lw[e] = le[e] = ls[e] = Q.cu, iv[e] = Q.it0;
      ls[CB] = Q.cu;
    }
    if (ZB && !use_tab) {  // boundary classes of the inverse diagonal (coarse levels): only inv knows the boundary
      const unsigned at = Q.a_apk + 4u * (2 * CB + 4) + ((i == 1) ? 12u : ((i == Q.n_) ? 24u : 0u));  // invtab row of the row class
      const float vm = *(const lds_f *)(uintptr_t)at;
      const float vf = *(const lds_f *)(uintptr_t)(at + 4u), vl = *(const lds_f *)(uintptr_t)(at + 8u);
#pragma unroll
      for (int e = 0; e < CB; ++e) iv[e] = vm;
      iv[0] = (Q.c == 0) ? vf : vm;
      iv[CB - 1] = (Q.c == 15) ? vl : vm;
    }
  }
  // (i, j0-1): lane c-1's last result of the previous step; lane c == 0 stands at the wall (ghost d = 0)
#ifdef AFC_TM_EXP_NODPP
  float L = S.up[CB - 1];
#else
  float L = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, S.up[CB - 1]), 0x111, 0xF, 0xF, false));
#endif
  float nw[CB];
#pragma unroll
  for (int e = 0; e < CB; ++e) {
    const float Rg = (e < CB - 1) ? S.Rr[e] : Xx;
    float t;
    if (GEN) {
      t = S.up[e] * lw[e] + Dd[e] * le[e];  // MG.pde:84-88, in this order
      t = t + L * ls[e];
      t = t + Rg * ls[e + 1];
      t = t - Rv[e];
      nw[e] = -t * iv[e];
    } else {
      // (a level without zeroed boundary faces -- the finest -- has the same inverse diagonal in every cell)
      const float ive = !ZB ? Q.it0 : ((e == 0) ? Q.ivF0 : ((e == CB - 1) ? Q.ivL0 : Q.it0));
      t = S.up[e] * Q.cu + Dd[e] * Q.cu;
      t = t + L * Q.cu;
      t = t + Rg * Q.cu;
      t = t - Rv[e];
      nw[e] = -t * ive;
    }
    L = nw[e];
  }
  const unsigned aW = S.base + oa + RINGB;
#ifdef AFC_TM_EXP_NOSTORE
  if (act && Q.n_ < 0) {
#else
  if (act) {
#endif
    if constexpr (CB >= 4) {
#pragma unroll
      for (int q = 0; q < CB / 4; ++q) {
        f4a o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = nw[4 * q + e];
        *(lds_f4 *)(uintptr_t)(aW + 16u * q) = o4;
      }
    } else {
      f2a8 o2;
      o2.x = nw[0], o2.y = nw[1];
      *(lds_f2a *)(uintptr_t)aW = o2;
    }
  }
#pragma unroll
  for (int e = 0; e < CB; ++e) {
    S.up[e] = (!GEN || act) ? nw[e] : S.up[e];
    if (e < CB - 1) S.Rr[e] = Dd[e + 1];
  }
  S.oa = S.ob;
  S.ob = S.oc;
  S.oc = tm_next<CB>(S.oc);
  S.i = i + 1;
  asm volatile("" ::: "memory");
}

#ifndef AFC_TM_SLEEP
#define AFC_TM_SLEEP 1
#endif
// A wait on one of the counters (wave-uniform).  The spin is BOUNDED (kTmSpinCap looks at the counter, >= 0.2 s; no wait of
// the protocol lasts longer than a few march steps): a protocol error must not hang the GPU -- a wait that runs out sets the
// up-leg's `broken` word, and the up-leg then poisons its result (NaN into x and the norms: the env comes out `diverged`,
// loudly).  The FORM of the loop matters far beyond its own cycles -- these functions' bodies decide what the compiler may
// assume around their call sites in the step kernel.  Same memory, per launch against the strip pipeline, 1 env / 256 envs:
//   this form (counter dead behind the loop, expiry found by looking once more)   -14.0 % /  -8.7 %
//   the same without the second look (expiry passes silently)                      -14.8 % /  -9.4 %
//   `if (it == cap) broken = 1` behind the loop                                    -10.9 % /  -6.4 %
//   `if (it == cap) __builtin_trap()`                                               -7.8 % /  -3.7 %
//   unbounded `while (word < need) s_sleep`                                         -6.0 % /  +1.2 %
// (s_sleep 0 instead of 1: no difference.)
constexpr int kTmSpinCap = 1 << 22;
__device__ __forceinline__ void tm_wait(const lds_i *w, int need, lds_i *broken) {
  // (the loop in exactly this form -- its counter dead behind it; the expiry is found by looking at the word once more)
  for (int it = 0; it < kTmSpinCap && uni(*(volatile const lds_i *)w) < need; ++it) __builtin_amdgcn_s_sleep(AFC_TM_SLEEP);
  if (uni(*(volatile const lds_i *)w) < need) *(volatile lds_i *)broken = 1;
  asm volatile("" ::: "memory");
}
// What the march needs from its caller (everything wave-uniform but `lane`): passed by value to a function of its own.
struct TmArgs {
  int n_, Dmax, has_sp, sp_i0, sp_i1, sb0, sb1, TW, TR, tc0;
  float cu, it0, it1, it2;
  unsigned a_rr, a_z, a_tab, a_apk, a_cF, a_cM, a_cB;
};
// The march as a function of its own (one call per level and up-leg): inlined into the step kernels its ~90 live
// registers compete with everything that is live across the call site, and the allocator answers with scratch reloads
// INSIDE the step loop -- each a vector-memory round trip on the one wave whose latency the whole workgroup waits for.
// As a callee it gets a clean allocation (capped at the kernels' 128 registers) and pays the calling convention's
// saves once per call.
template <int CB, bool ZB>
__device__ __noinline__ void tm_march(const TmArgs A_) {
  typedef TMC<CB> K;
  constexpr int PW = K::PW, NR = K::NR, LG = K::LG;
  constexpr unsigned PWB = PW * 4u;
  const int lane = (int)lane_fresh(), k = lane >> 4, c = lane & 15;
  const int n_ = uni(A_.n_), Dmax = uni(A_.Dmax);
  const bool has_sp = uni(A_.has_sp) != 0;
  const int sp_i0 = uni(A_.sp_i0), sp_i1 = uni(A_.sp_i1), sb0 = uni(A_.sb0), sb1 = uni(A_.sb1);
  const lds_i *cF = (const lds_i *)(uintptr_t)uni((int)A_.a_cF);
  lds_i *cB = (lds_i *)(uintptr_t)uni((int)A_.a_cB);
  TmState<CB> S;
  {
    const unsigned colb = 4u * (unsigned)(1 + CB * c);  // byte offset of the lane's first column in a ring row
    S.base = (unsigned)uni((int)A_.a_rr) + colb;
    S.zb = (unsigned)uni((int)A_.a_z) + colb;
    const int i0 = 0 - c - LG * k;            // the lane's row at step 0
    int s1 = (i0 + 1) % NR, s2 = (i0 + 2) % NR, s0 = i0 % NR;
    s0 = s0 < 0 ? s0 + NR : s0, s1 = s1 < 0 ? s1 + NR : s1, s2 = s2 < 0 ? s2 + NR : s2;
    S.oa = (unsigned)s0 * PWB, S.ob = (unsigned)s1 * PWB, S.oc = (unsigned)s2 * PWB;
    S.i = i0;
#pragma unroll
    for (int e = 0; e < CB; ++e) S.up[e] = 0.f, S.Dn[e] = 0.f, S.Rn[e] = 0.f;
#pragma unroll
    for (int e = 0; e < CB - 1; ++e) S.Rr[e] = 0.f;
    S.Xn = 0.f;
  }
  TmConst<CB> Q;
  Q.n_ = n_;
  Q.cu = uni(A_.cu);
  Q.zb = ZB;
  Q.it0 = uni(A_.it0);
  Q.ivF0 = (ZB && c == 0) ? uni(A_.it1) : Q.it0;
  Q.ivL0 = (ZB && c == 15) ? uni(A_.it2) : Q.it0;
  Q.c = c;
  Q.sp_i0 = sp_i0, Q.sp_n = sp_i1 - sp_i0;
  Q.sblk = has_sp && c >= sb0 && c <= sb1;
  Q.a_tab = (unsigned)uni((int)A_.a_tab) + 4u * (unsigned)(1 + CB * c - uni(A_.tc0));
  Q.a_apk = (unsigned)uni((int)A_.a_apk);
  Q.TWB = 4u * (unsigned)uni(A_.TW);
  Q.tabA = (unsigned)(uni(A_.TR) * uni(A_.TW)) * 4u;
  // step kinds: fast = every lane on an interior row (2 <= i, i + 1 <= n_), no lane of a special block on a special row
  const int Tend = n_ + K::SPAN;           // last step: the last lane on row n_
  const int Tf0 = K::SPAN + 2;             // first fast step: the last lane on row 2
  const int Tf1 = n_ - 1;                  // last fast step: the first lane on row n_ - 1
  // special phase: some lane of a special block stands on a row sp_i0 .. sp_i1
  const int Ts0 = has_sp ? sp_i0 + sb0 : (1 << 30), Ts1 = has_sp ? sp_i1 + sb1 + 3 * LG : -1;
  const unsigned a_m = (unsigned)uni((int)A_.a_cM) + 4u * (unsigned)lane;
  int fv = 0;  // the fill counter as last seen
  TM_STAMP(2);
  // one iteration = two steps: wait for the rows, two steps of one kind, publish.  A loop of its own per kind (a single
  // loop that picks its body per iteration joins three register assignments with ~40 copies per iteration).
#define TM_PAIR(GEN_, SP_)                                                                                     \
  {                                                                                                            \
    {                                                                                                          \
      int need = T + 3;                                                                                        \
      need = need > Dmax ? Dmax : need;                                                                        \
      if (uni(fv) < need) {                                                                                    \
        TM_WAIT_PROF_BEGIN                                                                                     \
        tm_wait(cF, need, cB);                                                                                 \
        TM_WAIT_PROF_END                                                                                       \
      }                                                                                                        \
    }                                                                                                          \
    TM_EXP_FV                                                                                                  \
    tm_step<CB, GEN_, SP_, ZB>(S, Q);                                                                          \
    tm_step<CB, GEN_, SP_, ZB>(S, Q);                                                                          \
    /* steps 0 .. T + 1 are stored (LDS executes in order); one word per lane, no lane mask: readers look at lane 0's */ \
    TM_EXP_POST                                                                                                \
    T += 2;                                                                                                    \
  }
#ifdef AFC_TM_EXP_NOFLAG
#define TM_EXP_FV fv = 1 << 20;
#define TM_EXP_POST
#else
#define TM_EXP_FV fv = *(volatile const lds_i *)cF; /* for the next iteration's test: on its way while this one computes */
#define TM_EXP_POST *(volatile lds_i *)(uintptr_t)a_m = T + 2;
#endif
#ifdef AFC_TM_PROF
#define TM_WAIT_PROF_BEGIN const long long w0_ = (long long)__builtin_readcyclecounter();
#define TM_WAIT_PROF_END TM_ACC(8, (long long)__builtin_readcyclecounter() - w0_); TM_ACC(9, 1);
#else
#define TM_WAIT_PROF_BEGIN
#define TM_WAIT_PROF_END
#endif
  tm_wait(cF, 3 > Dmax ? Dmax : 3, cB);
  tm_fetch<CB, true>(S, Q, S.i, S.oa, S.ob, true);  // prime: what step 0 works on
  int T = 0;
  // (T + 2 in the fast test: the second fetch of an iteration is for step T + 2)
  auto is_sp = [&](int t) { return (t + 1 >= Ts0) && (t <= Ts1); };
  auto is_fast = [&](int t) { return (t >= Tf0) && (t + 2 <= Tf1) && !is_sp(t); };
  while (T <= Tend) {
    if (is_fast(T)) {
      do TM_PAIR(false, false) while (T <= Tend && is_fast(T));
    } else if (is_sp(T)) {
      do TM_PAIR(true, true) while (T <= Tend && is_sp(T));
    } else {
      do TM_PAIR(true, false) while (T <= Tend && !is_fast(T) && !is_sp(T));
    }
  }
#undef TM_PAIR
#undef TM_EXP_FV
#undef TM_EXP_POST
#undef TM_WAIT_PROF_BEGIN
#undef TM_WAIT_PROF_END
  TM_STAMP(3);
}

// The whole up-leg as a function of its own in the shipped kernels (one call per level and up-leg): inlined, its helpers'
// two register sets and the call of the march push the allocation of the ~200 KB step kernel over the edge -- scratch
// reloads inside the row loops of every other phase (+38 % per launch at 1024 envs, measured).  The LDS region comes as an
// address and becomes an explicit LDS pointer again inside, so that the accesses stay ds_* instructions.
#ifndef AFC_TR_ATTR
#define AFC_TR_ATTR __noinline__
#endif
struct TrOut {
  float rr, linf, xs;
};
template <int CB>
__device__ AFC_TR_ATTR TrOut mg_up_tr(const bool HASC_, const Lv F_, const Lv Cc_, const bool need_r_, const unsigned lds_a,
                                      const int rmode_) {
  const bool HASC = uni((int)HASC_) != 0, need_r = uni((int)need_r_) != 0;
  const int rmode = uni(rmode_);
  const Lv F = uni(F_), Cc = uni(Cc_);
  float *lds = (float *)(lds_f *)(uintptr_t)(unsigned)uni((int)lds_a);
  typedef TMC<CB> K;
  constexpr int PW = K::PW, NR = K::NR, LG = K::LG;
  constexpr unsigned PWB = PW * 4u, RINGB = NR * PWB;
  const int n = F.n, m = F.m, n_ = n - 2, m_ = m - 2;
  const int tid = threadIdx.x;
  const LevelDesc &D = *F.desc;
  const DescU du = desc_u(D);
  const int sj0 = uni(D.sp_j0), sj1 = uni(D.sp_j1);
  const bool has_sp = du.sp_i1 >= du.sp_i0;
  const int sb0 = has_sp ? (sj0 - 1) / CB : 0, sb1 = has_sp ? (sj1 - 1) / CB : -1;
  const int TW = has_sp ? CB * (sb1 - sb0 + 1) + 4 : 0, TR = has_sp ? du.sp_i1 - du.sp_i0 + 2 : 0;
  const int tc0 = 1 + CB * sb0;  // grid column of table column 0
  // LDS (floats from a 16-byte boundary): [3 | r ring | d ring | zero row | table lx, ly, inv | analytic packet | counters]
  float *base = lds + (((16u - (lds_addr(lds) & 15u)) & 15u) >> 2);
  float *rr = base + 3, *dr = rr + NR * PW, *zrow = dr + NR * PW;  // ring column 1 on a 16-byte boundary
  float *tab = zrow + PW + 1;                                        // (zrow + PW + 1 = base + 4 + ...: 16-byte aligned)
  float *apk = tab + 3 * TR * TW;                                    // cu x (CB + 4) | iv0 x CB | invtab[9]
  lds_i *cnt = (lds_i *)(lds_f *)(apk + 2 * CB + 16);                 // F, X | M: one word per lane of the marching wave
  const int mc = Cc.m;
  const int CPR = m_ >> 2;
  double rr_part = 0, xs_part = 0;
  float linf = 0.f;
  UpGeom U{&F, &Cc, &D, n, m, n_, m_, mc, PW, CPR, rr, dr, du};
  U.NR = NR;
  U.al16 = true;
  U.store_r = (rmode != 1);
  U.update_x = (rmode != 2);
  const int Dmax = n_ + 15;  // diagonals 1 .. Dmax
  TM_STAMP(0);
  __syncthreads();
  for (int q = tid; q < PW; q += kThreads) zrow[q] = 0.f;
  if (tid < 2 * NR) dr[(tid >> 1) * PW + ((tid & 1) ? m_ + 1 : 0)] = 0.f;  // ghost columns of d read as zero (nobody writes them again)
#ifdef AFC_TM_EXP_SOLO
  if (tid < 2 + 64 + 1) cnt[tid] = (tid == 0) ? (1 << 20) : 0;  // everything "filled": the march alone (timing experiment, wrong data)
#else
  if (tid < 2 + 64 + 1) cnt[tid] = 0;
#endif
  if (tid < 2 * CB + 4) apk[tid] = (tid < CB + 4) ? du.cu : du.iv0;
  if (tid >= 64 && tid < 64 + 9) apk[2 * CB + 4 + tid - 64] = D.invtab[tid - 64];
  if (has_sp) {  // the level's special-cell table: rows sp_i0 .. sp_i1 + 1, columns tc0 .. tc0 + TW - 1 of lx, ly, inv
    const int per = TR * TW;
    for (int q = tid; q < 3 * per; q += kThreads) {
      const int a = q / per, rem = q - a * per, r_ = rem / TW, c_ = rem - r_ * TW;
      int col = tc0 + c_;
      col = col > m - 1 ? m - 1 : col;
      const float *src = (a == 0) ? F.lx : (a == 1 ? F.ly : F.inv);
      tab[q] = src[(du.sp_i0 + r_) * m + col];
    }
  }
  lds_barrier();
  TM_STAMP(1);
  lds_i *cF = cnt, *cX = cnt + 1, *cM = cnt + 2, *cB = cnt + 66;  // F, X | M: one word per lane of the march | broken
  if (tid < 64) {
    // ------------------------------------------------ the march ------------------------------------------------
    TmArgs A_;
    A_.n_ = n_, A_.Dmax = Dmax, A_.has_sp = has_sp ? 1 : 0, A_.sp_i0 = du.sp_i0, A_.sp_i1 = du.sp_i1, A_.sb0 = sb0, A_.sb1 = sb1;
    A_.TW = TW, A_.TR = TR, A_.tc0 = tc0;
    A_.cu = du.cu, A_.it0 = uni(D.invtab[0]), A_.it1 = uni(D.invtab[1]), A_.it2 = uni(D.invtab[2]);
    A_.a_rr = lds_addr(rr), A_.a_z = lds_addr(zrow), A_.a_tab = lds_addr(tab), A_.a_apk = lds_addr(apk);
    A_.a_cF = lds_addr((const float *)(lds_f *)cF), A_.a_cM = lds_addr((const float *)(lds_f *)cM);
    A_.a_cB = lds_addr((const float *)(lds_f *)cB);
    // (constant priorities here: this up-leg is picked for fewer than 2.5 workgroups per CU, where the base priorities do not
    // rotate (prio_rotate) -- and the save / restore form moved afc::lat's launch by +0.5 %: see the note on the spin loops)
    __builtin_amdgcn_s_setprio(3);
    if (du.zb)
      tm_march<CB, true>(A_);
    else
      tm_march<CB, false>(A_);
    __builtin_amdgcn_s_setprio(0);
  }
#ifdef AFC_TM_EXP_SOLO
  else if (n_ < 0) {
#else
  else {
#endif
    // ------------------------------------------------ the helpers ------------------------------------------------
    constexpr int CPD = K::CPD, ND = K::ND, KX = K::KX, GD = K::GD;
    const int hw = (tid >> 6) - 1, l = tid & 63;
    const int dd = l / CPD, a = l - dd * CPD;        // diagonal of the pass, four-column chunk of the row
    const int clo = (4 * a) / CB;                    // column block of the chunk's first column
    const int npass = (Dmax + ND - 1) / ND;
    // Roles: wave 1 fills (pass after pass, F is its own counter), waves 2 and 3 exit alternate passes (X advances in
    // order: a post waits for its predecessor's).  Who waits for whom:
    //   fill(p)   X >= last diagonal of p - NR + 1: the slots of the rows NR further up -- their exits and those of the
    //             cells around them (one diagonal further) are done;
    //   march(T)  F >= T + 3;
    //   exit(p)   M >= last diagonal of p + KX.
    // exit(p) thus needs the fills up to diagonal ND (p + 1) + KX + 2, which need exits up to that minus NR - 1: earlier
    // passes as long as NR > 2 ND + KX + 2 -- every wait points backwards in one total order, and the fill frontier may
    // run NR - KX - 2 ND - 2 diagonals ahead of the march.
    // The global loads of a pass are issued two passes of the wave ahead (two register sets): under load a round trip
    // takes 2-3 us.
    static_assert(NR > 2 * ND + KX + 2 + 4, "ring too short");
    (void)GD;
    TmH H;
    H.r = F.r, H.xc = Cc.x, H.x = F.x, H.rw = F.r;
    H.n = n, H.n_ = n_, H.m = m, H.mc = mc;
    H.cu = du.cu, H.iv0 = du.iv0, H.dg = -(du.cu + du.cu + du.cu + du.cu);
    H.a_rr = lds_addr(rr), H.a_inv = lds_addr(apk) + 4u * (2 * CB + 4);
    H.pwb = PWB, H.dofs = RINGB;
    H.hasc = HASC, H.need_r = need_r, H.store_r = U.store_r, H.update_x = U.update_x;
    H.a = a, H.c0 = 1 + dd - clo;
    H.first = (a == 0), H.last = (a == CPD - 1);
    const bool zb = du.zb != 0;
    if (hw == 0) {
      auto issue = [&](const int pq, TmFillIn &fin) {
        const int gf = ND * pq + H.c0;  // row of this lane's chunk (diagonal - column block)
        if ((pq < npass) && gf >= 1 && gf <= n_) fin = tm_fill_load(H, gf);
      };
      auto work = [&](const int pq, const TmFillIn &fin) {
        const int gf = ND * pq + H.c0;
        const bool fi_on = gf >= 1 && gf <= n_;
        const bool sp_fill = (ND * pq + ND + 1 >= du.sp_i0) && (ND * pq + 1 - 16 <= du.sp_i1);  // rows of a pass: first diagonal - 15 .. last diagonal
        const int needx = ND * pq + ND - NR + 1;
        if (needx > 0) tm_wait(cX, needx > Dmax ? Dmax : needx, cB);
#ifdef AFC_TM_EXP_NOAPPLY
        if (fi_on && n_ < 0) {
#else
        if (fi_on) {
#endif
          if (sp_fill) {  // a pass that crosses a body: the generic form with the real coefficient arrays
            FillIn gi;
            gi.r = fin.r;
#pragma unroll
            for (int e = 0; e < 4; ++e) gi.a[e] = fin.va.v[e];
            gi.w1 = fin.w1, gi.w2 = fin.w2, gi.e1 = fin.e1, gi.e2 = fin.e2;
            fill_apply(U, HASC, gf, a, gi, true);
          } else if (zb)
            tm_fill_apply<NR, true>(H, gf, fin);
          else
            tm_fill_apply<NR, false>(H, gf, fin);
        }
        const int v = ND * pq + ND;
        pipe_post_word(cF, v > Dmax ? Dmax : v);
      };
      TmFillIn f0, f1;
      issue(0, f0);
      issue(1, f1);
      for (int pq = 0; pq < npass; pq += 2) {
        work(pq, f0);
        issue(pq + 2, f0);
        if (pq + 1 < npass) {
          work(pq + 1, f1);
          issue(pq + 3, f1);
        }
      }
    } else {
      auto issue = [&](const int pe, TmExitIn &ein) {
        const int ge = ND * pe + H.c0;
        if ((pe < npass) && ge >= 1 && ge <= n_) ein = tm_exit_load(H, ge);
      };
      auto work = [&](const int pe, const TmExitIn &ein) {
        const int ge = ND * pe + H.c0;
        const bool ex_on = ge >= 1 && ge <= n_;
        const bool sp_exit = (ND * pe + ND + 1 >= du.sp_i0) && (ND * pe + 1 - 16 <= du.sp_i1);
        int needm = ND * pe + ND + KX;
        needm = needm > n_ + K::SPAN + 2 ? n_ + K::SPAN + 2 : needm;
        tm_wait(cM, needm, cB);
#ifdef AFC_TM_EXP_NOAPPLY
        if (ex_on && n_ < 0) {
#else
        if (ex_on) {
#endif
          if (sp_exit) {
            ExitIn gi;
            gi.x = ein.x, gi.a1 = ein.a1, gi.a2 = ein.a2, gi.gl = ein.gl, gi.gh = ein.gh;
            exit_apply(U, HASC, need_r, ge, a, gi, true, rr_part, xs_part, linf);
          } else if (zb)
            tm_exit_apply<NR, true>(H, ge, ein, rr_part, xs_part, linf);
          else
            tm_exit_apply<NR, false>(H, ge, ein, rr_part, xs_part, linf);
        }
        tm_wait(cX, ND * pe, cB);  // in order
        const int v = ND * pe + ND;
        pipe_post_word(cX, v > Dmax ? Dmax : v);
      };
      TmExitIn e0, e1;
      const int w = hw - 1;
      issue(w, e0);
      issue(w + 2, e1);
      for (int pe = w; pe < npass; pe += 4) {
        work(pe, e0);
        issue(pe + 4, e0);
        if (pe + 2 < npass) {
          work(pe + 2, e1);
          issue(pe + 6, e1);
        }
      }
    }
  }
  if (tid == 64) TM_STAMP(5);
  if (tid == 128) TM_STAMP(6);
  __syncthreads();  // phase boundary: the x (and r) this level stored must be visible to what follows
  TM_STAMP(4);
  const bool broken = uni(*(volatile const lds_i *)cB) != 0;  // a wait of the hand-over protocol ran out: never seen; must not pass silently
  TrOut o;
  o.rr = o.linf = o.xs = 0.f;
  if (need_r && rmode != 2) {
    block_reduce3(rr_part, xs_part, linf, lds);
    o.rr = uni((float)rr_part);
    o.linf = uni(linf);
    o.xs = uni((float)xs_part);
  }
  if (broken) {
    const float qnan = __builtin_nanf("");
    o.rr = o.linf = o.xs = qnan;
    if (tid == 0) F.x[m + 1] = qnan;  // the level's first interior cell: the NaN reaches p and the force, the env reports `diverged`
  }
  return o;
}
