// bdim_v2_multiwave.h -- the all-waves hyperplane smoother (gs_multiwave / mg_up_smooth_mw): a tested alternative to the
// strip march, selectable with dbg bit 2 in the DIAGNOSTIC instantiation only (docs/HISTORY.md).
// Included by bdim_v2_body.h inside namespace afc::diag; not compiled into the shipped or the perf-mode kernels.

// =============================================================================================
// Multi-wave hyperplane smoother: MG.smooth(4) (MG.pde:79-92) with ALL FOUR WAVES of the workgroup.
//
// Wave k runs sweep k; lane l (1..H) of wave k owns row i = I0 - k - 1 + l of a strip of H rows (the strip of
// sweep k is shifted up by k rows, so sweep k never needs a row its predecessor has not finished), lane 0 re-plays
// row I0 - k - 1, which the previous strip finished, so that every real lane finds d[i-1][j] in lane-1.
// At global step tau the lane is at column j = tau - l - S*k + 1.  The four neighbours of cell (i,j) in sweep k:
//   d[i-1][j] (sweep k)   computed one step earlier by lane l-1  -> DPP wave_shr:1
//   d[i][j-1] (sweep k)   computed one step earlier by this lane -> register
//   d[i+1][j] (sweep k-1) | both were written to the LDS array D by wave k-1 exactly S steps earlier (k == 0: d0 =
//   d[i][j+1] (sweep k-1) | r*inv left there by the fill) and are overwritten by this very sweep one step LATER
// which is the order the sequential loops of the reference see, so every cell gets the reference's bits.
// Steps are grouped in chunks of 4 with ONE LDS barrier per chunk: a chunk reads its operands from LDS one chunk
// ahead, computes 4 steps in registers and writes its 4 results; S = 8 guarantees that what a chunk prefetches was
// written at least one barrier earlier (4q+e+4-S <= 4q-1) and that nothing it still needs has been overwritten.
// D and R (r after the prolongation increment) are rings of NR = H + 6 full-width rows; r is read-only here.
// ---------------------------------------------------------------------------------------------
// dst = (lane is a re-play lane) ? b : a, as one v_cndmask on a constant lane mask (kept opaque: written as a C
// select the compiler predicates the whole update on it and pays an exec-mask branch per step)
__device__ __forceinline__ float sel_mask(float a, float b, unsigned long long mk) {
  float r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mk));
  return r;
}

// PER = sweeps per wave.  PER 1: wave k = sweep k, up to 62 rows per strip.  PER 2: waves 0 and 1 carry two sweeps
// each (lanes 0-31 / 32-63), up to 30 rows per strip -- the form for the finest level, whose strips are LDS-bound
// to ~28 rows: half the instruction streams for the same step count.  Waves wait for each other through progress
// words in LDS (a wave publishes the number of the chunk it has finished; the next sweep's wave checks it before it
// prefetches), not through workgroup barriers: LDS operations of one wave execute in order, so a consumer that sees
// the progress word also sees the data written before it.
// Not inlined on purpose: every instantiation gets its own register allocation instead of inflating the live ranges
// of the phases around it.
template <bool FAST, bool ZB, int PER>
__device__ __noinline__ void gs_multiwave(const Lv &L, const float *Rg, float *Dg, float *dummyg, int NR, int I0,
                                          int Lmax, int gbase) {
  const int wave = threadIdx.x >> 6;
  if (PER == 2 && wave >= 2) return;
  const lds_f *R = (const lds_f *)Rg;
  lds_f *D = (lds_f *)Dg;
  lds_f *dummy = (lds_f *)dummyg;
  volatile lds_i *prog = (volatile lds_i *)(dummy + 64);  // [4] progress words
  const int m = L.m, n_ = L.n - 2, m_ = L.m - 2;
  const int lane64 = threadIdx.x & 63;
  const int lane = (PER == 2) ? (lane64 & 31) : lane64;
  const int k = (PER == 2) ? (2 * wave + (lane64 >> 5)) : wave;
  const unsigned long long replay = (PER == 2) ? 0x100000001ull : 1ull;
  const int i = I0 - k - 1 + lane;
  const bool rowin = (i >= 1) && (i <= n_);
  const bool is0 = (lane == 0);
  const bool real = rowin && !is0 && (lane <= Lmax);
  const unsigned m_on = (real || (rowin && is0)) ? (unsigned)m_ : 0u;  // columns 1..m_on produce a value
  const unsigned m_wr = real ? (unsigned)m_ : 0u;                     // ... and are stored
  const int ic = i < 0 ? 0 : (i > n_ ? n_ : i);
  const int so = (ic % NR) * m;
  const int se = is0 ? so : ((ic + 1) % NR) * m;  // re-play lanes fetch their own (finished) row through the east stream
  const int jo = 1 - lane - kMwS * k;
  const LevelDesc &Dsc = *L.desc;
  float iv0 = 0.f, iv1 = 0.f, iv2 = 0.f;
  if (FAST) {
    const int rc = (i == 1) ? 1 : ((i == n_) ? 2 : 0);
    iv0 = Dsc.invtab[3 * rc + 0];
    iv1 = Dsc.invtab[3 * rc + 1];
    iv2 = Dsc.invtab[3 * rc + 2];
  }
  const float cu = uni(Dsc.cu);
  const int T = (m_ + Lmax + 3 * kMwS + 7) & ~7;  // two chunks per loop iteration
  const lds_f *pR = R + so + jo, *pE = D + se + jo, *pN = D + so + jo + 1;
  lds_f *pW = D + so + jo;
  lds_f *dmy = dummy + lane64;
  float aR[4], aE[4], aN[4], bR[4], bE[4], bN[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    aR[e] = pR[e];
    aE[e] = pE[e];
    aN[e] = pN[e];
  }
  // coefficients of the strips that touch a body: four columns per 16-byte load, one chunk ahead.  The east face
  // lx[i+1][j] is the west face of the row below, i.e. what lane+1 (one column behind) holds: no load for it.
  F4 a_lw, a_ls, a_iv, b_lw, b_ls, b_iv;
  float a_ln4 = 0.f, b_ln4 = 0.f;
  const int icf = i < 0 ? 0 : (i > n_ + 1 ? n_ + 1 : i);  // every lane loads its own row (also the one below the strip)
  const int cbase = icf * m + jo;
#define MW_COEF(TAU, lw_, ls_, iv_, ln4_) \
  {                                       \
    int kk_ = cbase + (TAU);              \
    kk_ = kk_ < 0 ? 0 : kk_;              \
    lw_ = load4u(L.lx + kk_);             \
    ls_ = load4u(L.ly + kk_);             \
    iv_ = load4u(L.inv + kk_);            \
    ln4_ = L.ly[kk_ + 4];                 \
  }
  if (!FAST) MW_COEF(0, a_lw, a_ls, a_iv, a_ln4)
  float cur = 0.f;
  const bool consumer = (wave > 0);
  volatile lds_i *pprod = prog + (wave > 0 ? wave - 1 : 0);
  int seen = gbase - 1;
  if (consumer) __builtin_amdgcn_s_sleep(8);  // start about one chunk behind the producer: its progress word is then
                                              // normally ahead of what is asked for and nobody spins
  // one chunk: prefetch the operands of the next chunk into (nR,nE,nN / n_*), run 4 steps on (cR,cE,cN / c_*)
#define MW_CHUNK(TAU0, cR, cE, cN, nR, nE, nN, c_lw, c_ls, c_iv, c_ln4, n_lw, n_ls, n_iv, n_ln4)                  \
  {                                                                                                               \
    const int tau0_ = (TAU0);                                                                                     \
    const int q_ = tau0_ >> 2;                                                                                    \
    if (consumer && q_ >= 1) { /* the previous sweep must have finished chunk q-1 (S = 8) */                      \
      const int need_ = gbase + q_ - 1;                                                                           \
      while (seen < need_) {                                                                                      \
        __builtin_amdgcn_s_sleep(1);                                                                              \
        seen = *pprod;                                                                                            \
      }                                                                                                           \
    }                                                                                                             \
    asm volatile("" ::: "memory"); /* the prefetches below must not be hoisted above the wait (volatile orders only volatile) */ \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                               \
      nR[e] = pR[tau0_ + 4 + e];                                                                                  \
      nE[e] = pE[tau0_ + 4 + e];                                                                                  \
      nN[e] = pN[tau0_ + 4 + e];                                                                                  \
    }                                                                                                             \
    if (consumer) seen = *pprod; /* for the next chunk: usually already far enough */                             \
    asm volatile("" ::: "memory");                                                                                \
    float le4[4] = {0.f, 0.f, 0.f, 0.f};                                                                          \
    if (!FAST) {                                                                                                  \
      MW_COEF(tau0_ + 4, n_lw, n_ls, n_iv, n_ln4)                                                                 \
      le4[0] = lane_next(c_lw.v[1]);                                                                              \
      le4[1] = lane_next(c_lw.v[2]);                                                                              \
      le4[2] = lane_next(c_lw.v[3]);                                                                              \
      le4[3] = lane_next(n_lw.v[0]);                                                                              \
    }                                                                                                             \
    const int jb = tau0_ + jo;                                                                                    \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                               \
      float lw, le, ls, ln, iv;                                                                                   \
      if (FAST) { /* out-of-domain neighbours read as d = 0: a boundary face may keep the uniform coefficient */  \
        lw = le = ls = ln = cu;                                                                                   \
        iv = iv0;                                                                                                 \
        if (ZB) {                                                                                                 \
          const int j = jb + e;                                                                                   \
          iv = (j == 1) ? iv1 : ((j == m_) ? iv2 : iv0);                                                          \
        }                                                                                                         \
      } else {                                                                                                    \
        lw = c_lw.v[e];                                                                                           \
        le = le4[e];                                                                                              \
        ls = c_ls.v[e];                                                                                           \
        ln = (e < 3) ? c_ls.v[(e + 1) & 3] : c_ln4;                                                               \
        iv = c_iv.v[e];                                                                                           \
      }                                                                                                           \
      const float prev = cur;                                                                                     \
      const float dW = lane_prev(prev);                                                                           \
      const float v = -(dW * lw + cE[e] * le + prev * ls + cN[e] * ln - cR[e]) * iv; /* MG.pde:84-88 */          \
      const float v2 = sel_mask(v, cE[e], replay);                                                                \
      cur = ((unsigned)(jb + e - 1) < m_on) ? v2 : 0.f;                                                           \
      lds_f *w = ((unsigned)(jb + e - 1) < m_wr) ? (pW + tau0_ + e) : dmy;                                        \
      *w = cur;                                                                                                   \
    }                                                                                                             \
    asm volatile("" ::: "memory"); /* the progress word goes out after the data */                                \
    if (lane64 == 0) prog[wave] = gbase + q_;                                                                     \
    asm volatile("" ::: "memory");                                                                                \
  }
  for (int tau0 = 0; tau0 < T; tau0 += 8) {
    MW_CHUNK(tau0, aR, aE, aN, bR, bE, bN, a_lw, a_ls, a_iv, a_ln4, b_lw, b_ls, b_iv, b_ln4)
    MW_CHUNK(tau0 + 4, bR, bE, bN, aR, aE, aN, b_lw, b_ls, b_iv, b_ln4, a_lw, a_ls, a_iv, a_ln4)
  }
#undef MW_CHUNK
#undef MW_COEF
}

// One call = everything a level does on the way up, like mg_up_smooth, with the multi-wave smoother: strips of
// H rows (H = LevelDesc.mw_H, chosen on the host so that both rings fit the workgroup's LDS).
__device__ inline void mg_up_smooth_mw(const bool HASC, const Lv &F, const Lv &Cc, bool need_r, float *lds, double *sm,
                                       float *rr_out, float *linf_out, ProfT &pf, float *xsum_out = nullptr,
                                       const int rmode = 0) {
  const int n = F.n, m = F.m, n_ = n - 2, m_ = m - 2;
  const LevelDesc &D = *F.desc;
  const int H = uni(D.mw_H), NR = H + 6;
  float *dr = lds + kMwGuardLo, *rr = dr + NR * m + kMwGuardMid, *dummy = rr + NR * m + kMwGuardMid;
  const int tid = threadIdx.x;
  const int mc = Cc.m;
  const int PW = m;
  const int rstep = kThreads / m, cstep = kThreads - rstep * m;
  const int r0 = tid / m, c0 = tid - r0 * m;
  double rr_part = 0, xs_part = 0;
  float linf = 0.f;
  const int CPR = m_ >> 2;
  const bool chunked = ((m_ & 3) == 0);
  const DescU du = desc_u(D);
  UpGeom U{&F, &Cc, &D, n, m, n_, m_, mc, PW, CPR, rr, dr, du};
  U.NR = NR;
  U.store_r = (rmode != 1);
  U.update_x = (rmode != 2);
  // a thread keeps its 4-column chunk and walks down the rows of its group: no division in the strip loops
  const int ngrp = chunked ? (kThreads / CPR) : 1, rg = chunked ? (tid / CPR) : 0, c4 = chunked ? (tid - rg * CPR) : 0;
  const bool grp_on = chunked && (rg < ngrp);
  __syncthreads();
  int gbase = 0;
  if (tid < 4) ((lds_i *)(lds_f *)dummy)[64 + tid] = -1;  // progress words of the four waves
  int loaded = 0;
  for (int I0 = 1; I0 - 3 <= n_; I0 += H) {
    const int hi = I0 + H;
    const bool sp_fill = (hi + 1 >= du.sp_i0) && (loaded <= du.sp_i1);
    if (chunked) {
      if (grp_on)
        for (int g = loaded + 1 + rg; g <= hi; g += ngrp) {
          const FillIn in = fill_load(U, HASC, g, c4);
          fill_apply(U, HASC, g, c4, in, sp_fill);
        }
    } else {
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
    if (need_r) pf.mark(PF_UP0);
    {
      const int lm = n_ - I0 + 4, Lmax = lm < H ? lm : H;
      const bool special = (hi >= du.sp_i0) && (I0 - 4 <= du.sp_i1);
      gbase += 4096;  // chunk numbers never repeat within a level call, so the progress words need no reset
      if (H <= 30) {  // two sweeps per wave
        if (special)
          gs_multiwave<false, true, 2>(F, rr, dr, dummy, NR, I0, Lmax, gbase);
        else if (du.zb)
          gs_multiwave<true, true, 2>(F, rr, dr, dummy, NR, I0, Lmax, gbase);
        else
          gs_multiwave<true, false, 2>(F, rr, dr, dummy, NR, I0, Lmax, gbase);
      } else {
        if (special)
          gs_multiwave<false, true, 1>(F, rr, dr, dummy, NR, I0, Lmax, gbase);
        else if (du.zb)
          gs_multiwave<true, true, 1>(F, rr, dr, dummy, NR, I0, Lmax, gbase);
        else
          gs_multiwave<true, false, 1>(F, rr, dr, dummy, NR, I0, Lmax, gbase);
      }
      lds_barrier();  // every sweep of the strip is complete
    }
    if (need_r) pf.mark(PF_GS0);
    // ---- exit: rows whose d (and d of both row neighbours) is final: d.setBC(); increment() ----
    const int e0 = (I0 - 4 < 1) ? 1 : I0 - 4, e1 = (I0 + H - 4 >= n_) ? n_ : I0 + H - 5;
    const bool sp_exit = (e1 + 1 >= du.sp_i0) && (e0 - 1 <= du.sp_i1);
    if (chunked) {
      if (grp_on)
        for (int g = e0 + rg; g <= e1; g += ngrp) {
          const ExitIn in = exit_load(U, HASC, g, c4);
          exit_apply(U, HASC, need_r, g, c4, in, sp_exit, rr_part, xs_part, linf);
        }
    } else {
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
          float xo = F.x[kk];
          if (HASC) xo = xo + prolong_at(Cc.x, mc, g, j, n_, m_);
          const float xnew = xo + d;
          F.x[kk] = xnew;
          if (need_r) xs_part += (double)xnew;
          const bool gi0 = (g == 1), gi1 = (g == n_), gj0 = (j == 1), gj1 = (j == m_);
          if (gi0 | gi1 | gj0 | gj1) {
            const float dp = HASC ? prolong_at(Cc.x, mc, g, j, n_, m_) : 0.f;
#define GHOST_UPD(GI, GJ)            \
  {                                  \
    float xg = F.x[(GI) * m + (GJ)]; \
    if (HASC) xg = xg + dp;          \
    F.x[(GI) * m + (GJ)] = xg + d;   \
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
    if (need_r) pf.mark(PF_INC0);
  }
  __syncthreads();  // phase boundary: the x (and r) this level stored must be visible to what follows
  if (need_r && rmode != 2) {
    *rr_out = uni((float)block_sum_d(rr_part, sm));
    *linf_out = uni(block_max_f(linf, sm));
    if (xsum_out) *xsum_out = uni((float)block_sum_d(xs_part, sm));
  }
}
