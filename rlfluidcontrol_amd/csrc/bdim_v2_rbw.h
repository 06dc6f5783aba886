// bdim_v2_rbw.h -- perf mode (cfg.gs_mode = 1): the up-leg of a level as register-resident, temporally blocked red-black
// row pipelines (mg_up_rbw) and its host-side shape test.  Included by bdim_v2_body.h inside every instantiation
// namespace (the host asks afc::prod for the shape test); the device code is compiled for afc::perf only (AFC_RB).

// (host side of mg_up_rbw, perf mode: which levels it serves -- in every instantiation, the host asks afc::prod)
__device__ __host__ __forceinline__ bool up_rbw_shape_ok(int n, int m) {
  const int n_ = n - 2, m_ = m - 2;
  return (m_ & 1) == 0 && m_ >= 2 && m_ <= 128 && n_ >= 2;
}
__device__ __host__ __forceinline__ int up_rbw_lds_floats(int m, int sp_i0, int sp_i1) {
  const int ns = (sp_i1 >= sp_i0) ? sp_i1 - sp_i0 + 1 : 0;
  return (3 * ns + 1) * m + 16;
}
#if AFC_RB
// =============================================================================================
// PERF MODE, round 5: the whole up-leg of a level -- prolongation increment, d = r*inv, the eight colour half-sweeps of
// four red-black Gauss-Seidel sweeps, d.setBC(), increment (MG.pde:73-97 with the smoother's loops in red-black order)
// -- as INDEPENDENT ROW PIPELINES, one per wave, temporally blocked, out of REGISTERS: no LDS ring, no workgroup
// barrier, no flags.  Wave w owns rows a .. b of the level and walks down rows lo = a - 9 .. hi = b + 9.  Lane l holds
// the two cells (2l+1, 2l+2) of every row of a window of twelve rows (d and r after the prolongation increment: 48
// registers); the four neighbours of a cell are the same lane's registers of the rows above / below, the lane's other
// cell, and one value from lane-1 or lane+1 (DPP wave_shr / wave_shl).  In step T the wave
//   fills row T                (r - A prolongate(coarse.x), d = r*inv: one 8-byte load per lane, the coarse row of the
//                               lane's column pair -- prolongation is constant over 2x2 cells)
//   applies half-sweep h to row T-1-h, h = 0 .. 7 (its inputs, the other colour of rows T-2-h .. T-h after half-sweep
//                               h-1, were produced earlier in this step / in step T-1: the dependency cone of the eight
//                               half-sweeps is what the nine halo rows on either side are for -- they are recomputed
//                               by both neighbours, identically, and written by neither)
//   lets row T-9 leave         (rows T-10 .. T-8 are final: d.setBC(), x += d (+ prolongate), r -= A d for the norms)
// Which cell of a lane a half-sweep updates depends on the parity of the row: the step function is instantiated for the
// twelve window slots, so slot indices and parities are compile-time constants and the window stays in registers.
// Rows that cross a body take their coefficients from a copy of those rows of lx / ly / inv staged in LDS by all waves
// at the start of the call.  Every cell sees the half-sweeps in the order of the barrier form (gs_rb_stage) with the
// same operands in the same order: same bits (tests/test_gpu_perfmode.py pins the solver on a numpy red-black multigrid).
// For levels of an even number of at most 128 columns whose staged rows fit the LDS region (LevelDesc::rbw, set by the
// host); on the finest level only when the new residual is not written back (rmode 1: a wave reads the old r of its
// neighbours' rows) -- the barrier form serves the rest.
constexpr int kRbwP = 12, kRbwHalo = 9;
struct RbwCtx {  // wave-uniform
  const float *r, *xc;
  float *x, *rout;
  const lds_f *slx, *sly, *sinv;  // the staged rows sp0 .. sp1 (+1 for lx) of the coefficient arrays
  int n_, m_, m, mc, lo, hi, a, b, sp0, sp1;
  bool hasc, need_r, store_r, update_x, zb;
  float cu, iv0;
};
struct RbwLane {
  int jA;
  bool on, first, last;
  float ivA0, ivA1, ivA2, ivB0, ivB1, ivB2;  // inverse diagonal of the lane's two cells by row class (levels with zeroed boundary faces)
  float lsA, lnB, diagA, diagB;              // interior rows, exact form: south face of cell A / north face of cell B, diagonals
};
struct RbwWin {
  float dA[kRbwP], dB[kRbwP], rA[kRbwP], rB[kRbwP];
  float cm, c0, cp;  // prolongate(coarse.x) of the lane's column pair in rows T-1, T, T+1
};
__device__ __forceinline__ float rbw_coarse(const RbwCtx &C, int t, int l) {  // MG.pde:139-152 incl. the setBC mirror
  const int tc = t < 1 ? 1 : (t > C.n_ ? C.n_ : t);
  return C.xc[((tc - 1) / 2 + 1) * C.mc + l + 1];
}
// coefficients of cell (i, jA + B); exact: boundary faces of a zero_bnd level are 0 (fill / exit), else every face = cu
// (the smoother: an out-of-domain neighbour reads as d = 0, so cu * 0 == 0 * ghost)
template <bool B>
__device__ __forceinline__ void rbw_coef(const RbwCtx &C, const RbwLane &Ln, int i, const bool exact, float &lw, float &le,
                                         float &ls, float &ln, float &iv) {
  if (i >= C.sp0 && i <= C.sp1) {
    const int o = (i - C.sp0) * C.m + Ln.jA + (B ? 1 : 0);
    lw = C.slx[o];
    le = C.slx[o + C.m];
    ls = C.sly[o];
    ln = C.sly[o + 1];
    iv = C.sinv[o];
  } else {
    const bool r1 = (i == 1), rn = (i == C.n_);
    const bool z = exact && C.zb;
    lw = (z && r1) ? 0.f : C.cu;
    le = (z && rn) ? 0.f : C.cu;
    ls = (z && !B && Ln.first) ? 0.f : C.cu;
    ln = (z && B && Ln.last) ? 0.f : C.cu;
    if (C.zb)
      iv = B ? (r1 ? Ln.ivB1 : (rn ? Ln.ivB2 : Ln.ivB0)) : (r1 ? Ln.ivA1 : (rn ? Ln.ivA2 : Ln.ivA0));
    else
      iv = C.iv0;
  }
}
// One step of the walk (see above).  S = T mod kRbwP: window slots and row parities are compile-time constants.
// FAST: every row the step touches (T-10 .. T) is an interior row of the grid (2 .. n_-1) without special cells -- no
// row tests, analytic coefficients from the lane's constants.  The general form tests each row (inside the grid?
// crossing a body? first / last row?) with wave-uniform branches.  Rows of the grid outside the wave's own cone
// (lo .. hi) are treated like any other: what they hold never reaches a row the wave owns.
template <int S, bool FAST>
__device__ __forceinline__ void rbw_step(const RbwCtx &C, const RbwLane &Ln, RbwWin &W, const int T, double &rr_part,
                                         double &xs_part, float &linf) {
#define RBW_SL(k) ((((S) - (k)) % kRbwP + kRbwP) % kRbwP)
  const int l = (Ln.jA - 1) >> 1;
  {  // ---- fill row T ----
    const int g = T;
    float rvA = 0.f, rvB = 0.f, dvA = 0.f, dvB = 0.f;
    if (FAST || (g >= 1 && g <= C.n_)) {
      float r2[2];
      __builtin_memcpy(r2, C.r + g * C.m + Ln.jA, 8);
      rvA = r2[0], rvB = r2[1];
      float lwA, leA, lsA, lnA, ivA, lwB, leB, lsB, lnB, ivB, diagA, diagB;
      if (FAST) {
        lwA = leA = lnA = lwB = leB = lsB = C.cu;
        lsA = Ln.lsA, lnB = Ln.lnB, diagA = Ln.diagA, diagB = Ln.diagB;
        ivA = Ln.ivA0, ivB = Ln.ivB0;
      } else {
        rbw_coef<false>(C, Ln, g, true, lwA, leA, lsA, lnA, ivA);
        rbw_coef<true>(C, Ln, g, true, lwB, leB, lsB, lnB, ivB);
        diagA = -(lwA + leA + lsA + lnA), diagB = -(lwB + leB + lsB + lnB);
      }
      if (C.hasc) {  // r -= A * prolongate(coarse.x)   (PoissonMatrix.pde:53-64 order, as fill_apply)
        const float c0 = W.c0, cw = W.cm, ce = W.cp;
        const float cprev = lane_prev_keep(c0, c0);
        const float cnx = lane_next(c0), cnext = Ln.last ? c0 : cnx;
        rvA = rvA - (c0 * diagA + cw * lwA + ce * leA + cprev * lsA + c0 * lnA);
        rvB = rvB - (c0 * diagB + cw * lwB + ce * leB + c0 * lsB + cnext * lnB);
      }
      rvA = Ln.on ? rvA : 0.f;
      rvB = Ln.on ? rvB : 0.f;
      dvA = rvA * ivA;  // MG.pde:80
      dvB = rvB * ivB;
    }
    if (C.hasc) {
      W.cm = W.c0;
      W.c0 = W.cp;
      W.cp = rbw_coarse(C, g + 2, l);
    }
    W.rA[RBW_SL(0)] = rvA;
    W.rB[RBW_SL(0)] = rvB;
    W.dA[RBW_SL(0)] = dvA;
    W.dB[RBW_SL(0)] = dvB;
  }
  // ---- half-sweep h on row T-1-h ----
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const int i = T - 1 - h;
    // the row's parity is that of S - 1 - h (T - S is a multiple of kRbwP, which is even): half-sweep h takes the cells with
    // (i + j) & 1 == (h & 1), i.e. cell B (even column) of the rows with (i & 1) == (h & 1)
    const bool cellB = ((((S - 1 - h) % 2) + 2) % 2) == (h & 1);
    const int si = RBW_SL(1 + h), sw = RBW_SL(2 + h), se = RBW_SL(h);
    if (FAST || (i >= 1 && i <= C.n_)) {
      float lw, le, ls, ln, iv;
      if (cellB) {
        if (FAST)
          lw = le = ls = ln = C.cu, iv = Ln.ivB0;
        else
          rbw_coef<true>(C, Ln, i, false, lw, le, ls, ln, iv);
        const float dW = (FAST || i > 1) ? W.dB[sw] : 0.f, dE = (FAST || i < C.n_) ? W.dB[se] : 0.f;
        const float dS = W.dA[si], dN = lane_next(W.dA[si]);
        const float t = dW * lw + dE * le + dS * ls + dN * ln - W.rB[si];
        const float v = -t * iv;  // MG.pde:84-88
        W.dB[si] = Ln.on ? v : 0.f;
      } else {
        if (FAST)
          lw = le = ls = ln = C.cu, iv = Ln.ivA0;
        else
          rbw_coef<false>(C, Ln, i, false, lw, le, ls, ln, iv);
        const float dW = (FAST || i > 1) ? W.dA[sw] : 0.f, dE = (FAST || i < C.n_) ? W.dA[se] : 0.f;
        const float dS = lane_prev(W.dB[si]), dN = W.dB[si];
        const float t = dW * lw + dE * le + dS * ls + dN * ln - W.rA[si];
        const float v = -t * iv;
        W.dA[si] = Ln.on ? v : 0.f;
      }
    }
  }
  {  // ---- row T-9 leaves: d.setBC(); x += d; r -= A d   (MG.pde:90-97, as exit_apply) ----
    const int g = T - 9;
    const bool own = (g >= C.a) && (g <= C.b);
    if (own) {  // (wave-uniform)
      const int sg = RBW_SL(9), sw = RBW_SL(10), se = RBW_SL(8);
      const float dA = W.dA[sg], dB = W.dB[sg];
      const int kk = g * C.m + Ln.jA;
      float dp = 0.f;
      if (C.hasc) dp = rbw_coarse(C, g, l);
      if (C.need_r) {
        const float dWa = (FAST || g > 1) ? W.dA[sw] : dA, dEa = (FAST || g < C.n_) ? W.dA[se] : dA;  // the ghost rows mirror (setBC)
        const float dWb = (FAST || g > 1) ? W.dB[sw] : dB, dEb = (FAST || g < C.n_) ? W.dB[se] : dB;
        const float pv = lane_prev(dB), nx = lane_next(dA);
        const float dSa = Ln.first ? dA : pv, dNb = Ln.last ? dB : nx;
        float lwA, leA, lsA, lnA, ivA, lwB, leB, lsB, lnB, ivB, diagA, diagB;
        if (FAST) {
          lwA = leA = lnA = lwB = leB = lsB = C.cu;
          lsA = Ln.lsA, lnB = Ln.lnB, diagA = Ln.diagA, diagB = Ln.diagB;
        } else {
          rbw_coef<false>(C, Ln, g, true, lwA, leA, lsA, lnA, ivA);
          rbw_coef<true>(C, Ln, g, true, lwB, leB, lsB, lnB, ivB);
          diagA = -(lwA + leA + lsA + lnA), diagB = -(lwB + leB + lsB + lnB);
        }
        const float rnA = W.rA[sg] - (dA * diagA + dWa * lwA + dEa * leA + dSa * lsA + dB * lnA);
        const float rnB = W.rB[sg] - (dB * diagB + dWb * lwB + dEb * leB + dA * lsB + dNb * lnB);
        if (Ln.on) {
          const float pa = rnA * rnA, pb = rnB * rnB;
          rr_part += (double)pa;
          rr_part += (double)pb;
          linf = jmax(linf, jabs(rnA));
          linf = jmax(linf, jabs(rnB));
          if (C.store_r) {
            const float o2[2] = {rnA, rnB};
            __builtin_memcpy(C.rout + kk, o2, 8);
          }
        }
      }
      if (C.update_x && Ln.on) {
        float x2[2];
        __builtin_memcpy(x2, C.x + kk, 8);
        float xa = x2[0], xb = x2[1];
        if (C.hasc) xa = xa + dp, xb = xb + dp;
        xa = xa + dA;
        xb = xb + dB;
        if (C.need_r) {
          xs_part += (double)xa;
          xs_part += (double)xb;
        }
        const float o2[2] = {xa, xb};
        __builtin_memcpy(C.x + kk, o2, 8);
        // ghosts mirror the adjacent interior cell (d.setBC(), Field.pde:211-224)
        if (Ln.first) {
          float xg = C.x[kk - 1];
          if (C.hasc) xg = xg + dp;
          C.x[kk - 1] = xg + dA;
        }
        if (Ln.last) {
          float xg = C.x[kk + 2];
          if (C.hasc) xg = xg + dp;
          C.x[kk + 2] = xg + dB;
        }
        if (!FAST && (g == 1 || g == C.n_)) {
          const int kg = ((g == 1) ? 0 : (C.n_ + 1)) * C.m + Ln.jA;
          float g2[2];
          __builtin_memcpy(g2, C.x + kg, 8);
          float ga = g2[0], gb = g2[1];
          if (C.hasc) ga = ga + dp, gb = gb + dp;
          const float q2[2] = {ga + dA, gb + dB};
          __builtin_memcpy(C.x + kg, q2, 8);
          if (Ln.first) {
            float xg = C.x[kg - 1];
            if (C.hasc) xg = xg + dp;
            C.x[kg - 1] = xg + dA;
          }
          if (Ln.last) {
            float xg = C.x[kg + 2];
            if (C.hasc) xg = xg + dp;
            C.x[kg + 2] = xg + dB;
          }
        }
      }
    }
  }
#undef RBW_SL
}
__device__ __forceinline__ void mg_up_rbw(const bool HASC, const Lv &F, const Lv &Cc, bool need_r, float *lds, float *rr_out,
                                 float *linf_out, float *xsum_out, const int rmode) {
  const int n = F.n, m = F.m, n_ = n - 2, m_ = m - 2;
  const int tid = threadIdx.x;
  const LevelDesc &D = *F.desc;
  const DescU du = desc_u(D);
  const int ns = (du.sp_i1 >= du.sp_i0) ? du.sp_i1 - du.sp_i0 + 1 : 0;
  // the rows sp_i0 .. sp_i1 of lx (one more: the east faces of the last row), ly and inv, staged once per call
  float *slx = lds + (((16u - (lds_addr(lds) & 15u)) & 15u) >> 2), *sly = slx + (ns + 1) * m, *sinv = sly + ns * m;
  __syncthreads();
  if (ns > 0) {
    const int o = du.sp_i0 * m;
    for (int q = tid; q < (ns + 1) * m; q += kThreads) slx[q] = F.lx[o + q];
    for (int q = tid; q < ns * m; q += kThreads) {
      sly[q] = F.ly[o + q];
      sinv[q] = F.inv[o + q];
    }
    if (tid == 0) sinv[ns * m] = 0.f;  // (ly of the cell past the last staged one is read by an idle lane)
  }
  __syncthreads();
  double rr_part = 0, xs_part = 0;
  float linf = 0.f;
  const int w = uni((int)(threadIdx.x >> 6));
  const int nw = (n_ >= 128) ? 4 : ((n_ >= 32) ? 2 : 1);
  const int RW = (n_ + nw - 1) / nw;
  const int a = 1 + w * RW, b = (a + RW - 1 < n_) ? a + RW - 1 : n_;
  if (w < nw && a <= n_) {
    RbwCtx C;
    C.r = F.r, C.xc = Cc.x, C.x = F.x, C.rout = F.r;
    C.slx = (const lds_f *)slx, C.sly = (const lds_f *)sly, C.sinv = (const lds_f *)sinv;
    C.n_ = n_, C.m_ = m_, C.m = m, C.mc = uni(Cc.m);
    C.a = a, C.b = b;
    C.lo = (a - kRbwHalo < 1) ? 1 : a - kRbwHalo;
    C.hi = (b + kRbwHalo > n_) ? n_ : b + kRbwHalo;
    C.sp0 = du.sp_i0, C.sp1 = du.sp_i1;
    C.hasc = HASC, C.need_r = need_r, C.store_r = (rmode != 1), C.update_x = (rmode != 2), C.zb = du.zb != 0;
    C.cu = du.cu, C.iv0 = du.iv0;
    RbwLane Ln;
    const int l0 = tid & 63;
    const int l = (2 * l0 + 2 <= m_) ? l0 : 0;
    Ln.on = (2 * l0 + 2 <= m_);
    Ln.jA = Ln.on ? 2 * l + 1 : 1;  // (idle lanes alias column 1: every address stays inside the arrays, every result is masked)
    Ln.first = (l0 == 0);
    Ln.last = (2 * l0 + 2 == m_);
    Ln.ivA0 = Ln.ivA1 = Ln.ivA2 = Ln.ivB0 = Ln.ivB1 = Ln.ivB2 = du.iv0;
    if (C.zb) {  // invtab[3 * row class + column class] (class 0 interior, 1 first, 2 last)
      const int ca = Ln.first ? 1 : 0, cb = Ln.last ? 2 : 0;
      Ln.ivA0 = D.invtab[ca], Ln.ivA1 = D.invtab[3 + ca], Ln.ivA2 = D.invtab[6 + ca];
      Ln.ivB0 = D.invtab[cb], Ln.ivB1 = D.invtab[3 + cb], Ln.ivB2 = D.invtab[6 + cb];
    }
    // interior rows, exact form (fill / exit): the only zeroed faces are the wall columns' (levels with zeroed boundary faces)
    Ln.lsA = (C.zb && Ln.first) ? 0.f : C.cu;
    Ln.lnB = (C.zb && Ln.last) ? 0.f : C.cu;
    Ln.diagA = -(C.cu + C.cu + Ln.lsA + C.cu);
    Ln.diagB = -(C.cu + C.cu + C.cu + Ln.lnB);
    RbwWin W;
#pragma unroll
    for (int q = 0; q < kRbwP; ++q) W.dA[q] = W.dB[q] = W.rA[q] = W.rB[q] = 0.f;
    W.cm = W.c0 = W.cp = 0.f;
    if (HASC) {
      W.cm = rbw_coarse(C, C.lo - 1, l);
      W.c0 = rbw_coarse(C, C.lo, l);
      W.cp = rbw_coarse(C, C.lo + 1, l);
    }
    const int Tend = C.b + 9;  // row b leaves in step b + 9 (rows past hi are filled like any other: harmless)
    for (int T0 = (C.lo / kRbwP) * kRbwP; T0 <= Tend; T0 += kRbwP) {
#define RBW_STEP(S_)                                                                                       \
  {                                                                                                        \
    const int T_ = T0 + (S_);                                                                              \
    if (T_ >= C.lo && T_ <= Tend) {                                                                        \
      if (T_ - 10 >= 2 && T_ <= n_ - 1 && (T_ < C.sp0 || T_ - 10 > C.sp1))                                \
        rbw_step<S_, true>(C, Ln, W, T_, rr_part, xs_part, linf);                                          \
      else                                                                                                 \
        rbw_step<S_, false>(C, Ln, W, T_, rr_part, xs_part, linf);                                         \
    }                                                                                                      \
  }
      RBW_STEP(0) RBW_STEP(1) RBW_STEP(2) RBW_STEP(3) RBW_STEP(4) RBW_STEP(5)
      RBW_STEP(6) RBW_STEP(7) RBW_STEP(8) RBW_STEP(9) RBW_STEP(10) RBW_STEP(11)
#undef RBW_STEP
    }
  }
  __syncthreads();  // phase boundary: the x (and r) this level stored must be visible to what follows
  if (need_r && rmode != 2) {
    block_reduce3(rr_part, xs_part, linf, lds);
    *rr_out = uni((float)rr_part);
    *linf_out = uni(linf);
    if (xsum_out) *xsum_out = uni((float)xs_part);
  }
}
#endif  // AFC_RB
