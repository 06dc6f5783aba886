// The two per-chunk bodies of an up-leg -- fill (prolongation increment, d0 = r inv) and exit (d.setBC, x += d, r -= A d,
// norms) of one four-column chunk of one row -- for rows WITHOUT special cells (included by bdim_v2_body.h inside the
// namespace of the instantiation; used by mg_up_tr's and mg_up_pipe's helper waves).
// ---- the helpers' two bodies, written for what they are: one four-column chunk of one row per lane, rows without special
// cells (fill_load / fill_apply / exit_load / exit_apply -- the generic forms, which the passes that cross a body keep --
// spend two thirds of their ~200 instructions per chunk on 64-bit lane addresses, boundary selects and per-cell
// coefficient look-ups).  Same loads, same expressions in the same order: same bits.
struct TmH {             // wave-uniform unless noted
  const float *r, *xc;   // the level's residual, the coarse level's x
  float *x, *rw;         // the level's x (and r, where it is written back)
  int n, n_, m, mc;
  float cu, iv0, dg;     // dg = -(((cu + cu) + cu) + cu)
  unsigned a_rr, a_inv;  // LDS byte addresses: ring slot 0 column 0 of the r ring, invtab[9]
  unsigned pwb, dofs;    // bytes per ring row; bytes from a cell of the r ring to the same cell of the d ring
  bool hasc, need_r, store_r, update_x;
  // per lane
  int a, c0;             // chunk of the row; row of the lane's chunk in pass p = ND p + c0
  bool first, last;      // the chunk at the lower / upper wall
};
struct TmFillIn {
  F4 r, va;
  float w1, w2, e1, e2;
};
__device__ __forceinline__ TmFillIn tm_fill_load(const TmH &H, const int g) {
  TmFillIn in;
  in.r = ld4b(H.r, 4u * (unsigned)(g * H.m + 1 + 4 * H.a));
  if (H.hasc) {
    const int gw = (g - 1 < 1) ? 1 : g - 1, ge = (g + 1 > H.n_) ? H.n_ : g + 1;
    const unsigned ca = 4u * (unsigned)((((g - 1) >> 1) + 1) * H.mc + 2 * H.a);  // coarse columns 2a .. 2a+3
    const unsigned cw = 4u * (unsigned)((((gw - 1) >> 1) + 1) * H.mc + 2 * H.a + 1), ce = 4u * (unsigned)((((ge - 1) >> 1) + 1) * H.mc + 2 * H.a + 1);
    in.va = ld4b(H.xc, ca);
    float t2[2];
    __builtin_memcpy(t2, (const char *)H.xc + cw, 8);
    in.w1 = t2[0], in.w2 = t2[1];
    __builtin_memcpy(t2, (const char *)H.xc + ce, 8);
    in.e1 = t2[0], in.e2 = t2[1];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) in.va.v[e] = 0.f;
    in.w1 = in.w2 = in.e1 = in.e2 = 0.f;
  }
  return in;
}
// r (after the prolongation increment) and d0 = r * inv of one chunk -> rings (fill_apply's arithmetic, PoissonMatrix.pde:53-64 order)
// NR: rows of the rings (slot of row g = g mod NR)
template <int NR, bool ZB>
__device__ __forceinline__ void tm_fill_apply(const TmH &H, const int g, const TmFillIn &in) {
  const unsigned PWB = H.pwb, RINGB = H.dofs;
  float lwv = H.cu, lev = H.cu, ls0 = H.cu, ln3 = H.cu, ivm = H.iv0, ivf = H.iv0, ivl = H.iv0;
  if (ZB) {  // zeroed boundary faces (coarse levels): the wall chunks and the first / last row know them
    lwv = (g == 1) ? 0.f : H.cu;
    lev = (g == H.n_) ? 0.f : H.cu;
    ls0 = H.first ? 0.f : H.cu;
    ln3 = H.last ? 0.f : H.cu;
    const unsigned at = H.a_inv + ((g == 1) ? 12u : ((g == H.n_) ? 24u : 0u));
    ivm = *(const lds_f *)(uintptr_t)at;
    ivf = H.first ? *(const lds_f *)(uintptr_t)(at + 4u) : ivm;
    ivl = H.last ? *(const lds_f *)(uintptr_t)(at + 8u) : ivm;
  }
  // (scalar on purpose: as packed f32 pairs the same arithmetic issued MORE instructions -- the pairs have to be assembled --
  // and took a third longer at 1024 workgroups; measured, tools/upleg_bench.hip)
  F4 rv = in.r, dv;
  if (H.hasc) {
    const float a1 = in.va.v[1], a2 = in.va.v[2];
    const float a0 = H.first ? a1 : in.va.v[0], a3 = H.last ? a2 : in.va.v[3];
    const float dc[4] = {a1, a1, a2, a2}, dw[4] = {in.w1, in.w1, in.w2, in.w2}, de[4] = {in.e1, in.e1, in.e2, in.e2};
    const float ds[4] = {a0, a1, a1, a2}, dn[4] = {a1, a2, a2, a3};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lsv = (e == 0) ? ls0 : H.cu, lnv = (e == 3) ? ln3 : H.cu;
      const float diag = ZB ? -(lwv + lev + lsv + lnv) : H.dg;
      rv.v[e] = rv.v[e] - (dc[e] * diag + dw[e] * lwv + de[e] * lev + ds[e] * lsv + dn[e] * lnv);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) dv.v[e] = rv.v[e] * ((e == 0) ? ivf : ((e == 3) ? ivl : ivm));  // MG.pde:80
  const unsigned sl = H.a_rr + (unsigned)(g % NR) * PWB + 4u * (unsigned)(1 + 4 * H.a);
  f4a r4, d4;
#pragma unroll
  for (int e = 0; e < 4; ++e) r4[e] = rv.v[e], d4[e] = dv.v[e];
  *(lds_f4 *)(uintptr_t)sl = r4;
  *(lds_f4 *)(uintptr_t)(sl + RINGB) = d4;
}
struct TmExitIn {
  F4 x;
  float a1, a2, gl, gh;
};
__device__ __forceinline__ TmExitIn tm_exit_load(const TmH &H, const int g) {
  TmExitIn in;
  in.a1 = in.a2 = in.gl = in.gh = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) in.x.v[e] = 0.f;
  if (!H.update_x) return in;
  const unsigned kb = 4u * (unsigned)(g * H.m + 1 + 4 * H.a);
  in.x = ld4b(H.x, kb);
  if (H.hasc) {
    float t2[2];
    __builtin_memcpy(t2, (const char *)H.xc + 4u * (unsigned)((((g - 1) >> 1) + 1) * H.mc + 2 * H.a + 1), 8);
    in.a1 = t2[0], in.a2 = t2[1];
  }
  if (H.first) in.gl = ld1b(H.x, kb - 4u);
  if (H.last) in.gh = ld1b(H.x, kb + 16u);
  return in;
}
// d.setBC(); x += d (+ the ghosts that mirror the chunk); r -= A d (exit_apply's arithmetic, MG.pde:90-97)
template <int NR, bool ZB>
__device__ __forceinline__ void tm_exit_apply(const TmH &H, const int g, const TmExitIn &in, double &rr_part, double &xs_part,
                                              float &linf) {
  const unsigned PWB = H.pwb, RINGB = H.dofs;
  const unsigned kb = 4u * (unsigned)(g * H.m + 1 + 4 * H.a);
  const int sg = g % NR;
  const unsigned colb = 4u * (unsigned)(1 + 4 * H.a);
  const unsigned sl = H.a_rr + (unsigned)sg * PWB + colb;  // row g of the r ring; the d ring is RINGB further
  float dp[4] = {0.f, 0.f, 0.f, 0.f};
  if (H.hasc) dp[0] = dp[1] = in.a1, dp[2] = dp[3] = in.a2;
  const f4a dc4 = *(const lds_f4 *)(uintptr_t)(sl + RINGB);
  float d6[6];  // d at columns j-1 .. j+4 with the setBC mirror
  d6[0] = *(const lds_f *)(uintptr_t)(sl + RINGB - 4u);
#pragma unroll
  for (int e = 0; e < 4; ++e) d6[1 + e] = dc4[e];
  d6[5] = *(const lds_f *)(uintptr_t)(sl + RINGB + 16u);
  if (H.first) d6[0] = d6[1];
  if (H.last) d6[5] = d6[4];
  if (H.need_r) {
    // rows g-1, g+1 (clamped to the grid: d.setBC mirrors the first / last row into the ghost rows)
    const int sw = (g == 1) ? sg : (sg == 0 ? NR - 1 : sg - 1), se = (g == H.n_) ? sg : (sg == NR - 1 ? 0 : sg + 1);
    const f4a w4 = *(const lds_f4 *)(uintptr_t)(H.a_rr + RINGB + (unsigned)sw * PWB + colb);
    const f4a e4 = *(const lds_f4 *)(uintptr_t)(H.a_rr + RINGB + (unsigned)se * PWB + colb);
    const f4a o4 = *(const lds_f4 *)(uintptr_t)sl;
    float lwv = H.cu, lev = H.cu, ls0 = H.cu, ln3 = H.cu;
    if (ZB) {
      lwv = (g == 1) ? 0.f : H.cu;
      lev = (g == H.n_) ? 0.f : H.cu;
      ls0 = H.first ? 0.f : H.cu;
      ln3 = H.last ? 0.f : H.cu;
    }
    F4 rn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lsv = (e == 0) ? ls0 : H.cu, lnv = (e == 3) ? ln3 : H.cu;
      const float diag = ZB ? -(lwv + lev + lsv + lnv) : H.dg;
      const float d = d6[1 + e];
      rn.v[e] = o4[e] - (d * diag + w4[e] * lwv + e4[e] * lev + d6[e] * lsv + d6[2 + e] * lnv);
      const float pr = rn.v[e] * rn.v[e];
      rr_part += (double)pr;
      linf = jmax(linf, jabs(rn.v[e]));
    }
    if (H.store_r) st4b(H.rw, kb, rn);
  }
  if (!H.update_x) return;
  F4 xn;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float xo = in.x.v[e];
    if (H.hasc) xo = xo + dp[e];
    xn.v[e] = xo + d6[1 + e];
    if (H.need_r) xs_part += (double)xn.v[e];
  }
  st4b(H.x, kb, xn);
  // ghosts mirror the adjacent interior cell (d.setBC(), Field.pde:211-224)
  if (H.first) {
    float xg = in.gl;
    if (H.hasc) xg = xg + dp[0];
    st1b(H.x, kb - 4u, xg + d6[1]);
  }
  if (H.last) {
    float xg = in.gh;
    if (H.hasc) xg = xg + dp[3];
    st1b(H.x, kb + 16u, xg + d6[4]);
  }
  if (g == 1 || g == H.n_) {
    const unsigned kg = 4u * (unsigned)(((g == 1) ? 0 : (H.n - 1)) * H.m + 1 + 4 * H.a);
    const F4 xg4 = ld4b(H.x, kg);
    F4 xgn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xg = xg4.v[e];
      if (H.hasc) xg = xg + dp[e];
      xgn.v[e] = xg + d6[1 + e];
    }
    st4b(H.x, kg, xgn);
    if (H.first) {
      float xg = ld1b(H.x, kg - 4u);
      if (H.hasc) xg = xg + dp[0];
      st1b(H.x, kg - 4u, xg + d6[1]);
    }
    if (H.last) {
      float xg = ld1b(H.x, kg + 16u);
      if (H.hasc) xg = xg + dp[3];
      st1b(H.x, kg + 16u, xg + d6[4]);
    }
  }
}

