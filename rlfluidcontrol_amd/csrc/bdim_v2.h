// bdim_v2.h -- optimised phases of the per-env BDIM step (same arithmetic, same bits as
// bdim_device.h's reference kernels; selected with StepParams.impl == 2).
//
// Velocity buffers rotate instead of being copied (BDIM.pde:82,91 copy u0/us):
//   A = (ux,uy)   state at the start of the step (= u0 of both half-steps)
//   B = (usx,usy) predictor result  (= `us` of the corrector)
//   C = (fx,fy)   corrector result; Heun average (A = (C+B)/2) lands back in A.
#pragma once
#include "bdim_device.h"

namespace afc {

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

// F = (advection + nu*diffusion)(src)*dt + base on interior cells (VectorField.AdvDif,
// VectorField.pde:170-179), streamed through an LDS tile of R+4 rows.  Wave w owns R/4
// consecutive rows of the block and carries the east-face flux down the rows; lane l owns C
// adjacent columns and carries the north-face flux across them.  Non-interior cells of dst are
// NOT written (setbc_vel defines them).
template <int C>
__device__ inline void advdif_v2(const StepParams &P, const float *__restrict__ sx, const float *__restrict__ sy,
                                 const float *__restrict__ bx, const float *__restrict__ by, float *__restrict__ dx,
                                 float *__restrict__ dy, float *lds) {
  const int n = P.n, m = P.m, N = P.N;
  const int R = P.adv_rows, TR = R + 4;
  float *tx = lds + 2, *ty = lds + 2 + TR * m + 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int RW = R >> 2;
  const int j0 = 1 + C * lane;
  const bool lane_on = (j0 <= m - 2);
  const float nu = P.nu, dt = P.dt;
  for (int a = 1; a <= n - 2; a += R) {
    // ---- stage rows a-2 .. a+R+1 of src into LDS (flat, coalesced float2) ----
    const int t0 = (a - 2) * m;                    // flat index of tile row 0
    const int lo = t0 < 0 ? 0 : t0;
    const int hi = ((a + R + 2) * m < N) ? (a + R + 2) * m : N;
    __syncthreads();  // previous block's readers are done
    for (int q = lo + 2 * threadIdx.x; q < hi; q += 2 * kThreads) {
      const float2 vx = *reinterpret_cast<const float2 *>(sx + q);
      const float2 vy = *reinterpret_cast<const float2 *>(sy + q);
      tx[q - t0] = vx.x;
      tx[q - t0 + 1] = vx.y;
      ty[q - t0] = vy.x;
      ty[q - t0 + 1] = vy.y;
    }
    __syncthreads();
#define TX(gi, gj) tx[((gi) - (a - 2)) * m + (gj)]
#define TY(gi, gj) ty[((gi) - (a - 2)) * m + (gj)]
    if (lane_on) {
      const int i_beg = a + wave * RW;
      float fEx[C], fEy[C];
      for (int ii = 0; ii < RW; ++ii) {
        const int i = i_beg + ii;
        if (i > n - 2) break;
        const bool rok = (i >= 2) && (i <= n - 2);       // row part of the fallback test for cells of row i
        const bool rokE = (i + 1 >= 2) && (i + 1 <= n - 2);
        const bool rokW = (i - 1 >= 2) && (i - 1 <= n - 2);
        float fSx = 0.f, fSy = 0.f;
#pragma unroll
        for (int e = 0; e < C; ++e) {
          const int j = j0 + e;
          if (j > m - 2) break;
          const bool cok = (j >= 2) && (j <= m - 2);
          const bool cokN = (j + 1 >= 2) && (j + 1 <= m - 2);
          const bool cokS = (j - 1 >= 2) && (j - 1 <= m - 2);
          const float xc = TX(i, j), yc = TY(i, j);
          // ---------- x component (btype 1): VectorField.pde:183-188 ----------
          {
            float fW;
            if (ii == 0) {
              const float uo = 0.5f * (TX(i - 1, j) + xc);
              fW = uo * quick_face(TX(i - 2, j), TX(i - 1, j), xc, TX(i + 1, j), uo, rokW && cok, rok && cok);
            } else {
              fW = fEx[e];
            }
            const float ue = 0.5f * (TX(i + 1, j) + xc);
            const float fE = ue * quick_face(TX(i - 1, j), xc, TX(i + 1, j), TX(i + 2, j), ue, rok && cok, rokE && cok);
            if (e == 0) {
              const float vs = 0.5f * (yc + TY(i - 1, j));
              fSx = vs * quick_face(TX(i, j - 2), TX(i, j - 1), xc, TX(i, j + 1), vs, rok && cokS, rok && cok);
            }
            const float vn = 0.5f * (TY(i, j + 1) + TY(i - 1, j + 1));
            const float fN = vn * quick_face(TX(i, j - 1), xc, TX(i, j + 1), TX(i, j + 2), vn, rok && cok, rok && cokN);
            const float adv = ((fW - fE) + (fSx - fN));
            const float dif = TX(i + 1, j) + TX(i, j + 1) - 4 * xc + TX(i - 1, j) + TX(i, j - 1);  // :198-200
            dx[i * m + j] = (adv + nu * dif) * dt + bx[i * m + j];                                  // :174
            fEx[e] = fE;
            fSx = fN;
          }
          // ---------- y component (btype 2): VectorField.pde:189-194 ----------
          {
            float fW;
            if (ii == 0) {
              const float uo = 0.5f * (TX(i, j - 1) + xc);
              fW = uo * quick_face(TY(i - 2, j), TY(i - 1, j), yc, TY(i + 1, j), uo, rokW && cok, rok && cok);
            } else {
              fW = fEy[e];
            }
            const float ue = 0.5f * (TX(i + 1, j - 1) + TX(i + 1, j));
            const float fE = ue * quick_face(TY(i - 1, j), yc, TY(i + 1, j), TY(i + 2, j), ue, rok && cok, rokE && cok);
            if (e == 0) {
              const float vs = 0.5f * (TY(i, j - 1) + yc);
              fSy = vs * quick_face(TY(i, j - 2), TY(i, j - 1), yc, TY(i, j + 1), vs, rok && cokS, rok && cok);
            }
            const float vn = 0.5f * (yc + TY(i, j + 1));
            const float fN = vn * quick_face(TY(i, j - 1), yc, TY(i, j + 1), TY(i, j + 2), vn, rok && cok, rok && cokN);
            const float adv = ((fW - fE) + (fSy - fN));
            const float dif = TY(i + 1, j) + TY(i, j + 1) - 4 * yc + TY(i - 1, j) + TY(i, j - 1);
            dy[i * m + j] = (adv + nu * dif) * dt + by[i * m + j];  // :175
            fEy[e] = fE;
            fSy = fN;
          }
        }
      }
    }
#undef TX
#undef TY
  }
  __syncthreads();
}

// BDIM.updateUP lines 118-120 restricted to the cells where it is not the identity:
// del != 1 or del1 != 0 (|d| < eps, plus the solid interior).  Everywhere else
// u = 1*F - ub*0 + 0*g == F.  Results are staged (scratch) before being scattered so that every
// band cell sees the un-forced F of its neighbours, as the reference's whole-array expression does.
__device__ inline void forcing_band(const StepParams &P, float *fx, float *fy, float *scratch, float dphi1,
                                    float dphi2) {
  const int m = P.m;
  for (int q = threadIdx.x; q < P.nband_x; q += kThreads) {
    const int k = P.band_x[q];
    const float del = P.delx[k], R = fx[k], ub = ub_x_at(P, k, dphi1, dphi2);
    float un = del * R - ub * (del + (-1.f));
    const float dE = fx[k + m] - ub_x_at(P, k + m, dphi1, dphi2);
    const float dW = fx[k - m] - ub_x_at(P, k - m, dphi1, dphi2);
    const float dN = fx[k + 1] - ub_x_at(P, k + 1, dphi1, dphi2);
    const float dS = fx[k - 1] - ub_x_at(P, k - 1, dphi1, dphi2);
    const float g = 0.5f * (P.wnxx[k] * (dE - dW) + P.wnyx[k] * (dN - dS));
    un = un + P.del1x[k] * g;
    scratch[q] = un;
  }
  for (int q = threadIdx.x; q < P.nband_y; q += kThreads) {
    const int k = P.band_y[q];
    const float del = P.dely[k], R = fy[k], ub = ub_y_at(P, k, dphi1, dphi2);
    float un = del * R - ub * (del + (-1.f));
    const float dE = fy[k + m] - ub_y_at(P, k + m, dphi1, dphi2);
    const float dW = fy[k - m] - ub_y_at(P, k - m, dphi1, dphi2);
    const float dN = fy[k + 1] - ub_y_at(P, k + 1, dphi1, dphi2);
    const float dS = fy[k - 1] - ub_y_at(P, k - 1, dphi1, dphi2);
    const float g = 0.5f * (P.wnxy[k] * (dE - dW) + P.wnyy[k] * (dN - dS));
    un = un + P.del1y[k] * g;
    scratch[P.nband_x + q] = un;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < P.nband_x; q += kThreads) fx[P.band_x[q]] = scratch[q];
  for (int q = threadIdx.x; q < P.nband_y; q += kThreads) fy[P.band_y[q]] = scratch[P.nband_x + q];
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// MG.smooth(4) without the trailing increment (MG.pde:80-89): d = r*inv, then four in-place
// lexicographic Gauss-Seidel sweeps, in the reference's exact dependency order.
//
// One wave marches a strip of 16 rows along j.  Lane = 16*k + c: sweep k (0..3), row
// i = I0 - k + c, at step tau it updates column j = tau - c - k + 1.  With this skew the four
// neighbours of cell (i,j) in sweep k were produced exactly one step earlier:
//   d[i-1][j] (sweep k)   by lane (k, c-1)   -> DPP row_shr:1      (c == 0: carry row in LDS)
//   d[i][j-1] (sweep k)   by this lane       -> register
//   d[i+1][j] (sweep k-1) by lane (k-1, c)   -> ds_bpermute        (k == 0: d0 = r*inv from LDS)
//   d[i][j+1] (sweep k-1) by lane (k-1, c-1) -> DPP of the permuted value (c == 0: carry row)
// which is what the sequential loops see, so every cell gets the reference's bits.  The last row
// of each sweep is kept in `carry` for the next strip; the strip of sweep k is shifted up by k
// rows so that sweep k never overtakes sweep k-1.  r and d live in 24-row LDS rings that all four
// waves fill/drain with coalesced accesses; out-of-domain neighbours read as 0 (ghost d is 0 on
// the finest level and multiplies a zero face coefficient on the coarse ones).
// Coefficients: on strips that do not touch the bodies they are synthesised from the level's
// analytic description (LevelDesc), otherwise loaded per cell.
// ---------------------------------------------------------------------------------------------
constexpr int kRing = 24;

__device__ __forceinline__ float dpp_row_shr1(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                0x111, 0xF, 0xF, false));
}

// value held by lane-16 (rows of 16 lanes): two gfx950 row swaps + one select, no LDS round trip
__device__ __forceinline__ float lane_minus16(float v, int lane) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);        // a[0] = [r0,r0,r2,r2], a[1] = [r1,r1,r3,r3]
  auto b = __builtin_amdgcn_permlane32_swap(a[0], a[1], false, false);  // b[0] = [r0,r0,r1,r1]
  return __builtin_bit_cast(float, ((lane >> 4) == 3) ? a[0] : b[0]);   //      -> [ *,r0,r1,r2]
}

struct F4 {
  float v[4];
};
__device__ __forceinline__ F4 load4u(const float *p) {  // 4-byte-aligned 16-byte load (global_load_dwordx4)
  F4 r;
  __builtin_memcpy(&r, p, 16);
  return r;
}

// LDS layout used by the smoother (floats): [pad 32][r ring 24*m][d ring 24*m][carry 4*m][pad 32][dummy 64]
__device__ __forceinline__ int gs_lds_floats(int m) { return 32 + (2 * kRing + 4) * m + 32 + 64; }

// analytic coefficients of an interior cell (LevelDesc) -- used on rows without special cells
__device__ __forceinline__ void coef_analytic(const LevelDesc &D, int i, int j, int n_, int m_, float &lw, float &le,
                                              float &ls, float &ln, float &iv) {
  const float b = D.zero_bnd ? 0.f : D.cu;
  lw = (i == 1) ? b : D.cu;
  le = (i == n_) ? b : D.cu;
  ls = (j == 1) ? b : D.cu;
  ln = (j == m_) ? b : D.cu;
  iv = D.invtab[3 * ((i == 1) ? 1 : ((i == n_) ? 2 : 0)) + ((j == 1) ? 1 : ((j == m_) ? 2 : 0))];
}

template <bool FAST>
__device__ __forceinline__ void gs_march(const Lv &L, const float *rr, float *dr, float *carry, float *dummy,
                                         int I0) {
  const int m = L.m, n_ = L.n - 2, m_ = L.m - 2;
  const int lane = threadIdx.x & 63, k = lane >> 4, c = lane & 15;
  const int i = I0 - k + c;
  const bool row_ok = (i >= 1) && (i <= n_);
  const int ic = i < 0 ? 0 : (i > n_ ? n_ : i);  // clamped for addressing only
  const int slot = (ic % kRing) * m, slotE = ((ic + 1) % kRing) * m;
  const LevelDesc &D = *L.desc;
  float lwU = D.cu, leU = D.cu, iv0 = 0.f, iv1 = 0.f, iv2 = 0.f;
  if (FAST) {  // row-dependent parts of the analytic coefficients
    const int rc = (i == 1) ? 1 : ((i == n_) ? 2 : 0);
    if (D.zero_bnd && i == 1) lwU = 0.f;
    if (D.zero_bnd && i == n_) leU = 0.f;
    iv0 = D.invtab[3 * rc + 0];
    iv1 = D.invtab[3 * rc + 1];
    iv2 = D.invtab[3 * rc + 2];
  }
  const float cu = D.cu;
  const bool zb = D.zero_bnd != 0;
  (void)lwU;
  (void)leU;
  const int steps = (m_ + 18 + 3) & ~3;  // multiple of 4 (the extra steps are inactive)
  // Running LDS pointers: at step tau this lane is at column j = tau + jo (jo = 1-c-k); nothing is
  // clamped -- before/after its row a lane reads whatever lies beside it in LDS and discards it.
  // Operands of step tau+1 are fetched during step tau, BEFORE that step's LDS writes: none of
  // them can be written in between (carry rows are re-written >= 14 steps after their last read,
  // d0 is overwritten by sweep 3 six columns behind sweep 0).  On strips that touch the bodies the
  // five coefficients come from memory, four columns per 4-byte-aligned 16-byte load, one chunk
  // ahead.  Writes of inactive lanes are steered to a dummy word.
  const int jo = 1 - c - k;
  const float *p_rv = rr + slot + jo;
  const float *p_cW = carry + k * m + jo;
  const float *p_cN = carry + ((k + 3) & 3) * m + 1 + jo;
  const float *p_dE = dr + slotE + jo;
  const float *p_dN = dr + slot + 1 + jo;
  float *w_d = dr + slot + jo;
  float *w_c = carry + k * m + jo;
  const bool wd_lane = row_ok && (k == 3);
  const bool wc_lane = (c == 15);
  float *const dmy = dummy + lane;
  // LDS operands of the four steps of a chunk are fetched one chunk ahead (<= 7 steps early: still
  // far inside the hazard margins above)
  float o_rv[4], o_cW[4], o_cN[4], o_dE[4], o_dN[4], y_rv[4], y_cW[4], y_cN[4], y_dE[4], y_dN[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    y_rv[e] = p_rv[e];
    y_cW[e] = p_cW[e];
    y_cN[e] = p_cN[e];
    y_dE[e] = p_dE[e];
    y_dN[e] = p_dN[e];
  }
  F4 c_lw, c_le, c_ls, c_iv, x_lw, x_le, x_ls, x_iv;
  float c_ln4 = 0.f, x_ln4 = 0.f;
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
  if (!FAST) GS_CHUNK(0, x_lw, x_le, x_ls, x_iv, x_ln4)
  float res = 0.f;
  for (int tau0 = 0; tau0 < steps; tau0 += 4) {
    if (!FAST) {
      c_lw = x_lw;
      c_le = x_le;
      c_ls = x_ls;
      c_iv = x_iv;
      c_ln4 = x_ln4;
      GS_CHUNK(tau0 + 4, x_lw, x_le, x_ls, x_iv, x_ln4)
    }
    const float *q_rv = p_rv + tau0 + 4, *q_cW = p_cW + tau0 + 4, *q_cN = p_cN + tau0 + 4, *q_dE = p_dE + tau0 + 4,
                *q_dN = p_dN + tau0 + 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o_rv[e] = y_rv[e];
      o_cW[e] = y_cW[e];
      o_cN[e] = y_cN[e];
      o_dE[e] = y_dE[e];
      o_dN[e] = y_dN[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y_rv[e] = q_rv[e];
      y_cW[e] = q_cW[e];
      y_cN[e] = q_cN[e];
      y_dE[e] = q_dE[e];
      y_dN[e] = q_dN[e];
    }
    float *qw_d = w_d + tau0, *qw_c = w_c + tau0;
    const int jb = tau0 + jo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = jb + e;
      const bool col_ok = (j >= 1) && (j <= m_);
      const bool act = row_ok && col_ok;
      const float rv = o_rv[e], cW = o_cW[e], cN = o_cN[e], dE0 = o_dE[e], dN0 = o_dN[e];
      float lw, le, ls, ln, iv;
      if (FAST) {
        // every out-of-domain neighbour reads as d = 0, so a boundary face may keep the uniform
        // coefficient: cu*0 == 0*ghost.  Only inv knows about the boundary (coarse levels).
        lw = cu;
        le = cu;
        ls = cu;
        ln = cu;
        iv = zb ? ((j == 1) ? iv1 : ((j == m_) ? iv2 : iv0)) : iv0;
      } else {
        lw = c_lw.v[e];
        le = c_le.v[e];
        ls = c_ls.v[e];
        ln = (e < 3) ? c_ls.v[(e + 1) & 3] : c_ln4;
        iv = c_iv.v[e];
      }
      const float prev = res;
      const float up16 = lane_minus16(prev, lane);  // (k-1, c): sweep k-1 at (i+1, j)
      const float dW = dpp_row_shr1(cW, prev);      // (k, c-1): sweep k   at (i-1, j)
      const float dNp = dpp_row_shr1(cN, up16);     // (k-1, c-1): sweep k-1 at (i, j+1)
      const float dE = (k == 0) ? dE0 : up16;
      const float dN = (k == 0) ? dN0 : dNp;
      const float v = -(dW * lw + dE * le + prev * ls + dN * ln - rv) * iv;  // MG.pde:84-88
      res = act ? v : 0.f;
      float *a_d = (wd_lane && col_ok) ? (qw_d + e) : dmy;
      float *a_c = (wc_lane && col_ok) ? (qw_c + e) : dmy;
      *a_d = res;
      *a_c = res;
    }
  }
#undef GS_CHUNK
}

// prolongate(coarse.x) at a (possibly ghost) cell of the fine level (MG.pde:139-152 incl. setBC)
__device__ __forceinline__ float prolong_at(const float *__restrict__ xc, int mc, int i, int j, int n_, int m_) {
  const int ci = i < 1 ? 1 : (i > n_ ? n_ : i), cj = j < 1 ? 1 : (j > m_ ? m_ : j);
  return xc[((ci - 1) / 2 + 1) * mc + ((cj - 1) / 2 + 1)];
}

// One call = everything a level does on the way up (MG.pde:73-76 seen from the level itself):
//   HASC:  d = prolongate(coarse.x); increment()            [x += d, r -= A d]
//   then   smooth(4): d = r*inv; 4 Gauss-Seidel sweeps; d.setBC(); increment()
// streamed strip by strip: r (and p on the finest level) cross HBM once, d never leaves LDS.
// need_r: also form/store r -= A d of the last increment and return r.r / L_inf (finest level).
template <bool HASC>
__device__ inline void mg_up_smooth(const Lv &F, const Lv &Cc, bool need_r, float *lds, double *sm, float *rr_out,
                                    float *linf_out) {
  const int n = F.n, m = F.m, n_ = n - 2, m_ = m - 2;
  float *rr = lds + 32, *dr = rr + kRing * m, *carry = dr + kRing * m, *dummy = carry + 4 * m + 32;
  const int tid = threadIdx.x;
  const LevelDesc &D = *F.desc;
  const int mc = Cc.m;
  const int rstep = kThreads / m, cstep = kThreads - rstep * m;  // flat (row, col) stepping by kThreads
  const int r0 = tid / m, c0 = tid - r0 * m;
  double rr_part = 0;
  float linf = 0.f;
  __syncthreads();
  for (int q = tid; q < 4 * m + 32; q += kThreads) carry[q] = 0.f;
  int loaded = 0;  // rows 1..loaded are (or were) in the rings
  for (int I0 = 1; I0 - 3 <= n_; I0 += 16) {
    // ---- fill rows loaded+1 .. I0+16: r (after the prolongation increment) and d0 = r*inv ----
    const int hi = I0 + 16;
    const bool sp_fill = (hi + 1 >= D.sp_i0) && (loaded <= D.sp_i1);
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
            coef_analytic(D, g, j, n_, m_, lw, le, ls, ln, iv);
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
        const int sl = (g % kRing) * m + j;
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
    __syncthreads();
    if (tid < 64) {
      const bool special = (I0 + 16 >= D.sp_i0) && (I0 - 3 <= D.sp_i1);
      if (special)
        gs_march<false>(F, rr, dr, carry, dummy, I0);
      else
        gs_march<true>(F, rr, dr, carry, dummy, I0);
    }
    __syncthreads();
    // ---- exit: rows whose d (and d of both row neighbours) is final: d.setBC(); increment() ----
    const int e0 = (I0 - 4 < 1) ? 1 : I0 - 4, e1 = (I0 + 12 >= n_) ? n_ : I0 + 11;
    const bool sp_exit = (e1 + 1 >= D.sp_i0) && (e0 - 1 <= D.sp_i1);
    {
      int g = e0 + r0, j = c0;
      while (g <= e1) {
        if (j >= 1 && j <= m_) {
          const int kk = g * m + j;
          const int gw = (g - 1 < 1) ? 1 : g - 1, ge = (g + 1 > n_) ? n_ : g + 1;
          const int js = (j - 1 < 1) ? 1 : j - 1, jn = (j + 1 > m_) ? m_ : j + 1;
          const int sl = (g % kRing) * m;
          const float d = dr[sl + j];
          if (need_r) {
            const float dW = dr[(gw % kRing) * m + j], dE = dr[(ge % kRing) * m + j];
            const float dS = dr[sl + js], dN = dr[sl + jn];
            float lw, le, ls, ln, iv;
            if (sp_exit) {
              lw = F.lx[kk];
              le = F.lx[kk + m];
              ls = F.ly[kk];
              ln = F.ly[kk + 1];
            } else {
              coef_analytic(D, g, j, n_, m_, lw, le, ls, ln, iv);
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
          F.x[kk] = xo + d;
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
    __syncthreads();  // the next strip's fill re-uses ring slots this exit has just read
  }
  if (need_r) {
    *rr_out = (float)block_sum_d(rr_part, sm);
    *linf_out = block_max_f(linf, sm);
  }
}

// MGsolver (MG.pde:30-38) with the fused ascending leg; descending leg = reference kernels
__device__ inline int mg_solve_v2(const StepParams &P, size_t e, double *sm, float *linf_out, Prof &pf, float *lds) {
  const int kc = P.nlev - 1;
  int iter = 0;
  const Lv L0 = make_level(P, e, 0);
  float linf = 0.f;
  while (iter < P.mg_max_iters) {
    ++iter;
    for (int l = 0; l < kc; ++l) {  // descending leg of vCycle (MG.pde:69-70)
      const Lv F = make_level(P, e, l), Cc = make_level(P, e, l + 1);
      mg_dinv(F);  // smooth(0)
      mg_increment(F);
      mg_restrict(F, Cc);
      pf.mark(l == 0 ? PF_DOWN0 : PF_COARSE_DOWN);
    }
    float rr = 0.f;
    {
      const Lv Lk = make_level(P, e, kc);  // coarsest: coarse.smooth(its), MG.pde:73
      mg_up_smooth<false>(Lk, Lk, false, lds, sm, &rr, &linf);
      pf.mark(PF_COARSE_GS);
    }
    for (int l = kc - 1; l >= 0; --l) {
      const Lv F = make_level(P, e, l), Cc = make_level(P, e, l + 1);
      mg_up_smooth<true>(F, Cc, l == 0, lds, sm, &rr, &linf);
      pf.mark(l == 0 ? PF_GS0 : PF_COARSE_GS);
    }
    if (P.exact_sums) rr = seq_inner(L0, sm);
    pf.mark(PF_RR);
    if (rr < P.mg_tol) break;
  }
  *linf_out = linf;
  return iter;
}

template <int C>
__device__ inline void sim_step_v2(const StepParams &P, size_t e, float xi1_m, float xi2_m, double *sm, ShBcast *sh,
                                   float *lds) {
  const size_t eo = e * (size_t)P.Ns;
  const float dphi1 = (2 * xi1_m * P.dt) / (P.dR * P.D);  // AFCCylinder.pde:48-49
  const float dphi2 = (2 * xi2_m * P.dt) / (P.dR * P.D);
  float *Ax = P.ux + eo, *Ay = P.uy + eo, *Bx = P.usx + eo, *By = P.usy + eo, *Cx = P.fx + eo, *Cy = P.fy + eo;
  float *scratch = P.u0x + eo;
  float l0, l1;
  Prof pf;
  pf.start(P);
  // predictor, BDIM.update(): F = AdvDif(u; u0 = u); updateUP
  advdif_v2<C>(P, Ax, Ay, Ax, Ay, Bx, By, lds);
  pf.mark(PF_ADVDIF);
  forcing_band(P, Bx, By, scratch, dphi1, dphi2);
  pf.mark(PF_FORCING);
  setbc_vel(P, Bx, By);
  pf.mark(PF_SETBC);
  const int it0 = project_vel(P, e, Bx, By, sm, &l0, pf, lds);
  // corrector, BDIM.update2(): us = u (= B); F = AdvDif(u; u0); updateUP; u = (u+us)/2
  advdif_v2<C>(P, Bx, By, Ax, Ay, Cx, Cy, lds);
  pf.mark(PF_ADVDIF);
  forcing_band(P, Cx, Cy, scratch, dphi1, dphi2);
  pf.mark(PF_FORCING);
  setbc_vel(P, Cx, Cy);
  pf.mark(PF_SETBC);
  const int it1 = project_vel(P, e, Cx, Cy, sm, &l1, pf, lds);
  for (int k = threadIdx.x; k < P.N; k += kThreads) {  // BDIM.pde:95-96, whole array, no setBC
    Ax[k] = (Cx[k] + Bx[k]) * 0.5f;
    Ay[k] = (Cy[k] + By[k]) * 0.5f;
  }
  if (threadIdx.x == 0) {
    float fx, fy;
    eval_force(P, P.p + eo, &fx, &fy);
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
