// A PROJECTION TAIL INSIDE THE ADVECTION PASS THAT FOLLOWS IT (included by bdim_v2_body.h, inside the namespace of the
// instantiation).
//
// u.project ends with u -= c grad(p + s), u.setBC() (VectorField.pde:136-140) -- in the corrector followed by the Heun
// average u = (u + us) / 2 (BDIM.pde:95-96) -- and the next thing that happens to that field is AdvDif (BDIM.pde:82, 90-93;
// VectorField.pde:170-179).  As two passes (project_tail_v2, then advdif_v3) the projected field is written to memory
// (2 floats per cell) and read back (2 floats) a gigabyte of other traffic later -- beyond L2 and beyond the Infinity
// Cache (tools/mall_test.hip).  Here the advection's row pipelines stage the RAW rows (F as forcing + setBC left it in
// front of the solve) and each wave projects a row IN ITS LDS RING one iteration before its stencils read it:
//
//   * everything arrives by LDS-DMA, as advdif_v3's source rows do: the raw velocity rows two rows ahead (rings of 6
//     rows per component instead of 7), the row of p one row ahead into a single row slot (p of the row above is the
//     previous row's, kept in registers), for rows that cross a body the two rows of face coefficients likewise, and
//     the base rows (HEUN: the rows of us) double-buffered: 19 row slots per wave instead of advdif_v3's 18.  Nothing
//     comes through registers asynchronously: a load whose destination the compiler may copy or spill before the
//     hand-placed wait is a race (tried: loads by inline assembly into tied operands -- the compiler moved the set between
//     registers while the load was in flight; wrong bits under load only)
//   * the projected row is written back into the ring slot (with the wall columns setBC gives it) and -- by the wave
//     that OWNS the row -- to the field `q`.  q is never the raw field: that stays intact, so the two halo rows either
//     side of a wave's quarter are projected a second time by the neighbour, in its own ring, from the same operands --
//     no flags, no ordering between waves, no barrier
//   * what setbc_rows does after the tail moves to the owners of rows 1 and n-2: row 0 = the projected row 1, then
//     u.x of row 1 = 1 (inflow); row n-1 = the projected row n-2 with the outflow correction 1 - mean(row n-2), the
//     mean added in the reference's order (Field.pde:302-318) through v_readlane
//
// Same operands, same operations, same order as project_tail_v2 + setbc_rows + advdif_v3: the bits do not change
// (tests/test_gpu_parity.py::test_fused_projection_tail_is_bit_exact, tests/test_gpu_round2.py::test_fused_tails_*).
// All waits are counted vmcnt waits formed from a running count of the wave's issued vector-memory operations and the
// count at the operation waited for (`issued - tag`): the wave's operations complete in order, and the producer's
// operations per row vary (own rows store, halo rows do not; body rows copy two more rows; rows 1 and n-2 store twice).
//
// One column tile (256x128, 128x64), >= 8 rows per wave, folded walls; two tiles behind AFC_FUSE_T2 (below).
constexpr int kAdvfNS = 6;  // ring rows per component: i-2 .. i+3
__device__ __host__ __forceinline__ int advf_lds_floats(int C) { return 4 * (2 * kAdvfNS + 7) * (64 * C + 4) + 4; }
__device__ __host__ __forceinline__ bool advf_shape_ok(int n, int m, int C) {
  return (C == 1 || C == 2) && (m - 2) <= 64 * C && ((m - 2) % C) == 0 && ((n - 2) & 3) == 0 && (n - 2) >= 4 * 8;
}
// two column tiles of two cells per lane (T2): 384x192, 512x256
__device__ __host__ __forceinline__ bool advf2_shape_ok(int n, int m) {
  return (m - 2) > 128 && (m - 2) <= 256 && ((m - 2) & 3) == 0 && ((n - 2) & 1) == 0 && (n - 2) >= 2 * 8;
}
// Two-tile grids: compiled into their step kernels only with -DAFC_FUSE_T2=1 (tools/build_variant.sh t2 -DAFC_FUSE_T2=1).
// Bit-identical there too (tools/fuse_dbg.py 24 / 32 on such a build), but not faster: on the same memory at 1024 envs the
// fused form is +3.2 % per launch at 512x256 (-0.9 % at 512 envs), and in the 384x192 kernel the mere presence of the
// second advection function moves the allocation of the whole kernel: +23 % per launch fused or not.
#ifndef AFC_FUSE_T2
#define AFC_FUSE_T2 0
#endif
// by the columns-per-lane class of the step kernel (CPL 1, 2: one tile; 3, 4: two tiles)
__device__ __host__ __forceinline__ bool advf_ok_for(int n, int m, int cpl) {
  return cpl <= 2 ? advf_shape_ok(n, m, cpl) : (AFC_FUSE_T2 != 0 && advf2_shape_ok(n, m));
}

// HEUN = false: the corrector's advection (BDIM.update2) on the projected predictor field: raw = F of the predictor,
//   b = the base field u0 (row copies, one row ahead), q = where the projected field goes (`us` of the Heun average).
// HEUN = true: the NEXT sim-step's predictor advection (BDIM.update) on u = (project(F2) + us) / 2 (BDIM.pde:95-96):
//   raw = F of the corrector, b = us (its rows are averaged into the projected row as it is formed), q = u itself --
//   the advection's base is its own source, taken from the ring.  The sim-step before left out its projection tail.
// T2 = true: grids of two column tiles (384x192, 512x256; C = 2): the waves are a 2 x 2 arrangement of row halves and
//   column tiles as in advdif_v3.  A tile's ring rows reach two columns into the neighbouring tile, and those must hold
//   PROJECTED values too: after its own cells a wave projects that pair of halo cells (one lane; their p above in two
//   more registers).  The outflow row's sequential mean crosses the two waves of the lower row half: the left tile's
//   wave adds its columns and hands the partial sum over through LDS, the right tile's wave finishes it and hands back
//   the correction.
template <int C, bool HEUN, bool T2 = false>
__device__ AFC_ADV_ATTR void advdif_v3f(const StepParams &P, const float *__restrict__ sx, const float *__restrict__ sy,
                                        const float *__restrict__ bx, const float *__restrict__ by,
                                        float *__restrict__ dx, float *__restrict__ dy, float *lds,
                                        const float *__restrict__ pp, const float shift, float *__restrict__ qx_,
                                        float *__restrict__ qy_) {
  static_assert(C == 1 || C == 2, "column tiles of 64*C columns");
  static_assert(!T2 || C == 2, "two tiles: two cells per lane");
  constexpr int PR = 64 * C + 4, NS = kAdvfNS;
  constexpr unsigned PRB = PR * 4u;
  constexpr int YO = NS * PR;
  constexpr int ST = 4;
  const int n = uni(P.n), n_ = n - 2, m = uni(P.m), m_ = m - 2;
  const int wave = uni((int)(threadIdx.x >> 6));
  const int ncol = T2 ? 2 : 1;
  const int tcol = T2 ? (wave & 1) : 0, trow = T2 ? (wave >> 1) : wave;
  const int W = T2 ? (m_ >> 1) : m_, lanes_on = W / C;
  const int jlo = 1 + tcol * W, c0 = jlo - 2;  // first interior column of the tile; column at ring index 0
  const int pofs = 2 * tcol;  // the row of p sits two columns further right in its slot for a right-hand tile (whose left halo cells need p of column jlo-3; it has no right halo)
  const int LQ = (W + 7) >> 2;
  const int RPW = T2 ? ((n_ + 1) >> 1) : ((n_ + 3) >> 2);
  const int i_first = 1 + trow * RPW, i_last = (i_first + RPW - 1 < n_) ? i_first + RPW - 1 : n_;
  const unsigned ring = uni((int)(((lds_addr(lds) + 15u) & ~15u) + (unsigned)wave * ((2 * NS + 7) * PRB)));
  // base x, y of even rows | of odd rows | p, cx, cy
  const unsigned bslot = ring + (unsigned)(2 * NS) * PRB, pslot = bslot + 4u * PRB;
  const float nu = uni(P.nu), dt = uni(P.dt);
  const float *sxc = sx + c0, *syc = sy + c0, *bxc = bx + c0, *byc = by + c0;  // ring index 0 is column c0 (one tile: "column -1")
  const float *ppc = uni(pp) + (c0 - pofs);
  // two tiles: four words behind the waves' regions for the hand-over of the outflow sum: partial sum | flag | correction | flag
  const unsigned hand = uni((int)(((lds_addr(lds) + 15u) & ~15u) + 4u * ((2 * NS + 7) * PRB)));
  if constexpr (T2) {
    if (threadIdx.x == 0) {
      *(volatile __attribute__((address_space(3))) int *)(uintptr_t)(hand + 4u) = 0;
      *(volatile __attribute__((address_space(3))) int *)(uintptr_t)(hand + 12u) = 0;
    }
    __syncthreads();
  }
  const int sp0 = uni(P.lev[0].sp_i0), sp1 = uni(P.lev[0].sp_i1);
  const float cu = uni(P.lev[0].cu);
  const float *pcx = uni(G(P.cx)) + c0, *pcy = uni(G(P.cy)) + c0;
  if (i_first <= i_last) {
    const int r_hi = (i_last + 2 < n - 1) ? i_last + 2 : n - 1;  // last row the stencils of this wave read
    const int p_hi = (r_hi < n_) ? r_hi : n_;                    // last row this wave stages and projects
    const int g0 = (i_first - 2 > 1) ? i_first - 2 : 1;          // first row it projects
    int issued = 0;                                              // vector-memory operations of this wave so far
    int r_next = i_first - 2, s_next = 0;
#define ADVF_STAGE()                                                                   \
  {                                                                                    \
    if (r_next >= 1 && r_next <= p_hi) {                                               \
      const unsigned ln_ = lane_fresh();                                               \
      if (ln_ < (unsigned)LQ) {                                                        \
        const unsigned so_ = 4u * (unsigned)(r_next * m) + 16u * ln_;                  \
        lds_dma16(sxc, so_, ring + (unsigned)s_next * PRB);                            \
        lds_dma16(syc, so_, ring + (unsigned)(s_next + NS) * PRB);                     \
      }                                                                                \
      issued += 2;                                                                     \
    }                                                                                  \
    ++r_next;                                                                          \
    s_next = (s_next + 1 == NS) ? 0 : s_next + 1;                                      \
  }
#define ADVF_BASE(row)                                                                 \
  {                                                                                    \
    const unsigned ln_ = lane_fresh();                                                 \
    if (ln_ < (unsigned)LQ) {                                                          \
      const unsigned so_ = 4u * (unsigned)((row) * m) + 16u * ln_;                     \
      const unsigned lb_ = bslot + (unsigned)(2 * ((row) & 1)) * PRB;                  \
      lds_dma16(bxc, so_, lb_);                                                        \
      lds_dma16(byc, so_, lb_ + PRB);                                                  \
    }                                                                                  \
    issued += 2;                                                                       \
  }
    // p of row g (and, for a row that crosses a body, its face coefficients) into the wave's three row slots
#define ADVF_ISSUE(g)                                                                  \
  {                                                                                    \
    const unsigned ln_ = lane_fresh();                                                 \
    const unsigned so_ = 4u * (unsigned)((g) * m) + 16u * ln_;                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the slots' readers are done */ \
    if (ln_ < (unsigned)LQ) lds_dma16(ppc, so_, pslot);                                \
    issued += 1;                                                                       \
    if ((g) >= sp0 && (g) <= sp1) {                                                    \
      if (ln_ < (unsigned)LQ) {                                                        \
        lds_dma16(pcx, so_, pslot + PRB);                                              \
        lds_dma16(pcy, so_, pslot + 2u * PRB);                                         \
      }                                                                                \
      issued += 2;                                                                     \
    }                                                                                  \
  }
    float pw[C];   // p of the row above the next row to project (this lane's columns)
    float pwh[C];  // two tiles: the same for the pair of halo cells (lane 0)
    // Row g of the raw field, in ring slot s, becomes the projected row: VectorField.pde:138 + the wall columns of
    // Field.pde:217-224 (as project_tail_v2 folds them) + the row part of setBC for rows 1 and n-2 (setbc_rows)
    auto project_row = [&](const int g, const int s, const bool own_row, const unsigned prow) {
      const unsigned ln_ = lane_fresh();
      float corr = 0.f;  // row n-2: the outflow correction, formed by the pass over the wave's own cells
      // C cells of the row at ring index bidx: the wave's own cells (halo = false: lane l has bidx = 2 + C l and column
      // jlo + C l) or, two tiles, the pair of cells of the neighbouring tile its stencils reach (halo = true: ring only)
      auto cells = [&](const unsigned bidx, const int j0, const bool on, const bool halo, float (&pwr)[C]) {
        const bool own = own_row && !halo;
        lds_f *ru = (lds_f *)(uintptr_t)(ring + 4u * bidx + (unsigned)s * PRB);
        float u[C], v[C], pc[C], ps0, cx[C], cy[C];
        const lds_f *rp = (const lds_f *)(uintptr_t)(prow + 4u * (bidx + (unsigned)pofs));
        const lds_f *rc = (const lds_f *)(uintptr_t)(pslot + PRB + 4u * bidx);
        if constexpr (C == 2) {
          const f2 a = *(const lds_f2a *)ru, b = *(const lds_f2a *)(ru + YO), c = *(const lds_f2a *)rp;
          u[0] = a.x, u[1] = a.y, v[0] = b.x, v[1] = b.y, pc[0] = c.x, pc[1] = c.y;
        } else {
          u[0] = ru[0], v[0] = ru[YO], pc[0] = rp[0];
        }
        ps0 = rp[-1];
        if (g >= sp0 && g <= sp1) {  // rows that cross a body: the real coefficient arrays
          if constexpr (C == 2) {
            const f2 a = *(const lds_f2a *)rc, b = *(const lds_f2a *)(rc + PR);
            cx[0] = a.x, cx[1] = a.y, cy[0] = b.x, cy[1] = b.y;
          } else {
            cx[0] = rc[0], cy[0] = rc[PR];
          }
        } else {  // verified on the host: every face coefficient of the row equals cu (LevelDesc)
#pragma unroll
          for (int e = 0; e < C; ++e) cx[e] = cy[e] = cu;
        }
        float un[C], vn[C], vs[C];
#pragma unroll
        for (int e = 0; e < C; ++e) {
          const float pcs = pc[e] + shift, pws = pwr[e] + shift;
          const float pss = ((e > 0) ? pc[e > 0 ? e - 1 : 0] : ps0) + shift;
          const float gx = pcs - pws, gy = pcs - pss;  // Field.pde:76-77
          un[e] = u[e] + cx[e] * (gx * -1.f);          // VectorField.pde:138
          vn[e] = v[e] + cy[e] * (gy * -1.f);
          vs[e] = vn[e];
          pwr[e] = pc[e];
        }
        // the lane at the lower wall (lane 0 of a left-hand tile) and at the upper wall (the last lane of a right-hand one)
        const bool first = !halo && (ln_ == 0u) && (tcol == 0), last = !halo && (ln_ == (unsigned)(lanes_on - 1)) && (tcol == ncol - 1);
        if (first) vs[0] = 0.f;
        const float wu = first ? un[0] : un[C - 1], wv = first ? vn[0] : 0.f;  // the lane's wall column (lane 0: 0, last lane: m-1)
        const int wi = first ? -1 : C;                                          // ... and its place in the ring row
        // one row of the result: into ring row r (lane position) and -- by the owner -- into row `grow` of the field q.
        // HEUN: averaged with row `grow` of us (in the base slot of its parity), wall columns and all (BDIM.pde:95-96 after
        // setBC: project_tail_v2's `(un + b) * 0.5f`, `(0.f + d) * 0.5f` at the walls, and its four redone rows)
        auto emit = [&](lds_f *r, const int grow, const float *xu, const float xw) {
          float au[C], av[C], aw = xw, awv = wv;
#pragma unroll
          for (int e = 0; e < C; ++e) au[e] = xu[e], av[e] = vs[e];
          if constexpr (HEUN) {
            const lds_f *rq = (const lds_f *)(uintptr_t)(bslot + (unsigned)(2 * (grow & 1)) * PRB + 4u * bidx);
            float qx[C], qy[C];
            if constexpr (C == 2) {
              const f2 a = *(const lds_f2a *)rq, b = *(const lds_f2a *)(rq + PR);
              qx[0] = a.x, qx[1] = a.y, qy[0] = b.x, qy[1] = b.y;
            } else {
              qx[0] = rq[0], qy[0] = rq[PR];
            }
#pragma unroll
            for (int e = 0; e < C; ++e) au[e] = (xu[e] + qx[e]) * 0.5f, av[e] = (vs[e] + qy[e]) * 0.5f;
            if (first || last) aw = (xw + rq[wi]) * 0.5f, awv = (wv + rq[PR + wi]) * 0.5f;
          }
          if (on) {
#pragma unroll
            for (int e = 0; e < C; ++e) r[e] = au[e], r[YO + e] = av[e];
            if (first || last) r[wi] = aw, r[YO + wi] = awv;
          }
          if (own) {
            if (first || last) {
              const unsigned gb = 4u * (unsigned)(grow * m + (first ? 0 : m - 1));
              st1b(qx_, gb, aw);
              st1b(qy_, gb, awv);
            }
            if (on) {
              const unsigned ob = 4u * (unsigned)(grow * m + j0);
              __builtin_memcpy((char *)qx_ + ob, au, sizeof(float) * C);
              __builtin_memcpy((char *)qy_ + ob, av, sizeof(float) * C);
            }
            issued += ST;
          }
        };
        if (g == 1) {  // setBC: row 0 = row 1, then u.x[1][.] = 1 (Field.pde:207-212)
          float one[C];
#pragma unroll
          for (int e = 0; e < C; ++e) one[e] = 1.f;
          emit((lds_f *)(uintptr_t)(ring + 4u * bidx + (unsigned)(s == 0 ? NS - 1 : s - 1) * PRB), 0, un, wu);
          emit(ru, 1, one, 1.f);
        } else {
          emit(ru, g, un, wu);
        }
        if (g == n_) {  // row n-1 = row n-2, u.x with the flux correction 1 - mean (Field.pde:213-216, 302-318: in this order)
          if (!halo) {
            typedef volatile __attribute__((address_space(3))) int lds_vi;
            float sum = 0;
            if (T2 && tcol == 1) {  // the left tile's columns come first
              while (*(lds_vi *)(uintptr_t)(hand + 4u) == 0) __builtin_amdgcn_s_sleep(2);
              sum = __builtin_bit_cast(float, *(lds_vi *)(uintptr_t)hand);
            }
            for (int l = 0; l < lanes_on; ++l) {
#pragma unroll
              for (int e = 0; e < C; ++e)
                sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, un[e]), l));
            }
            if (T2 && tcol == 0) {
              *(lds_vi *)(uintptr_t)hand = __builtin_bit_cast(int, sum);
              *(lds_vi *)(uintptr_t)(hand + 4u) = 1;
              while (*(lds_vi *)(uintptr_t)(hand + 12u) == 0) __builtin_amdgcn_s_sleep(2);
              corr = __builtin_bit_cast(float, *(lds_vi *)(uintptr_t)(hand + 8u));
            } else {
              sum /= (float)(m - 2);
              corr = 1.f - sum;
              if (T2) {
                *(lds_vi *)(uintptr_t)(hand + 8u) = __builtin_bit_cast(int, corr);
                *(lds_vi *)(uintptr_t)(hand + 12u) = 1;
              }
            }
          }
          float uc[C];
#pragma unroll
          for (int e = 0; e < C; ++e) uc[e] = un[e] + corr;
          emit((lds_f *)(uintptr_t)(ring + 4u * bidx + (unsigned)(s + 1 == NS ? 0 : s + 1) * PRB), n - 1, uc, wu);
        }
      };
      cells(2u + (unsigned)C * ln_, jlo + C * (int)ln_, ln_ < (unsigned)lanes_on, false, pw);
      if constexpr (T2) cells(tcol ? 0u : (unsigned)(W + 2), 0, ln_ == 0u, true, pwh);
    };
    // ---- prologue: raw rows i_first-2 .. i_first+2 staged and projected; then row i_first+3, base row i_first and p
    // of row i_first+3 on their way.
    // HEUN = false: the up to six rows of p the five projections need (g0-1 .. i_first+2) arrive TOGETHER, in the six
    // row slots that are still idle -- ring slot 5 of both components and the four base slots -- so that the prologue
    // is one round trip plus one per row that crosses a body (the single pair of coefficient slots), not one per row.
    // HEUN = true: the base slots carry the rows of us; one round trip per row. ----
    const int q_hi = (p_hi == n_) ? n - 1 : p_hi;  // HEUN: last row of us this wave needs (the last wave: row n-1 too)
    int tag_q = 0;
    for (int q = 0; q < 5; ++q) ADVF_STAGE()
    if constexpr (!HEUN) {
      {
        const unsigned ln_ = lane_fresh();
        for (int g = g0 - 1; g <= i_first + 2; ++g) {
          const int k = g - (i_first - 3);  // 0 .. 5
          const unsigned slot = (k < 2) ? ring + (unsigned)(5 + k * NS) * PRB : bslot + (unsigned)(k - 2) * PRB;
          if (ln_ < (unsigned)LQ) lds_dma16(ppc, 4u * (unsigned)(g * m) + 16u * ln_, slot);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int k0 = (g0 - 1) - (i_first - 3);
        const unsigned slot0 = (k0 < 2) ? ring + (unsigned)(5 + k0 * NS) * PRB : bslot + (unsigned)(k0 - 2) * PRB;
        const lds_f *rp = (const lds_f *)(uintptr_t)(slot0 + 4u * (unsigned)(2 + C * (int)ln_ + pofs));
        const lds_f *rh = (const lds_f *)(uintptr_t)(slot0 + 4u * (unsigned)((tcol ? 0 : W + 2) + pofs));
#pragma unroll
        for (int e = 0; e < C; ++e) pw[e] = rp[e], pwh[e] = rh[e];
      }
      for (int g = g0; g <= i_first + 2; ++g) {
        const int k = g - (i_first - 3);
        const unsigned slot = (k < 2) ? ring + (unsigned)(5 + k * NS) * PRB : bslot + (unsigned)(k - 2) * PRB;
        if (g >= sp0 && g <= sp1) {
          const unsigned ln_ = lane_fresh();
          const unsigned so_ = 4u * (unsigned)(g * m) + 16u * ln_;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (ln_ < (unsigned)LQ) {
            lds_dma16(pcx, so_, pslot + PRB);
            lds_dma16(pcy, so_, pslot + 2u * PRB);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        project_row(g, g - (i_first - 2), g >= i_first, slot);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the readers of the borrowed slots are done
    } else if (g0 > 1 && (sp1 < g0 || sp0 > g0 + 4)) {
      // no row of the prologue crosses a body: the two coefficient slots serve as p slots, three rows of p and two of us
      // per round trip -- three round trips instead of six
      const unsigned ln_ = lane_fresh();
      auto p_to = [&](const int g, const int k) {
        if (ln_ < (unsigned)LQ) lds_dma16(ppc, 4u * (unsigned)(g * m) + 16u * ln_, pslot + (unsigned)k * PRB);
        issued += 1;
      };
      for (int q = 0; q < 5; ++q) {  // row g0 + q (ring slot q), its p in slot (q + 1) % 3
        if (q == 0) {
          p_to(g0 - 1, 0), p_to(g0, 1), p_to(g0 + 1, 2);
          ADVF_BASE(g0)
          ADVF_BASE(g0 + 1)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const lds_f *rp = (const lds_f *)(uintptr_t)(pslot + 4u * (unsigned)(2 + C * (int)ln_ + pofs));
          const lds_f *rh = (const lds_f *)(uintptr_t)(pslot + 4u * (unsigned)((tcol ? 0 : W + 2) + pofs));
#pragma unroll
          for (int e = 0; e < C; ++e) pw[e] = rp[e], pwh[e] = rh[e];
        } else if (q == 2) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          p_to(g0 + 2, 0), p_to(g0 + 3, 1), p_to(g0 + 4, 2);
          ADVF_BASE(g0 + 2)
          ADVF_BASE(g0 + 3)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (q == 4) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          ADVF_BASE(g0 + 4)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        project_row(g0 + q, q, g0 + q >= i_first, pslot + (unsigned)((q + 1) % 3) * PRB);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      ADVF_ISSUE(g0 - 1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      {
        const lds_f *rp = (const lds_f *)(uintptr_t)(pslot + 4u * (unsigned)(2 + C * (int)lane_fresh() + pofs));
        const lds_f *rh = (const lds_f *)(uintptr_t)(pslot + 4u * (unsigned)((tcol ? 0 : W + 2) + pofs));
#pragma unroll
        for (int e = 0; e < C; ++e) pw[e] = rp[e], pwh[e] = rh[e];
      }
      for (int g = g0; g <= i_first + 2; ++g) {
        ADVF_ISSUE(g)
        if (g == 1) ADVF_BASE(0)
        ADVF_BASE(g)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        project_row(g, g - (i_first - 2), g >= i_first, pslot);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the next row's copy of us goes where the row before this one's was read)
      }
    }
    ADVF_STAGE()  // row i_first+3
    ADVF_BASE(HEUN ? i_first + 3 : i_first)
    int tag_b = issued;  // base row i
    ADVF_ISSUE(i_first + 3)
    int tag_p = issued;
    f2 fE2[C];
    int s0 = 0;  // ring slot of row i-2
    for (int i = i_first; i <= i_last; ++i) {
      const int t = i - i_first;
      int tag_bn = tag_b;
      if constexpr (!HEUN) {
        if (i < i_last) {
          ADVF_BASE(i + 1)
          tag_bn = issued;
        }
      } else {
        if (i + 4 <= q_hi) {  // us of row i+4 (read when that row is projected, at the end of the next iteration)
          ADVF_BASE(i + 4)
          tag_q = issued;
        }
      }
      if (t > 0) ADVF_STAGE()  // raw row i+4 replaces row i-2
      const int j0 = jlo + C * (int)lane_fresh();
      const unsigned lo = ring + 4u * (unsigned)(j0 - c0);
      const int s1 = (s0 + 1 >= NS) ? s0 + 1 - NS : s0 + 1, s2 = (s0 + 2 >= NS) ? s0 + 2 - NS : s0 + 2,
                s3 = (s0 + 3 >= NS) ? s0 + 3 - NS : s0 + 3, s4 = (s0 + 4 >= NS) ? s0 + 4 - NS : s0 + 4,
                s5 = (s0 + 5 >= NS) ? s0 + 5 - NS : s0 + 5;
      const lds_f *r1 = (const lds_f *)(uintptr_t)(lo + (unsigned)s1 * PRB);
      const lds_f *r2 = (const lds_f *)(uintptr_t)(lo + (unsigned)s2 * PRB);
      const lds_f *r3 = (const lds_f *)(uintptr_t)(lo + (unsigned)s3 * PRB);
      const lds_f *r4 = (const lds_f *)(uintptr_t)(lo + (unsigned)s4 * PRB);
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
#define XY(r, dj) f2{X_##r[(dj) + O_##r], Y_##r[(dj) + O_##r]}
      const bool rok = (i >= 2) && (i <= n - 2);
      const bool rokE = (i + 1 >= 2) && (i + 1 <= n - 2);
      float oxv[C], oyv[C];
      f2 fEn[C], fNn[C], difv[C], ccv[C];
      f2 fSw = {0.f, 0.f};
#pragma unroll
      for (int e = 0; e < C; ++e) {
        const int j = j0 + e;
        const lmask mC = lanes_if((unsigned)(j - 2) <= (unsigned)(m - 4));
        const lmask mCN = lanes_if((unsigned)(j - 1) <= (unsigned)(m - 4));
        const f2 cc = XY(r2, e), ce = XY(r3, e), cw = XY(r1, e), cn = XY(r2, e + 1), cs = XY(r2, e - 1);
        ccv[e] = cc;
        if (t == 0) {
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
      if (T2 && tcol) {  // right-hand tile: lane 0's south face = the north face of column j0-1, evaluated here (e = -1 above)
        const lmask mC = lanes_if((unsigned)(j0 - 3) <= (unsigned)(m - 4));
        const lmask mCN = lanes_if((unsigned)(j0 - 2) <= (unsigned)(m - 4));
        const f2 cc = XY(r2, -1), cn = XY(r2, 0), cs = XY(r2, -2);
        const f2 vn = 0.5f * (f2{cn.y, cc.y} + f2{Y_r1[0], cn.y});
        fSw = vn * quick_face2(cs, cc, cn, XY(r2, 1), vn, rok ? mC : 0, rok ? mCN : 0);
      }
#undef XY
      fSw.x = lane_prev_keep(fSw.x, fNn[C - 1].x);
      fSw.y = lane_prev_keep(fSw.y, fNn[C - 1].y);
      f2 basev[C];
      if constexpr (!HEUN) {
        wait_vm_le(uni(issued - tag_b));  // base row i: copied in iteration i-1 or by the prologue
        const lds_f *bb = (const lds_f *)(uintptr_t)(bslot + (unsigned)(2 * (i & 1)) * PRB + 4u * (unsigned)(j0 - c0));
        if constexpr (C == 2) {
          const f2 ab = LDP(bb, 0), bb2 = LDP(bb + PR, 0);
          basev[0] = f2{ab.x, bb2.x}, basev[1] = f2{ab.y, bb2.y};
        } else {
#pragma unroll
          for (int e = 0; e < C; ++e) basev[e] = f2{bb[e], bb[PR + e]};
        }
      } else {  // BDIM.update: u0 = u
#pragma unroll
        for (int e = 0; e < C; ++e) basev[e] = ccv[e];
      }
      tag_b = tag_bn;
#pragma unroll
      for (int e = 0; e < C; ++e) {
        const f2 fW = fE2[e], fS = (e == 0) ? fSw : fNn[e > 0 ? e - 1 : 0];
        const f2 adv = ((fW - fEn[e]) + (fS - fNn[e]));
        const f2 o = (adv + nu * difv[e]) * dt + basev[e];  // VectorField.pde:174-175
        oxv[e] = o.x;
        oyv[e] = o.y;
        fE2[e] = fEn[e];
      }
      {
        const unsigned ob = 4u * (unsigned)(i * m + j0);
        const unsigned ln_ = lane_fresh();
        const bool first = (ln_ == 0u) && (tcol == 0), last = (ln_ == (unsigned)(lanes_on - 1)) && (tcol == ncol - 1);
        if (first || last) {
          const unsigned gb = 4u * (unsigned)(i * m + (first ? 0 : m - 1));
          st1b(dx, gb, first ? oxv[0] : oxv[C - 1]);
          st1b(dy, gb, first ? oyv[0] : 0.f);
        }
        if (first) oyv[0] = 0.f;
        if (ln_ < (unsigned)lanes_on) {
          __builtin_memcpy((char *)dx + ob, oxv, sizeof(float) * C);
          __builtin_memcpy((char *)dy + ob, oyv, sizeof(float) * C);
        }
        issued += ST;
      }
      asm volatile("" ::: "memory");
      if (t == 0) ADVF_STAGE()  // row i+4 replaces row i-2, whose last reader was this row's west face
      // ---- the projection of row i+3 (the stencils of the next iteration read up to it); p of row i+4 on its way ----
      if (i + 3 <= p_hi) {
        // (HEUN: us of row i+3 was copied an iteration before its p; the last wave's row n-1 of us only in this one)
        wait_vm_le(uni(issued - ((HEUN && i + 3 == n_ && tag_q > tag_p) ? tag_q : tag_p)));
        project_row(i + 3, s5, i + 3 <= i_last, pslot);
        if (i + 4 <= p_hi) {
          ADVF_ISSUE(i + 4)
          tag_p = issued;
        }
      }
      asm volatile("" ::: "memory");
      s0 = s1;
    }
#undef ADVF_STAGE
#undef ADVF_BASE
#undef ADVF_ISSUE
#undef LDP
  }
  __syncthreads();
}
