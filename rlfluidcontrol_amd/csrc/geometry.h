// geometry.h -- host-side, once-per-config construction of everything in the BDIM step that
// does not depend on the flow state: immersed-boundary kernels, wall normals, body-velocity
// weights and the whole multigrid coefficient hierarchy.
//
// In the reference these are recomputed every actuated step (BDIM.get_coeffs, BDIM.pde:126-138)
// and every V-cycle (MG.restrict(A), MG.pde:70,108-122).  They are constant because the circles
// rotate without moving (CircleBody.rotate only stores dphi, Body.pde:412-415) and dt is fixed
// (AFCCylinder.pde:47), so the product builds them once, on the host, with Java float semantics
// (float32, no FMA, sin/cos/sqrt in double then narrowed, the reference's evaluation order) --
// in particular the tie-sensitive wall-normal selection (Body.pde:223-229) is evaluated exactly
// once and shared by all envs.
#pragma once
#include <cstdint>
#include <vector>

namespace afc {

struct CaseParams {  // AFCCylinder ctor arguments, AFCCylinder.pde:11
  int resolution = 24, x_lengths = 16, y_lengths = 8, reynolds = 500;
  float t_step = 0.0075f, dR = 0.125f, gR = 0.2f, theta = 1.0471976f;
};

struct MGLevel {
  int n = 0, m = 0;              // array sizes incl. ghosts
  std::vector<float> lx, ly, inv;  // PoissonMatrix.lower.x/.y and .inv (PoissonMatrix.pde:38-51)
};

struct PressSample {  // one bilinear sample of p (Field.linear, Field.pde:175-190)
  int i, j;           // base cell
  float s, t;         // weights; exact==1 means s==0&&t==0 -> returns a[i][j] directly
  int exact;
};

struct StaticFields {
  int n = 0, m = 0;  // BDIM.n, BDIM.m (ghosts included)
  float dt = 0, nu = 0, D = 0;
  // BDIM coefficient fields (BDIM.pde:132-196), reference layout a[i*m+j]
  std::vector<float> delx, dely, del1x, del1y, wnxx, wnxy, wnyx, wnyy, cx, cy;
  // body velocity ub (BDIM.pde:140-149, BodyUnion.pde:84-92, Body.pde:234-240):
  //   ub.x = ((0 - ry1*dphi1)/dt)*w1 + ((0 - ry2*dphi2)/dt)*w2     (u points)
  //   ub.y = ((0 + rx1*dphi1)/dt)*w1 + ((0 + rx2*dphi2)/dt)*w2     (v points)
  std::vector<float> w1x, w2x, ry1x, ry2x, w1y, w2y, rx1y, rx2y;
  std::vector<MGLevel> levels;  // level 0 uses lower = c
  float mg_tol = 0;             // MG.pde:49-50
  // pressure force on the main cylinder (Body.pressForce, Body.pde:296-303): 40 segments
  std::vector<PressSample> force_samples;
  std::vector<float> seg_l, seg_nx, seg_ny;
  // band of cells touched by the immersed-boundary forcing (|d|<eps) -- informational
  int band_cells = 0, ub_cells = 0;
  // interior cells where BDIM.updateUP is not the identity (del != 1 or del1 != 0), per component
  std::vector<int> band_x, band_y;
};

// Builds all of the above.  Throws std::runtime_error on an invalid configuration
// (e.g. a grid the multigrid cannot coarsen, MG.pde:99-106).
StaticFields build_static(const CaseParams &cp);

// SaveScalar probe positions (SaveScalar.pde:28-41,61-72) as bilinear samples of p
std::vector<PressSample> build_probes(const CaseParams &cp, int num);

}  // namespace afc
