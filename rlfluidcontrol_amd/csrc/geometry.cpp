// geometry.cpp -- see geometry.h.  Host code, strict float32 (compiled with -ffp-contract=off;
// x86-64 baseline has no FMA anyway).  Follows the reference's arithmetic statement by statement:
//   CircleBody/EllipseBody/Body   Body.pde:57-121,174-240,382-417
//   OrthoNormal                   OrthoNormal.pde:8-28
//   BodyUnion                     BodyUnion.pde:67-92,145-166
//   BDIM.get_coeffs, delta0/1     BDIM.pde:132-215
//   PoissonMatrix ctor            PoissonMatrix.pde:38-51
//   MG.restrict(PoissonMatrix)    MG.pde:108-122
#include "geometry.h"

#include <cmath>
#include <stdexcept>

namespace afc {
namespace {

// Processing float constants and helpers (PApplet): results of java.lang.Math narrowed to float
constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 6.28318530717958647693f;
inline float jmin(float a, float b) { return (a < b) ? a : b; }
inline float jmax(float a, float b) { return (a > b) ? a : b; }
inline float jabs(float a) { return (a < 0) ? -a : a; }
inline float jsin(float a) { return static_cast<float>(std::sin(static_cast<double>(a))); }
inline float jcos(float a) { return static_cast<float>(std::cos(static_cast<double>(a))); }
inline float jsqrt(float a) { return static_cast<float>(std::sqrt(static_cast<double>(a))); }

// One edge of a polygon with its unit tangent/normal (OrthoNormal.pde:8-18)
struct Edge {
  float len, nx, ny, tx, ty, off, cx, cy;
  Edge(float ax, float ay, float bx, float by) {
    const float dx = ax - bx, dy = ay - by;
    len = jsqrt(dx * dx + dy * dy + 0.0f * 0.0f);  // PVector.mag incl. z
    tx = (bx - ax) / len;
    ty = (by - ay) / len;
    nx = -ty;
    ny = tx;
    off = ax * nx + ay * ny;
    cx = (ax + bx) / 2.f;
    cy = (ay + by) / 2.f;
  }
  float signed_dist(float x, float y) const { return x * nx + y * ny - off; }  // :20-23
};

// The 40-gon the reference uses for every cylinder (EllipseBody ctor, Body.pde:386-395) plus its
// axis-aligned bounding box (Body.end, Body.pde:73-87).  Distance is the analytic circle
// (CircleBody.distance, Body.pde:408-410); normals come from the polygon faces.
class Cylinder {
 public:
  Cylinder(float xc, float yc, float diameter) : xc_(xc), yc_(yc), h_(diameter) {
    const float a = 1.f / 1.0f;
    const float rx = 0.5f * h_ * a, ry = 0.5f * h_;
    constexpr int kSides = 40;
    std::vector<float> px(kSides), py(kSides);
    for (int k = 0; k < kSides; ++k) {
      const float th = -kTwoPi * k / static_cast<float>(kSides);
      px[k] = xc_ + rx * jcos(th);
      py[k] = yc_ + ry * jsin(th);
    }
    for (int k = 0; k < kSides; ++k) faces_.emplace_back(px[k], py[k], px[(k + 1) % kSides], py[(k + 1) % kSides]);
    float lox = xc_, loy = yc_, hix = xc_, hiy = yc_;
    for (int k = 0; k < kSides; ++k) {
      lox = jmin(lox, px[k]);
      loy = jmin(loy, py[k]);
      hix = jmax(hix, px[k]);
      hiy = jmax(hiy, py[k]);
    }
    box_.emplace_back(lox, loy, lox, hiy);
    box_.emplace_back(lox, hiy, hix, hiy);
    box_.emplace_back(hix, hiy, hix, loy);
    box_.emplace_back(hix, loy, lox, loy);
  }
  float xc() const { return xc_; }
  float yc() const { return yc_; }
  const std::vector<Edge> &faces() const { return faces_; }

  float distance(float x, float y) const {  // Body.pde:408-410 + PApplet.mag
    const float ax = x - xc_, ay = y - yc_;
    return jsqrt(ax * ax + ay * ay) - 0.5f * h_;
  }
  float box_distance(float x, float y) const {  // convex 4-gon: Body.pde:181-185
    float d = -1e10f;
    for (const Edge &e : box_) d = jmax(d, e.signed_dist(x, y));
    return d;
  }
  // Body.WallNormal, Body.pde:215-232: strict '>' keeps the FIRST face among ties
  void wall_normal(float x, float y, float &wx, float &wy) const {
    wx = 0;
    wy = 0;
    if (box_distance(x, y) > 3) return;
    float best = -1e10f;
    for (const Edge &e : faces_) {
      const float d = e.signed_dist(x, y);
      if (d > best) {
        best = d;
        wx = e.nx;
        wy = e.ny;
      }
    }
  }

 private:
  float xc_, yc_, h_;
  std::vector<Edge> faces_, box_;
};

float union_kernel(float d) {  // BodyUnion.delta0, BodyUnion.pde:158-166
  if (d <= -1) return 0;
  if (d >= 1) return 1;
  return 0.5f * (1.f + d + jsin(kPi * d) / kPi);
}
float delta0(float eps, float d) {  // BDIM.pde:199-207
  if (d <= -eps) return 0;
  if (d >= eps) return 1;
  return 0.5f * (1.f + d / eps + jsin(kPi * d / eps) / kPi);
}
float delta1(float eps, float d) {  // BDIM.pde:209-215
  if (jabs(d) >= eps) return 0;
  return 0.25f * (eps - (d * d) / eps) -
         1 / kTwoPi * (d * jsin(d * kPi / eps) + eps / kPi * (1 + jcos(d * kPi / eps)));
}

struct Trio {
  const Cylinder *c[3];
  void weights(float x, float y, float w[3]) const {  // BodyUnion.pde:145-156
    float s = 0;
    for (int k = 0; k < 3; ++k) {
      w[k] = union_kernel(-c[k]->distance(x, y) / 3.f);
      s += w[k];
    }
    for (int k = 0; k < 3 && s > 0; ++k) w[k] /= s;
  }
  float distance(float x, float y) const {  // BodyUnion.pde:67-72
    float d = 1e6f;
    for (int k = 0; k < 3; ++k) d = jmin(d, c[k]->distance(x, y));
    return d;
  }
  void wall_normal(float x, float y, float &mx, float &my) const {  // BodyUnion.pde:74-82
    float w[3];
    weights(x, y, w);
    mx = 0;
    my = 0;
    for (int k = 0; k < 3; ++k) {
      float nx, ny;
      c[k]->wall_normal(x, y, nx, ny);
      nx *= w[k];
      ny *= w[k];
      mx += nx;
      my += ny;
    }
  }
};

// Field.setBC (Field.pde:209-234) for the static, non-gradientExit coefficient fields
void set_bc(std::vector<float> &a, int n, int m, int btype, float bval) {
  auto at = [&](int i, int j) -> float & { return a[static_cast<size_t>(i) * m + j]; };
  for (int j = 0; j < m; ++j) {
    at(0, j) = at(1, j);
    at(n - 1, j) = at(n - 2, j);
    if (btype == 1) {
      at(1, j) = bval;
      at(n - 1, j) = bval;
    }
  }
  for (int i = 0; i < n; ++i) {
    at(i, 0) = at(i, 1);
    at(i, m - 1) = at(i, m - 2);
    if (btype == 2) {
      at(i, 1) = bval;
      at(i, m - 1) = bval;
    }
  }
}

void fill_inverse(MGLevel &L) {  // PoissonMatrix.pde:38-51
  const int n = L.n, m = L.m;
  L.inv.assign(static_cast<size_t>(n) * m, 1.0f);
  for (int i = 1; i < n - 1; ++i)
    for (int j = 1; j < m - 1; ++j) {
      const size_t k = static_cast<size_t>(i) * m + j;
      const float sumd = L.lx[k] + L.lx[k + m] + L.ly[k] + L.ly[k + 1];
      if (sumd > 1e-5f) L.inv[k] = -1.f / sumd;
    }
}

bool divisible(int n, int m) {  // MG.pde:99-106 (evaluated on the grid to be coarsened further)
  return (n - 2) % 2 == 0 && (m - 2) % 2 == 0 && n > 4 && m > 4;
}

PressSample make_sample(int n, int m, float x0, float y0) {  // Field.linear btype 0, Field.pde:175-190
  PressSample ps;
  const float x = jmin(jmax(0.5f, x0), n - 1.5f);
  int i = static_cast<int>(x);
  if (i > n - 2) i = n - 2;
  const float y = jmin(jmax(0.5f, y0), m - 1.5f);
  int j = static_cast<int>(y);
  if (j > m - 2) j = m - 2;
  ps.i = i;
  ps.j = j;
  ps.s = x - i;
  ps.t = y - j;
  ps.exact = (ps.s == 0 && ps.t == 0) ? 1 : 0;
  return ps;
}

}  // namespace

StaticFields build_static(const CaseParams &cp) {
  if (cp.resolution < 1 || cp.x_lengths < 1 || cp.y_lengths < 1 || cp.reynolds < 1)
    throw std::runtime_error("afc: resolution, x_lengths, y_lengths and reynolds must be positive");
  StaticFields sf;
  const int ni = cp.x_lengths * cp.resolution, mi = cp.y_lengths * cp.resolution;  // AFCCylinder.pde:12-13
  const int n = ni + 2, m = mi + 2;                                                 // BDIM.pde:42
  sf.n = n;
  sf.m = m;
  sf.dt = cp.t_step * cp.resolution;  // AFCCylinder.pde:20
  sf.D = static_cast<float>(cp.resolution);
  sf.nu = sf.D / cp.reynolds;  // AFCCylinder.pde:34
  const float D = sf.D, eps = 2.0f;
  const float r = (D / 2 + cp.gR * D + cp.dR * D / 2);  // AFCCylinder.pde:29
  const Cylinder main_c(static_cast<float>(ni / 4), static_cast<float>(mi / 2), D);
  const Cylinder ctl1(ni / 4 + r * jcos(cp.theta), mi / 2 - r * jsin(cp.theta), cp.dR * D);
  const Cylinder ctl2(ni / 4 + r * jcos(cp.theta), mi / 2 + r * jsin(cp.theta), cp.dR * D);
  const Trio trio{{&main_c, &ctl1, &ctl2}};

  const size_t N = static_cast<size_t>(n) * m;
  sf.delx.assign(N, 1.f);
  sf.dely.assign(N, 1.f);
  for (auto *v : {&sf.del1x, &sf.del1y, &sf.wnxx, &sf.wnxy, &sf.wnyx, &sf.wnyy, &sf.w1x, &sf.w2x, &sf.ry1x,
                  &sf.ry2x, &sf.w1y, &sf.w2y, &sf.rx1y, &sf.rx2y})
    v->assign(N, 0.f);

  for (int i = 1; i < n - 1; ++i)
    for (int j = 1; j < m - 1; ++j) {
      const size_t k = static_cast<size_t>(i) * m + j;
      const float xu = static_cast<float>(i - 0.5f), yu = static_cast<float>(j);  // u point
      const float xv = static_cast<float>(i), yv = static_cast<float>(j - 0.5f);  // v point
      const float du = trio.distance(xu, yu), dv = trio.distance(xv, yv);         // BDIM.pde:165-172
      sf.delx[k] = delta0(eps, du);
      sf.dely[k] = delta0(eps, dv);
      sf.del1x[k] = delta1(eps, du);
      sf.del1y[k] = delta1(eps, dv);
      if (jabs(du) < eps || jabs(dv) < eps) sf.band_cells++;
      float w[3];
      trio.weights(xu, yu, w);  // BDIM.pde:146 -> BodyUnion.velocity
      sf.w1x[k] = w[1];
      sf.w2x[k] = w[2];
      sf.ry1x[k] = yu - ctl1.yc();  // PVector.sub, Body.pde:236-237
      sf.ry2x[k] = yu - ctl2.yc();
      trio.weights(xv, yv, w);
      sf.w1y[k] = w[1];
      sf.w2y[k] = w[2];
      sf.rx1y[k] = xv - ctl1.xc();
      sf.rx2y[k] = xv - ctl2.xc();
      if (sf.w1x[k] != 0 || sf.w2x[k] != 0 || sf.w1y[k] != 0 || sf.w2y[k] != 0) sf.ub_cells++;
      trio.wall_normal(xu, yu, sf.wnxx[k], sf.wnyx[k]);  // BDIM.pde:156-158: wnx.x, wny.x
      trio.wall_normal(xv, yv, sf.wnxy[k], sf.wnyy[k]);  // :159-161: wnx.y, wny.y
    }
  set_bc(sf.delx, n, m, 1, 1.f);  // del.setBC(), BDIM.pde:183 (bval 1, no gradientExit)
  set_bc(sf.dely, n, m, 2, 1.f);
  set_bc(sf.del1x, n, m, 1, 0.f);  // BDIM.pde:195
  set_bc(sf.del1y, n, m, 2, 0.f);

  for (int i = 1; i < n - 1; ++i)
    for (int j = 1; j < m - 1; ++j) {
      const int k = i * m + j;
      if (sf.delx[k] != 1.f || sf.del1x[k] != 0.f) sf.band_x.push_back(k);
      if (sf.dely[k] != 1.f || sf.del1y[k] != 0.f) sf.band_y.push_back(k);
    }

  // c = del * (rhoi*dt), BDIM.pde:81 (rhoi == 1 everywhere, BDIM.pde:61)
  sf.cx.resize(N);
  sf.cy.resize(N);
  for (size_t k = 0; k < N; ++k) {
    const float rdt = 1.f * sf.dt;
    sf.cx[k] = sf.delx[k] * rdt;
    sf.cy[k] = sf.dely[k] * rdt;
  }

  // multigrid hierarchy: level 0 = PoissonMatrix(c); then MG.restrict(A) while divisible
  MGLevel L0;
  L0.n = n;
  L0.m = m;
  L0.lx = sf.cx;
  L0.ly = sf.cy;
  fill_inverse(L0);
  sf.levels.push_back(std::move(L0));
  if (!divisible(n, m)) throw std::runtime_error("afc: grid is not coarsenable (MG.divisible)");
  while (true) {
    const MGLevel &F = sf.levels.back();
    MGLevel C;
    C.n = (F.n - 2) / 2 + 2;
    C.m = (F.m - 2) / 2 + 2;
    C.lx.assign(static_cast<size_t>(C.n) * C.m, 0.f);
    C.ly.assign(static_cast<size_t>(C.n) * C.m, 0.f);
    for (int i = 1; i < C.n - 1; ++i)
      for (int j = 1; j < C.m - 1; ++j) {
        const int ii = (i - 1) * 2 + 1, jj = (j - 1) * 2 + 1;
        const size_t kf = static_cast<size_t>(ii) * F.m + jj;
        C.lx[static_cast<size_t>(i) * C.m + j] = (F.lx[kf] + F.lx[kf + 1]) * 0.5f;
        C.ly[static_cast<size_t>(i) * C.m + j] = (F.ly[kf] + F.ly[kf + F.m]) * 0.5f;
      }
    set_bc(C.lx, C.n, C.m, 1, 0.f);  // lower.setBC(), MG.pde:120
    set_bc(C.ly, C.n, C.m, 2, 0.f);
    fill_inverse(C);
    const bool more = divisible(C.n, C.m);
    if (!more && C.n > 9 && C.m > 9)
      throw std::runtime_error("afc: MultiGrid requires 2^p * small sizes (MG.pde:101-104)");
    sf.levels.push_back(std::move(C));
    if (!more) break;
  }

  // tol = r.inner(r) with r == 1e-4 on the whole array (MG.pde:49-50): float products, double sum
  {
    double s = 0;
    const float t = 1e-4f;
    const float prod = t * t;
    for (int i = 1; i < n - 1; ++i)
      for (int j = 1; j < m - 1; ++j) s += prod;
    sf.mg_tol = static_cast<float>(s);
  }

  for (const Edge &e : main_c.faces()) {  // Body.pressForce, Body.pde:296-303
    sf.force_samples.push_back(make_sample(n, m, e.cx, e.cy));
    sf.seg_l.push_back(e.len);
    sf.seg_nx.push_back(e.nx);
    sf.seg_ny.push_back(e.ny);
  }
  return sf;
}

std::vector<PressSample> build_probes(const CaseParams &cp, int num) {  // SaveScalar.pde:28-41,61-72
  const float res = static_cast<float>(cp.resolution);
  const float Dp = res;
  const float nn = static_cast<float>(cp.x_lengths) * res, mm = static_cast<float>(cp.y_lengths) * res;
  const float cx = nn / 4, cy = mm / 2;
  const int n = cp.x_lengths * cp.resolution + 2, m = cp.y_lengths * cp.resolution + 2;
  std::vector<PressSample> out;
  for (int i = 0; i < num; ++i) {
    const float ang = static_cast<float>(i) / num * kPi * 2;
    const float xp = jcos(ang) * Dp / 2 + cx;
    const float yp = jsin(ang) * Dp / 2 + cy;
    out.push_back(make_sample(n, m, xp, yp));
  }
  return out;
}

}  // namespace afc
