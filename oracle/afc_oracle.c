/*
 * afc_oracle.c -- literal float32 CPU restatement of the reference's Lilypad 2-D BDIM
 * hot path (AFCCylinder.update2 and everything under it).
 *
 * TEST INFRASTRUCTURE ONLY (see afc_oracle.h).  Not part of the product path.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see Makefile).  Java `float`
 * arithmetic is IEEE binary32 without FMA; Processing's sin/cos/sqrt/mag evaluate
 * java.lang.Math in double on a float argument and narrow the result.
 *
 * Citations are file:line under /root/reference/clientLilypad/.
 */
#include "afc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------- Processing numeric helpers (PApplet) ---------- */
static const float PI_F = 3.14159265358979323846f;     /* PConstants.PI (float) */
static const float TWO_PI_F = 6.28318530717958647693f; /* PConstants.TWO_PI */

static inline float p_min(float a, float b) { return (a < b) ? a : b; }
static inline float p_max(float a, float b) { return (a > b) ? a : b; }
static inline float p_abs(float a) { return (a < 0) ? -a : a; }
static inline float p_sin(float a) { return (float)sin((double)a); }
static inline float p_cos(float a) { return (float)cos((double)a); }
static inline float p_sq(float a) { return a * a; }
static inline float p_mag2(float a, float b) {
  float s = a * a + b * b;
  return (float)sqrt((double)s);
}
/* PVector.mag(): sqrt(x*x+y*y+z*z) with z = 0 */
static inline float pv_mag(float x, float y) {
  float s = x * x + y * y + 0.0f * 0.0f;
  return (float)sqrt((double)s);
}

/* ---------- Field (Field.pde:24-58) ---------- */
typedef struct {
  int n, m, btype;
  float bval;
  int gradientExit;
  float *a; /* a[i*m+j] */
} Field;

#define A(f, i, j) ((f)->a[(size_t)(i) * (f)->m + (j)])

static Field *field_new(int n, int m, int btype, float bval) { /* Field.pde:30-37 */
  Field *f = (Field *)malloc(sizeof(Field));
  f->n = n;
  f->m = m;
  f->btype = btype;
  f->bval = bval;
  f->gradientExit = 0;
  f->a = (float *)malloc(sizeof(float) * (size_t)n * m);
  for (int k = 0; k < n * m; k++) f->a[k] = bval;
  return f;
}
static Field *field_copy(const Field *b) { /* Field.pde:51-58: flag NOT inherited */
  Field *f = (Field *)malloc(sizeof(Field));
  f->n = b->n;
  f->m = b->m;
  f->btype = b->btype;
  f->bval = b->bval;
  f->gradientExit = 0;
  f->a = (float *)malloc(sizeof(float) * (size_t)b->n * b->m);
  memcpy(f->a, b->a, sizeof(float) * (size_t)b->n * b->m);
  return f;
}
static void field_free(Field *f) {
  if (f) {
    free(f->a);
    free(f);
  }
}
static void field_eq(Field *f, const Field *b) { /* Field.pde:326-331 values only */
  memcpy(f->a, b->a, sizeof(float) * (size_t)f->n * f->m);
}
static void field_eqs(Field *f, float b) {
  for (int k = 0; k < f->n * f->m; k++) f->a[k] = b;
}

/* Field.setBC, Field.pde:209-234 -- statement order matters */
static void field_setBC(Field *f) {
  int n = f->n, m = f->m;
  float s = 0;
  for (int j = 0; j < m; j++) {
    A(f, 0, j) = A(f, 1, j);
    A(f, n - 1, j) = A(f, n - 2, j);
    if (f->btype == 1) {
      if (f->gradientExit) {
        A(f, 1, j) = f->bval;
        if (j > 0 && j < m - 1) s += A(f, n - 1, j);
      } else {
        A(f, 1, j) = f->bval;
        A(f, n - 1, j) = f->bval;
      }
    }
  }
  for (int i = 0; i < n; i++) {
    A(f, i, 0) = A(f, i, 1);
    A(f, i, m - 1) = A(f, i, m - 2);
    if (f->btype == 2) {
      A(f, i, 1) = f->bval;
      A(f, i, m - 1) = f->bval;
    }
  }
  if (f->gradientExit) {
    s /= (float)(m - 2);
    for (int j = 1; j < m - 1; j++) A(f, n - 1, j) += f->bval - s;
  }
}

/* elementwise ops, Field.pde:245-301 (whole array, ghosts included) */
static Field *field_times_s(const Field *f, float b) {
  Field *c = field_copy(f);
  for (int k = 0; k < c->n * c->m; k++) c->a[k] *= b;
  return c;
}
static Field *field_times(const Field *f, const Field *b) {
  Field *c = field_copy(f);
  for (int k = 0; k < c->n * c->m; k++) c->a[k] *= b->a[k];
  return c;
}
static Field *field_plus_s(const Field *f, float b) {
  Field *c = field_copy(f);
  for (int k = 0; k < c->n * c->m; k++) c->a[k] += b;
  return c;
}
static Field *field_plus(const Field *f, const Field *b) {
  Field *c = field_copy(f);
  for (int k = 0; k < c->n * c->m; k++) c->a[k] += b->a[k];
  return c;
}
static Field *field_minus(const Field *f, const Field *b) {
  Field *c = field_copy(f);
  for (int k = 0; k < c->n * c->m; k++) c->a[k] -= b->a[k];
  return c;
}
static void field_timesEq_s(Field *f, float b) {
  Field *c = field_times_s(f, b);
  field_eq(f, c);
  field_free(c);
}
static void field_plusEq(Field *f, const Field *b) {
  Field *c = field_plus(f, b);
  field_eq(f, c);
  field_free(c);
}
static void field_plusEq_s(Field *f, float b) {
  Field *c = field_plus_s(f, b);
  field_eq(f, c);
  field_free(c);
}
static void field_minusEq(Field *f, const Field *b) {
  Field *c = field_minus(f, b);
  field_eq(f, c);
  field_free(c);
}
/* Field.inner, Field.pde:302-310: float products, double accumulator */
static float field_inner(const Field *f, const Field *b) {
  double s = 0;
  for (int i = 1; i < f->n - 1; i++)
    for (int j = 1; j < f->m - 1; j++) {
      float prod = A(f, i, j) * A(b, i, j);
      s += prod;
    }
  return (float)s;
}
/* Field.sum, Field.pde:311-318: sequential float */
static float field_sum(const Field *f) {
  float s = 0;
  for (int i = 1; i < f->n - 1; i++)
    for (int j = 1; j < f->m - 1; j++) s += A(f, i, j);
  return s;
}
/* Field.L_inf, Field.pde:340-347 */
static float field_Linf(const Field *f) {
  float mx = 0;
  for (int k = 0; k < f->n * f->m; k++) mx = p_max(mx, p_abs(f->a[k]));
  return mx;
}
/* Field.linear, Field.pde:175-190 */
static float field_linear(const Field *f, float x0, float y0) {
  int n = f->n, m = f->m;
  float x = p_min(p_max(0.5f, x0), n - 1.5f);
  if (f->btype == 1 || f->btype == 3) x += 0.5f;
  int i = (int)x;
  if (i > n - 2) i = n - 2;
  float s = x - i;
  float y = p_min(p_max(0.5f, y0), m - 1.5f);
  if (f->btype == 2 || f->btype == 3) y += 0.5f;
  int j = (int)y;
  if (j > m - 2) j = m - 2;
  float t = y - j;
  if (s == 0 && t == 0) {
    return A(f, i, j);
  } else {
    return s * (t * A(f, i + 1, j + 1) + (1 - t) * A(f, i + 1, j)) +
           (1 - s) * (t * A(f, i, j + 1) + (1 - t) * A(f, i, j));
  }
}

/* ---------- VectorField (VectorField.pde:22-44) ---------- */
typedef struct {
  Field *x, *y;
  int n, m;
} VField;

static VField *vf_new(int n, int m, float xval, float yval) { /* :27-32 */
  VField *v = (VField *)malloc(sizeof(VField));
  v->n = n;
  v->m = m;
  v->x = field_new(n, m, 1, xval);
  v->y = field_new(n, m, 2, yval);
  return v;
}
static VField *vf_copy(const VField *b) { /* :33-39 */
  VField *v = (VField *)malloc(sizeof(VField));
  v->n = b->n;
  v->m = b->m;
  v->x = field_copy(b->x);
  v->y = field_copy(b->y);
  return v;
}
static void vf_free(VField *v) {
  if (v) {
    field_free(v->x);
    field_free(v->y);
    free(v);
  }
}
static void vf_setBC(VField *v) {
  field_setBC(v->x);
  field_setBC(v->y);
}
static void vf_eq(VField *v, const VField *b) {
  field_eq(v->x, b->x);
  field_eq(v->y, b->y);
}
static void vf_eqs(VField *v, float b) {
  field_eqs(v->x, b);
  field_eqs(v->y, b);
}
static VField *vf_times(const VField *a, const VField *b) { /* :237-241 */
  VField *g = vf_copy(a);
  Field *t = field_times(g->x, b->x);
  field_eq(g->x, t);
  field_free(t);
  t = field_times(g->y, b->y);
  field_eq(g->y, t);
  field_free(t);
  return g;
}
static VField *vf_times_s(const VField *a, float b) { /* :243-247 */
  VField *g = vf_copy(a);
  field_timesEq_s(g->x, b);
  field_timesEq_s(g->y, b);
  return g;
}
static VField *vf_plus_s(const VField *a, float b) { /* :261-265 */
  VField *g = vf_copy(a);
  field_plusEq_s(g->x, b);
  field_plusEq_s(g->y, b);
  return g;
}
static VField *vf_minus(const VField *a, const VField *b) { /* :255-259 */
  VField *g = vf_copy(a);
  field_minusEq(g->x, b->x);
  field_minusEq(g->y, b->y);
  return g;
}
static void vf_plusEq(VField *a, const VField *b) {
  field_plusEq(a->x, b->x);
  field_plusEq(a->y, b->y);
}
static void vf_timesEq_s(VField *a, float b) {
  field_timesEq_s(a->x, b);
  field_timesEq_s(a->y, b);
}

/* VectorField.normalGrad, VectorField.pde:46-54 */
static VField *vf_normalGrad(const VField *u, const VField *wnx, const VField *wny) {
  int n = u->n, m = u->m;
  VField *g = vf_new(n, m, 0, 0);
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++) {
      A(g->x, i, j) = 0.5f * (A(wnx->x, i, j) * (A(u->x, i + 1, j) - A(u->x, i - 1, j)) +
                              A(wny->x, i, j) * (A(u->x, i, j + 1) - A(u->x, i, j - 1)));
      A(g->y, i, j) = 0.5f * (A(wnx->y, i, j) * (A(u->y, i + 1, j) - A(u->y, i - 1, j)) +
                              A(wny->y, i, j) * (A(u->y, i, j + 1) - A(u->y, i, j - 1)));
    }
  return g;
}
/* VectorField.divergence, VectorField.pde:56-65 */
static Field *vf_divergence(const VField *u) {
  int n = u->n, m = u->m;
  Field *d = field_new(n, m, 0, 0);
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++)
      A(d, i, j) = A(u->x, i + 1, j) - A(u->x, i, j) + A(u->y, i, j + 1) - A(u->y, i, j);
  return d;
}
/* VectorField.curl, VectorField.pde:78-87 */
static Field *vf_curl(const VField *u) {
  int n = u->n, m = u->m;
  Field *d = field_new(n, m, 3, 0);
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++)
      A(d, i, j) = A(u->x, i, j - 1) - A(u->x, i, j) + A(u->y, i, j) - A(u->y, i - 1, j);
  return d;
}
/* Field.gradient, Field.pde:71-81 */
static VField *field_gradient(const Field *p) {
  int n = p->n, m = p->m;
  VField *g = vf_new(n, m, 0, 0);
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++) {
      A(g->x, i, j) = A(p, i, j) - A(p, i - 1, j);
      A(g->y, i, j) = A(p, i, j) - A(p, i, j - 1);
    }
  vf_setBC(g);
  return g;
}

/* QUICK, VectorField.pde:202-223 */
static const float CF = 1.f / 6.f, S_Q = 10.f;
static inline float med(float a, float b, float c) { /* :221-223 */
  return p_max(p_min(a, b), p_min(p_max(a, b), c));
}
static float bho(const VField *u, const Field *b, int i, int j, int d1, int d2, float uf) {
  int n = u->n, m = u->m;
  float bf = 0.5f * (A(b, i + d1, j + d2) + A(b, i, j));
  if (d1 * uf < 0) {
    i += d1;
    d1 = -d1;
  }
  if (d2 * uf < 0) {
    j += d2;
    d2 = -d2;
  }
  if (i > n - 2 || i < 2 || j > m - 2 || j < 2) return bf;
  float bc = A(b, i, j);
  float bd = A(b, i + d1, j + d2);
  float bu = A(b, i - d1, j - d2);
  bf -= CF * (bd - 2 * bc + bu);
  float b1 = bu + S_Q * (bc - bu);
  return med(bf, bc, med(bc, bd, b1));
}
/* VectorField.advection, :181-196 */
static float advection(const VField *u, const Field *b, int i, int j) {
  const Field *x = u->x, *y = u->y;
  float uo, ue, vs, vn;
  if (b->btype == 1) {
    uo = 0.5f * (A(x, i - 1, j) + A(x, i, j));
    ue = 0.5f * (A(x, i + 1, j) + A(x, i, j));
    vs = 0.5f * (A(y, i, j) + A(y, i - 1, j));
    vn = 0.5f * (A(y, i, j + 1) + A(y, i - 1, j + 1));
  } else {
    uo = 0.5f * (A(x, i, j - 1) + A(x, i, j));
    ue = 0.5f * (A(x, i + 1, j - 1) + A(x, i + 1, j));
    vs = 0.5f * (A(y, i, j - 1) + A(y, i, j));
    vn = 0.5f * (A(y, i, j) + A(y, i, j + 1));
  }
  return ((uo * bho(u, b, i, j, -1, 0, uo) - ue * bho(u, b, i, j, 1, 0, ue)) +
          (vs * bho(u, b, i, j, 0, -1, vs) - vn * bho(u, b, i, j, 0, 1, vn)));
}
/* VectorField.diffusion, :198-200 */
static float diffusion(const Field *b, int i, int j) {
  return A(b, i + 1, j) + A(b, i, j + 1) - 4 * A(b, i, j) + A(b, i - 1, j) + A(b, i, j - 1);
}
/* VectorField.AdvDif, :170-179 */
static void vf_AdvDif(VField *u, const VField *u0, float dt, float nu) {
  int n = u->n, m = u->m;
  VField *v = vf_copy(u);
  for (int j = 1; j < m - 1; j++)
    for (int i = 1; i < n - 1; i++) {
      A(v->x, i, j) = (advection(u, u->x, i, j) + nu * diffusion(u->x, i, j)) * dt + A(u0->x, i, j);
      A(v->y, i, j) = (advection(u, u->y, i, j) + nu * diffusion(u->y, i, j)) * dt + A(u0->y, i, j);
    }
  vf_eq(u, v);
  vf_free(v);
}

/* ---------- PoissonMatrix (PoissonMatrix.pde:33-69) ---------- */
typedef struct {
  int n, m;
  VField *lower;
  Field *diagonal, *inv;
} PMatrix;

static PMatrix *pm_new(const VField *lower) { /* :38-51 */
  PMatrix *A_ = (PMatrix *)malloc(sizeof(PMatrix));
  int n = lower->n, m = lower->m;
  A_->n = n;
  A_->m = m;
  A_->lower = vf_copy(lower);
  A_->diagonal = field_new(n, m, 0, 0);
  A_->inv = field_new(n, m, 0, 1);
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++) {
      float sumd = A(lower->x, i, j) + A(lower->x, i + 1, j) + A(lower->y, i, j) + A(lower->y, i, j + 1);
      A(A_->diagonal, i, j) = -sumd;
      if (sumd > 1e-5f) A(A_->inv, i, j) = -1.f / sumd;
    }
  return A_;
}
static void pm_free(PMatrix *A_) {
  if (A_) {
    vf_free(A_->lower);
    field_free(A_->diagonal);
    field_free(A_->inv);
    free(A_);
  }
}
static Field *pm_times(const PMatrix *A_, const Field *x) { /* :53-64 */
  int n = A_->n, m = A_->m;
  Field *ab = field_new(n, m, 0, 0);
  const Field *lx = A_->lower->x, *ly = A_->lower->y;
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++)
      A(ab, i, j) = A(x, i, j) * A(A_->diagonal, i, j) + A(x, i - 1, j) * A(lx, i, j) +
                    A(x, i + 1, j) * A(lx, i + 1, j) + A(x, i, j - 1) * A(ly, i, j) +
                    A(x, i, j + 1) * A(ly, i, j + 1);
  return ab;
}
static Field *pm_residual(const PMatrix *A_, const Field *b, const Field *x) { /* :66-68 */
  Field *ax = pm_times(A_, x);
  Field *r = field_minus(b, ax);
  field_free(ax);
  return r;
}

/* ---------- MG (MG.pde:40-153) ---------- */
typedef struct {
  PMatrix *A_; /* owned iff own_A */
  int own_A;
  Field *r, *x, *d; /* x owned iff own_x */
  int own_x;
  int iter, level, its;
  float tol;
} MG;

static MG *mg_new_top(PMatrix *A_, Field *x, const Field *b) { /* MG.pde:46-52 */
  MG *s = (MG *)calloc(1, sizeof(MG));
  s->A_ = A_;
  s->own_A = 0;
  s->x = x;
  s->own_x = 0;
  s->its = 4;
  s->tol = 1e-4f;
  Field *r = field_new(x->n, x->m, 0, s->tol);
  s->tol = field_inner(r, r);
  field_free(r);
  s->r = pm_residual(A_, b, x);
  s->d = NULL;
  return s;
}
static MG *mg_new_coarse(PMatrix *A_, Field *r, int level, int iter) { /* :53-59 */
  MG *s = (MG *)calloc(1, sizeof(MG));
  s->A_ = A_;
  s->own_A = 1;
  s->r = r;
  s->x = field_new(r->n, r->m, 0, 0);
  s->own_x = 1;
  s->level = level;
  s->iter = iter;
  s->its = 4;
  s->tol = 1e-4f;
  return s;
}
static void mg_free(MG *s) {
  if (!s) return;
  if (s->own_A) pm_free(s->A_);
  field_free(s->r);
  if (s->own_x) field_free(s->x);
  field_free(s->d);
  free(s);
}
static void mg_increment(MG *s) { /* :94-97 */
  field_plusEq(s->x, s->d);
  Field *ad = pm_times(s->A_, s->d);
  field_minusEq(s->r, ad);
  field_free(ad);
}
static void mg_smooth(MG *s, int itmx) { /* :79-92 */
  field_free(s->d);
  s->d = field_times(s->r, s->A_->inv);
  Field *d = s->d;
  const Field *lx = s->A_->lower->x, *ly = s->A_->lower->y, *inv = s->A_->inv, *r = s->r;
  for (int it = 0; it < itmx; it++)
    for (int i = 1; i < r->n - 1; i++)
      for (int j = 1; j < r->m - 1; j++)
        A(d, i, j) = -(A(d, i - 1, j) * A(lx, i, j) + A(d, i + 1, j) * A(lx, i + 1, j) +
                       A(d, i, j - 1) * A(ly, i, j) + A(d, i, j + 1) * A(ly, i, j + 1) - A(r, i, j)) *
                     A(inv, i, j);
  field_setBC(d);
  mg_increment(s);
}
static int mg_divisible(const MG *s) { /* :99-106 */
  return (s->x->n - 2) % 2 == 0 && (s->x->m - 2) % 2 == 0 && s->x->n > 4 && s->x->m > 4;
}
static PMatrix *mg_restrictA(const PMatrix *A_) { /* :108-122 */
  int n = (A_->lower->n - 2) / 2 + 2;
  int m = (A_->lower->m - 2) / 2 + 2;
  VField *lower = vf_new(n, m, 0, 0);
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++) {
      int ii = (i - 1) * 2 + 1;
      int jj = (j - 1) * 2 + 1;
      A(lower->x, i, j) = (A(A_->lower->x, ii, jj) + A(A_->lower->x, ii, jj + 1)) * 0.5f;
      A(lower->y, i, j) = (A(A_->lower->y, ii, jj) + A(A_->lower->y, ii + 1, jj)) * 0.5f;
    }
  vf_setBC(lower);
  PMatrix *c = pm_new(lower);
  vf_free(lower);
  return c;
}
static Field *mg_restrictF(const Field *a) { /* :124-137 */
  int n = (a->n - 2) / 2 + 2;
  int m = (a->m - 2) / 2 + 2;
  Field *b = field_new(n, m, 0, 0);
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++) {
      int ii = (i - 1) * 2 + 1;
      int jj = (j - 1) * 2 + 1;
      A(b, i, j) = A(a, ii, jj) + A(a, ii, jj + 1) + A(a, ii + 1, jj) + A(a, ii + 1, jj + 1);
    }
  field_setBC(b);
  return b;
}
static Field *mg_prolongate(const Field *a) { /* :139-152, Java truncating division */
  int n = (a->n - 2) * 2 + 2;
  int m = (a->m - 2) * 2 + 2;
  Field *b = field_new(n, m, 0, 0);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < m; j++) {
      int ii = (i - 1) / 2 + 1; /* C99 '/' truncates toward zero like Java */
      int jj = (j - 1) / 2 + 1;
      A(b, i, j) = A(a, ii, jj);
    }
  field_setBC(b);
  return b;
}
static void mg_vCycle(MG *s) { /* :68-77 */
  mg_smooth(s, 0);
  MG *coarse = mg_new_coarse(mg_restrictA(s->A_), mg_restrictF(s->r), s->level + 1, s->iter);
  if (mg_divisible(coarse)) mg_vCycle(coarse);
  mg_smooth(coarse, s->its);
  field_free(s->d);
  s->d = mg_prolongate(coarse->x);
  mg_increment(s);
  mg_free(coarse);
}
static void mg_update(MG *s) { /* :61-66 */
  s->iter++;
  mg_vCycle(s);
  mg_smooth(s, s->its);
}
/* MGsolver, MG.pde:30-38 (x updated in place and returned) */
static int MGsolver(float itmx, PMatrix *A_, Field *x, const Field *b, float *linf) {
  MG *solver = mg_new_top(A_, x, b);
  while (solver->iter < itmx) {
    mg_update(solver);
    if (field_inner(solver->r, solver->r) < solver->tol) break;
  }
  if (linf) *linf = field_Linf(solver->r);
  int it = solver->iter;
  mg_free(solver);
  return it;
}

/* VectorField.project, VectorField.pde:130-143 */
static int vf_project(VField *u, const VField *coeffs, Field *p, float *linf) {
  int n = u->n, m = u->m;
  Field *s = vf_divergence(u);
  PMatrix *A_ = pm_new(coeffs);
  int it = MGsolver(20, A_, p, s, linf);
  pm_free(A_);
  field_free(s);
  field_plusEq_s(p, -1 * field_sum(p) / (float)((n - 2) * (m - 2)));
  VField *dp = field_gradient(p);
  Field *t1 = field_times_s(dp->x, -1);
  Field *t2 = field_times(coeffs->x, t1);
  field_plusEq(u->x, t2);
  field_free(t1);
  field_free(t2);
  t1 = field_times_s(dp->y, -1);
  t2 = field_times(coeffs->y, t1);
  field_plusEq(u->y, t2);
  field_free(t1);
  field_free(t2);
  vf_free(dp);
  vf_setBC(u);
  return it;
}

/* ---------- OrthoNormal (OrthoNormal.pde:8-28) ---------- */
typedef struct {
  float l, nx, ny, tx, ty, off, t1, t2, cenx, ceny;
} Ortho;

static void ortho_init(Ortho *o, float x1x, float x1y, float x2x, float x2y) { /* :8-18 */
  o->l = pv_mag(x1x - x2x, x1y - x2y);
  o->tx = (x2x - x1x) / o->l;
  o->ty = (x2y - x1y) / o->l;
  o->t1 = x1x * o->tx + x1y * o->ty;
  o->t2 = x2x * o->tx + x2y * o->ty;
  o->nx = -o->ty;
  o->ny = o->tx;
  o->off = x1x * o->nx + x1y * o->ny;
  o->cenx = (x1x + x2x) / 2.f;
  o->ceny = (x1y + x2y) / 2.f;
}
static float ortho_distance(const Ortho *o, float x, float y, int projected) { /* :20-28 */
  float d = x * o->nx + y * o->ny - o->off;
  if (projected) return d;
  float d1 = x * o->tx + y * o->ty - o->t1;
  float d2 = x * o->tx + y * o->ty - o->t2;
  return p_abs(d) + p_max(0, -d1) + p_max(0, d2);
}

/* ---------- Body / EllipseBody / CircleBody (Body.pde) ---------- */
typedef struct Body {
  float xcx, xcy;
  int n; /* number of coords */
  float *cx, *cy;
  int north;
  Ortho *orth;
  int convex;
  struct Body *box;
  float dphi, dxcx, dxcy, phi;
  float h;       /* CircleBody diameter */
  int is_circle; /* CircleBody.distance override, Body.pde:408-410 */
} Body;

static Body *body_new(float x, float y) { /* Body.pde:57-60 */
  Body *b = (Body *)calloc(1, sizeof(Body));
  b->xcx = x;
  b->xcy = y;
  return b;
}
static void body_add(Body *b, float x, float y) { /* :62-64 */
  b->cx = (float *)realloc(b->cx, sizeof(float) * (b->n + 1));
  b->cy = (float *)realloc(b->cy, sizeof(float) * (b->n + 1));
  b->cx[b->n] = x;
  b->cy[b->n] = y;
  b->n++;
}
static void body_free(Body *b) {
  if (!b) return;
  if (b->box) body_free(b->box);
  free(b->cx);
  free(b->cy);
  free(b->orth);
  free(b);
}
static void body_end(Body *b) { /* Body.pde:66-97 (closed) */
  int n = b->n;
  b->north = n;
  b->orth = (Ortho *)malloc(sizeof(Ortho) * n);
  for (int i = 0; i < n; i++) /* getOrth :102-108 */
    ortho_init(&b->orth[i], b->cx[i], b->cy[i], b->cx[(i + 1) % n], b->cy[(i + 1) % n]);
  if (n > 4) { /* bounding box :73-87 */
    float mnx = b->xcx, mny = b->xcy, mxx = b->xcx, mxy = b->xcy;
    for (int i = 0; i < n; i++) {
      mnx = p_min(mnx, b->cx[i]);
      mny = p_min(mny, b->cy[i]);
      mxx = p_max(mxx, b->cx[i]);
      mxy = p_max(mxy, b->cy[i]);
    }
    b->box = body_new(b->xcx, b->xcy);
    body_add(b->box, mnx, mny);
    body_add(b->box, mnx, mxy);
    body_add(b->box, mxx, mxy);
    body_add(b->box, mxx, mny);
    body_end(b->box);
  }
  b->convex = 1; /* :90-96 */
  for (int i = 0; i < b->north && b->convex; i++)
    for (int j = 0; j < b->north; j++)
      if (ortho_distance(&b->orth[i], b->orth[j].cenx, b->orth[j].ceny, 1) > 0.001f) {
        b->convex = 0;
        break;
      }
}
static int body_wn(const Body *b, float x, float y) { /* :199-213 */
  int wn = 0;
  for (int i = 0; i < b->n - 1; i++) {
    float yi = b->cy[i], yi1 = b->cy[i + 1];
    const Ortho *o = &b->orth[i];
    if (yi <= y) {
      if (yi1 > y && ortho_distance(o, x, y, 1) > 0) wn++;
    } else {
      if (yi1 <= y && ortho_distance(o, x, y, 1) < 0) wn--;
    }
  }
  return wn;
}
static float body_distance_poly(const Body *b, float x, float y) { /* :174-192 */
  float dis;
  if (b->n > 4) {
    dis = body_distance_poly(b->box, x, y);
    if (dis > 3) return dis;
  }
  if (b->convex) {
    dis = -1e10f;
    for (int i = 0; i < b->north; i++) dis = p_max(dis, ortho_distance(&b->orth[i], x, y, 1));
    return dis;
  } else {
    dis = 1e10f;
    for (int i = 0; i < b->north; i++) dis = p_min(dis, ortho_distance(&b->orth[i], x, y, 0));
    return (body_wn(b, x, y) == 0) ? dis : -dis;
  }
}
static float body_distance(const Body *b, float x, float y) {
  if (b->is_circle) return p_mag2(x - b->xcx, y - b->xcy) - 0.5f * b->h; /* :408-410 */
  return body_distance_poly(b, x, y);
}
static void body_WallNormal(const Body *b, float x, float y, float *wx, float *wy) { /* :215-232 */
  *wx = 0;
  *wy = 0;
  float dis = -1e10f, dis2 = -1e10f;
  if (b->n > 4) {
    if (body_distance_poly(b->box, x, y) > 3) return;
  }
  for (int i = 0; i < b->north; i++) {
    dis2 = ortho_distance(&b->orth[i], x, y, 1);
    if (dis2 > dis) {
      dis = dis2;
      *wx = b->orth[i].nx;
      *wy = b->orth[i].ny;
    }
  }
}
static float body_velocity(const Body *b, int d, float dt, float x, float y) { /* :234-240 */
  float rx = x - b->xcx, ry = y - b->xcy;
  if (d == 1)
    return (b->dxcx - ry * b->dphi) / dt;
  else
    return (b->dxcy + rx * b->dphi) / dt;
}
static int body_unsteady(const Body *b) { /* :242 */
  return (pv_mag(b->dxcx, b->dxcy) != 0) | (b->dphi != 0);
}
static void body_pressForce(const Body *b, const Field *p, float *fx, float *fy) { /* :296-303 */
  float px = 0, py = 0;
  for (int i = 0; i < b->north; i++) {
    const Ortho *o = &b->orth[i];
    float pdl = field_linear(p, o->cenx, o->ceny) * o->l;
    px += pdl * o->nx;
    py += pdl * o->ny;
  }
  *fx = px;
  *fy = py;
}
/* EllipseBody ctor with a=1 (CircleBody), Body.pde:386-406 */
static Body *circle_new(float x, float y, float d) {
  Body *b = body_new(x, y);
  b->h = d;
  b->is_circle = 1;
  float a = 1.f / 1.0f;
  float dx = 0.5f * b->h * a, dy = 0.5f * b->h;
  int m = 40;
  for (int i = 0; i < m; i++) {
    float theta = -TWO_PI_F * i / ((float)m);
    body_add(b, b->xcx + dx * p_cos(theta), b->xcy + dy * p_sin(theta));
  }
  body_end(b);
  return b;
}

/* ---------- BodyUnion (BodyUnion.pde) ---------- */
#define NBODY 3
static float union_delta0(float d) { /* BodyUnion.pde:158-166 */
  if (d <= -1) {
    return 0;
  } else if (d >= 1) {
    return 1;
  } else {
    return 0.5f * (1.f + d + p_sin(PI_F * d) / PI_F);
  }
}
static void union_weights(Body *const *bl, float x, float y, float *w) { /* :145-156 */
  float s = 0;
  for (int i = 0; i < NBODY; i++) {
    float d = body_distance(bl[i], x, y);
    w[i] = union_delta0(-d / 3.f);
    s += w[i];
  }
  for (int i = 0; i < NBODY && s > 0; i++) w[i] /= s;
}
static float union_distance(Body *const *bl, float x, float y) { /* :67-72 */
  float d = 1e6f;
  for (int i = 0; i < NBODY; i++) d = p_min(d, body_distance(bl[i], x, y));
  return d;
}
static void union_WallNormal(Body *const *bl, float x, float y, float *mx, float *my) { /* :74-82 */
  float w[NBODY];
  union_weights(bl, x, y, w);
  float ax = 0, ay = 0;
  for (int i = 0; i < NBODY; i++) {
    float nx, ny;
    body_WallNormal(bl[i], x, y, &nx, &ny);
    nx *= w[i]; /* PVector.mult */
    ny *= w[i];
    ax += nx; /* PVector.add */
    ay += ny;
  }
  *mx = ax;
  *my = ay;
}
static float union_velocity(Body *const *bl, int d, float dt, float x, float y) { /* :84-92 */
  float w[NBODY];
  union_weights(bl, x, y, w);
  float v = 0;
  for (int i = 0; i < NBODY; i++) {
    float u = body_velocity(bl[i], d, dt, x, y);
    v += u * w[i];
  }
  return v;
}
static int union_unsteady(Body *const *bl) { /* :116-122 */
  int un = 0;
  for (int i = 0; i < NBODY; i++) un = un | body_unsteady(bl[i]);
  return un;
}

/* ---------- BDIM (BDIM.pde) ---------- */
struct Oracle {
  /* AFCCylinder */
  int resolution, n_, m_;
  float dt, t, D, dR, gR, theta;
  Body *bodies[NBODY];
  float force[2];
  /* BDIM */
  int n, m;
  float flow_t, flow_dt, nu, eps;
  VField *u, *del, *del1, *c, *u0, *ub, *wnx, *wny, *distance, *rhoi;
  Field *p;
  int last_iters[2];
  float last_resid[2];
};

static float bdim_delta0(float eps, float d) { /* BDIM.pde:199-207 */
  if (d <= -eps) {
    return 0;
  } else if (d >= eps) {
    return 1;
  } else {
    return 0.5f * (1.f + d / eps + p_sin(PI_F * d / eps) / PI_F);
  }
}
static float bdim_delta1(float eps, float d) { /* BDIM.pde:209-215 */
  if (p_abs(d) >= eps) {
    return 0;
  } else {
    return 0.25f * (eps - p_sq(d) / eps) -
           1 / TWO_PI_F * (d * p_sin(d * PI_F / eps) + eps / PI_F * (1 + p_cos(d * PI_F / eps)));
  }
}
static void bdim_get_coeffs(Oracle *o) { /* BDIM.pde:132-196 */
  int n = o->n, m = o->m;
  Body *const *bl = o->bodies;
  /* get_dist :165-172 */
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++) {
      A(o->distance->x, i, j) = union_distance(bl, (float)(i - 0.5f), j);
      A(o->distance->y, i, j) = union_distance(bl, i, (float)(j - 0.5f));
    }
  /* get_del :174-184 */
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++) {
      A(o->del->x, i, j) = bdim_delta0(o->eps, A(o->distance->x, i, j));
      A(o->del->y, i, j) = bdim_delta0(o->eps, A(o->distance->y, i, j));
    }
  vf_setBC(o->del);
  /* get_del1 :186-196 */
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++) {
      A(o->del1->x, i, j) = bdim_delta1(o->eps, A(o->distance->x, i, j));
      A(o->del1->y, i, j) = bdim_delta1(o->eps, A(o->distance->y, i, j));
    }
  vf_setBC(o->del1);
  /* get_ub :140-149 */
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++) {
      A(o->ub->x, i, j) = union_velocity(bl, 1, o->flow_dt, (float)(i - 0.5f), j);
      A(o->ub->y, i, j) = union_velocity(bl, 2, o->flow_dt, i, (float)(j - 0.5f));
    }
  /* get_wn :151-163 */
  for (int i = 1; i < n - 1; i++)
    for (int j = 1; j < m - 1; j++) {
      float wx, wy;
      union_WallNormal(bl, (float)(i - 0.5f), j, &wx, &wy);
      A(o->wnx->x, i, j) = wx;
      A(o->wny->x, i, j) = wy;
      union_WallNormal(bl, i, (float)(j - 0.5f), &wx, &wy);
      A(o->wnx->y, i, j) = wx;
      A(o->wny->y, i, j) = wy;
    }
}
/* BDIM.updateUP, BDIM.pde:109-124 (g = 0 so R.plusEq(g*dt) adds +0) */
static void bdim_updateUP(Oracle *o, VField *R, const VField *coeff, int which) {
  field_plusEq_s(R->x, 0.f * o->flow_dt); /* PVector.mult(g,dt) with g=(0,0) */
  field_plusEq_s(R->y, 0.f * o->flow_dt);
  VField *du = vf_minus(R, o->ub); /* :124 evaluated by the caller before the call */
  /* note: Java evaluates R.minus(ub) BEFORE entering updateUP, i.e. before R.plusEq(g*dt);
   * with g=0 the two orders give identical values (x + 0 == x). */
  VField *dm1 = vf_plus_s(o->del, -1);
  VField *t1 = vf_times(o->ub, dm1);
  VField *t2 = vf_times(o->del, R);
  VField *unew = vf_minus(t2, t1);
  vf_eq(o->u, unew);
  vf_free(dm1);
  vf_free(t1);
  vf_free(t2);
  vf_free(unew);
  VField *g = vf_normalGrad(du, o->wnx, o->wny); /* mu1 = true :120 */
  VField *t3 = vf_times(o->del1, g);
  vf_plusEq(o->u, t3);
  vf_free(g);
  vf_free(t3);
  vf_free(du);
  vf_setBC(o->u);
  o->last_iters[which] = vf_project(o->u, coeff, o->p, &o->last_resid[which]);
}
static void bdim_update(Oracle *o) { /* BDIM.pde:79-87 */
  VField *rd = vf_times_s(o->rhoi, o->flow_dt);
  VField *cc = vf_times(o->del, rd);
  vf_eq(o->c, cc);
  vf_free(rd);
  vf_free(cc);
  vf_eq(o->u0, o->u);
  VField *F = vf_copy(o->u);
  vf_AdvDif(F, o->u0, o->flow_dt, o->nu);
  bdim_updateUP(o, F, o->c, 0);
  vf_free(F);
}
static void bdim_update2(Oracle *o) { /* BDIM.pde:89-107, QUICK branch */
  VField *us = vf_copy(o->u), *F = vf_copy(o->u);
  vf_AdvDif(F, o->u0, o->flow_dt, o->nu);
  bdim_updateUP(o, F, o->c, 1);
  vf_plusEq(o->u, us);
  vf_timesEq_s(o->u, 0.5f);
  o->flow_t += o->flow_dt;
  vf_free(us);
  vf_free(F);
}

/* ---------- public API ---------- */
Oracle *orc_create(int resolution, int Re, float dR, float gR, float theta, float dtReal,
                   int xLengths, int yLengths) { /* AFCCylinder.pde:11-42 + BDIM.pde:41-71 */
  Oracle *o = (Oracle *)calloc(1, sizeof(Oracle));
  o->n_ = xLengths * resolution;
  o->m_ = yLengths * resolution;
  o->resolution = resolution;
  o->dR = dR;
  o->gR = gR;
  o->theta = theta;
  o->dt = dtReal * resolution;
  o->D = resolution;
  float D = o->D;
  float r = (D / 2 + gR * D + dR * D / 2);
  int n_ = o->n_, m_ = o->m_;
  o->bodies[0] = circle_new(n_ / 4, m_ / 2, D);
  o->bodies[1] = circle_new(n_ / 4 + r * p_cos(theta), m_ / 2 - r * p_sin(theta), dR * D);
  o->bodies[2] = circle_new(n_ / 4 + r * p_cos(theta), m_ / 2 + r * p_sin(theta), dR * D);
  /* BDIM(n,m,dt,body,(float)D/Re,QUICK) */
  o->n = n_ + 2;
  o->m = m_ + 2;
  int n = o->n, m = o->m;
  o->flow_dt = o->dt;
  o->nu = (float)D / Re;
  o->eps = 2.0f;
  o->u = vf_new(n, m, 1, 0);
  o->u->x->gradientExit = 1; /* BDIM.pde:52 */
  o->u0 = vf_new(n, m, 0, 0);
  o->p = field_new(n, m, 0, 0);
  o->ub = vf_new(n, m, 0, 0);
  o->distance = vf_new(n, m, 10, 10);
  o->del = vf_new(n, m, 1, 1);
  o->del1 = vf_new(n, m, 0, 0);
  o->rhoi = vf_copy(o->del);
  o->c = vf_copy(o->del);
  o->wnx = vf_new(n, m, 0, 0);
  o->wny = vf_new(n, m, 0, 0);
  bdim_get_coeffs(o);
  return o;
}
void orc_destroy(Oracle *o) {
  if (!o) return;
  for (int i = 0; i < NBODY; i++) body_free(o->bodies[i]);
  vf_free(o->u);
  vf_free(o->u0);
  vf_free(o->ub);
  vf_free(o->distance);
  vf_free(o->del);
  vf_free(o->del1);
  vf_free(o->rhoi);
  vf_free(o->c);
  vf_free(o->wnx);
  vf_free(o->wny);
  field_free(o->p);
  free(o);
}
int orc_n(const Oracle *o) { return o->n; }
int orc_m(const Oracle *o) { return o->m; }
float orc_dt(const Oracle *o) { return o->dt; }
float orc_nu(const Oracle *o) { return o->nu; }
float orc_t(const Oracle *o) { return o->t; }
float orc_flow_t(const Oracle *o) { return o->flow_t; }
void orc_set_t(Oracle *o, float t) { o->t = t; }

void orc_set_state(Oracle *o, const float *u, const float *v, const float *p) {
  size_t N = (size_t)o->n * o->m;
  memcpy(o->u->x->a, u, N * sizeof(float));
  memcpy(o->u->y->a, v, N * sizeof(float));
  memcpy(o->p->a, p, N * sizeof(float));
}
void orc_get_state(const Oracle *o, float *u, float *v, float *p) {
  size_t N = (size_t)o->n * o->m;
  memcpy(u, o->u->x->a, N * sizeof(float));
  memcpy(v, o->u->y->a, N * sizeof(float));
  memcpy(p, o->p->a, N * sizeof(float));
}
int orc_resume(Oracle *o, const char *path) { /* BDIM.pde:239-251 */
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  char line[512];
  if (!fgets(line, sizeof line, f)) goto bad;
  o->flow_t = strtof(line, NULL);
  if (!fgets(line, sizeof line, f)) goto bad;
  o->flow_dt = strtof(line, NULL);
  for (int i = 0; i < o->n; i++)
    for (int j = 0; j < o->m; j++) {
      if (!fgets(line, sizeof line, f)) goto bad;
      char *q = line;
      float a = strtof(q, &q);
      while (*q == ',' || *q == ' ') q++;
      float b = strtof(q, &q);
      while (*q == ',' || *q == ' ') q++;
      float c = strtof(q, &q);
      A(o->u->x, i, j) = a;
      A(o->u->y, i, j) = b;
      A(o->p, i, j) = c;
    }
  fclose(f);
  return 0;
bad:
  fclose(f);
  return -2;
}
int orc_write(const Oracle *o, const char *path) { /* BDIM.pde:226-237 (%.9g round-trips) */
  FILE *f = fopen(path, "w");
  if (!f) return -1;
  fprintf(f, "%.9g\n%.9g\n", o->flow_t, o->flow_dt);
  for (int i = 0; i < o->n; i++)
    for (int j = 0; j < o->m; j++)
      fprintf(f, "%.9g, %.9g, %.9g\n", A(o->u->x, i, j), A(o->u->y, i, j), A(o->p, i, j));
  fclose(f);
  return 0;
}

void orc_update2(Oracle *o, float xi1_m, float xi2_m, float *force, int *iters, float *resid) {
  /* AFCCylinder.update2, AFCCylinder.pde:45-61 */
  o->flow_dt = o->dt;
  float dphi1 = (2 * xi1_m * o->dt) / (o->dR * o->D);
  float dphi2 = (2 * xi2_m * o->dt) / (o->dR * o->D);
  o->bodies[1]->dphi = dphi1; /* CircleBody.rotate, Body.pde:412-415 */
  o->bodies[1]->phi += dphi1;
  o->bodies[2]->dphi = dphi2;
  o->bodies[2]->phi += dphi2;
  /* flow.update(body), BDIM.pde:126-129 */
  if (union_unsteady(o->bodies)) {
    bdim_get_coeffs(o);
  } else {
    vf_eqs(o->ub, 0.f);
  }
  bdim_update(o);
  bdim_update2(o); /* order2 */
  o->t += o->dt / o->resolution;
  float fx, fy;
  body_pressForce(o->bodies[0], o->p, &fx, &fy);
  o->force[0] = fx * -1; /* PVector.mult(-1) */
  o->force[1] = fy * -1;
  if (force) {
    force[0] = o->force[0];
    force[1] = o->force[1];
  }
  if (iters) {
    iters[0] = o->last_iters[0];
    iters[1] = o->last_iters[1];
  }
  if (resid) {
    resid[0] = o->last_resid[0];
    resid[1] = o->last_resid[1];
  }
}

void orc_get_coeff(const Oracle *o, int which, float *out) {
  const Field *f = NULL;
  switch (which) {
    case 0: f = o->del->x; break;
    case 1: f = o->del->y; break;
    case 2: f = o->del1->x; break;
    case 3: f = o->del1->y; break;
    case 4: f = o->wnx->x; break;
    case 5: f = o->wnx->y; break;
    case 6: f = o->wny->x; break;
    case 7: f = o->wny->y; break;
    case 8: f = o->ub->x; break;
    case 9: f = o->ub->y; break;
    case 10: f = o->distance->x; break;
    case 11: f = o->distance->y; break;
    case 12: f = o->c->x; break;
    case 13: f = o->c->y; break;
    default: return;
  }
  memcpy(out, f->a, sizeof(float) * (size_t)o->n * o->m);
}
void orc_refresh_coeffs(Oracle *o, float dphi1, float dphi2) {
  o->bodies[1]->dphi = dphi1;
  o->bodies[2]->dphi = dphi2;
  bdim_get_coeffs(o);
}
void orc_weights(const Oracle *o, float x, float y, float *w3) { union_weights(o->bodies, x, y, w3); }

float orc_cfl(const Oracle *o) { /* VectorField.CFL, VectorField.pde:225-235 (BDIM.checkCFL without the min) */
  const VField *u = o->u;
  int n = o->n, m = o->m;
  float b = p_abs(A(u->x, 0, 0)) + p_abs(A(u->y, 0, 0));
  float c;
  for (int i = 1; i < n - 1; i++) {
    for (int j = 1; j < m - 1; j++) {
      c = p_abs(A(u->x, i, j)) + p_abs(A(u->y, i, j));
      if (c > b) b = c;
    }
  }
  return 1.f / (b + 3.f * o->nu);
}
void orc_vorticity(const Oracle *o, float *out) {
  Field *w = vf_curl(o->u);
  memcpy(out, w->a, sizeof(float) * (size_t)o->n * o->m);
  field_free(w);
}
void orc_probes(const Oracle *o, int num, float *out) { /* SaveScalar.pde:28-41,61-72 */
  float res = (float)o->resolution;
  float D = res;
  float nn = 16.f * res; /* xLen*res with xLen = (float)xLengths */
  float mm = 8.f * res;
  nn = (float)(o->n_);
  mm = (float)(o->m_);
  float centX = nn / 4, centY = mm / 2;
  for (int i = 0; i < num; i++) {
    float xPre = p_cos((float)i / num * PI_F * 2) * D / 2 + centX;
    float yPre = p_sin((float)i / num * PI_F * 2) * D / 2 + centY;
    out[i] = field_linear(o->p, xPre, yPre);
  }
}
void orc_force(const Oracle *o, float *force) {
  float fx, fy;
  body_pressForce(o->bodies[0], o->p, &fx, &fy);
  force[0] = fx * -1;
  force[1] = fy * -1;
}

/* ---- unit-level entry points ---- */
static VField *vf_from(int n, int m, const float *lx, const float *ly) {
  VField *v = vf_new(n, m, 0, 0);
  memcpy(v->x->a, lx, sizeof(float) * (size_t)n * m);
  memcpy(v->y->a, ly, sizeof(float) * (size_t)n * m);
  return v;
}
static Field *f_from(int n, int m, const float *a, int btype, float bval) {
  Field *f = field_new(n, m, btype, bval);
  memcpy(f->a, a, sizeof(float) * (size_t)n * m);
  return f;
}
void orc_poisson_times(int n, int m, const float *lx, const float *ly, const float *x, float *out) {
  VField *l = vf_from(n, m, lx, ly);
  PMatrix *A_ = pm_new(l);
  Field *xf = f_from(n, m, x, 0, 0);
  Field *ab = pm_times(A_, xf);
  memcpy(out, ab->a, sizeof(float) * (size_t)n * m);
  field_free(ab);
  field_free(xf);
  pm_free(A_);
  vf_free(l);
}
void orc_poisson_diag_inv(int n, int m, const float *lx, const float *ly, float *diag, float *inv) {
  VField *l = vf_from(n, m, lx, ly);
  PMatrix *A_ = pm_new(l);
  memcpy(diag, A_->diagonal->a, sizeof(float) * (size_t)n * m);
  memcpy(inv, A_->inv->a, sizeof(float) * (size_t)n * m);
  pm_free(A_);
  vf_free(l);
}
float orc_mg_updates(int n, int m, const float *lx, const float *ly, float *x, const float *b,
                     int nupdate, float *r_out) {
  VField *l = vf_from(n, m, lx, ly);
  PMatrix *A_ = pm_new(l);
  Field *xf = f_from(n, m, x, 0, 0);
  Field *bf = f_from(n, m, b, 0, 0);
  MG *s = mg_new_top(A_, xf, bf);
  for (int k = 0; k < nupdate; k++) mg_update(s);
  float tol = s->tol;
  memcpy(x, xf->a, sizeof(float) * (size_t)n * m);
  if (r_out) memcpy(r_out, s->r->a, sizeof(float) * (size_t)n * m);
  mg_free(s);
  field_free(xf);
  field_free(bf);
  pm_free(A_);
  vf_free(l);
  return tol;
}
int orc_mg_solve(int n, int m, const float *lx, const float *ly, float *x, const float *b, int itmx,
                 float *resid_linf) {
  VField *l = vf_from(n, m, lx, ly);
  PMatrix *A_ = pm_new(l);
  Field *xf = f_from(n, m, x, 0, 0);
  Field *bf = f_from(n, m, b, 0, 0);
  int it = MGsolver((float)itmx, A_, xf, bf, resid_linf);
  memcpy(x, xf->a, sizeof(float) * (size_t)n * m);
  field_free(xf);
  field_free(bf);
  pm_free(A_);
  vf_free(l);
  return it;
}
void orc_mg_smooth(int n, int m, const float *lx, const float *ly, float *x, float *r, int itmx) {
  VField *l = vf_from(n, m, lx, ly);
  PMatrix *A_ = pm_new(l);
  MG *s = (MG *)calloc(1, sizeof(MG));
  s->A_ = A_;
  s->x = f_from(n, m, x, 0, 0);
  s->own_x = 1;
  s->r = f_from(n, m, r, 0, 0);
  s->its = 4;
  mg_smooth(s, itmx);
  memcpy(x, s->x->a, sizeof(float) * (size_t)n * m);
  memcpy(r, s->r->a, sizeof(float) * (size_t)n * m);
  mg_free(s);
  pm_free(A_);
  vf_free(l);
}
void orc_setbc(int n, int m, float *a, int btype, float bval, int gradientExit) {
  Field *f = f_from(n, m, a, btype, bval);
  f->gradientExit = gradientExit;
  field_setBC(f);
  memcpy(a, f->a, sizeof(float) * (size_t)n * m);
  field_free(f);
}
void orc_advdif(int n, int m, float *fx, float *fy, const float *u0x, const float *u0y, float dt,
                float nu) {
  VField *F = vf_from(n, m, fx, fy);
  VField *u0 = vf_from(n, m, u0x, u0y);
  vf_AdvDif(F, u0, dt, nu);
  memcpy(fx, F->x->a, sizeof(float) * (size_t)n * m);
  memcpy(fy, F->y->a, sizeof(float) * (size_t)n * m);
  vf_free(F);
  vf_free(u0);
}
float orc_polygon_distance(int npts, const float *xs, const float *ys, float xc, float yc, float x,
                           float y) {
  Body *b = body_new(xc, yc);
  for (int i = 0; i < npts; i++) body_add(b, xs[i], ys[i]);
  body_end(b);
  float d = body_distance_poly(b, x, y);
  body_free(b);
  return d;
}
float orc_linear(int n, int m, const float *a, int btype, float x0, float y0) {
  Field f;
  f.n = n;
  f.m = m;
  f.btype = btype;
  f.bval = 0;
  f.gradientExit = 0;
  f.a = (float *)a;
  return field_linear(&f, x0, y0);
}
float orc_delta0(float d) { return bdim_delta0(2.0f, d); }
float orc_delta1(float d) { return bdim_delta1(2.0f, d); }
float orc_union_delta0(float d) { return union_delta0(d); }
