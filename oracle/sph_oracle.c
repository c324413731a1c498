/*
 * sph_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see sph_oracle.h).
 *
 * CPU restatement of grzegorz-k-karch/sph compute/compute.cpp.  Each function cites the
 * reference lines it follows.  Build: gcc -O2 -fopenmp -ffp-contract=off (no FMA contraction,
 * matching the reference's x86-64 object which contains no FMA instructions).
 *
 * Arithmetic notes that make the restatement bit-exact:
 *  - M_PI is a double, so every kernel coefficient is a double and the product
 *    coefficient * polynomial is formed in double and rounded ONCE to float
 *    (compute.cpp:45,56,68,80,92).  The polynomials themselves are fp32.
 *  - r2 is (dx*dx + dy*dy) + dz*dz in fp32, the neighbour test is sqrtf(r2) <= h
 *    (compute.cpp:143-144 and the `r <= h` guards).
 *  - Wpoly6 & co. square the ROUNDED r again (compute.cpp:65,77,89); they do not reuse r2.
 */
#include "sph_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  float h, m, rho0, c2, visc_m, dt, tank, sigma, g[3];
  sph_oracle_consts k;
  int tension;
} ctx_t;

/* ---- derived constants -------------------------------------------------------------------- */

static float largest_r2_within(float h) {
  /* max { x : sqrtf(x) <= h }.  sqrtf is monotone, so walk from h*h. */
  float x = h * h;
  while (sqrtf(x) <= h) x = nextafterf(x, INFINITY);
  while (sqrtf(x) > h) x = nextafterf(x, -INFINITY);
  return x;
}

void sph_oracle_derive(const sph_params* p, sph_oracle_consts* c) {
  /* compute.cpp:100-102 */
  c->h2 = p->h * p->h;
  c->h6 = c->h2 * c->h2 * c->h2;
  c->h9 = p->h * c->h2 * c->h6;
  /* coefficient sub-expressions exactly as written in the kernels (float literals, double pi) */
  c->c_visc_lapl = 45.0f / (M_PI * c->h6);             /* :45 */
  c->c_spiky_grad = -45.0f / (M_PI * c->h6);           /* :56 */
  c->c_poly6 = 315.0f / (64.0f * M_PI * c->h9);        /* :68 */
  c->c_poly6_grad = -945.0f / (32.0f * M_PI * c->h9);  /* :80 */
  c->c_poly6_lapl = 945.0f / (32.0f * M_PI * c->h9);   /* :92 */
  c->r2_max = largest_r2_within(p->h);
  c->cell_size = p->cell_size > 0.0f ? p->cell_size : p->h * 1.000244140625f; /* 1 + 2^-12 */
  int nc = (int)ceil((double)p->tank_size / (double)c->cell_size);
  c->ncell = nc < 1 ? 1 : nc;
}

static void make_ctx(const sph_params* p, ctx_t* c) {
  c->h = p->h;
  c->m = p->mass;
  c->rho0 = p->rho0;
  c->c2 = p->sound_speed * p->sound_speed; /* :153 */
  c->visc_m = p->dyn_visc * p->mass;       /* left-most product of :204 */
  c->dt = p->timestep;
  c->tank = p->tank_size;
  c->sigma = p->surf_tension;
  memcpy(c->g, p->gravity, sizeof c->g);
  sph_oracle_derive(p, &c->k);
  c->tension = (p->flags & SPH_FLAG_SURFACE_TENSION) != 0;
}

/* ---- smoothing kernels, compute.cpp:41-95 --------------------------------------------------- */

static inline float k_lapl_visc(const ctx_t* c, float r) { /* :41-48 */
  if (!(r <= c->h)) return 0.0f;
  return (float)(c->k.c_visc_lapl * (double)(c->h - r));
}
static inline float k_grad_spiky(const ctx_t* c, float r) { /* :50-59 */
  if (!(r <= c->h)) return 0.0f;
  float d = c->h - r;
  float d2 = d * d;
  return (float)(c->k.c_spiky_grad * (double)d2);
}
static inline float k_poly6(const ctx_t* c, float r) { /* :61-71 */
  if (!(r <= c->h)) return 0.0f;
  float rr = r * r;
  float d = c->k.h2 - rr;
  float d3 = d * d * d;
  return (float)(c->k.c_poly6 * (double)d3);
}
static inline float k_grad_poly6(const ctx_t* c, float r) { /* :73-83 */
  if (!(r <= c->h)) return 0.0f;
  float rr = r * r;
  float d = c->k.h2 - rr;
  float d2 = d * d;
  return (float)(c->k.c_poly6_grad * (double)r * (double)d2);
}
static inline float k_lapl_poly6(const ctx_t* c, float r) { /* :85-95 */
  if (!(r <= c->h)) return 0.0f;
  float rr = r * r;
  float d = c->k.h2 - rr;
  float d2 = d * d;
  float poly = 4.0f * rr * d - d2;
  return (float)(c->k.c_poly6_lapl * (double)poly);
}

float sph_oracle_lapl_viscosity(const sph_params* p, float r) { ctx_t c; make_ctx(p, &c); return k_lapl_visc(&c, r); }
float sph_oracle_grad_spiky(const sph_params* p, float r) { ctx_t c; make_ctx(p, &c); return k_grad_spiky(&c, r); }
float sph_oracle_w_poly6(const sph_params* p, float r) { ctx_t c; make_ctx(p, &c); return k_poly6(&c, r); }
float sph_oracle_grad_poly6(const sph_params* p, float r) { ctx_t c; make_ctx(p, &c); return k_grad_poly6(&c, r); }
float sph_oracle_lapl_poly6(const sph_params* p, float r) { ctx_t c; make_ctx(p, &c); return k_lapl_poly6(&c, r); }

/* compute.cpp:23-39.  (float)sqrt((double)x) == sqrtf(x) for every float x, so the double
 * sqrt the reference's vecLength resolves to gives the same bits as sqrtf. */
static inline float len3(const float* a) { return sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static inline void unit3(const float* a, float* b) {
  float len = len3(a);
  if (len > 0.0f) {
    float x = a[0] / len, y = a[1] / len, z = a[2] / len;
    b[0] = x; b[1] = y; b[2] = z;
  } else {
    b[0] = 0.0f; b[1] = 0.0f; b[2] = 0.0f;
  }
}
void sph_oracle_vec_normalize(const float* a, float* b) { unit3(a, b); }

/* ---- per-pair terms (shared by the all-pairs and the grid formulations) --------------------- */

static inline void sep(const float* pos, int i, int j, float d[3]) { /* :140-142 */
  d[0] = pos[3 * i + 0] - pos[3 * j + 0];
  d[1] = pos[3 * i + 1] - pos[3 * j + 1];
  d[2] = pos[3 * i + 2] - pos[3 * j + 2];
}

static inline void acc_density(const ctx_t* c, const float* pos, int i, int j, float* rho) {
  float d[3];
  sep(pos, i, j, d);
  float r = len3(d); /* :143-144 */
  *rho += c->m * k_poly6(c, r); /* :146 */
}

static inline void acc_pressure(const ctx_t* c, const float* pos, const float* rho, float pi,
                                int i, int j, float f[3]) { /* :163-178 */
  float d[3];
  float pj = c->c2 * (rho[j] - c->rho0);
  sep(pos, i, j, d);
  float r = len3(d);
  float mfp = -c->m * (pi + pj) / (2.0f * rho[j]) * k_grad_spiky(c, r);
  unit3(d, d);
  f[0] += mfp * d[0];
  f[1] += mfp * d[1];
  f[2] += mfp * d[2];
}

static inline void acc_viscosity(const ctx_t* c, const float* pos, const float* vel,
                                 const float* rho, int i, int j, float f[3]) { /* :196-208 */
  float d[3];
  float dv0 = vel[3 * j + 0] - vel[3 * i + 0];
  float dv1 = vel[3 * j + 1] - vel[3 * i + 1];
  float dv2 = vel[3 * j + 2] - vel[3 * i + 2];
  sep(pos, i, j, d);
  float r = len3(d);
  float t = c->visc_m * k_lapl_visc(c, r) / rho[j];
  f[0] += dv0 * t;
  f[1] += dv1 * t;
  f[2] += dv2 * t;
}

static inline void acc_color(const ctx_t* c, const float* pos, const float* rho, int i, int j,
                             float grad[3], float* lapl) { /* :235-252 */
  float d[3], u[3];
  sep(pos, i, j, d);
  float r = len3(d);
  float mg = c->m * k_grad_poly6(c, r) / rho[j];
  unit3(d, u);
  u[0] *= mg; u[1] *= mg; u[2] *= mg;
  grad[0] += u[0];
  grad[1] += u[1];
  grad[2] += u[2];
  *lapl += c->m * k_lapl_poly6(c, r) / rho[j];
}

/* :255-261 — tension force from the colour-field sums */
static inline void tension_force(const ctx_t* c, float grad[3], float lapl, float out[3]) {
  unit3(grad, grad);
  float t = -c->sigma * lapl;
  out[0] = t * grad[0];
  out[1] = t * grad[1];
  out[2] = t * grad[2];
}

/* :265-280 and :286-319 */
static inline void integrate_and_walls(const ctx_t* c, float* x, float* v, const float* f,
                                       float rho) {
  float sc = c->dt / rho;
  v[0] += f[0] * sc;
  v[1] += f[1] * sc;
  v[2] += f[2] * sc;
  x[0] = x[0] + v[0] * c->dt;
  x[1] = x[1] + v[1] * c->dt;
  x[2] = x[2] + v[2] * c->dt;
}
static inline void walls(const ctx_t* c, float* x, float* v) {
  const float damping = 1.0f;
  for (int a = 0; a < 3; a++) {
    if (x[a] < 0.0f) { x[a] = 0.0f; v[a] = -damping * v[a]; }
    if (x[a] > c->tank) { x[a] = c->tank; v[a] = -damping * v[a]; }
  }
}

/* ---- all-pairs step: literal restatement of compute.cpp:120-320 ------------------------------ */

void sph_oracle_step_allpairs(const sph_params* p, float* pos, float* vel, int n,
                              float* rho_out, float* force_out) {
  ctx_t c;
  make_ctx(p, &c);
  float* f = (float*)calloc((size_t)n * 3 + 1, sizeof(float)); /* f_total, :122-125 */
  float* rho = (float*)malloc(((size_t)n + 1) * sizeof(float)); /* densities, :128 */

#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) { /* density, :131-150 */
    float s = 0.0f;
    for (int j = 0; j < n; j++) acc_density(&c, pos, i, j, &s);
    rho[i] = s;
  }

#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) { /* pressure, :153-184 */
    float pi = c.c2 * (rho[i] - c.rho0);
    float a[3] = {0.0f, 0.0f, 0.0f};
    for (int j = 0; j < n; j++) acc_pressure(&c, pos, rho, pi, i, j, a);
    f[3 * i + 0] += a[0];
    f[3 * i + 1] += a[1];
    f[3 * i + 2] += a[2];
  }

#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) { /* viscosity, :187-214 */
    float a[3] = {0.0f, 0.0f, 0.0f};
    for (int j = 0; j < n; j++) acc_viscosity(&c, pos, vel, rho, i, j, a);
    f[3 * i + 0] += a[0];
    f[3 * i + 1] += a[1];
    f[3 * i + 2] += a[2];
  }

#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) { /* gravity, :217-223 */
    f[3 * i + 0] += c.g[0] * rho[i];
    f[3 * i + 1] += c.g[1] * rho[i];
    f[3 * i + 2] += c.g[2] * rho[i];
  }

  if (c.tension) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) { /* surface tension, :226-262 */
      float grad[3] = {0.0f, 0.0f, 0.0f}, lapl = 0.0f, t[3];
      for (int j = 0; j < n; j++) acc_color(&c, pos, rho, i, j, grad, &lapl);
      tension_force(&c, grad, lapl, t);
      f[3 * i + 0] += t[0];
      f[3 * i + 1] += t[1];
      f[3 * i + 2] += t[2];
    }
  }

  if (rho_out) memcpy(rho_out, rho, (size_t)n * sizeof(float));
  if (force_out) memcpy(force_out, f, (size_t)n * 3 * sizeof(float));

#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) /* :265-280 */
    integrate_and_walls(&c, &pos[3 * i], &vel[3 * i], &f[3 * i], rho[i]);

  free(rho);
  free(f);

#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) walls(&c, &pos[3 * i], &vel[3 * i]); /* :286-319 */
}

void sph_oracle_neighbor_counts(const sph_params* p, const float* pos, int n, int32_t* out) {
  ctx_t c;
  make_ctx(p, &c);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    int k = 0;
    for (int j = 0; j < n; j++) {
      float d[3];
      sep(pos, i, j, d);
      if (len3(d) <= c.h) k++;
    }
    out[i] = k;
  }
}

/* ---- uniform grid ---------------------------------------------------------------------------- */

static inline int cell_coord(float x, float cell, int nc) {
  float q = floorf(x / cell);
  if (!(q >= 0.0f)) q = 0.0f; /* also catches NaN */
  if (q > (float)(nc - 1)) q = (float)(nc - 1);
  return (int)q;
}

void sph_oracle_cell_ids(const sph_params* p, const float* pos, int n, int32_t* out) {
  sph_oracle_consts k;
  sph_oracle_derive(p, &k);
  for (int i = 0; i < n; i++) {
    int cx = cell_coord(pos[3 * i + 0], k.cell_size, k.ncell);
    int cy = cell_coord(pos[3 * i + 1], k.cell_size, k.ncell);
    int cz = cell_coord(pos[3 * i + 2], k.cell_size, k.ncell);
    out[i] = (cx * k.ncell + cy) * k.ncell + cz;
  }
}

/* Order of the grid twin's sorted copy: by cell; inside a cell the particles in the lower half of the cell in z
 * first, then those in the upper half; inside a half by particle id.  `half` is the low bit of the particle's
 * coordinate on a grid of half the cell edge in z (x / (cell/2) = 2 (x / cell) exactly in fp32, so the two grids agree
 * about every cell boundary).  The summation order of order 0 is this sorted order, run by run. */
typedef struct { int32_t cell, half, id, idx; } key_t_;
static int cmp_key(const void* a, const void* b) {
  const key_t_* x = (const key_t_*)a; const key_t_* y = (const key_t_*)b;
  if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
  if (x->half != y->half) return x->half < y->half ? -1 : 1;
  if (x->id != y->id) return x->id < y->id ? -1 : 1;
  return 0;
}
typedef struct { int32_t id, j; } nb_t;
static int cmp_nb(const void* a, const void* b) {
  int32_t x = ((const nb_t*)a)->id, y = ((const nb_t*)b)->id;
  return x < y ? -1 : (x > y);
}

/* Neighbour list of sorted particle s in the requested order; returns the count. */
static int gather_neighbors(const ctx_t* c, const float* spos, const int32_t* sid,
                            const int32_t* cstart, int s, int order, nb_t** buf, int* cap) {
  const int nc = c->k.ncell;
  int cx = cell_coord(spos[3 * s + 0], c->k.cell_size, nc);
  int cy = cell_coord(spos[3 * s + 1], c->k.cell_size, nc);
  int cz = cell_coord(spos[3 * s + 2], c->k.cell_size, nc);
  int k = 0;
  for (int dx = -1; dx <= 1; dx++) {
    int x = cx + dx;
    if (x < 0 || x >= nc) continue;
    for (int dy = -1; dy <= 1; dy++) {
      int y = cy + dy;
      if (y < 0 || y >= nc) continue;
      int z0 = cz > 0 ? cz - 1 : 0, z1 = cz < nc - 1 ? cz + 1 : nc - 1;
      int row = (x * nc + y) * nc;
      for (int j = cstart[row + z0]; j < cstart[row + z1 + 1]; j++) {
        float d[3];
        sep(spos, s, j, d);
        float r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (r2 <= c->k.r2_max) {
          if (k == *cap) { *cap = *cap * 2 + 64; *buf = (nb_t*)realloc(*buf, (size_t)*cap * sizeof(nb_t)); }
          (*buf)[k].id = sid[j];
          (*buf)[k].j = j;
          k++;
        }
      }
    }
  }
  if (order == 1) qsort(*buf, (size_t)k, sizeof(nb_t), cmp_nb);
  return k;
}

void sph_oracle_step_grid(const sph_params* p, float* pos, float* vel, const int32_t* ids,
                          int n, int order, float* rho_out, float* force_out,
                          int32_t* ncount_out) {
  ctx_t c;
  make_ctx(p, &c);
  const int nc = c.k.ncell;
  const size_t ncells = (size_t)nc * nc * nc;
  key_t_* keys = (key_t_*)malloc(((size_t)n + 1) * sizeof(key_t_));
  int32_t* cid = (int32_t*)malloc(((size_t)n + 1) * sizeof(int32_t));
  sph_oracle_cell_ids(p, pos, n, cid);
  for (int i = 0; i < n; i++) {
    keys[i].cell = cid[i];
    keys[i].half = cell_coord(pos[3 * i + 2], c.k.cell_size * 0.5f, 2 * nc) & 1;
    keys[i].id = ids ? ids[i] : i;
    keys[i].idx = i;
  }
  qsort(keys, (size_t)n, sizeof(key_t_), cmp_key);

  float* spos = (float*)malloc(((size_t)n * 3 + 1) * sizeof(float));
  float* svel = (float*)malloc(((size_t)n * 3 + 1) * sizeof(float));
  int32_t* sid = (int32_t*)malloc(((size_t)n + 1) * sizeof(int32_t));
  int32_t* cstart = (int32_t*)calloc(ncells + 1, sizeof(int32_t));
  for (int s = 0; s < n; s++) {
    memcpy(&spos[3 * s], &pos[3 * keys[s].idx], 3 * sizeof(float));
    memcpy(&svel[3 * s], &vel[3 * keys[s].idx], 3 * sizeof(float));
    sid[s] = keys[s].id;
    cstart[keys[s].cell + 1]++;
  }
  for (size_t k = 0; k < ncells; k++) cstart[k + 1] += cstart[k];

  float* rho = (float*)malloc(((size_t)n + 1) * sizeof(float));
  float* f = (float*)calloc((size_t)n * 3 + 1, sizeof(float));
  int32_t* ncnt = (int32_t*)malloc(((size_t)n + 1) * sizeof(int32_t));

#pragma omp parallel
  {
    nb_t* nb = NULL;
    int cap = 0;
#pragma omp for schedule(static)
    for (int s = 0; s < n; s++) {
      int k = gather_neighbors(&c, spos, sid, cstart, s, order, &nb, &cap);
      float r = 0.0f;
      for (int t = 0; t < k; t++) acc_density(&c, spos, s, nb[t].j, &r);
      rho[s] = r;
      ncnt[s] = k;
    }
#pragma omp for schedule(static)
    for (int s = 0; s < n; s++) {
      int k = gather_neighbors(&c, spos, sid, cstart, s, order, &nb, &cap);
      float pi = c.c2 * (rho[s] - c.rho0);
      float fp[3] = {0, 0, 0}, fv[3] = {0, 0, 0}, grad[3] = {0, 0, 0}, lapl = 0.0f, ft[3];
      for (int t = 0; t < k; t++) acc_pressure(&c, spos, rho, pi, s, nb[t].j, fp);
      for (int t = 0; t < k; t++) acc_viscosity(&c, spos, svel, rho, s, nb[t].j, fv);
      float* fs = &f[3 * s];
      for (int a = 0; a < 3; a++) { fs[a] += fp[a]; fs[a] += fv[a]; fs[a] += c.g[a] * rho[s]; }
      if (c.tension) {
        for (int t = 0; t < k; t++) acc_color(&c, spos, rho, s, nb[t].j, grad, &lapl);
        tension_force(&c, grad, lapl, ft);
        for (int a = 0; a < 3; a++) fs[a] += ft[a];
      }
    }
    free(nb);
  }

  /* integrate on copies so that the force sweep above saw only old positions */
  for (int s = 0; s < n; s++) {
    int i = keys[s].idx;
    if (rho_out) rho_out[i] = rho[s];
    if (force_out) memcpy(&force_out[3 * i], &f[3 * s], 3 * sizeof(float));
    if (ncount_out) ncount_out[i] = ncnt[s];
    integrate_and_walls(&c, &pos[3 * i], &vel[3 * i], &f[3 * s], rho[s]);
    walls(&c, &pos[3 * i], &vel[3 * i]);
  }
  free(ncnt); free(f); free(rho); free(cstart); free(sid); free(svel); free(spos); free(cid); free(keys);
}

int sph_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* The OpenMP runtime is shared by everything in the process that links it (the reference build included). */
void sph_oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
