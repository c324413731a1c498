// Device-side arithmetic shared by every neighbour sweep.  All of it mirrors the reference expression by
// expression (reference compute/compute.cpp, lines cited inline); translation units that include this
// header must be compiled with -ffp-contract=off.
#pragma once

#include "sph_internal.hpp"

namespace sph {

// ---- cell index ---------------------------------------------------------------------------------
// c = clamp(floorf(x / cell), 0, gn-1).  True fp32 division: x * (1/cell) rounds differently.
// fmaxf(NaN, 0) = 0 keeps NaN positions inside the table.
__device__ inline int cell_coord(float x, float cell, int gn) {
  float q = floorf(x / cell);
  q = fminf(fmaxf(q, 0.0f), (float)(gn - 1));
  return (int)q;
}

// 1 if z lies in the upper half of its cell (in z), 0 in the lower.  x / (cell/2) = 2 (x / cell) exactly in fp32 (a
// power of two commutes with rounding), so this is the low bit of floor(2 q) for the very q whose floor is the cell
// coordinate: the two never disagree about which cell the halves belong to, clamps included.
// Sorting a cell's particles lower half first lets a particle leave out the far half of the cell beyond its own half:
// a particle in the lower half of cell cz and one in the upper half of cell cz + 1 are more than one padded cell edge
// minus half an ulp of the quotient apart, the same margin that keeps cells two apart from interacting.
__device__ inline int upper_half(const DevParams& P, float z) { return cell_coord(z, P.cell * 0.5f, 2 * P.nz) & 1; }

__device__ inline float dist2(float dx, float dy, float dz) { return dx * dx + dy * dy + dz * dz; }  // :143
// Grid-sorted position of slot j (w = 0).
__device__ inline float4 sorted_pos(const Sorted& st, int j) {
  const float* p = st.pos3 + 3 * (size_t)j;
  return make_float4(p[0], p[1], p[2], 0.0f);
}

__device__ inline float dist2_to(const float4& a, const float4& b, float& dx, float& dy, float& dz) {
  dx = a.x - b.x;  // :140-142
  dy = a.y - b.y;
  dz = a.z - b.z;
  return dist2(dx, dy, dz);
}

// Correctly rounded fp32 square root (what sqrtf gives on the reference's host), for the r of :143.
// hipcc expands sqrtf into v_sqrt_f32 (1 ulp) plus a correction: the candidates one ulp below and above are
// tried with exact fma residuals.  Around that it scales arguments below 2^-96 (v_sqrt_f32 does not take
// denormals) and patches 0 and inf.  Here every argument is a squared distance: 0 for the particle itself,
// which the correction below leaves at 0 (both residual tests fail), or far above 2^-96 for any two fp32
// positions that differ - the scaling is kept for the arguments that need it, behind a branch no wave takes.
__device__ inline float sqrt_rn(float x) {
  if (__builtin_expect(x < 0x1p-96f && x > 0.0f, 0)) return sqrtf(x);
  float s = __builtin_amdgcn_sqrtf(x);
  const int bits = __float_as_int(s);
  const float below = __int_as_float(bits - 1), above = __int_as_float(bits + 1);
  const float e_below = __builtin_fmaf(-below, s, x), e_above = __builtin_fmaf(-above, s, x);
  s = e_below <= 0.0f ? below : s;
  s = e_above > 0.0f ? above : s;
  return s;
}

// ---- smoothing kernels, compute.cpp:41-95 (callers guarantee r <= h) -------------------------------
__device__ inline float w_poly6(const DevParams& P, float r) {  // :61-71
  float rr = r * r;
  float d = P.h2 - rr;
  float d3 = d * d * d;
  return (float)(P.c_poly6 * (double)d3);
}
__device__ inline float grad_spiky(const DevParams& P, float r) {  // :50-59
  float d = P.h - r;
  float d2 = d * d;
  return (float)(P.c_spiky_grad * (double)d2);
}
__device__ inline float lapl_visc(const DevParams& P, float r) {  // :41-48
  return (float)(P.c_visc_lapl * (double)(P.h - r));
}
__device__ inline float grad_poly6(const DevParams& P, float r) {  // :73-83
  float rr = r * r;
  float d = P.h2 - rr;
  float d2 = d * d;
  return (float)(P.c_poly6_grad * (double)r * (double)d2);
}
__device__ inline float lapl_poly6(const DevParams& P, float r) {  // :85-95
  float rr = r * r;
  float d = P.h2 - rr;
  float d2 = d * d;
  float poly = 4.0f * rr * d - d2;
  return (float)(P.c_poly6_lapl * (double)poly);
}

// ---- one neighbour's contribution to the three pair sums (:161-179, :194-209, :233-253) -------------
constexpr int kForceComps = 10;
struct PairOut {
  float v[kForceComps];  // 0-2 pressure (:176-178)  3-5 viscosity (:206-208)  6-8 colour gradient (:248-250)  9 colour laplacian (:252)
};

// ---- IEEE-754 fp32 division with a shared divisor -------------------------------------------------------
// A pair divides three numbers by r and four by rho_j.  hipcc expands every `a / b` into
//   v_div_scale x2, v_rcp, 2 fma (Newton step on the reciprocal), mul, 4 fma, v_div_fmas, v_div_fixup.
// The scale/fixup instructions only act on operands near the ends of the exponent range (|b| or |a/b|
// within ~2^-126 / 2^+96 of the limits), which r in (0, h], rho >= m*W(0) and the numerators here never are.
// Without them the quotient is exactly the fma chain below, so it is computed with ONE refined reciprocal
// per divisor: same bits as `a / b`, a third of the instructions.  (Sign of a zero quotient aside, which
// cannot reach a result: the sums start at +0 and x + (+-0) = x.)
__device__ inline float refined_rcp(float b) {
  float y = __builtin_amdgcn_rcpf(b);
  float e = __builtin_fmaf(-b, y, 1.0f);
  return __builtin_fmaf(e, y, y);
}
__device__ inline float div_by(float a, float b, float y) {  // y = refined_rcp(b)
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}

__device__ inline PairOut pair_eval(const DevParams& P, float dx, float dy, float dz, float r2, float pi,
                                    float rho_j, float dvx, float dvy, float dvz) {
  PairOut o;
  float r = sqrt_rn(r2);
  float ux = 0.0f, uy = 0.0f, uz = 0.0f;  // vecNormalize(r, r)  :28-39
  if (r > 0.0f) {
    const float yr = refined_rcp(r);
    ux = div_by(dx, r, yr);
    uy = div_by(dy, r, yr);
    uz = div_by(dz, r, yr);
  }
  const float yrho = refined_rcp(rho_j);
  float pj = P.c2 * (rho_j - P.rho0);  // :164
  // :170  x / (2*rho) == (x / rho) / 2 exactly: scaling by a power of two commutes with rounding
  float mfp = div_by(P.neg_m * (pi + pj), rho_j, yrho) * 0.5f * grad_spiky(P, r);
  o.v[0] = mfp * ux;
  o.v[1] = mfp * uy;
  o.v[2] = mfp * uz;
  float t = div_by(P.visc_m * lapl_visc(P, r), rho_j, yrho);  // :204
  o.v[3] = dvx * t;
  o.v[4] = dvy * t;
  o.v[5] = dvz * t;
  if (P.tension) {
    float mg = div_by(P.m * grad_poly6(P, r), rho_j, yrho);  // :242
    o.v[6] = ux * mg;
    o.v[7] = uy * mg;
    o.v[8] = uz * mg;
    o.v[9] = div_by(P.m * lapl_poly6(P, r), rho_j, yrho);  // :252
  } else {
    o.v[6] = o.v[7] = o.v[8] = o.v[9] = 0.0f;
  }
  return o;
}

// ---- the same contribution in fast arithmetic ------------------------------------------------------------------
// Same terms as pair_eval, evaluated the cheap way: fp32 coefficients folded with the mass factors on the host,
// 1/r from v_rsq_f32 with one Newton step, 1/rho_j precomputed per particle by the density pass, fused multiply-adds.
// Differs from the reference's value of each term by a few ulp (the reference itself is a few ulp from the exact
// value: sqrt, up to three divisions and a double product, each rounded), never in which pairs contribute: that is
// decided by the neighbour mask, which is exact.  Uses r2 where the reference re-squares its rounded r.
//   pressure   -m (p_i + p_j) / (2 rho_j) * c_spiky (h - r)^2 * d/r                         (:56,:170-178)
//   viscosity  (v_j - v_i) * mu m c_visc (h - r) / rho_j                                    (:45,:204-208)
//   colour     grad: m c_grad r (h2 - r2)^2 / rho_j * d/r = m c_grad (h2 - r2)^2 / rho_j * d      (:80,:242-250)
//              lapl: m c_lapl (4 r2 (h2 - r2) - (h2 - r2)^2) / rho_j                        (:92,:252)
__device__ inline float fast_rsqrt(float x) {  // 1/sqrt(x), x > 0
  const float y = __builtin_amdgcn_rsqf(x);
  const float e = __builtin_fmaf(-x * y, y, 1.0f);  // 1 - x y^2
  return __builtin_fmaf(0.5f * y, e, y);
}
__device__ inline float fast_rcp(float x) {
  const float y = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, y, 1.0f);
  return __builtin_fmaf(e, y, y);
}
__device__ inline void pair_accumulate_fast(const DevParams& P, float* a, float dx, float dy, float dz, float pi,
                                            float rho_j, float inv_rho_j, float dvx, float dvy, float dvz) {
  const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
  const float rinv = r2 > 0.0f ? fast_rsqrt(r2) : 0.0f;  // coincident particles: no direction (vecNormalize of 0, :36-37)
  const float r = r2 * rinv;
  const float hr = P.h - r;
  const float pj = __builtin_fmaf(P.c2, rho_j, -P.c2rho0);  // :164
  const float sp = (P.k_press * (hr * hr)) * ((pi + pj) * inv_rho_j) * rinv;
  a[0] = __builtin_fmaf(sp, dx, a[0]);
  a[1] = __builtin_fmaf(sp, dy, a[1]);
  a[2] = __builtin_fmaf(sp, dz, a[2]);
  const float tv = (P.k_visc * hr) * inv_rho_j;
  a[3] = __builtin_fmaf(dvx, tv, a[3]);
  a[4] = __builtin_fmaf(dvy, tv, a[4]);
  a[5] = __builtin_fmaf(dvz, tv, a[5]);
  if (P.tension) {
    const float d = P.h2 - r2;
    const float d2 = d * d;
    const float mg = (P.k_tgrad * d2) * inv_rho_j;
    a[6] = __builtin_fmaf(mg, dx, a[6]);
    a[7] = __builtin_fmaf(mg, dy, a[7]);
    a[8] = __builtin_fmaf(mg, dz, a[8]);
    const float poly = __builtin_fmaf(4.0f * r2, d, -d2);
    a[9] = __builtin_fmaf(P.k_tlapl * poly, inv_rho_j, a[9]);
  }
}

// ---- per-particle epilogue: total force, symplectic Euler, tank walls --------------------------------
struct Advanced {
  float x, y, z, vx, vy, vz, fx, fy, fz;
};

// a[] holds the ten sums in PairOut order.
__device__ inline Advanced finish_particle(const DevParams& P, const float* a, float xi, float yi, float zi,
                                           float vxi, float vyi, float vzi, float rho_i) {
  // f_total = (((0 + pressure) + viscosity) + gravity*rho) + tension   (:181-183,:211-213,:220-222,:259-261)
  float fx = 0.0f + a[0], fy = 0.0f + a[1], fz = 0.0f + a[2];
  fx += a[3];
  fy += a[4];
  fz += a[5];
  fx += P.gx * rho_i;
  fy += P.gy * rho_i;
  fz += P.gz * rho_i;
  if (P.tension) {
    float len = sqrtf(a[6] * a[6] + a[7] * a[7] + a[8] * a[8]);  // vecNormalize(gradcolor)  :255
    float nx = 0.0f, ny = 0.0f, nz = 0.0f;
    if (len > 0.0f) {
      nx = a[6] / len;
      ny = a[7] / len;
      nz = a[8] / len;
    }
    float t = P.neg_sigma * a[9];  // :257
    fx += t * nx;
    fy += t * ny;
    fz += t * nz;
  }
  // symplectic Euler (:265-280)
  float scl = P.dt / rho_i;
  vxi += fx * scl;
  vyi += fy * scl;
  vzi += fz * scl;
  xi = xi + vxi * P.dt;
  yi = yi + vyi * P.dt;
  zi = zi + vzi * P.dt;
  // tank walls, damping 1 (:286-319)
  if (xi < 0.0f) { xi = 0.0f; vxi = -1.0f * vxi; }
  if (xi > P.tank) { xi = P.tank; vxi = -1.0f * vxi; }
  if (yi < 0.0f) { yi = 0.0f; vyi = -1.0f * vyi; }
  if (yi > P.tank) { yi = P.tank; vyi = -1.0f * vyi; }
  if (zi < 0.0f) { zi = 0.0f; vzi = -1.0f * vzi; }
  if (zi > P.tank) { zi = P.tank; vzi = -1.0f * vzi; }
  return Advanced{xi, yi, zi, vxi, vyi, vzi, fx, fy, fz};
}

}  // namespace sph
