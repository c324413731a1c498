/*
 * ref_tap.cpp — TEST INFRASTRUCTURE.  Thin C shim around the UNMODIFIED reference
 * compute/compute.cpp, which oracle/Makefile compiles from where it lies under
 * /root/reference into oracle/_ref/libsph_ref.so (never copied into this repository).
 *
 * The reference keeps `densities` and `f_total` in malloc'd temporaries that it frees inside
 * updateParticles (compute.cpp:122,128,282-283).  The library is linked with
 * -Wl,--wrap=malloc,--wrap=free, so the reference object's malloc/free calls land in the
 * wrappers below, which copy the two buffers out just before they are released.
 */
#include <cstddef>
#include <cstdlib>
#include <cstring>

/* reference API, compute/compute.h:4-8 (C++ linkage) */
void initParticles(float** particles, float** velocities, int numParticles);
void updateParticles(float* particles, float* velocities, int numParticles);
void vecNormalize(const float* a, float* b);
/* reference kernels and globals, compute.cpp:8-20,41-95 (non-static, so linkable) */
float laplWviscosity(const float r);
float gradWspiky(const float r);
float Wpoly6(const float r);
float gradWpoly6(const float r);
float laplWpoly6(const float r);
extern float soundSpeed, m, rho0, dynVisc, a_gravity[3], timestep, tankSize, surfTension, h, h2, h6, h9;

extern "C" {
void* __real_malloc(size_t);
void __real_free(void*);
}

namespace {
struct Tap {
  bool armed = false;
  int nmalloc = 0;
  void* f_total = nullptr;
  void* densities = nullptr;
  size_t f_bytes = 0, d_bytes = 0;
  float* f_out = nullptr;
  float* d_out = nullptr;
} tap;
}  // namespace

extern "C" void* __wrap_malloc(size_t bytes) {
  void* p = __real_malloc(bytes);
  if (tap.armed) {
    if (tap.nmalloc == 0) { tap.f_total = p; tap.f_bytes = bytes; }      /* compute.cpp:122 */
    else if (tap.nmalloc == 1) { tap.densities = p; tap.d_bytes = bytes; } /* compute.cpp:128 */
    tap.nmalloc++;
  }
  return p;
}

extern "C" void __wrap_free(void* p) {
  if (tap.armed && p) {
    if (p == tap.densities && tap.d_out) std::memcpy(tap.d_out, p, tap.d_bytes);
    if (p == tap.f_total && tap.f_out) std::memcpy(tap.f_out, p, tap.f_bytes);
  }
  __real_free(p);
}

extern "C" {

/* Runs the reference's own initParticles once so that h2/h6/h9 are set by its code
 * (compute.cpp:100-102), then discards the random arrays it allocated. */
void ref_prepare(void) {
  float *p = nullptr, *v = nullptr;
  initParticles(&p, &v, 1);
  delete[] p;
  delete[] v;
}

/* Field order follows sph_params (include/sph_amd.h). */
void ref_set_params(float sound_speed, float mass, float rho0_, float dyn_visc, const float* g,
                    float dt, float tank, float sigma, float h_) {
  soundSpeed = sound_speed; m = mass; rho0 = rho0_; dynVisc = dyn_visc;
  a_gravity[0] = g[0]; a_gravity[1] = g[1]; a_gravity[2] = g[2];
  timestep = dt; tankSize = tank; surfTension = sigma; h = h_;
  ref_prepare();
}

void ref_get_consts(float* out3) { out3[0] = h2; out3[1] = h6; out3[2] = h9; }

/* One reference step with the intermediates tapped; rho_out[n] / force_out[3n] may be NULL. */
void ref_step(float* pos, float* vel, int n, float* rho_out, float* force_out) {
  tap = Tap();
  tap.d_out = rho_out;
  tap.f_out = force_out;
  tap.armed = true;
  updateParticles(pos, vel, n);
  tap.armed = false;
}

/* Untapped steps, for timing the reference as the CPU baseline. */
void ref_run(float* pos, float* vel, int n, int steps) {
  for (int s = 0; s < steps; s++) updateParticles(pos, vel, n);
}

float ref_lapl_viscosity(float r) { return laplWviscosity(r); }
float ref_grad_spiky(float r) { return gradWspiky(r); }
float ref_w_poly6(float r) { return Wpoly6(r); }
float ref_grad_poly6(float r) { return gradWpoly6(r); }
float ref_lapl_poly6(float r) { return laplWpoly6(r); }
void ref_vec_normalize(const float* a, float* b) { vecNormalize(a, b); }

}  // extern "C"
