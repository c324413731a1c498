// libsph_compute.so: drop-in replacement for the reference's compute.o (see include/sph_compute_shim.h).
// Host glue only; the step itself runs in libsph_amd.so on the GPU.
#include "../../include/sph_compute_shim.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>

#include "../../include/sph_amd.h"

// Same names, types and initial values as reference compute/compute.cpp:8-20.
float soundSpeed = 1.0f;
float m = 0.00020543f;
float rho0 = 600.0f;
float dynVisc = 0.2f;
float a_gravity[3] = {0.0f, -10.0f, 0.0f};
float timestep = 0.005f;
float tankSize = 0.2154f;
float surfTension = 0.0728f;
float h = 0.01f;
float h9 = 0.0f;
float h2 = 0.0f;
float h6 = 0.0f;

namespace {

sph_params params_from_globals() {
  sph_params p;
  sph_params_default(&p);
  p.sound_speed = soundSpeed;
  p.mass = m;
  p.rho0 = rho0;
  p.dyn_visc = dynVisc;
  p.gravity[0] = a_gravity[0];
  p.gravity[1] = a_gravity[1];
  p.gravity[2] = a_gravity[2];
  p.timestep = timestep;
  p.tank_size = tankSize;
  p.surf_tension = surfTension;
  p.h = h;
  if (const char* e = std::getenv("SPH_AMD_EXACT"))
    if (e[0] == '1') p.flags |= SPH_FLAG_EXACT_ORDER;  // bit-identical to the reference, slower
  if (const char* e = std::getenv("SPH_AMD_REFERENCE_ARITH"))
    if (e[0] == '1') p.flags |= SPH_FLAG_REFERENCE_ARITH;  // the reference's term arithmetic, stencil order
  return p;
}

[[noreturn]] void die(const char* what) {
  std::fprintf(stderr, "sph_amd: %s failed: %s\n", what, sph_last_error());
  std::abort();
}

}  // namespace

void initParticles(float** particles, float** velocities, int numParticles) {
  h2 = h * h;  // compute.cpp:100-102
  h6 = h2 * h2 * h2;
  h9 = h * h2 * h6;
  const size_t count = (size_t)(numParticles > 0 ? numParticles : 0) * 3;
  *particles = static_cast<float*>(std::malloc((count ? count : 1) * sizeof(float)));
  *velocities = static_cast<float*>(std::malloc((count ? count : 1) * sizeof(float)));
  if (!*particles || !*velocities) die("initParticles allocation");
  // The caller keeps these two arrays for the whole run and hands them back every step (main.cpp:27-35): page-locked,
  // the 24 bytes per particle each way are direct DMA.  Plain malloc memory before and after (main.cpp:37 frees one of
  // them with free()); a refusal (no device yet, locked-memory limit) only costs speed.
  if (count * sizeof(float) >= (size_t)1 << 18) {
    (void)sph_host_register(*particles, count * sizeof(float));
    (void)sph_host_register(*velocities, count * sizeof(float));
  }
  unsigned long long seed = (unsigned long long)std::time(nullptr);
  if (const char* e = std::getenv("SPH_AMD_SEED")) seed = std::strtoull(e, nullptr, 10);
  sph_params p = params_from_globals();
  if (sph_fill_random_block(&p, seed, *particles, *velocities, numParticles) != SPH_OK) die("initParticles");
}

void updateParticles(float* particles, float* velocities, int numParticles) {
  static const bool once = [] {  // SPH_AMD_RESIDENT=1: the caller does not edit the arrays between steps (main.cpp:31-35)
    const char* e = std::getenv("SPH_AMD_RESIDENT");
    if (e && e[0] == '1') sph_one_call_mode(1);
    return true;
  }();
  (void)once;
  sph_params p = params_from_globals();
  if (sph_update_particles(&p, particles, velocities, numParticles) != SPH_OK) die("updateParticles");
}

void vecNormalize(const float* a, float* b) {
  // |a| = sqrt((a0^2 + a1^2) + a2^2) in fp32, then three divisions (compute.cpp:23-39)
  const float len = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  if (len > 0.0f) {
    const float x = a[0] / len, y = a[1] / len, z = a[2] / len;
    b[0] = x;
    b[1] = y;
    b[2] = z;
  } else {
    b[0] = b[1] = b[2] = 0.0f;
  }
}
