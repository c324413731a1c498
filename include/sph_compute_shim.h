/*
 * sph_compute_shim.h — the reference's compute/ entry points, re-exported by libsph_compute.so.
 *
 * grzegorz-k-karch/sph declares its compute subsystem in compute/compute.h:4-8 with C++ linkage
 * and configures it through mutable globals defined in compute/compute.cpp:8-20 (main.cpp:12
 * declares `extern float tankSize`).  libsph_compute.so defines the same three functions with the
 * same Itanium-mangled names
 *     _Z13initParticlesPPfS0_i   _Z15updateParticlesPfS_i   _Z12vecNormalizePKfPf
 * and the same global variables, so the reference's main.o / render.o link against it unchanged
 * in place of compute.o (reference Makefile:25-26,29).  Each updateParticles call reads the
 * globals, then runs upload -> one GPU step -> download through the C ABI (sph_amd.h).
 *
 * This header exists for code in THIS repository (the headless driver, tests); when building the
 * reference, keep using its own compute/compute.h.
 */
#ifndef SPH_COMPUTE_SHIM_H
#define SPH_COMPUTE_SHIM_H

/* compute.h:4 — allocates both arrays (ordinary heap memory: the reference releases
 * `particles` with free(), main.cpp:37-39), fills the reference's random block in
 * [0, tankSize/2]^3 with zero velocities and derives h2/h6/h9 (compute.cpp:97-118).
 * The seed is time(0) like the reference's, or $SPH_AMD_SEED when set. */
void initParticles(float** particles, float** velocities, int numParticles);

/* compute.h:6 — one time step in place on host arrays of 3*numParticles floats.  There are no
 * error returns in the reference; on a GPU failure the shim prints the error and aborts. */
void updateParticles(float* particles, float* velocities, int numParticles);

/* compute.h:8 — b = a/|a|, or the zero vector when |a| == 0; a == b allowed (compute.cpp:28-39).
 * Host arithmetic: render.cpp:133 calls it for 65x64 sphere normals at start-up. */
void vecNormalize(const float* a, float* b);

/* compute.cpp:8-20 */
extern float soundSpeed, m, rho0, dynVisc, a_gravity[3], timestep, tankSize, surfTension;
extern float h, h9, h2, h6;

#endif
