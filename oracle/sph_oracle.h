/*
 * sph_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's updateParticles (grzegorz-k-karch/sph,
 * compute/compute.cpp:120-320) used only as the checker for the HIP path: by tests/,
 * by __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  Nothing under sph_amd/
 * may link, load or call it.
 *
 * Parity pin: every function here is checked bit-for-bit against the reference's own
 * compute.cpp compiled unmodified into oracle/_ref/ (see oracle/Makefile, tests/test_oracle_vs_ref.py)
 * and against the golden vectors under tests/golden/ that were generated from that build.
 */
#ifndef SPH_ORACLE_H
#define SPH_ORACLE_H

#include <stdint.h>
#include "../include/sph_amd.h" /* sph_params only: the oracle shares the parameter struct */

#ifdef __cplusplus
extern "C" {
#endif

/* Quantities the reference derives from its globals. */
typedef struct sph_oracle_consts {
  float h2, h6, h9;            /* compute.cpp:100-102 */
  double c_poly6;              /* 315/(64 pi h9)      compute.cpp:68 */
  double c_spiky_grad;         /* -45/(pi h6)         compute.cpp:56 */
  double c_visc_lapl;          /* 45/(pi h6)          compute.cpp:45 */
  double c_poly6_grad;         /* -945/(32 pi h9)     compute.cpp:80 */
  double c_poly6_lapl;         /* 945/(32 pi h9)      compute.cpp:92 */
  float r2_max;                /* largest fp32 x with sqrtf(x) <= h */
  float cell_size;             /* resolved cell edge */
  int ncell;                   /* cells per axis */
} sph_oracle_consts;

void sph_oracle_derive(const sph_params* p, sph_oracle_consts* c);

/* The five smoothing-kernel functions, compute.cpp:41-95. */
float sph_oracle_lapl_viscosity(const sph_params* p, float r);
float sph_oracle_grad_spiky(const sph_params* p, float r);
float sph_oracle_w_poly6(const sph_params* p, float r);
float sph_oracle_grad_poly6(const sph_params* p, float r);
float sph_oracle_lapl_poly6(const sph_params* p, float r);
/* compute.cpp:28-39 */
void sph_oracle_vec_normalize(const float* a, float* b);

/*
 * One all-pairs step, literal restatement of compute.cpp:120-320 (every j visited, zero
 * terms included, same accumulation order).  pos/vel: float[3n] AoS, updated in place.
 * rho_out[n] / force_out[3n] (may be NULL) receive `densities` and `f_total`.
 * OpenMP over i exactly where the reference has its pragmas; the result does not depend
 * on the thread count.
 */
void sph_oracle_step_allpairs(const sph_params* p, float* pos, float* vel, int n,
                              float* rho_out, float* force_out);

/* #{j : sqrtf(r2_ij) <= h}, self included, with the reference's r2 arithmetic (:140-144). */
void sph_oracle_neighbor_counts(const sph_params* p, const float* pos, int n, int32_t* out);

/* Linear cell id (cx*nc + cy)*nc + cz (z fastest, x slowest: x is the slab-decomposition axis)
 * with c = clamp(floorf(x / cell_size), 0, nc-1). */
void sph_oracle_cell_ids(const sph_params* p, const float* pos, int n, int32_t* out);

/*
 * Uniform-grid twin: same physics restricted to the 27-cell stencil.
 *   order = 0: neighbours are summed in stencil order (dx,dy outer; cells ascending in z;
 *              inside a cell the lower half in z before the upper half, ascending id inside a
 *              half) — the order of the HIP fast path;
 *   order = 1: ascending particle id — bit-identical to sph_oracle_step_allpairs.
 * ids[n] may be NULL (id = index).  ncount_out (may be NULL) gets neighbour counts.
 * Not the reference algorithm: it exists to debug kernels and to time "what a CPU grid
 * code would do" next to the all-pairs baseline.
 */
void sph_oracle_step_grid(const sph_params* p, float* pos, float* vel, const int32_t* ids,
                          int n, int order, float* rho_out, float* force_out,
                          int32_t* ncount_out);

int sph_oracle_num_threads(void);
void sph_oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
