// Internal interface between the C ABI (sph_api.cpp) and the gfx950 kernels (sph_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sph_amd.h"

namespace sph {

// Everything a kernel needs, passed by value as a kernel argument.
// The coefficients are doubles on purpose: the reference forms coefficient*polynomial in
// double (M_PI is a double) and rounds once to float (reference compute.cpp:45,56,68,80,92).
struct DevParams {
  double c_poly6, c_spiky_grad, c_visc_lapl, c_poly6_grad, c_poly6_lapl;
  float h, h2, r2_max;       // r2_max = max{x : sqrtf(x) <= h}: the reference's `r <= h` on r2
  float m, neg_m, rho0, c2;  // c2 = soundSpeed^2 (compute.cpp:153)
  float visc_m;              // dynVisc*m, the left-most product of compute.cpp:204
  float gx, gy, gz, dt, tank, neg_sigma;
  float cell;                // cell edge
  int nx, ny, nz;            // cells per axis of the local grid
  // c / ny and c / nz for 0 <= c < 2^31 as a multiply-high and a shift (division_constants()); unused when the
  // divisor is 1
  uint32_t ny_mul, ny_shift, nz_mul, nz_shift;
  int cx_off;                // global x-column of local column 0 (slab decomposition)
  int gnx;                   // global cells per axis (for clamping before the offset)
  int own_lo, own_hi;        // local x-columns this context integrates: [own_lo, own_hi)
  int force_lo, force_hi;    // local x-columns one launch of the force pass advances: all own columns, or a part of them
                             // (faces first, interior while the halo records travel)
  int dens_lo, dens_hi;      // local x-columns whose density is needed: own plus one ghost column per side
  int tension;               // surface-tension term on/off
  // fp32 products of mass/viscosity and the kernel coefficients, for the fast-arithmetic sweeps (rounded once on
  // the host from the double coefficients above)
  float k_dens;              // m * c_poly6                      density, compute.cpp:68,146
  float k_press;             // -m/2 * c_spiky_grad              pressure, :56,:170
  float k_visc;              // dynVisc * m * c_visc_lapl        viscosity, :45,:204
  float k_tgrad;             // m * c_poly6_grad                 colour gradient, :80,:242
  float k_tlapl;             // m * c_poly6_lapl                 colour laplacian, :92,:252
  float c2rho0;              // c2 * rho0
};

// Constants that turn c / d, 0 <= c < 2^31, d >= 2, into (uint32)(((uint64)c * mul) >> 32) >> shift: with
// s = ceil(log2 d), mul = floor(2^(31+s) / d) + 1 fits 32 bits and overshoots 2^(31+s) / d by less than 2^-31 of it.
inline void division_constants(uint32_t d, uint32_t* mul, uint32_t* shift) {
  uint32_t s = 0;
  while ((1ull << s) < d) s++;
  *mul = d < 2 ? 0u : (uint32_t)(((1ull << (31 + s)) / d) + 1);
  *shift = d < 2 ? 0u : s - 1;
}

// Structure-of-arrays particle state in HBM.
struct State {
  float *x, *y, *z, *vx, *vy, *vz;
  int32_t* id;
};

// Grid-sorted working copy of the state, laid out for neighbour gathers.  The search and the density pass
// read positions only: 16 bytes per candidate from `pos`.  The force pass needs everything about a neighbour:
// `rec` holds position and velocity + density side by side, so one 128-byte line serves both loads (with two
// separate arrays the pass fetched twice as many lines into L1, which bounds it where the fluid is compressed).
struct ParticleRec {
  float4 pos;  // x, y, z, 1/density (w is written by the fast-arithmetic density pass only)
  float4 vel;  // vx, vy, vz, density (w is written by the density pass)
};
// The neighbour search reads up to kSearchOverread slots past the end of a stencil run (and masks them off), so
// `pos` is allocated that much longer than the particle capacity.
constexpr int kSearchOverread = 4;
// Stencil runs are clipped by half cells (see run_ends() in sph_sweeps.hip) only for particles of cells that hold
// kClipFrom particles or more.  The counting sort knows a cell's size when it places a particle, and leaves the
// verdict in bit 31 of the particle's entry of the sorted cell array: the sweeps then need no look at the start table
// before they can ask for their eighteen run ends.
#ifndef SPH_CLIP_FROM
#define SPH_CLIP_FROM 8
#endif
constexpr int kClipFrom = SPH_CLIP_FROM;
constexpr int32_t kCellClipBit = (int32_t)0x80000000u;
constexpr int32_t kCellMask = 0x7fffffff;
struct Sorted {
  float* pos3;  // x y z packed, 12 bytes per particle: four candidates of the search are three 16-byte loads
  ParticleRec* rec;
  int32_t* id;
};

// Which candidates of a particle's flattened 27-cell stencil are neighbours (distance test passed), one
// bit per candidate, written by the neighbour search and walked by the density and force passes.  Word w of particle slot s
// lives at words[w * stride + s]; ncand[s] is the number of candidates (bits) of slot s, nnb[s] the number
// of neighbours (set bits).
constexpr int kMaskWords = 64;
struct NeighbourMask {
  uint32_t* words;
  int32_t* ncand;
  int32_t* nnb;
  size_t stride;
};

struct GridBuffers {
  // ncells counts the real cells PLUS one trailing trash cell for particles outside the local grid.
  int32_t* count;      // [ncells]   histogram, left zeroed by the scan
  int32_t* start;      // [ncells+1] exclusive prefix sum: cell c owns slots [start[c], start[c+1])
  int32_t* mid;        // [ncells]   first slot of cell c's particles in the upper half of the cell in z: inside a cell
                       //            the lower half comes first (ascending id), then the upper half (ascending id)
  int32_t* block_sums; // scan scratch
  int32_t* cell_of;    // [n] cell of particle i (pre-sort order)
  int32_t* rank_of;    // [n] arrival rank of particle i in its cell (arbitrary, from the atomic)
  int32_t* src_of;     // [n] provisional slot -> pre-sort index
  int32_t* id_prov;    // [n] provisional slot -> particle id
  int32_t* cell_s;     // [n] sorted slot -> cell; bit 31 (kCellClipBit): the cell holds kClipFrom particles or more
  size_t ncells;
  size_t scan_blocks;
};

struct DebugBuffers {
  float* force;     // [3n] SoA planes fx|fy|fz, sorted order
  int32_t* ncount;  // [n]
};

constexpr int kScanItems = 4096;  // cells per scan block

// Flavour of the density/force sweeps.  All but SWEEP_PLAIN walk the neighbour bitmask written by the search kernel.
// SWEEP_FAST is the default: fp32 coefficients and hardware reciprocals in the pair arithmetic.  The other three
// evaluate every term with the reference's arithmetic (double coefficient products, IEEE division and square root):
// MASKED in stencil order (bit-identical to the CPU grid twin), PLAIN the same without the mask (ablation), EXACT in
// ascending particle id (bit-identical to the reference).
enum SweepMode { SWEEP_MASKED = 0, SWEEP_PLAIN = 1, SWEEP_EXACT = 2, SWEEP_FAST = 3 };

void launch_split_aos(hipStream_t s, const float* pos_aos, const float* vel_aos, int n, State st);
void launch_hash(hipStream_t s, const DevParams& P, const State& in, int n, GridBuffers g);
void launch_scan(hipStream_t s, GridBuffers g);
void launch_reorder(hipStream_t s, const State& in, const Sorted& out, int n, GridBuffers g);
void launch_neighbours(hipStream_t s, const DevParams& P, const Sorted& st, int n, const GridBuffers& g,
                       const NeighbourMask& nm, bool with_density, int32_t* ncount);
void launch_density(hipStream_t s, const DevParams& P, const Sorted& st, int n, const GridBuffers& g,
                    int32_t* ncount, const NeighbourMask& nm, SweepMode mode);
void launch_force(hipStream_t s, const DevParams& P, const Sorted& st, const State& out, int n,
                  const GridBuffers& g, float* force_dbg, int32_t* n_own_out, const NeighbourMask& nm,
                  SweepMode mode);
void launch_slab_pack(hipStream_t s, const DevParams& P, const State& st, int n, const int32_t* n_dev, int col_lo,
                      int col_hi, bool has_left, bool has_right, float* rec_left, float* rec_right, int cap,
                      int32_t* counters);
// Overlapped slab step: records from the freshly advanced particles of the local columns [a_lo, a_hi) and [b_lo, b_hi)
// only (the columns next to the faces), written behind a header record that carries the count; and the check that no
// particle of the remaining own columns ended up where a neighbour would have needed it.
void launch_slab_pack_faces(hipStream_t s, const DevParams& P, const State& st, int n_bound, const int32_t* start,
                            int a_lo, int a_hi, int b_lo, int b_hi, int col_lo, int col_hi, bool has_left,
                            bool has_right, float* msg_left, float* msg_right, int cap, int32_t* counters);
void launch_slab_headers(hipStream_t s, const int32_t* counters, float* msg_left, float* msg_right);
void launch_slab_check_interior(hipStream_t s, const DevParams& P, const State& st, int n_bound, const int32_t* start,
                                int lo, int hi, int col_lo, int col_hi, bool has_left, bool has_right,
                                int32_t* missed);

void launch_slab_append(hipStream_t s, const DevParams& P, const float* rec, int n_rec, const State& st, int at,
                        int32_t* stray);
void launch_max_column_shift(hipStream_t s, const DevParams& P, const State& st, int n, int32_t* out);
void launch_split_aos_ids(hipStream_t s, const float* pos, const float* vel, const int32_t* ids, int n, State st);
void launch_join_aos(hipStream_t s, const State& st, int n, float* pos, float* vel, int32_t* ids);
void launch_fill_lattice(hipStream_t s, const State& st, long long first, int n, int ny, int nz, double spacing,
                         double jitter, unsigned long long seed, const double origin[3]);
void launch_column_counts(hipStream_t s, const int32_t* start, int nx, int cpc, int32_t* out);
void launch_column_work(hipStream_t s, const int32_t* start, int col_lo, int col_hi, int cpc, const int32_t* nnb,
                        int32_t* out);

// Camera of a rendered frame in the form the splat kernel uses: view = m * position + t.
struct RenderCamera {
  float m[3][3], t[3];
  float focal, aspect;  // 1 / tan(fov / 2), width / height
  float z_near, z_far, radius;
};
void launch_render(hipStream_t s, const State& st, int n, const RenderCamera& cam, int width, int height,
                   unsigned long long* fb, unsigned char* rgb);
void launch_join_debug(hipStream_t s, const DevParams& P, const int32_t* start, int n_own, int n_stride,
                       const ParticleRec* rec_s, const float* force_o, const int32_t* ncount_s, const int32_t* cell_s,
                       float* rho, float* force, int32_t* ncount, int32_t* cell);
void launch_gather_aos(hipStream_t s, const State& st, int n, float* pos_aos, float* vel_aos);
void launch_gather_debug(hipStream_t s, const int32_t* id, int n, const ParticleRec* rec_s,
                         const float* force_s, const int32_t* ncount_s, const int32_t* cell_s,
                         float* rho, float* force, int32_t* ncount, int32_t* cell);

}  // namespace sph
