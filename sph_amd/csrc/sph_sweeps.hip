// Neighbour sweeps for gfx950: density pass and force+integration pass of one SPH step
// (reference compute/compute.cpp:131-150 and :153-319).  Compile with -ffp-contract=off.
//
// Data the sweeps read is the grid-sorted copy of the state, packed for gathers: one float4 per particle
// for the position and one for velocity + density, so a neighbour costs one 16-byte load per array instead
// of one 4-byte load per component.  Cells are linearised as c = (cx*ny + cy)*nz + cz (z fastest; x, the
// slab axis, slowest): the 27-cell stencil of a particle is 9 runs of up to 3 z-adjacent cells, each run ONE
// contiguous slot range.  Runs are visited dx-major, dy-minor; inside a cell particles are in ascending id
// order.  That order is a pure function of (positions, ids), so results do not depend on how particles
// share waves, contexts or slabs.
//
// Three flavours, same results bit for bit except SWEEP_EXACT's summation order:
//   queued (default)  candidates that pass the distance test are pushed on a per-lane FIFO in LDS and the
//                     pair arithmetic runs when a queue is full, with most lanes of the wave busy.  The
//                     density pass also records WHICH candidates passed as a per-particle bitmask over the
//                     flattened stencil, so the force pass tests nothing: it walks the set bits;
//   plain             evaluates a neighbour on the spot (ablation);
//   exact             sums neighbours in ascending particle id, the order in which the reference's
//                     j = 0..N-1 loops meet the non-zero terms: bit-identical to the reference.
#include "sph_device.hpp"

namespace sph {

namespace {

constexpr int kBlock = 256;
constexpr int kQueue = 16;
constexpr int kAhead = 4;  // candidates fetched per loop trip: independent loads in flight per lane
constexpr int kMaskBits = 32 * kMaskWords;

__host__ inline int div_up(long long a, int b) { return (int)((a + b - 1) / b); }

// Workgroups are dealt round-robin to the 8 XCDs, each with a private L2.  Remap the block index so that every
// XCD works on runs of kXcdRun CONSECUTIVE blocks of the sorted particle range (a few y-rows of cells, whose
// stencils overlap) instead of every eighth block; the runs themselves stay interleaved over the XCDs so that
// crowded and empty regions of the tank are still spread evenly.  Bijective for any grid size (the tail that
// does not fill 8 runs keeps its index); affects locality only, never results.
constexpr int kXcdRun = 32;
__device__ inline int xcd_block(int b, int nb) {
  const int span = 8 * kXcdRun;
  if (b >= nb / span * span) return b;
  const int xcd = b & 7, i = b >> 3;
  return ((i / kXcdRun) * 8 + xcd) * kXcdRun + i % kXcdRun;
}

struct CellCoord {
  int x, y, z;
};
__device__ inline CellCoord decode_cell(const DevParams& P, int c) {
  CellCoord q;
  q.z = c % P.nz;
  int t = c / P.nz;
  q.y = t % P.ny;
  q.x = t / P.ny;
  return q;
}

struct RunCursor {
  int k, j, e;  // current run, next slot, end of run
};

// Advance to the next stencil run of cell cc; returns false past the last run.
__device__ inline bool next_run(const DevParams& P, const int32_t* __restrict__ start, const CellCoord cc,
                                RunCursor& c) {
  if (++c.k >= 9) return false;
  const int x = cc.x + c.k / 3 - 1, y = cc.y + c.k % 3 - 1;
  const bool inside = x >= 0 && x < P.nx && y >= 0 && y < P.ny;  // rows beyond the grid are empty runs
  const int row = inside ? (x * P.ny + y) * P.nz : 0;
  const int z0 = cc.z > 0 ? cc.z - 1 : 0, z1 = cc.z < P.nz - 1 ? cc.z + 1 : P.nz - 1;
  const int b = start[row + z0], e = start[row + z1 + 1];
  c.j = inside ? b : 0;
  c.e = inside ? e : 0;
  return true;
}

__device__ inline float dist2_to(const float4& a, const float4& b, float& dx, float& dy, float& dz) {
  dx = a.x - b.x;  // :140-142
  dy = a.y - b.y;
  dz = a.z - b.z;
  return dist2(dx, dy, dz);
}

// ---- density ----------------------------------------------------------------------------------------------------
// rho_i = sum_j m * Wpoly6(|r_i - r_j|), self term included (:131-150).  Written to vel[s].w.
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_density(DevParams P, Sorted st, int n, const int32_t* __restrict__ start,
                                                    const int32_t* __restrict__ cell_s,
                                                    int32_t* __restrict__ ncount, NeighbourMask nm) {
  __shared__ float queue[MODE == SWEEP_QUEUED ? kQueue : 1][kBlock];  // r2 of the queued neighbours
  // slots of the x-columns [dens_lo, dens_hi): own columns plus one ghost column on each side
  const int cpc = P.ny * P.nz;
  const int s = start[P.dens_lo * cpc] + xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (s >= start[P.dens_hi * cpc]) return;
  const float4 pi = st.pos[s];
  const CellCoord cc = decode_cell(P, cell_s[s]);
  float acc = 0.0f;
  int cnt = 0;
  float dx, dy, dz;
  if (MODE == SWEEP_QUEUED) {
    int fill = 0;
    auto drain = [&]() {
#pragma unroll 4
      for (int t = 0; t < kQueue; t++)
        if (t < fill) acc += P.m * w_poly6(P, sqrtf(queue[t][threadIdx.x]));  // :144-146
      cnt += fill;
      fill = 0;
    };
    // bit q of the mask <=> the q-th candidate of the flattened stencil is a neighbour
    unsigned word = 0;
    int q = 0;
    RunCursor c{-1, 0, 0};
    for (;;) {
      if (c.j >= c.e) {
        if (!next_run(P, start, cc, c)) break;
        continue;
      }
      const int m = c.e - c.j < kAhead ? c.e - c.j : kAhead;
      float4 cand[kAhead];
#pragma unroll
      for (int u = 0; u < kAhead; u++) cand[u] = st.pos[u < m ? c.j + u : c.j];
#pragma unroll
      for (int u = 0; u < kAhead; u++) {
        if (u < m) {
          const float r2 = dist2_to(pi, cand[u], dx, dy, dz);
          const bool pass = r2 <= P.r2_max;  // sqrtf(r2) <= h
          if (pass) queue[fill++][threadIdx.x] = r2;
          word |= (pass ? 1u : 0u) << (q & 31);
          q++;
          if ((q & 31) == 0) {
            if (q <= kMaskBits) nm.words[(size_t)(q / 32 - 1) * nm.stride + s] = word;
            word = 0;
          }
        }
      }
      c.j += m;
      if (__any(fill > kQueue - kAhead)) drain();
    }
    drain();
    if ((q & 31) != 0 && q < kMaskBits) nm.words[(size_t)(q / 32) * nm.stride + s] = word;
    nm.ncand[s] = q;
  } else if (MODE == SWEEP_PLAIN) {
    RunCursor c{-1, 0, 0};
    while (next_run(P, start, cc, c)) {
      for (int j = c.j; j < c.e; j++) {
        const float r2 = dist2_to(pi, st.pos[j], dx, dy, dz);
        if (r2 <= P.r2_max) {
          acc += P.m * w_poly6(P, sqrtf(r2));
          cnt++;
        }
      }
    }
  } else {  // SWEEP_EXACT
    int last = -1;
    for (;;) {
      int best = 0x7fffffff;
      float best_r2 = 0.0f;
      RunCursor c{-1, 0, 0};
      while (next_run(P, start, cc, c)) {
        for (int j = c.j; j < c.e; j++) {
          const float r2 = dist2_to(pi, st.pos[j], dx, dy, dz);
          const int idj = st.id[j];
          if (r2 <= P.r2_max && idj > last && idj < best) {
            best = idj;
            best_r2 = r2;
          }
        }
      }
      if (best == 0x7fffffff) break;
      acc += P.m * w_poly6(P, sqrtf(best_r2));
      cnt++;
      last = best;
    }
  }
  st.vel[s].w = acc;
  if (ncount) ncount[s] = cnt;
}

// ---- forces, integration, walls -------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_force(DevParams P, Sorted st, State out, int n,
                                                  const int32_t* __restrict__ start,
                                                  const int32_t* __restrict__ cell_s, float* __restrict__ force_dbg,
                                                  int32_t* __restrict__ n_own_out, NeighbourMask nm) {
  __shared__ int queue[MODE == SWEEP_QUEUED ? kQueue : 1][kBlock];  // slot indices of the queued neighbours
  // slots of the own x-columns [own_lo, own_hi); the new state is written compactly from index 0
  const int cpc = P.ny * P.nz;
  const int base = start[P.own_lo * cpc], bound = start[P.own_hi * cpc];
  const int o = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (o == 0 && n_own_out) {
    const int real = P.nx * cpc;
    n_own_out[0] = bound - base;                   // particles this context keeps
    n_own_out[1] = start[real + 1] - start[real];  // trash cell: particles outside the local grid
  }
  const int s = base + o;
  if (s >= bound) return;
  const float4 pi4 = st.pos[s];
  const float4 vi4 = st.vel[s];  // w = density
  const float rho_i = vi4.w;
  const float pi = P.c2 * (rho_i - P.rho0);  // :158
  const CellCoord cc = decode_cell(P, cell_s[s]);
  float a[kForceComps] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float dx, dy, dz;

  auto add_neighbour = [&](int j) {
    const float r2 = dist2_to(pi4, st.pos[j], dx, dy, dz);
    const float4 vj = st.vel[j];
    const PairOut w = pair_eval(P, dx, dy, dz, r2, pi, vj.w, vj.x - vi4.x, vj.y - vi4.y, vj.z - vi4.z);
#pragma unroll
    for (int q = 0; q < kForceComps; q++) a[q] += w.v[q];
  };

  const int ncand = MODE == SWEEP_QUEUED ? nm.ncand[s] : 0;
  if (MODE == SWEEP_QUEUED && ncand <= kMaskBits) {
    // Walk the set bits of the density pass's neighbour mask: no distance tests, every trip is a neighbour.
    RunCursor c{-1, 0, 0};
    int run_q = 0;  // flattened index of the first candidate of the current run
    int w = -1;
    unsigned m = 0;
    const int nwords = (ncand + 31) >> 5;
    for (;;) {
      if (m == 0) {
        if (++w >= nwords) break;
        m = nm.words[(size_t)w * nm.stride + s];
        continue;
      }
      const int q = (w << 5) + __ffs(m) - 1;
      m &= m - 1;
      while (q >= run_q + (c.e - c.j)) {  // runs are met in ascending order
        run_q += c.e - c.j;
        next_run(P, start, cc, c);
      }
      add_neighbour(c.j + (q - run_q));
    }
  } else if (MODE == SWEEP_QUEUED) {
    // more candidates than mask bits (extremely crowded cells): test again, queueing the hits
    int fill = 0;
    auto drain = [&]() {
#pragma unroll 1
      for (int t = 0; t < kQueue; t++)
        if (t < fill) add_neighbour(queue[t][threadIdx.x]);
      fill = 0;
    };
    RunCursor c{-1, 0, 0};
    for (;;) {
      if (c.j >= c.e) {
        if (!next_run(P, start, cc, c)) break;
        continue;
      }
      const int m = c.e - c.j < kAhead ? c.e - c.j : kAhead;
      float4 cand[kAhead];
#pragma unroll
      for (int u = 0; u < kAhead; u++) cand[u] = st.pos[u < m ? c.j + u : c.j];
#pragma unroll
      for (int u = 0; u < kAhead; u++) {
        const float r2 = dist2_to(pi4, cand[u], dx, dy, dz);
        if (u < m && r2 <= P.r2_max) queue[fill++][threadIdx.x] = c.j + u;
      }
      c.j += m;
      if (__any(fill > kQueue - kAhead)) drain();
    }
    drain();
  } else if (MODE == SWEEP_PLAIN) {
    RunCursor c{-1, 0, 0};
    while (next_run(P, start, cc, c)) {
      for (int j = c.j; j < c.e; j++) {
        const float r2 = dist2_to(pi4, st.pos[j], dx, dy, dz);
        if (r2 <= P.r2_max) add_neighbour(j);
      }
    }
  } else {  // SWEEP_EXACT
    int last = -1;
    for (;;) {
      int best = 0x7fffffff, bj = -1;
      RunCursor c{-1, 0, 0};
      while (next_run(P, start, cc, c)) {
        for (int j = c.j; j < c.e; j++) {
          const float r2 = dist2_to(pi4, st.pos[j], dx, dy, dz);
          const int idj = st.id[j];
          if (r2 <= P.r2_max && idj > last && idj < best) {
            best = idj;
            bj = j;
          }
        }
      }
      if (bj < 0) break;
      add_neighbour(bj);
      last = best;
    }
  }

  const Advanced w = finish_particle(P, a, pi4.x, pi4.y, pi4.z, vi4.x, vi4.y, vi4.z, rho_i);
  if (force_dbg) {
    force_dbg[o] = w.fx;
    force_dbg[(size_t)n + o] = w.fy;
    force_dbg[2 * (size_t)n + o] = w.fz;
  }
  out.x[o] = w.x;
  out.y[o] = w.y;
  out.z[o] = w.z;
  out.vx[o] = w.vx;
  out.vy[o] = w.vy;
  out.vz[o] = w.vz;
  if (out.id) out.id[o] = st.id[s];
}

}  // namespace

void launch_density(hipStream_t s, const DevParams& P, const Sorted& st, int n, const GridBuffers& g,
                    int32_t* ncount, const NeighbourMask& nm, SweepMode mode) {
  if (n <= 0) return;
  dim3 grid(div_up(n, kBlock)), block(kBlock);
  if (mode == SWEEP_EXACT)
    hipLaunchKernelGGL(k_density<SWEEP_EXACT>, grid, block, 0, s, P, st, n, g.start, g.cell_s, ncount, nm);
  else if (mode == SWEEP_PLAIN)
    hipLaunchKernelGGL(k_density<SWEEP_PLAIN>, grid, block, 0, s, P, st, n, g.start, g.cell_s, ncount, nm);
  else
    hipLaunchKernelGGL(k_density<SWEEP_QUEUED>, grid, block, 0, s, P, st, n, g.start, g.cell_s, ncount, nm);
}

void launch_force(hipStream_t s, const DevParams& P, const Sorted& st, const State& out, int n,
                  const GridBuffers& g, float* force_dbg, int32_t* n_own_out, const NeighbourMask& nm,
                  SweepMode mode) {
  if (n <= 0) return;
  dim3 grid(div_up(n, kBlock)), block(kBlock);
  if (mode == SWEEP_EXACT)
    hipLaunchKernelGGL(k_force<SWEEP_EXACT>, grid, block, 0, s, P, st, out, n, g.start, g.cell_s, force_dbg, n_own_out, nm);
  else if (mode == SWEEP_PLAIN)
    hipLaunchKernelGGL(k_force<SWEEP_PLAIN>, grid, block, 0, s, P, st, out, n, g.start, g.cell_s, force_dbg, n_own_out, nm);
  else
    hipLaunchKernelGGL(k_force<SWEEP_QUEUED>, grid, block, 0, s, P, st, out, n, g.start, g.cell_s, force_dbg, n_own_out, nm);
}

}  // namespace sph
