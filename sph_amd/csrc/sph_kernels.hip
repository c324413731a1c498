// gfx950 kernels of the SPH step.  Compile with -ffp-contract=off: the reference's x86-64 object
// contains no FMA instructions, and the neighbour predicate / cell index must match it bit for bit.
//
// Pipeline of one step (reference compute/compute.cpp:120-320 restricted to true neighbours):
//   k_hash      position -> cell id, per-cell histogram
//   k_scan_*    exclusive prefix sum over the histogram -> cell start table
//   k_scatter / k_reorder   counting sort; inside a cell particles end up in ascending id order,
//               so the summation order of every later pass is a pure function of (positions, ids)
//   density and force sweeps: sph_sweeps.hip
#include "sph_device.hpp"

namespace sph {

namespace {

constexpr int kBlock = 256;

__host__ __device__ inline int div_up(long long a, int b) { return (int)((a + b - 1) / b); }

// ---- AoS <-> SoA ------------------------------------------------------------------------------------
__global__ void k_split_aos(const float* __restrict__ pos, const float* __restrict__ vel, int n, State st) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st.x[i] = pos[3 * i + 0];
  st.y[i] = pos[3 * i + 1];
  st.z[i] = pos[3 * i + 2];
  st.vx[i] = vel[3 * i + 0];
  st.vy[i] = vel[3 * i + 1];
  st.vz[i] = vel[3 * i + 2];
  st.id[i] = i;
}

__global__ void k_gather_aos(State st, int n, float* __restrict__ pos, float* __restrict__ vel) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  int i = st.id[s];
  if (pos) {
    pos[3 * i + 0] = st.x[s];
    pos[3 * i + 1] = st.y[s];
    pos[3 * i + 2] = st.z[s];
  }
  if (vel) {
    vel[3 * i + 0] = st.vx[s];
    vel[3 * i + 1] = st.vy[s];
    vel[3 * i + 2] = st.vz[s];
  }
}

__global__ void k_gather_debug(const int32_t* __restrict__ id, int n, const ParticleRec* rec_s,
                               const float* force_s, const int32_t* ncount_s, const int32_t* cell_s,
                               float* rho, float* force, int32_t* ncount, int32_t* cell) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  int i = id[s];
  if (rho) rho[i] = rec_s[s].vel.w;
  if (force) {
    force[3 * i + 0] = force_s[s];
    force[3 * i + 1] = force_s[(size_t)n + s];
    force[3 * i + 2] = force_s[2 * (size_t)n + s];
  }
  if (ncount) ncount[i] = ncount_s[s];
  if (cell) cell[i] = cell_s[s] & kCellMask;
}

// ---- hash + histogram ---------------------------------------------------------------------------------
// Global atomics are the expensive part of a histogram on this chip (they are served at the memory side of the
// fabric), so each WAVE first counts its 64 particles per cell in a small LDS table of its own (open addressing, LDS
// atomics; a run of equal cells in consecutive lanes is entered once, by its first lane) and then issues ONE global
// atomic per distinct cell.  The state arrives in the previous step's cell order, so a wave meets few distinct cells,
// however much the order has been stirred since.  Waves never wait for one another: no block barrier.  The ranks are
// provisional either way; k_reorder replaces them by the rank of the id within the cell.
constexpr int kWaveSlots = 128;  // 64 keys at most: load factor <= 1/2
__device__ inline void wave_lds_fence() {
  // LDS operations of one wave execute in program order; this only stops the compiler from moving them
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__global__ void __launch_bounds__(kBlock) k_hash(DevParams P, State in, int n, int32_t* __restrict__ cell_of,
                                                 int32_t* __restrict__ rank_of, int32_t* __restrict__ count) {
  __shared__ int key_all[kBlock / 64][kWaveSlots];
  __shared__ int cnt_all[kBlock / 64][kWaveSlots];  // particles of the wave in that cell; afterwards the base rank
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int* key = key_all[threadIdx.x >> 6];
  int* cnt = cnt_all[threadIdx.x >> 6];
  key[lane] = -1; key[lane + 64] = -1;
  cnt[lane] = 0; cnt[lane + 64] = 0;
  const bool valid = i < n;
  int c = -1 - lane;  // lanes without a particle: distinct keys, never entered
  int upper = 0;      // which half of its cell (in z) the particle is in: bit 31 of its rank word
  if (valid) {
    // A particle whose x-column lies outside this context's local grid (possible only in slab mode) goes to
    // the trash cell appended after the last real cell: it sorts behind everything and no pass visits it.
    const int cx = cell_coord(in.x[i], P.cell, P.gnx) - P.cx_off;
    const int cy = cell_coord(in.y[i], P.cell, P.ny);
    const int cz = cell_coord(in.z[i], P.cell, P.nz);
    c = (cx < 0 || cx >= P.nx) ? P.nx * P.ny * P.nz : (cx * P.ny + cy) * P.nz + cz;
    upper = upper_half(P, in.z[i]);
  }
  const int prev = __shfl_up(c, 1, 64);
  const bool head = lane == 0 || prev != c;
  const unsigned long long heads = __ballot(head);
  const unsigned long long upto = (2ull << lane) - 1ull;  // bits 0..lane (all 64 for lane 63)
  const int head_lane = 63 - __clzll((long long)(heads & upto));
  const unsigned long long above = heads & ~upto;
  const int next_head = above ? __ffsll((long long)above) - 1 : 64;
  wave_lds_fence();
  int at = 0, local = 0;
  if (head && valid) {
    at = (int)(((uint32_t)c * 0x9E3779B1u) >> 25) & (kWaveSlots - 1);
    for (;;) {  // at most 64 distinct keys in 128 slots: always terminates
      const int old = atomicCAS(&key[at], -1, c);
      if (old == -1 || old == c) break;
      at = (at + 1) & (kWaveSlots - 1);
    }
    local = atomicAdd(&cnt[at], next_head - head_lane);
  }
  wave_lds_fence();
#pragma unroll
  for (int t = lane; t < kWaveSlots; t += 64) {
    const int k = key[t];
    if (k >= 0) cnt[t] = atomicAdd(&count[k], cnt[t]);
  }
  wave_lds_fence();
  at = __shfl(at, head_lane, 64);
  local = __shfl(local, head_lane, 64);
  if (valid) {
    cell_of[i] = c;
    rank_of[i] = (cnt[at] + local + lane - head_lane) | (upper << 31);
  }
}

// ---- exclusive scan over the cell histogram -----------------------------------------------------------
__device__ inline int wave_inclusive_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// Exclusive scan of one value per thread across a 256-thread block; returns the block total in *total.
__device__ inline int block_exclusive_scan(int v, int* total) {
  __shared__ int wsum[kBlock / 64];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = wave_inclusive_scan(v, lane);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kBlock / 64; k++) {
    int s = wsum[k];
    if (k < w) off += s;
    tot += s;
  }
  *total = tot;
  __syncthreads();
  return off + inc - v;
}

constexpr int kPerThread = kScanItems / kBlock;  // 16 cells per thread

__global__ void __launch_bounds__(kBlock) k_scan_reduce(const int32_t* __restrict__ count, size_t ncells,
                                                        int32_t* __restrict__ block_sums) {
  size_t base = (size_t)blockIdx.x * kScanItems + (size_t)threadIdx.x * kPerThread;
  int s = 0;
  if (base + kPerThread <= ncells) {
    const int4* p = reinterpret_cast<const int4*>(count + base);
#pragma unroll
    for (int k = 0; k < kPerThread / 4; k++) {
      int4 v = p[k];
      s += v.x + v.y + v.z + v.w;
    }
  } else {
    for (int k = 0; k < kPerThread; k++)
      if (base + k < ncells) s += count[base + k];
  }
  int tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// Single block: in-place exclusive scan of the per-block sums.
__global__ void __launch_bounds__(kBlock) k_scan_top(int32_t* __restrict__ block_sums, int nb) {
  int carry = 0;
  for (int base = 0; base < nb; base += kBlock) {
    int i = base + threadIdx.x;
    int v = i < nb ? block_sums[i] : 0;
    int tot;
    int ex = block_exclusive_scan(v, &tot);
    if (i < nb) block_sums[i] = carry + ex;
    carry += tot;
  }
}

// Writes start[] and leaves count[] zeroed for the next step's histogram.  With sums_are_raw the per-block sums
// have not been through k_scan_top: every block adds up the sums of the blocks before it itself (a few KB from
// L2 on grids of up to kFusedTopBlocks tiles), which saves a launch whose single block only waits on memory.
constexpr int kFusedTopBlocks = 2048;
__global__ void __launch_bounds__(kBlock) k_scan_final(int32_t* __restrict__ count, size_t ncells,
                                                       const int32_t* __restrict__ block_sums,
                                                       int32_t* __restrict__ start, int32_t* __restrict__ mid,
                                                       int sums_are_raw) {
  size_t base = (size_t)blockIdx.x * kScanItems + (size_t)threadIdx.x * kPerThread;
  int v[kPerThread];
  int s = 0;
  bool full = base + kPerThread <= ncells;
  // A tile without particles (most of a tank whose fluid fills a corner): its raw sum says so, and then its cells
  // need neither be read nor re-zeroed - only their start entries written, all equal.
  const bool empty_tile = sums_are_raw && block_sums[blockIdx.x] == 0;
  if (empty_tile) {
#pragma unroll
    for (int k = 0; k < kPerThread; k++) v[k] = 0;
  } else if (full) {
    int4* p = reinterpret_cast<int4*>(count + base);
#pragma unroll
    for (int k = 0; k < kPerThread / 4; k++) {
      int4 q = p[k];
      v[4 * k + 0] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
      p[k] = make_int4(0, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int k = 0; k < kPerThread; k++) {
      v[k] = base + k < ncells ? count[base + k] : 0;
      if (base + k < ncells) count[base + k] = 0;
    }
  }
#pragma unroll
  for (int k = 0; k < kPerThread; k++) s += v[k];
  int tot;
  int before;
  if (sums_are_raw) {
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += kBlock) part += block_sums[b];
    block_exclusive_scan(part, &before);
  } else {
    before = block_sums[blockIdx.x];
  }
  int run = before + block_exclusive_scan(s, &tot);
  // mid[] starts out equal to start[]: k_reorder overwrites the entries of the cells that hold particles
  if (full) {
    int4* o = reinterpret_cast<int4*>(start + base);
    int4* om = reinterpret_cast<int4*>(mid + base);
#pragma unroll
    for (int k = 0; k < kPerThread / 4; k++) {
      int4 q;
      q.x = run; run += v[4 * k + 0];
      q.y = run; run += v[4 * k + 1];
      q.z = run; run += v[4 * k + 2];
      q.w = run; run += v[4 * k + 3];
      o[k] = q;
      om[k] = q;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kPerThread; k++) {
      if (base + k < ncells) {
        start[base + k] = run;
        mid[base + k] = run;
      }
      run += v[k];
    }
  }
  // the thread that owns the last cell also writes the grand total
  if (base < ncells && base + kPerThread >= ncells) start[ncells] = run;
}

// ---- counting sort ---------------------------------------------------------------------------------------
// Provisional placement: particle i goes to the slot its arrival rank gives it inside its cell's range.  What lands
// there is what the final pass needs in slot order: where the particle came from, its id and its cell (the cell of a
// slot is final already: the final pass only permutes particles inside their cell's range).
__global__ void k_scatter(int n, const int32_t* __restrict__ cell_of, const int32_t* __restrict__ rank_of,
                          const int32_t* __restrict__ start, const int32_t* __restrict__ id_in,
                          int32_t* __restrict__ src_of, int32_t* __restrict__ id_prov, int32_t* __restrict__ cell_s) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = cell_of[i];
  const int rw = rank_of[i];  // bit 31: upper half of the cell
  const int b = start[c], e = start[c + 1];
  const int slot = b + (rw & 0x7fffffff);
  src_of[slot] = i;
  id_prov[slot] = id_in[i] | (rw & (int)0x80000000u);  // ids are below 2^31: the sort key is (half, id)
  cell_s[slot] = c | (e - b >= kClipFrom ? kCellClipBit : 0);  // cells are counted in 31 bits; bit 31: runs clip
}

// Final slot = cell start + number of cell-mates with a smaller key, the key being (half of the cell in z, id): the
// in-cell order no longer depends on the order in which the histogram atomics happened to be served.  The lanes of a
// wave are consecutive slots, hence mostly cell-mates that count over the same keys: those loads are one request per
// wave.  The cell's first particle in the new order also records where the cell's upper half begins.
//
// The final slot lies in the particle's own cell, so all but the particles of the two cells that straddle the block's
// ends stay inside the block's own 256 slots: they are put in order in LDS and leave as whole lines (per lane a
// 12-byte, a 32-byte and a 4-byte piece at its own stride would be five partial-line stores per particle).  A slot
// of the block's range that is filled by a neighbouring block stays marked empty here and is not written.
#ifndef SPH_REORDER_STAGED
#define SPH_REORDER_STAGED 1
#endif
__global__ void __launch_bounds__(kBlock) k_reorder(int n, State in, Sorted out, const int32_t* __restrict__ start,
                                                    int32_t* __restrict__ mid, const int32_t* __restrict__ src_of,
                                                    const int32_t* __restrict__ id_prov,
                                                    const int32_t* __restrict__ cell_s) {
  __shared__ float s_pos3[3 * kBlock];
  __shared__ float4 s_rec[2 * kBlock];
  __shared__ int s_id[kBlock];  // -1: no particle of this block goes there
  const int base = blockIdx.x * kBlock;
  const int s = base + threadIdx.x;
  s_id[threadIdx.x] = -1;
  __syncthreads();
  if (s < n) {
    const int i = src_of[s];
    const unsigned my = (unsigned)id_prov[s];
    const int c = cell_s[s] & kCellMask;
    const int b = start[c], e = start[c + 1];
    const float4 pos = make_float4(in.x[i], in.y[i], in.z[i], 0.0f);  // w receives 1/density
    const float4 vel = make_float4(in.vx[i], in.vy[i], in.vz[i], 0.0f);  // w receives the density
    int r = 0, uppers = 0;
    int t = b;
    for (; t + 4 <= e; t += 4) {  // four independent loads per trip
      const unsigned a0 = (unsigned)id_prov[t], a1 = (unsigned)id_prov[t + 1], a2 = (unsigned)id_prov[t + 2],
                     a3 = (unsigned)id_prov[t + 3];
      r += (a0 < my) + (a1 < my) + (a2 < my) + (a3 < my);
      uppers += (int)((a0 >> 31) + (a1 >> 31) + (a2 >> 31) + (a3 >> 31));
    }
    for (; t < e; t++) {
      const unsigned a = (unsigned)id_prov[t];
      r += a < my;
      uppers += (int)(a >> 31);
    }
    const int d = b + r;
    if (r == 0) mid[c] = e - uppers;
    const int id = (int)(my & 0x7fffffffu);
    const unsigned l = (unsigned)(d - base);
    if (SPH_REORDER_STAGED && l < (unsigned)kBlock) {
      s_pos3[3 * l + 0] = pos.x;
      s_pos3[3 * l + 1] = pos.y;
      s_pos3[3 * l + 2] = pos.z;
      s_rec[2 * l + 0] = pos;
      s_rec[2 * l + 1] = vel;
      s_id[l] = id;
    } else {
      out.pos3[3 * (size_t)d + 0] = pos.x;
      out.pos3[3 * (size_t)d + 1] = pos.y;
      out.pos3[3 * (size_t)d + 2] = pos.z;
      out.rec[d].pos = pos;
      out.rec[d].vel = vel;
      out.id[d] = id;
    }
  }
  if (!SPH_REORDER_STAGED) return;
  __syncthreads();
  const int mine = s_id[threadIdx.x];
  if (mine >= 0) out.id[s] = mine;
  float* pos3_out = out.pos3 + 3 * (size_t)base;
  float4* rec_out = reinterpret_cast<float4*>(out.rec + base);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int w = k * kBlock + threadIdx.x;
    if (s_id[(w * 21846) >> 16] >= 0) pos3_out[w] = s_pos3[w];  // w / 3 for w < 768
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int w = k * kBlock + threadIdx.x;
    if (s_id[w >> 1] >= 0) rec_out[w] = s_rec[w];
  }
}

// ---- slab decomposition: halo/migration records ---------------------------------------------------------------
// A record is 7 words: x, y, z, vx, vy, vz, id (id bit-cast).  Records travel between neighbouring slabs.
constexpr int kRecWords = 7;

// Every particle whose x-column is within two columns of (or beyond) a slab face is copied to the buffer
// for the neighbour behind that face: it is either a migrant (new owner: the neighbour) or a ghost for it.
// n_dev, when given, holds the true particle count on the device (the launch then covers an upper bound n).
__global__ void __launch_bounds__(kBlock) k_slab_pack(DevParams P, State st, int n, const int32_t* __restrict__ n_dev,
                                                      int col_lo, int col_hi, int has_left, int has_right,
                                                      float* __restrict__ rec_left, float* __restrict__ rec_right,
                                                      int cap, int32_t* __restrict__ counters) {
  __shared__ int block_base[2];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int count = n_dev ? *n_dev : n;
  if ((int)(blockIdx.x * blockDim.x) >= count) return;  // whole block beyond the particles
  bool to_left = false, to_right = false;
  if (i < count) {
    const int g = cell_coord(st.x[i], P.cell, P.gnx);
    to_left = has_left && g < col_lo + 2;
    to_right = has_right && g >= col_hi - 2;
  }
  // one global atomic per block and side: the block's records take consecutive places in the buffer
  int n_left, n_right;
  const int at_left = block_exclusive_scan(to_left ? 1 : 0, &n_left);
  const int at_right = block_exclusive_scan(to_right ? 1 : 0, &n_right);
  if (threadIdx.x == 0) {
    block_base[0] = n_left ? atomicAdd(&counters[0], n_left) : 0;
    block_base[1] = n_right ? atomicAdd(&counters[1], n_right) : 0;
  }
  __syncthreads();
  if (!to_left && !to_right) return;
  const float r[kRecWords] = {st.x[i], st.y[i], st.z[i], st.vx[i], st.vy[i], st.vz[i], __int_as_float(st.id[i])};
  if (to_left) {
    const int k = block_base[0] + at_left;
    if (k < cap)
      for (int w = 0; w < kRecWords; w++) rec_left[(size_t)k * kRecWords + w] = r[w];
  }
  if (to_right) {
    const int k = block_base[1] + at_right;
    if (k < cap)
      for (int w = 0; w < kRecWords; w++) rec_right[(size_t)k * kRecWords + w] = r[w];
  }
}

// Overlapped slab step.  The force pass advances the columns next to the faces first; this kernel then packs records
// from exactly those freshly written particles - the local columns [a_lo, a_hi) and [b_lo, b_hi), disjoint - while the
// interior columns are still being advanced.  Records go behind a header record (record 0 of each message; its first
// word receives the count from k_slab_headers).  `st` is the NEW state: own slot s of the last sort sits at index
// s - (first own slot).  Faces: [col_lo, col_hi) in global columns, the ones the NEXT step will use.
__global__ void __launch_bounds__(kBlock) k_slab_pack_faces(DevParams P, State st, const int32_t* __restrict__ start,
                                                            int a_lo, int a_hi, int b_lo, int b_hi, int col_lo,
                                                            int col_hi, int has_left, int has_right,
                                                            float* __restrict__ msg_left, float* __restrict__ msg_right,
                                                            int cap, int32_t* __restrict__ counters) {
  __shared__ int block_base[2];
  const int cpc = P.ny * P.nz;
  const int own_base = start[P.own_lo * cpc];
  const int a0 = start[a_lo * cpc], a1 = start[a_hi * cpc], b0 = start[b_lo * cpc], b1 = start[b_hi * cpc];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ((int)(blockIdx.x * blockDim.x) >= (a1 - a0) + (b1 - b0)) return;  // whole block beyond the face ranges
  int s = -1;
  if (i < a1 - a0) {
    s = a0 + i;
  } else {
    i -= a1 - a0;
    if (i < b1 - b0) s = b0 + i;
  }
  bool to_left = false, to_right = false;
  int o = 0;
  if (s >= 0) {
    o = s - own_base;
    const int g = cell_coord(st.x[o], P.cell, P.gnx);
    to_left = has_left && g < col_lo + 2;
    to_right = has_right && g >= col_hi - 2;
  }
  // one global atomic per block and side: the block's records take consecutive places in the message
  int n_left, n_right;
  const int at_left = block_exclusive_scan(to_left ? 1 : 0, &n_left);
  const int at_right = block_exclusive_scan(to_right ? 1 : 0, &n_right);
  if (threadIdx.x == 0) {
    block_base[0] = n_left ? atomicAdd(&counters[0], n_left) : 0;
    block_base[1] = n_right ? atomicAdd(&counters[1], n_right) : 0;
  }
  __syncthreads();
  if (!to_left && !to_right) return;
  const float r[kRecWords] = {st.x[o], st.y[o], st.z[o], st.vx[o], st.vy[o], st.vz[o], __int_as_float(st.id[o])};
  if (to_left) {
    const int k = block_base[0] + at_left;
    if (k < cap)
      for (int w = 0; w < kRecWords; w++) msg_left[(size_t)(k + 1) * kRecWords + w] = r[w];
  }
  if (to_right) {
    const int k = block_base[1] + at_right;
    if (k < cap)
      for (int w = 0; w < kRecWords; w++) msg_right[(size_t)(k + 1) * kRecWords + w] = r[w];
  }
}

__global__ void k_slab_headers(const int32_t* __restrict__ counters, float* __restrict__ msg_left,
                               float* __restrict__ msg_right) {
  if (threadIdx.x == 0 && msg_left) msg_left[0] = __int_as_float(counters[0]);
  if (threadIdx.x == 1 && msg_right) msg_right[0] = __int_as_float(counters[1]);
}

// The interior columns [lo, hi) were advanced after the records had been packed: none of their particles may have
// ended up within two columns of a face or beyond it (the neighbour would have needed a record of it).  Counts the
// ones that did; the caller widens the face ranges when this is not zero.
__global__ void k_slab_check_interior(DevParams P, State st, const int32_t* __restrict__ start, int lo, int hi,
                                      int col_lo, int col_hi, int has_left, int has_right,
                                      int32_t* __restrict__ missed) {
  const int cpc = P.ny * P.nz;
  const int own_base = start[P.own_lo * cpc];
  const int s0 = start[lo * cpc], s1 = start[hi * cpc];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s1 - s0) return;
  const int o = s0 + i - own_base;
  const int g = cell_coord(st.x[o], P.cell, P.gnx);
  if ((has_left && g < col_lo + 2) || (has_right && g >= col_hi - 2)) atomicAdd(missed, 1);
}

// Received records join the particle set.  A record whose x-column lies outside this context's local grid
// (own columns +- 2) can be neither integrated nor used as a ghost here and nobody else holds it: it is counted
// in *stray (the slabs are too narrow for the distance particles travel per step).
__global__ void k_slab_append(DevParams P, const float* __restrict__ rec, int n_rec, State st, int at,
                              int32_t* __restrict__ stray) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_rec) return;
  const float* r = rec + (size_t)k * kRecWords;
  int i = at + k;
  st.x[i] = r[0];
  st.y[i] = r[1];
  st.z[i] = r[2];
  st.vx[i] = r[3];
  st.vy[i] = r[4];
  st.vz[i] = r[5];
  st.id[i] = __float_as_int(r[6]);
  const int cx = cell_coord(r[0], P.cell, P.gnx) - P.cx_off;
  if (cx < 0 || cx >= P.nx) atomicAdd(stray, 1);
}

// Largest |vx| * dt / cell over the particles: how many cell columns a particle can cross in the next step.
// Non-negative floats order like their bit patterns, so the maximum is taken with an integer atomic.
// One atomic per BLOCK of a fixed, small grid: atomics on a single word are served one after the other at the memory side
// (about 11 ns each), so one per wave of a million particles would take 200 us.
constexpr int kReduceBlocks = 512;
__global__ void __launch_bounds__(kBlock) k_max_column_shift(DevParams P, State st, int n, int32_t* __restrict__ out) {
  __shared__ float wmax[kBlock / 64];
  float v = 0.0f;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float a = fabsf(st.vx[i]) * P.dt / P.cell;
    v = a > v ? a : v;  // a NaN never wins
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kBlock / 64; w++) v = fmaxf(v, wmax[w]);
    if (v > 0.0f) atomicMax(out, __float_as_int(v));
  }
}

__global__ void k_split_aos_ids(const float* __restrict__ pos, const float* __restrict__ vel,
                                const int32_t* __restrict__ ids, int n, State st) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st.x[i] = pos[3 * i + 0];
  st.y[i] = pos[3 * i + 1];
  st.z[i] = pos[3 * i + 2];
  st.vx[i] = vel[3 * i + 0];
  st.vy[i] = vel[3 * i + 1];
  st.vz[i] = vel[3 * i + 2];
  st.id[i] = ids[i];
}

// SoA -> AoS in storage order (no id permutation), ids alongside.
__global__ void k_join_aos(State st, int n, float* __restrict__ pos, float* __restrict__ vel,
                           int32_t* __restrict__ ids) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (pos) {
    pos[3 * i + 0] = st.x[i];
    pos[3 * i + 1] = st.y[i];
    pos[3 * i + 2] = st.z[i];
  }
  if (vel) {
    vel[3 * i + 0] = st.vx[i];
    vel[3 * i + 1] = st.vy[i];
    vel[3 * i + 2] = st.vz[i];
  }
  if (ids) ids[i] = st.id[i];
}

// Intermediates of the last step for the own particles, in the order the new state was written
// (index o <-> sorted slot base + o).
__global__ void k_join_debug(DevParams P, const int32_t* __restrict__ start, int n_own, int n_stride,
                             const ParticleRec* __restrict__ rec_s, const float* __restrict__ force_o,
                             const int32_t* __restrict__ ncount_s, const int32_t* __restrict__ cell_s,
                             float* rho, float* force, int32_t* ncount, int32_t* cell) {
  int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_own) return;
  int s = start[P.own_lo * P.ny * P.nz] + o;
  if (rho) rho[o] = rec_s[s].vel.w;
  if (force) {
    force[3 * o + 0] = force_o[o];
    force[3 * o + 1] = force_o[(size_t)n_stride + o];
    force[3 * o + 2] = force_o[2 * (size_t)n_stride + o];
  }
  if (ncount) ncount[o] = ncount_s[s];
  if (cell) cell[o] = cell_s[s] & kCellMask;
}

// ---- scene generation on the device -----------------------------------------------------------------------------------
// Jittered simple-cubic lattice, the counterpart of initParticles' fill (compute.cpp:108-117) for fields too large to
// generate on the host: particle (i, j, k) has index idx = (i*ny + j)*nz + k and sits at
// origin + (i+0.5, j+0.5, k+0.5)*spacing + (2u-1)*jitter*spacing, u = uniform [0,1) from a counter-based generator
// keyed by (idx, axis, seed) - two rounds of the splitmix64 finaliser - so any context can generate any x-planes of
// a field and get the same particles.  Evaluated in double, left to right, rounded once to float.
__device__ inline unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void k_fill_lattice(State st, long long first, int n, int ny, int nz, double spacing, double jitter,
                               unsigned long long seed, double ox, double oy, double oz) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long long idx = first + t;
  const long long k = idx % nz, j = (idx / nz) % ny, i = idx / ((long long)ny * nz);
  const double ijk[3] = {(double)i, (double)j, (double)k};
  const double o[3] = {ox, oy, oz};
  float p[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const unsigned long long key = (unsigned long long)idx * 3ull + (unsigned long long)a + seed * 0x100000001B3ull;
    const unsigned long long bits = splitmix64(splitmix64(key));
    const double u = (double)(bits >> 11) * (1.0 / 9007199254740992.0);
    const double jit = u * 2.0 - 1.0;
    p[a] = (float)(o[a] + (ijk[a] + 0.5) * spacing + jit * (jitter * spacing));
  }
  st.x[t] = p[0];
  st.y[t] = p[1];
  st.z[t] = p[2];
  st.vx[t] = 0.0f;  // compute.cpp:114-116
  st.vy[t] = 0.0f;
  st.vz[t] = 0.0f;
  st.id[t] = (int32_t)idx;
}

// Particles per local x-column, from the cell start table of the last sort.
__global__ void k_column_counts(const int32_t* __restrict__ start, int nx, int cpc, int32_t* __restrict__ out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nx) return;
  out[c] = start[(size_t)(c + 1) * cpc] - start[(size_t)c * cpc];
}

// Work per local x-column: sum over the column's particles of (kWorkBase + neighbours), the neighbour counts being
// those of the last search.  A step costs about a + b * neighbours per particle (sort and search per particle, search
// and force per neighbour); measured on the dam break, 0.225 ms per million particles at 13 neighbours and 0.60 ms at
// 50 put a / b at 9.  One block per column; the column's slots are one contiguous range of the sorted arrays.
constexpr int kWorkBase = 9;
constexpr int kWorkChunks = 32;  // blocks per column: a slab has a few dozen columns of tens of thousands of particles each
__global__ void __launch_bounds__(kBlock) k_column_work(const int32_t* __restrict__ start, int cpc,
                                                        const int32_t* __restrict__ nnb, int32_t* __restrict__ out) {
  const int c = blockIdx.x;
  const int lo = start[(size_t)c * cpc], hi = start[(size_t)(c + 1) * cpc];
  int acc = 0;  // a block's share of a column stays far below 2^31
  for (int s = lo + blockIdx.y * kBlock + threadIdx.x; s < hi; s += kWorkChunks * kBlock) acc += kWorkBase + nnb[s];
  int tot;
  block_exclusive_scan(acc, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(&out[c], tot);
}

}  // namespace

// ---- launchers -----------------------------------------------------------------------------------------------
void launch_split_aos(hipStream_t s, const float* pos_aos, const float* vel_aos, int n, State st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_split_aos, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, pos_aos, vel_aos, n, st);
}

void launch_hash(hipStream_t s, const DevParams& P, const State& in, int n, GridBuffers g) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_hash, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, P, in, n, g.cell_of, g.rank_of, g.count);
}

void launch_scan(hipStream_t s, GridBuffers g) {
  int nb = (int)g.scan_blocks;
  hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(kBlock), 0, s, g.count, g.ncells, g.block_sums);
  const bool fused = nb <= kFusedTopBlocks;
  if (!fused) hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kBlock), 0, s, g.block_sums, nb);
  hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(kBlock), 0, s, g.count, g.ncells, g.block_sums, g.start, g.mid,
                     fused ? 1 : 0);
}

void launch_reorder(hipStream_t s, const State& in, const Sorted& out, int n, GridBuffers g) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_scatter, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, n, g.cell_of, g.rank_of, g.start,
                     in.id, g.src_of, g.id_prov, g.cell_s);
  hipLaunchKernelGGL(k_reorder, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, n, in, out, g.start, g.mid, g.src_of,
                     g.id_prov, g.cell_s);
}

void launch_slab_pack(hipStream_t s, const DevParams& P, const State& st, int n, const int32_t* n_dev, int col_lo,
                      int col_hi, bool has_left, bool has_right, float* rec_left, float* rec_right, int cap,
                      int32_t* counters) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_slab_pack, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, P, st, n, n_dev, col_lo, col_hi,
                     has_left ? 1 : 0, has_right ? 1 : 0, rec_left, rec_right, cap, counters);
}

void launch_slab_pack_faces(hipStream_t s, const DevParams& P, const State& st, int n_bound, const int32_t* start,
                            int a_lo, int a_hi, int b_lo, int b_hi, int col_lo, int col_hi, bool has_left,
                            bool has_right, float* msg_left, float* msg_right, int cap, int32_t* counters) {
  if (n_bound <= 0 || (!has_left && !has_right)) return;
  hipLaunchKernelGGL(k_slab_pack_faces, dim3(div_up(n_bound, kBlock)), dim3(kBlock), 0, s, P, st, start, a_lo, a_hi, b_lo,
                     b_hi, col_lo, col_hi, has_left ? 1 : 0, has_right ? 1 : 0, msg_left, msg_right, cap, counters);
}

void launch_slab_headers(hipStream_t s, const int32_t* counters, float* msg_left, float* msg_right) {
  hipLaunchKernelGGL(k_slab_headers, dim3(1), dim3(64), 0, s, counters, msg_left, msg_right);
}

void launch_slab_check_interior(hipStream_t s, const DevParams& P, const State& st, int n_bound, const int32_t* start,
                                int lo, int hi, int col_lo, int col_hi, bool has_left, bool has_right,
                                int32_t* missed) {
  if (n_bound <= 0 || hi <= lo || (!has_left && !has_right)) return;
  hipLaunchKernelGGL(k_slab_check_interior, dim3(div_up(n_bound, kBlock)), dim3(kBlock), 0, s, P, st, start, lo, hi,
                     col_lo, col_hi, has_left ? 1 : 0, has_right ? 1 : 0, missed);
}

void launch_max_column_shift(hipStream_t s, const DevParams& P, const State& st, int n, int32_t* out) {
  if (n <= 0) return;
  const int blocks = div_up(n, kBlock) < kReduceBlocks ? div_up(n, kBlock) : kReduceBlocks;
  hipLaunchKernelGGL(k_max_column_shift, dim3(blocks), dim3(kBlock), 0, s, P, st, n, out);
}

void launch_slab_append(hipStream_t s, const DevParams& P, const float* rec, int n_rec, const State& st, int at,
                        int32_t* stray) {
  if (n_rec <= 0) return;
  hipLaunchKernelGGL(k_slab_append, dim3(div_up(n_rec, kBlock)), dim3(kBlock), 0, s, P, rec, n_rec, st, at, stray);
}

void launch_split_aos_ids(hipStream_t s, const float* pos, const float* vel, const int32_t* ids, int n, State st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_split_aos_ids, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, pos, vel, ids, n, st);
}

void launch_join_aos(hipStream_t s, const State& st, int n, float* pos, float* vel, int32_t* ids) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_join_aos, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, st, n, pos, vel, ids);
}

void launch_join_debug(hipStream_t s, const DevParams& P, const int32_t* start, int n_own, int n_stride,
                       const ParticleRec* rec_s, const float* force_o, const int32_t* ncount_s, const int32_t* cell_s,
                       float* rho, float* force, int32_t* ncount, int32_t* cell) {
  if (n_own <= 0) return;
  hipLaunchKernelGGL(k_join_debug, dim3(div_up(n_own, kBlock)), dim3(kBlock), 0, s, P, start, n_own, n_stride,
                     rec_s, force_o, ncount_s, cell_s, rho, force, ncount, cell);
}

void launch_fill_lattice(hipStream_t s, const State& st, long long first, int n, int ny, int nz, double spacing,
                         double jitter, unsigned long long seed, const double origin[3]) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_fill_lattice, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, st, first, n, ny, nz, spacing, jitter,
                     seed, origin[0], origin[1], origin[2]);
}

void launch_column_counts(hipStream_t s, const int32_t* start, int nx, int cpc, int32_t* out) {
  hipLaunchKernelGGL(k_column_counts, dim3(div_up(nx, kBlock)), dim3(kBlock), 0, s, start, nx, cpc, out);
}

void launch_column_work(hipStream_t s, const int32_t* start, int col_lo, int col_hi, int cpc, const int32_t* nnb,
                        int32_t* out) {
  if (col_hi <= col_lo) return;
  (void)hipMemsetAsync(out + col_lo, 0, (size_t)(col_hi - col_lo) * sizeof(int32_t), s);
  hipLaunchKernelGGL(k_column_work, dim3(col_hi - col_lo, kWorkChunks), dim3(kBlock), 0, s, start + (size_t)col_lo * cpc,
                     cpc, nnb, out + col_lo);
}

void launch_gather_aos(hipStream_t s, const State& st, int n, float* pos_aos, float* vel_aos) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_aos, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, st, n, pos_aos, vel_aos);
}

void launch_gather_debug(hipStream_t s, const int32_t* id, int n, const ParticleRec* rec_s, const float* force_s,
                         const int32_t* ncount_s, const int32_t* cell_s, float* rho, float* force,
                         int32_t* ncount, int32_t* cell) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_debug, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, id, n, rec_s, force_s, ncount_s,
                     cell_s, rho, force, ncount, cell);
}

}  // namespace sph
