// Neighbour sweeps for gfx950: neighbour search, density pass and force+integration pass of one SPH step
// (reference compute/compute.cpp:131-150 and :153-319).  Compile with -ffp-contract=off.
//
// The sweeps read the grid-sorted copy of the state, packed for gathers: 12 bytes per particle for the
// position (search, density) and a 32-byte record with position and velocity + density (force).  Cells are
// linearised as c = (cx*ny + cy)*nz + cz (z fastest; x, the slab axis, slowest): the 27-cell stencil of a
// particle is 9 runs of up to 3 z-adjacent cells, each run ONE contiguous slot range.  Runs are visited
// dx-major, dy-minor; inside a cell the particles of its lower half in z come first, then those of its upper half,
// each half in ascending id order (so a run can leave out the half of an end cell that is out of the particle's
// reach).  That order is a pure function of (positions, ids), so results do not depend on how particles share waves,
// blocks, contexts or slabs.
//
// Default (SWEEP_MASKED) pipeline, one lane per particle in every kernel:
//   k_neighbours  the only kernel that tests distances: walks the flattened stencil four candidates per trip
//                 (48 packed bytes, three 16-byte gathers) and records WHICH candidates satisfy sqrtf(r2) <= h as a
//                 bitmask per particle, plus the candidate and neighbour counts.  Every run of the stencil is padded to
//                 whole trips in the mask (a run of len candidates owns 4 * ceil(len / 4) bit positions), so trip t of a
//                 particle fills bits 4t .. 4t+3 whatever came before;
//   k_density     walks the set bits: every loop trip is a true neighbour, so the sqrt / double-precision
//                 kernel arithmetic runs with the wave's lanes busy instead of ~20 % of them;
//   k_force       same walk with the pair arithmetic of :153-262, then Euler step and wall reflection.
//
// Other flavours, for verification and ablation:
//   SWEEP_PLAIN  tests and evaluates in one loop (no mask); same results bit for bit as SWEEP_MASKED;
//   SWEEP_EXACT  sums neighbours in ascending particle id, the order in which the reference's j = 0..N-1
//                loops meet the non-zero terms: bit-identical to the reference.
#include <type_traits>

#include "sph_device.hpp"

namespace sph {

namespace {

// Threads per block of the sweeps: 256 for large launches; 128 below kSmallLaunch particles, where the grid is only a
// few rounds of blocks deep and finer blocks shorten its tail (1M-particle dam break at peak compression: step 0.477 ->
// 0.467 ms; at 8M and 16.8M particles the smaller record tile costs the force pass 1 %).
constexpr int kBlock = 256;
constexpr int kSmallBlock = 128;
constexpr int kSmallLaunch = 4 << 20;
constexpr int kAhead = 4;  // candidates fetched per loop trip of the search
static_assert(kAhead == 4, "the search unpacks four packed positions out of three 16-byte words");
struct __attribute__((aligned(4))) Quad {  // 16 bytes at dword alignment
  float a, b, c, d;
};
constexpr int kMaskBits = 32 * kMaskWords;
// Record tile of the force pass: the block's own slots (plus kTilePad on either side).  Measured at peak compression,
// run ends held as 16-bit: 204 us without margin (7 blocks per CU), 215 us with 32 slots (6 blocks); with 32-bit run
// ends 226 / 225 / 254 us for margins of 32 / 64 / 128 slots (5, 5, 4 blocks): what a margin adds in hits it loses in
// resident blocks.
constexpr int kTilePad = 0;
constexpr int tile_records(int block) { return block + 2 * kTilePad; }  // 8 KB of LDS for 256 threads

__host__ inline int div_up(long long a, int b) { return (int)((a + b - 1) / b); }

// Workgroups are dealt round-robin to the 8 XCDs, each with a private 4 MiB L2.  Remap the block index so that
// every XCD works on runs of `run` CONSECUTIVE blocks of the sorted particle range instead of every eighth block.
// With run = blocks / (8 * kXcdWindows) (range_first_slot below) each XCD sweeps kXcdWindows contiguous windows of the
// domain one after the other, so the blocks it has in flight at any time are neighbours in the sorted order and
// their stencils (three x-planes of cells) overlap in its L2.  Several windows per XCD rather than one spread
// crowded and empty regions of the tank over the XCDs.  Measured on the 1M dam break, L2 misses of the force
// pass: 270 MB per launch with 32-block runs, 87 MB with 4 windows, 70 MB with 1 window (which loses 7 % of
// time to imbalance).  Bijective for any grid size (the tail that does not fill 8 runs keeps its index);
// affects locality only, never results.
constexpr int kXcdWindows = 8;
// a / b for 0 <= a, b < 2^24, b >= 1 (block counts): float reciprocal and corrections instead of the ~30 instructions
// of a general 32-bit division - every lane computes these block-uniform values.  The estimate is within 2^-22 of
// a / b relatively, i.e. at most 2 off the quotient either way after rounding and truncation.
__device__ inline int div_small(int a, int b) {
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
  int r = a - q * b;
  if (r < 0) { q--; r += b; }
  if (r < 0) { q--; r += b; }
  if (r >= b) { q++; r -= b; }
  if (r >= b) q++;
  return q;
}
__device__ inline int xcd_block(int b, int nb, int run) {
  const int span = 8 * run;
  if (b >= div_small(nb, span) * span) return b;
  const int xcd = b & 7, i = b >> 3;
  const int w = div_small(i, run);
  return (w * 8 + xcd) * run + (i - w * run);
}

// First slot offset of this block within a range of `nslots` slots that a launch of at least ceil(nslots / B) blocks
// covers, or -1 for a block beyond it.  The launch is sized for an upper bound known to the host (all particles); the
// range (some columns of them) is known on the device only.  The windows are laid over the blocks that HAVE work:
// hardware deals block indices round-robin to the XCDs whether they exit at once or not, so the first
// ceil(nslots / B) indices are spread evenly, and mapping those keeps every XCD busy for any range.
// A window should be a few x-columns deep, or most of what its blocks gather (three columns per block) is fetched for
// that window alone: ranges of fewer than 24 * kXcdWindows columns (slabs, face ranges) get fewer windows per XCD.
__device__ inline int range_first_slot(int nslots, int B, int ncols) {
  const int nact = (nslots + B - 1) / B;
  if ((int)blockIdx.x >= nact) return -1;
  // ncols / 24 below 24 * kXcdWindows columns: (ncols * 171) >> 12 is exact up to 192
  static_assert(kXcdWindows == 8, "the multiply-shift below stands for ncols / 24 up to 24 * 8 columns");
  int windows = ncols >= 24 * kXcdWindows ? kXcdWindows : (ncols * 171) >> 12;
  windows = windows < 1 ? 1 : windows;
  int run = div_small(nact, 8 * windows);
  run = run > 0 ? run : 1;
  return xcd_block(blockIdx.x, nact, run) * B;
}

struct CellCoord {
  int x, y, z;
};
// Two divisions by grid dimensions per particle and pass: by multiply-high and shift with constants from the host
// (an integer division by a run-time divisor is some 35 vector instructions on this chip).
__device__ inline int div_by_dim(int c, int d, uint32_t mul, uint32_t shift) {
  return d == 1 ? c : (int)(__umulhi((uint32_t)c, mul) >> shift);
}
__device__ inline CellCoord decode_cell(const DevParams& P, int c) {
  CellCoord q;
  const int t = div_by_dim(c, P.nz, P.nz_mul, P.nz_shift);
  q.z = c - t * P.nz;
  q.x = div_by_dim(t, P.ny, P.ny_mul, P.ny_shift);
  q.y = t - q.x * P.ny;
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

// The 9 stencil runs of every lane's particle, fetched up front (18 independent loads, one latency) into a
// per-lane column of LDS, so that stepping from run to run inside the sweeps is an LDS read instead of two
// dependent global loads that would stall the whole wave each time one lane steps.  A run is stored by where it
// sits in the particle's flattened candidate sequence: positions [qend[k-1], qend[k]) are run k (padded to a multiple
// of kAhead), and position q of run k is sorted slot shift[k] + q.
// QE: int where the whole stencil is walked whatever its size (search, density); unsigned short in the force pass, which
// consults the table only for stencils of at most kMaskBits candidates and wants the LDS for a record tile.
template <class QE, int B>
struct RunTableT {
  int (*shift)[B];  // first slot of run k minus the number of candidates before run k
  QE (*qend)[B];    // number of candidates up to and including run k (saturated at the largest QE)
};
// Returns the number of kAhead-wide trips a walk over all runs takes.
// Where a stencil run of a particle in cell (.., .., cz) begins and ends in a row of cells, as (table, index) pairs: the
// cells cz-1 .. cz+1, minus the half of an end cell that the particle's own half of its cell puts out of reach (the
// lower half of cz-1 for a particle in the upper half of cz, the upper half of cz+1 for one in the lower half; a
// cell's particles are sorted lower half first and mid[] is where its upper half begins).
struct RunEnds {
  const int32_t* first_table;
  const int32_t* last_table;
  int first, last;  // cell offsets within a row: the run is [first_table[row + first], last_table[row + last])
};
// Only particles of well-filled cells clip their runs (kClipFrom particles or more in the particle's own cell; the
// counting sort leaves that verdict in bit 31 of the particle's sorted cell word): the lanes of a cell then fall into
// two groups with different runs, and where cells hold three particles that costs the vector cache more merged lookups
// than the shorter runs save.
__device__ inline RunEnds run_ends(const DevParams& P, const int32_t* __restrict__ start,
                                   const int32_t* __restrict__ mid, bool clip, int cz, int upper) {
  RunEnds r;
  r.first_table = (clip && upper && cz > 0) ? mid : start;
  r.first = cz > 0 ? cz - 1 : 0;
  const bool clip_top = clip && !upper && cz < P.nz - 1;
  r.last_table = clip_top ? mid : start;
  r.last = clip_top ? cz + 1 : (cz < P.nz - 1 ? cz + 2 : P.nz);
  return r;
}

template <class QE, int B>
__device__ inline int fill_runs(const DevParams& P, const int32_t* __restrict__ start,
                                const int32_t* __restrict__ mid, const CellCoord cc, bool clip, int upper,
                                const RunTableT<QE, B>& rt) {
  const RunEnds re = run_ends(P, start, mid, clip, cc.z, upper);
  // rows beyond the grid are read as the nearest row inside it (any valid address) and recorded as empty runs;
  // no branch, so that all 18 loads are issued before the first is waited for
  int b[9], e[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int x = cc.x + k / 3 - 1, y = cc.y + k % 3 - 1;
    const int xc = x < 0 ? 0 : (x > P.nx - 1 ? P.nx - 1 : x), yc = y < 0 ? 0 : (y > P.ny - 1 ? P.ny - 1 : y);
    const int row = (xc * P.ny + yc) * P.nz;
    b[k] = re.first_table[row + re.first];
    e[k] = re.last_table[row + re.last];
  }
  int q = 0, trips = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int x = cc.x + k / 3 - 1, y = cc.y + k % 3 - 1;
    const bool inside = x >= 0 && x < P.nx && y >= 0 && y < P.ny;
    const int len = inside ? e[k] - b[k] : 0;
    rt.shift[k][threadIdx.x] = b[k] - q;
    q += (len + kAhead - 1) / kAhead * kAhead;  // the mask pads every run to whole trips of the search
    trips += (len + kAhead - 1) / kAhead;
    rt.qend[k][threadIdx.x] = sizeof(QE) == 2 ? (QE)(q < 65535 ? q : 65535) : (QE)q;
  }
  return trips;
}

// Slot range [lo, hi) of the local x-columns [col_lo, col_hi).
__device__ inline void column_slots(const DevParams& P, const int32_t* __restrict__ start, int col_lo, int col_hi,
                                    int& lo, int& hi) {
  const int cpc = P.ny * P.nz;
  lo = start[col_lo * cpc];
  hi = start[col_hi * cpc];
}

// ---- neighbour search ---------------------------------------------------------------------------------------------
// The only kernel that tests distances.  One lane per particle walks its flattened stencil four candidates per
// trip (three independent 16-byte gathers in flight) and packs the outcomes into the particle's mask:
// bit 4t + u <=> candidate u of trip t passes sqrtf(r2) <= h, tested as r2 <= r2_max with
// r2 = (dx*dx + dy*dy) + dz*dz in uncontracted fp32 (:140-144).
// WITH_DENSITY (fast arithmetic): the density is accumulated on the spot, rho_i = (m c_poly6)_fp32 * sum over the
// neighbours of (h2 - r2)^3 (:65-68,:146 with r2 in place of the re-squared rounded r; no square root, no double
// product), and written with its refined reciprocal into the particle's record: no separate density pass.
//
// The kernel is bound by the issue of its own instructions (switching the gathers or the distance arithmetic off
// barely shortens it), so the loop is built to need few of them:
//  * the search has a run table of its own (fill_search_runs): non-empty runs only, as (byte offset of the first
//    position, length); the entry after the current run is fetched from LDS when the current one is entered, so stepping
//    from run to run is a few register moves and never waits;
//  * trip t fills bits 4t .. 4t+3: word index and shift are the same for every lane of the wave (no per-lane word
//    bookkeeping, stores of completed words are uniform and coalesced);
//  * a candidate past the end of its run gets a squared distance nothing passes, which serves the outcome bit and the
//    density term alike.
constexpr int kSearchRuns = 10;  // nine runs and a zero-length entry behind the last
template <int B>
struct SearchRuns {
  unsigned (*boff)[B];  // byte offset of the run's first packed position
  int (*len)[B];        // candidates in the run
};
// Returns the number of kAhead-wide trips the particle's stencil takes.  *tight is set when fewer than kAhead - 1
// slots lie between the end of one of the particle's runs and the start of its next one (or the end of the sorted
// array): only then can a slot fetched past the end of a run be a particle that is ALSO a candidate of a later run (or
// no particle at all).  Everywhere else the slots behind a run belong to cells outside the 27-cell stencil, at least
// one padded cell away, and fail the distance test by themselves.
template <int B>
__device__ inline int fill_search_runs(const DevParams& P, const int32_t* __restrict__ start,
                                       const int32_t* __restrict__ mid, const CellCoord cc, bool clip, int upper,
                                       const SearchRuns<B>& rt, int n, bool* tight) {
  const RunEnds re = run_ends(P, start, mid, clip, cc.z, upper);
  int b[9], e[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {  // branch-free: all 18 loads are issued before the first is waited for
    const int x = cc.x + k / 3 - 1, y = cc.y + k % 3 - 1;
    const int xc = x < 0 ? 0 : (x > P.nx - 1 ? P.nx - 1 : x), yc = y < 0 ? 0 : (y > P.ny - 1 ? P.ny - 1 : y);
    const int row = (xc * P.ny + yc) * P.nz;
    b[k] = re.first_table[row + re.first];
    e[k] = re.last_table[row + re.last];
  }
  int nrun = 0, trips = 0, last_end = -kAhead;
  bool close = false;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int x = cc.x + k / 3 - 1, y = cc.y + k % 3 - 1;
    const bool inside = x >= 0 && x < P.nx && y >= 0 && y < P.ny;
    const int len = inside ? e[k] - b[k] : 0;
    // branch-free: an empty run's entry is written too and overwritten by the next run's (or by the closing entry)
    rt.boff[nrun][threadIdx.x] = (unsigned)b[k] * 12u;  // a context holds far fewer than 2^32 / 12 particles
    rt.len[nrun][threadIdx.x] = len;
    nrun += len > 0 ? 1 : 0;
    trips += (len + kAhead - 1) / kAhead;
    close = close || (len > 0 && b[k] - last_end < kAhead - 1);  // runs are met in ascending slot order
    last_end = len > 0 ? e[k] : last_end;
  }
  rt.boff[nrun][threadIdx.x] = 0u;
  rt.len[nrun][threadIdx.x] = 0;
  *tight = close || n - last_end < kAhead - 1;
  return trips;
}

template <int B, bool WITH_DENSITY>
__global__ void __launch_bounds__(B) k_neighbours(DevParams P, Sorted st, const int32_t* __restrict__ start,
                                                       const int32_t* __restrict__ mid,
                                                       const int32_t* __restrict__ cell_s, NeighbourMask nm,
                                                       int32_t* __restrict__ ncount, int n) {
  __shared__ unsigned rt_boff[kSearchRuns][B];
  __shared__ int rt_len[kSearchRuns][B];
  int lo, hi;
  column_slots(P, start, P.dens_lo, P.dens_hi, lo, hi);
  const int off = range_first_slot(hi - lo, B, P.dens_hi - P.dens_lo);
  const int s = lo + off + threadIdx.x;
  if (off < 0 || s >= hi) return;
  const float4 pi = sorted_pos(st, s);
  bool tight;
  const int cw = cell_s[s];  // bit 31: the particle's cell is full enough for clipped runs
  const int trips = fill_search_runs(P, start, mid, decode_cell(P, cw & kCellMask), cw < 0, upper_half(P, pi.z),
                                     SearchRuns<B>{rt_boff, rt_len}, n, &tight);
  // One wave, one loop: with the checks of run ends only if some lane of the wave needs them (a wave of the bulk of a
  // fluid does not).
  const bool check_ends = __any(tight);
  constexpr float kFar = 3.0e38f;  // squared distance of a candidate past the end of its run
  unsigned word = 0;
  int cnt = 0;
  float sum = 0.0f;
  float dx, dy, dz;
  // current run: byte offset of the next candidate and candidates left (<= 0: enter the next run); the run after it
  int k = 0, rem = 0, len_n = rt_len[0][threadIdx.x];
  unsigned boff = 0, boff_n = rt_boff[0][threadIdx.x];
  uint32_t* wp = nm.words + s;  // where the word being assembled goes
  // the number of trips is known up front, so the loop needs no end-of-stencil test: a trip always finds a run
  auto walk = [&](auto checked) {
  for (int t = 0; t < trips; t++) {
    if (rem <= 0) {
      boff = boff_n;
      rem = len_n;
      ++k;
      boff_n = rt_boff[k][threadIdx.x];
      len_n = rt_len[k][threadIdx.x];
    }
    // Always fetch kAhead consecutive slots from one base address: past the end of the run they belong to the
    // next cell or to the array's padding, and they are kept out below.
    // kAhead = 4 packed positions are 48 bytes: three 16-byte loads (dword-aligned, as global loads may be), addressed
    // by a 32-bit byte offset from the array's base
    const Quad* base = reinterpret_cast<const Quad*>(reinterpret_cast<const char*>(st.pos3) + boff);
    const Quad w0 = base[0], w1 = base[1], w2 = base[2];
    float4 cand[kAhead];
    cand[0] = make_float4(w0.a, w0.b, w0.c, 0.0f);
    cand[1] = make_float4(w0.d, w1.a, w1.b, 0.0f);
    cand[2] = make_float4(w1.c, w1.d, w2.a, 0.0f);
    cand[3] = make_float4(w2.b, w2.c, w2.d, 0.0f);
    float r2[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; u++) {
      r2[u] = dist2_to(pi, cand[u], dx, dy, dz);
      if (decltype(checked)::value && u > 0) r2[u] = u < rem ? r2[u] : kFar;  // candidate 0 always belongs to the run
    }
    unsigned bits = 0;
#pragma unroll
    for (int u = kAhead - 1; u >= 0; u--) bits = bits + bits + (r2[u] <= P.r2_max ? 1u : 0u);
    if (WITH_DENSITY) {
      // (h2 - r2) clamped at 0 in the subtraction itself: a candidate beyond h (or beyond its run) adds nothing
#pragma unroll
      for (int u = 0; u < kAhead; u++) {
        const float d = __builtin_amdgcn_fmed3f(P.h2 - r2[u], 0.0f, 1.0f);
        sum = __builtin_fmaf(d * d, d, sum);
      }
    }
    cnt += __popc(bits);
    word |= bits << ((t & 7) * kAhead);
    if ((t & 7) == 7) {  // uniform: the word is complete for every lane still walking
      if ((t >> 3) < kMaskWords) *wp = word;
      wp += nm.stride;
      word = 0;
    }
    boff += 12u * kAhead;
    rem -= kAhead;
  }
  };
  if (check_ends) walk(std::true_type{});
  else walk(std::false_type{});
  if ((trips & 7) != 0 && (trips >> 3) < kMaskWords) *wp = word;
  nm.ncand[s] = trips * kAhead;
  nm.nnb[s] = cnt;
  if (WITH_DENSITY) {
    const float rho = P.k_dens * sum;
    st.rec[s].vel.w = rho;
    st.rec[s].pos.w = fast_rcp(rho);
    if (ncount) ncount[s] = cnt;
  }
}

// Lanes of a wave run in lockstep, so a wave takes as long as its particle with the most neighbours.  Dealing the
// block's particles to lanes by neighbour count (a counting sort in LDS, bins of eight) used to even that out; with the
// kernels as they are it loses everywhere it was measured - consecutive slots per wave share more cache lines than equal
// loads save, and the sort cost eight barriers per block: force pass 1.40 -> 1.27 ms on the 16.8M-particle cube,
// 2.34 -> 2.16 ms on the 8M-particle dam break at its peak, 71 -> 69 us on the 1M dam break at rest density and
// 198 -> 195 us at its peak.  The lanes of a block are its slots in order.

// Walks the set bits of particle slot s's candidate mask in order and calls visit(pos_j, vel_j) for each; load(j,
// pos, vel) fetches candidate index j, where j counts from the run starts in column `col` of the run table (slots
// of the sorted arrays, or positions in a staged tile).  The loads of the NEXT neighbour are issued before the
// current one is evaluated (two neighbours ahead), so their latency hides behind the pair arithmetic.
template <class Table, class Load, class Visit>
__device__ inline void walk_mask(const NeighbourMask& nm, int s, int nn, const Table& rt, int col, Load load,
                                 Visit visit) {
  if (nn <= 0) return;
  // nn set bits are known to be there, so the walk needs no end-of-mask checks: the word after the current one
  // is always fetched (the array has a spare row behind the last word) and next_index() always finds a bit.
  const uint32_t* wp = nm.words + s;  // word after the one in m
  const size_t stride = nm.stride;
  unsigned m = *wp;  // current word (bits still to visit)
  wp += stride;
  unsigned m_next = *wp;  // the word after it
  int qbase = 0;          // candidate index of bit 0 of the current word
  int k = 0, shift = rt.shift[0][col], qe = rt.qend[0][col];  // current run
  auto next_index = [&]() -> int {
    while (m == 0) {
      qbase += 32;
      m = m_next;
      wp += stride;
      m_next = *wp;
    }
    const int q = qbase + __ffs(m) - 1;
    m &= m - 1;
    while (q >= qe) {  // runs are met in ascending order
      ++k;
      shift = rt.shift[k][col];
      qe = rt.qend[k][col];
    }
    return shift + q;
  };
  float4 pj = make_float4(0.0f, 0.0f, 0.0f, 0.0f), vj = pj, pn = pj, vn = pj, pm = pj, vm = pj;
  load(next_index(), pj, vj);
  if (nn > 1) load(next_index(), pn, vn);
  for (int i = 2; i < nn; i++) {  // two neighbours' records in flight behind the one being evaluated
    load(next_index(), pm, vm);
    visit(pj, vj);
    pj = pn;
    vj = vn;
    pn = pm;
    vn = vm;
  }
  visit(pj, vj);
  if (nn > 1) visit(pn, vn);
}

// Calls visit(pos_j, vel_j) for every neighbour of particle slot s, in stencil order, from the mask written by
// the neighbour search.  Falls back to testing again when the stencil holds more candidates than the mask has
// bits (cells crowded beyond ~75 particles each) or the search left no mask.
// A copy in LDS of the records of the slots [lo, lo + n): the block's own slots and a margin on either side.  The
// neighbours a particle finds in its own (x, y) row of cells - the row that holds most of them - are its neighbours in
// slot order too, so a good third of the force pass's gathers land in this window and are served from LDS instead of
// through the vector cache.  Purely a cache: a slot outside the window is fetched from global memory as before.
struct RecordTile {
  const float4* pos;
  const float4* vel;
  int lo;
  unsigned n;
};

template <bool WITH_VEL, class Table, class Visit>
__device__ inline void for_each_neighbour(const DevParams& P, const Sorted& st, const int32_t* __restrict__ start,
                                          const NeighbourMask& nm, int s, const float4& pi, const Table& rt,
                                          int col, int cell_of_s, Visit visit,
                                          const RecordTile tile = RecordTile{nullptr, nullptr, 0, 0u}) {
  const int ncand = nm.ncand[s];
  if (ncand <= kMaskBits) {
    walk_mask(nm, s, nm.nnb[s], rt, col,
              [&](int j, float4& pj, float4& vj) {
                if (WITH_VEL) {
                  const unsigned t = (unsigned)(j - tile.lo);
                  if (t < tile.n) {
                    pj = tile.pos[t];
                    vj = tile.vel[t];
                  } else {
                    pj = st.rec[j].pos;
                    vj = st.rec[j].vel;
                  }
                } else {
                  pj = sorted_pos(st, j);
                }
              },
              visit);
  } else {
    float dx, dy, dz;
    const CellCoord cc = decode_cell(P, cell_of_s);  // the table may be saturated for a stencil this large
    RunCursor c{-1, 0, 0};
    while (next_run(P, start, cc, c)) {
      for (int j = c.j; j < c.e; j++) {
        const float4 pj = sorted_pos(st, j);
        if (dist2_to(pi, pj, dx, dy, dz) <= P.r2_max) {
          if (WITH_VEL) visit(st.rec[j].pos, st.rec[j].vel);  // the record's w components carry 1/density and density
          else visit(pj, pj);
        }
      }
    }
  }
}

// ---- density ----------------------------------------------------------------------------------------------------
// rho_i = sum_j m * Wpoly6(|r_i - r_j|), self term included (:131-150).  Written to rec[s].vel.w.
// FAST (masked flavour only): rho_i = (m c_poly6)_fp32 * sum (h2 - r2)^3 - no square root, no double product, r2 in
// place of the re-squared rounded r - and 1/rho_i, refined, stored beside it for the force pass.
template <int B, int MODE, bool FAST>
__global__ void __launch_bounds__(B) k_density(DevParams P, Sorted st, const int32_t* __restrict__ start,
                                                    const int32_t* __restrict__ mid,
                                                    const int32_t* __restrict__ cell_s,
                                                    int32_t* __restrict__ ncount, NeighbourMask nm) {
  __shared__ int rt_shift[9][B], rt_qend[9][B];  // used by the masked flavour only
  using RunTable = RunTableT<int, B>;
  // slots of the x-columns [dens_lo, dens_hi): own columns plus one ghost column on each side
  int lo, hi;
  column_slots(P, start, P.dens_lo, P.dens_hi, lo, hi);
  const int off = range_first_slot(hi - lo, B, P.dens_hi - P.dens_lo);
  const int first = lo + off;
  const int col = threadIdx.x;
  const int s = first + col;
  if (off < 0 || s >= hi) return;
  const float4 pi = sorted_pos(st, s);
  const int cw = cell_s[s];  // bit 31: the particle's cell is full enough for clipped runs
  const CellCoord cc = decode_cell(P, cw & kCellMask);
  if (MODE == SWEEP_MASKED)  // own column: no barrier needed
    fill_runs(P, start, mid, cc, cw < 0, upper_half(P, pi.z), RunTable{rt_shift, rt_qend});
  float acc = 0.0f;
  int cnt = 0;
  float dx, dy, dz;
  auto add = [&](float r2) {
    if (FAST) {
      const float d = P.h2 - r2;
      acc = __builtin_fmaf(d * d, d, acc);
    } else {
      acc += P.m * w_poly6(P, sqrt_rn(r2));  // :144-146
    }
    cnt++;
  };
  if (MODE == SWEEP_MASKED) {
    for_each_neighbour<false>(P, st, start, nm, s, pi, RunTable{rt_shift, rt_qend}, col, cw & kCellMask,
                              [&](const float4& pj, const float4&) { add(dist2_to(pi, pj, dx, dy, dz)); });
  } else if (MODE == SWEEP_PLAIN) {
    RunCursor c{-1, 0, 0};
    while (next_run(P, start, cc, c)) {
      for (int j = c.j; j < c.e; j++) {
        const float r2 = dist2_to(pi, sorted_pos(st, j), dx, dy, dz);
        if (r2 <= P.r2_max) add(r2);
      }
    }
  } else {  // SWEEP_EXACT: ascending particle id
    int last = -1;
    for (;;) {
      int best = 0x7fffffff;
      float best_r2 = 0.0f;
      RunCursor c{-1, 0, 0};
      while (next_run(P, start, cc, c)) {
        for (int j = c.j; j < c.e; j++) {
          const float r2 = dist2_to(pi, sorted_pos(st, j), dx, dy, dz);
          const int idj = st.id[j];
          if (r2 <= P.r2_max && idj > last && idj < best) {
            best = idj;
            best_r2 = r2;
          }
        }
      }
      if (best == 0x7fffffff) break;
      add(best_r2);
      last = best;
    }
  }
  if (FAST) {
    acc *= P.k_dens;
    st.rec[s].pos.w = fast_rcp(acc);
  }
  st.rec[s].vel.w = acc;
  if (ncount) ncount[s] = cnt;
}

// ---- forces, integration, walls -------------------------------------------------------------------------------------
template <int B, int MODE, bool FAST>
__global__ void __launch_bounds__(B) k_force(DevParams P, Sorted st, State out, int n,
                                                  const int32_t* __restrict__ start,
                                                  const int32_t* __restrict__ mid,
                                                  const int32_t* __restrict__ cell_s, float* __restrict__ force_dbg,
                                                  int32_t* __restrict__ n_own_out, NeighbourMask nm) {
  __shared__ int rt_shift[9][B];  // used by the masked flavour only
  __shared__ unsigned short rt_qend[9][B];
  using ForceRuns = RunTableT<unsigned short, B>;
  constexpr int kTileRecords = tile_records(B);
  // This launch advances the slots of the x-columns [force_lo, force_hi), a part (usually all) of the own columns
  // [own_lo, own_hi); the new state is written compactly: own slot s goes to index s - (first own slot).
  int base, bound, own_base, own_bound;
  column_slots(P, start, P.force_lo, P.force_hi, base, bound);
  column_slots(P, start, P.own_lo, P.own_hi, own_base, own_bound);
  if (blockIdx.x == 0 && threadIdx.x == 0 && n_own_out) {
    const int real = P.nx * P.ny * P.nz;
    n_own_out[0] = own_bound - own_base;           // particles this context keeps
    n_own_out[1] = start[real + 1] - start[real];  // trash cell: particles outside the local grid
  }
  const int first = range_first_slot(bound - base, B, P.force_hi - P.force_lo);
  if (first < 0) return;  // whole block beyond the range (uniform: before any barrier)
  const int col = threadIdx.x;
  __shared__ float4 tile_pos[MODE == SWEEP_MASKED ? kTileRecords : 1], tile_vel[MODE == SWEEP_MASKED ? kTileRecords : 1];
  RecordTile tile{tile_pos, tile_vel, 0, 0u};
  if (MODE == SWEEP_MASKED) {
    const int s_nat = base + first + threadIdx.x;
    if (s_nat < bound) {
      const int cw = cell_s[s_nat];  // bit 31: the particle's cell is full enough for clipped runs
      fill_runs(P, start, mid, decode_cell(P, cw & kCellMask), cw < 0, upper_half(P, st.pos3[3 * (size_t)s_nat + 2]),
                ForceRuns{rt_shift, rt_qend});
    }
    tile.lo = base + first - kTilePad > 0 ? base + first - kTilePad : 0;
    const int tile_hi = tile.lo + kTileRecords < n ? tile.lo + kTileRecords : n;
    tile.n = (unsigned)(tile_hi - tile.lo);
    for (int t = threadIdx.x; t < (int)tile.n; t += B) {
      tile_pos[t] = st.rec[tile.lo + t].pos;
      tile_vel[t] = st.rec[tile.lo + t].vel;
    }
    __syncthreads();  // publishes the tile (every thread of the block gets here: the range test above is block-wide)
  }
  const int s = base + first + col;
  const int o = s - own_base;
  if (s >= bound) return;
  const float4 pi4 = sorted_pos(st, s);
  const float4 vi4 = st.rec[s].vel;  // w = density
  const float rho_i = vi4.w;
  const float pi = P.c2 * (rho_i - P.rho0);  // :158
  const int cell_i = cell_s[s] & kCellMask;
  const CellCoord cc = decode_cell(P, cell_i);
  float a[kForceComps] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float dx, dy, dz;

  auto add_pair = [&](const float4& pj, const float4& vj) {
    const float r2 = dist2_to(pi4, pj, dx, dy, dz);
    if (FAST) {
      pair_accumulate_fast(P, a, dx, dy, dz, pi, vj.w, pj.w, vj.x - vi4.x, vj.y - vi4.y, vj.z - vi4.z);
    } else {
      const PairOut w = pair_eval(P, dx, dy, dz, r2, pi, vj.w, vj.x - vi4.x, vj.y - vi4.y, vj.z - vi4.z);
#pragma unroll
      for (int q = 0; q < kForceComps; q++) a[q] += w.v[q];
    }
  };

  if (MODE == SWEEP_MASKED) {
    for_each_neighbour<true>(P, st, start, nm, s, pi4, ForceRuns{rt_shift, rt_qend}, col, cell_i, add_pair,
                             tile);
  } else if (MODE == SWEEP_PLAIN) {
    RunCursor c{-1, 0, 0};
    while (next_run(P, start, cc, c)) {
      for (int j = c.j; j < c.e; j++) {
        const float r2 = dist2_to(pi4, sorted_pos(st, j), dx, dy, dz);
        if (r2 <= P.r2_max) add_pair(sorted_pos(st, j), st.rec[j].vel);
      }
    }
  } else {  // SWEEP_EXACT: ascending particle id
    int last = -1;
    for (;;) {
      int best = 0x7fffffff, bj = -1;
      RunCursor c{-1, 0, 0};
      while (next_run(P, start, cc, c)) {
        for (int j = c.j; j < c.e; j++) {
          const float r2 = dist2_to(pi4, sorted_pos(st, j), dx, dy, dz);
          const int idj = st.id[j];
          if (r2 <= P.r2_max && idj > last && idj < best) {
            best = idj;
            bj = j;
          }
        }
      }
      if (bj < 0) break;
      add_pair(sorted_pos(st, bj), st.rec[bj].vel);
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

namespace {
template <int B>
void neighbours_on(hipStream_t s, const DevParams& P, const Sorted& st, int n, const GridBuffers& g,
                   const NeighbourMask& nm, bool with_density, int32_t* ncount) {
  const dim3 grid(div_up(n, B)), block(B);
  if (with_density)
    hipLaunchKernelGGL((k_neighbours<B, true>), grid, block, 0, s, P, st, g.start, g.mid, g.cell_s, nm, ncount, n);
  else
    hipLaunchKernelGGL((k_neighbours<B, false>), grid, block, 0, s, P, st, g.start, g.mid, g.cell_s, nm, ncount, n);
}

template <int B>
void density_on(hipStream_t s, const DevParams& P, const Sorted& st, int n, const GridBuffers& g, int32_t* ncount,
                const NeighbourMask& nm, SweepMode mode) {
  const dim3 grid(div_up(n, B)), block(B);
  if (mode == SWEEP_EXACT)
    hipLaunchKernelGGL((k_density<B, SWEEP_EXACT, false>), grid, block, 0, s, P, st, g.start, g.mid, g.cell_s, ncount, nm);
  else if (mode == SWEEP_PLAIN)
    hipLaunchKernelGGL((k_density<B, SWEEP_PLAIN, false>), grid, block, 0, s, P, st, g.start, g.mid, g.cell_s, ncount, nm);
  else if (mode == SWEEP_FAST)
    hipLaunchKernelGGL((k_density<B, SWEEP_MASKED, true>), grid, block, 0, s, P, st, g.start, g.mid, g.cell_s, ncount, nm);
  else
    hipLaunchKernelGGL((k_density<B, SWEEP_MASKED, false>), grid, block, 0, s, P, st, g.start, g.mid, g.cell_s, ncount, nm);
}

template <int B>
void force_on(hipStream_t s, const DevParams& P, const Sorted& st, const State& out, int n, const GridBuffers& g,
              float* force_dbg, int32_t* n_own_out, const NeighbourMask& nm, SweepMode mode) {
  const dim3 grid(div_up(n, B)), block(B);
  if (mode == SWEEP_EXACT)
    hipLaunchKernelGGL((k_force<B, SWEEP_EXACT, false>), grid, block, 0, s, P, st, out, n, g.start, g.mid, g.cell_s, force_dbg, n_own_out, nm);
  else if (mode == SWEEP_PLAIN)
    hipLaunchKernelGGL((k_force<B, SWEEP_PLAIN, false>), grid, block, 0, s, P, st, out, n, g.start, g.mid, g.cell_s, force_dbg, n_own_out, nm);
  else if (mode == SWEEP_FAST)
    hipLaunchKernelGGL((k_force<B, SWEEP_MASKED, true>), grid, block, 0, s, P, st, out, n, g.start, g.mid, g.cell_s, force_dbg, n_own_out, nm);
  else
    hipLaunchKernelGGL((k_force<B, SWEEP_MASKED, false>), grid, block, 0, s, P, st, out, n, g.start, g.mid, g.cell_s, force_dbg, n_own_out, nm);
}
}  // namespace

// The launches cover n slots (an upper bound of the range a pass works on); their block size follows n.
void launch_neighbours(hipStream_t s, const DevParams& P, const Sorted& st, int n, const GridBuffers& g,
                       const NeighbourMask& nm, bool with_density, int32_t* ncount) {
  if (n <= 0) return;
  if (n < kSmallLaunch) neighbours_on<kSmallBlock>(s, P, st, n, g, nm, with_density, ncount);
  else neighbours_on<kBlock>(s, P, st, n, g, nm, with_density, ncount);
}

void launch_density(hipStream_t s, const DevParams& P, const Sorted& st, int n, const GridBuffers& g,
                    int32_t* ncount, const NeighbourMask& nm, SweepMode mode) {
  if (n <= 0) return;
  if (n < kSmallLaunch) density_on<kSmallBlock>(s, P, st, n, g, ncount, nm, mode);
  else density_on<kBlock>(s, P, st, n, g, ncount, nm, mode);
}

void launch_force(hipStream_t s, const DevParams& P, const Sorted& st, const State& out, int n,
                  const GridBuffers& g, float* force_dbg, int32_t* n_own_out, const NeighbourMask& nm,
                  SweepMode mode) {
  if (n <= 0) return;
  if (n < kSmallLaunch) force_on<kSmallBlock>(s, P, st, out, n, g, force_dbg, n_own_out, nm, mode);
  else force_on<kBlock>(s, P, st, out, n, g, force_dbg, n_own_out, nm, mode);
}

}  // namespace sph
