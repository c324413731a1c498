// C ABI of libsph_amd.so (declared in include/sph_amd.h): context management, HBM buffers, step driver.
// Host-only code; every compute call ends in a kernel launch from sph_kernels.hip — there is no CPU path.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "sph_internal.hpp"
#include "sph_rccl.hpp"

using namespace sph;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(SPH_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

constexpr size_t kRecordBytes = SPH_SLAB_RECORD_WORDS * sizeof(float);

enum Cat { CAT_HASH, CAT_SCAN, CAT_REORDER, CAT_SEARCH, CAT_DENSITY, CAT_FORCE, CAT_OTHER, CAT_N };

struct EventPair {
  int cat;
  hipEvent_t a, b;
};

}  // namespace

struct sph_context {
  int device = 0;
  hipStream_t stream = nullptr;
  sph_params params{};
  DevParams dev{};
  int n_max = 0, n = 0;
  State cur{};      // SoA state the next step starts from (sorted order of the previous step)
  Sorted sorted{};  // this step's grid-ordered, gather-packed copy
  NeighbourMask nmask{};
  GridBuffers grid{};
  size_t cell_capacity = 0;
  DebugBuffers dbg{};
  float *stage_pos = nullptr, *stage_vel = nullptr;  // AoS staging in HBM, 3*n_max floats each
  void* stage_dbg = nullptr;                         // 6*n_max words for debug downloads
  bool uploaded = false, stepped = false;
  // slab decomposition: this context owns global x-columns [col_lo, col_hi) of gn; single-GPU = [0, gn)
  bool slab = false;
  bool own_stream = true;
  int gn = 0, col_lo = 0, col_hi = 0, cols_capacity = 0;
  int n_last_own = 0;                 // particles integrated by the last step (debug downloads)
  int dbg_stride = 0;                 // plane stride of dbg.force = particle count the last step was launched with
  DevParams sorted_dev{};             // grid (nx, own_lo, ...) of the last sort: what the start table and the intermediates
  int sorted_col_lo = 0, sorted_col_hi = 0;  // of that step are indexed by, whatever set_columns/set_params did since
  hipEvent_t wait_event = nullptr;    // sph_stream_wait
  int32_t* d_counters = nullptr;      // [0],[1] pack counters  [2] own count  [3] trash count after a step
                                      // [4] stray received records (cumulative)  [5] max column shift (float bits)
  int32_t* h_counters = nullptr;      // pinned mirror
  int32_t* d_colcounts = nullptr;     // [2 * (cols_capacity + 4)]: particles per local column, then work per local column
  bool n_pending = false;             // true count of `cur` still travelling to h_counters[2]
  // Records for the neighbours, packed right behind the force pass of a step into buffers of the context's own
  // (the counts travel home with the own count): the next sph_slab_pack then only copies them out.
  float* pre_rec[2] = {nullptr, nullptr};
  int pre_cap = 0;
  bool pre_side[2] = {false, false};  // which faces the caller packs for (as of its last sph_slab_pack)
  bool pre_valid = false;             // pre_rec and h_counters[0..1] describe the current `cur` and cuts
  int64_t lost = 0;
  // ---- exchange driven by the library (overlapped slab step) ----
  enum { XFER_NONE = 0, XFER_RCCL = 1, XFER_LOCAL = 2 };
  int xfer = XFER_NONE;
  void* rccl = nullptr;                    // ncclComm_t
  int rank = -1, nranks = 0;
  sph_context* peer[2] = {nullptr, nullptr};  // local transport: the neighbouring contexts of this process (left, right)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_pack = nullptr, ev_comm = nullptr;
  int xcap = 0;                            // records a message buffer holds behind its header record
  float* msg_out[2] = {nullptr, nullptr};  // [(xcap + 1) records] message to the left / right neighbour
  float* msg_in[2] = {nullptr, nullptr};   // message from the left / right neighbour
  int cap_out[2] = {0, 0}, cap_in[2] = {0, 0};  // records the next message to / from each neighbour holds (agreed sizes)
  int posted_out[2] = {0, 0}, posted_in[2] = {0, 0};  // the sizes the exchange in flight was posted with
  uint64_t remainder_rounds = 0;           // exchanges so far that needed a second round (a message outgrew its size)
  float size_headroom = 0.25f;             // next message = last true count * (1 + headroom), rounded up to 256 records,
  int size_min = 1024;                     // at least size_min, at most xcap: identical on every rank
  bool halo_current = false;               // `cur` holds own particles AND the neighbours' records for the next sort
  bool in_flight = false;                  // a begin() has posted an exchange that no end() has completed yet
  uint64_t begun = 0;                      // exchanges posted so far (LOCAL transport: a neighbour must have kept up)
  bool next_pending = false;
  int next_lo = 0, next_hi = 0;            // faces the NEXT exchange packs for (set_next_columns)
  float face_shift = 4.0f;                 // columns a particle may cross per step: how wide the face ranges are
  int overlap_mode = 2;                    // 0 one force launch, then pack; 1 faces first, interior beside the exchange; 2 by size
  int64_t missed = 0;
  // frame rendering (viewer hand-off): depth-keyed frame buffer and the resolved picture, grown on demand
  unsigned long long* frame_fb = nullptr;
  unsigned char* frame_rgb = nullptr;
  size_t frame_pixels = 0;
  int timing = 0;  // 0 off, 1 every kernel category, 2 the force pass only
  std::vector<EventPair> pending;
  std::vector<hipEvent_t> pool;
  double acc_ms[CAT_N] = {0, 0, 0, 0, 0, 0, 0};
  uint64_t acc_steps = 0, acc_particles = 0;
};

namespace {

// RAII bracket: records an event before and after the launches issued in its scope.
struct Timed {
  sph_context* c;
  EventPair p{};
  bool on;
  Timed(sph_context* ctx, int cat) : c(ctx), on(ctx->timing == 1 || (ctx->timing == 2 && cat == CAT_FORCE)) {
    if (!on) return;
    p.cat = cat;
    p.a = take();
    p.b = take();
    (void)hipEventRecord(p.a, c->stream);
  }
  ~Timed() {
    if (!on) return;
    (void)hipEventRecord(p.b, c->stream);
    c->pending.push_back(p);
  }
  hipEvent_t take() {
    if (!c->pool.empty()) {
      hipEvent_t e = c->pool.back();
      c->pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
};

int drain_timings(sph_context* c) {
  if (c->pending.empty()) return SPH_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (auto& p : c->pending) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) c->acc_ms[p.cat] += ms;
    c->pool.push_back(p.a);
    c->pool.push_back(p.b);
  }
  c->pending.clear();
  return SPH_OK;
}

bool finite_pos(float v) { return isfinite(v) && v > 0.0f; }

int check_params(const sph_params* p) {
  if (!p) return fail(SPH_ERR_INVALID, "params is NULL");
  if (!finite_pos(p->h)) return fail(SPH_ERR_INVALID, "h must be finite and > 0 (got %g)", p->h);
  if (!finite_pos(p->tank_size)) return fail(SPH_ERR_INVALID, "tank_size must be finite and > 0 (got %g)", p->tank_size);
  if (!finite_pos(p->mass) || !isfinite(p->rho0) || !isfinite(p->sound_speed) || !isfinite(p->dyn_visc) ||
      !isfinite(p->timestep) || !isfinite(p->surf_tension) || !isfinite(p->gravity[0]) ||
      !isfinite(p->gravity[1]) || !isfinite(p->gravity[2]))
    return fail(SPH_ERR_INVALID, "non-finite simulation parameter");
  if (p->cell_size != 0.0f && !(p->cell_size >= p->h))
    return fail(SPH_ERR_INVALID, "cell_size (%g) must be 0 or >= h (%g)", p->cell_size, p->h);
  return SPH_OK;
}

float resolved_cell(const sph_params* p) {
  return p->cell_size > 0.0f ? p->cell_size : p->h * 1.000244140625f;  // h * (1 + 2^-12)
}

int cells_per_axis(const sph_params* p) {
  double nc = ceil((double)p->tank_size / (double)resolved_cell(p));
  if (nc < 1.0) nc = 1.0;
  if (nc > 2048.0) return -1;
  return (int)nc;
}

// Largest fp32 x with sqrtf(x) <= h: the reference tests `sqrt(r2) <= h` (compute.cpp:144 and the
// guards at :44,:53,:64,:76,:88); on r2 that is `r2 <= r2_max`, and r2_max is NOT always h*h.
float largest_r2_within(float h) {
  float x = h * h;
  while (sqrtf(x) <= h) x = nextafterf(x, INFINITY);
  while (sqrtf(x) > h) x = nextafterf(x, -INFINITY);
  return x;
}

void derive(const sph_params* p, int nx, int cx_off, int gn, DevParams* d) {
  const float h = p->h;
  const float h2 = h * h;         // compute.cpp:100
  const float h6 = h2 * h2 * h2;  // :101
  const float h9 = h * h2 * h6;   // :102
  d->c_visc_lapl = 45.0f / (M_PI * h6);            // :45
  d->c_spiky_grad = -45.0f / (M_PI * h6);          // :56
  d->c_poly6 = 315.0f / (64.0f * M_PI * h9);       // :68
  d->c_poly6_grad = -945.0f / (32.0f * M_PI * h9); // :80
  d->c_poly6_lapl = 945.0f / (32.0f * M_PI * h9);  // :92
  d->h = h;
  d->h2 = h2;
  d->r2_max = largest_r2_within(h);
  d->m = p->mass;
  d->neg_m = -p->mass;
  d->rho0 = p->rho0;
  d->c2 = p->sound_speed * p->sound_speed;
  d->visc_m = p->dyn_visc * p->mass;
  d->gx = p->gravity[0];
  d->gy = p->gravity[1];
  d->gz = p->gravity[2];
  d->dt = p->timestep;
  d->tank = p->tank_size;
  d->neg_sigma = -p->surf_tension;
  d->cell = resolved_cell(p);
  d->nx = nx;
  d->ny = gn;
  d->nz = gn;
  division_constants((uint32_t)d->ny, &d->ny_mul, &d->ny_shift);
  division_constants((uint32_t)d->nz, &d->nz_mul, &d->nz_shift);
  d->cx_off = cx_off;
  d->gnx = gn;
  d->tension = (p->flags & SPH_FLAG_SURFACE_TENSION) ? 1 : 0;
  d->k_dens = (float)((double)p->mass * d->c_poly6);
  d->k_press = (float)(-0.5 * (double)p->mass * d->c_spiky_grad);
  d->k_visc = (float)((double)d->visc_m * d->c_visc_lapl);
  d->k_tgrad = (float)((double)p->mass * d->c_poly6_grad);
  d->k_tlapl = (float)((double)p->mass * d->c_poly6_lapl);
  d->c2rho0 = d->c2 * p->rho0;
}

// Where a context's device buffers come from: one hipMalloc each (sph_create), or consecutive 256-byte-aligned
// pieces of a region that already exists (the no-allocation entry point).
struct MallocSource {
  int take(void** p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes ? bytes : 1));
    return SPH_OK;
  }
};
struct RegionSource {
  unsigned char* base;
  size_t size, used = 0;
  RegionSource(void* b, size_t s) : base((unsigned char*)b), size(s) {}
  int take(void** p, size_t bytes) {
    const size_t at = (used + 255) & ~(size_t)255;
    if (at + bytes > size) return fail(SPH_ERR_CAPACITY, "device region of %zu bytes is too small", size);
    *p = base + at;
    used = at + bytes;
    return SPH_OK;
  }
};

template <class Source, class T>
int dev_alloc(Source& src, T** p, size_t count) {
  *p = nullptr;
  return src.take((void**)p, (count ? count : 1) * sizeof(T));
}

// Cells of a context's tables, rounded up to whole scan tiles (the scan kernels move int4s).
size_t padded_cells(size_t ncells) { return ((ncells + kScanItems - 1) / kScanItems) * kScanItems; }

// Every device buffer one step touches, for n particles and ncells cells.
template <class Source>
int layout_step_buffers(Source& src, sph_context* c, size_t n, size_t ncells) {
  int rc;
  State* s = &c->cur;
  if ((rc = dev_alloc(src, &s->x, n)) || (rc = dev_alloc(src, &s->y, n)) || (rc = dev_alloc(src, &s->z, n)) ||
      (rc = dev_alloc(src, &s->vx, n)) || (rc = dev_alloc(src, &s->vy, n)) || (rc = dev_alloc(src, &s->vz, n)) ||
      (rc = dev_alloc(src, &s->id, n)))
    return rc;
  if ((rc = dev_alloc(src, &c->sorted.pos3, 3 * (n + kSearchOverread))) || (rc = dev_alloc(src, &c->sorted.rec, n)) ||
      (rc = dev_alloc(src, &c->sorted.id, n)) || (rc = dev_alloc(src, &c->nmask.words, (size_t)(kMaskWords + 2) * n)) ||
      (rc = dev_alloc(src, &c->nmask.ncand, n)) || (rc = dev_alloc(src, &c->nmask.nnb, n)))
    return rc;
  c->nmask.stride = n;
  const size_t padded = padded_cells(ncells);
  if ((rc = dev_alloc(src, &c->grid.count, padded)) || (rc = dev_alloc(src, &c->grid.start, padded + 4)) ||
      (rc = dev_alloc(src, &c->grid.mid, padded + 4)) ||
      (rc = dev_alloc(src, &c->grid.block_sums, padded / kScanItems + 1)) || (rc = dev_alloc(src, &c->grid.cell_of, n)) ||
      (rc = dev_alloc(src, &c->grid.rank_of, n)) || (rc = dev_alloc(src, &c->grid.src_of, n)) ||
      (rc = dev_alloc(src, &c->grid.id_prov, n)) || (rc = dev_alloc(src, &c->grid.cell_s, n)))
    return rc;
  return SPH_OK;
}

void free_state(State* s) {
  (void)hipFree(s->x); (void)hipFree(s->y); (void)hipFree(s->z);
  (void)hipFree(s->vx); (void)hipFree(s->vy); (void)hipFree(s->vz);
  (void)hipFree(s->id);
  *s = State{};
}

int set_grid(sph_context* c, const sph_params* p, int col_lo, int col_hi) {
  c->pre_valid = false;  // packed for the old cut planes / cell size
  int gn = cells_per_axis(p);
  if (gn < 0) return fail(SPH_ERR_CAPACITY, "tank_size/cell_size needs more than 2048 cells per axis");
  if (!c->slab) { col_lo = 0; col_hi = gn; }
  if (col_lo < 0 || col_hi > gn || col_hi <= col_lo)
    return fail(SPH_ERR_INVALID, "slab columns [%d, %d) outside the %d-column grid", col_lo, col_hi, gn);
  if (c->slab && col_hi - col_lo < SPH_SLAB_MIN_COLUMNS && !(col_lo == 0 && col_hi == gn))
    return fail(SPH_ERR_INVALID, "slab [%d, %d) is narrower than %d columns", col_lo, col_hi, SPH_SLAB_MIN_COLUMNS);
  // two ghost columns per interior face
  const int g_lo = col_lo - 2 < 0 ? 0 : col_lo - 2;
  const int g_hi = col_hi + 2 > gn ? gn : col_hi + 2;
  const int nx = g_hi - g_lo;
  size_t ncells = (size_t)nx * gn * gn + 1;  // + trash cell
  if (ncells > c->cell_capacity)
    return fail(SPH_ERR_CAPACITY, "local grid of %d x %d x %d cells exceeds the context's capacity of %zu cells", nx,
                gn, gn, c->cell_capacity);
  if (c->gn != gn || c->col_lo != col_lo || c->col_hi != col_hi || c->dev.cell != resolved_cell(p))
    c->halo_current = false;  // other cells or other faces: the neighbours' records held no longer fit
  derive(p, nx, g_lo, gn, &c->dev);
  c->dev.own_lo = col_lo - g_lo;
  c->dev.own_hi = col_hi - g_lo;
  c->dev.force_lo = c->dev.own_lo;
  c->dev.force_hi = c->dev.own_hi;
  c->dev.dens_lo = c->dev.own_lo > 0 ? c->dev.own_lo - 1 : 0;
  c->dev.dens_hi = c->dev.own_hi < nx ? c->dev.own_hi + 1 : nx;
  c->grid.ncells = ncells;
  c->grid.scan_blocks = (ncells + kScanItems - 1) / kScanItems;
  c->params = *p;
  c->gn = gn;
  c->col_lo = col_lo;
  c->col_hi = col_hi;
  return SPH_OK;
}

// Slab mode: the number of particles a step keeps is only known on the device; fetch it when needed.
int resolve_n(sph_context* c) {
  if (!c->n_pending) return SPH_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->n = c->h_counters[2];
  c->n_last_own = c->n;
  c->lost = c->h_counters[4];  // cumulative on the device
  c->n_pending = false;
  return SPH_OK;
}

int one_step(sph_context* c) {
  const bool exact = (c->params.flags & SPH_FLAG_EXACT_ORDER) != 0;
  const bool keep = (c->params.flags & SPH_FLAG_KEEP_DEBUG) != 0;
  int rc = resolve_n(c);
  if (rc) return rc;
  const int n = c->n;
  { Timed t(c, CAT_HASH); launch_hash(c->stream, c->dev, c->cur, n, c->grid); }
  { Timed t(c, CAT_SCAN); launch_scan(c->stream, c->grid); }
  { Timed t(c, CAT_REORDER); launch_reorder(c->stream, c->cur, c->sorted, n, c->grid); }
  SweepMode mode = SWEEP_FAST;
  if (exact) mode = SWEEP_EXACT;
  else if (c->params.flags & SPH_FLAG_LANE_SWEEP) mode = SWEEP_PLAIN;
  else if (c->params.flags & SPH_FLAG_REFERENCE_ARITH) mode = SWEEP_MASKED;
  if (mode == SWEEP_MASKED || mode == SWEEP_FAST) {  // fast arithmetic: the search computes the density on the way
    Timed t(c, CAT_SEARCH);
    launch_neighbours(c->stream, c->dev, c->sorted, n, c->grid, c->nmask, mode == SWEEP_FAST,
                      keep ? c->dbg.ncount : nullptr);
  }
  if (mode != SWEEP_FAST) {
    Timed t(c, CAT_DENSITY);
    launch_density(c->stream, c->dev, c->sorted, n, c->grid, keep ? c->dbg.ncount : nullptr, c->nmask, mode);
  }
  State out = c->cur;
  if (!c->slab) out.id = nullptr;  // same particles, same order: the id array is swapped below instead
  { Timed t(c, CAT_FORCE);
    launch_force(c->stream, c->dev, c->sorted, out, n, c->grid, keep ? c->dbg.force : nullptr,
                 c->slab ? c->d_counters + 2 : nullptr, c->nmask, mode); }
  HIP_TRY(hipGetLastError());
  c->dbg_stride = n;
  c->sorted_dev = c->dev;
  c->sorted_col_lo = c->col_lo;
  c->sorted_col_hi = c->col_hi;
  if (c->slab) {
    // ghosts and emigrants drop out here: only own-column particles were written back
    c->pre_valid = false;
    c->halo_current = false;
    if (c->pre_cap > 0) {
      const bool has_left = c->col_lo > 0 && c->pre_side[0], has_right = c->col_hi < c->gn && c->pre_side[1];
      HIP_TRY(hipMemsetAsync(c->d_counters, 0, 2 * sizeof(int32_t), c->stream));
      Timed t(c, CAT_OTHER);
      launch_slab_pack(c->stream, c->dev, c->cur, n, c->d_counters + 2, c->col_lo, c->col_hi, has_left, has_right,
                       c->pre_rec[0], c->pre_rec[1], c->pre_cap, c->d_counters);
      c->pre_valid = true;
    }
    if (n == 0) HIP_TRY(hipMemsetAsync(c->d_counters, 0, 4 * sizeof(int32_t), c->stream));  // no kernel ran to write them
    HIP_TRY(hipMemcpyAsync(c->h_counters, c->d_counters, 5 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    c->n_pending = true;
  } else {
    int32_t* t = c->cur.id;
    c->cur.id = c->sorted.id;
    c->sorted.id = t;
    c->n_last_own = n;
  }
  c->stepped = true;
  c->acc_steps += 1;
  c->acc_particles += (uint64_t)n;
  if (c->pending.size() > 8192) return drain_timings(c);
  return SPH_OK;
}

int create_common(const sph_params* params, int n_max, int device, bool slab, int col_lo, int col_hi,
                  int cols_capacity, sph_context** out);

}  // namespace

namespace {
int create_common(const sph_params* params, int n_max, int device, bool slab, int col_lo, int col_hi,
                  int cols_capacity, sph_context** out) {
  if (!out) return fail(SPH_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int rc = check_params(params);
  if (rc) return rc;
  if (n_max < 1) return fail(SPH_ERR_INVALID, "n_max must be >= 1 (got %d)", n_max);
  int gn = cells_per_axis(params);
  if (gn < 0) return fail(SPH_ERR_CAPACITY, "tank_size/cell_size needs more than 2048 cells per axis");
  if (!slab) { col_lo = 0; col_hi = gn; cols_capacity = gn; }
  if (cols_capacity < col_hi - col_lo) cols_capacity = col_hi - col_lo;
  if (cols_capacity > gn) cols_capacity = gn;
  int cap_cols = cols_capacity + 4 > gn ? gn : cols_capacity + 4;
  size_t ncells = (size_t)cap_cols * gn * gn + 1;
  if (ncells + 1 > (size_t)0x7fffffff) return fail(SPH_ERR_CAPACITY, "%zu cells overflow the 32-bit cell index", ncells);

  int ndev = sph_device_count();
  if (ndev <= 0) return fail(SPH_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU path");
  if (device < 0 || device >= ndev) return fail(SPH_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, ndev - 1);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(SPH_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
  HIP_TRY(hipSetDevice(device));

  sph_context* c = new sph_context();
  c->device = device;
  c->n_max = n_max;
  c->slab = slab;
  c->cols_capacity = cols_capacity;
  c->cell_capacity = ncells;
  auto bail = [&](int code) { sph_destroy(c); return code; };
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(SPH_ERR_HIP, "hipStreamCreate failed"));
  const size_t n = (size_t)n_max;
  MallocSource heap;
  if ((rc = layout_step_buffers(heap, c, n, ncells))) return bail(rc);
  const size_t padded = padded_cells(ncells);
  if ((rc = dev_alloc(heap, &c->dbg.force, 3 * n)) || (rc = dev_alloc(heap, &c->dbg.ncount, n)) ||
      (rc = dev_alloc(heap, &c->stage_pos, 3 * n)) || (rc = dev_alloc(heap, &c->stage_vel, 3 * n)) ||
      (rc = dev_alloc(heap, (uint32_t**)&c->stage_dbg, 6 * n)) || (rc = dev_alloc(heap, &c->d_counters, 8)) ||
      (rc = dev_alloc(heap, &c->d_colcounts, 2 * ((size_t)cap_cols + 4))))
    return bail(rc);
  if (hipHostMalloc((void**)&c->h_counters, 16 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess)
    return bail(fail(SPH_ERR_HIP, "hipHostMalloc failed"));
  memset(c->h_counters, 0, 16 * sizeof(int32_t));
  if (hipMemsetAsync(c->grid.count, 0, padded * sizeof(int32_t), c->stream) != hipSuccess ||
      hipMemsetAsync(c->d_counters, 0, 8 * sizeof(int32_t), c->stream) != hipSuccess)
    return bail(fail(SPH_ERR_HIP, "hipMemsetAsync failed"));
  if ((rc = set_grid(c, params, col_lo, col_hi))) return bail(rc);
  if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(fail(SPH_ERR_HIP, "stream sync failed"));
  *out = c;
  return SPH_OK;
}
}  // namespace

extern "C" {

const char* sph_version(void) { return "sph_amd 0.1 gfx950"; }
const char* sph_last_error(void) { return g_err; }

int sph_params_default(sph_params* p) {
  if (!p) return fail(SPH_ERR_INVALID, "params is NULL");
  memset(p, 0, sizeof *p);
  p->sound_speed = 1.0f;
  p->mass = 0.00020543f;
  p->rho0 = 600.0f;
  p->dyn_visc = 0.2f;
  p->gravity[0] = 0.0f;
  p->gravity[1] = -10.0f;
  p->gravity[2] = 0.0f;
  p->timestep = 0.005f;
  p->tank_size = 0.2154f;
  p->surf_tension = 0.0728f;
  p->h = 0.01f;
  p->cell_size = 0.0f;
  p->flags = SPH_FLAG_SURFACE_TENSION;
  return SPH_OK;
}

int sph_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sph_create(const sph_params* params, int n_max, int device, sph_context** out) {
  return create_common(params, n_max, device, false, 0, 0, 0, out);
}

int sph_destroy(sph_context* c) {
  if (!c) return SPH_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : c->pool) (void)hipEventDestroy(e);
  if (c->wait_event) (void)hipEventDestroy(c->wait_event);
  free_state(&c->cur);
  (void)hipFree(c->sorted.pos3); (void)hipFree(c->sorted.rec); (void)hipFree(c->sorted.id);
  (void)hipFree(c->nmask.words); (void)hipFree(c->nmask.ncand); (void)hipFree(c->nmask.nnb);
  (void)hipFree(c->grid.count); (void)hipFree(c->grid.start); (void)hipFree(c->grid.mid);
  (void)hipFree(c->grid.block_sums);
  (void)hipFree(c->grid.cell_of); (void)hipFree(c->grid.rank_of); (void)hipFree(c->grid.src_of);
  (void)hipFree(c->grid.id_prov); (void)hipFree(c->grid.cell_s);
  (void)hipFree(c->dbg.force); (void)hipFree(c->dbg.ncount);
  (void)hipFree(c->stage_pos); (void)hipFree(c->stage_vel); (void)hipFree(c->stage_dbg);
  (void)hipFree(c->d_counters); (void)hipFree(c->d_colcounts);
  (void)hipFree(c->pre_rec[0]); (void)hipFree(c->pre_rec[1]);
  for (int k = 0; k < 2; k++) { (void)hipFree(c->msg_out[k]); (void)hipFree(c->msg_in[k]); }
  if (c->comm_stream) { (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamDestroy(c->comm_stream); }
  if (c->ev_pack) (void)hipEventDestroy(c->ev_pack);
  if (c->ev_comm) (void)hipEventDestroy(c->ev_comm);
  if (c->rccl) rccl_comm_destroy(c->rccl);
  if (c->h_counters) (void)hipHostFree(c->h_counters);
  (void)hipFree(c->frame_fb);
  (void)hipFree(c->frame_rgb);
  if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return SPH_OK;
}

int sph_set_params(sph_context* c, const sph_params* p) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  int rc = check_params(p);
  if (rc) return rc;
  return set_grid(c, p, c->col_lo, c->col_hi);
}

int sph_get_params(const sph_context* c, sph_params* p) {
  if (!c || !p) return fail(SPH_ERR_INVALID, "NULL argument");
  *p = c->params;
  return SPH_OK;
}

static int upload_common(sph_context* c, const float* pos, const float* vel, int n, hipMemcpyKind kind) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (n < 0 || n > c->n_max) return fail(SPH_ERR_CAPACITY, "n = %d outside 0..n_max = %d", n, c->n_max);
  if (n > 0 && (!pos || !vel)) return fail(SPH_ERR_INVALID, "positions/velocities pointer is NULL");
  HIP_TRY(hipSetDevice(c->device));
  if (n > 0) {
    HIP_TRY(hipMemcpyAsync(c->stage_pos, pos, (size_t)n * 3 * sizeof(float), kind, c->stream));
    HIP_TRY(hipMemcpyAsync(c->stage_vel, vel, (size_t)n * 3 * sizeof(float), kind, c->stream));
    Timed t(c, CAT_OTHER);
    launch_split_aos(c->stream, c->stage_pos, c->stage_vel, n, c->cur);
  }
  HIP_TRY(hipGetLastError());
  if (kind == hipMemcpyHostToDevice) HIP_TRY(hipStreamSynchronize(c->stream));  // caller may reuse its buffers
  c->n = n;
  c->n_pending = false;
  c->uploaded = true;
  c->stepped = false;
  c->pre_valid = false;
  c->halo_current = false;
  return SPH_OK;
}

int sph_upload(sph_context* c, const float* pos, const float* vel, int n) {
  return upload_common(c, pos, vel, n, hipMemcpyHostToDevice);
}
int sph_upload_device(sph_context* c, const float* pos, const float* vel, int n) {
  return upload_common(c, pos, vel, n, hipMemcpyDeviceToDevice);
}

int sph_step(sph_context* c, int nsteps) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (nsteps < 0) return fail(SPH_ERR_INVALID, "nsteps < 0");
  if (!c->uploaded) return fail(SPH_ERR_STATE, "sph_step before sph_upload");
  HIP_TRY(hipSetDevice(c->device));
  for (int s = 0; s < nsteps; s++) {
    int rc = one_step(c);
    if (rc) return rc;
  }
  return SPH_OK;
}

int sph_step_profile(sph_context* c, int nsteps, float* step_ms) {
  if (!c || !step_ms) return fail(SPH_ERR_INVALID, "NULL argument");
  if (nsteps < 0) return fail(SPH_ERR_INVALID, "nsteps < 0");
  if (!c->uploaded) return fail(SPH_ERR_STATE, "sph_step_profile before sph_upload");
  HIP_TRY(hipSetDevice(c->device));
  std::vector<hipEvent_t> ev((size_t)nsteps + 1, nullptr);
  int rc = SPH_OK;
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) rc = fail(SPH_ERR_HIP, "hipEventCreate failed");
  for (int s = 0; s < nsteps && rc == SPH_OK; s++) {
    if (hipEventRecord(ev[s], c->stream) != hipSuccess) rc = fail(SPH_ERR_HIP, "hipEventRecord failed");
    if (rc == SPH_OK) rc = one_step(c);
  }
  if (rc == SPH_OK && hipEventRecord(ev[nsteps], c->stream) != hipSuccess) rc = fail(SPH_ERR_HIP, "hipEventRecord failed");
  if (rc == SPH_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(SPH_ERR_HIP, "stream sync failed");
  for (int s = 0; s < nsteps && rc == SPH_OK; s++)
    if (hipEventElapsedTime(&step_ms[s], ev[s], ev[s + 1]) != hipSuccess) rc = fail(SPH_ERR_HIP, "hipEventElapsedTime failed");
  for (auto e : ev)
    if (e) (void)hipEventDestroy(e);
  return rc;
}

int sph_sync(sph_context* c) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return SPH_OK;
}

static int download_common(sph_context* c, float* pos, float* vel, hipMemcpyKind kind) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->uploaded) return fail(SPH_ERR_STATE, "download before upload");
  if (c->slab) return fail(SPH_ERR_STATE, "slab contexts hold global ids: use sph_slab_download");
  HIP_TRY(hipSetDevice(c->device));
  const int n = c->n;
  if (n > 0 && (pos || vel)) {
    {
      Timed t(c, CAT_OTHER);
      launch_gather_aos(c->stream, c->cur, n, pos ? c->stage_pos : nullptr, vel ? c->stage_vel : nullptr);
    }
    HIP_TRY(hipGetLastError());
    if (pos) HIP_TRY(hipMemcpyAsync(pos, c->stage_pos, (size_t)n * 3 * sizeof(float), kind, c->stream));
    if (vel) HIP_TRY(hipMemcpyAsync(vel, c->stage_vel, (size_t)n * 3 * sizeof(float), kind, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return SPH_OK;
}

int sph_download(sph_context* c, float* pos, float* vel) { return download_common(c, pos, vel, hipMemcpyDeviceToHost); }
int sph_download_device(sph_context* c, float* pos, float* vel) {
  return download_common(c, pos, vel, hipMemcpyDeviceToDevice);
}

int sph_download_debug(sph_context* c, float* rho, float* force, int32_t* ncount, int32_t* cell) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->stepped) return fail(SPH_ERR_STATE, "sph_download_debug needs at least one step after the upload");
  if ((force || ncount) && !(c->params.flags & SPH_FLAG_KEEP_DEBUG))
    return fail(SPH_ERR_STATE, "force/ncount need SPH_FLAG_KEEP_DEBUG set before the step");
  if (c->slab) return fail(SPH_ERR_STATE, "slab contexts hold global ids: use sph_slab_download_debug");
  HIP_TRY(hipSetDevice(c->device));
  const size_t n = (size_t)c->n;
  if (n == 0) return SPH_OK;
  float* d_rho = (float*)c->stage_dbg;
  float* d_force = d_rho + n;
  int32_t* d_ncount = (int32_t*)(d_force + 3 * n);
  int32_t* d_cell = d_ncount + n;
  launch_gather_debug(c->stream, c->cur.id, (int)n, c->sorted.rec, c->dbg.force, c->dbg.ncount, c->grid.cell_s,
                      rho ? d_rho : nullptr, force ? d_force : nullptr, ncount ? d_ncount : nullptr,
                      cell ? d_cell : nullptr);
  HIP_TRY(hipGetLastError());
  if (rho) HIP_TRY(hipMemcpyAsync(rho, d_rho, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (force) HIP_TRY(hipMemcpyAsync(force, d_force, 3 * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (ncount) HIP_TRY(hipMemcpyAsync(ncount, d_ncount, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (cell) HIP_TRY(hipMemcpyAsync(cell, d_cell, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return SPH_OK;
}

int sph_camera_default(sph_camera* cam) {
  if (!cam) return fail(SPH_ERR_INVALID, "cam is NULL");
  cam->translation[0] = 0.0f; cam->translation[1] = 0.0f; cam->translation[2] = -1.5f;  // render.cpp:26
  cam->quat[0] = 0.0f; cam->quat[1] = 0.0f; cam->quat[2] = 0.0f; cam->quat[3] = 1.0f;     // trackball(q, 0, 0, 0, 0)
  cam->fov_deg = 45.0f;   // render.cpp:157
  cam->z_near = 0.01f;    // :155
  cam->z_far = 10.0f;     // :156
  cam->radius = 0.0025f;  // :187
  return SPH_OK;
}

int sph_render_frame(sph_context* c, const sph_camera* cam, int width, int height, unsigned char* rgb) {
  if (!c || !cam || !rgb) return fail(SPH_ERR_INVALID, "NULL argument");
  if (width < 1 || height < 1 || width > 16384 || height > 16384)
    return fail(SPH_ERR_INVALID, "frame size %d x %d outside [1, 16384]", width, height);
  if (!c->uploaded) return fail(SPH_ERR_STATE, "sph_render_frame before upload");
  if (!(cam->fov_deg > 0.0f && cam->fov_deg < 180.0f) || !(cam->z_near > 0.0f) || !(cam->z_far > cam->z_near) ||
      !(cam->radius > 0.0f))
    return fail(SPH_ERR_INVALID, "camera needs 0 < fov < 180, 0 < z_near < z_far, radius > 0");
  HIP_TRY(hipSetDevice(c->device));
  int rc = resolve_n(c);
  if (rc != SPH_OK) return rc;
  const size_t pixels = (size_t)width * height;
  if (pixels > c->frame_pixels) {
    (void)hipFree(c->frame_fb);
    (void)hipFree(c->frame_rgb);
    c->frame_fb = nullptr;
    c->frame_rgb = nullptr;
    c->frame_pixels = 0;
    HIP_TRY(hipMalloc((void**)&c->frame_fb, pixels * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void**)&c->frame_rgb, pixels * 3));
    c->frame_pixels = pixels;
  }
  RenderCamera rcam;
  const float* q = cam->quat;
  // build_rotmatrix(glm::mat4&, q) fills m[column][row] (render/trackball.cpp:326-346); M[row][column] below
  rcam.m[0][0] = 1.0f - 2.0f * (q[1] * q[1] + q[2] * q[2]);
  rcam.m[1][0] = 2.0f * (q[0] * q[1] - q[2] * q[3]);
  rcam.m[2][0] = 2.0f * (q[2] * q[0] + q[1] * q[3]);
  rcam.m[0][1] = 2.0f * (q[0] * q[1] + q[2] * q[3]);
  rcam.m[1][1] = 1.0f - 2.0f * (q[2] * q[2] + q[0] * q[0]);
  rcam.m[2][1] = 2.0f * (q[1] * q[2] - q[0] * q[3]);
  rcam.m[0][2] = 2.0f * (q[2] * q[0] - q[1] * q[3]);
  rcam.m[1][2] = 2.0f * (q[1] * q[2] + q[0] * q[3]);
  rcam.m[2][2] = 1.0f - 2.0f * (q[1] * q[1] + q[0] * q[0]);
  for (int a = 0; a < 3; a++) rcam.t[a] = cam->translation[a];
  rcam.focal = 1.0f / tanf(0.5f * cam->fov_deg * 3.14159265358979323846f / 180.0f);
  rcam.aspect = (float)width / (float)height;
  rcam.z_near = cam->z_near;
  rcam.z_far = cam->z_far;
  rcam.radius = cam->radius;
  {
    Timed t(c, CAT_OTHER);
    launch_render(c->stream, c->cur, c->n, rcam, width, height, c->frame_fb, c->frame_rgb);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(rgb, c->frame_rgb, pixels * 3, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return SPH_OK;
}

int sph_num_particles(const sph_context* c) {
  if (!c) return 0;
  if (c->n_pending) (void)resolve_n(const_cast<sph_context*>(c));
  return c->n;
}

int sph_slab_last_own(const sph_context* c) {
  if (!c) return 0;
  if (c->n_pending) (void)resolve_n(const_cast<sph_context*>(c));
  return c->n_last_own;
}

int sph_grid_dims(const sph_context* c, int dims[3]) {
  if (!c || !dims) return fail(SPH_ERR_INVALID, "NULL argument");
  dims[0] = c->dev.nx;
  dims[1] = c->dev.ny;
  dims[2] = c->dev.nz;
  return SPH_OK;
}

int sph_enable_timing(sph_context* c, int enable) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  c->timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
  return SPH_OK;
}

int sph_get_timings(sph_context* c, sph_timings* out) {
  if (!c || !out) return fail(SPH_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  int rc = drain_timings(c);
  if (rc) return rc;
  out->hash_ms = c->acc_ms[CAT_HASH];
  out->scan_ms = c->acc_ms[CAT_SCAN];
  out->reorder_ms = c->acc_ms[CAT_REORDER];
  out->search_ms = c->acc_ms[CAT_SEARCH];
  out->density_ms = c->acc_ms[CAT_DENSITY];
  out->force_ms = c->acc_ms[CAT_FORCE];
  out->other_ms = c->acc_ms[CAT_OTHER];
  out->steps = c->acc_steps;
  out->particles = c->acc_particles;
  return SPH_OK;
}

int sph_reset_timings(sph_context* c) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  HIP_TRY(hipSetDevice(c->device));
  int rc = drain_timings(c);
  if (rc) return rc;
  for (double& v : c->acc_ms) v = 0.0;
  c->acc_steps = 0;
  c->acc_particles = 0;
  return SPH_OK;
}

// ---- slab decomposition ------------------------------------------------------------------------------------------

int sph_global_cells(const sph_params* p, int* cells) {
  int rc = check_params(p);
  if (rc) return rc;
  if (!cells) return fail(SPH_ERR_INVALID, "NULL argument");
  int gn = cells_per_axis(p);
  if (gn < 0) return fail(SPH_ERR_CAPACITY, "tank_size/cell_size needs more than 2048 cells per axis");
  *cells = gn;
  return SPH_OK;
}

int sph_column_of(const sph_params* p, float x, int* column) {
  int rc = check_params(p);
  if (rc) return rc;
  if (!column) return fail(SPH_ERR_INVALID, "NULL argument");
  int gn = cells_per_axis(p);
  if (gn < 0) return fail(SPH_ERR_CAPACITY, "tank_size/cell_size needs more than 2048 cells per axis");
  // the expression of the device's cell_coord (fp32 division, then clamp; a NaN lands in column 0)
  float q = floorf(x / resolved_cell(p));
  q = fminf(fmaxf(q, 0.0f), (float)(gn - 1));
  *column = (int)q;
  return SPH_OK;
}

// ---- cut planes (host arithmetic only) ---------------------------------------------------------------------------

namespace {

int ceil_columns(float v) { return (int)ceilf(v > 0.0f ? v : 0.0f); }

// cuts[0..nslabs]: boundaries whose cumulative load is closest to r/nslabs of the total, every slab >= min_cols wide.
void balanced_cuts(const int64_t* load, int ncols, int nslabs, int min_cols, std::vector<int>* out) {
  std::vector<int64_t> csum((size_t)ncols + 1, 0);
  for (int i = 0; i < ncols; i++) csum[i + 1] = csum[i] + load[i];
  const int64_t total = csum[ncols];
  std::vector<int>& cuts = *out;
  cuts.assign(1, 0);
  for (int r = 1; r < nslabs; r++) {
    const double target = (double)(total * r) / (double)nslabs;
    int c = 0;
    while (c < ncols && (double)csum[c] < target) c++;  // first boundary at or beyond the target
    if (c > 0 && fabs((double)csum[c - 1] - target) <= fabs((double)csum[c] - target)) c--;
    const int lo = cuts.back() + min_cols, hi = ncols - (nslabs - r) * min_cols;
    cuts.push_back(c < lo ? lo : (c > hi ? hi : c));
  }
  cuts.push_back(ncols);
}

int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace

int sph_slab_plan_partition(const int64_t* load, int ncols, int nslabs, int min_cols, int32_t* cuts) {
  if (!load || !cuts) return fail(SPH_ERR_INVALID, "NULL argument");
  if (ncols < 1 || nslabs < 1 || min_cols < 1) return fail(SPH_ERR_INVALID, "ncols, nslabs and min_cols must be >= 1");
  if ((int64_t)nslabs * min_cols > ncols)
    return fail(SPH_ERR_INVALID, "%d columns cannot hold %d slabs of >= %d columns", ncols, nslabs, min_cols);
  for (int i = 0; i < ncols; i++)
    if (load[i] < 0) return fail(SPH_ERR_INVALID, "load[%d] is negative", i);
  std::vector<int> v;
  balanced_cuts(load, ncols, nslabs, min_cols, &v);
  for (int r = 0; r <= nslabs; r++) cuts[r] = v[r];
  return SPH_OK;
}

int sph_slab_plan_next(const int64_t* load, int ncols, int nslabs, const int32_t* cuts, float column_shift,
                       float shift_margin, const int32_t* cols_capacity, int32_t* next_cuts, int* face_move,
                       float* face_shift) {
  if (!load || !cuts || !next_cuts) return fail(SPH_ERR_INVALID, "NULL argument");
  if (ncols < 1 || nslabs < 1) return fail(SPH_ERR_INVALID, "ncols and nslabs must be >= 1");
  if (!(column_shift >= 0.0f) || !(shift_margin >= 0.0f)) return fail(SPH_ERR_INVALID, "shifts must be >= 0");
  if (cuts[0] != 0 || cuts[nslabs] != ncols) return fail(SPH_ERR_INVALID, "cuts must run from 0 to ncols");
  int narrowest = ncols;
  for (int r = 0; r < nslabs; r++) {
    if (cuts[r + 1] <= cuts[r]) return fail(SPH_ERR_INVALID, "cuts must increase");
    narrowest = std::min(narrowest, cuts[r + 1] - cuts[r]);
  }
  for (int i = 0; i < ncols; i++)
    if (load[i] < 0) return fail(SPH_ERR_INVALID, "load[%d] is negative", i);

  // Columns a plane may travel: as many as the fastest particle crosses, so that planes can follow a front, within
  // what one exchange covers; one column less than the narrowest slab could afford, so that a slab under pressure
  // (fluid piling up at a wall) can still narrow by a column per call.
  int move = std::min(SPH_SLAB_MAX_FACE_MOVE, std::max(1, ceil_columns(column_shift)));
  move = std::max(1, std::min(move, narrowest - ceil_columns(column_shift) - 3));
  const float shift = column_shift + shift_margin;
  // narrowest slab one exchange per step can serve: ceil(shift) + 2 ghost columns + the planes' own travel
  int min_cols = std::max(SPH_SLAB_MIN_COLUMNS, ceil_columns(shift) + 2 + move);
  min_cols = std::min(min_cols, ncols / nslabs);  // never more than the tank can give
  if (min_cols < 1) return fail(SPH_ERR_INVALID, "%d columns cannot hold %d slabs", ncols, nslabs);

  std::vector<int> ideal, now(cuts, cuts + nslabs + 1), next;
  balanced_cuts(load, ncols, nslabs, min_cols, &ideal);
  next = now;
  for (int r = 1; r < nslabs; r++) next[r] = now[r] + clampi(ideal[r] - now[r], -move, move);
  for (int r = 1; r < nslabs; r++) next[r] = std::max(next[r], next[r - 1] + min_cols);  // minimum widths ...
  for (int r = nslabs - 1; r > 0; r--) next[r] = std::min(next[r], next[r + 1] - min_cols);
  for (int r = 1; r < nslabs; r++) next[r] = clampi(next[r], now[r] - move, now[r] + move);  // ... never beyond `move`
  if (cols_capacity) {
    const int32_t* cap = cols_capacity;
    std::vector<int> fit = next;
    for (int r = 0; r + 1 < nslabs; r++) fit[r + 1] = std::min(fit[r + 1], fit[r] + cap[r]);  // too wide: columns go right
    for (int r = nslabs - 1; r > 0; r--) fit[r] = std::max(fit[r], fit[r + 1] - cap[r]);      // ... and the last ones left
    bool ok = true, too_wide = false;
    for (int r = 0; r < nslabs; r++) {
      const int w = fit[r + 1] - fit[r];
      ok = ok && w <= cap[r] && w >= min_cols;
      too_wide = too_wide || next[r + 1] - next[r] > cap[r];
    }
    for (int r = 1; r < nslabs; r++) ok = ok && abs(fit[r] - now[r]) <= move;
    if (ok) {
      next = fit;
    } else if (too_wide) {
      // plane by plane from where they stand: a plane takes its move if both slabs beside it stay admissible
      std::vector<int> work = now;
      for (int r = 1; r < nslabs; r++) {
        const int left = next[r] - work[r - 1], right = work[r + 1] - next[r];
        if (left >= min_cols && left <= cap[r - 1] && right >= min_cols && right <= cap[r]) work[r] = next[r];
      }
      next = work;
    }
  }
  for (int r = 0; r <= nslabs; r++) next_cuts[r] = next[r];
  if (face_move) *face_move = move;
  if (face_shift) *face_shift = shift;
  return SPH_OK;
}

int sph_slab_create(const sph_params* params, int n_max, int device, int col_lo, int col_hi, int cols_capacity,
                    sph_context** out) {
  return create_common(params, n_max, device, true, col_lo, col_hi, cols_capacity, out);
}

int sph_slab_set_columns(sph_context* c, int col_lo, int col_hi) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->slab) return fail(SPH_ERR_STATE, "not a slab context");
  if (col_hi - col_lo > c->cols_capacity)
    return fail(SPH_ERR_CAPACITY, "%d columns exceed cols_capacity = %d", col_hi - col_lo, c->cols_capacity);
  return set_grid(c, &c->params, col_lo, col_hi);
}

int sph_slab_get_columns(const sph_context* c, int* col_lo, int* col_hi, int* global_cols) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (col_lo) *col_lo = c->col_lo;
  if (col_hi) *col_hi = c->col_hi;
  if (global_cols) *global_cols = c->gn;
  return SPH_OK;
}

int sph_slab_upload(sph_context* c, const float* pos, const float* vel, const int32_t* ids, int n) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (n < 0 || n > c->n_max) return fail(SPH_ERR_CAPACITY, "n = %d outside 0..n_max = %d", n, c->n_max);
  if (n > 0 && (!pos || !vel || !ids)) return fail(SPH_ERR_INVALID, "positions/velocities/ids pointer is NULL");
  HIP_TRY(hipSetDevice(c->device));
  if (n > 0) {
    int32_t* d_ids = (int32_t*)c->stage_dbg;
    HIP_TRY(hipMemcpyAsync(c->stage_pos, pos, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->stage_vel, vel, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_ids, ids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    launch_split_aos_ids(c->stream, c->stage_pos, c->stage_vel, d_ids, n, c->cur);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->n = n;
  c->n_pending = false;
  c->uploaded = true;
  c->stepped = false;
  c->pre_valid = false;
  c->halo_current = false;
  return SPH_OK;
}

int sph_slab_download(sph_context* c, float* pos, float* vel, int32_t* ids, int capacity, int* n_out) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->uploaded) return fail(SPH_ERR_STATE, "download before upload");
  HIP_TRY(hipSetDevice(c->device));
  int rc = resolve_n(c);
  if (rc) return rc;
  const int n = c->n;
  if (n_out) *n_out = n;
  if (n > capacity) return fail(SPH_ERR_CAPACITY, "%d particles do not fit the caller's capacity %d", n, capacity);
  if (n > 0) {
    int32_t* d_ids = (int32_t*)c->stage_dbg;
    launch_join_aos(c->stream, c->cur, n, pos ? c->stage_pos : nullptr, vel ? c->stage_vel : nullptr,
                    ids ? d_ids : nullptr);
    HIP_TRY(hipGetLastError());
    if (pos) HIP_TRY(hipMemcpyAsync(pos, c->stage_pos, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (vel) HIP_TRY(hipMemcpyAsync(vel, c->stage_vel, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (ids) HIP_TRY(hipMemcpyAsync(ids, d_ids, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return SPH_OK;
}

int sph_slab_download_debug(sph_context* c, float* rho, float* force, int32_t* ncount, int32_t* cell, int capacity) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->stepped) return fail(SPH_ERR_STATE, "sph_slab_download_debug needs at least one step after the upload");
  if ((force || ncount) && !(c->params.flags & SPH_FLAG_KEEP_DEBUG))
    return fail(SPH_ERR_STATE, "force/ncount need SPH_FLAG_KEEP_DEBUG set before the step");
  HIP_TRY(hipSetDevice(c->device));
  int rc = resolve_n(c);
  if (rc) return rc;
  const size_t n = (size_t)c->n_last_own;
  if ((int)n > capacity) return fail(SPH_ERR_CAPACITY, "%zu particles do not fit the caller's capacity %d", n, capacity);
  if (n == 0) return SPH_OK;
  float* d_rho = (float*)c->stage_dbg;
  float* d_force = d_rho + n;
  int32_t* d_ncount = (int32_t*)(d_force + 3 * n);
  int32_t* d_cell = d_ncount + n;
  // the force plane stride is the launch size of the step that wrote it: recover it from the step's n
  launch_join_debug(c->stream, c->sorted_dev, c->grid.start, (int)n, c->dbg_stride, c->sorted.rec, c->dbg.force, c->dbg.ncount,
                    c->grid.cell_s, rho ? d_rho : nullptr, force ? d_force : nullptr, ncount ? d_ncount : nullptr,
                    cell ? d_cell : nullptr);
  HIP_TRY(hipGetLastError());
  if (rho) HIP_TRY(hipMemcpyAsync(rho, d_rho, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (force) HIP_TRY(hipMemcpyAsync(force, d_force, 3 * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (ncount) HIP_TRY(hipMemcpyAsync(ncount, d_ncount, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (cell) HIP_TRY(hipMemcpyAsync(cell, d_cell, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return SPH_OK;
}

int sph_slab_pack(sph_context* c, float* d_left, float* d_right, int capacity, int* n_left, int* n_right) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->slab) return fail(SPH_ERR_STATE, "not a slab context");
  if (!c->uploaded) return fail(SPH_ERR_STATE, "sph_slab_pack before upload");
  if (!n_left || !n_right || capacity < 0) return fail(SPH_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  int rc = resolve_n(c);
  if (rc) return rc;
  const bool has_left = c->col_lo > 0 && d_left, has_right = c->col_hi < c->gn && d_right;
  const bool same_request = c->pre_cap == capacity && c->pre_side[0] == (d_left != nullptr) &&
                            c->pre_side[1] == (d_right != nullptr);
  if (c->pre_valid && same_request) {
    // packed behind the last step: the counts came home with the own count, the records are copied out
    c->pre_valid = false;
    *n_left = c->h_counters[0];
    *n_right = c->h_counters[1];
    if (*n_left > capacity || *n_right > capacity)
      return fail(SPH_ERR_CAPACITY, "halo/migration records (%d left, %d right) exceed the buffer capacity %d",
                  *n_left, *n_right, capacity);
    const size_t rec = kRecordBytes;
    if (*n_left > 0)
      HIP_TRY(hipMemcpyAsync(d_left, c->pre_rec[0], (size_t)*n_left * rec, hipMemcpyDeviceToDevice, c->stream));
    if (*n_right > 0)
      HIP_TRY(hipMemcpyAsync(d_right, c->pre_rec[1], (size_t)*n_right * rec, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPH_OK;
  }
  c->pre_valid = false;
  HIP_TRY(hipMemsetAsync(c->d_counters, 0, 2 * sizeof(int32_t), c->stream));
  {
    Timed t(c, CAT_OTHER);
    launch_slab_pack(c->stream, c->dev, c->cur, c->n, nullptr, c->col_lo, c->col_hi, has_left, has_right, d_left,
                     d_right, capacity, c->d_counters);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(c->h_counters, c->d_counters, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  *n_left = c->h_counters[0];
  *n_right = c->h_counters[1];
  if (*n_left > capacity || *n_right > capacity)
    return fail(SPH_ERR_CAPACITY, "halo/migration records (%d left, %d right) exceed the buffer capacity %d", *n_left,
                *n_right, capacity);
  // from now on the records of the next exchange are packed right behind each step
  if (!same_request && capacity > 0) {
    for (auto& p : c->pre_rec) {
      (void)hipFree(p);
      p = nullptr;
    }
    c->pre_cap = 0;
    if (hipMalloc((void**)&c->pre_rec[0], (size_t)capacity * kRecordBytes) == hipSuccess &&
        hipMalloc((void**)&c->pre_rec[1], (size_t)capacity * kRecordBytes) == hipSuccess) {
      c->pre_cap = capacity;
      c->pre_side[0] = d_left != nullptr;
      c->pre_side[1] = d_right != nullptr;
    } else {
      (void)hipGetLastError();  // no room for the shortcut: keep packing on demand
      for (auto& p : c->pre_rec) {
        (void)hipFree(p);
        p = nullptr;
      }
    }
  }
  return SPH_OK;
}

int sph_slab_append(sph_context* c, const float* d_rec, int n_rec) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->slab) return fail(SPH_ERR_STATE, "not a slab context");
  if (n_rec < 0 || (n_rec > 0 && !d_rec)) return fail(SPH_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  int rc = resolve_n(c);
  if (rc) return rc;
  if (c->n + n_rec > c->n_max)
    return fail(SPH_ERR_CAPACITY, "%d own + %d received particles exceed n_max = %d", c->n, n_rec, c->n_max);
  {
    Timed t(c, CAT_OTHER);
    launch_slab_append(c->stream, c->dev, d_rec, n_rec, c->cur, c->n, c->d_counters + 4);
  }
  HIP_TRY(hipGetLastError());
  c->n += n_rec;
  c->uploaded = true;
  if (n_rec > 0) c->pre_valid = false;
  return SPH_OK;
}

// Column counts and (optionally) work per column and the column shift, in one round trip to the device.
static int load_info(sph_context* c, int32_t* counts, int32_t* work, int capacity, int* ncols, float* max_shift) {
  if (!c || !counts) return fail(SPH_ERR_INVALID, "NULL argument");
  if (!c->stepped) return fail(SPH_ERR_STATE, "column counts come from the last step's sort: step first");
  HIP_TRY(hipSetDevice(c->device));
  // the start table belongs to the grid of the last sort, not to what set_columns/set_params may have set since
  const DevParams& G = c->sorted_dev;
  const int own = c->sorted_col_hi - c->sorted_col_lo;
  if (ncols) *ncols = own;
  if (own > capacity) return fail(SPH_ERR_CAPACITY, "%d own columns do not fit the caller's capacity %d", own, capacity);
  if (max_shift) {
    int rc = resolve_n(c);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->d_counters + 5, 0, sizeof(int32_t), c->stream));
    launch_max_column_shift(c->stream, c->dev, c->cur, c->n, c->d_counters + 5);
    HIP_TRY(hipMemcpyAsync(c->h_counters + 5, c->d_counters + 5, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  }
  launch_column_counts(c->stream, c->grid.start, G.nx, G.ny * G.nz, c->d_colcounts);
  int32_t* d_work = c->d_colcounts + c->cols_capacity + 4;
  if (work) launch_column_work(c->stream, c->grid.start, G.own_lo, G.own_hi, G.ny * G.nz, c->nmask.nnb, d_work);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(counts, c->d_colcounts + G.own_lo, (size_t)own * sizeof(int32_t), hipMemcpyDeviceToHost,
                         c->stream));
  if (work)
    HIP_TRY(hipMemcpyAsync(work, d_work + G.own_lo, (size_t)own * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (max_shift) memcpy(max_shift, c->h_counters + 5, sizeof(float));
  return SPH_OK;
}

int sph_slab_column_counts(sph_context* c, int32_t* counts, int capacity, int* ncols) {
  return load_info(c, counts, nullptr, capacity, ncols, nullptr);
}

int sph_slab_load_info(sph_context* c, int32_t* counts, int capacity, int* ncols, float* max_shift) {
  if (!max_shift) return fail(SPH_ERR_INVALID, "NULL argument");
  return load_info(c, counts, nullptr, capacity, ncols, max_shift);
}

int sph_slab_load_work(sph_context* c, int32_t* counts, int32_t* work, int capacity, int* first_col, int* ncols,
                       float* max_shift) {
  if (!max_shift || !work || !first_col) return fail(SPH_ERR_INVALID, "NULL argument");
  const int rc = load_info(c, counts, work, capacity, ncols, max_shift);
  if (rc == SPH_OK) *first_col = c->sorted_col_lo;
  return rc;
}

int sph_slab_lost(sph_context* c, int64_t* lost) {
  if (!c || !lost) return fail(SPH_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  int rc = resolve_n(c);
  if (rc) return rc;
  // records appended since the last step have not been through resolve_n yet
  HIP_TRY(hipMemcpyAsync(c->h_counters + 4, c->d_counters + 4, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->lost = c->h_counters[4];
  *lost = c->lost;
  return SPH_OK;
}

int sph_slab_max_column_shift(sph_context* c, float* columns) {
  if (!c || !columns) return fail(SPH_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  int rc = resolve_n(c);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(c->d_counters + 5, 0, sizeof(int32_t), c->stream));
  launch_max_column_shift(c->stream, c->dev, c->cur, c->n, c->d_counters + 5);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(c->h_counters + 5, c->d_counters + 5, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(columns, c->h_counters + 5, sizeof(float));
  return SPH_OK;
}


// ---- exchange driven by the library: overlapped slab step ---------------------------------------------------------------
// The reference has no counterpart (single process).  A slab context that has been attached to a transport exchanges
// its halo and migration records itself:
//   transport RCCL   ncclSend/ncclRecv with the two neighbouring ranks inside one ncclGroup, on the context's own
//                    communication stream (xGMI: each neighbour pair has its own direct link; no collective);
//   transport LOCAL  the neighbours are contexts of the same process (several slabs on one GPU, tests): a device copy
//                    out of the neighbour's message buffer, ordered by events.
// A message holds an agreed number of records behind a header record whose first word is the sender's true count:
// both ends size the next message from the last true count (the sender knows what it sent, the receiver what it got),
// so sizes are known at both ends without a handshake, and the rare message that outgrows its size sends the
// remainder in a second round.
namespace {

constexpr int kHdr = 1;  // header records in front of a message
constexpr int kOverlapMinParticles = 4000000;  // sph_slab_set_overlap(ctx, 2): smaller slabs do not split their force pass

int next_capacity(const sph_context* c, int count) {
  long long want = (long long)count + (long long)ceil((double)count * (double)c->size_headroom) + 1;
  want = (want + 255) / 256 * 256;
  if (want < c->size_min) want = c->size_min;
  return (int)(want > c->xcap ? c->xcap : want);
}

bool has_face(const sph_context* c, int side, bool next) {
  const int lo = next && c->next_pending ? c->next_lo : c->col_lo, hi = next && c->next_pending ? c->next_hi : c->col_hi;
  return side == 0 ? lo > 0 : hi < c->gn;
}

int attach_common(sph_context* c, int rec_capacity) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->slab) return fail(SPH_ERR_STATE, "not a slab context");
  if (c->xfer != sph_context::XFER_NONE) return fail(SPH_ERR_STATE, "the context is already attached to a transport");
  if (rec_capacity < 1024) return fail(SPH_ERR_INVALID, "rec_capacity must be at least 1024 records");
  HIP_TRY(hipSetDevice(c->device));
  // an earlier attach that failed half-way may have left some of these behind: reuse what exists, replace the buffers
  if (!c->comm_stream) HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  if (!c->ev_pack) HIP_TRY(hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming));
  if (!c->ev_comm) HIP_TRY(hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming));
  const size_t bytes = (size_t)(rec_capacity + kHdr) * kRecordBytes;
  for (int k = 0; k < 2; k++) {
    (void)hipFree(c->msg_out[k]);
    (void)hipFree(c->msg_in[k]);
    c->msg_out[k] = c->msg_in[k] = nullptr;
    HIP_TRY(hipMalloc((void**)&c->msg_out[k], bytes));
    HIP_TRY(hipMalloc((void**)&c->msg_in[k], bytes));
  }
  c->xcap = rec_capacity;
  c->halo_current = false;
  // an event that has never been recorded counts as complete: the first waits on ev_comm pass
  return SPH_OK;
}

// One round of the transport on the communication stream: towards each neighbour `n_out[k]` words of msg_out[k] from
// word `off_out[k]` on, from each neighbour `n_in[k]` words into msg_in[k] at word `off_in[k]`.  Both ends of a face
// pass the same numbers (what one sends the other receives, at the same offset).
//   RCCL   ncclSend/ncclRecv with ranks rank-1 and rank+1, all inside one group;
//   LOCAL  the receiver copies out of the neighbour's msg_out (the neighbour's own call moves nothing towards me).
int transfer_round(sph_context* c, const bool face[2], const size_t off_out[2], const size_t n_out[2],
                   const size_t off_in[2], const size_t n_in[2], const char* what) {
  if (c->xfer == sph_context::XFER_RCCL) {
    char err[256] = "";
    if (rccl_exchange(c->rccl, c->comm_stream, face[0] ? c->rank - 1 : -1, c->msg_out[0] + off_out[0], n_out[0],
                      c->msg_in[0] + off_in[0], n_in[0], face[1] ? c->rank + 1 : -1, c->msg_out[1] + off_out[1], n_out[1],
                      c->msg_in[1] + off_in[1], n_in[1], err, sizeof err))
      return fail(SPH_ERR_HIP, "%s: %s", what, err);
    return SPH_OK;
  }
  for (int k = 0; k < 2; k++) {
    if (!face[k] || n_in[k] == 0) continue;
    HIP_TRY(hipMemcpyAsync(c->msg_in[k] + off_in[k], c->peer[k]->msg_out[1 - k] + off_in[k], n_in[k] * sizeof(float),
                           hipMemcpyDeviceToDevice, c->comm_stream));
  }
  return SPH_OK;
}

// The header word (true record count) of each received message, to pinned host memory behind the round that brought it.
int fetch_headers(sph_context* c, const bool face[2]) {
  for (int k = 0; k < 2; k++)
    if (face[k])
      HIP_TRY(hipMemcpyAsync(c->h_counters + 8 + k, c->msg_in[k], sizeof(int32_t), hipMemcpyDeviceToHost, c->comm_stream));
  return SPH_OK;
}

// Posts the exchange of the messages in msg_out (packed and headed on the compute stream, ev_pack recorded behind
// them) with cap_out[side] records towards and cap_in[side] records from each neighbour, plus the header record.
// RCCL: enqueued on the communication stream now.  LOCAL: the neighbour may not have packed yet - every context copies
// its incoming messages in its complete_exchange, after all contexts of the process have begun.
int post_exchange(sph_context* c) {
  HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
  const bool face[2] = {has_face(c, 0, true), has_face(c, 1, true)};
  for (int k = 0; k < 2; k++) {
    c->posted_out[k] = c->cap_out[k];
    c->posted_in[k] = c->cap_in[k];
  }
  if (c->xfer == sph_context::XFER_RCCL) {
    const size_t w = SPH_SLAB_RECORD_WORDS;
    const size_t zero[2] = {0, 0};
    const size_t n_out[2] = {(size_t)(c->cap_out[0] + kHdr) * w, (size_t)(c->cap_out[1] + kHdr) * w};
    const size_t n_in[2] = {(size_t)(c->cap_in[0] + kHdr) * w, (size_t)(c->cap_in[1] + kHdr) * w};
    int rc = transfer_round(c, face, zero, n_out, zero, n_in, "halo exchange");
    if (rc == SPH_OK) rc = fetch_headers(c, face);
    if (rc != SPH_OK) return rc;
  }
  // own count, trash count, what was packed, stray and missed counters: one small copy behind the exchange
  HIP_TRY(hipMemcpyAsync(c->h_counters, c->d_counters, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, c->comm_stream));
  if (c->xfer == sph_context::XFER_RCCL) HIP_TRY(hipEventRecord(c->ev_comm, c->comm_stream));
  c->in_flight = true;
  c->begun += 1;
  return SPH_OK;
}

// An exchange that cannot be completed leaves the context without a current halo and with nothing in flight: the
// caller may upload and start over, but the step that was under way is lost - on every rank, since the neighbours
// completed their side of it.  Returns `code` for use in a return statement.
int abandon_exchange(sph_context* c, int code) {
  (void)hipStreamSynchronize(c->comm_stream);
  c->in_flight = false;
  c->halo_current = false;
  return code;
}

// Completes the exchange: the ONE host wait of a step (on the communication stream; the compute stream keeps running
// the interior force pass), then the received records join the particle set behind the own particles.
// `own_known` >= 0: the own count is that (halo exchange of an uploaded state); < 0: it comes home with the counters.
// Both transports follow the same size protocol: round one moved posted_out / posted_in records (+ header) whatever
// the true counts were; the header says how many records the sender really had; what did not fit follows in round two,
// whose sizes both ends derive from the same two numbers.
int complete_exchange(sph_context* c, int own_known, bool apply_next) {
  const bool face[2] = {has_face(c, 0, true), has_face(c, 1, true)};
  const size_t w = SPH_SLAB_RECORD_WORDS;
  if (c->xfer == sph_context::XFER_LOCAL) {
    for (int k = 0; k < 2; k++) {
      if (!face[k]) continue;
      sph_context* p = c->peer[k];
      if (!p || p->begun < c->begun)
        return abandon_exchange(c, fail(SPH_ERR_STATE, "the neighbouring context has not begun this exchange"));
      if (p->begun > c->begun)
        return abandon_exchange(c, fail(SPH_ERR_STATE, "the neighbouring context has begun its next exchange and overwritten "
                                                       "the message: call begin on all contexts, then end on all of them"));
      if (p->posted_out[1 - k] != c->posted_in[k])
        return abandon_exchange(c, fail(SPH_ERR_STATE, "message sizes disagree: the neighbour sends %d records, this context "
                                                       "expects %d", p->posted_out[1 - k], c->posted_in[k]));
      // the neighbour's ev_pack was recorded in its begin, which (begin on all, then end on all) came before this call
      HIP_TRY(hipStreamWaitEvent(c->comm_stream, p->ev_pack, 0));
    }
    const size_t zero[2] = {0, 0};
    const size_t n_in1[2] = {(size_t)(c->posted_in[0] + kHdr) * w, (size_t)(c->posted_in[1] + kHdr) * w};
    int rc = transfer_round(c, face, zero, zero, zero, n_in1, "halo exchange");
    if (rc == SPH_OK) rc = fetch_headers(c, face);
    if (rc != SPH_OK) return abandon_exchange(c, rc);
    if (hipStreamSynchronize(c->comm_stream) != hipSuccess)
      return abandon_exchange(c, fail(SPH_ERR_HIP, "waiting for the halo exchange failed"));
  } else {
    if (hipEventSynchronize(c->ev_comm) != hipSuccess)
      return abandon_exchange(c, fail(SPH_ERR_HIP, "waiting for the halo exchange failed"));
  }
  int n_out[2], n_in[2];
  for (int k = 0; k < 2; k++) {
    n_out[k] = face[k] ? c->h_counters[k] : 0;
    n_in[k] = face[k] ? c->h_counters[8 + k] : 0;
  }
  // A face that needs more than rec_capacity records fails the run, on BOTH of its ends (they read the same count).
  // The remainder round is still posted, with the counts clamped the same way at both ends, so that a neighbour on the
  // other, healthy face is not left waiting for a message that never comes.
  int over = 0;  // the largest count beyond the capacity (a negative header cannot come from a sender of this library)
  for (int k = 0; k < 2; k++) {
    if (n_in[k] < 0) n_in[k] = 0x7fffffff;
    const int worst = n_in[k] > n_out[k] ? n_in[k] : n_out[k];
    if (worst > c->xcap && worst > over) over = worst;
    if (n_in[k] > c->xcap) n_in[k] = c->xcap;
    if (n_out[k] > c->xcap) n_out[k] = c->xcap;
  }
  size_t off_out[2], off_in[2], more_out[2], more_in[2];
  bool any = false;
  for (int k = 0; k < 2; k++) {
    off_out[k] = (size_t)(c->posted_out[k] + kHdr) * w;
    off_in[k] = (size_t)(c->posted_in[k] + kHdr) * w;
    more_out[k] = n_out[k] > c->posted_out[k] ? (size_t)(n_out[k] - c->posted_out[k]) * w : 0;
    more_in[k] = n_in[k] > c->posted_in[k] ? (size_t)(n_in[k] - c->posted_in[k]) * w : 0;
    any = any || more_out[k] || more_in[k];
  }
  if (any) {
    const int rc = transfer_round(c, face, off_out, more_out, off_in, more_in, "halo exchange (remainder)");
    if (rc != SPH_OK) return abandon_exchange(c, rc);
    c->remainder_rounds += 1;
  }
  if (any || c->xfer == sph_context::XFER_LOCAL) HIP_TRY(hipEventRecord(c->ev_comm, c->comm_stream));
  for (int k = 0; k < 2; k++) {
    c->cap_out[k] = next_capacity(c, n_out[k]);
    c->cap_in[k] = next_capacity(c, n_in[k]);
  }
  if (over)
    return abandon_exchange(c, fail(SPH_ERR_CAPACITY, "%d halo/migration records exceed the message capacity %d", over, c->xcap));
  const int own = own_known >= 0 ? own_known : c->h_counters[2];
  c->lost = c->h_counters[4];
  c->missed = c->h_counters[6];
  if ((long long)own + n_in[0] + n_in[1] > c->n_max)
    return abandon_exchange(c, fail(SPH_ERR_CAPACITY, "%d own + %d received particles exceed n_max = %d", own,
                                    n_in[0] + n_in[1], c->n_max));
  // The records just exchanged were packed for the faces of the NEXT step: those take effect now, before the received
  // records are checked against the local grid (kernels already launched carry their own copy of the old one).
  if (apply_next && c->next_pending) {
    c->next_pending = false;
    const int rc = set_grid(c, &c->params, c->next_lo, c->next_hi);
    if (rc != SPH_OK) return abandon_exchange(c, rc);
  }
  // the received records join the set on the compute stream, behind whatever it is still doing
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
  c->n = own;
  c->n_last_own = own;
  c->n_pending = false;
  for (int k = 0; k < 2; k++) {
    if (n_in[k] <= 0) continue;
    Timed t(c, CAT_OTHER);
    launch_slab_append(c->stream, c->dev, c->msg_in[k] + kHdr * SPH_SLAB_RECORD_WORDS, n_in[k], c->cur, c->n,
                       c->d_counters + 4);
    c->n += n_in[k];
  }
  HIP_TRY(hipGetLastError());
  c->in_flight = false;
  c->halo_current = true;
  return SPH_OK;
}

// LOCAL transport: my next pack overwrites msg_out, which the neighbours copy from in THEIR complete_exchange.
int wait_for_readers(sph_context* c) {
  if (c->xfer != sph_context::XFER_LOCAL) return SPH_OK;
  for (int k = 0; k < 2; k++)
    if (c->peer[k] && c->peer[k]->ev_comm) HIP_TRY(hipStreamWaitEvent(c->stream, c->peer[k]->ev_comm, 0));
  return SPH_OK;
}

}  // namespace

int sph_comm_unique_id(void* id128) {
  if (!id128) return fail(SPH_ERR_INVALID, "NULL argument");
  char err[256] = "";
  if (rccl_unique_id(id128, err, sizeof err)) return fail(SPH_ERR_HIP, "%s", err);
  return SPH_OK;
}

int sph_slab_attach_rccl(sph_context* c, const void* id128, int rank, int nranks, int rec_capacity) {
  if (!id128 || rank < 0 || rank >= nranks) return fail(SPH_ERR_INVALID, "bad rank %d of %d", rank, nranks);
  int rc = attach_common(c, rec_capacity);
  if (rc) return rc;
  char err[256] = "";
  if (rccl_comm_create(&c->rccl, id128, rank, nranks, err, sizeof err)) return fail(SPH_ERR_HIP, "%s", err);
  c->rank = rank;
  c->nranks = nranks;
  c->xfer = sph_context::XFER_RCCL;
  return SPH_OK;
}

int sph_slab_attach_local(sph_context* c, sph_context* left, sph_context* right, int rec_capacity) {
  int rc = attach_common(c, rec_capacity);
  if (rc) return rc;
  c->peer[0] = left;
  c->peer[1] = right;
  c->xfer = sph_context::XFER_LOCAL;
  return SPH_OK;
}

int sph_slab_set_message_sizing(sph_context* c, float headroom, int min_records) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!(headroom >= 0.0f) || headroom > 16.0f || min_records < 1)
    return fail(SPH_ERR_INVALID, "bad message sizing (headroom %g, minimum %d records)", headroom, min_records);
  if (c->in_flight) return fail(SPH_ERR_STATE, "an exchange is in flight");
  c->size_headroom = headroom;
  c->size_min = min_records;
  return SPH_OK;
}

int sph_slab_exchange_stats(const sph_context* c, uint64_t* exchanges, uint64_t* remainder_rounds, int next_out[2],
                            int next_in[2]) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (exchanges) *exchanges = c->begun;
  if (remainder_rounds) *remainder_rounds = c->remainder_rounds;
  for (int k = 0; k < 2; k++) {
    if (next_out) next_out[k] = c->cap_out[k];
    if (next_in) next_in[k] = c->cap_in[k];
  }
  return SPH_OK;
}

// The transport's own primitive on this device: the group of ncclSend/ncclRecv the slab exchange posts, with this
// rank as both neighbours (RCCL matches a rank's sends to itself with its receives from itself, in order).
int sph_comm_self_exchange(const float* d_send_left, float* d_recv_left, size_t words_left, const float* d_send_right,
                           float* d_recv_right, size_t words_right) {
  if ((words_left && (!d_send_left || !d_recv_left)) || (words_right && (!d_send_right || !d_recv_right)))
    return fail(SPH_ERR_INVALID, "NULL buffer");
  char err[256] = "";
  unsigned char id[kRcclIdBytes];
  if (rccl_unique_id(id, err, sizeof err)) return fail(SPH_ERR_HIP, "%s", err);
  void* comm = nullptr;
  if (rccl_comm_create(&comm, id, 0, 1, err, sizeof err)) return fail(SPH_ERR_HIP, "%s", err);
  hipStream_t stream = nullptr;
  int rc = SPH_OK;
  if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) rc = fail(SPH_ERR_HIP, "hipStreamCreate failed");
  if (rc == SPH_OK && rccl_exchange(comm, stream, 0, d_send_left, words_left, d_recv_left, words_left, 0, d_send_right,
                                    words_right, d_recv_right, words_right, err, sizeof err))
    rc = fail(SPH_ERR_HIP, "self exchange: %s", err);
  if (rc == SPH_OK && hipStreamSynchronize(stream) != hipSuccess) rc = fail(SPH_ERR_HIP, "self exchange did not complete");
  if (stream) (void)hipStreamDestroy(stream);
  rccl_comm_destroy(comm);
  return rc;
}

int sph_slab_set_next_columns(sph_context* c, int col_lo, int col_hi) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->slab) return fail(SPH_ERR_STATE, "not a slab context");
  if (col_lo < 0 || col_hi > c->gn || col_hi - col_lo < SPH_SLAB_MIN_COLUMNS)
    return fail(SPH_ERR_INVALID, "slab columns [%d, %d) invalid", col_lo, col_hi);
  if (col_hi - col_lo > c->cols_capacity)
    return fail(SPH_ERR_CAPACITY, "%d columns exceed cols_capacity = %d", col_hi - col_lo, c->cols_capacity);
  if (abs(col_lo - c->col_lo) > SPH_SLAB_MAX_FACE_MOVE || abs(col_hi - c->col_hi) > SPH_SLAB_MAX_FACE_MOVE)
    return fail(SPH_ERR_INVALID, "a face may move by at most %d columns per step", SPH_SLAB_MAX_FACE_MOVE);
  if (c->in_flight)
    return fail(SPH_ERR_STATE, "an exchange is in flight: its records were packed for the faces as they stood at sph_slab_step_begin");
  c->next_lo = col_lo;
  c->next_hi = col_hi;
  c->next_pending = col_lo != c->col_lo || col_hi != c->col_hi;
  return SPH_OK;
}

int sph_slab_set_overlap(sph_context* c, int mode) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (mode < 0 || mode > 2) return fail(SPH_ERR_INVALID, "overlap mode %d is not 0, 1 or 2", mode);
  c->overlap_mode = mode;
  return SPH_OK;
}

int sph_slab_set_face_shift(sph_context* c, float columns) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!(columns >= 0.0f) || columns > 1024.0f) return fail(SPH_ERR_INVALID, "bad column shift %g", columns);
  c->face_shift = columns;
  return SPH_OK;
}

int sph_slab_missed(sph_context* c, int64_t* missed) {
  if (!c || !missed) return fail(SPH_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(c->h_counters + 6, c->d_counters + 6, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->missed = c->h_counters[6];
  *missed = c->missed;
  return SPH_OK;
}

// Halo exchange of the state as it stands (after an upload): every particle near a face is packed.
int sph_slab_halo_begin(sph_context* c) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (c->xfer == sph_context::XFER_NONE) return fail(SPH_ERR_STATE, "attach a transport first");
  if (!c->uploaded) return fail(SPH_ERR_STATE, "halo exchange before upload");
  if (c->in_flight) return fail(SPH_ERR_STATE, "an exchange is already in flight");
  HIP_TRY(hipSetDevice(c->device));
  int rc = resolve_n(c);
  if (rc) return rc;
  if ((rc = wait_for_readers(c))) return rc;
  c->next_pending = false;
  const bool l = has_face(c, 0, false), r = has_face(c, 1, false);
  HIP_TRY(hipMemsetAsync(c->d_counters, 0, 2 * sizeof(int32_t), c->stream));
  {
    Timed t(c, CAT_OTHER);
    launch_slab_pack(c->stream, c->dev, c->cur, c->n, nullptr, c->col_lo, c->col_hi, l, r,
                     c->msg_out[0] + kHdr * SPH_SLAB_RECORD_WORDS, c->msg_out[1] + kHdr * SPH_SLAB_RECORD_WORDS, c->xcap,
                     c->d_counters);
    launch_slab_headers(c->stream, c->d_counters, l ? c->msg_out[0] : nullptr, r ? c->msg_out[1] : nullptr);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev_pack, c->stream));
  // first message of a run: full-size both ways (every rank uses the same rec_capacity)
  c->cap_out[0] = c->cap_out[1] = c->cap_in[0] = c->cap_in[1] = c->xcap;
  return post_exchange(c);
}

int sph_slab_halo_end(sph_context* c) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->in_flight) return fail(SPH_ERR_STATE, "no exchange in flight");
  HIP_TRY(hipSetDevice(c->device));
  return complete_exchange(c, c->n, false);
}

// One time step, records for the next one packed and under way before the interior columns are advanced.
int sph_slab_step_begin(sph_context* c) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (c->xfer == sph_context::XFER_NONE) return fail(SPH_ERR_STATE, "attach a transport first");
  if (!c->halo_current) return fail(SPH_ERR_STATE, "the halo is not current: sph_slab_halo_begin/end first");
  if (c->in_flight) return fail(SPH_ERR_STATE, "an exchange is already in flight");
  if (c->params.flags & (SPH_FLAG_EXACT_ORDER | SPH_FLAG_LANE_SWEEP))
    return fail(SPH_ERR_STATE, "the overlapped step runs the masked sweeps only");
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  const bool keep = (c->params.flags & SPH_FLAG_KEEP_DEBUG) != 0;
  const SweepMode mode = (c->params.flags & SPH_FLAG_REFERENCE_ARITH) ? SWEEP_MASKED : SWEEP_FAST;
  const int n = c->n;
  { Timed t(c, CAT_HASH); launch_hash(c->stream, c->dev, c->cur, n, c->grid); }
  { Timed t(c, CAT_SCAN); launch_scan(c->stream, c->grid); }
  { Timed t(c, CAT_REORDER); launch_reorder(c->stream, c->cur, c->sorted, n, c->grid); }
  { Timed t(c, CAT_SEARCH);
    launch_neighbours(c->stream, c->dev, c->sorted, n, c->grid, c->nmask, mode == SWEEP_FAST, keep ? c->dbg.ncount : nullptr); }
  if (mode != SWEEP_FAST) {
    Timed t(c, CAT_DENSITY);
    launch_density(c->stream, c->dev, c->sorted, n, c->grid, keep ? c->dbg.ncount : nullptr, c->nmask, mode);
  }
  // face ranges (local columns): wide enough that whatever can reach the neighbour's halo within this step, and the
  // column a face may move by, starts inside them
  const bool l = has_face(c, 0, true), r = has_face(c, 1, true);
  // ... plus the columns a face is about to move by: what the neighbour gains must have its new state first as well
  const int move_lo = c->next_pending ? abs(c->next_lo - c->col_lo) : 0, move_hi = c->next_pending ? abs(c->next_hi - c->col_hi) : 0;
  const int width = 3 + (int)ceilf(c->face_shift) + (move_lo > move_hi ? move_lo : move_hi);
  const DevParams& G = c->dev;
  const int a_lo = G.own_lo, a_hi = l ? (G.own_lo + width < G.own_hi ? G.own_lo + width : G.own_hi) : G.own_lo;
  const int b_hi = G.own_hi, b_lo = r ? (G.own_hi - width > a_hi ? G.own_hi - width : a_hi) : G.own_hi;
  const int send_lo = c->next_pending ? c->next_lo : c->col_lo, send_hi = c->next_pending ? c->next_hi : c->col_hi;
  int32_t* own_out = c->d_counters + 2;
  auto force_range = [&](int lo, int hi) {
    if (hi <= lo && own_out == nullptr) return;
    DevParams Q = c->dev;
    Q.force_lo = lo;
    Q.force_hi = hi;
    Timed t(c, CAT_FORCE);
    launch_force(c->stream, Q, c->sorted, c->cur, n, c->grid, keep ? c->dbg.force : nullptr, own_out, c->nmask, mode);
    own_out = nullptr;  // the first launch reports the own and trash counts
  };
  if ((rc = wait_for_readers(c))) return rc;
  // an empty slab launches nothing that would write the own and trash counts
  if (n == 0) HIP_TRY(hipMemsetAsync(c->d_counters, 0, 4 * sizeof(int32_t), c->stream));
  // Faces first and the interior beside the exchange pays only for slabs of several million particles: three launches
  // of a few hundred thousand particles each leave the chip half empty in their tails (measured on slabs of a million:
  // 0.30 ms of force pass split in three against 0.23 ms in one launch), which is more than a neighbour exchange of
  // that size takes.  Smaller slabs advance all own columns in one launch and pack afterwards.
  const bool split = c->overlap_mode == 1 || (c->overlap_mode == 2 && n >= kOverlapMinParticles);
  if (!split || (b_lo == a_hi && b_hi > b_lo && a_hi > a_lo)) {
    force_range(G.own_lo, G.own_hi);
  } else {
    force_range(a_lo, a_hi);
    if (b_hi > b_lo) force_range(b_lo, b_hi);
  }
  HIP_TRY(hipMemsetAsync(c->d_counters, 0, 2 * sizeof(int32_t), c->stream));
  {
    Timed t(c, CAT_OTHER);
    launch_slab_pack_faces(c->stream, c->dev, c->cur, n, c->grid.start, a_lo, a_hi, b_lo, b_hi, send_lo, send_hi, l, r,
                           c->msg_out[0], c->msg_out[1], c->xcap, c->d_counters);
    launch_slab_headers(c->stream, c->d_counters, l ? c->msg_out[0] : nullptr, r ? c->msg_out[1] : nullptr);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev_pack, c->stream));
  if ((rc = post_exchange(c))) return rc;
  // the interior columns, while the records travel
  if (b_lo > a_hi) {
    if (split) force_range(a_hi, b_lo);
    launch_slab_check_interior(c->stream, c->dev, c->cur, n, c->grid.start, a_hi, b_lo, send_lo, send_hi, l, r,
                               c->d_counters + 6);
  }
  HIP_TRY(hipGetLastError());
  c->dbg_stride = n;
  c->sorted_dev = c->dev;
  c->sorted_col_lo = c->col_lo;
  c->sorted_col_hi = c->col_hi;
  c->pre_valid = false;
  c->halo_current = false;
  c->stepped = true;
  c->acc_steps += 1;
  c->acc_particles += (uint64_t)n;
  if (c->pending.size() > 8192) return drain_timings(c);
  return SPH_OK;
}

int sph_slab_step_end(sph_context* c) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->in_flight) return fail(SPH_ERR_STATE, "no exchange in flight");
  HIP_TRY(hipSetDevice(c->device));
  return complete_exchange(c, -1, true);
}

int sph_slab_exchange_step(sph_context* c) {
  int rc = sph_slab_step_begin(c);
  return rc ? rc : sph_slab_step_end(c);
}

int sph_slab_halo_exchange(sph_context* c) {
  int rc = sph_slab_halo_begin(c);
  return rc ? rc : sph_slab_halo_end(c);
}

int sph_get_stream(sph_context* c, void** stream) {
  if (!c || !stream) return fail(SPH_ERR_INVALID, "NULL argument");
  *stream = (void*)c->stream;
  return SPH_OK;
}

int sph_stream_wait(sph_context* c, void* producer_stream) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  HIP_TRY(hipSetDevice(c->device));
  if ((hipStream_t)producer_stream == c->stream) return SPH_OK;  // same stream: already ordered
  if (!c->wait_event) HIP_TRY(hipEventCreateWithFlags(&c->wait_event, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(c->wait_event, (hipStream_t)producer_stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->wait_event, 0));
  return SPH_OK;
}

int sph_set_stream(sph_context* c, void* stream) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  c->stream = (hipStream_t)stream;
  c->own_stream = false;
  return SPH_OK;
}

// ---- one-call form of updateParticles on a process-wide default context ------------------------------------
static std::mutex g_default_mu;
static sph_context* g_default = nullptr;
static int g_one_call_mode = 0;                 // sph_one_call_mode
static const float* g_last_pos = nullptr;       // host arrays of the last successful call
static const float* g_last_vel = nullptr;
static int g_last_n = -1;

int sph_one_call_mode(int mode) {
  std::lock_guard<std::mutex> lock(g_default_mu);
  const int before = g_one_call_mode;
  if (mode == 0 || mode == 1) {
    g_one_call_mode = mode;
    g_last_n = -1;  // the next call uploads whatever the mode
  }
  return before;
}

int sph_update_particles(const sph_params* params, float* particles, float* velocities, int num_particles) {
  std::lock_guard<std::mutex> lock(g_default_mu);
  sph_params p;
  if (params) p = *params; else sph_params_default(&p);
  int rc = check_params(&p);
  if (rc) return rc;
  if (num_particles < 0) return fail(SPH_ERR_INVALID, "num_particles < 0");
  if (num_particles == 0) return SPH_OK;
  bool need_new = !g_default || g_default->n_max < num_particles;
  if (!need_new && sph_set_params(g_default, &p) != SPH_OK) need_new = true;  // grid outgrew the context
  if (need_new) {
    if (g_default) sph_destroy(g_default);
    g_default = nullptr;
    g_last_n = -1;
    int dev = 0;
    if (const char* e = getenv("SPH_AMD_DEVICE")) dev = atoi(e);
    if ((rc = sph_create(&p, num_particles, dev, &g_default))) return rc;
  }
  // mode 1: the caller promises not to touch the arrays between calls (the reference's main loop does not,
  // main.cpp:31-35), so the state in HBM is what the host arrays hold and the upload can be skipped
  const bool resident = g_one_call_mode == 1 && g_last_pos == particles && g_last_vel == velocities &&
                        g_last_n == num_particles;
  g_last_n = -1;
  if (!resident && (rc = sph_upload(g_default, particles, velocities, num_particles))) return rc;
  if ((rc = sph_step(g_default, 1))) return rc;
  if ((rc = sph_download(g_default, particles, velocities))) return rc;
  g_last_pos = particles;
  g_last_vel = velocities;
  g_last_n = num_particles;
  return SPH_OK;
}

// Page-locks a host range the caller owns so that uploads and downloads from it are direct DMA transfers instead of
// copies through the runtime's staging buffers.  The pages stay locked until sph_host_unregister (or process exit).
namespace {
std::mutex g_pin_mu;
std::vector<std::pair<void*, void*>> g_pinned;  // (caller's pointer, base the runtime was given)
}  // namespace

int sph_host_register(void* p, size_t bytes) {
  if (!p || bytes == 0) return fail(SPH_ERR_INVALID, "bad host range");
  std::lock_guard<std::mutex> lock(g_pin_mu);
  for (auto& e : g_pinned)
    if (e.first == p) return SPH_OK;
  // whole pages: a malloc'd block rarely starts on a page boundary, and the first and last page may hold other
  // heap data, which being page-locked does not harm
  const uintptr_t page = 4096, a = (uintptr_t)p & ~(page - 1), b = ((uintptr_t)p + bytes + page - 1) & ~(page - 1);
  hipError_t e = hipHostRegister((void*)a, b - a, hipHostRegisterDefault);
  if (e == hipErrorHostMemoryAlreadyRegistered) {
    (void)hipGetLastError();
    return SPH_OK;
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(SPH_ERR_HIP, "hipHostRegister(%zu bytes): %s", (size_t)(b - a), hipGetErrorString(e));
  }
  g_pinned.emplace_back(p, (void*)a);
  return SPH_OK;
}

int sph_host_unregister(void* p) {
  if (!p) return fail(SPH_ERR_INVALID, "NULL pointer");
  std::lock_guard<std::mutex> lock(g_pin_mu);
  for (size_t i = 0; i < g_pinned.size(); i++) {
    if (g_pinned[i].first != p) continue;
    const hipError_t e = hipHostUnregister(g_pinned[i].second);
    g_pinned.erase(g_pinned.begin() + i);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(SPH_ERR_HIP, "hipHostUnregister: %s", hipGetErrorString(e));
    }
    return SPH_OK;
  }
  return fail(SPH_ERR_INVALID, "this pointer was not registered through sph_host_register");
}

// ---- one-call step that allocates nothing -------------------------------------------------------------------
// Same step as sph_update_particles, for callers that cannot allocate device memory (HIP graph capture, tight
// serving loops, harnesses that forbid hipMalloc).  Every intermediate buffer of the step - SoA state, grid-sorted
// copies, neighbour masks, cell tables - is a piece of ONE file-scope device array, `scratch`, sized for
// SPH_NOALLOC_MAX_PARTICLES particles and SPH_NOALLOC_MAX_CELLS cells.  In host code the array's name is only a
// handle: its device address on each GPU comes from hipGetSymbolAddress, once per device, and that pointer is what
// kernels and memsets are given.
constexpr size_t kNoAllocCells = (size_t)SPH_NOALLOC_MAX_CELLS + 1;  // + trash cell
constexpr size_t kNoAllocPerParticle =
    7 * 4 + sizeof(float4) + sizeof(ParticleRec) + 4 + (size_t)(kMaskWords + 2) * 4 + 2 * 4 + 5 * 4;
constexpr size_t kNoAllocPerCell = 3 * sizeof(int32_t);
constexpr size_t kScratchBytes = kNoAllocPerParticle * SPH_NOALLOC_MAX_PARTICLES +
                                 kNoAllocPerCell * (kNoAllocCells + 2 * kScanItems) + 64 * 1024;  // + alignment slack
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
struct NoAllocSlot {
  bool ready = false;
  void* base = nullptr;  // device address of `scratch` on this GPU
  sph_context ctx;
};
NoAllocSlot g_noalloc[kMaxDevices];
std::mutex g_noalloc_mu;
}  // namespace

int sph_update_particles_noalloc(const sph_params* params, float* d_particles, float* d_velocities,
                                 int num_particles, void* stream) {
  sph_params p;
  if (params) p = *params; else sph_params_default(&p);
  int rc = check_params(&p);
  if (rc) return rc;
  if (num_particles < 0) return fail(SPH_ERR_INVALID, "num_particles < 0");
  if (num_particles > SPH_NOALLOC_MAX_PARTICLES)
    return fail(SPH_ERR_CAPACITY, "%d particles exceed the %d this entry point's fixed scratch holds", num_particles,
                SPH_NOALLOC_MAX_PARTICLES);
  if (p.flags & SPH_FLAG_KEEP_DEBUG)
    return fail(SPH_ERR_INVALID, "SPH_FLAG_KEEP_DEBUG needs buffers this entry point does not have");
  const int gn = cells_per_axis(&p);
  if (gn < 0 || (size_t)gn * gn * gn > (size_t)SPH_NOALLOC_MAX_CELLS)
    return fail(SPH_ERR_CAPACITY, "a tank of %g with cells of %g needs more than the %d cells this entry point's fixed scratch holds",
                p.tank_size, resolved_cell(&p), SPH_NOALLOC_MAX_CELLS);
  if (num_particles == 0) return SPH_OK;
  if (!d_particles || !d_velocities) return fail(SPH_ERR_INVALID, "particles/velocities pointer is NULL");

  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(SPH_ERR_NO_DEVICE, "device %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_noalloc_mu);
  NoAllocSlot& slot = g_noalloc[dev];
  sph_context* c = &slot.ctx;
  if (!slot.ready) {
    // once per device, with that device current: where this GPU keeps its copy of the array
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(SPH_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 only", dev, prop.gcnArchName);
    void* base = nullptr;
    HIP_TRY(hipGetSymbolAddress(&base, HIP_SYMBOL(scratch)));
    if (!base) return fail(SPH_ERR_HIP, "hipGetSymbolAddress returned a null address");
    RegionSource region(base, kScratchBytes);
    if ((rc = layout_step_buffers(region, c, (size_t)SPH_NOALLOC_MAX_PARTICLES, kNoAllocCells))) return rc;
    c->device = dev;
    c->n_max = SPH_NOALLOC_MAX_PARTICLES;
    c->cell_capacity = kNoAllocCells;
    c->own_stream = false;
    slot.base = base;
    slot.ready = true;
  }
  c->stream = (hipStream_t)stream;
  if ((rc = set_grid(c, &p, 0, 0))) return rc;
  // The array keeps the last call's contents.  The one table a step expects zeroed is the cell histogram (a step
  // leaves it zeroed behind itself, but a call that failed half-way or a different grid would not have).
  HIP_TRY(hipMemsetAsync(c->grid.count, 0, padded_cells(c->grid.ncells) * sizeof(int32_t), c->stream));
  launch_split_aos(c->stream, d_particles, d_velocities, num_particles, c->cur);
  c->n = num_particles;
  c->n_pending = false;
  c->uploaded = true;
  if ((rc = one_step(c))) return rc;
  launch_gather_aos(c->stream, c->cur, num_particles, d_particles, d_velocities);
  HIP_TRY(hipGetLastError());
  return SPH_OK;
}

int sph_fill_lattice(sph_context* c, int nx, int ny, int nz, int x_lo, int x_hi, double spacing, double jitter,
                     uint64_t seed, const double origin[3]) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (nx < 1 || ny < 1 || nz < 1 || x_lo < 0 || x_hi > nx || x_hi < x_lo || !(spacing > 0.0) || !(jitter >= 0.0))
    return fail(SPH_ERR_INVALID, "bad lattice %d x %d x %d, planes [%d, %d)", nx, ny, nz, x_lo, x_hi);
  if ((long long)nx * ny * nz > 0x7fffffffLL) return fail(SPH_ERR_CAPACITY, "particle ids are 32-bit");
  if (!c->slab && (x_lo != 0 || x_hi != nx))
    return fail(SPH_ERR_INVALID, "a plain context numbers its particles from 0: it takes the whole lattice");
  const long long n = (long long)(x_hi - x_lo) * ny * nz;
  if (n > c->n_max) return fail(SPH_ERR_CAPACITY, "%lld particles exceed n_max = %d", n, c->n_max);
  HIP_TRY(hipSetDevice(c->device));
  const double zero[3] = {0.0, 0.0, 0.0};
  launch_fill_lattice(c->stream, c->cur, (long long)x_lo * ny * nz, (int)n, ny, nz, spacing, jitter, seed,
                      origin ? origin : zero);
  HIP_TRY(hipGetLastError());
  c->n = (int)n;
  c->n_pending = false;
  c->uploaded = true;
  c->stepped = false;
  c->pre_valid = false;
  c->halo_current = false;
  return SPH_OK;
}

int sph_fill_random_block(const sph_params* params, uint64_t seed, float* pos, float* vel, int n) {
  if (!params || !pos || !vel || n < 0) return fail(SPH_ERR_INVALID, "bad argument");
  // 64-bit LCG (Knuth MMIX constants); the top 24 bits give a uniform fp32 in [0,1].
  uint64_t s = seed * 2862933555777941757ull + 3037000493ull;
  const float half = params->tank_size / 2.0f;  // compute.cpp:110-112
  for (int i = 0; i < 3 * n; i++) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    float u = (float)(s >> 40) / 16777215.0f;
    pos[i] = u * half;
    vel[i] = 0.0f;  // compute.cpp:114-116
  }
  return SPH_OK;
}

}  // extern "C"
