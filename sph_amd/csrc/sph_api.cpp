// C ABI of libsph_amd.so (declared in include/sph_amd.h): context management, HBM buffers, step driver.
// Host-only code; every compute call ends in a kernel launch from sph_kernels.hip — there is no CPU path.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "sph_internal.hpp"

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

enum Cat { CAT_HASH, CAT_SCAN, CAT_REORDER, CAT_DENSITY, CAT_FORCE, CAT_OTHER, CAT_N };

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
  State cur{}, sorted{};  // cur: state the next step starts from; sorted: this step's grid-ordered copy
  GridBuffers grid{};
  size_t cell_capacity = 0;
  float* rho = nullptr;
  DebugBuffers dbg{};
  float *stage_pos = nullptr, *stage_vel = nullptr;  // AoS staging in HBM, 3*n_max floats each
  void* stage_dbg = nullptr;                         // 6*n_max words for debug downloads
  bool uploaded = false, stepped = false;
  bool timing = false;
  std::vector<EventPair> pending;
  std::vector<hipEvent_t> pool;
  double acc_ms[CAT_N] = {0, 0, 0, 0, 0, 0};
  uint64_t acc_steps = 0, acc_particles = 0;
};

namespace {

// RAII bracket: records an event before and after the launches issued in its scope.
struct Timed {
  sph_context* c;
  EventPair p{};
  bool on;
  Timed(sph_context* ctx, int cat) : c(ctx), on(ctx->timing) {
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
  d->cx_off = cx_off;
  d->gnx = gn;
  d->tension = (p->flags & SPH_FLAG_SURFACE_TENSION) ? 1 : 0;
}

template <class T>
int dev_alloc(T** p, size_t count) {
  *p = nullptr;
  HIP_TRY(hipMalloc((void**)p, (count ? count : 1) * sizeof(T)));
  return SPH_OK;
}

int alloc_state(State* s, size_t n) {
  int rc;
  if ((rc = dev_alloc(&s->x, n))) return rc;
  if ((rc = dev_alloc(&s->y, n))) return rc;
  if ((rc = dev_alloc(&s->z, n))) return rc;
  if ((rc = dev_alloc(&s->vx, n))) return rc;
  if ((rc = dev_alloc(&s->vy, n))) return rc;
  if ((rc = dev_alloc(&s->vz, n))) return rc;
  if ((rc = dev_alloc(&s->id, n))) return rc;
  return SPH_OK;
}

void free_state(State* s) {
  (void)hipFree(s->x); (void)hipFree(s->y); (void)hipFree(s->z);
  (void)hipFree(s->vx); (void)hipFree(s->vy); (void)hipFree(s->vz);
  (void)hipFree(s->id);
  *s = State{};
}

int set_grid(sph_context* c, const sph_params* p) {
  int gn = cells_per_axis(p);
  if (gn < 0) return fail(SPH_ERR_CAPACITY, "tank_size/cell_size needs more than 2048 cells per axis");
  size_t ncells = (size_t)gn * gn * gn;
  if (ncells > c->cell_capacity)
    return fail(SPH_ERR_CAPACITY, "grid of %d^3 cells exceeds the context's capacity of %zu cells", gn,
                c->cell_capacity);
  derive(p, gn, 0, gn, &c->dev);
  c->grid.ncells = ncells;
  c->grid.scan_blocks = (ncells + kScanItems - 1) / kScanItems;
  c->params = *p;
  return SPH_OK;
}

int one_step(sph_context* c) {
  const bool exact = (c->params.flags & SPH_FLAG_EXACT_ORDER) != 0;
  const bool keep = (c->params.flags & SPH_FLAG_KEEP_DEBUG) != 0;
  const int n = c->n;
  { Timed t(c, CAT_HASH); launch_hash(c->stream, c->dev, c->cur, n, c->grid); }
  { Timed t(c, CAT_SCAN); launch_scan(c->stream, c->grid); }
  { Timed t(c, CAT_REORDER); launch_reorder(c->stream, c->cur, c->sorted, n, c->grid); }
  { Timed t(c, CAT_DENSITY);
    launch_density(c->stream, c->dev, c->sorted, n, c->grid, c->rho, keep ? c->dbg.ncount : nullptr, exact); }
  { Timed t(c, CAT_FORCE);
    launch_force(c->stream, c->dev, c->sorted, c->cur, n, c->grid, c->rho, keep ? c->dbg.force : nullptr, exact); }
  // cur now holds the new state in this step's sorted order; its ids are the sorted ids.
  int32_t* t = c->cur.id;
  c->cur.id = c->sorted.id;
  c->sorted.id = t;
  HIP_TRY(hipGetLastError());
  c->stepped = true;
  c->acc_steps += 1;
  c->acc_particles += (uint64_t)n;
  if (c->pending.size() > 8192) return drain_timings(c);
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
  if (!out) return fail(SPH_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int rc = check_params(params);
  if (rc) return rc;
  if (n_max < 1) return fail(SPH_ERR_INVALID, "n_max must be >= 1 (got %d)", n_max);
  int ndev = sph_device_count();
  if (ndev <= 0) return fail(SPH_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU path");
  if (device < 0 || device >= ndev) return fail(SPH_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, ndev - 1);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(SPH_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
  HIP_TRY(hipSetDevice(device));

  int gn = cells_per_axis(params);
  if (gn < 0) return fail(SPH_ERR_CAPACITY, "tank_size/cell_size needs more than 2048 cells per axis");
  size_t ncells = (size_t)gn * gn * gn;
  if (ncells + 1 > (size_t)0x7fffffff) return fail(SPH_ERR_CAPACITY, "%zu cells overflow the 32-bit cell index", ncells);

  sph_context* c = new sph_context();
  c->device = device;
  c->n_max = n_max;
  c->cell_capacity = ncells;
  auto bail = [&](int code) { sph_destroy(c); return code; };
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(SPH_ERR_HIP, "hipStreamCreate failed"));
  const size_t n = (size_t)n_max;
  if ((rc = alloc_state(&c->cur, n)) || (rc = alloc_state(&c->sorted, n))) return bail(rc);
  // scan kernels read/write whole 4096-cell tiles through int4: round the tables up to a tile
  size_t padded = ((ncells + kScanItems - 1) / kScanItems) * kScanItems;
  if ((rc = dev_alloc(&c->grid.count, padded)) || (rc = dev_alloc(&c->grid.start, padded + 4)) ||
      (rc = dev_alloc(&c->grid.block_sums, padded / kScanItems + 1)) || (rc = dev_alloc(&c->grid.cell_of, n)) ||
      (rc = dev_alloc(&c->grid.rank_of, n)) || (rc = dev_alloc(&c->grid.src_of, n)) ||
      (rc = dev_alloc(&c->grid.id_prov, n)) || (rc = dev_alloc(&c->grid.cell_s, n)) ||
      (rc = dev_alloc(&c->rho, n)) || (rc = dev_alloc(&c->dbg.force, 3 * n)) ||
      (rc = dev_alloc(&c->dbg.ncount, n)) || (rc = dev_alloc(&c->stage_pos, 3 * n)) ||
      (rc = dev_alloc(&c->stage_vel, 3 * n)) || (rc = dev_alloc((uint32_t**)&c->stage_dbg, 6 * n)))
    return bail(rc);
  if (hipMemsetAsync(c->grid.count, 0, padded * sizeof(int32_t), c->stream) != hipSuccess)
    return bail(fail(SPH_ERR_HIP, "hipMemsetAsync failed"));
  if ((rc = set_grid(c, params))) return bail(rc);
  if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(fail(SPH_ERR_HIP, "stream sync failed"));
  *out = c;
  return SPH_OK;
}

int sph_destroy(sph_context* c) {
  if (!c) return SPH_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : c->pool) (void)hipEventDestroy(e);
  free_state(&c->cur);
  free_state(&c->sorted);
  (void)hipFree(c->grid.count); (void)hipFree(c->grid.start); (void)hipFree(c->grid.block_sums);
  (void)hipFree(c->grid.cell_of); (void)hipFree(c->grid.rank_of); (void)hipFree(c->grid.src_of);
  (void)hipFree(c->grid.id_prov); (void)hipFree(c->grid.cell_s);
  (void)hipFree(c->rho); (void)hipFree(c->dbg.force); (void)hipFree(c->dbg.ncount);
  (void)hipFree(c->stage_pos); (void)hipFree(c->stage_vel); (void)hipFree(c->stage_dbg);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return SPH_OK;
}

int sph_set_params(sph_context* c, const sph_params* p) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  int rc = check_params(p);
  if (rc) return rc;
  return set_grid(c, p);
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
  c->uploaded = true;
  c->stepped = false;
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

int sph_sync(sph_context* c) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return SPH_OK;
}

static int download_common(sph_context* c, float* pos, float* vel, hipMemcpyKind kind) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  if (!c->uploaded) return fail(SPH_ERR_STATE, "download before upload");
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
  HIP_TRY(hipSetDevice(c->device));
  const size_t n = (size_t)c->n;
  if (n == 0) return SPH_OK;
  float* d_rho = (float*)c->stage_dbg;
  float* d_force = d_rho + n;
  int32_t* d_ncount = (int32_t*)(d_force + 3 * n);
  int32_t* d_cell = d_ncount + n;
  launch_gather_debug(c->stream, c->cur.id, (int)n, c->rho, c->dbg.force, c->dbg.ncount, c->grid.cell_s,
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

int sph_num_particles(const sph_context* c) { return c ? c->n : 0; }

int sph_grid_dims(const sph_context* c, int dims[3]) {
  if (!c || !dims) return fail(SPH_ERR_INVALID, "NULL argument");
  dims[0] = c->dev.nx;
  dims[1] = c->dev.ny;
  dims[2] = c->dev.nz;
  return SPH_OK;
}

int sph_enable_timing(sph_context* c, int enable) {
  if (!c) return fail(SPH_ERR_INVALID, "context is NULL");
  c->timing = enable != 0;
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

// ---- one-call form of updateParticles on a process-wide default context ------------------------------------
static std::mutex g_default_mu;
static sph_context* g_default = nullptr;

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
    int dev = 0;
    if (const char* e = getenv("SPH_AMD_DEVICE")) dev = atoi(e);
    if ((rc = sph_create(&p, num_particles, dev, &g_default))) return rc;
  }
  if ((rc = sph_upload(g_default, particles, velocities, num_particles))) return rc;
  if ((rc = sph_step(g_default, 1))) return rc;
  return sph_download(g_default, particles, velocities);
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
