/*
 * sph_amd.h — C ABI of the MI355X (gfx950) SPH compute library, libsph_amd.so.
 *
 * This library replaces the compute/ subsystem of grzegorz-k-karch/sph:
 *   initParticles / updateParticles / vecNormalize   (reference compute/compute.h:4-8)
 *   the mutable parameter globals                    (reference compute/compute.cpp:8-20)
 *
 * All entry points are extern "C", take plain pointers and sizes, and return an int status
 * (SPH_OK == 0).  On failure sph_last_error() returns a thread-local message.  Nothing here
 * runs on the CPU: every compute entry point launches hand-written HIP kernels and fails with
 * SPH_ERR_NO_DEVICE if no gfx950 device is usable.
 *
 * Particle arrays handed across this boundary use the reference's layout: array-of-structures
 * xyz, float[3*n] for positions and float[3*n] for velocities (reference main.cpp:9-10,
 * compute/compute.cpp:104-105).  Inside the library the state is structure-of-arrays fp32 in
 * HBM, kept in grid-sorted order between steps, with a particle-id array so that downloads
 * come back in caller order.
 */
#ifndef SPH_AMD_H
#define SPH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------ */
enum {
  SPH_OK = 0,
  SPH_ERR_INVALID = 1,    /* bad argument (null pointer, n > n_max, non-finite parameter ...)    */
  SPH_ERR_NO_DEVICE = 2,  /* no usable HIP device / wrong architecture                           */
  SPH_ERR_HIP = 3,        /* a HIP runtime call or kernel launch failed                          */
  SPH_ERR_CAPACITY = 4,   /* particle or cell capacity of the context exceeded                   */
  SPH_ERR_STATE = 5       /* call order violated (e.g. step before upload)                       */
};

/* ---- flags for sph_params.flags ------------------------------------------------------------ */
enum {
  /* Surface-tension force (reference compute.cpp:225-262).  ON by default: the reference
   * computes it although its README lists it as unused. */
  SPH_FLAG_SURFACE_TENSION = 1u << 0,
  /* Neighbour contributions are summed in ascending particle-id order, which is the order in
   * which the reference's j = 0..N-1 loops meet them (non-neighbour terms there are exactly
   * +-0).  Results are then bit-identical to the reference.  Costs O(neighbours x candidates)
   * per particle instead of O(candidates); meant for verification, not for throughput. */
  SPH_FLAG_EXACT_ORDER = 1u << 1,
  /* Keep per-particle total force density and neighbour counts of the last step for
   * sph_download_debug (costs 16 B/particle of extra stores per step). */
  SPH_FLAG_KEEP_DEBUG = 1u << 2,
  /* Ablation switch for the density and force passes: plain one-lane-per-particle sweep, distance test and pair
   * arithmetic in one loop, no candidate mask, reference arithmetic; same results bit for bit as
   * SPH_FLAG_REFERENCE_ARITH. */
  SPH_FLAG_LANE_SWEEP = 1u << 3,
  /* Evaluate every pair term the way the reference's x86-64 object does: kernel coefficient times polynomial
   * formed in double and rounded once (compute.cpp:45,56,68,80,92), correctly rounded sqrt and IEEE divisions
   * (:143,:170,:204,:242,:252), no fused multiply-adds.  With it the default summation order (stencil order) is
   * reproduced bit for bit by a CPU program that sums in the same order; SPH_FLAG_EXACT_ORDER implies it.
   * Without it (the default) the same terms are evaluated with fp32 coefficients, hardware reciprocal and
   * reciprocal square root refined by one Newton step, and fused multiply-adds: each term within a few ulp of
   * the reference's, about a third of the instructions.  Which pairs interact (the test sqrtf(r2) <= h), cell
   * ids and neighbour counts are exact in every mode. */
  SPH_FLAG_REFERENCE_ARITH = 1u << 4
};

/*
 * Simulation parameters.  One field per mutable global of the reference
 * (compute/compute.cpp:8-17); defaults from sph_params_default() are the reference's values.
 * h2/h6/h9 and the kernel coefficients are derived inside the library exactly as the reference
 * derives them (compute.cpp:100-102 and the expressions at :45,:56,:68,:80,:92).
 */
typedef struct sph_params {
  float sound_speed;   /* soundSpeed  = 1.0          compute.cpp:8  */
  float mass;          /* m           = 0.00020543   compute.cpp:9  */
  float rho0;          /* rho0        = 600.0        compute.cpp:10 */
  float dyn_visc;      /* dynVisc     = 0.2          compute.cpp:11 */
  float gravity[3];    /* a_gravity   = {0,-10,0}    compute.cpp:12 */
  float timestep;      /* timestep    = 0.005        compute.cpp:13 */
  float tank_size;     /* tankSize    = 0.2154       compute.cpp:14 */
  float surf_tension;  /* surfTension = 0.0728       compute.cpp:15 */
  float h;             /* h           = 0.01         compute.cpp:17 */
  /* Edge of a neighbour-search cell.  0 selects h*(1+2^-12): with that pad the 27-cell
   * stencil provably contains every pair whose fp32 distance test sqrtf(r2) <= h passes. */
  float cell_size;
  uint32_t flags;      /* SPH_FLAG_*  */
  uint32_t reserved;
} sph_params;

/* Opaque simulation context: one per GPU (one per process rank in multi-GPU runs). */
typedef struct sph_context sph_context;

/* Per-kernel timings accumulated by HIP events on the context's stream. */
typedef struct sph_timings {
  double hash_ms;       /* cell hash + histogram                        */
  double scan_ms;       /* prefix sum over cell counts                  */
  double reorder_ms;    /* scatter + in-cell id ordering + state gather */
  double search_ms;     /* neighbour search: distance tests -> per-particle neighbour bitmask */
  double density_ms;    /* density pass  (reference compute.cpp:131-150) */
  double force_ms;      /* pressure+viscosity+gravity+tension+integrate+walls (:153-319) */
  double other_ms;      /* pack/unpack, memsets                          */
  uint64_t steps;       /* steps accumulated since the last reset        */
  uint64_t particles;   /* sum over steps of particles advanced          */
} sph_timings;

/* Library identification, e.g. "sph_amd 0.1 gfx950". */
const char* sph_version(void);
/* Thread-local text of the last error on the calling thread ("" if none). */
const char* sph_last_error(void);

/* Fill *p with the reference's default parameters (compute.cpp:8-17), surface tension on. */
int sph_params_default(sph_params* p);

/* Number of HIP devices visible to the process; 0 without a GPU (never fails). */
int sph_device_count(void);

/*
 * Create a context on HIP device `device` able to hold n_max particles (own + ghost).
 * The neighbour grid is sized from params->tank_size / cell size.  Counterpart of the
 * allocation half of initParticles (compute.cpp:97-105).
 */
int sph_create(const sph_params* params, int n_max, int device, sph_context** out);
int sph_destroy(sph_context* ctx);

/* Replace mutable parameters between steps (the reference's globals can be edited between
 * updateParticles calls, e.g. a_gravity for sloshing).  h, tank_size and cell_size define the
 * grid: changing them beyond the created grid's capacity is SPH_ERR_CAPACITY. */
int sph_set_params(sph_context* ctx, const sph_params* params);
int sph_get_params(const sph_context* ctx, sph_params* params);

/* Host -> HBM.  pos_aos/vel_aos: float[3*n] in caller order; particle i gets id i. */
int sph_upload(sph_context* ctx, const float* pos_aos, const float* vel_aos, int n);
/* Same, from device pointers (HBM-resident inputs; no PCIe traffic).  Read on the context's stream: see
 * sph_stream_wait when another stream wrote them. */
int sph_upload_device(sph_context* ctx, const float* d_pos_aos, const float* d_vel_aos, int n);

/*
 * Advance nsteps time steps; each step is one updateParticles (compute.cpp:120-320):
 * cell hash -> counting sort -> neighbour search -> density -> forces -> integrate -> tank walls.
 * Asynchronous on the context's stream; sph_download / sph_sync wait for it.
 */
int sph_step(sph_context* ctx, int nsteps);
int sph_sync(sph_context* ctx);
/* sph_step with a HIP event recorded on the context's stream before every step and behind the last one;
 * waits for the steps and returns the device time of each in step_ms[nsteps] (milliseconds).  Measurement aid:
 * the reference has no timers (its main loop, main.cpp:31-35, is not instrumented). */
int sph_step_profile(sph_context* ctx, int nsteps, float* step_ms);

/* HBM -> host, caller order (via the id permutation).  Either pointer may be NULL. */
int sph_download(sph_context* ctx, float* pos_aos, float* vel_aos);
/* Same, to device pointers. */
int sph_download_device(sph_context* ctx, float* d_pos_aos, float* d_vel_aos);

/*
 * Intermediates of the LAST step in caller order; any pointer may be NULL.
 *   rho[n]      densities                          (compute.cpp:148)
 *   force[3n]   total force density f_total        (compute.cpp:122, before integration)
 *   ncount[n]   #{j : sqrtf(r2_ij) <= h}, self included
 *   cell[n]     linear cell id (cx*ny + cy)*nz + cz (z fastest) of the position the step started from
 * force and ncount need SPH_FLAG_KEEP_DEBUG.
 */
int sph_download_debug(sph_context* ctx, float* rho, float* force, int32_t* ncount,
                       int32_t* cell);

/* Number of particles currently held, grid dimensions (cells per axis). */
int sph_num_particles(const sph_context* ctx);
int sph_grid_dims(const sph_context* ctx, int dims[3]);

/* Event timing: when enabled every kernel launch is bracketed by HIP events on the stream it
 * is launched on; sph_get_timings synchronises and returns the accumulated sums.  enable = 1 brackets every
 * kernel category, enable = 2 the force pass only: an event record is a barrier packet on the queue, and twelve of
 * them per step cost a 1M-particle step 8 % (0.476 against 0.436 ms), two cost 1 %. */
int sph_enable_timing(sph_context* ctx, int enable);
int sph_get_timings(sph_context* ctx, sph_timings* out);
int sph_reset_timings(sph_context* ctx);

/*
 * One-call form of the reference's updateParticles(particles, velocities, numParticles)
 * (compute.h:6): upload -> one step -> download on a process-wide default context created on
 * first use (device 0, or $SPH_AMD_DEVICE).  `params` NULL means the reference defaults.
 */
int sph_update_particles(const sph_params* params, float* particles, float* velocities,
                         int num_particles);

/*
 * How sph_update_particles treats the caller's arrays.  Mode 0 (default): upload, step, download on every call - the
 * caller may edit the arrays between calls, as the reference's interface allows.  Mode 1: a call with the same two
 * pointers and particle count as the previous call skips the upload (the state in HBM is what those arrays hold,
 * provided the caller has not written to them since - the reference's main loop, main.cpp:31-35, does not); both
 * arrays are still written on return.  Any other pointers, count or a grid that outgrows the context upload as in
 * mode 0.  Returns the previous mode; any other argument only queries.
 */
int sph_one_call_mode(int mode);

/*
 * The one-call form moves 48 bytes per particle over PCIe every step (the reference's contract: host arrays valid on
 * return, render.cpp:240-253).  From ordinary (pageable) memory those transfers go through the runtime's staging
 * buffers; from page-locked memory they are direct DMA.  sph_host_register page-locks a range the caller allocated
 * (malloc, new[], ...) in place - the pointer stays what it was, free() still works on it at process end - and
 * sph_host_unregister undoes it; unregister before freeing memory that the process goes on to reuse.  Registering a
 * range twice is not an error.  The drop-in initParticles registers the two arrays it hands out.
 */
int sph_host_register(void* host_ptr, size_t bytes);
int sph_host_unregister(void* host_ptr);

/*
 * The same step for callers that cannot allocate: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  (No counterpart in the reference, whose updateParticles mallocs
 * its temporaries on every call, compute.cpp:122,128.)  One updateParticles on DEVICE arrays of
 * 3*num_particles floats each, advanced in place, issued on `stream` (a hipStream_t; NULL = the null
 * stream) of the CURRENT device and not waited for.  The call allocates and frees nothing: every
 * intermediate buffer is a piece of one static device array inside the library
 * (about 15.5 MB, held by each GPU the library runs on until the process ends), which bounds the input:
 * at most SPH_NOALLOC_MAX_PARTICLES particles and SPH_NOALLOC_MAX_CELLS grid cells
 * (SPH_ERR_CAPACITY beyond); SPH_FLAG_KEEP_DEBUG is rejected.  All calls on one GPU share that
 * array: calls that overlap in time on the same GPU (different streams or threads) race.
 */
#define SPH_NOALLOC_MAX_PARTICLES 32768
#define SPH_NOALLOC_MAX_CELLS 262144 /* 64^3: a tank of up to 64 cell edges per axis */
int sph_update_particles_noalloc(const sph_params* params, float* d_particles, float* d_velocities,
                                 int num_particles, void* stream);

/*
 * Counterpart of initParticles' fill (compute.cpp:106-117): uniform random block in
 * [0, tank/2]^3 from a 64-bit LCG seeded with `seed`, zero velocities, written to host arrays
 * the caller owns.  (The reference seeds glibc rand() with time(0), which is not reproducible;
 * a seed is the one deliberate difference.)
 */
int sph_fill_random_block(const sph_params* params, uint64_t seed, float* pos_aos,
                          float* vel_aos, int n);

/*
 * Scene generation in HBM, for fields too large to build on the host and push over PCIe (counterpart of
 * initParticles' fill, compute.cpp:108-117, which draws its block from glibc rand()): the x-planes [x_lo, x_hi) of
 * an nx*ny*nz simple-cubic lattice of the given spacing, every coordinate jittered by +-jitter*spacing with a
 * counter-based generator keyed by (particle index, axis, seed); zero velocities; particle (i,j,k) gets id
 * (i*ny + j)*nz + k.  Any context generates any planes of the same field identically (slabs generate their own).
 * A plain context takes the whole lattice (x_lo = 0, x_hi = nx).  origin NULL = (0,0,0).  Asynchronous.
 */
int sph_fill_lattice(sph_context* ctx, int nx, int ny, int nz, int x_lo, int x_hi, double spacing, double jitter,
                     uint64_t seed, const double origin[3]);

/* ---- viewer hand-off ------------------------------------------------------------------------------------------
 * Counterpart of the reference's display(particles, numParticles) (render/render.cpp:240-260), which copies the
 * positions to the GPU every frame and draws one sphere per particle.  Here the state is drawn where it lives (HBM)
 * and only the finished picture leaves the device.  Camera and shading follow the reference: model matrix =
 * translate(translation) * rotation(quat) with the trackball's quaternion layout (x, y, z, w) and its matrix
 * (render/trackball.cpp:326-346), view matrix = identity, perspective(fov, width/height, z_near, z_far)
 * (render.cpp:155-171), spheres of `radius` (render.cpp:187), light and material of vs_basic.glsl / fs_simple.glsl,
 * white background (render.cpp:297).  Spheres are drawn as per-pixel impostors, not tessellated. */
typedef struct sph_camera {
  float translation[3]; /* modelTranslation = {0, 0, -1.5}   render.cpp:26  */
  float quat[4];        /* curquat = identity {0, 0, 0, 1}   render.cpp:295 */
  float fov_deg;        /* 45                                render.cpp:157 */
  float z_near, z_far;  /* 0.01, 10                          render.cpp:155-156 */
  float radius;         /* 0.0025                            render.cpp:187 */
} sph_camera;
int sph_camera_default(sph_camera* cam);
/* Draw the context's particles into rgb (HOST buffer, 3 * width * height bytes, top row first, R G B).
 * Synchronous.  width, height in [1, 16384]. */
int sph_render_frame(sph_context* ctx, const sph_camera* cam, int width, int height, unsigned char* rgb);

/* ---- slab decomposition (one context per GPU / per process rank) -----------------------------
 *
 * The reference is a single shared-memory process (OpenMP only); it has no counterpart of this
 * section.  Interactions reach no further than h (the kernels vanish beyond it, reference
 * compute.cpp:44,53,64,76,88), so the tank is cut into slabs along x on cell-column boundaries.
 * A slab context owns the global cell columns [col_lo, col_hi) and keeps two ghost columns on
 * each side: the inner one gets its density recomputed locally (it needs the outer one), so a
 * step needs ONE neighbour exchange, which carries migrants and ghosts together:
 *
 *   sph_slab_pack     every particle within two columns of (or beyond) a face -> record buffer
 *   <exchange>        caller moves the buffers to the neighbouring ranks (RCCL send/recv, ...)
 *   sph_slab_append   received records join the particle set
 *   sph_step(ctx, 1)  sort over the local grid, density on own+1 columns, forces/integration on
 *                     own columns; afterwards the context holds exactly the particles whose
 *                     pre-step column was in [col_lo, col_hi)
 *
 * Ownership is decided by column alone, so a migrant is integrated by exactly one rank, and the
 * per-particle summation order is the same as on a single GPU: an N-slab run is bit-identical
 * to the 1-GPU run.  Slabs must be at least 4 columns wide.
 * A record is 7 consecutive 32-bit words: x, y, z, vx, vy, vz, id.
 */
#define SPH_SLAB_RECORD_WORDS 7
#define SPH_SLAB_MIN_COLUMNS 4

/* Cells per axis of the global grid for these parameters. */
int sph_global_cells(const sph_params* params, int* cells_per_axis);
/* Global cell column of the coordinate x: the device's own expression (fp32 x / cell edge, floor, clamped to the grid),
 * so that a host that deals particles out to slabs puts every one where the sort will find it.  Host arithmetic. */
int sph_column_of(const sph_params* params, float x, int* column);

/* Cut planes.  Host arithmetic only (no device is touched): one implementation for drivers in any language.
 * load[c] is what global column c costs - particles, or the work figure of sph_slab_load_work; cuts[0..nslabs] with
 * cuts[0] = 0 and cuts[nslabs] = ncols, slab r owning columns [cuts[r], cuts[r+1]).
 *   sph_slab_plan_partition   the cuts whose cumulative load is closest to equal shares, every slab >= min_cols wide.
 *   sph_slab_plan_next        the cuts for the records of the next step: every interior plane moves towards the
 *                             balanced partition by at most *face_move columns - as many as the fastest particle
 *                             crosses per step (column_shift, the largest sph_slab_max_column_shift of all slabs), at
 *                             most SPH_SLAB_MAX_FACE_MOVE, fewer when the narrowest slab cannot afford them - keeping
 *                             every slab at least ceil(column_shift + shift_margin) + 2 + *face_move columns wide and,
 *                             with cols_capacity (one entry per slab, or NULL), no wider than its context allows.
 *                             shift_margin: columns of allowance for a shift measured some steps before the planes it
 *                             sizes are used (0 when measured between the same two steps).  *face_shift = column_shift
 *                             + shift_margin is what sph_slab_set_face_shift wants; next_cuts[r], next_cuts[r+1] what
 *                             sph_slab_set_next_columns of slab r wants.  face_move and face_shift may be NULL. */
int sph_slab_plan_partition(const int64_t* load, int ncols, int nslabs, int min_cols, int32_t* cuts);
int sph_slab_plan_next(const int64_t* load, int ncols, int nslabs, const int32_t* cuts, float column_shift,
                       float shift_margin, const int32_t* cols_capacity, int32_t* next_cuts, int* face_move,
                       float* face_shift);

/* Like sph_create, for the slab [col_lo, col_hi); cols_capacity (>= col_hi - col_lo) bounds how
 * wide the slab may become through sph_slab_set_columns. */
int sph_slab_create(const sph_params* params, int n_max, int device, int col_lo, int col_hi,
                    int cols_capacity, sph_context** out);
/* Move the slab's faces (load balancing).  All ranks must switch between the same two steps and
 * a face may move by at most one column per step. */
int sph_slab_set_columns(sph_context* ctx, int col_lo, int col_hi);
int sph_slab_get_columns(const sph_context* ctx, int* col_lo, int* col_hi, int* global_cols);

/* Host -> HBM with explicit global particle ids (int32). */
int sph_slab_upload(sph_context* ctx, const float* pos_aos, const float* vel_aos,
                    const int32_t* ids, int n);
/* HBM -> host in storage order with ids; *n_out particles (<= capacity). */
int sph_slab_download(sph_context* ctx, float* pos_aos, float* vel_aos, int32_t* ids,
                      int capacity, int* n_out);
/* Number of particles the last step advanced (the context's own particles; what sph_slab_download_debug describes).
 * sph_num_particles may be larger: it includes records received since for the next step. */
int sph_slab_last_own(const sph_context* ctx);
/* Intermediates of the last step for the first sph_slab_last_own particles sph_slab_download returns, same order. */
int sph_slab_download_debug(sph_context* ctx, float* rho, float* force, int32_t* ncount,
                            int32_t* cell, int capacity);

/* Fill the caller's DEVICE buffers (capacity records each) with the records the left / right
 * neighbour needs; counts come back on the host.  Pass NULL for a face at the tank wall.
 * After the first call a context packs these records right behind every step (into buffers of its
 * own, counts travelling home with the particle count), so that the call between two steps is a
 * copy instead of a kernel and a second round trip; anything that changes the particle set or the
 * cut planes in between makes it pack on demand again.  Those buffers are allocated by the first call and
 * again only when `capacity` differs from the previous call's: keep it constant over a run.  The header
 * count of this caller-driven protocol is the caller's to transport; sph_amd/slab.py sends it as an fp32
 * word, exact to 2^24 records per face (the library-driven exchange below carries it as an integer). */
int sph_slab_pack(sph_context* ctx, float* d_rec_left, float* d_rec_right, int capacity,
                  int* n_left, int* n_right);
/* Add n_rec records from a DEVICE buffer to the particle set.  Read on the context's stream: see sph_stream_wait
 * when another stream (the transport's) wrote them. */
int sph_slab_append(sph_context* ctx, const float* d_rec, int n_rec);

/* Particles per OWN column at the last sort (host array, ncols = col_hi - col_lo entries). */
int sph_slab_column_counts(sph_context* ctx, int32_t* counts, int capacity, int* ncols);
/* sph_slab_column_counts and sph_slab_max_column_shift in one round trip to the device: what a load
 * balancer asks for every time it runs. */
int sph_slab_load_info(sph_context* ctx, int32_t* counts, int capacity, int* ncols, float* max_shift);
/* The same plus work[col] = sum over the column's particles of (9 + neighbours at the last search): what a step costs
 * is about proportional to that, not to the particle count (a particle in fluid compressed to 50 neighbours costs
 * 2.7 times one at the rest density's 13).  Entry k describes global column *first_col + k: the own columns AT THE LAST
 * SORT, which lag sph_slab_set_next_columns by a step. */
int sph_slab_load_work(sph_context* ctx, int32_t* counts, int32_t* work, int capacity, int* first_col, int* ncols,
                       float* max_shift);
/* Total number of RECEIVED records so far whose x-column lay outside this context's local grid (own
 * columns +- 2): particles that no context integrates any more.  Must stay 0; it cannot stay 0 when
 * particles cross more columns per step than the neighbouring slab is wide (see
 * sph_slab_max_column_shift).  Own particles that leave the local grid are not counted: sph_slab_pack
 * handed them to the neighbour first. */
int sph_slab_lost(sph_context* ctx, int64_t* lost);
/* max |vx| * timestep / cell edge over the context's particles: the number of columns a particle can
 * cross in the next step.  With one exchange per step between direct neighbours, every slab must be
 * at least ceil(that) + 2 columns wide (+ 1 while cut planes move).  Synchronous. */
int sph_slab_max_column_shift(sph_context* ctx, float* columns);

/* ---- exchange issued by the library, overlapped with the interior columns ----------------------------------------
 *
 * Instead of pack / <caller's exchange> / append, a slab context attached to a transport moves its records itself:
 *   RCCL   ncclSend/ncclRecv with ranks rank-1 and rank+1 inside one ncclGroup on a communication stream of the
 *          context's own (librccl is loaded at run time by the first sph_comm_unique_id / sph_slab_attach_rccl);
 *   LOCAL  the neighbours are contexts of the same process (several slabs on one GPU, tests): device copies.
 * Protocol per step, one host wait in all:
 *   sph_slab_step_begin   sort and search over own + received particles; force pass over the columns next to the
 *                         faces; records for the NEXT step packed from them and posted on the communication stream;
 *                         force pass over the interior columns meanwhile on the compute stream.  Asynchronous.
 *                         (Slabs below four million particles advance all columns in one launch and post the records
 *                         afterwards: see sph_slab_set_overlap.)
 *   sph_slab_step_end     waits for the exchange only (counts arrive in pinned host memory), sends the remainder of a
 *                         message that outgrew its agreed size, and queues the received records behind the interior
 *                         force pass.  Afterwards sph_num_particles is exact and the next step may begin.
 * With several contexts in one process call begin on all of them, then end on all of them.  Before the first step,
 * and after anything that replaces the particle set, sph_slab_halo_begin/_end exchange the halo of the state as it
 * stands.  rec_capacity (records per message, the same on every rank, at most 2^31/28) bounds what one face can send
 * per step: SPH_ERR_CAPACITY beyond.  Results are bit-identical to the pack/append protocol and to a single context.
 * The face ranges are sph_slab_set_face_shift (+3) columns wide - pass the largest sph_slab_max_column_shift of all
 * slabs; sph_slab_missed counts interior particles that reached a neighbour's halo anyway (must stay 0). */
int sph_comm_unique_id(void* id128);  /* 128 bytes, made on one rank and handed to all (any out-of-band channel) */
int sph_slab_attach_rccl(sph_context* ctx, const void* id128, int rank, int nranks, int rec_capacity);
/* LOCAL: the neighbours are borrowed, not owned: destroy the contexts of a chain together, after the last exchange. */
int sph_slab_attach_local(sph_context* ctx, sph_context* left, sph_context* right, int rec_capacity);
int sph_slab_halo_begin(sph_context* ctx);
int sph_slab_halo_end(sph_context* ctx);
int sph_slab_halo_exchange(sph_context* ctx); /* begin + end: one context per process */
int sph_slab_step_begin(sph_context* ctx);
int sph_slab_step_end(sph_context* ctx);
int sph_slab_exchange_step(sph_context* ctx); /* begin + end: one context per process */
/* Faces for the records of the next sph_slab_step_begin and for every step after it (load balancing): all ranks
 * must call it between the same two steps; a face moves by at most SPH_SLAB_MAX_FACE_MOVE columns per step (the face
 * ranges of that step widen by as many columns: a front that crosses three columns per step can be followed at its
 * own pace). */
#define SPH_SLAB_MAX_FACE_MOVE 8
int sph_slab_set_next_columns(sph_context* ctx, int col_lo, int col_hi);
int sph_slab_set_face_shift(sph_context* ctx, float columns);
/* How sph_slab_step_begin orders its force pass: 1 = face columns first, interior columns while the records travel;
 * 0 = all own columns in one launch, records packed and posted afterwards (nothing hidden, but no small launches);
 * 2 (default) = 1 for slabs of four million particles or more, 0 below.  Results are the same bits either way. */
int sph_slab_set_overlap(sph_context* ctx, int mode);
/* sph_slab_missed fetches the counter when called.  The interior check of a step runs behind that step's interior
 * force pass, after the step's counters have gone home: the figure sph_slab_step_end keeps internally lags by a step,
 * this call does not.  Cumulative either way. */
int sph_slab_missed(sph_context* ctx, int64_t* missed);

/* Message sizes.  A message's size must be known at both ends before it is posted, so both ends derive it from the
 * true count of the previous message on that face (the sender knows what it sent, the receiver reads it from the
 * header record): next = min(rec_capacity, max(min_records, roundup256(count * (1 + headroom) + 1))); defaults 0.25 and
 * 1024.  A message that outgrows its size sends the remainder in a second round inside sph_slab_step_end.  Both
 * transports run this protocol: LOCAL copies exactly the agreed number of records per round and refuses
 * (SPH_ERR_STATE) when the two ends of a face disagree on a size.  Set the same values on every rank, before the first
 * exchange.  An error from sph_slab_step_end / sph_slab_halo_end (SPH_ERR_CAPACITY: a face needed more than rec_capacity
 * records, or own + received particles exceed n_max) ends the run on every rank: the failing context posts its
 * remainder round first, so no neighbour is left waiting, and is left without a current halo. */
int sph_slab_set_message_sizing(sph_context* ctx, float headroom, int min_records);
/* Exchanges posted so far, how many of them needed a remainder round, and the sizes (records behind the header) of the
 * next message to / from the left and right neighbour.  Any pointer may be NULL. */
int sph_slab_exchange_stats(const sph_context* ctx, uint64_t* exchanges, uint64_t* remainder_rounds, int next_out[2],
                            int next_in[2]);
/* Diagnostic of the RCCL transport on the current device: a one-rank communicator is created and the exchange's group
 * of ncclSend/ncclRecv is posted with this rank as both neighbours - words_left floats from d_send_left arrive in
 * d_recv_left, words_right floats from d_send_right in d_recv_right (device buffers, none overlapping).  Synchronous. */
int sph_comm_self_exchange(const float* d_send_left, float* d_recv_left, size_t words_left, const float* d_send_right,
                           float* d_recv_right, size_t words_right);

/* The HIP stream (hipStream_t) all of the context's work is issued on; sph_set_stream makes the
 * context issue on a caller-owned stream instead (e.g. the stream a communication library
 * orders its transfers against). */
int sph_get_stream(sph_context* ctx, void** stream);
int sph_set_stream(sph_context* ctx, void* stream);
/* Device buffers handed to sph_upload_device / sph_slab_append are read on the context's stream.  When another
 * stream produced them (a communication library's, a framework's), the caller must order the two: this call
 * makes everything the context issues from now on wait for the work submitted to producer_stream so far
 * (event record + stream wait, no host synchronisation).  The other direction needs nothing: sph_download_device
 * and sph_slab_pack return only when their output is complete. */
int sph_stream_wait(sph_context* ctx, void* producer_stream);

#ifdef __cplusplus
}
#endif
#endif /* SPH_AMD_H */
