"""ctypes binding of libsph_amd.so (the C ABI in include/sph_amd.h).

The library is the product; this module only marshals numpy arrays and raw device pointers
across the C boundary.  Loading fails loudly when the in-tree library has not been built —
there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from .params import (SPH_FLAG_EXACT_ORDER, SPH_FLAG_KEEP_DEBUG, SPH_FLAG_SURFACE_TENSION,  # noqa: F401
                     SphParams, SphTimings, default_params)

LIB_PATH = os.environ.get("SPH_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsph_amd.so")

# Every symbol include/sph_amd.h declares: (name, restype, argtypes)
_P = C.POINTER(SphParams)
_CTX = C.c_void_p


class SphCamera(C.Structure):
    """struct sph_camera of include/sph_amd.h (the reference's camera, render/render.cpp:26,155-157,187)."""
    _fields_ = [("translation", C.c_float * 3), ("quat", C.c_float * 4), ("fov_deg", C.c_float),
                ("z_near", C.c_float), ("z_far", C.c_float), ("radius", C.c_float)]


_SIGNATURES = [
    ("sph_version", C.c_char_p, []),
    ("sph_last_error", C.c_char_p, []),
    ("sph_params_default", C.c_int, [_P]),
    ("sph_device_count", C.c_int, []),
    ("sph_create", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_CTX)]),
    ("sph_destroy", C.c_int, [_CTX]),
    ("sph_set_params", C.c_int, [_CTX, _P]),
    ("sph_get_params", C.c_int, [_CTX, _P]),
    ("sph_upload", C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_int]),
    ("sph_upload_device", C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_int]),
    ("sph_step", C.c_int, [_CTX, C.c_int]),
    ("sph_sync", C.c_int, [_CTX]),
    ("sph_step_profile", C.c_int, [_CTX, C.c_int, C.c_void_p]),
    ("sph_download", C.c_int, [_CTX, C.c_void_p, C.c_void_p]),
    ("sph_download_device", C.c_int, [_CTX, C.c_void_p, C.c_void_p]),
    ("sph_download_debug", C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("sph_num_particles", C.c_int, [_CTX]),
    ("sph_grid_dims", C.c_int, [_CTX, C.POINTER(C.c_int * 3)]),
    ("sph_enable_timing", C.c_int, [_CTX, C.c_int]),
    ("sph_get_timings", C.c_int, [_CTX, C.POINTER(SphTimings)]),
    ("sph_reset_timings", C.c_int, [_CTX]),
    ("sph_update_particles", C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_int]),
    ("sph_one_call_mode", C.c_int, [C.c_int]),
    ("sph_host_register", C.c_int, [C.c_void_p, C.c_size_t]),
    ("sph_host_unregister", C.c_int, [C.c_void_p]),
    ("sph_update_particles_noalloc", C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("sph_fill_lattice", C.c_int, [_CTX, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint64,
                                   C.c_void_p]),
    ("sph_fill_random_block", C.c_int, [_P, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]),
    # slab decomposition
    ("sph_global_cells", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("sph_column_of", C.c_int, [_P, C.c_float, C.POINTER(C.c_int)]),
    ("sph_slab_plan_partition", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("sph_slab_plan_next", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                     C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    ("sph_slab_create", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_CTX)]),
    ("sph_slab_set_columns", C.c_int, [_CTX, C.c_int, C.c_int]),
    ("sph_slab_get_columns", C.c_int, [_CTX, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("sph_slab_upload", C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("sph_slab_download", C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    ("sph_slab_last_own", C.c_int, [_CTX]),
    ("sph_slab_download_debug", C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("sph_slab_pack", C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("sph_slab_append", C.c_int, [_CTX, C.c_void_p, C.c_int]),
    ("sph_slab_column_counts", C.c_int, [_CTX, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    ("sph_slab_lost", C.c_int, [_CTX, C.POINTER(C.c_int64)]),
    ("sph_slab_max_column_shift", C.c_int, [_CTX, C.POINTER(C.c_float)]),
    ("sph_slab_load_info", C.c_int, [_CTX, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    ("sph_slab_load_work", C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_float)]),
    ("sph_comm_unique_id", C.c_int, [C.c_void_p]),
    ("sph_slab_attach_rccl", C.c_int, [_CTX, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    ("sph_slab_attach_local", C.c_int, [_CTX, _CTX, _CTX, C.c_int]),
    ("sph_slab_halo_begin", C.c_int, [_CTX]),
    ("sph_slab_halo_end", C.c_int, [_CTX]),
    ("sph_slab_halo_exchange", C.c_int, [_CTX]),
    ("sph_slab_step_begin", C.c_int, [_CTX]),
    ("sph_slab_step_end", C.c_int, [_CTX]),
    ("sph_slab_exchange_step", C.c_int, [_CTX]),
    ("sph_slab_set_next_columns", C.c_int, [_CTX, C.c_int, C.c_int]),
    ("sph_slab_set_face_shift", C.c_int, [_CTX, C.c_float]),
    ("sph_slab_set_overlap", C.c_int, [_CTX, C.c_int]),
    ("sph_slab_missed", C.c_int, [_CTX, C.POINTER(C.c_int64)]),
    ("sph_slab_set_message_sizing", C.c_int, [_CTX, C.c_float, C.c_int]),
    ("sph_slab_exchange_stats", C.c_int, [_CTX, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int * 2),
                                          C.POINTER(C.c_int * 2)]),
    ("sph_comm_self_exchange", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("sph_get_stream", C.c_int, [_CTX, C.POINTER(C.c_void_p)]),
    ("sph_set_stream", C.c_int, [_CTX, C.c_void_p]),
    ("sph_stream_wait", C.c_int, [_CTX, C.c_void_p]),
    ("sph_camera_default", C.c_int, [C.POINTER(SphCamera)]),
    ("sph_render_frame", C.c_int, [_CTX, C.POINTER(SphCamera), C.c_int, C.c_int, C.c_void_p]),
]

NOALLOC_MAX_PARTICLES = 32768  # SPH_NOALLOC_MAX_PARTICLES
NOALLOC_MAX_CELLS = 262144     # SPH_NOALLOC_MAX_CELLS
RECORD_WORDS = 7      # SPH_SLAB_RECORD_WORDS: x, y, z, vx, vy, vz, id
MIN_SLAB_COLUMNS = 4  # SPH_SLAB_MIN_COLUMNS
MAX_FACE_MOVE = 8     # SPH_SLAB_MAX_FACE_MOVE

_lib = None


class SphError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"sph_amd error {code}: {message}")
        self.code = code


def load():
    """Load libsph_amd.so and bind every declared symbol; raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch ships its own libamdhip64/libhsa-runtime64.  A process must run on ONE HIP runtime:
        # if libsph_amd.so pulled in /opt/rocm's copy first, a later `import torch` would find no
        # GPU.  Loading torch first makes both sides resolve to the same runtime.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP library first (make -C sph_amd/csrc, or "
            "__graft_entry__.build()).  sph_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in _SIGNATURES:
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def exported_symbols():
    return [name for name, _, _ in _SIGNATURES]


def _check(rc):
    if rc != 0:
        raise SphError(rc, load().sph_last_error().decode())


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def default_camera():
    cam = SphCamera()
    _check(load().sph_camera_default(C.byref(cam)))
    return cam


def _render(sim, width, height, camera=None):
    """The context's particles drawn as shaded spheres: uint8 array [height, width, 3], top row first."""
    cam = camera if camera is not None else default_camera()
    img = np.empty((int(height), int(width), 3), dtype=np.uint8)
    _check(sim.lib.sph_render_frame(sim._ctx, C.byref(cam), int(width), int(height), img.ctypes.data))
    return img


def _fill_lattice(sim, nx, ny, nz, x_range=None, spacing=None, jitter=0.1, seed=12345, origin=None):
    from .scenes import REST_SPACING
    lo, hi = (0, nx) if x_range is None else x_range
    o = (C.c_double * 3)(*origin) if origin is not None else None
    _check(sim.lib.sph_fill_lattice(sim._ctx, int(nx), int(ny), int(nz), int(lo), int(hi),
                                    float(REST_SPACING if spacing is None else spacing), float(jitter), int(seed), o))


class Simulation:
    """One GPU context: mirrors the reference's initParticles / updateParticles life cycle
    (reference compute/compute.h:4-6) with the state resident in HBM between steps."""

    def __init__(self, params=None, n_max=3000, device=0):
        self.lib = load()
        self.params = params.copy() if params is not None else default_params()
        self._ctx = _CTX()
        _check(self.lib.sph_create(C.byref(self.params), int(n_max), int(device), C.byref(self._ctx)))
        self.n_max = int(n_max)

    def close(self):
        if self._ctx:
            self.lib.sph_destroy(self._ctx)
            self._ctx = _CTX()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def n(self):
        return int(self.lib.sph_num_particles(self._ctx))

    def grid_dims(self):
        d = (C.c_int * 3)()
        _check(self.lib.sph_grid_dims(self._ctx, C.byref(d)))
        return tuple(d)

    def set_params(self, params):
        _check(self.lib.sph_set_params(self._ctx, C.byref(params)))
        self.params = params.copy()

    def set_gravity(self, g):
        p = self.params.copy()
        p.gravity[0], p.gravity[1], p.gravity[2] = float(g[0]), float(g[1]), float(g[2])
        self.set_params(p)

    def upload(self, pos, vel):
        pos, vel = _f32(pos), _f32(vel)
        n = pos.size // 3
        assert vel.size == pos.size
        _check(self.lib.sph_upload(self._ctx, pos.ctypes.data, vel.ctypes.data, n))

    def fill_lattice(self, nx, ny, nz, **kw):
        """Generate scenes.jittered_lattice(nx, ny, nz, ...) directly in HBM (same particles, bit for bit)."""
        _fill_lattice(self, nx, ny, nz, None, **kw)

    def render(self, width, height, camera=None):
        return _render(self, width, height, camera)

    def upload_device(self, d_pos_ptr, d_vel_ptr, n, producer_stream=None):
        """producer_stream: the stream that wrote the buffers (a hipStream_t as int); the context waits for it."""
        if producer_stream is not None:
            _check(self.lib.sph_stream_wait(self._ctx, C.c_void_p(producer_stream)))
        _check(self.lib.sph_upload_device(self._ctx, C.c_void_p(d_pos_ptr), C.c_void_p(d_vel_ptr), int(n)))

    def step(self, nsteps=1):
        _check(self.lib.sph_step(self._ctx, int(nsteps)))

    def sync(self):
        _check(self.lib.sph_sync(self._ctx))

    def step_profile(self, nsteps):
        """nsteps steps, waited for; returns the device time of every step in ms (HIP events on the context's stream)."""
        ms = np.zeros(int(nsteps), np.float32)
        _check(self.lib.sph_step_profile(self._ctx, int(nsteps), ms.ctypes.data))
        return ms

    def download(self):
        n = self.n
        pos = np.empty((n, 3), np.float32)
        vel = np.empty((n, 3), np.float32)
        _check(self.lib.sph_download(self._ctx, pos.ctypes.data, vel.ctypes.data))
        return pos, vel

    def download_device(self, d_pos_ptr, d_vel_ptr):
        _check(self.lib.sph_download_device(self._ctx, C.c_void_p(d_pos_ptr), C.c_void_p(d_vel_ptr)))

    def download_debug(self, force=True, ncount=True):
        n = self.n
        rho = np.empty(n, np.float32)
        f = np.empty((n, 3), np.float32) if force else None
        nc = np.empty(n, np.int32) if ncount else None
        cell = np.empty(n, np.int32)
        _check(self.lib.sph_download_debug(self._ctx, rho.ctypes.data, f.ctypes.data if force else None,
                                           nc.ctypes.data if ncount else None, cell.ctypes.data))
        return rho, f, nc, cell

    def enable_timing(self, on=True):
        """on: False/0 off, True/1 every kernel category, 2 the force pass only (cheap enough for a timed region)."""
        _check(self.lib.sph_enable_timing(self._ctx, int(on)))

    def reset_timings(self):
        _check(self.lib.sph_reset_timings(self._ctx))

    def timings(self):
        t = SphTimings()
        _check(self.lib.sph_get_timings(self._ctx, C.byref(t)))
        return {k: getattr(t, k) for k, _ in SphTimings._fields_}


def comm_unique_id():
    """128 opaque bytes identifying a new RCCL communicator: make them on one rank, hand them to all."""
    buf = (C.c_char * 128)()
    _check(load().sph_comm_unique_id(buf))
    return bytes(buf)


def comm_self_exchange(send_left, recv_left, send_right, recv_right):
    """The slab exchange's RCCL group (send left, receive left, send right, receive right) with this rank as both
    neighbours: torch CUDA float32 tensors, send and receive of a side equally long."""
    assert send_left.numel() == recv_left.numel() and send_right.numel() == recv_right.numel()
    _check(load().sph_comm_self_exchange(C.c_void_p(send_left.data_ptr()), C.c_void_p(recv_left.data_ptr()),
                                         send_left.numel(), C.c_void_p(send_right.data_ptr()),
                                         C.c_void_p(recv_right.data_ptr()), send_right.numel()))


def global_cells(params):
    n = C.c_int()
    _check(load().sph_global_cells(C.byref(params), C.byref(n)))
    return n.value


def column_of(params, x):
    """Global cell column of coordinate x, by the library's own expression (host arithmetic)."""
    c = C.c_int()
    _check(load().sph_column_of(C.byref(params), float(x), C.byref(c)))
    return c.value


def plan_partition(load_per_column, nslabs, min_cols=MIN_SLAB_COLUMNS):
    """sph_slab_plan_partition: balanced cuts[0..nslabs] over the columns (host arithmetic, no device)."""
    hist = np.ascontiguousarray(load_per_column, dtype=np.int64)
    cuts = np.zeros(nslabs + 1, np.int32)
    _check(load().sph_slab_plan_partition(hist.ctypes.data, hist.size, nslabs, min_cols, cuts.ctypes.data))
    return [int(c) for c in cuts]


def plan_next(load_per_column, cuts, column_shift, shift_margin=0.0, cols_capacity=None):
    """sph_slab_plan_next: (next cuts, face move, face shift) for the records of the next step."""
    hist = np.ascontiguousarray(load_per_column, dtype=np.int64)
    now = np.ascontiguousarray(cuts, dtype=np.int32)
    nslabs = now.size - 1
    caps = None if cols_capacity is None else np.ascontiguousarray(cols_capacity, dtype=np.int32)
    if caps is not None and caps.size != nslabs:
        raise ValueError("cols_capacity needs one entry per slab")
    out = np.zeros(nslabs + 1, np.int32)
    move, shift = C.c_int(), C.c_float()
    _check(load().sph_slab_plan_next(hist.ctypes.data, hist.size, nslabs, now.ctypes.data, float(column_shift),
                                     float(shift_margin), None if caps is None else caps.ctypes.data, out.ctypes.data,
                                     C.byref(move), C.byref(shift)))
    return [int(c) for c in out], move.value, shift.value


class SlabSimulation:
    """One slab of the x-decomposition on one GPU (one per process rank).

    Record buffers handed to / returned by pack() and append() are torch CUDA tensors of shape
    [k, RECORD_WORDS] float32: torch only provides the device memory and the transport.
    """

    def __init__(self, params, n_max, col_lo, col_hi, cols_capacity=0, device=0, rec_capacity=None):
        import torch
        self.torch = torch
        self.lib = load()
        self.params = params.copy()
        self._ctx = _CTX()
        _check(self.lib.sph_slab_create(C.byref(self.params), int(n_max), int(device), int(col_lo), int(col_hi),
                                        int(cols_capacity), C.byref(self._ctx)))
        self.n_max = int(n_max)
        self.cols_capacity = max(int(cols_capacity), int(col_hi) - int(col_lo))  # the widest this slab may become
        self.device = torch.device("cuda", device)
        self.rec_capacity = int(rec_capacity if rec_capacity is not None else max(1024, n_max // 4))
        self._send_left = torch.empty((self.rec_capacity, RECORD_WORDS), dtype=torch.float32, device=self.device)
        self._send_right = torch.empty((self.rec_capacity, RECORD_WORDS), dtype=torch.float32, device=self.device)

    def close(self):
        if self._ctx:
            self.lib.sph_destroy(self._ctx)
            self._ctx = _CTX()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- columns ----------------------------------------------------------------------------------
    def columns(self):
        lo, hi, g = C.c_int(), C.c_int(), C.c_int()
        _check(self.lib.sph_slab_get_columns(self._ctx, C.byref(lo), C.byref(hi), C.byref(g)))
        return lo.value, hi.value, g.value

    def set_columns(self, lo, hi):
        _check(self.lib.sph_slab_set_columns(self._ctx, int(lo), int(hi)))

    def column_counts(self):
        out = np.empty(self.columns()[2], np.int32)
        n = C.c_int()
        _check(self.lib.sph_slab_column_counts(self._ctx, out.ctypes.data, out.size, C.byref(n)))
        return out[:n.value]  # the columns of the last sort (fewer than now if the slab has been widened since)

    # -- state ------------------------------------------------------------------------------------
    @property
    def n(self):
        return int(self.lib.sph_num_particles(self._ctx))

    def upload(self, pos, vel, ids):
        pos, vel = _f32(pos), _f32(vel)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        _check(self.lib.sph_slab_upload(self._ctx, pos.ctypes.data, vel.ctypes.data, ids.ctypes.data, ids.size))

    def fill_lattice(self, nx, ny, nz, x_range, **kw):
        """Generate the x-planes x_range of scenes.jittered_lattice(nx, ny, nz, ...) directly in HBM, global ids."""
        _fill_lattice(self, nx, ny, nz, x_range, **kw)

    def download(self):
        cap = self.n_max
        pos = np.empty((cap, 3), np.float32)
        vel = np.empty((cap, 3), np.float32)
        ids = np.empty(cap, np.int32)
        n = C.c_int()
        _check(self.lib.sph_slab_download(self._ctx, pos.ctypes.data, vel.ctypes.data, ids.ctypes.data, cap,
                                          C.byref(n)))
        return pos[:n.value], vel[:n.value], ids[:n.value]

    def download_debug(self):
        cap = self.n_max
        rho = np.empty(cap, np.float32)
        f = np.empty((cap, 3), np.float32)
        nc = np.empty(cap, np.int32)
        cell = np.empty(cap, np.int32)
        _check(self.lib.sph_slab_download_debug(self._ctx, rho.ctypes.data, f.ctypes.data, nc.ctypes.data,
                                                cell.ctypes.data, cap))
        n = int(self.lib.sph_slab_last_own(self._ctx))  # the particles the last step advanced
        return rho[:n], f[:n], nc[:n], cell[:n]

    def render(self, width, height, camera=None):
        return _render(self, width, height, camera)

    def last_own(self):
        """Particles the last step advanced (the first ones download() returns)."""
        return int(self.lib.sph_slab_last_own(self._ctx))

    def set_params(self, params):
        _check(self.lib.sph_set_params(self._ctx, C.byref(params)))
        self.params = params.copy()

    def set_gravity(self, g):
        p = self.params.copy()
        p.gravity[0], p.gravity[1], p.gravity[2] = float(g[0]), float(g[1]), float(g[2])
        self.set_params(p)

    def lost(self):
        v = C.c_int64()
        _check(self.lib.sph_slab_lost(self._ctx, C.byref(v)))
        return v.value

    def load_info(self):
        """(particles per own column at the last sort, max column shift): one device round trip."""
        out = np.empty(self.columns()[2], np.int32)
        n, v = C.c_int(), C.c_float()
        _check(self.lib.sph_slab_load_info(self._ctx, out.ctypes.data, out.size, C.byref(n), C.byref(v)))
        return out[:n.value], float(v.value)

    def load_work(self):
        """(first column, particles per column, work per column = sum of 9 + neighbours, max column shift) for the own
        columns of the last sort: one round trip."""
        gn = self.columns()[2]
        counts, work = np.empty(gn, np.int32), np.empty(gn, np.int32)
        first, n, v = C.c_int(), C.c_int(), C.c_float()
        _check(self.lib.sph_slab_load_work(self._ctx, counts.ctypes.data, work.ctypes.data, gn, C.byref(first), C.byref(n),
                                           C.byref(v)))
        return first.value, counts[:n.value], work[:n.value], float(v.value)

    def max_column_shift(self):
        """Columns a particle can cross in the next step: max |vx| * dt / cell over this slab's particles."""
        v = C.c_float()
        _check(self.lib.sph_slab_max_column_shift(self._ctx, C.byref(v)))
        return float(v.value)

    # -- the per-step protocol -----------------------------------------------------------------------
    def pack(self):
        """Records for the (left, right) neighbour: views of this object's send buffers."""
        nl, nr = C.c_int(), C.c_int()
        _check(self.lib.sph_slab_pack(self._ctx, self._send_left.data_ptr(), self._send_right.data_ptr(),
                                      self.rec_capacity, C.byref(nl), C.byref(nr)))
        return self._send_left[:nl.value], self._send_right[:nr.value]

    def append(self, records):
        if records is None or records.shape[0] == 0:
            return
        assert records.is_cuda and records.dtype == self.torch.float32 and records.is_contiguous()
        # torch (or RCCL through it) filled `records` on torch's current stream; the context reads them on its own
        _check(self.lib.sph_stream_wait(self._ctx, C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)))
        _check(self.lib.sph_slab_append(self._ctx, records.data_ptr(), int(records.shape[0])))

    def step(self, nsteps=1):
        _check(self.lib.sph_step(self._ctx, int(nsteps)))

    def sync(self):
        _check(self.lib.sph_sync(self._ctx))

    def use_stream(self, stream_ptr):
        _check(self.lib.sph_set_stream(self._ctx, C.c_void_p(stream_ptr)))

    # -- exchange issued by the library -----------------------------------------------------------------
    def attach_rccl(self, unique_id, rank, nranks, rec_capacity=None):
        """unique_id: the 128 bytes of sph_amd.capi.comm_unique_id(), made on one rank and handed to all."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        _check(self.lib.sph_slab_attach_rccl(self._ctx, buf, int(rank), int(nranks),
                                             int(rec_capacity or self.rec_capacity)))

    def attach_local(self, left, right, rec_capacity=None):
        _check(self.lib.sph_slab_attach_local(self._ctx, left._ctx if left is not None else None,
                                              right._ctx if right is not None else None,
                                              int(rec_capacity or self.rec_capacity)))

    def halo_begin(self):
        _check(self.lib.sph_slab_halo_begin(self._ctx))

    def halo_end(self):
        _check(self.lib.sph_slab_halo_end(self._ctx))

    def step_begin(self):
        _check(self.lib.sph_slab_step_begin(self._ctx))

    def step_end(self):
        _check(self.lib.sph_slab_step_end(self._ctx))

    def set_next_columns(self, lo, hi):
        _check(self.lib.sph_slab_set_next_columns(self._ctx, int(lo), int(hi)))

    def set_overlap(self, mode):
        """0: one force launch, then pack; 1: faces first, interior beside the exchange; 2 (default): by slab size."""
        _check(self.lib.sph_slab_set_overlap(self._ctx, int(mode)))

    def set_face_shift(self, columns):
        _check(self.lib.sph_slab_set_face_shift(self._ctx, float(columns)))

    def missed(self):
        v = C.c_int64()
        _check(self.lib.sph_slab_missed(self._ctx, C.byref(v)))
        return v.value

    def set_message_sizing(self, headroom, min_records):
        """How the next message is sized from the last true count (the same on every rank)."""
        _check(self.lib.sph_slab_set_message_sizing(self._ctx, float(headroom), int(min_records)))

    def exchange_stats(self):
        """(exchanges posted, exchanges that needed a remainder round, next sizes out [left, right], next sizes in)."""
        a, b = C.c_uint64(), C.c_uint64()
        o, i = (C.c_int * 2)(), (C.c_int * 2)()
        _check(self.lib.sph_slab_exchange_stats(self._ctx, C.byref(a), C.byref(b), C.byref(o), C.byref(i)))
        return a.value, b.value, list(o), list(i)

    def empty_records(self, n):
        return self.torch.empty((n, RECORD_WORDS), dtype=self.torch.float32, device=self.device)

    def enable_timing(self, on=True):
        """on: False/0 off, True/1 every kernel category, 2 the force pass only (cheap enough for a timed region)."""
        _check(self.lib.sph_enable_timing(self._ctx, int(on)))

    def reset_timings(self):
        _check(self.lib.sph_reset_timings(self._ctx))

    def timings(self):
        t = SphTimings()
        _check(self.lib.sph_get_timings(self._ctx, C.byref(t)))
        return {k: getattr(t, k) for k, _ in SphTimings._fields_}


def update_particles(pos, vel, params=None):
    """One-call updateParticles (reference compute.h:6): arrays are advanced in place."""
    lib = load()
    assert pos.dtype == np.float32 and vel.dtype == np.float32 and pos.flags.c_contiguous and vel.flags.c_contiguous
    _check(lib.sph_update_particles(C.byref(params) if params is not None else None, pos.ctypes.data,
                                    vel.ctypes.data, pos.size // 3))


def one_call_mode(mode):
    """0: every update_particles uploads; 1: a call with the same arrays as the last one skips the upload.  Returns
    the previous mode."""
    return int(load().sph_one_call_mode(int(mode)))


def host_register(array):
    """Page-lock a numpy array in place (direct DMA for uploads and downloads from it)."""
    _check(load().sph_host_register(C.c_void_p(array.ctypes.data), array.nbytes))


def host_unregister(array):
    _check(load().sph_host_unregister(C.c_void_p(array.ctypes.data)))


def update_particles_noalloc(d_pos_ptr, d_vel_ptr, n, params=None, stream_ptr=None):
    """updateParticles on DEVICE arrays, in place, without any allocation; asynchronous on `stream_ptr`."""
    lib = load()
    _check(lib.sph_update_particles_noalloc(C.byref(params) if params is not None else None, C.c_void_p(d_pos_ptr),
                                            C.c_void_p(d_vel_ptr), int(n), C.c_void_p(stream_ptr)))


def fill_random_block(n, params=None, seed=1):
    lib = load()
    params = params if params is not None else default_params()
    pos = np.empty((n, 3), np.float32)
    vel = np.empty((n, 3), np.float32)
    _check(lib.sph_fill_random_block(C.byref(params), seed, pos.ctypes.data, vel.ctypes.data, n))
    return pos, vel
