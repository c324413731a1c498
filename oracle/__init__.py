"""TEST INFRASTRUCTURE — Python bindings of the CPU oracle (never imported by sph_amd/).

``Oracle``  wraps oracle/libsph_oracle.so: this repository's CPU restatement of the reference's
            updateParticles (reference compute/compute.cpp:120-320).
``Ref``     wraps oracle/_ref/libsph_ref.so: the reference's own compute.cpp, compiled unmodified
            by oracle/Makefile, with its temporaries tapped (oracle/ref_tap.cpp).  Present only
            where that build has been run; ``Ref.available()`` says so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import ctypes as C
import os

import numpy as np

from sph_amd.params import SphParams, default_params  # parameter struct only

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_I = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class OracleConsts(C.Structure):
    _fields_ = [
        ("h2", C.c_float), ("h6", C.c_float), ("h9", C.c_float),
        ("c_poly6", C.c_double), ("c_spiky_grad", C.c_double), ("c_visc_lapl", C.c_double),
        ("c_poly6_grad", C.c_double), ("c_poly6_lapl", C.c_double),
        ("r2_max", C.c_float), ("cell_size", C.c_float), ("ncell", C.c_int),
    ]


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    PATH = os.path.join(_HERE, "libsph_oracle.so")

    def __init__(self):
        if not os.path.exists(self.PATH):
            raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
        L = self.lib = C.CDLL(self.PATH)
        P = C.POINTER(SphParams)
        L.sph_oracle_derive.argtypes = [P, C.POINTER(OracleConsts)]
        for name in ("lapl_viscosity", "grad_spiky", "w_poly6", "grad_poly6", "lapl_poly6"):
            f = getattr(L, "sph_oracle_" + name)
            f.argtypes = [P, C.c_float]
            f.restype = C.c_float
        L.sph_oracle_vec_normalize.argtypes = [_F, _F]
        L.sph_oracle_step_allpairs.argtypes = [P, _F, _F, C.c_int, C.c_void_p, C.c_void_p]
        L.sph_oracle_neighbor_counts.argtypes = [P, _F, C.c_int, _I]
        L.sph_oracle_cell_ids.argtypes = [P, _F, C.c_int, _I]
        L.sph_oracle_step_grid.argtypes = [P, _F, _F, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
        L.sph_oracle_num_threads.restype = C.c_int
        L.sph_oracle_set_num_threads.argtypes = [C.c_int]
        L.sph_oracle_set_num_threads.restype = None

    def consts(self, params):
        k = OracleConsts()
        self.lib.sph_oracle_derive(C.byref(params), C.byref(k))
        return k

    def kernel(self, name, params, r):
        return float(getattr(self.lib, "sph_oracle_" + name)(C.byref(params), C.c_float(r)))

    def vec_normalize(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.empty(3, dtype=np.float32)
        self.lib.sph_oracle_vec_normalize(a, b)
        return b

    def step_allpairs(self, params, pos, vel, want_debug=False):
        """In-place all-pairs step on float32 [n,3] arrays; returns (rho, force) if asked."""
        n = pos.shape[0]
        rho = np.empty(n, np.float32) if want_debug else None
        force = np.empty((n, 3), np.float32) if want_debug else None
        self.lib.sph_oracle_step_allpairs(C.byref(params), pos.reshape(-1), vel.reshape(-1), n,
                                          _opt(rho), _opt(force))
        return rho, force

    def step_grid(self, params, pos, vel, ids=None, order=0, want_debug=False):
        n = pos.shape[0]
        rho = np.empty(n, np.float32) if want_debug else None
        force = np.empty((n, 3), np.float32) if want_debug else None
        ncount = np.empty(n, np.int32) if want_debug else None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.int32)
        self.lib.sph_oracle_step_grid(C.byref(params), pos.reshape(-1), vel.reshape(-1), _opt(ids),
                                      n, order, _opt(rho), _opt(force), _opt(ncount))
        return rho, force, ncount

    def neighbor_counts(self, params, pos):
        out = np.empty(pos.shape[0], np.int32)
        self.lib.sph_oracle_neighbor_counts(C.byref(params), pos.reshape(-1), pos.shape[0], out)
        return out

    def cell_ids(self, params, pos):
        out = np.empty(pos.shape[0], np.int32)
        self.lib.sph_oracle_cell_ids(C.byref(params), pos.reshape(-1), pos.shape[0], out)
        return out

    def num_threads(self):
        return int(self.lib.sph_oracle_num_threads())

    def set_num_threads(self, n):
        """OpenMP threads of this process from now on (the reference build shares the runtime)."""
        self.lib.sph_oracle_set_num_threads(int(n))


class Ref:
    """The reference's own compute.cpp (compiled, unmodified) behind oracle/ref_tap.cpp."""
    PATH = os.path.join(_HERE, "_ref", "libsph_ref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self, params=None):
        L = self.lib = C.CDLL(self.PATH)
        L.ref_set_params.argtypes = [C.c_float] * 4 + [_F] + [C.c_float] * 4
        L.ref_get_consts.argtypes = [_F]
        L.ref_step.argtypes = [_F, _F, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_run.argtypes = [_F, _F, C.c_int, C.c_int]
        for name in ("lapl_viscosity", "grad_spiky", "w_poly6", "grad_poly6", "lapl_poly6"):
            f = getattr(L, "ref_" + name)
            f.argtypes = [C.c_float]
            f.restype = C.c_float
        L.ref_vec_normalize.argtypes = [_F, _F]
        self.set_params(params if params is not None else default_params())

    def set_params(self, p):
        g = np.array([p.gravity[0], p.gravity[1], p.gravity[2]], dtype=np.float32)
        self.lib.ref_set_params(p.sound_speed, p.mass, p.rho0, p.dyn_visc, g, p.timestep,
                                p.tank_size, p.surf_tension, p.h)

    def consts(self):
        out = np.empty(3, np.float32)
        self.lib.ref_get_consts(out)
        return out

    def kernel(self, name, r):
        return float(getattr(self.lib, "ref_" + name)(C.c_float(r)))

    def vec_normalize(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.empty(3, dtype=np.float32)
        self.lib.ref_vec_normalize(a, b)
        return b

    def step(self, pos, vel, want_debug=False):
        n = pos.shape[0]
        rho = np.empty(n, np.float32) if want_debug else None
        force = np.empty((n, 3), np.float32) if want_debug else None
        self.lib.ref_step(pos.reshape(-1), vel.reshape(-1), n, _opt(rho), _opt(force))
        return rho, force

    def run(self, pos, vel, steps):
        self.lib.ref_run(pos.reshape(-1), vel.reshape(-1), pos.shape[0], steps)
