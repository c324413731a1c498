"""ctypes mirror of ``sph_params`` / ``sph_timings`` from include/sph_amd.h.

Field order, names and defaults follow the reference's mutable globals
(reference compute/compute.cpp:8-17).
"""
import ctypes as C

SPH_FLAG_SURFACE_TENSION = 1 << 0
SPH_FLAG_EXACT_ORDER = 1 << 1
SPH_FLAG_KEEP_DEBUG = 1 << 2
SPH_FLAG_LANE_SWEEP = 1 << 3
SPH_FLAG_REFERENCE_ARITH = 1 << 4


class SphParams(C.Structure):
    _fields_ = [
        ("sound_speed", C.c_float),
        ("mass", C.c_float),
        ("rho0", C.c_float),
        ("dyn_visc", C.c_float),
        ("gravity", C.c_float * 3),
        ("timestep", C.c_float),
        ("tank_size", C.c_float),
        ("surf_tension", C.c_float),
        ("h", C.c_float),
        ("cell_size", C.c_float),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]

    def copy(self):
        out = SphParams()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(SphParams))
        return out


class SphTimings(C.Structure):
    _fields_ = [
        ("hash_ms", C.c_double),
        ("scan_ms", C.c_double),
        ("reorder_ms", C.c_double),
        ("search_ms", C.c_double),
        ("density_ms", C.c_double),
        ("force_ms", C.c_double),
        ("other_ms", C.c_double),
        ("steps", C.c_uint64),
        ("particles", C.c_uint64),
    ]


def default_params(tank_size=None, flags=SPH_FLAG_SURFACE_TENSION):
    """Reference defaults (compute.cpp:8-17); pure Python so it needs no library."""
    p = SphParams()
    p.sound_speed = 1.0
    p.mass = 0.00020543
    p.rho0 = 600.0
    p.dyn_visc = 0.2
    p.gravity[0], p.gravity[1], p.gravity[2] = 0.0, -10.0, 0.0
    p.timestep = 0.005
    p.tank_size = 0.2154 if tank_size is None else tank_size
    p.surf_tension = 0.0728
    p.h = 0.01
    p.cell_size = 0.0
    p.flags = flags
    p.reserved = 0
    return p
