"""Reader/writer of the headless driver's particle dumps (.sphd; format in csrc/sph_headless.cpp).

The reference has no on-disk format (its state is two host arrays, reference main.cpp:9-10);
this one doubles as checkpoint and as the hand-off to an offline viewer."""
import ctypes as C

import numpy as np

from .params import SphParams

MAGIC = b"SPHD"


def write_dump(path, params, step, pos, vel):
    pos = np.ascontiguousarray(pos, np.float32)
    vel = np.ascontiguousarray(vel, np.float32)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(np.array([1, pos.size // 3, step], np.uint32).tobytes())
        f.write(bytes(params))
        f.write(pos.tobytes())
        f.write(vel.tobytes())


def read_dump(path):
    with open(path, "rb") as f:
        if f.read(4) != MAGIC:
            raise ValueError(f"{path}: not an SPHD dump")
        version, n, step = np.frombuffer(f.read(12), np.uint32)
        if version != 1:
            raise ValueError(f"{path}: unsupported dump version {version}")
        params = SphParams.from_buffer_copy(f.read(C.sizeof(SphParams)))
        pos = np.frombuffer(f.read(12 * int(n)), np.float32).reshape(-1, 3).copy()
        vel = np.frombuffer(f.read(12 * int(n)), np.float32).reshape(-1, 3).copy()
    if pos.shape[0] != n or vel.shape[0] != n:
        raise ValueError(f"{path}: truncated dump")
    return params, int(step), pos, vel
