"""The drop-in boundary: libsph_compute.so must look like the reference's compute.o to its callers
(reference compute/compute.h:4-8, main.cpp:12,27,33, render.cpp:133)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT, bits_equal, max_rel

from sph_amd import dumpio, scenes
from sph_amd.params import default_params

SHIM = os.path.join(ROOT, "sph_amd", "libsph_compute.so")
HEADLESS = os.path.join(ROOT, "sph_amd", "sph_headless")
REF_HEADER_DIR = "/root/reference/compute"

MANGLED = {  # Itanium names of the reference's three C++-linkage entry points
    "initParticles": "_Z13initParticlesPPfS0_i",
    "updateParticles": "_Z15updateParticlesPfS_i",
    "vecNormalize": "_Z12vecNormalizePKfPf",
}
GLOBALS = {"soundSpeed": 1.0, "m": 0.00020543, "rho0": 600.0, "dynVisc": 0.2, "timestep": 0.005,
           "tankSize": 0.2154, "surfTension": 0.0728, "h": 0.01}


@pytest.fixture(scope="module")
def shim():
    import sph_amd.capi as capi
    capi.load()  # one HIP runtime per process (see capi.load)
    return C.CDLL(SHIM)


def test_exports_reference_symbols(shim):
    for name in MANGLED.values():
        assert hasattr(shim, name), name
    for name, value in GLOBALS.items():
        assert C.c_float.in_dll(shim, name).value == np.float32(value), name
    g = (C.c_float * 3).in_dll(shim, "a_gravity")
    assert tuple(g) == (0.0, -10.0, 0.0)
    for name in ("h2", "h6", "h9"):  # zero until initParticles runs, like the reference (compute.cpp:18-20)
        C.c_float.in_dll(shim, name)


def test_vec_normalize_matches_reference(shim):
    fn = getattr(shim, MANGLED["vecNormalize"])
    with np.load(os.path.join(GOLDEN_DIR, "kernels.npz")) as z:
        for a, want in zip(z["vecnorm_in"], z["vecnorm_out"]):
            a = np.ascontiguousarray(a, np.float32)
            b = np.empty(3, np.float32)
            fn(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
            assert bits_equal(b, want)
            fn(a.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p))  # in place, as compute.cpp:174 does
            assert bits_equal(a, want)


@pytest.mark.skipif(not os.path.isdir(REF_HEADER_DIR) or shutil.which("g++") is None,
                    reason="needs the reference checkout (build container only)")
def test_links_against_the_reference_header(tmp_path):
    """A translation unit that includes the REFERENCE's own compute.h links against the shim."""
    src = tmp_path / "caller.cpp"
    src.write_text(
        '#include <cstdio>\n#include "compute.h"\nextern float tankSize;\n'
        "int main() { float a[3] = {3, 4, 0}, b[3]; vecNormalize(a, b);\n"
        "  void (*u)(float*, float*, int) = updateParticles; void (*i)(float**, float**, int) = initParticles;\n"
        '  std::printf("%g %g %g %g %d\\n", b[0], b[1], b[2], tankSize, u != nullptr && i != nullptr); return 0; }\n')
    exe = tmp_path / "caller"
    libdir = os.path.join(ROOT, "sph_amd")
    subprocess.run(["g++", "-std=c++11", f"-I{REF_HEADER_DIR}", str(src), "-o", str(exe), f"-L{libdir}",
                    "-lsph_compute", "-lsph_amd", f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["0.6", "0.8", "0", "0.2154", "1"]


@pytest.mark.gpu
def test_update_particles_through_the_shim(shim, oracle):
    init = getattr(shim, MANGLED["initParticles"])
    upd = getattr(shim, MANGLED["updateParticles"])
    os.environ["SPH_AMD_SEED"] = "42"
    pp, vv = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
    n = 3000  # main.cpp:8
    init(C.byref(pp), C.byref(vv), n)
    pos = np.ctypeslib.as_array(pp, shape=(n, 3))
    vel = np.ctypeslib.as_array(vv, shape=(n, 3))
    assert pos.min() >= 0 and pos.max() <= np.float32(0.2154) / 2 and np.all(vel == 0)
    assert C.c_float.in_dll(shim, "h2").value == np.float32(0.01) * np.float32(0.01)
    p = default_params()
    po, vo = pos.copy(), vel.copy()
    for _ in range(3):
        upd(pp, vv, n)
        oracle.step_allpairs(p, po, vo)
    assert max_rel(pos, po) <= 3e-5 and max_rel(vel, vo) <= 1e-4
    # the reference's globals are the configuration API: tilt gravity and step again
    g = (C.c_float * 3).in_dll(shim, "a_gravity")
    g[0], g[1] = 5.0, -8.0
    q = default_params()
    q.gravity[0], q.gravity[1] = 5.0, -8.0
    upd(pp, vv, n)
    oracle.step_allpairs(q, po, vo)
    assert max_rel(pos, po) <= 5e-5
    g[0], g[1] = 0.0, -10.0
    libc = C.CDLL(None)
    libc.free(pp)  # main.cpp:37-39 frees with free()
    libc.free(vv)


@pytest.mark.gpu
def test_headless_driver_modes_and_resume(tmp_path, oracle):
    """Fixed-step loop, dumps and resume; `--exact` makes the drop-in path bit-identical to the
    reference algorithm, so shim mode, resident mode and the oracle agree to the last bit."""
    def run(*args):
        r = subprocess.run([HEADLESS, *args], capture_output=True, text=True, cwd=tmp_path, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        return r.stdout

    run("--n", "1500", "--seed", "9", "--steps", "6", "--mode", "resident", "--exact", "--dump-every", "3", "--out", "res")
    run("--n", "1500", "--seed", "9", "--steps", "6", "--mode", "shim", "--exact", "--out", "shim")
    pr, s3, pos3, vel3 = dumpio.read_dump(tmp_path / "res_000003.sphd")
    _, s6, pos6, vel6 = dumpio.read_dump(tmp_path / "res_final.sphd")
    _, _, spos6, svel6 = dumpio.read_dump(tmp_path / "shim_final.sphd")
    assert (s3, s6) == (3, 6) and pos6.shape == (1500, 3)
    assert bits_equal(pos6, spos6) and bits_equal(vel6, svel6)
    # resume from the step-3 dump and land on the same state
    run("--resume", "res_000003.sphd", "--steps", "3", "--mode", "resident", "--exact", "--out", "cont")
    _, s, cpos, cvel = dumpio.read_dump(tmp_path / "cont_final.sphd")
    assert s == 6 and bits_equal(cpos, pos6) and bits_equal(cvel, vel6)
    # and the oracle, started from the step-3 dump, agrees bit for bit
    p = default_params()
    po, vo = pos3.copy(), vel3.copy()
    for _ in range(3):
        oracle.step_allpairs(p, po, vo)
    assert bits_equal(po, pos6) and bits_equal(vo, vel6)
    # python writer round-trips through the C++ reader
    dumpio.write_dump(tmp_path / "py.sphd", p, 0, pos3, vel3)
    run("--resume", "py.sphd", "--steps", "3", "--mode", "resident", "--exact", "--out", "py")
    _, _, ppos, _ = dumpio.read_dump(tmp_path / "py_final.sphd")
    assert bits_equal(ppos, pos6)


@pytest.mark.gpu
def test_headless_sloshing_scene(tmp_path, oracle):
    """--scene slosh turns the gravity vector every step (the reference's mutable a_gravity, compute.cpp:12): resident
    mode (sph_set_params) and shim mode (the a_gravity global) agree with each other and with the oracle bit for bit."""
    from sph_amd import scenes

    def run(*args):
        r = subprocess.run([HEADLESS, *args], capture_output=True, text=True, cwd=tmp_path, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout

    run("--scene", "slosh", "--side", "11", "--steps", "5", "--mode", "resident", "--exact", "--dump-every", "1", "--out", "res")
    run("--scene", "slosh", "--side", "11", "--steps", "5", "--mode", "shim", "--exact", "--out", "shim")
    p, s1, pos1, vel1 = dumpio.read_dump(tmp_path / "res_000001.sphd")
    pf, s5, pos5, vel5 = dumpio.read_dump(tmp_path / "res_final.sphd")
    _, _, spos5, svel5 = dumpio.read_dump(tmp_path / "shim_final.sphd")
    assert (s1, s5) == (1, 5) and bits_equal(pos5, spos5) and bits_equal(vel5, svel5)
    po, vo = pos1.copy(), vel1.copy()
    for step in range(1, 5):  # steps 2..5 of the run use the gravity of step indices 1..4
        q = p.copy()
        g = scenes.sloshing_gravity(step)
        q.gravity[0], q.gravity[1], q.gravity[2] = float(g[0]), float(g[1]), float(g[2])
        oracle.step_allpairs(q, po, vo)
    assert bits_equal(po, pos5) and bits_equal(vo, vel5)
    assert abs(pf.gravity[0]) > 0.0  # the dump records the turned vector


@pytest.mark.gpu
def test_headless_writes_pictures(tmp_path):
    """--frames-every: display()'s counterpart (render.cpp:240-260) writes a PPM drawn from the resident state; it is the
    picture sph_render_frame gives for the dumped state."""
    import numpy as np
    from sph_amd import capi

    r = subprocess.run([HEADLESS, "--scene", "dam", "--side", "12", "--steps", "4", "--mode", "resident", "--frames-every", "2",
                        "--frame-size", "160x120", "--camera-z", "-0.8", "--dump-every", "4", "--out", "pic"],
                       capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    raw = (tmp_path / "pic_000004.ppm").read_bytes()
    assert raw.startswith(b"P6\n160 120\n255\n") and (tmp_path / "pic_000002.ppm").exists()
    img = np.frombuffer(raw[len(b"P6\n160 120\n255\n"):], dtype=np.uint8).reshape(120, 160, 3)
    p, step, pos, vel = dumpio.read_dump(tmp_path / "pic_000004.sphd")
    with capi.Simulation(p, n_max=pos.shape[0]) as sim:
        sim.upload(pos, vel)
        cam = capi.default_camera()
        cam.translation[2] = -0.8
        again = sim.render(160, 120, cam)
    assert (img != 255).any() and np.array_equal(img, again)
    # shim mode keeps the state on the host: no pictures there
    r = subprocess.run([HEADLESS, "--n", "100", "--steps", "1", "--mode", "shim", "--frames-every", "1"], capture_output=True,
                       text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 2


@pytest.mark.gpu
@pytest.mark.parametrize("scene,side,slabs", [("dam", 40, 4), ("slosh", 36, 3)])
def test_headless_slabs_match_the_single_context(tmp_path, scene, side, slabs):
    """--slabs K: K slab contexts driven from C++ alone (sph_column_of, sph_slab_plan_partition, sph_slab_upload,
    sph_slab_attach_local, sph_slab_step_begin/_end, sph_slab_load_work -> sph_slab_plan_next -> sph_slab_set_next_columns)
    give the single context's dumps bit for bit, mid-run and at the end, with the cut planes moving; and a run resumed
    from a dump of the slab run continues to the same bits."""
    import json

    def run(*args):
        r = subprocess.run([HEADLESS, *args], capture_output=True, text=True, cwd=tmp_path, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        return json.loads(r.stdout.strip().splitlines()[-1])

    common = ["--scene", scene, "--side", str(side), "--steps", "40", "--mode", "resident", "--dump-every", "20"]
    one = run(*common, "--out", "one")
    many = run(*common, "--slabs", str(slabs), "--rebalance-every", "2", "--out", "many")
    assert many["slabs"] == slabs and many["lost"] == 0 and many["missed"] == 0 and many["rebalances"] == 20
    assert sum(many["own"]) == side ** 3 == one["n"]
    for name in ("000020", "final"):
        _, s1, p1, v1 = dumpio.read_dump(tmp_path / f"one_{name}.sphd")
        _, s2, p2, v2 = dumpio.read_dump(tmp_path / f"many_{name}.sphd")
        assert s1 == s2 and bits_equal(p1, p2) and bits_equal(v1, v2), name
    cont = run("--resume", "many_000020.sphd", "--scene", scene, "--steps", "20", "--mode", "resident", "--slabs", str(slabs),
               "--out", "cont")
    _, s3, p3, v3 = dumpio.read_dump(tmp_path / "cont_final.sphd")
    assert s3 == 40 and bits_equal(p3, p2) and bits_equal(v3, v2) and cont["lost"] == 0
    # the collapsing dam drags the cut planes away from where the resting block put them (40 steps of a 15-degree swing
    # do not)
    start = run("--scene", scene, "--side", str(side), "--steps", "1", "--mode", "resident", "--slabs", str(slabs),
                "--rebalance-every", "0", "--out", "start")
    assert start["rebalances"] == 0 and (start["cuts"] != many["cuts"] or scene != "dam")


@pytest.mark.gpu
def test_headless_slabs_refuse_what_they_cannot_do(tmp_path):
    r = subprocess.run([HEADLESS, "--n", "500", "--steps", "1", "--mode", "shim", "--slabs", "2"], capture_output=True, text=True,
                       cwd=tmp_path, timeout=300)
    assert r.returncode == 2
    # more slabs than the grid has room for (four columns each)
    r = subprocess.run([HEADLESS, "--n", "500", "--steps", "1", "--mode", "resident", "--slabs", "500"], capture_output=True,
                       text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 1 and "cannot hold" in r.stderr
