"""Viewer hand-off: frames rendered from the state in HBM against the CPU restatement of the reference's camera
and shaders (oracle/render_ref.py; parity unpinned: the reference's OpenGL picture cannot be produced here)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import render_ref  # noqa: E402
from sph_amd import capi, scenes  # noqa: E402

pytestmark = pytest.mark.gpu


def _scene(n=400, seed=3):
    rng = np.random.default_rng(seed)
    p = capi.default_params()
    pos = (rng.random((n, 3), dtype=np.float32) * np.float32(0.9) + np.float32(0.05)) * np.float32(p.tank_size)
    vel = np.zeros_like(pos)
    return p, pos, vel


def _compare(img, ref, depth):
    assert img.shape == ref.shape and img.dtype == np.uint8
    covered = np.isfinite(depth)
    assert covered.sum() > 50  # the picture shows something
    differs = np.abs(img.astype(np.int32) - ref.astype(np.int32)).max(axis=2) > 2
    # pixels may flip only where a pixel centre lies on a silhouette or two surfaces tie in depth (last-bit
    # differences between the device's and numpy's rounding): a small fraction of the covered pixels
    assert differs.sum() <= max(3, 0.01 * covered.sum()), (int(differs.sum()), int(covered.sum()))
    assert (img[~covered & ~differs] == 255).all()  # background is the reference's clear colour


def test_default_camera_is_the_references():
    cam = capi.default_camera()
    assert tuple(cam.translation) == (0.0, 0.0, -1.5)
    assert tuple(cam.quat) == (0.0, 0.0, 0.0, 1.0)
    assert (cam.fov_deg, cam.z_near, cam.z_far) == (45.0, np.float32(0.01), 10.0)
    assert cam.radius == np.float32(0.0025)


def test_frame_matches_cpu_restatement_default_camera():
    p, pos, vel = _scene()
    with capi.Simulation(p, n_max=pos.shape[0]) as sim:
        sim.upload(pos, vel)
        cam = capi.default_camera()
        cam.translation[2] = -0.6  # closer than the reference's -1.5 so that spheres cover several pixels
        img = sim.render(320, 240, cam)
    ref, depth = render_ref.render(pos, np.arange(pos.shape[0]), 320, 240, translation=(0.0, 0.0, -0.6))
    _compare(img, ref, depth)


def test_frame_with_rotated_camera_after_steps():
    p, pos, vel = _scene(300, seed=5)
    quat = np.array([0.2, -0.3, 0.1, 0.0], dtype=np.float32)
    quat[3] = np.sqrt(1.0 - (quat[:3] ** 2).sum())
    with capi.Simulation(p, n_max=pos.shape[0]) as sim:
        sim.upload(pos, vel)
        sim.step(3)  # the state is then in sorted order: colours must still follow the particle ids
        now, _ = sim.download()
        cam = capi.default_camera()
        cam.translation[0], cam.translation[1], cam.translation[2] = -0.05, -0.1, -0.5
        for k in range(4):
            cam.quat[k] = float(quat[k])
        img = sim.render(200, 200, cam)
    ref, depth = render_ref.render(now, np.arange(now.shape[0]), 200, 200, translation=(-0.05, -0.1, -0.5), quat=quat)
    _compare(img, ref, depth)


def test_render_rejects_bad_arguments():
    p, pos, vel = _scene(10)
    with capi.Simulation(p, n_max=10) as sim:
        with pytest.raises(capi.SphError):
            sim.render(64, 64)  # nothing uploaded
        sim.upload(pos, vel)
        with pytest.raises(capi.SphError):
            sim.render(0, 64)
        cam = capi.default_camera()
        cam.fov_deg = 0.0
        with pytest.raises(capi.SphError):
            sim.render(64, 64, cam)
        assert sim.render(8, 8).shape == (8, 8, 3)
