"""CPU restatement of the reference's picture of a particle set — TEST INFRASTRUCTURE, never on the product path.

Parity UNPINNED: the reference draws with OpenGL (render/render.cpp:240-260), which cannot run here, and holds no
image fixtures.  This file restates the reference's camera and shader arithmetic per pixel:
  model matrix   translate(modelTranslation) * build_rotmatrix(curquat)   render.cpp:164-171, trackball.cpp:326-346
  view matrix    identity                                                 render.cpp:163
  projection     glm::perspective(fov 45, aspect, 0.01, 10)               render.cpp:155-162
  geometry       sphere of radius 0.0025 per particle                     render.cpp:187,253
  colour         normalize(vec3(gl_InstanceID / 2000, 1, 1))              shaders/vs_basic.glsl
  light          at (10, 10, 10); N = model-space normal (normal matrix without the rotation, render.cpp:167)
  material       ka 0.5, kd 0.45 max(0, N.L), ks 0.5 max(0, N.H)^128, H = normalize(L + (0, 0, 1))   fs_simple.glsl
  background     white                                                    render.cpp:297
with the sphere as a per-pixel impostor (pixel centre inside the projected disc of radius focal * r / depth).
Pure numpy / Python loops: small scenes only.
"""
import numpy as np

f32 = np.float32


def rotation(q):
    """M[row][col] of build_rotmatrix(glm::mat4&, q): glm indexes m[col][row] (trackball.cpp:326-346)."""
    q = [f32(v) for v in q]
    one, two = f32(1), f32(2)
    m = np.zeros((3, 3), dtype=np.float32)
    m[0][0] = one - two * (q[1] * q[1] + q[2] * q[2])
    m[1][0] = two * (q[0] * q[1] - q[2] * q[3])
    m[2][0] = two * (q[2] * q[0] + q[1] * q[3])
    m[0][1] = two * (q[0] * q[1] + q[2] * q[3])
    m[1][1] = one - two * (q[2] * q[2] + q[0] * q[0])
    m[2][1] = two * (q[1] * q[2] - q[0] * q[3])
    m[0][2] = two * (q[2] * q[0] - q[1] * q[3])
    m[1][2] = two * (q[1] * q[2] + q[0] * q[3])
    m[2][2] = one - two * (q[1] * q[1] + q[0] * q[0])
    return m


def _norm(v):
    return v / np.sqrt((v * v).sum(dtype=np.float32), dtype=np.float32)


def render(pos, ids, width, height, translation=(0.0, 0.0, -1.5), quat=(0.0, 0.0, 0.0, 1.0), fov_deg=45.0,
           z_near=0.01, z_far=10.0, radius=0.0025):
    """uint8 [height, width, 3] picture and float32 [height, width] depth (inf where empty)."""
    pos = np.asarray(pos, dtype=np.float32)
    M = rotation(quat)
    t = np.asarray(translation, dtype=np.float32)
    focal = f32(1.0 / np.tan(0.5 * np.radians(fov_deg)))
    aspect = f32(width) / f32(height)
    sx, sy = focal / aspect * f32(0.5) * f32(width), focal * f32(0.5) * f32(height)
    r = f32(radius)
    img = np.full((height, width, 3), 255, dtype=np.uint8)
    depth = np.full((height, width), np.inf, dtype=np.float32)
    light = np.array([10.0, 10.0, 10.0], dtype=np.float32)
    for p, pid in zip(pos, ids):
        v = M @ p + t
        zc = -v[2]
        if not (zc - r > z_near) or not (zc + r < z_far):
            continue
        px = f32(0.5) * f32(width) + sx * v[0] / zc
        py = f32(0.5) * f32(height) - sy * v[1] / zc
        rp = sy * r / zc
        x0, x1 = max(int(np.floor(px - rp)), 0), min(int(np.ceil(px + rp)), width - 1)
        y0, y1 = max(int(np.floor(py - rp)), 0), min(int(np.ceil(py + rp)), height - 1)
        tint = _norm(np.array([f32(pid) / f32(2000.0), 1.0, 1.0], dtype=np.float32))
        for iy in range(y0, y1 + 1):
            for ix in range(x0, x1 + 1):
                u = (f32(ix) + f32(0.5) - px) / rp
                w = -(f32(iy) + f32(0.5) - py) / rp
                d2 = u * u + w * w
                if d2 > 1.0:
                    continue
                nz = np.sqrt(f32(1.0) - d2, dtype=np.float32)
                d = zc - r * nz
                if not d < depth[iy, ix]:
                    continue
                n_view = np.array([u, w, nz], dtype=np.float32)
                world = v + r * n_view
                L = _norm(light - world)
                H = _norm(L + np.array([0.0, 0.0, 1.0], dtype=np.float32))
                N = M.T @ n_view
                diff = f32(0.5) + f32(0.45) * max(f32(0.0), f32(L @ N))
                spec = f32(0.5) * f32(max(f32(0.0), f32(H @ N)) ** 128.0)
                c = np.minimum(f32(1.0), diff * tint + spec)
                img[iy, ix] = (c * f32(255.0) + f32(0.5)).astype(np.uint8)
                depth[iy, ix] = d
    return img, depth
