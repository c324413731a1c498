// Viewer hand-off: the particle state is drawn where it lives.  The reference's display() (render/render.cpp:240-260)
// copies the positions to the GPU every frame and draws one tessellated sphere per particle (radius 0.0025,
// render.cpp:187) under a perspective camera (fov 45 degrees, near 0.01, far 10, render.cpp:155-162) whose model
// matrix is translate(0, 0, -1.5) * rotation(trackball quaternion) (render.cpp:26,164-171), shaded by
// vs_basic.glsl / fs_simple.glsl.  Here the positions are already in HBM: one lane per particle splats a sphere
// impostor (per-pixel depth and normal of the sphere, no tessellation) into a depth-keyed frame buffer, and only the
// finished image leaves the device.
//
// Frame buffer: one 64-bit word per pixel, (depth as float bits) << 32 | 0x00RRGGBB, resolved by atomicMin: the
// nearest surface wins whatever the order of the lanes, so the picture is a pure function of the state.
#include "sph_internal.hpp"

namespace sph {

namespace {

constexpr unsigned long long kEmptyPixel = ~0ull;
constexpr float kMaxSplat = 512.0f;

__global__ void k_frame_clear(unsigned long long* __restrict__ fb, int pixels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < pixels) fb[i] = kEmptyPixel;
}

__device__ inline float3 normalized(float x, float y, float z) {
  const float inv = 1.0f / sqrtf(x * x + y * y + z * z);
  return make_float3(x * inv, y * inv, z * inv);
}

__global__ void k_frame_splat(State st, int n, RenderCamera cam, int width, int height,
                              unsigned long long* __restrict__ fb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = st.x[i], y = st.y[i], z = st.z[i];
  // view = model * position (the reference's view matrix is the identity, render.cpp:163)
  const float vx = cam.m[0][0] * x + cam.m[0][1] * y + cam.m[0][2] * z + cam.t[0];
  const float vy = cam.m[1][0] * x + cam.m[1][1] * y + cam.m[1][2] * z + cam.t[1];
  const float vz = cam.m[2][0] * x + cam.m[2][1] * y + cam.m[2][2] * z + cam.t[2];
  const float zc = -vz;  // distance in front of the camera
  if (!(zc - cam.radius > cam.z_near) || !(zc + cam.radius < cam.z_far)) return;
  const float sx = cam.focal / cam.aspect * 0.5f * (float)width, sy = cam.focal * 0.5f * (float)height;
  const float px = 0.5f * (float)width + sx * vx / zc;
  const float py = 0.5f * (float)height - sy * vy / zc;  // row 0 is the top of the picture
  const float rp = sy * cam.radius / zc;                 // radius in pixels
  if (!(rp > 0.0f)) return;
  // one lane draws one sphere: a sphere that fills more than kMaxSplat pixels either way (a particle at the near plane of
  // a very large frame) is drawn out to that distance from its centre only, which bounds the lane's work
  const float reach = rp < kMaxSplat ? rp : kMaxSplat;
  int x0 = (int)floorf(px - reach), x1 = (int)ceilf(px + reach), y0 = (int)floorf(py - reach), y1 = (int)ceilf(py + reach);
  x0 = x0 < 0 ? 0 : x0;
  y0 = y0 < 0 ? 0 : y0;
  x1 = x1 > width - 1 ? width - 1 : x1;
  y1 = y1 > height - 1 ? height - 1 : y1;
  // instance colour of vs_basic.glsl: normalize(vec3(id / 2000, 1, 1))
  const int id = st.id ? st.id[i] : i;
  const float3 tint = normalized((float)id / 2000.0f, 1.0f, 1.0f);
  const float inv_rp = 1.0f / rp;
  for (int iy = y0; iy <= y1; iy++) {
    for (int ix = x0; ix <= x1; ix++) {
      const float u = ((float)ix + 0.5f - px) * inv_rp, v = -((float)iy + 0.5f - py) * inv_rp;
      const float d2 = u * u + v * v;
      if (d2 > 1.0f) continue;
      const float nz = sqrtf(1.0f - d2);
      const float depth = zc - cam.radius * nz;
      // surface point and the lighting of fs_simple.glsl: light at (10, 10, 10), half vector with (0, 0, 1),
      // ka 0.5, kd 0.45, ks 0.5 with exponent 128; the normal is the model-space one (the reference's normal
      // matrix leaves the trackball rotation out, render.cpp:167)
      const float wx = vx + cam.radius * u, wy = vy + cam.radius * v, wz = vz + cam.radius * nz;
      const float3 L = normalized(10.0f - wx, 10.0f - wy, 10.0f - wz);
      const float3 H = normalized(L.x, L.y, L.z + 1.0f);
      const float Nx = cam.m[0][0] * u + cam.m[1][0] * v + cam.m[2][0] * nz;
      const float Ny = cam.m[0][1] * u + cam.m[1][1] * v + cam.m[2][1] * nz;
      const float Nz = cam.m[0][2] * u + cam.m[1][2] * v + cam.m[2][2] * nz;
      const float nl = fmaxf(0.0f, L.x * Nx + L.y * Ny + L.z * Nz);
      const float nh = fmaxf(0.0f, H.x * Nx + H.y * Ny + H.z * Nz);
      const float diff = 0.5f + 0.45f * nl, spec = 0.5f * powf(nh, 128.0f);
      const float cr = fminf(1.0f, diff * tint.x + spec), cg = fminf(1.0f, diff * tint.y + spec),
                  cb = fminf(1.0f, diff * tint.z + spec);
      const unsigned rgb = ((unsigned)(cr * 255.0f + 0.5f) << 16) | ((unsigned)(cg * 255.0f + 0.5f) << 8) |
                           (unsigned)(cb * 255.0f + 0.5f);
      const unsigned long long key = ((unsigned long long)__float_as_uint(depth) << 32) | rgb;
      atomicMin(&fb[(size_t)iy * width + ix], key);
    }
  }
}

__global__ void k_frame_resolve(const unsigned long long* __restrict__ fb, int pixels, unsigned char* __restrict__ rgb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pixels) return;
  const unsigned long long key = fb[i];
  const unsigned c = key == kEmptyPixel ? 0x00ffffffu : (unsigned)key;  // glClearColor(1, 1, 1, 1), render.cpp:297
  rgb[3 * i + 0] = (unsigned char)(c >> 16);
  rgb[3 * i + 1] = (unsigned char)(c >> 8);
  rgb[3 * i + 2] = (unsigned char)c;
}

}  // namespace

void launch_render(hipStream_t s, const State& st, int n, const RenderCamera& cam, int width, int height,
                   unsigned long long* fb, unsigned char* rgb) {
  const int pixels = width * height;
  hipLaunchKernelGGL(k_frame_clear, dim3((pixels + 255) / 256), dim3(256), 0, s, fb, pixels);
  if (n > 0) hipLaunchKernelGGL(k_frame_splat, dim3((n + 255) / 256), dim3(256), 0, s, st, n, cam, width, height, fb);
  hipLaunchKernelGGL(k_frame_resolve, dim3((pixels + 255) / 256), dim3(256), 0, s, fb, pixels, rgb);
}

}  // namespace sph
