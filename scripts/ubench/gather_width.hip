// Microbenchmark (development aid): does a gathered wave64 load cost per lane or per byte?  L1-resident table,
// loads of 4, 8, 12 and 16 bytes per lane, contiguous / random / groups-of-13 addresses.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 2000;
constexpr int kLoads = 4;
__device__ inline unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
struct F3 { float x, y, z; };
template <class T> __device__ inline float first(const T& v);
template <> __device__ inline float first<float>(const float& v) { return v; }
template <> __device__ inline float first<float2>(const float2& v) { return v.x + v.y; }
template <> __device__ inline float first<F3>(const F3& v) { return v.x + v.z; }
template <> __device__ inline float first<float4>(const float4& v) { return v.x + v.w; }

template <class T, int PATTERN>
__global__ void __launch_bounds__(256) k(const T* __restrict__ table, unsigned mask, float* out) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  float acc = 0.0f;
  for (int it = 0; it < kIters; it++) {
    const unsigned r = mix(wave * 7331u + it);
    T v[kLoads];
#pragma unroll
    for (int u = 0; u < kLoads; u++) {
      unsigned i = PATTERN == 0 ? ((r + u * 131u) * 64u + lane) : PATTERN == 1 ? mix(r + lane * 7919u + u) : ((r + (lane / 13) * 977u) * 4u + u);
      v[u] = table[i & mask];
    }
#pragma unroll
    for (int u = 0; u < kLoads; u++) acc += first(v[u]);
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <class T, int PATTERN>
double run(const void* table, unsigned entries, float* out) {
  const int blocks = 256 * 8, threads = 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<T, PATTERN>), dim3(blocks), dim3(threads), 0, 0, (const T*)table, entries - 1, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<T, PATTERN>), dim3(blocks), dim3(threads), 0, 0, (const T*)table, entries - 1, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e-3 * 2.4e9 * 256 / ((double)blocks * (threads / 64) * kIters * kLoads);
}

int main() {
  float* out; (void)hipMalloc(&out, 4);
  void* table; (void)hipMalloc(&table, 1 << 20); (void)hipMemset(table, 0, 1 << 20);
  const unsigned entries = 1024;  // entries of each type: 4-16 KB, L1-resident
  std::printf("CU-cycles per wave load      contiguous   random   groups of 13 x 4 consecutive\n");
  std::printf("  4 B per lane               %8.1f %8.1f %8.1f\n", run<float, 0>(table, entries, out), run<float, 1>(table, entries, out), run<float, 2>(table, entries, out));
  std::printf("  8 B per lane               %8.1f %8.1f %8.1f\n", run<float2, 0>(table, entries, out), run<float2, 1>(table, entries, out), run<float2, 2>(table, entries, out));
  std::printf(" 12 B per lane               %8.1f %8.1f %8.1f\n", run<F3, 0>(table, entries, out), run<F3, 1>(table, entries, out), run<F3, 2>(table, entries, out));
  std::printf(" 16 B per lane               %8.1f %8.1f %8.1f\n", run<float4, 0>(table, entries, out), run<float4, 1>(table, entries, out), run<float4, 2>(table, entries, out));
  return 0;
}
