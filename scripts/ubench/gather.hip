// Microbenchmark (development aid): cost of a wave64 global_load_dwordx4 on gfx950 as a function of how the 64 lane
// addresses are spread, and of a ds_read_b128 with random lane addresses.  Prints CU-cycles per wave instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIters = 2000;
constexpr int kLoads = 4;  // independent loads per iteration

__device__ inline unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int PATTERN>
__device__ inline unsigned lane_index(unsigned r, int lane, int u, unsigned mask) {
  switch (PATTERN) {
    case 0: return (r + u * 131u) & mask;                                        // one address for the wave
    case 1: return (r + (lane / 13) * 977u + u * 131u) & mask;                    // 5 groups of 13 lanes, one address each
    case 2: return (r + lane * 8u * 37u + u * 8u * 4099u) & mask;                 // 64 lanes, 64 different lines
    case 3: return ((r + u * 131u) * 64u + lane) & mask;                          // contiguous 1 KB
    case 4: return ((r + (lane / 4) * 1031u + u * 131u) * 4u + (lane & 3)) & mask;  // quads: 64 contiguous bytes each
    case 5: return ((r + (lane / 4) * 1031u + u * 131u) * 8u + (lane & 3) * 2u) & mask;  // 4 lanes per line, 32-B stride
    case 6: return ((r + (lane / 2) * 1031u + u * 131u) * 8u + (lane & 1) * 2u) & mask;  // 2 lanes per line
    case 7: return ((r + (lane / 13) * 977u) * 4u + u) & mask;                    // search-like: group shares 4 consecutive
    case 8: return ((r + (lane / 3) * 977u) * 4u + u) & mask;                     // same, groups of 3 lanes (rest density)
    case 9: return (mix(r + lane * 7919u + u) ) & mask;                            // fully random 16-B words
  }
  return 0;
}

template <int PATTERN>
__global__ void __launch_bounds__(256) k_gather(const float4* __restrict__ table, unsigned mask, float* out) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  float acc = 0.0f;
  for (int it = 0; it < kIters; it++) {
    const unsigned r = mix(wave * 7331u + it);
    float4 v[kLoads];
#pragma unroll
    for (int u = 0; u < kLoads; u++) v[u] = table[lane_index<PATTERN>(r, lane, u, mask)];
#pragma unroll
    for (int u = 0; u < kLoads; u++) acc += v[u].x + v[u].w;
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int PATTERN>
__global__ void __launch_bounds__(256) k_lds(const float4* __restrict__ table, float* out) {
  __shared__ float4 tile[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) tile[i] = table[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  float acc = 0.0f;
  for (int it = 0; it < kIters; it++) {
    const unsigned r = mix(wave * 7331u + it);
    float4 v[kLoads];
#pragma unroll
    for (int u = 0; u < kLoads; u++) v[u] = tile[lane_index<PATTERN>(r, lane, u, 2047u)];
#pragma unroll
    for (int u = 0; u < kLoads; u++) acc += v[u].x + v[u].w;
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <class F>
double time_ms(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  const int blocks = 256 * 8, threads = 256;
  const double wave_instr = (double)blocks * (threads / 64) * kIters * kLoads;
  float* out; hipMalloc(&out, 4);
  const char* names[] = {"one address per wave", "5 groups of 13 lanes", "64 different lines", "contiguous 1 KB",
                         "quads of 64 contiguous B", "4 lanes per line (32-B stride)", "2 lanes per line",
                         "groups of 13, 4 consecutive", "groups of 3, 4 consecutive", "random 16-B words"};
  for (unsigned entries : {1024u, 131072u, 4194304u}) {
    float4* table; hipMalloc(&table, (size_t)entries * 16);
    hipMemset(table, 0, (size_t)entries * 16);
    std::printf("table %u KB\n", entries * 16 / 1024);
    double ms[10];
    ms[0] = time_ms([&] { hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, out); });
    ms[1] = time_ms([&] { hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, out); });
    ms[2] = time_ms([&] { hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, out); });
    ms[3] = time_ms([&] { hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, out); });
    ms[4] = time_ms([&] { hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, out); });
    ms[5] = time_ms([&] { hipLaunchKernelGGL(k_gather<5>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, out); });
    ms[6] = time_ms([&] { hipLaunchKernelGGL(k_gather<6>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, out); });
    ms[7] = time_ms([&] { hipLaunchKernelGGL(k_gather<7>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, out); });
    ms[8] = time_ms([&] { hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, out); });
    ms[9] = time_ms([&] { hipLaunchKernelGGL(k_gather<9>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, out); });
    for (int p = 0; p < 10; p++)
      std::printf("  global  %-32s %8.3f ms  %7.1f CU-cycles per wave load\n", names[p], ms[p],
                  ms[p] * 1e-3 * 2.4e9 * 256 / wave_instr);
    std::fflush(stdout);
    if (entries == 1024u) {
      double l0 = time_ms([&] { hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(threads), 0, 0, table, out); });
      double l9 = time_ms([&] { hipLaunchKernelGGL(k_lds<9>, dim3(blocks), dim3(threads), 0, 0, table, out); });
      double l3 = time_ms([&] { hipLaunchKernelGGL(k_lds<3>, dim3(blocks), dim3(threads), 0, 0, table, out); });
      double l5 = time_ms([&] { hipLaunchKernelGGL(k_lds<5>, dim3(blocks), dim3(threads), 0, 0, table, out); });
      std::printf("  LDS     one address %7.1f | random %7.1f | contiguous %7.1f | 32-B stride groups %7.1f CU-cycles per wave ds_read_b128\n",
                  l0 * 1e-3 * 2.4e9 * 256 / wave_instr, l9 * 1e-3 * 2.4e9 * 256 / wave_instr,
                  l3 * 1e-3 * 2.4e9 * 256 / wave_instr, l5 * 1e-3 * 2.4e9 * 256 / wave_instr);
    }
    hipFree(table);
  }
  return 0;
}
