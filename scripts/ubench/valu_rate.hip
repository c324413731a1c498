// Micro-benchmark: sustained issue rate of common fp32/integer vector instructions on one SIMD with 8 waves resident
// (256 CUs x 8 blocks of 256 threads).  Prints cycles per wave-instruction per SIMD for several instruction mixes.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  unsigned bits = 0;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // 128 independent v_fma_f32
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
    } else if (MODE == 1) {  // 128 v_mul_f32 / v_add_f32 / v_sub_f32 (VOP2)
      REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_sub_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %9\n v_sub_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
    } else if (MODE == 2) {  // 64 x (v_cmp_ge_f32 vcc ; v_cndmask_b32 reading vcc), with the two wait states filled by nops
      REP16(asm volatile("v_cmp_ge_f32 vcc, %8, %0\n s_nop 1\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_ge_f32 vcc, %8, %2\n s_nop 1\n v_cndmask_b32 %3, %3, %9, vcc\n"
                         "v_cmp_ge_f32 vcc, %8, %4\n s_nop 1\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_ge_f32 vcc, %8, %6\n s_nop 1\n v_cndmask_b32 %7, %7, %9, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");)
    } else if (MODE == 3) {  // 128 v_cmp_ge_f32 only
      REP16(asm volatile("v_cmp_ge_f32 vcc, %8, %0\n v_cmp_ge_f32 vcc, %8, %1\n v_cmp_ge_f32 vcc, %8, %2\n v_cmp_ge_f32 vcc, %8, %3\n"
                         "v_cmp_ge_f32 vcc, %8, %4\n v_cmp_ge_f32 vcc, %8, %5\n v_cmp_ge_f32 vcc, %8, %6\n v_cmp_ge_f32 vcc, %8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");)
    } else if (MODE == 4) {  // 128 v_sub_f32 with clamp (VOP3)
      REP16(asm volatile("v_sub_f32_e64 %0, %8, %0 clamp\n v_sub_f32_e64 %1, %8, %1 clamp\n v_sub_f32_e64 %2, %8, %2 clamp\n v_sub_f32_e64 %3, %8, %3 clamp\n"
                         "v_sub_f32_e64 %4, %8, %4 clamp\n v_sub_f32_e64 %5, %8, %5 clamp\n v_sub_f32_e64 %6, %8, %6 clamp\n v_sub_f32_e64 %7, %8, %7 clamp\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
    } else if (MODE == 5) {  // 128 v_mov_b32
      REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                         "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
    } else if (MODE == 6) {  // 128 v_mul_lo_u32
      REP16(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                         "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
    } else if (MODE == 7) {  // 128 v_lshl_add_u64
      unsigned long long y0 = (unsigned long long)x0, y1 = (unsigned long long)x1, s = (unsigned long long)a;
      REP16(asm volatile("v_lshl_add_u64 %0, %0, 2, %2\n v_lshl_add_u64 %1, %1, 2, %2\n v_lshl_add_u64 %0, %0, 2, %2\n v_lshl_add_u64 %1, %1, 2, %2\n"
                         "v_lshl_add_u64 %0, %0, 2, %2\n v_lshl_add_u64 %1, %1, 2, %2\n v_lshl_add_u64 %0, %0, 2, %2\n v_lshl_add_u64 %1, %1, 2, %2\n"
                         : "+v"(y0), "+v"(y1) : "v"(s));)
      x0 += (float)y0; x1 += (float)y1;
    } else if (MODE == 8) {  // 128 v_mov_b64
      double d0 = x0, d1 = x1, d2 = x2, d3 = x3;
      REP16(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n"
                         "v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
      x0 += (float)d0; x1 += (float)d1; x2 += (float)d2; x3 += (float)d3;
    } else if (MODE == 9) {  // 128 v_rsq_f32
      REP16(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                         "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
    } else if (MODE == 10) {  // 128 v_addc_co_u32 with the carry from vcc
      REP16(asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n"
                         "v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n"
                         : "+v"(bits) : : "vcc");)
    } else if (MODE == 11 || MODE == 12 || MODE == 13) {  // 128 packed fp32 operations on register pairs
      double d0 = x0, d1 = x1, d2 = x2, d3 = x3, pa = a, pb = b;
      if (MODE == 11) {
        REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                           "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                           : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(pa), "v"(pb));)
      } else if (MODE == 12) {
        REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                           "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                           : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(pa), "v"(pb));)
      } else {
        REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                           "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                           : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(pa), "v"(pb));)
      }
      x0 += (float)d0; x1 += (float)d1; x2 += (float)d2; x3 += (float)d3;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + bits;
}

template <int MODE>
void run(const char* name, int per_iter, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, iters = 2000;
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 10, 1.0001f, 0.5f);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: blocks * 4 waves * iters * per_iter / 1024 SIMDs
  const double insts = (double)blocks * 4 * iters * per_iter / 1024.0;
  printf("%-44s %d waves/SIMD  %.3f ms  %.2f cycles per instruction per SIMD (at 2.4 GHz)\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / insts);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32 (independent)", 128, w);
    run<1>("v_mul/add/sub_f32", 128, w);
  }
  run<2>("v_cmp + s_nop 1 + v_cndmask (vcc)", 128, 8);
  run<3>("v_cmp_ge_f32", 128, 8);
  run<4>("v_sub_f32 clamp (VOP3)", 128, 8);
  run<5>("v_mov_b32", 128, 8);
  run<6>("v_mul_lo_u32", 128, 8);
  run<7>("v_lshl_add_u64", 128, 8);
  run<8>("v_mov_b64", 128, 8);
  run<9>("v_rsq_f32", 128, 8);
  run<10>("v_addc_co_u32 (vcc chain)", 128, 8);
  run<11>("v_pk_fma_f32 (two fp32 per lane)", 128, 8);
  run<12>("v_pk_mul_f32", 128, 8);
  run<13>("v_pk_add_f32", 128, 8);
  return 0;
}
