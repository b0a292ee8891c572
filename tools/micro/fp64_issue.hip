// Issue cost (s_memtime ticks = shader cycles at 2.4 GHz here) of the instructions on the Cholesky pivot chain, for ONE wave
// alone on its SIMD: 64 back-to-back copies, independent (different destination registers) and dependent (a chain).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/fp64_issue.hip -o tools/micro/fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
#define TIME(slot, code) { unsigned long long t0 = __builtin_amdgcn_s_memtime(); for (int r = 0; r < reps; r++) { code } \
    unsigned long long t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[slot] = t1 - t0; }
__global__ void k(double* out, unsigned long long* cyc, int reps) {
  double a = 1.0 + 1e-3 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x, c = 0.5, d0 = 1.5, d1 = 2.5, d2 = 3.5, d3 = 4.5;
  float f = 1.0f + threadIdx.x, g0 = 0.f, g1 = 0.f;
  // independent: 4 destinations round-robin
  TIME(0, R16(asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));))
  TIME(1, R64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(b), "v"(c));))
  TIME(2, R16(asm volatile("v_mul_f64 %0, %4, %0\n v_mul_f64 %1, %4, %1\n v_mul_f64 %2, %4, %2\n v_mul_f64 %3, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b));))
  TIME(3, R16(asm volatile("v_rsq_f64 %0, %4\n v_rsq_f64 %1, %4\n v_rsq_f64 %2, %4\n v_rsq_f64 %3, %4" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a));))
  TIME(4, R64(asm volatile("v_rsq_f64 %0, %0" : "+v"(d0));))
  TIME(5, R16(asm volatile("v_rcp_f64 %0, %4\n v_rcp_f64 %1, %4\n v_rcp_f64 %2, %4\n v_rcp_f64 %3, %4" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a));))
  TIME(6, R16(asm volatile("v_fmac_f64_dpp %0, -%4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, -%4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                           "v_fmac_f64_dpp %2, -%4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, -%4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf"
                           : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));))
  TIME(7, R64(asm volatile("s_nop 1\n v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(d1) : "v"(d0));))
  TIME(8, R64(asm volatile("s_nop 1");))
  TIME(9, R16(asm volatile("v_rsq_f32 %0, %2\n v_rsq_f32 %1, %2\n v_rsq_f32 %0, %2\n v_rsq_f32 %1, %2" : "=v"(g0), "=v"(g1) : "v"(f));))
  TIME(10, R64(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(g0) : "v"(a));))
  TIME(11, R64(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d2) : "v"(f));))
  TIME(12, R64(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(g0) : "v"(f));))
  TIME(13, R16(asm volatile("v_fma_f32 %0, %2, %2, %0\n v_fma_f32 %1, %2, %2, %1\n v_fma_f32 %0, %2, %2, %0\n v_fma_f32 %1, %2, %2, %1" : "+v"(g0), "+v"(g1) : "v"(f));))
  TIME(14, R64(asm volatile("v_rsq_f64 %0, %0\n v_fma_f64 %1, %2, %3, %1" : "+v"(d0), "+v"(d1) : "v"(a), "v"(b));))   // rsq chain + an independent fma each
  out[threadIdx.x] = d0 + d1 + d2 + d3 + g0 + g1;
}
int main() {
  double* d; unsigned long long* c;
  (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&c, 256);
  const int reps = 100;
  k<<<1, 64>>>(d, c, reps); k<<<1, 64>>>(d, c, reps);
  (void)hipDeviceSynchronize();
  unsigned long long h[16];
  (void)hipMemcpy(h, c, 128, hipMemcpyDeviceToHost);
  const char* nm[] = {"v_fma_f64 independent", "v_fma_f64 dependent", "v_mul_f64 independent", "v_rsq_f64 independent", "v_rsq_f64 dependent",
                      "v_rcp_f64 independent", "v_fmac_f64_dpp independent", "s_nop 1 + v_mov_b64_dpp", "s_nop 1", "v_rsq_f32", "v_cvt_f32_f64",
                      "v_cvt_f64_f32", "v_fma_f32 dependent", "v_fma_f32 independent", "v_rsq_f64 dependent + independent v_fma_f64 (pair)"};
  for (int i = 0; i < 15; i++) printf("%-55s %.2f ticks per instruction (pair for the last)\n", nm[i], (double)h[i] / (64.0 * reps));
  return 0;
}
