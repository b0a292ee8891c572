// Cost of the column operation of the in-register 16x16 Cholesky (chol.hip wave_potrf16) on ONE wave that runs alone
// on its SIMD: multiplier broadcast by DPP inside the FMA (v_fmac_f64_dpp row_newbcast, serves the 16 lanes of each
// row) against v_readlane x2 + v_fma_f64 with an SGPR-pair multiplier (serves all 64 lanes).  s_memtime cycles per
// column operation, 15 dependent-free operations per "pivot" like the real chain.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/bcast_fma.hip -o tools/micro/bcast_fma
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, unsigned long long* cyc, int reps) {
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; c++) a[c] = 1.0 + 1e-3 * (threadIdx.x + c);
  double piv = 1e-6 * threadIdx.x;
  // ---- DPP form
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
#define DPPOP(CC) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:" #CC " row_mask:0xf bank_mask:0xf" : "+v"(a[CC]) : "v"(piv), "v"(piv));
    DPPOP(1) DPPOP(2) DPPOP(3) DPPOP(4) DPPOP(5) DPPOP(6) DPPOP(7) DPPOP(8) DPPOP(9) DPPOP(10) DPPOP(11) DPPOP(12) DPPOP(13) DPPOP(14) DPPOP(15)
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  // ---- readlane + SGPR form: multiplier of column CC = lane CC's piv
  for (int r = 0; r < reps; r++) {
#define SOP(CC) { int lo = __builtin_amdgcn_readlane(__double2loint(piv), CC), hi = __builtin_amdgcn_readlane(__double2hiint(piv), CC); \
                  double m = __hiloint2double(hi, lo); a[CC] = fma(-m, piv, a[CC]); }
    SOP(1) SOP(2) SOP(3) SOP(4) SOP(5) SOP(6) SOP(7) SOP(8) SOP(9) SOP(10) SOP(11) SOP(12) SOP(13) SOP(14) SOP(15)
    asm volatile("" : "+v"(piv));
  }
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  // ---- plain VALU fp64 FMA with vector operands (reference issue rate)
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int c = 1; c < 16; c++) a[c] = fma(-piv, piv, a[c]);
    asm volatile("" : "+v"(piv));
  }
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int c = 0; c < 16; c++) s += a[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}
int main() {
  double* d; unsigned long long* c;
  (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&c, 64);
  const int reps = 2000;
  for (int lanes : {64, 32, 16}) {   // does a wave with only its first 16 / 32 lanes active issue faster?
    k<<<1, lanes>>>(d, c, reps);
    k<<<1, lanes>>>(d, c, reps);
    (void)hipDeviceSynchronize();
    unsigned long long h[3];
    (void)hipMemcpy(h, c, 24, hipMemcpyDeviceToHost);
    const double ops = 15.0 * reps;
    printf("%2d active lanes: s_memtime ticks per column operation: DPP fmac %.2f | 2x readlane + SGPR fma %.2f | plain v_fma_f64 %.2f\n",
           lanes, h[0] / ops, h[1] / ops, h[2] / ops);
  }
  return 0;
}
