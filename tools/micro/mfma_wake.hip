// Does the matrix pipe go to sleep?  One wave: N dependent v_fma_f64 (VALU only), then four dependent
// v_mfma_f64_16x16x4_f64; s_memtime around the MFMA group, for growing N.  Also: LDS operand loads + MFMAs + LDS store
// (the shape of chol.hip's wave_gemm_nt16) after the same idle gaps.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_wake.hip -o tools/micro/mfma_wake
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, unsigned long long* cyc, int nvalu, int mode) {
  __shared__ double T[16 * 66 * 2];
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x; i < 16 * 66 * 2; i += blockDim.x) T[i] = 1e-3 * i;
  __syncthreads();
  if (threadIdx.x >= 64) return;
  double x = 1.0 + 1e-9 * lane;
  unsigned long long tot = 0;
  f64x4 acc = {0, 0, 0, 0};
  for (int rep = 0; rep < 64; rep++) {
    for (int i = 0; i < nvalu; i++) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x));
    asm volatile("s_nop 4" ::: "memory");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
      asm volatile("" : "+v"(acc));
      x += acc[0] * 1e-300;
    } else if (mode == 2) {   // the same product with FLAT (generic-pointer) accesses to LDS, as chol.hip's wave_gemm_nt16
      double* Tf = (double*)T;
      asm volatile("" : "+v"(Tf));
      f64x4 a2 = {0, 0, 0, 0};
#pragma unroll
      for (int kk = 0; kk < 16; kk += 4) a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Tf[r * 66 + kk + g], Tf[16 * 66 + r * 66 + kk + g], a2, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; i++) Tf[(g + 4 * i) * 66 + r + 32] -= a2[i];
      __builtin_amdgcn_s_waitcnt(0);
    } else if (mode == 3) {   // one global load (L2 hit)
      x += out[64 + lane];
      asm volatile("" : "+v"(x));
    } else {
      f64x4 a2 = {0, 0, 0, 0};
#pragma unroll
      for (int kk = 0; kk < 16; kk += 4) a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[r * 66 + kk + g], T[16 * 66 + r * 66 + kk + g], a2, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; i++) T[(g + 4 * i) * 66 + r + 32] -= a2[i];
      __builtin_amdgcn_s_waitcnt(0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (rep >= 8) tot += t1 - t0;
  }
  out[threadIdx.x] = x + acc[1] + T[lane];
  if (threadIdx.x == 0) cyc[0] = tot / 56;
}
int main() {
  double* d; unsigned long long* c;
  (void)hipMalloc(&d, 512 * 8); (void)hipMalloc(&c, 64);
  for (int mode = 0; mode < 4; mode++)
    for (int n : {0, 16, 64, 256, 1024, 4096}) {
      k<<<1, 64>>>(d, c, n, mode); k<<<1, 64>>>(d, c, n, mode);
      (void)hipDeviceSynchronize();
      unsigned long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
      printf("%s after %5d dependent v_fma_f64 (~%6d cycles idle matrix pipe): %llu ticks\n", mode == 0 ? "4 dependent f64 MFMAs" : mode == 1 ? "LDS gemm 16x16x16 (ds loads, 4 MFMA, RMW store)" : mode == 2 ? "same product with FLAT accesses to LDS" : "one global load (L2 hit)", n, n * 8, h);
    }
  return 0;
}
