// Do VALU instructions of one wave issue while the matrix pipe works on another wave's MFMAs (same SIMD)?
// 8 waves per workgroup, one workgroup per CU: waves 0-3 run chains of v_mfma_f32_16x16x4_f32 (or fp64 / bf16),
// waves 4-7 run chains of v_fma_f32; timed alone and together.  "together ~ max" = overlap, "~ sum" = none.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int KIND>  // 0: f32 16x16x4, 1: f64 16x16x4
__device__ __forceinline__ void mfma_loop(float* out, int n) {
  if (KIND == 0) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < n; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
      }
    }
    out[threadIdx.x] = c0[0] + c1[1];
  } else {
    f64x4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < n; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
      }
    }
    out[threadIdx.x] = (float)(c0[0] + c1[1]);
  }
}
__device__ __forceinline__ void valu_loop(float* out, int n) {
  float x[8];
#pragma unroll
  for (int u = 0; u < 8; u++) x[u] = threadIdx.x * 0.001f + u;
  const float m = 1.0001f, q = 0.5f;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = fmaf(x[u], m, q);   // 8 independent chains, 32 FMAs per iteration
  }
  float s = 0;
#pragma unroll
  for (int u = 0; u < 8; u++) s += x[u];
  out[threadIdx.x] = s;
}
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int nm, int nv, int mode) {
  const int wave = threadIdx.x >> 6;
  float* o = out + blockIdx.x * 512;
  if (wave < 4) { if (mode & 1) mfma_loop<KIND>(o, nm); }
  else          { if (mode & 2) valu_loop(o, nv); }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int KIND>
int run(const char* name, int nm, int nv) {
  float* d; CK(hipMalloc(&d, 256 * 512 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float t[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; mode++) {
    k<KIND><<<256, 512>>>(d, nm, nv, mode);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) k<KIND><<<256, 512>>>(d, nm, nv, mode);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&t[mode], e0, e1));
    t[mode] *= 200.f;  // us per launch
  }
  const double mf = 256.0 * 4 * nm * 16 * (KIND == 0 ? 2048.0 : 2048.0) / (t[1] * 1e-6) / 1e12;
  printf("%-22s MFMA alone %7.1f us (%.1f TFLOP/s)  VALU alone %7.1f us  together %7.1f us  (sum %.1f, max %.1f)\n", name, t[1], mf, t[2],
         t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
  CK(hipFree(d));
  return 0;
}
int main() {
  if (run<0>("f32 16x16x4, equal", 2000, 4000)) return 1;
  if (run<0>("f32 16x16x4, valu/2", 2000, 2000)) return 1;
  if (run<1>("f64 16x16x4, equal", 1000, 4000)) return 1;
  return 0;
}
