// Launch overhead of hipLaunchCooperativeKernel against a plain launch on gfx950 (VERDICT r01 task 7: can the
// single-launch factorisation afford a cooperative launch, which guarantees residency or fails at launch?).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/coop_launch.hip -o tools/micro/coop_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(512) void k_small(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ __launch_bounds__(512) void k_lds(int* p) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += lds[5];
}
int main() {
  int* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 500;
  void* args[] = {&d};
  for (int lds = 0; lds < 2; lds++) {
    const size_t shmem = lds ? 145 * 1024 : 0;
    if (lds) CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const void* fn = lds ? (const void*)k_lds : (const void*)k_small;
    for (int coop = 0; coop < 2; coop++) {
      for (int warm = 0; warm < 2; warm++) {
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) {
          if (coop) CK(hipLaunchCooperativeKernel(fn, dim3(256), dim3(512), args, shmem, s));
          else CK(hipLaunchKernel(fn, dim3(256), dim3(512), args, shmem, s));
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (warm) printf("%-12s %-22s 256 x 512 threads: %.2f us per dependent launch\n", coop ? "cooperative" : "plain", lds ? "145 KB LDS per WG" : "no LDS", ms * 1e3 / reps);
      }
      // interleaved with a plain kernel (the BA stream alternates kernel kinds)
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; r++) {
        CK(hipLaunchKernel((const void*)k_small, dim3(256), dim3(512), args, 0, s));
        if (coop) CK(hipLaunchCooperativeKernel(fn, dim3(256), dim3(512), args, shmem, s));
        else CK(hipLaunchKernel(fn, dim3(256), dim3(512), args, shmem, s));
      }
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-12s %-22s alternating with a plain kernel: %.2f us per pair\n", coop ? "cooperative" : "plain", lds ? "145 KB LDS per WG" : "no LDS", ms * 1e3 / reps);
    }
  }
  // over-subscription: does the cooperative launch refuse a grid that cannot be resident?
  hipError_t e = hipLaunchCooperativeKernel((const void*)k_lds, dim3(257 * 2), dim3(512), args, 145 * 1024, s);
  printf("cooperative launch of 514 x 145 KB workgroups: %s\n", hipGetErrorString(e));
  (void)hipGetLastError();
  CK(hipStreamSynchronize(s));
  return 0;
}
