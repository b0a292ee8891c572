// Dispatch rate of single-wave workgroups with a 15.6 KB LDS allocation each (the shape of altcorr_wave_f16): how long
// does a grid of N such workgroups take when the body is empty, and when it sleeps ~W cycles (concurrency check)?
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/wgrate.hip -o tools/micro/wgrate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDSF, int THREADS>
__global__ __launch_bounds__(THREADS) void k(unsigned* out, int spin) {
  __shared__ float lds[LDSF];
  lds[threadIdx.x] = (float)spin;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) __builtin_amdgcn_s_sleep(8);
  if (lds[threadIdx.x ^ 1] == 12345.f) out[blockIdx.x] = 1;
}
int main() {
  unsigned* out; hipMalloc(&out, 1 << 22);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int spins[] = {0, 2000, 8000};
  for (int cfg = 0; cfg < 3; cfg++)
    for (int sp : spins) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        if (cfg == 0) hipLaunchKernelGGL((k<3904, 64>), dim3(24576), dim3(64), 0, 0, out, sp);
        if (cfg == 1) hipLaunchKernelGGL((k<1024, 64>), dim3(24576), dim3(64), 0, 0, out, sp);
        if (cfg == 2) hipLaunchKernelGGL((k<3904 * 4, 256>), dim3(6144), dim3(256), 0, 0, out, sp);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      const char* names[] = {"24576 x 64 threads, 15.6 KB LDS", "24576 x 64 threads, 4 KB LDS", "6144 x 256 threads, 62 KB LDS"};
      printf("%s, spin %d memtime ticks: %.1f us\n", names[cfg], sp, best * 1e3);
    }
  return 0;
}
