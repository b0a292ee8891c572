// How much code survives in the instruction cache between two uses?  One wave alternates between a 4 KB block of
// straight-line VALU code (timed) and FILL KB of other straight-line code.  If the timed block slows down once
// FILL exceeds some size, that size is the usable instruction-cache capacity for a loop body.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/icache.hip -o tools/micro/icache
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
#define R256(x) R64(x) R64(x) R64(x) R64(x)
// v_add_f32 e32 = 4 bytes; 1024 of them = 4 KB
#define BLOCK4K(reg) R256(asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(reg));)
template <int FILL4K>
__device__ __attribute__((noinline)) void region(float& x, float& y, unsigned long long& tot, int reps) {
  for (int r = 0; r < reps; r++) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    BLOCK4K(x)
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (r >= 4) tot += t1 - t0;
    if (FILL4K >= 2) { BLOCK4K(y) BLOCK4K(y) }
    if (FILL4K >= 4) { BLOCK4K(y) BLOCK4K(y) }
    if (FILL4K >= 6) { BLOCK4K(y) BLOCK4K(y) }
    if (FILL4K >= 8) { BLOCK4K(y) BLOCK4K(y) }
    if (FILL4K >= 12) { BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) }
  }
}
template <int FILL4K>
__device__ __attribute__((noinline)) void region2(float& x, float& y, unsigned long long& tot, int reps) {
  for (int r = 0; r < reps; r++) {   // the same shape at other addresses (and other instructions, so that nothing is merged)
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    R256(asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0" : "+v"(x));)
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (r >= 4) tot += t1 - t0;
#define MBLK R256(asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0" : "+v"(y));)
    if (FILL4K >= 2) { MBLK MBLK }
    if (FILL4K >= 4) { MBLK MBLK }
    if (FILL4K >= 6) { MBLK MBLK }
    if (FILL4K >= 8) { MBLK MBLK }
    if (FILL4K >= 12) { MBLK MBLK MBLK MBLK }
  }
}
// neighbouring workgroups (one per CU) run DIFFERENT code regions: is the instruction cache shared between CUs?
template <int FILL4K>
__global__ void k2(float* out, unsigned long long* cyc, int reps) {
  float x = 1e-30f * threadIdx.x, y = 2e-30f;
  unsigned long long tot = 0;
  if (blockIdx.x & 1) region2<FILL4K>(x, y, tot, reps); else region<FILL4K>(x, y, tot, reps);
  out[threadIdx.x] = x + y;
  if (threadIdx.x == 0 && blockIdx.x < 2) cyc[blockIdx.x] = tot / (reps - 4);
}
template <int FILL4K>
__global__ void k(float* out, unsigned long long* cyc, int reps) {
  float x = 1e-30f * threadIdx.x, y = 2e-30f;
  unsigned long long tot = 0;
  for (int r = 0; r < reps; r++) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    BLOCK4K(x)
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (r >= 4) tot += t1 - t0;
    if (FILL4K >= 1) { BLOCK4K(y) }
    if (FILL4K >= 2) { BLOCK4K(y) }
    if (FILL4K >= 4) { BLOCK4K(y) BLOCK4K(y) }
    if (FILL4K >= 6) { BLOCK4K(y) BLOCK4K(y) }
    if (FILL4K >= 8) { BLOCK4K(y) BLOCK4K(y) }
    if (FILL4K >= 12) { BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) }
    if (FILL4K >= 16) { BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) }
    if (FILL4K >= 24) { BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) }
    if (FILL4K >= 32) { BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) BLOCK4K(y) }
  }
  out[threadIdx.x] = x + y;
  if (threadIdx.x == 0) cyc[0] = tot / (reps - 4);
}
template <int F>
void run(float* d, unsigned long long* c, int grid) {
  k<F><<<grid, 64>>>(d, c, 40); k<F><<<grid, 64>>>(d, c, 40);
  (void)hipDeviceSynchronize();
  unsigned long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("grid %4d: 4 KB block (1024 v_add_f32) with %3d KB of other code between uses: %llu ticks\n", grid, 4 * F, h);
}
template <int F>
void run2(float* d, unsigned long long* c, int grid) {
  k2<F><<<grid, 64>>>(d, c, 40); k2<F><<<grid, 64>>>(d, c, 40);
  (void)hipDeviceSynchronize();
  unsigned long long h[2]; (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
  printf("two code regions (even / odd workgroups), grid %4d: 4 KB block with %3d KB between uses: %llu / %llu ticks\n", grid, 4 * F, h[0], h[1]);
}
int main() {
  float* d; unsigned long long* c;
  (void)hipMalloc(&d, 64 * 4); (void)hipMalloc(&c, 64);
  for (int grid : {1, 512}) {
    run<0>(d, c, grid); run<1>(d, c, grid); run<2>(d, c, grid); run<4>(d, c, grid); run<6>(d, c, grid); run<8>(d, c, grid);
    run<12>(d, c, grid); run<16>(d, c, grid); run<24>(d, c, grid); run<32>(d, c, grid);
  }
  for (int grid : {2, 256, 512}) { run2<0>(d, c, grid); run2<4>(d, c, grid); run2<6>(d, c, grid); run2<8>(d, c, grid); run2<12>(d, c, grid); }
  return 0;
}
