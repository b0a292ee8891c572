// Semantics of the immediate offset of global_load_lds_dwordx4 (saddr + 32-bit voffset form): does `offset:N` move the
// global address only, or the LDS destination too?  One wave; global buffer g[i] = i (dwords); LDS preset to 0xFFFFFFFF.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/glds_off.hip -o tools/micro/glds_off
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(const unsigned* __restrict__ g, unsigned* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[1024];
  const int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = 0xFFFFFFFFu;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned*)lds);
  const unsigned voff = l * 16;               // bytes: lane-linear source
  const unsigned m = base + 1024;             // LDS destination base: byte 1024
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:64\n\ts_waitcnt vmcnt(0)"
               :: "v"(voff), "s"(g), "s"(m) : "memory");
  __syncthreads();
  for (int i = l; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  unsigned *g, *out, h[4096], o[1024];
  for (int i = 0; i < 4096; i++) h[i] = i;
  hipMalloc(&g, sizeof(h)); hipMalloc(&out, sizeof(o));
  hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, out);
  hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
  int first = -1, last = -1;
  for (int i = 0; i < 1024; i++) if (o[i] != 0xFFFFFFFFu) { if (first < 0) first = i; last = i; }
  printf("LDS dwords written: [%d, %d] (expected base dword 256 if the offset is global-only, 272 if it moves the LDS side too)\n", first, last);
  if (first >= 0) printf("first values: %u %u %u %u (16 = global byte offset 64 applied)\n", o[first], o[first + 1], o[first + 2], o[first + 3]);
  return 0;
}
