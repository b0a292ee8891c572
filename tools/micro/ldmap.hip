// How does the vector memory path price a 16-byte-per-lane load by its lane -> address map when the data sit in L2?
// (design question of the f16 alt-corr kernel: the operand layout of v_mfma_f32_16x16x32_f16 wants lane (row = l & 15,
// k-group = l >> 4), i.e. the four lanes that read the 64 contiguous bytes of one row are 16 lanes apart.)
// Rows of 256 B (128 halves), a wave-instruction reads 16 rows x 64 B; rows of a block are consecutive positions of a
// small map (good locality), blocks are random.  Modes:
//   0  lane -> (row = l >> 2, chunk = l & 3): four ADJACENT lanes share a 64-B run          (the LDS-DMA staging map)
//   1  lane -> (row = l & 15, chunk = l >> 4): the MFMA operand map, straight from memory
//   2  as 0 but via global_load_lds_dwordx4 (LDS-DMA) followed by ds_read_b128 in the MFMA map
//   3  LDS-DMA with lane -> (row = l >> 4, chunk = l & 15): one instruction = 4 whole rows of 256 B (16 adjacent lanes per
//      row, chunks XOR-permuted inside the row), 4 instructions per block of 16 rows; ds_read_b128 in the MFMA map
// Reports bytes/clk/CU.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/ldmap.hip -o tools/micro/ldmap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ __launch_bounds__(64) void k(const char* __restrict__ buf, unsigned* __restrict__ out, int nrows, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[4][1024];
  const int l = threadIdx.x;
  const int row = (MODE == 1) ? (l & 15) : (l >> 2), chunk = (MODE == 1) ? (l >> 4) : (l & 3);
  if (MODE == 3) {
    u4 acc = {0, 0, 0, 0};
    unsigned h = hash32(blockIdx.x * 977u + 13u);
    const int i = l & 15, g = l >> 4;
    for (int it = 0; it < iters; it++) {
      h = hash32(h + it);
      const unsigned r0 = h % (unsigned)(nrows - 16);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int pos = 4 * t + (l >> 4);
        const char* src = buf + (size_t)(r0 + pos) * 256 + 16 * ((l & 15) ^ pos);
        const unsigned dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds[t]);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int s = 0; s < 4; s++) acc += *reinterpret_cast<const u4*>(&lds[0][0] + i * 256 + 64 * (s ^ (i >> 2)) + 16 * (g ^ (i & 3)));
    }
    out[blockIdx.x * 64 + l] = acc.x ^ acc.y ^ acc.z ^ acc.w;
    return;
  }
  u4 acc = {0, 0, 0, 0};
  unsigned h = hash32(blockIdx.x * 977u + 13u);
  for (int it = 0; it < iters; it++) {
    h = hash32(h + it);
    const unsigned r0 = h % (unsigned)(nrows - 16);
    const char* p = buf + (size_t)(r0 + row) * 256 + chunk * 16;
    if (MODE == 2) {
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const unsigned dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds[s]);
        const char* src = p + 64 * s;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int s = 0; s < 4; s++) acc += *reinterpret_cast<const u4*>(lds[s] + 64 * (l & 15) + 16 * ((l >> 4) ^ ((l >> 1) & 3)));
    } else {
      u4 v[4];
#pragma unroll
      for (int s = 0; s < 4; s++) v[s] = *reinterpret_cast<const u4*>(p + 64 * s);
#pragma unroll
      for (int s = 0; s < 4; s++) acc += v[s];
    }
  }
  out[blockIdx.x * 64 + l] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main(int argc, char** argv) {
  const int nrows = 3072 * 4;  // 3 MB: inside one XCD's 4 MB L2
  const int iters = 200, wgs = 256 * 10 * 4;
  char* buf; unsigned* out;
  hipMalloc(&buf, (size_t)nrows * 256); hipMemset(buf, 1, (size_t)nrows * 256);
  hipMalloc(&out, (size_t)wgs * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 4; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(64), 0, 0, buf, out, nrows, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(64), 0, 0, buf, out, nrows, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(64), 0, 0, buf, out, nrows, iters);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(64), 0, 0, buf, out, nrows, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)wgs * iters * 4096.0;
      if (rep == 2)
        printf("mode %d: %.3f ms, %.1f GB/s chip, %.1f B/clk/CU at 2.4 GHz (%d single-wave workgroups x %d blocks of 4 KB)\n", mode, ms,
               bytes / ms / 1e6, bytes / ms / 1e6 / 256 / 2.4, wgs, iters);
    }
  }
  return 0;
}
