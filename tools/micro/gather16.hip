// What does a 16-byte load from a 128-byte line cost at the memory side on gfx950: a 64-byte or a 128-byte request?
// (VERDICT r01 task 3: the level-0 correlation lookup reads 8 window rows x 16 B per query, every row in its own
// 128-B line; rocprofv3's FETCH_SIZE is RDREQ x 64 B and cannot tell the two apart by itself.)
//
// Access pattern = level 0 of the fp16 volume lookup: lane <-> plane of 6144 B (48 x 64 halfs, row pitch 128 B),
// 8 consecutive rows per plane starting at a hashed row; 786432 planes = the 256-edge level-0 volume (4.8 GB), so one
// launch touches 805 MB of distinct 128-B lines and nothing survives in the 256 MiB Infinity Cache between launches.
// Modes (all with the same number of lines touched unless noted):
//   0  stream: 16 B per lane, fully coalesced, the whole buffer          (calibration: known bytes, 128-B requests)
//   1  one 16-B load per line, always in the FIRST 64-B half
//   2  one 16-B load per line, half chosen per lane at random
//   3  two 16-B loads per line, one in EACH 64-B half                    (2 sectors per line)
//   4  two 16-B loads per line, both in the first half                   (1 sector per line, 2 instructions)
//   5  one 16-B load straddling the 64-B boundary (offset 56)            (alignment 8)
// If memory moves 64-B sectors: t(3) ~ 2 t(1), t(4) ~ t(1), FETCH_SIZE(3) = 2 FETCH_SIZE(1).
// If it moves whole 128-B lines: t(3) ~ t(4) ~ t(1) (only the instruction count differs), same FETCH_SIZE.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/gather16.hip -o tools/micro/gather16
// Run:   tools/micro/gather16 [mode|-1 = all]   (rocprofv3 --pmc FETCH_SIZE --kernel-trace -- tools/micro/gather16 N)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr int PLANE = 6144, PITCH = 128, ROWS = 8;

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const char* __restrict__ buf, unsigned* __restrict__ out, int nplanes) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= nplanes) return;
  const unsigned h = hash32((unsigned)p);
  const int y0 = h % 40;
  const int xo = (MODE == 2) ? (((h >> 8) & 1) * 64 + ((h >> 9) & 3) * 16) : (MODE == 5 ? 56 : ((h >> 9) & 3) * 16);
  const char* base = buf + (size_t)p * PLANE + (size_t)y0 * PITCH;
  u4 acc = {0, 0, 0, 0};
  u4 v[ROWS], w[ROWS];
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    if (MODE == 5) {
      const uint2* q = reinterpret_cast<const uint2*>(base + j * PITCH + xo);
      uint2 a = q[0], b = q[1];
      v[j] = u4{a.x, a.y, b.x, b.y};
    } else {
      v[j] = *reinterpret_cast<const u4*>(base + j * PITCH + xo);
    }
    if (MODE == 3) w[j] = *reinterpret_cast<const u4*>(base + j * PITCH + (xo ^ 64) % 128);
    if (MODE == 4) w[j] = *reinterpret_cast<const u4*>(base + j * PITCH + ((xo + 16) & 63));
  }
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    acc += v[j];
    if (MODE == 3 || MODE == 4) acc += w[j];
  }
  out[p] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

__global__ __launch_bounds__(256) void stream_kernel(const u4* __restrict__ buf, unsigned* __restrict__ out, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  u4 acc = {0, 0, 0, 0};
  for (; i < n16; i += stride) acc += buf[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
static void run(const char* buf, unsigned* out, int nplanes, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 10;
  gather_kernel<MODE><<<(nplanes + 255) / 256, 256>>>(buf, out, nplanes);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) gather_kernel<MODE><<<(nplanes + 255) / 256, 256>>>(buf, out, nplanes);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, lines = (double)nplanes * ROWS;
  printf("mode %d  %-52s %8.1f us  %6.2f Glines/s  = %5.2f TB/s @64B  %5.2f TB/s @128B\n", MODE, what, us,
         lines / us * 1e-3, lines * 64 / us * 1e-6, lines * 128 / us * 1e-6);
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : -1;
  const int nplanes = 786432;                       // 4.8 GB: one gather launch touches 805 MB of distinct lines (> 256 MiB Infinity Cache)
  const size_t bytes = (size_t)nplanes * PLANE;
  char* buf; unsigned* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, sizeof(unsigned) * nplanes));
  CK(hipMemset(buf, 1, bytes));
  if (mode < 0 || mode == 0) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    stream_kernel<<<4096, 256>>>(reinterpret_cast<const u4*>(buf), out, bytes / 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; r++) stream_kernel<<<4096, 256>>>(reinterpret_cast<const u4*>(buf), out, bytes / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode 0  %-52s %8.1f us  %6.2f TB/s (%zu bytes per launch)\n", "stream, 16 B per lane coalesced", ms * 100, bytes / (ms * 100) * 1e-6, bytes);
  }
  if (mode < 0 || mode == 1) run<1>(buf, out, nplanes, "1 x 16 B per line, first half");
  if (mode < 0 || mode == 2) run<2>(buf, out, nplanes, "1 x 16 B per line, random half");
  if (mode < 0 || mode == 3) run<3>(buf, out, nplanes, "2 x 16 B per line, both halves");
  if (mode < 0 || mode == 4) run<4>(buf, out, nplanes, "2 x 16 B per line, same half");
  if (mode < 0 || mode == 5) run<5>(buf, out, nplanes, "2 x 8 B straddling the 64-B boundary");
  printf("lines touched per gather launch: %d (x64 = %.1f MB, x128 = %.1f MB)\n", nplanes * ROWS, nplanes * ROWS * 64e-6, nplanes * ROWS * 128e-6);
  CK(hipFree(buf)); CK(hipFree(out));
  return 0;
}
