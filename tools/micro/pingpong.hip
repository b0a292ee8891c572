// Hand-off latency between two workgroups: flag ping-pong, same XCD vs different XCD, sc1 vs L2-scope loads.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/pingpong.hip -o tools/micro/pingpong ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// never hang the GPU: a poll that does not see its flag within 200k tries raises g_abort and every loop drains
__device__ int g_abort;
#define SPIN(cond) { int spins_ = 0; while (cond) { if (++spins_ > 200000) { g_abort = 1; } if (*(volatile int*)&g_abort) break; } }
// L2-scope load: invalidate this CU's L1, then a plain load (served by the XCD's L2)
__device__ __forceinline__ int l2_load(const int* p) {
  asm volatile("buffer_inv sc0" ::: "memory");
  return *(const volatile int*)p;
}
__device__ __forceinline__ double l2_loadd(const double* p) { return *(const volatile double*)p; }

template <int LOADSCOPE>
__global__ void pingpong(int* flags, unsigned long long* out, int* xcc, int a, int b, int rounds) {
  const int wg = blockIdx.x;
  if (threadIdx.x == 0) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    xcc[wg] = v & 0xf;
  }
  if (wg != a && wg != b) return;
  if (threadIdx.x != 0) return;
  int* mine = flags + (wg == a ? 0 : 64);
  int* other = flags + (wg == a ? 64 : 0);
  const unsigned long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; r++) {
    if (*(volatile int*)&g_abort) break;
    if (wg == a) {
      __hip_atomic_store(other, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      SPIN((LOADSCOPE ? __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : l2_load(mine)) < r)
    } else {
      SPIN((LOADSCOPE ? __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : l2_load(mine)) < r)
      __hip_atomic_store(other, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (wg == a) out[0] = wall_clock64() - t0;
}

// bulk hand-off: producer writes a 32 KB tile then the flag; consumer polls, then loads the tile.
// STORE: 0 = 8-byte sc1 (agent-scope atomic), 1 = plain 16-byte (stays in the XCD's L2), 2 = 16-byte sc1 (asm)
// LOAD:  0 = 8-byte sc1,                      1 = L1 invalidate + plain 16-byte (L2),   2 = 16-byte sc1 (asm)
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <int STORE, int LOAD, int THREADS>
__global__ void tilepass(double* buf, int* flags, unsigned long long* out, int a, int b, int rounds) {
  const int wg = blockIdx.x, t = threadIdx.x;
  if (wg != a && wg != b) return;
  __shared__ double sink[512];
  double acc = 0;
  const unsigned long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; r++) {
    if (*(volatile int*)&g_abort) break;
    const bool producer = ((r & 1) == 1) == (wg == a);
    double* tile = buf + (size_t)(r & 1) * 4096;
    int* flag = flags + (r & 1) * 64;
    if (producer) {
      if (STORE == 0) {
        for (int i = t; i < 4096; i += THREADS) __hip_atomic_store(&tile[i], (double)r + acc * 1e-30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        for (int i = t; i < 2048; i += THREADS) {
          f64x2 v = {(double)r + acc * 1e-30, (double)r};
          if (STORE == 1) *(f64x2*)&tile[2 * i] = v;
          else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(&tile[2 * i]), "v"(v) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) __hip_atomic_store(flag, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (t == 0) SPIN((LOAD != 1 ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : l2_load(flag)) < r)
      __syncthreads();
      if (LOAD == 0) {
        for (int i = t; i < 4096; i += THREADS) acc += __hip_atomic_load(&tile[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (LOAD == 1) {
        asm volatile("buffer_inv sc0" ::: "memory");
        for (int i = t; i < 2048; i += THREADS) { f64x2 v = *(volatile f64x2*)&tile[2 * i]; acc += v[0] + v[1]; }
      } else {
        f64x2 v[2048 / THREADS];
#pragma unroll
        for (int q = 0; q < 2048 / THREADS; q++)
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[q]) : "v"(&tile[2 * (t + q * THREADS)]) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 2048 / THREADS; q++) { asm volatile("" : "+v"(v[q])); acc += v[q][0] + v[q][1]; }
      }
    }
  }
  sink[t] = acc;
  if (wg == a && t == 0) out[0] = wall_clock64() - t0;
  if (acc == 12345.678) out[1] = (unsigned long long)sink[(t + 1) & 255];
}

// Strip hand-off exactly as the single-launch Cholesky does it (chol.hip: chol_strip_out -> load_strip64): ONE wave of the
// producer writes a 16-column strip of a 64x64 tile (8 KB, 16-byte stores) into a data-tagged slot (preset to 0xFF
// bytes, fresh memory every round: no flag, no acknowledgement); 256 threads of the consumer re-read their two 16-byte
// units until no value carries the tag, meet at a barrier, and the consumer answers with a strip of its own.
// MODE 0: sc1 stores + sc1 loads (what ships; works across XCDs).  MODE 1: plain stores + L1 invalidate + plain loads
// (through the producer XCD's own L2: same XCD only) -- candidate (ii) of DESIGN section 8.
template <int MODE>
__global__ __launch_bounds__(256) void strippass(double* slots, unsigned long long* out, int a, int b, int rounds) {
  const int wg = blockIdx.x, t = threadIdx.x;
  if (wg != a && wg != b) return;
  double acc = 0;
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < rounds; r++) {
    if (*(volatile int*)&g_abort) break;
    for (int half = 0; half < 2; half++) {
      const bool producer = (half == 0) == (wg == a);
      double* slot = slots + ((size_t)2 * r + half) * 4096;       // 64 x 64 doubles, strip = columns 48..63
      if (producer) {
        if (t < 64) {
          const int i0 = t >> 3, j = 48 + (t & 7) * 2;
#pragma unroll
          for (int it = 0; it < 8; it++) {
            f64x2 v = {(double)r + acc * 1e-30, (double)(r + it)};
            double* q = &slot[(i0 + 8 * it) * 64 + j];
            if (MODE == 1) *(f64x2*)q = v;
            else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
          }
        }
      } else {
        const double* p0 = &slot[(t >> 3) * 64 + 48 + (t & 7) * 2];
        const double* p1 = p0 + 32 * 64;
        f64x2 v0, v1;
        int spins = 0;
        while (true) {
          if (MODE == 1) {
            asm volatile("buffer_inv sc0" ::: "memory");
            v0 = *(volatile f64x2*)p0; v1 = *(volatile f64x2*)p1;
          } else {
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v0) : "v"(p0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v1) : "v"(p1) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(v0), "+v"(v1));
          }
          const bool there = __double_as_longlong(v0[0]) != -1LL && __double_as_longlong(v0[1]) != -1LL &&
                             __double_as_longlong(v1[0]) != -1LL && __double_as_longlong(v1[1]) != -1LL;
          if (there) break;
          if (++spins > 200000) { g_abort = 1; }
          if (*(volatile int*)&g_abort) break;
        }
        acc += v0[0] + v1[1];
        __syncthreads();
      }
    }
  }
  if (wg == a && t == 0) out[0] = wall_clock64() - t0;
  if (acc == 12345.678) out[1] = 1;
}

template <int MODE>
void run_strip(const char* what, double* slots, size_t slot_bytes, unsigned long long* out, int a, int b, int rounds, int rate) {
  hipMemset(slots, 0xFF, slot_bytes);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((strippass<MODE>), dim3(256), dim3(256), 0, 0, slots, out, a, b, rounds);
  unsigned long long h = 0; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  printf("8 KB strip, data-tagged slot, WG %d -> WG %d, %s: %.0f ns per hand-off\n", a, b, what, (double)h / rate * 1e6 / rounds / 2);
}

template <int STORE, int LOAD, int THREADS>
void run_tile(const char* what, double* buf, int* flags, unsigned long long* out, int a, int b, int rounds, int rate) {
  hipMemset(flags, 0, 1024);
  hipLaunchKernelGGL((tilepass<STORE, LOAD, THREADS>), dim3(256), dim3(THREADS), 0, 0, buf, flags, out, a, b, rounds);
  unsigned long long h = 0; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  printf("32 KB tile + flag WG %d -> WG %d, %d threads, %s: %.0f ns per hand-off\n", a, b, THREADS, what, (double)h / rate * 1e6 / rounds);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  int *flags, *xcc; unsigned long long* out; double* buf;
  hipMalloc(&flags, 1024); hipMalloc(&xcc, 4096); hipMalloc(&out, 64); hipMalloc(&buf, 8192 * 8);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate %d kHz\n", rate);
  const int rounds = 2000;
  hipMemset(flags, 0, 1024);
  hipLaunchKernelGGL(pingpong<1>, dim3(256), dim3(64), 0, 0, flags, out, xcc, 0, 8, 1);   // records the XCC ids
  std::vector<int> x0(256); hipMemcpy(x0.data(), xcc, 1024, hipMemcpyDeviceToHost);
  printf("xcc of WG 0..15:"); for (int i = 0; i < 16; i++) printf(" %d", x0[i]); printf("\n");
  int same = -1, diff = -1;
  for (int i = 1; i < 256; i++) { if (same < 0 && x0[i] == x0[0]) same = i; if (diff < 0 && x0[i] != x0[0]) diff = i; }
  if (same < 0 || diff < 0) { printf("no pair found\n"); return 1; }
  const int pairs[2][2] = {{0, same}, {0, diff}};
  for (auto& p : pairs) for (int scope = 0; scope < 2; scope++) {
    if (!scope && p[1] == diff) continue;  // an L2-scope poll across XCDs is not coherent
    hipMemset(flags, 0, 1024);
    if (scope) hipLaunchKernelGGL(pingpong<1>, dim3(256), dim3(64), 0, 0, flags, out, xcc, p[0], p[1], rounds);
    else hipLaunchKernelGGL(pingpong<0>, dim3(256), dim3(64), 0, 0, flags, out, xcc, p[0], p[1], rounds);
    unsigned long long h = 0; std::vector<int> x(256);
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); hipMemcpy(x.data(), xcc, 1024, hipMemcpyDeviceToHost);
    fflush(stdout);
    printf("flag ping-pong WG %d (xcc %d) <-> WG %d (xcc %d), %s loads: %.0f ns one way\n", p[0], x[p[0]], p[1], x[p[1]],
           scope ? "agent(sc1)" : "L2-scope (L1 invalidate + plain)", (double)h / rate * 1e6 / rounds / 2);
  }
  run_tile<0, 0, 256>("8-byte sc1 stores, 8-byte sc1 loads", buf, flags, out, 0, same, rounds, rate);
  run_tile<0, 0, 256>("8-byte sc1 stores, 8-byte sc1 loads", buf, flags, out, 0, diff, rounds, rate);
  run_tile<0, 0, 512>("8-byte sc1 stores, 8-byte sc1 loads", buf, flags, out, 0, diff, rounds, rate);
  run_tile<2, 2, 256>("16-byte sc1 stores, 16-byte sc1 loads", buf, flags, out, 0, diff, rounds, rate);
  run_tile<2, 2, 512>("16-byte sc1 stores, 16-byte sc1 loads", buf, flags, out, 0, diff, rounds, rate);
  run_tile<2, 2, 512>("16-byte sc1 stores, 16-byte sc1 loads", buf, flags, out, 0, same, rounds, rate);
  run_tile<1, 1, 256>("plain 16-byte stores, L1-invalidate + plain loads (same XCD only)", buf, flags, out, 0, same, rounds, rate);
  run_tile<1, 1, 512>("plain 16-byte stores, L1-invalidate + plain loads (same XCD only)", buf, flags, out, 0, same, rounds, rate);
  run_tile<2, 1, 512>("16-byte sc1 stores, L1-invalidate + plain loads (same XCD only)", buf, flags, out, 0, same, rounds, rate);
  {
    const int srounds = 1000;
    const size_t sbytes = (size_t)srounds * 2 * 4096 * 8;
    double* slots; hipMalloc(&slots, sbytes);
    run_strip<0>("16-byte sc1 stores, 16-byte sc1 polling loads, same XCD", slots, sbytes, out, 0, same, srounds, rate);
    run_strip<0>("16-byte sc1 stores, 16-byte sc1 polling loads, other XCD", slots, sbytes, out, 0, diff, srounds, rate);
    run_strip<1>("plain stores, L1 invalidate + plain polling loads, same XCD (through its L2)", slots, sbytes, out, 0, same, srounds, rate);
    hipFree(slots);
  }
  int ab = 0; hipMemcpyFromSymbol(&ab, HIP_SYMBOL(g_abort), 4);
  printf("abort flag: %d (1 = some poll never saw its flag; numbers above are then meaningless)\n", ab);
  return 0;
}
