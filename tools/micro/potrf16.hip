// Variants of the in-register 16x16 Cholesky + inverse (chol.hip wave_potrf16) on ONE wave alone on its SIMD:
//   0 = shipped form (rsqrt on the chain, scaled columns)
//   1 = shipped form without the identity rows (timing only)
//   2 = reciprocal on the chain: t = u/d, updates a[r][c] -= u[r] t[c]; rsqrt + scaling off the chain
//   3 = variant 2 without the identity rows
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/potrf16.hip -o tools/micro/potrf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "potrf16_asm.inc"
__device__ __forceinline__ double rsqrt_h(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double e = fma(-d * y, y, 1.0);
  const double p = fma(0.375, e, 0.5);
  return fma(y * e, p, y);
}
__device__ __forceinline__ double rcp_h(double d) {  // 1/d: v_rcp_f64 seed + one cubic step
  double r = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, r, 1.0);
  const double q = fma(e, e, e);   // e + e^2
  return fma(r, q, r);
}
template <int C>
__device__ __forceinline__ void fmac_bcast(double& acc, const double& piv, const double& own) {
  asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(piv), "v"(own), "n"(C));
}
template <int C>
__device__ __forceinline__ double mul_bcast(const double& piv, const double& own) {  // piv[lane C] * own (v_mul_f64 has no DPP form)
  double r = 0.0;
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(piv), "v"(own), "n"(C));
  return r;
}
template <int V, int J>
__device__ __forceinline__ void pivot(double (&a)[16], double (&w)[16], double& ylast) {
  if (V < 2) {
    double d;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(a[J]), "n"(J));
    const double y = rsqrt_h(d);
    if (J == 15) ylast = y;
    a[J] *= y;
    if (V == 0) w[J] *= y;
    asm volatile("s_nop 1" : "+v"(a[J]));
#define CASE01(CC) if (CC > J) { fmac_bcast<CC>(a[CC], a[J], a[J]); if (V == 0) fmac_bcast<CC>(w[CC], a[J], w[J]); }
    CASE01(1) CASE01(2) CASE01(3) CASE01(4) CASE01(5) CASE01(6) CASE01(7) CASE01(8) CASE01(9) CASE01(10) CASE01(11) CASE01(12) CASE01(13) CASE01(14) CASE01(15)
  } else {
    // every lane inverts its own a[r][J]; only lane J's value is used (DPP broadcast inside the multiply)
    const double inv = rcp_h(a[J]);
    asm volatile("s_nop 1" ::: "memory");
    double t = mul_bcast<J>(inv, a[J]);          // t[r] = a[r][J] / d
    asm volatile("s_nop 1" : "+v"(t));
#define CASE23(CC) if (CC > J) { fmac_bcast<CC>(a[CC], t, a[J]); if (V == 2) fmac_bcast<CC>(w[CC], t, w[J]); }
    CASE23(1) CASE23(2) CASE23(3) CASE23(4) CASE23(5) CASE23(6) CASE23(7) CASE23(8) CASE23(9) CASE23(10) CASE23(11) CASE23(12) CASE23(13) CASE23(14) CASE23(15)
    // off the chain: the scaling of column J
    const double y = rsqrt_h(a[J]);
    asm volatile("s_nop 1" ::: "memory");
    a[J] = mul_bcast<J>(y, a[J]);
    if (V == 2) w[J] = mul_bcast<J>(y, w[J]);
    if (J == 15) ylast = y;
  }
}

// ---- variant 4: the shipped arithmetic, software-pipelined by hand: the reciprocal-root chain of pivot J+1 (every
// instruction depends on the one before; a lone wave issues in order, so a waiting instruction blocks everything behind it)
// is interleaved with the independent column updates of pivot J.  All chain instructions are volatile asm so that the
// order below is the issue order.
#define A_MOVB(d, src, L) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(src), "n"(L))
#define A_RSQ(y, d) asm volatile("v_rsq_f64 %0, %1" : "=v"(y) : "v"(d))
#define A_MUL(r, x, y) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y))
#define A_FMA(r, x, y, z) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z))
#define A_FMAN(r, x, y, z) asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z))
template <int J, int U>
__device__ __forceinline__ void upd(double (&a)[16], double (&w)[16]) {
  // update number U of pivot J (U = 0: a[J+1]; then w[J+1], a[J+2], w[J+2], ...)
  constexpr int c = J + 1 + (U >> 1);
  if (c < 16) {
    if ((U & 1) == 0) fmac_bcast<c>(a[c], a[J], a[J]);
    else fmac_bcast<c>(w[c], a[J], w[J]);
  }
}
template <int J>
__device__ __forceinline__ void pivot_sp(double (&a)[16], double (&w)[16], double& y, double& ylast) {
  // y = 1/sqrt(pivot J) on entry; leaves 1/sqrt(pivot J+1)
  const double c38 = 0.375, c12 = 0.5, one = 1.0;
  A_MUL(a[J], a[J], y);
  A_MUL(w[J], w[J], y);
  if (J == 15) { ylast = y; return; }
  asm volatile("s_nop 1");
  upd<J, 0>(a, w);                       // a[J+1]: the next pivot column
  upd<J, 1>(a, w); upd<J, 2>(a, w);      // (also the two wait states in front of the DPP read of a[J+1])
  if (J >= 14) asm volatile("s_nop 1");
  double d, y0, t, e, pp, ye;
  A_MOVB(d, a[J + 1], J + 1);
  upd<J, 3>(a, w); upd<J, 4>(a, w);
  A_RSQ(y0, d);
  upd<J, 5>(a, w); upd<J, 6>(a, w); upd<J, 7>(a, w);
  A_MUL(t, d, y0);
  upd<J, 8>(a, w); upd<J, 9>(a, w); upd<J, 10>(a, w);
  A_FMAN(e, t, y0, one);
  upd<J, 11>(a, w); upd<J, 12>(a, w); upd<J, 13>(a, w);
  A_FMA(pp, c38, e, c12);
  A_MUL(ye, y0, e);
  upd<J, 14>(a, w); upd<J, 15>(a, w); upd<J, 16>(a, w);
  A_FMA(y, ye, pp, y0);
  upd<J, 17>(a, w); upd<J, 18>(a, w); upd<J, 19>(a, w); upd<J, 20>(a, w); upd<J, 21>(a, w); upd<J, 22>(a, w);
  upd<J, 23>(a, w); upd<J, 24>(a, w); upd<J, 25>(a, w); upd<J, 26>(a, w); upd<J, 27>(a, w); upd<J, 28>(a, w); upd<J, 29>(a, w);
}
template <int V>
__global__ void k(const double* A, double* L, double* W, unsigned long long* cyc, int reps, int nactive = 1) {
  __shared__ double LbAll[8][16 * 18], Idn[256], WlAll[8][256], Asrc[256];
  double* Lb = LbAll[threadIdx.x >> 6];
  double* Wl = WlAll[threadIdx.x >> 6];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) Asrc[i] = A[i];
  const int lane = threadIdx.x & 63, row = lane & 15;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) Idn[i] = ((i >> 4) == (i & 15)) ? 1.0 : 0.0;
  __syncthreads();
  if ((int)threadIdx.x >= 64 * nactive) return;
  unsigned long long t0 = 0, t1 = 0;
  for (int r = 0; r < reps; r++) {
    for (int c = lane; c < 256; c += 64) Lb[(c >> 4) * 18 + (c & 15)] = Asrc[c];
    __builtin_amdgcn_s_waitcnt(0);
    if (r == 1) t0 = __builtin_amdgcn_s_memtime();
    double a[16], w[16];
#pragma unroll
    for (int c = 0; c < 16; c++) { a[c] = Lb[row * 18 + c]; w[c] = Idn[row * 16 + c]; }
    double ylast = 0;
    if (V == 5) {   // the generated single-block form (tools/gen_potrf16_asm.py)
      if (lane < 16) {
        const unsigned pa = (unsigned)(size_t)((__attribute__((address_space(3))) double*)&Lb[row * 18]);
        const unsigned pi = (unsigned)(size_t)((__attribute__((address_space(3))) double*)&Idn[row * 16]);
        const unsigned pw = (unsigned)(size_t)((__attribute__((address_space(3))) double*)&Wl[row]);
        double yl;
        asm volatile(DROID_POTRF16_ASM : "=&v"(yl) : "v"(pa), "v"(pi), "v"(pw) : DROID_POTRF16_CLOBBERS);
        asm volatile("" :: "v"(yl));
      }
      __builtin_amdgcn_s_waitcnt(0);
      continue;
    }
    if (V == 4) {
      double d0, y;
      asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(d0) : "v"(a[0]));
      y = rsqrt_h(d0);
      pivot_sp<0>(a, w, y, ylast); pivot_sp<1>(a, w, y, ylast); pivot_sp<2>(a, w, y, ylast); pivot_sp<3>(a, w, y, ylast);
      pivot_sp<4>(a, w, y, ylast); pivot_sp<5>(a, w, y, ylast); pivot_sp<6>(a, w, y, ylast); pivot_sp<7>(a, w, y, ylast);
      pivot_sp<8>(a, w, y, ylast); pivot_sp<9>(a, w, y, ylast); pivot_sp<10>(a, w, y, ylast); pivot_sp<11>(a, w, y, ylast);
      pivot_sp<12>(a, w, y, ylast); pivot_sp<13>(a, w, y, ylast); pivot_sp<14>(a, w, y, ylast); pivot_sp<15>(a, w, y, ylast);
    } else {
    pivot<(V == 4 ? 0 : V), 0>(a, w, ylast); pivot<(V == 4 ? 0 : V), 1>(a, w, ylast); pivot<(V == 4 ? 0 : V), 2>(a, w, ylast); pivot<(V == 4 ? 0 : V), 3>(a, w, ylast);
    pivot<(V == 4 ? 0 : V), 4>(a, w, ylast); pivot<(V == 4 ? 0 : V), 5>(a, w, ylast); pivot<(V == 4 ? 0 : V), 6>(a, w, ylast); pivot<(V == 4 ? 0 : V), 7>(a, w, ylast);
    pivot<(V == 4 ? 0 : V), 8>(a, w, ylast); pivot<(V == 4 ? 0 : V), 9>(a, w, ylast); pivot<(V == 4 ? 0 : V), 10>(a, w, ylast); pivot<(V == 4 ? 0 : V), 11>(a, w, ylast);
    pivot<(V == 4 ? 0 : V), 12>(a, w, ylast); pivot<(V == 4 ? 0 : V), 13>(a, w, ylast); pivot<(V == 4 ? 0 : V), 14>(a, w, ylast); pivot<(V == 4 ? 0 : V), 15>(a, w, ylast);
    }
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 16; c++) { Lb[row * 18 + c] = a[c]; Wl[c * 16 + row] = w[c]; }
    }
    __builtin_amdgcn_s_waitcnt(0);
  }
  t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x < 64) for (int c = lane; c < 256; c += 64) { L[c] = Lb[(c >> 4) * 18 + (c & 15)]; W[c] = Wl[c]; }
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V>
void run(const double* dA, const std::vector<double>& A, int threads, int nactive = 1) {
  double *dL, *dW; unsigned long long* dc;
  (void)hipMalloc(&dL, 2048); (void)hipMalloc(&dW, 2048); (void)hipMalloc(&dc, 64);
  const int reps = 2001;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<V><<<1, threads>>>(dA, dL, dW, dc, reps, nactive);
  (void)hipEventRecord(e0);
  k<V><<<1, threads>>>(dA, dL, dW, dc, reps, nactive);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<double> L(256), W(256); unsigned long long cyc;
  (void)hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost); (void)hipMemcpy(W.data(), dW, 2048, hipMemcpyDeviceToHost);
  (void)hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
  // residuals: L L^T - A (lower), W = L^-1 (W[j*16+k] = Linv[j][k])
  double e1m = 0, e2m = 0;
  for (int i = 0; i < 16; i++) for (int j = 0; j <= i; j++) {
    double s = 0; for (int kk = 0; kk <= j; kk++) s += L[i * 16 + kk] * L[j * 16 + kk];
    e1m = fmax(e1m, fabs(s - A[i * 16 + j]));
    double t = 0; for (int kk = j; kk <= i; kk++) t += W[i * 16 + kk] * L[kk * 16 + j];
    e2m = fmax(e2m, fabs(t - (i == j ? 1.0 : 0.0)));
  }
  double cs = 0; for (int i = 0; i < 16; i++) for (int j = 0; j <= i; j++) cs += L[i * 16 + j] * (1 + i + 17 * j) + W[i * 16 + j] * (3 + j + 13 * i);
  printf("checksum %.17g  ", cs);
  printf("[%d active waves] ", nactive);
  printf("variant %d (%d threads): %.3f us per block (events), %.1f s_memtime ticks per block | |LL^T-A| %.2e |L^-1 L - I| %.2e\n",
         V, threads, ms * 1e3 / reps, (double)cyc / (reps - 1), e1m, e2m);
}
int main() {
  std::vector<double> B(256), A(256);
  for (int i = 0; i < 256; i++) B[i] = sin(0.37 * i) + ((i % 17) == 0 ? 3.0 : 0.0);
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = (i == j) ? 4.0 : 0.0; for (int kk = 0; kk < 16; kk++) s += B[i * 16 + kk] * B[j * 16 + kk]; A[i * 16 + j] = s; }
  double* dA; (void)hipMalloc(&dA, 2048); (void)hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
  for (int threads : {64, 512}) { run<0>(dA, A, threads); run<1>(dA, A, threads); run<2>(dA, A, threads); run<3>(dA, A, threads); run<4>(dA, A, threads); run<5>(dA, A, threads); }
  for (int na : {1, 2, 3, 4, 5, 8}) run<0>(dA, A, 512, na);
  return 0;
}
