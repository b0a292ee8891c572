// Can v_mfma_f32_16x16x32_bf16 replace v_mfma_f32_16x16x4_f32 in an fp32 SYRK?  Every fp32 operand is split into three
// bf16 terms (x = b0 + b1 + b2 to 2^-27, round to nearest) and a 32-deep k step becomes the six products of order <= 2
// (b0b0, b0b1, b1b0, b0b2, b1b1, b2b0: every bf16 x bf16 product is exact in fp32): 6 x 16 cycles instead of 8 x 32.
// What the test answers: is the accumulation inside / between those MFMAs round-to-nearest (unbiased), and how does the
// result compare with the fp32 MFMA chain of the same data?  Variants: (a) one accumulator for all six products,
// (b) low-order products in an accumulator of their own, (c) a fresh accumulator per k step, totals added on the VALU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_acc_bf16.hip -o tools/micro/mfma_acc_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& b0, __bf16& b1, __bf16& b2) {
  b0 = (__bf16)x;
  const float r1 = x - (float)b0;
  b1 = (__bf16)r1;
  const float r2 = r1 - (float)b1;
  b2 = (__bf16)r2;
}

// A [16][K], B [16][K] -> C = A B^T; lane (r = lane & 15, g = lane >> 4) feeds k = 32 s + 8 g .. + 7 of row r
__global__ void chains(const float* A, const float* B, float* C32, float* Ca, float* Cb, float* Cc, int K) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  f32x4 c32 = {0, 0, 0, 0}, ca = {0, 0, 0, 0}, cb_hi = {0, 0, 0, 0}, cb_lo = {0, 0, 0, 0}, cc = {0, 0, 0, 0};
  for (int s = 0; s < K; s += 32) {
    // fp32 MFMA chain over the same 32 k values (any order of k is a valid chain)
    for (int e = 0; e < 8; e++)
      c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + s + 4 * e + g], B[r * K + s + 4 * e + g], c32, 0, 0, 0);
    bf16x8 a[3], b[3];
    for (int e = 0; e < 8; e++) {
      __bf16 t0, t1, t2;
      split3(A[r * K + s + 8 * g + e], t0, t1, t2);
      a[0][e] = t0; a[1][e] = t1; a[2][e] = t2;
      split3(B[r * K + s + 8 * g + e], t0, t1, t2);
      b[0][e] = t0; b[1][e] = t1; b[2][e] = t2;
    }
    // (a) one accumulator, small terms first
    ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], ca, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], ca, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], ca, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], ca, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], ca, 0, 0, 0);
    ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], ca, 0, 0, 0);
    // (b) low-order products apart
    cb_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], cb_lo, 0, 0, 0);
    cb_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], cb_lo, 0, 0, 0);
    cb_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], cb_lo, 0, 0, 0);
    cb_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], cb_lo, 0, 0, 0);
    cb_lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], cb_lo, 0, 0, 0);
    cb_hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], cb_hi, 0, 0, 0);
    // (c) fresh accumulator per k step, fp32 total on the VALU
    f32x4 z = {0, 0, 0, 0};
    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], z, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], z, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], z, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], z, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], z, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], z, 0, 0, 0);
    cc += z;
  }
  for (int x = 0; x < 4; x++) {
    const int o = (4 * g + x) * 16 + r;
    C32[o] = c32[x]; Ca[o] = ca[x]; Cb[o] = cb_hi[x] + cb_lo[x]; Cc[o] = cc[x];
  }
}

int main() {
  for (int K : {64, 512, 1536, 4096}) {
    for (int sign = 0; sign < 3; sign++) {  // 2: positive terms with a wide dynamic range along k (x 2^-12..2^0 per k)
      std::vector<float> A(16 * K), B(16 * K);
      srand(1 + K);
      for (auto& x : A) x = (sign == 1 ? (rand() % 2 ? 1.f : -1.f) : 1.f) * (0.5f + rand() / (float)RAND_MAX);
      for (auto& x : B) x = 0.5f + rand() / (float)RAND_MAX;
      if (sign == 2)
        for (int k = 0; k < K; k++) {
          const float sc = ldexpf(1.f, -(rand() % 13));
          for (int i = 0; i < 16; i++) A[i * K + k] *= sc;
        }
      float *dA, *dB, *dC;
      hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      chains<<<1, 64>>>(dA, dB, dC, dC + 256, dC + 512, dC + 768, K);
      float C[1024];
      hipMemcpy(C, dC, 4096, hipMemcpyDeviceToHost);
      const char* name[4] = {"fp32 MFMA chain      ", "bf16x3, one acc      ", "bf16x3, hi / lo accs ", "bf16x3, per-step + add"};
      printf("K=%5d %s terms:\n", K, sign == 1 ? "mixed-sign" : (sign ? "positive, wide range" : "positive  "));
      for (int v = 0; v < 4; v++) {
        double bm = 0, rm = 0, mx = 0;
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
          double ref = 0, mag = 0;
          for (int k = 0; k < K; k++) { ref += (double)A[i * K + k] * B[j * K + k]; mag += fabs((double)A[i * K + k] * B[j * K + k]); }
          const double e = (C[256 * v + i * 16 + j] - ref) / mag;
          bm += e; rm += e * e; mx = fmax(mx, fabs(e));
        }
        printf("   %s mean rel err %+.3e rms %.3e max %.3e\n", name[v], bm / 256, sqrt(rm / 256), mx);
      }
      hipFree(dA); hipFree(dB); hipFree(dC);
    }
  }
  printf("(fp32 eps/2 = 5.96e-08; errors relative to sum|terms|)\n");
  return 0;
}
