// Rounding behaviour of the fp32 MFMA accumulation chain (v_mfma_f32_16x16x4_f32) against an fp64 reference and
// a v_fma_f32 chain: is the accumulate round-to-nearest (error ~ sqrt(n) ulp, unbiased) or truncating (~ n ulp, biased)?
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_acc.hip -o tools/micro/mfma_acc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// A [16][K], B [16][K] -> C[16][16] = A B^T ; one wave; lane (r = lane&15, g = lane>>4) feeds A[r][4s+g], B[r][4s+g]
__global__ void mfma_chain(const float* A, const float* B, float* C, float* Cf, int K) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  f32x4 c = {0, 0, 0, 0};
  for (int s = 0; s < K; s += 4) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + s + g], B[r * K + s + g], c, 0, 0, 0);
  for (int x = 0; x < 4; x++) C[(4 * g + x) * 16 + r] = c[x];   // row 4g+x of A side, column r of B side
  // plain FMA chain for the same 4 outputs
  for (int x = 0; x < 4; x++) {
    float acc = 0.f;
    const int i = 4 * g + x;
    for (int k = 0; k < K; k++) acc = fmaf(A[i * K + k], B[r * K + k], acc);
    Cf[i * 16 + r] = acc;
  }
}
int main() {
  for (int K : {64, 512, 4096}) {
    for (int sign = 0; sign < 2; sign++) {
      std::vector<float> A(16 * K), B(16 * K);
      srand(1 + K);
      for (auto& x : A) x = (sign ? (rand() % 2 ? 1.f : -1.f) : 1.f) * (0.5f + rand() / (float)RAND_MAX);
      for (auto& x : B) x = 0.5f + rand() / (float)RAND_MAX;
      float *dA, *dB, *dC, *dCf;
      hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024); hipMalloc(&dCf, 1024);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      mfma_chain<<<1, 64>>>(dA, dB, dC, dCf, K);
      float C[256], Cf[256];
      hipMemcpy(C, dC, 1024, hipMemcpyDeviceToHost); hipMemcpy(Cf, dCf, 1024, hipMemcpyDeviceToHost);
      double bm = 0, rm = 0, bf = 0, rf = 0;
      for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
        double ref = 0, mag = 0;
        for (int k = 0; k < K; k++) { ref += (double)A[i * K + k] * B[j * K + k]; mag += fabs((double)A[i * K + k] * B[j * K + k]); }
        const double em = (C[i * 16 + j] - ref) / mag, ef = (Cf[i * 16 + j] - ref) / mag;
        bm += em; rm += em * em; bf += ef; rf += ef * ef;
      }
      printf("K=%5d %s terms: MFMA mean rel err %+.3e rms %.3e | FMA chain mean %+.3e rms %.3e   (fp32 eps/2 = 5.96e-08; relative to sum|terms|)\n",
             K, sign ? "mixed-sign" : "positive  ", bm / 256, sqrt(rm / 256), bf / 256, sqrt(rf / 256));
    }
  }
  return 0;
}
