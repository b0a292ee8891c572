// chol.hip -- on-device dense fp64 Cholesky solve of the reduced camera system.
//
// Replaces the reference's host-side Eigen::SimplicialLLT<double> (SparseBlock::solve,
// /root/reference/src/droid_kernels.cu:1192-1213): same damping `diag += ep + lm*diag` (:1197),
// same fp64 LL^T factorisation of the same matrix, failure => caller zeroes dx (:1207-1210).
//
// Layout: S is ld x ld row-major with ld = n + 1; the lower triangle of S[0:n,0:n] is the
// matrix, row n holds the right-hand side.  Factoring the augmented matrix turns row n into
// y^T = (L^-1 b)^T for free (the forward substitution rides along with the panel TRSM), so only
// the backward substitution L^T x = y remains.
//
// Blocked right-looking factorisation, NB = 64, two launches per block column:
//   panel : every row block re-factors the 64x64 diagonal block in LDS (cheaper than a grid
//           hand-off) and solves its own rows against it,
//   update: A22 -= L21 L21^T on 64x64 tiles (lower tiles only).
#include <hip/hip_runtime.h>

#include "ba_internal.hpp"

namespace droid {

constexpr int NB = CHOL_NB;

__global__ void chol_damp_kernel(double* __restrict__ S, int n, int ld, double lm, double ep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double d = S[(size_t)i * ld + i];
    S[(size_t)i * ld + i] = d + (ep + lm * d);
  }
}

__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ S, int n, int ld, int k,
                                                         int* __restrict__ fail) {
  __shared__ double L[NB][NB + 1];
  __shared__ double X[NB][NB + 1];
  __shared__ double dg[NB];
  const int t = threadIdx.x;
  const int c0 = k * NB;
  const int wk = min(NB, n - c0);
  const int rb = k + blockIdx.x;
  const int r0 = rb * NB;
  const int r1 = min(r0 + NB, ld);

  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx / NB, j = idx % NB;
    double val = 0.0;
    if (i < wk && j <= i) val = S[(size_t)(c0 + i) * ld + c0 + j];
    L[i][j] = val;
  }
  // rows of this block that lie below the diagonal block
  const int s0 = max(r0, c0 + wk);
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx / NB, j = idx % NB;
    double val = 0.0;
    if (r0 + i >= s0 && r0 + i < r1 && j < wk) val = S[(size_t)(r0 + i) * ld + c0 + j];
    X[i][j] = val;
  }
  __syncthreads();

  // unblocked Cholesky of the diagonal block in LDS
  for (int j = 0; j < wk; j++) {
    double d = L[j][j];
    if (!(d > 0.0)) {
      if (t == 0 && blockIdx.x == 0) *fail = 1;
      d = 1.0;
    }
    d = sqrt(d);
    if (t == 0) dg[j] = d;
    if (t > j && t < wk) L[t][j] = L[t][j] / d;
    __syncthreads();
    const int i = t & 63;
    if (i > j && i < wk) {
      const double lij = L[i][j];
      for (int c = j + 1 + (t >> 6); c <= i; c += 4) L[i][c] -= lij * L[c][j];
    }
    __syncthreads();
  }

  // write the factor of the diagonal block (the row block that contains it)
  if (rb == k) {
    for (int idx = t; idx < NB * NB; idx += 256) {
      const int i = idx / NB, j = idx % NB;
      if (i < wk && j <= i) S[(size_t)(c0 + i) * ld + c0 + j] = (i == j) ? dg[i] : L[i][j];
    }
  }

  // X L^T = A  for the rows below: 4 lanes per row, each owning the columns c = part (mod 4)
  {
    const int i = t >> 2, part = t & 3;
    const bool active = (r0 + i >= s0) && (r0 + i < r1);
    for (int j = 0; j < wk; j++) {
      double s = 0.0;
      if (active)
        for (int c = part; c < j; c += 4) s += X[i][c] * L[j][c];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (active && part == (j & 3)) X[i][j] = (X[i][j] - s) / dg[j];
    }
  }
  __syncthreads();
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx / NB, j = idx % NB;
    if (r0 + i >= s0 && r0 + i < r1 && j < wk) S[(size_t)(r0 + i) * ld + c0 + j] = X[i][j];
  }
}

__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ S, int n, int ld,
                                                          int k) {
  __shared__ double Lr[NB][NB + 1];
  __shared__ double Lc[NB][NB + 1];
  const int bi = k + 1 + blockIdx.x, bj = k + 1 + blockIdx.y;
  if (bj > bi) return;
  const int t = threadIdx.x;
  const int p0 = k * NB;  // the panel is a full block whenever a trailing column block exists
  const int r0 = bi * NB, q0 = bj * NB;
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx / NB, j = idx % NB;
    Lr[i][j] = (r0 + i < ld) ? S[(size_t)(r0 + i) * ld + p0 + j] : 0.0;
    Lc[i][j] = (q0 + i < n) ? S[(size_t)(q0 + i) * ld + p0 + j] : 0.0;
  }
  __syncthreads();
  const int ti = t >> 4, tj = t & 15;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
  for (int j = 0; j < NB; j++) {
    double a[4], b[4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
      a[x] = Lr[4 * ti + x][j];
      b[x] = Lc[4 * tj + x][j];
    }
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
      for (int y = 0; y < 4; y++) acc[x][y] += a[x] * b[y];
  }
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const int r = r0 + 4 * ti + x;
    if (r >= ld) continue;
#pragma unroll
    for (int y = 0; y < 4; y++) {
      const int c = q0 + 4 * tj + y;
      if (c < n && c <= r) S[(size_t)r * ld + c] -= acc[x][y];
    }
  }
}

// One block column of the backward substitution L^T x = y (y = row n of S, consumed in place):
// x_k = L_kk^-T y_k, then y[0:c0] -= L[k rows, 0:c0]^T x_k.
__global__ __launch_bounds__(256) void chol_backsolve_kernel(double* __restrict__ S, int n, int ld,
                                                             int k, double* __restrict__ x) {
  __shared__ double L[NB][NB + 1];
  __shared__ double xk[NB];
  const int t = threadIdx.x;
  const int c0 = k * NB;
  const int wk = min(NB, n - c0);
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx / NB, j = idx % NB;
    L[i][j] = (i < wk && j <= i) ? S[(size_t)(c0 + i) * ld + c0 + j] : 0.0;
  }
  __syncthreads();
  if (t < 64) {
    double z = (t < wk) ? S[(size_t)n * ld + c0 + t] : 0.0;
    for (int i = wk - 1; i >= 0; i--) {
      const double xi = __shfl(z, i) / L[i][i];
      if (t == i) z = xi;
      if (t < i) z -= L[i][t] * xi;
    }
    xk[t] = z;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (t < wk) x[c0 + t] = xk[t];
    return;
  }
  const int c = (blockIdx.x - 1) * 256 + t;
  if (c < c0) {
    double s = 0.0;
    for (int r = 0; r < wk; r++) s += S[(size_t)(c0 + r) * ld + c] * xk[r];
    S[(size_t)n * ld + c] -= s;
  }
}

void launch_chol_factor(double* sys, int n, int ld, double lm, double ep, int* fail_flag,
                        hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(chol_damp_kernel, dim3((n + 255) / 256), dim3(256), 0, s, sys, n, ld, lm, ep);
  const int nb = (n + NB - 1) / NB;
  const int nrb = (ld + NB - 1) / NB;
  for (int k = 0; k < nb; k++) {
    hipLaunchKernelGGL(chol_panel_kernel, dim3(nrb - k), dim3(256), 0, s, sys, n, ld, k, fail_flag);
    if (k + 1 < nb)
      hipLaunchKernelGGL(chol_update_kernel, dim3(nrb - k - 1, nb - k - 1), dim3(256), 0, s, sys, n,
                         ld, k);
  }
}

void launch_chol_backsolve(double* sys, int n, int ld, double* x, hipStream_t s) {
  const int nb = (n + NB - 1) / NB;
  for (int k = nb - 1; k >= 0; k--) {
    const int c0 = k * NB;
    hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1 + (c0 + 255) / 256), dim3(256), 0, s, sys, n,
                       ld, k, x);
  }
}

void launch_chol_solve(double* sys, int n, int ld, double lm, double ep, double* x, int* fail_flag,
                       hipStream_t s) {
  if (n <= 0) return;
  launch_chol_factor(sys, n, ld, lm, ep, fail_flag, s);
  launch_chol_backsolve(sys, n, ld, x, s);
}

// helper for droid_chol_solve: pack (A, b) into the augmented layout
__global__ void chol_pack_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                 double* __restrict__ S, int n, int ld) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)ld * ld) return;
  const int r = (int)(idx / ld), c = (int)(idx % ld);
  double val = 0.0;
  if (r < n && c < n) val = A[(size_t)r * n + c];
  else if (r == n && c < n) val = b[c];
  S[idx] = val;
}

void launch_chol_pack(const double* A, const double* b, double* S, int n, int ld, hipStream_t s) {
  const size_t tot = (size_t)ld * ld;
  hipLaunchKernelGGL(chol_pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, b,
                     S, n, ld);
}

}  // namespace droid
