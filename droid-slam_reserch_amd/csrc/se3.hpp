// se3.hpp -- unit-quaternion SE3 device helpers and the per-pixel projective geometry shared by
// the BA and geometry kernels.  Restates the arithmetic of
// /root/reference/src/droid_kernels.cu:58-175 (group ops) and :290-354 (projection + Jacobians).
#pragma once
#include <hip/hip_runtime.h>

#define DROID_MIN_DEPTH 0.25f  // droid_kernels.cu:26

namespace droid {

struct Intr {
  float fx, fy, cx, cy;
};

struct Rel {       // relative transform of one edge
  float t[3];
  float q[4];
};

__device__ __forceinline__ void act_so3(const float* q, const float* X, float* Y) {  // dk:58-68
  float uv0 = 2.0f * (q[1] * X[2] - q[2] * X[1]);
  float uv1 = 2.0f * (q[2] * X[0] - q[0] * X[2]);
  float uv2 = 2.0f * (q[0] * X[1] - q[1] * X[0]);
  Y[0] = X[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
  Y[1] = X[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
  Y[2] = X[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}

// Y = Adj(T)^T X, dk:79-94
__device__ __forceinline__ void adj_se3(const float* t, const float* q, const float* X, float* Y) {
  float qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  act_so3(qinv, &X[0], &Y[0]);
  act_so3(qinv, &X[3], &Y[3]);
  float u[3], v[3];
  u[0] = t[2] * X[1] - t[1] * X[2];
  u[1] = t[0] * X[2] - t[2] * X[0];
  u[2] = t[1] * X[0] - t[0] * X[1];
  act_so3(qinv, u, v);
  Y[3] += v[0];
  Y[4] += v[1];
  Y[5] += v[2];
}

// Tij = Tj * Ti^-1, dk:96-107; stereo pair (ix == jx): fixed baseline, dk:219-229.
// Evaluated in fp64 and rounded once: t = tj - R ti is a difference of two vectors of the size of the
// trajectory (12 m at 256 keyframes against a baseline of 0.1 m), which in fp32 costs two digits of the
// relative translation -- an error shared by ALL pixels of the edge, so it does not average out in the
// normal equations (measured: reduced rhs 4e-6 -> see DESIGN.md section 5).  One evaluation per edge and
// workgroup; the per-pixel arithmetic stays fp32 like the reference's.
template <bool STEREO = true>
__device__ __forceinline__ Rel rel_pose(const float* __restrict__ poses, int ix, int jx) {
  Rel r;
  if (STEREO && ix == jx) {
    r.t[0] = -0.1f; r.t[1] = 0.f; r.t[2] = 0.f;
    r.q[0] = 0.f; r.q[1] = 0.f; r.q[2] = 0.f; r.q[3] = 1.f;
    return r;
  }
  const float* pi = poses + 7 * (size_t)ix;
  const float* pj = poses + 7 * (size_t)jx;
  const double ti[3] = {pi[0], pi[1], pi[2]};
  const double qi[4] = {pi[3], pi[4], pi[5], pi[6]};
  const double tj[3] = {pj[0], pj[1], pj[2]};
  const double qj[4] = {pj[3], pj[4], pj[5], pj[6]};
  double q[4];
  q[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  q[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  q[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  q[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  // rt = q * ti (dk:58-68 in double)
  const double uv0 = 2.0 * (q[1] * ti[2] - q[2] * ti[1]);
  const double uv1 = 2.0 * (q[2] * ti[0] - q[0] * ti[2]);
  const double uv2 = 2.0 * (q[0] * ti[1] - q[1] * ti[0]);
  const double rt0 = ti[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
  const double rt1 = ti[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
  const double rt2 = ti[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
  r.t[0] = (float)(tj[0] - rt0);
  r.t[1] = (float)(tj[1] - rt1);
  r.t[2] = (float)(tj[2] - rt2);
#pragma unroll
  for (int n = 0; n < 4; n++) r.q[n] = (float)q[n];
  return r;
}

// The same relative transform as a rotation MATRIX + translation: x' = R x + d t costs 12 FMAs per pixel
// against ~30 operations of the quaternion sandwich (act_so3), and Adj^T needs only R^T and t.
// R = I + 2 w [v]x + 2 [v]x^2 is act_so3's own expression (dk:58-68) collected per matrix entry, so it
// agrees with it for the not-renormalised quaternions of the reference too.  S = double: per-edge values for
// the residual evaluation of the linearisation; S = float: per-pixel Jacobian arithmetic.
template <typename S>
struct RelMat {
  S R[9];  // row-major
  S t[3];
};

template <typename S, bool STEREO = true>
__device__ __forceinline__ RelMat<S> rel_pose_mat(const float* __restrict__ poses, int ix, int jx) {
  RelMat<S> r;
  if (STEREO && ix == jx) {
#pragma unroll
    for (int n = 0; n < 9; n++) r.R[n] = (S)((n % 4) == 0 ? 1.0 : 0.0);
    r.t[0] = (S)(-0.1f); r.t[1] = (S)0.0; r.t[2] = (S)0.0;   // the reference's fp32 literal (dk:221)
    return r;
  }
  const float* pi = poses + 7 * (size_t)ix;
  const float* pj = poses + 7 * (size_t)jx;
  const double ti[3] = {pi[0], pi[1], pi[2]};
  const double qi[4] = {pi[3], pi[4], pi[5], pi[6]};
  const double tj[3] = {pj[0], pj[1], pj[2]};
  const double qj[4] = {pj[3], pj[4], pj[5], pj[6]};
  const double x = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  const double y = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  const double z = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  const double w = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  double R[9];
  R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - z * w);       R[2] = 2.0 * (x * z + y * w);
  R[3] = 2.0 * (x * y + z * w);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - x * w);
  R[6] = 2.0 * (x * z - y * w);       R[7] = 2.0 * (y * z + x * w);       R[8] = 1.0 - 2.0 * (x * x + y * y);
#pragma unroll
  for (int n = 0; n < 9; n++) r.R[n] = (S)R[n];
#pragma unroll
  for (int n = 0; n < 3; n++) r.t[n] = (S)(tj[n] - (R[3 * n] * ti[0] + R[3 * n + 1] * ti[1] + R[3 * n + 2] * ti[2]));
  return r;
}

// Y = Adj(T)^T X (dk:79-94) from the matrix form: Y[0:3] = R^T X[0:3], Y[3:6] = R^T (X[3:6] + X[0:3] x t)
template <typename S, typename V>
__device__ __forceinline__ void adjT_mat(const S* R, const S* t, const V* X, V* Y) {
  const V u0 = X[3] + (X[1] * (V)t[2] - X[2] * (V)t[1]);
  const V u1 = X[4] + (X[2] * (V)t[0] - X[0] * (V)t[2]);
  const V u2 = X[5] + (X[0] * (V)t[1] - X[1] * (V)t[0]);
#pragma unroll
  for (int n = 0; n < 3; n++) {
    Y[n] = (V)R[n] * X[0] + (V)R[3 + n] * X[1] + (V)R[6 + n] * X[2];
    Y[3 + n] = (V)R[n] * u0 + (V)R[3 + n] * u1 + (V)R[6 + n] * u2;
  }
}

// relSE3 without the stereo special case (projmap / frame_distance / depth_filter, dk:480, :582)
__device__ __forceinline__ Rel rel_pose_plain(const float* __restrict__ poses, int ix, int jx) {
  return rel_pose<false>(poses, ix, jx);
}

// Transform the back-projected pixel (u,v,disp) into frame j.  dk:290-296
__device__ __forceinline__ void transform_pixel(const Intr& K, const Rel& T, float u, float v,
                                                float disp, float* Xj) {
  float Xi[3];
  Xi[0] = (u - K.cx) / K.fx;
  Xi[1] = (v - K.cy) / K.fy;
  Xi[2] = 1.f;
  act_so3(T.q, Xi, Xj);
  Xj[3] = disp;
  Xj[0] += disp * T.t[0];
  Xj[1] += disp * T.t[1];
  Xj[2] += disp * T.t[2];
}

// Per-pixel linearisation of one edge: both residual rows.  Jj rows have a structural zero each
// (Jj_u[1] = 0, Jj_v[0] = 0; dk:313, :345).
struct PixLin {
  float Ju[6], Jv[6];  // d(proj)/d(xi_j), u and v rows                 dk:312-317, :345-350
  float Jzu, Jzv;      // d(proj)/d(disp_i)                             dk:319, :352
  float ru, rv;        // residuals target - proj                       dk:307-308
  float valid;         // 0 when Z < MIN_DEPTH (weights are zeroed)     dk:302-306
};

// Jacobian rows from the transformed point (x, y, 1/Z = d, disparity h) and the edge's translation
__device__ __forceinline__ void pix_jacobians(const Intr& K, float x, float y, float d, float h, float t0,
                                              float t1, float t2, PixLin& L) {
  const float d2 = d * d;
  L.Ju[0] = K.fx * (h * d);
  L.Ju[1] = 0.f;
  L.Ju[2] = K.fx * (-x * h * d2);
  L.Ju[3] = K.fx * (-x * y * d2);
  L.Ju[4] = K.fx * (1.f + x * x * d2);
  L.Ju[5] = K.fx * (-y * d);
  L.Jzu = K.fx * (t0 * d - t2 * (x * d2));
  L.Jv[0] = 0.f;
  L.Jv[1] = K.fy * (h * d);
  L.Jv[2] = K.fy * (-y * h * d2);
  L.Jv[3] = K.fy * (-1.f - y * y * d2);
  L.Jv[4] = K.fy * (x * y * d2);
  L.Jv[5] = K.fy * (x * d);
  L.Jzv = K.fy * (t1 * d - t2 * (y * d2));
}

__device__ __forceinline__ PixLin linearize_pixel(const Intr& K, const Rel& T, float u, float v,
                                                  float disp, float tu, float tv) {
  float Xj[4];
  transform_pixel(K, T, u, v, disp, Xj);
  const float x = Xj[0], y = Xj[1], h = Xj[3];
  const bool bad = Xj[2] < DROID_MIN_DEPTH;
  const float d = bad ? 0.f : 1.0f / Xj[2];
  PixLin L;
  L.valid = bad ? 0.f : 1.f;
  L.ru = tu - (K.fx * d * x + K.cx);
  L.rv = tv - (K.fy * d * y + K.cy);
  pix_jacobians(K, x, y, d, h, T.t[0], T.t[1], T.t[2], L);
  return L;
}

// Jacobians only (E rows of the Schur complement and of the back-substitution), fp32 matrix form:
// 12 FMAs for the transform.  ru / rv are not set.
// X0, X1: the back-projected pixel ((u - cx) / fx, (v - cy) / fy), formed once per pixel by the caller
__device__ __forceinline__ PixLin jacobians_pixel(const Intr& K, const float* R, const float* t, float X0, float X1,
                                                  float disp) {
  const float x = R[0] * X0 + R[1] * X1 + R[2] + disp * t[0];
  const float y = R[3] * X0 + R[4] * X1 + R[5] + disp * t[1];
  const float z = R[6] * X0 + R[7] * X1 + R[8] + disp * t[2];
  const bool bad = z < DROID_MIN_DEPTH;
  const float d = bad ? 0.f : 1.0f / z;
  PixLin L;
  L.valid = bad ? 0.f : 1.f;
  L.ru = 0.f;
  L.rv = 0.f;
  pix_jacobians(K, x, y, d, disp, t[0], t[1], t[2], L);
  return L;
}

// Linearisation with the reprojection and the residual evaluated in fp64 (fp64 FMAs issue at the fp32
// rate on gfx950): in fp32 the projected coordinate (up to 64..128 px) carries 1e-5 px of rounding, i.e.
// 3e-5 of a typical 0.3 px residual, which the depth back-substitution dz = Q (w - E^T dx) hands through
// to weakly observed pixels (measured: 1e-4 of a disparity that moves by 1.4 in two iterations).  The
// Jacobians are formed in fp32 from the rounded point, like the reference's.
// X0, X1: the back-projected pixel in fp64, ((double)u - cx) / fx etc., formed once per pixel by the caller (two fp64
// divisions that would otherwise be repeated for every edge of the slot)
__device__ __forceinline__ PixLin linearize_pixel_d(const Intr& K, const double* R, const double* t, double X0, double X1,
                                                    float disp, float tu, float tv) {
  const double dd = (double)disp;
  const double x = R[0] * X0 + R[1] * X1 + R[2] + dd * t[0];
  const double y = R[3] * X0 + R[4] * X1 + R[5] + dd * t[1];
  const double z = R[6] * X0 + R[7] * X1 + R[8] + dd * t[2];
  const bool bad = z < (double)DROID_MIN_DEPTH;
  // 1/z: fp32 reciprocal refined by two Newton steps in fp64 (relative error 1e-7 -> 1e-14 -> 1e-28)
  double d = (double)(1.0f / (float)z);
  d = d * (2.0 - z * d);
  d = d * (2.0 - z * d);
  if (bad) d = 0.0;
  PixLin L;
  L.valid = bad ? 0.f : 1.f;
  L.ru = (float)((double)tu - ((double)K.fx * d * x + (double)K.cx));
  L.rv = (float)((double)tv - ((double)K.fy * d * y + (double)K.cy));
  pix_jacobians(K, (float)x, (float)y, (float)d, disp, (float)t[0], (float)t[1], (float)t[2], L);
  return L;
}

// SO3 / SE3 exponential and the retraction T <- exp(xi) * T.  dk:110-175, :877-895
__device__ __forceinline__ void exp_so3(const float* phi, float* q) {
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float theta_p4 = theta_sq * theta_sq;
  float theta = sqrtf(theta_sq);
  float imag, real;
  if (theta_sq < 1e-8f) {
    imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_p4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_p4;
  } else {
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  q[0] = imag * phi[0];
  q[1] = imag * phi[1];
  q[2] = imag * phi[2];
  q[3] = real;
}

__device__ __forceinline__ void cross_inplace(const float* a, float* b) {
  float x0 = a[1] * b[2] - a[2] * b[1];
  float x1 = a[2] * b[0] - a[0] * b[2];
  float x2 = a[0] * b[1] - a[1] * b[0];
  b[0] = x0; b[1] = x1; b[2] = x2;
}

__device__ __forceinline__ void exp_se3(const float* xi, float* t, float* q) {
  exp_so3(xi + 3, q);
  float tau[3] = {xi[0], xi[1], xi[2]};
  float phi[3] = {xi[3], xi[4], xi[5]};
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float theta = sqrtf(theta_sq);
  t[0] = tau[0]; t[1] = tau[1]; t[2] = tau[2];
  if (theta > 1e-4f) {
    float a = (1.f - cosf(theta)) / theta_sq;
    cross_inplace(phi, tau);
    t[0] += a * tau[0]; t[1] += a * tau[1]; t[2] += a * tau[2];
    float b = (theta - sinf(theta)) / (theta * theta_sq);
    cross_inplace(phi, tau);
    t[0] += b * tau[0]; t[1] += b * tau[1]; t[2] += b * tau[2];
  }
}

__device__ __forceinline__ void retr_se3(const float* xi, const float* t, const float* q, float* t1,
                                         float* q1) {
  float dt[3] = {0.f, 0.f, 0.f};
  float dq[4] = {0.f, 0.f, 0.f, 1.f};
  exp_se3(xi, dt, dq);
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  act_so3(dq, t, t1);
  t1[0] += dt[0]; t1[1] += dt[1]; t1[2] += dt[2];
}

}  // namespace droid
