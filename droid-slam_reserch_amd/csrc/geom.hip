// geom.hip -- the geometry operators of the droid_backends ABI besides `ba`:
// frame_distance, projmap, iproj, depth_filter (/root/reference/src/droid_kernels.cu:427-850).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "se3.hpp"

namespace droid {

__device__ __forceinline__ float wave_sum(float s) {
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  return s;
}

// frame_distance_kernel dk:518-657: mean induced flow, beta * full + (1-beta) * translation-only.
// One workgroup per edge; three wave-shuffle reductions instead of three 256-float LDS trees.
__global__ __launch_bounds__(256) void frame_distance_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intrinsics, const int64_t* __restrict__ ii,
    const int64_t* __restrict__ jj, float* __restrict__ dist, int nbuf, int H, int W, float beta) {
  __shared__ float red[3][4];
  const int e = blockIdx.x, tid = threadIdx.x;
  const int64_t i64 = ii[e], j64 = jj[e];
  if (i64 < 0 || i64 >= nbuf || j64 < 0 || j64 >= nbuf) {
    if (tid == 0) dist[e] = 1000.0f;
    return;
  }
  const int ix = (int)i64, jx = (int)j64;
  const Intr K = {intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  const Rel T = rel_pose_plain(poses, ix, jx);
  const int HW = H * W;
  float accum = 0.f, valid = 0.f, total = 0.f;
  for (int k = tid; k < HW; k += 256) {
    const float u = (float)(k % W), v = (float)(k / W);
    const float d0 = disps[(size_t)ix * HW + k];
    float Xj[4];
    transform_pixel(K, T, u, v, d0, Xj);
    float du = K.fx * (Xj[0] / Xj[2]) + K.cx - u;
    float dv = K.fy * (Xj[1] / Xj[2]) + K.cy - v;
    float d = sqrtf(du * du + dv * dv);
    total += beta;
    if (Xj[2] > DROID_MIN_DEPTH) {
      accum += beta * d;
      valid += beta;
    }
    const float X0 = (u - K.cx) / K.fx, X1 = (v - K.cy) / K.fy;
    Xj[0] = X0 + d0 * T.t[0];
    Xj[1] = X1 + d0 * T.t[1];
    Xj[2] = 1.f + d0 * T.t[2];
    du = K.fx * (Xj[0] / Xj[2]) + K.cx - u;
    dv = K.fy * (Xj[1] / Xj[2]) + K.cy - v;
    d = sqrtf(du * du + dv * dv);
    total += (1.f - beta);
    if (Xj[2] > DROID_MIN_DEPTH) {
      accum += (1.f - beta) * d;
      valid += (1.f - beta);
    }
  }
  accum = wave_sum(accum);
  valid = wave_sum(valid);
  total = wave_sum(total);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = accum;
    red[1][tid >> 6] = valid;
    red[2][tid >> 6] = total;
  }
  __syncthreads();
  if (tid == 0) {
    const float a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float vl = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float tt = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    dist[e] = (vl / (tt + 1e-8f) < 0.75f) ? 1000.0f : a / vl;  // dk:655
  }
}

// All ordered pairs (i, j), i, j < n, in ONE launch (`DepthVideo.distance()` with ii = None,
// depth_video.py:160-190; the candidate matrix of add_proximity_factors, factor_graph.py:318-326): a workgroup owns
// a source frame i and FD_JB target frames: the depth map of frame i (12 KB at 48 x 64) is fetched from memory once
// per (i, target block) and then served by L1 / L2 instead of once per pair, the relative poses of the block are
// formed once, and no index tensors exist.  Same arithmetic per pair as frame_distance_kernel (dk:518-657).
// dist[i * n + j] = distance i -> j.
constexpr int FD_JB = 32;    // targets per workgroup
__global__ __launch_bounds__(256) void frame_distance_matrix_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intrinsics,
    float* __restrict__ dist, int n, int H, int W, float beta) {
  __shared__ float Ts[FD_JB][8];
  __shared__ float red[FD_JB][2][4];
  __shared__ float tot[4];
  const int i = blockIdx.x, j0 = blockIdx.y * FD_JB, tid = threadIdx.x;
  const int nj = min(FD_JB, n - j0);
  const Intr K = {intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  const int HW = H * W;
  if (tid < nj) {
    const Rel T = rel_pose_plain(poses, i, j0 + tid);
#pragma unroll
    for (int c = 0; c < 3; c++) Ts[tid][c] = T.t[c];
#pragma unroll
    for (int c = 0; c < 4; c++) Ts[tid][3 + c] = T.q[c];
  }
  {  // `total` of dk:618-655 (beta + (1 - beta) per pixel, accumulated like the per-pair kernel): the same for all targets
    float total = 0.f;
    for (int k = tid; k < HW; k += 256) {
      total += beta;
      total += (1.f - beta);
    }
    total = wave_sum(total);
    if ((tid & 63) == 0) tot[tid >> 6] = total;
  }
  __syncthreads();
  for (int jj = 0; jj < nj; jj++) {
    Rel T;
#pragma unroll
    for (int c = 0; c < 3; c++) T.t[c] = Ts[jj][c];
#pragma unroll
    for (int c = 0; c < 4; c++) T.q[c] = Ts[jj][3 + c];
    float accum = 0.f, valid = 0.f;
    for (int k = tid; k < HW; k += 256) {   // the depth map stays in L1 / L2 across the targets of the block
      const float u = (float)(k % W), v = (float)(k / W);
      const float d0 = disps[(size_t)i * HW + k];
      float Xj[4];
      transform_pixel(K, T, u, v, d0, Xj);
      float du = K.fx * (Xj[0] / Xj[2]) + K.cx - u;
      float dv = K.fy * (Xj[1] / Xj[2]) + K.cy - v;
      float d = sqrtf(du * du + dv * dv);
      if (Xj[2] > DROID_MIN_DEPTH) {
        accum += beta * d;
        valid += beta;
      }
      const float X0 = (u - K.cx) / K.fx, X1 = (v - K.cy) / K.fy;
      Xj[0] = X0 + d0 * T.t[0];
      Xj[1] = X1 + d0 * T.t[1];
      Xj[2] = 1.f + d0 * T.t[2];
      du = K.fx * (Xj[0] / Xj[2]) + K.cx - u;
      dv = K.fy * (Xj[1] / Xj[2]) + K.cy - v;
      d = sqrtf(du * du + dv * dv);
      if (Xj[2] > DROID_MIN_DEPTH) {
        accum += (1.f - beta) * d;
        valid += (1.f - beta);
      }
    }
    accum = wave_sum(accum);
    valid = wave_sum(valid);
    if ((tid & 63) == 0) {
      red[jj][0][tid >> 6] = accum;
      red[jj][1][tid >> 6] = valid;
    }
  }
  __syncthreads();
  if (tid < nj) {
    const float a = red[tid][0][0] + red[tid][0][1] + red[tid][0][2] + red[tid][0][3];
    const float vl = red[tid][1][0] + red[tid][1][1] + red[tid][1][2] + red[tid][1][3];
    const float tt = tot[0] + tot[1] + tot[2] + tot[3];
    dist[(size_t)i * n + j0 + tid] = (vl / (tt + 1e-8f) < 0.75f) ? 1000.0f : a / vl;  // dk:655
  }
}

// projmap_kernel dk:427-516
__global__ __launch_bounds__(256) void projmap_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intrinsics, const int64_t* __restrict__ ii,
    const int64_t* __restrict__ jj, float* __restrict__ coords, float* __restrict__ valid, int nbuf,
    int H, int W) {
  const int e = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W;
  if (k >= HW) return;
  const int64_t i64 = ii[e], j64 = jj[e];
  if (i64 < 0 || i64 >= nbuf || j64 < 0 || j64 >= nbuf) {  // edge with a bad index: zeros, no fault
    float* c = coords + ((size_t)e * HW + k) * 3;
    c[0] = c[1] = c[2] = 0.f;
    valid[(size_t)e * HW + k] = 0.f;
    return;
  }
  const Intr K = {intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  const Rel T = rel_pose_plain(poses, (int)i64, (int)j64);
  const float u = (float)(k % W), v = (float)(k / W);
  float Xj[4];
  transform_pixel(K, T, u, v, disps[(size_t)i64 * HW + k], Xj);
  float cu = u, cv = v;
  if (Xj[2] > 0.01f) {
    cu = K.fx * (Xj[0] / Xj[2]) + K.cx;
    cv = K.fy * (Xj[1] / Xj[2]) + K.cy;
  }
  float* c = coords + ((size_t)e * HW + k) * 3;
  c[0] = cu;
  c[1] = cv;
  c[2] = 0.f;
  valid[(size_t)e * HW + k] = (Xj[2] > DROID_MIN_DEPTH) ? 1.0f : 0.0f;
}


// reproject + motion features of the update operator, one pass:
//   depth_video.py:150-158 `reproject` -> geom/projective_ops.py:96-125 `projective_transform`
//   (back-projection with the SOURCE frame's intrinsics :99, Gij = Gj * Gi^-1 :102, stereo edges ii == jj use
//   the fixed baseline (-0.1, 0, 0) :105, projection with the TARGET frame's intrinsics where a depth below
//   0.5 * MIN_DEPTH is replaced by 1 :46-51, valid = depth > MIN_DEPTH = 0.2 :113 -- these are the Python-side
//   constants, not the 0.25 of droid_kernels.cu), then factor_graph.py:203-205:
//   motn = clamp(cat(coords1 - coords0, target - coords1), -64, 64) laid out [E,4,H,W].
// One thread per pixel; coords/valid/motn rows are written with unit stride (motn planes are H*W apart).
__global__ __launch_bounds__(256) void reproject_motion_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intrinsics,
    int intr_stride, const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
    const float* __restrict__ target, float* __restrict__ coords, float* __restrict__ valid,
    float* __restrict__ motn, int nbuf, int H, int W) {
  const int e = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W;
  if (k >= HW) return;
  const int64_t i64 = ii[e], j64 = jj[e];
  const size_t o = (size_t)e * HW + k;
  if (i64 < 0 || i64 >= nbuf || j64 < 0 || j64 >= nbuf) {  // edge with a bad index: zeros, no fault
    reinterpret_cast<float2*>(coords)[o] = make_float2(0.f, 0.f);
    valid[o] = 0.f;
    if (motn)
      for (int c = 0; c < 4; c++) motn[((size_t)e * 4 + c) * HW + k] = 0.f;
    return;
  }
  const float* ki = intrinsics + (size_t)i64 * intr_stride;
  const float* kj = intrinsics + (size_t)j64 * intr_stride;
  const Intr Ki = {ki[0], ki[1], ki[2], ki[3]};
  const Rel T = rel_pose<true>(poses, (int)i64, (int)j64);
  const float u = (float)(k % W), v = (float)(k / W);
  float Xj[4];
  transform_pixel(Ki, T, u, v, disps[(size_t)i64 * HW + k], Xj);
  const float Z = Xj[2];
  const float d = 1.0f / ((Z < 0.5f * 0.2f) ? 1.0f : Z);
  const float x = kj[0] * (Xj[0] * d) + kj[2];
  const float y = kj[1] * (Xj[1] * d) + kj[3];
  reinterpret_cast<float2*>(coords)[o] = make_float2(x, y);
  valid[o] = (Z > 0.2f) ? 1.0f : 0.0f;  // the source point has depth 1 > MIN_DEPTH by construction
  if (motn) {
    const float2 tg = reinterpret_cast<const float2*>(target)[o];
    auto clamp64 = [](float a) { return fminf(fmaxf(a, -64.0f), 64.0f); };
    float* m = motn + (size_t)e * 4 * HW + k;
    m[0] = clamp64(x - u);
    m[HW] = clamp64(y - v);
    m[2 * (size_t)HW] = clamp64(tg.x - x);
    m[3 * (size_t)HW] = clamp64(tg.y - y);
  }
}

// iproj_kernel dk:779-850
__global__ __launch_bounds__(256) void iproj_kernel(const float* __restrict__ poses,
                                                    const float* __restrict__ disps,
                                                    const float* __restrict__ intrinsics,
                                                    float* __restrict__ points, int H, int W) {
  const int f = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W;
  if (k >= HW) return;
  const Intr K = {intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  Rel T;
  for (int n = 0; n < 3; n++) T.t[n] = poses[7 * (size_t)f + n];
  for (int n = 0; n < 4; n++) T.q[n] = poses[7 * (size_t)f + 3 + n];
  float Xj[4];
  transform_pixel(K, T, (float)(k % W), (float)(k / W), disps[(size_t)f * HW + k], Xj);
  float* p = points + ((size_t)f * HW + k) * 3;
  p[0] = Xj[0] / Xj[3];
  p[1] = Xj[1] / Xj[3];
  p[2] = Xj[2] / Xj[3];
}

// depth_filter_kernel dk:661-775: count of the 6 temporal neighbours whose depth agrees.
// The reference adds with atomicAdd over a (num,6,tiles) grid; here one thread owns a pixel and
// loops over the neighbours, so the count is a plain register sum.
__global__ __launch_bounds__(256) void depth_filter_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intrinsics, const int64_t* __restrict__ inds,
    const float* __restrict__ thresh, float* __restrict__ counter, int nbuf, int H, int W) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W;
  if (k >= HW) return;
  const int ix = (int)inds[b];
  float cnt = 0.f;
  if (ix >= 0 && ix < nbuf) {
    const Intr K = {intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
    const float t = thresh[b];
    const float ui = (float)(k % W), vi = (float)(k / W);
    const float di = disps[(size_t)ix * HW + k];
    for (int nb = 0; nb < 6; nb++) {
      const int jx = (nb < 3) ? ix - nb - 1 : ix + nb;  // dk:695
      if (jx < 0 || jx >= nbuf) continue;
      const Rel T = rel_pose_plain(poses, ix, jx);
      float Xj[4];
      transform_pixel(K, T, ui, vi, di, Xj);
      const float uj = K.fx * (Xj[0] / Xj[2]) + K.cx;
      const float vj = K.fy * (Xj[1] / Xj[2]) + K.cy;
      const float dj = Xj[3] / Xj[2];
      const int u0 = (int)floorf(uj), v0 = (int)floorf(vj);
      if (u0 >= 0 && v0 >= 0 && u0 < W - 1 && v0 < H - 1) {
        const float* dp = disps + (size_t)jx * HW;
        const float d00 = dp[(v0 + 0) * W + u0 + 0], d01 = dp[(v0 + 0) * W + u0 + 1];
        const float d10 = dp[(v0 + 1) * W + u0 + 0], d11 = dp[(v0 + 1) * W + u0 + 1];
        const float idj = 1.0f / dj;
        if (fabsf(idj - 1.0f / d00) < t) cnt += 1.f;
        else if (fabsf(idj - 1.0f / d01) < t) cnt += 1.f;
        else if (fabsf(idj - 1.0f / d10) < t) cnt += 1.f;
        else if (fabsf(idj - 1.0f / d11) < t) cnt += 1.f;
      }
    }
  }
  counter[(size_t)b * HW + k] = cnt;
}

void launch_frame_distance(const float* poses, const float* disps, const float* intr,
                           const int64_t* ii, const int64_t* jj, int E, int nbuf, int H, int W,
                           float beta, float* dist, hipStream_t s) {
  hipLaunchKernelGGL(frame_distance_kernel, dim3(E), dim3(256), 0, s, poses, disps, intr, ii, jj,
                     dist, nbuf, H, W, beta);
}

void launch_frame_distance_matrix(const float* poses, const float* disps, const float* intr, int n, int H, int W,
                                  float beta, float* dist, hipStream_t s) {
  hipLaunchKernelGGL(frame_distance_matrix_kernel, dim3(n, (n + FD_JB - 1) / FD_JB), dim3(256), 0, s, poses, disps,
                     intr, dist, n, H, W, beta);
}

void launch_projmap(const float* poses, const float* disps, const float* intr, const int64_t* ii,
                    const int64_t* jj, int E, int nbuf, int H, int W, float* coords, float* valid,
                    hipStream_t s) {
  hipLaunchKernelGGL(projmap_kernel, dim3((H * W + 255) / 256, E), dim3(256), 0, s, poses, disps,
                     intr, ii, jj, coords, valid, nbuf, H, W);
}

void launch_reproject_motion(const float* poses, const float* disps, const float* intr, int intr_stride,
                             const int64_t* ii, const int64_t* jj, const float* target, int E, int nbuf, int H,
                             int W, float* coords, float* valid, float* motn, hipStream_t s) {
  hipLaunchKernelGGL(reproject_motion_kernel, dim3((H * W + 255) / 256, E), dim3(256), 0, s, poses, disps, intr,
                     intr_stride, ii, jj, target, coords, valid, motn, nbuf, H, W);
}

void launch_iproj(const float* poses, const float* disps, const float* intr, int nm, int H, int W,
                  float* points, hipStream_t s) {
  hipLaunchKernelGGL(iproj_kernel, dim3((H * W + 255) / 256, nm), dim3(256), 0, s, poses, disps,
                     intr, points, H, W);
}

void launch_depth_filter(const float* poses, const float* disps, const float* intr,
                         const int64_t* ix, const float* thresh, int num, int nbuf, int H, int W,
                         float* counter, hipStream_t s) {
  hipLaunchKernelGGL(depth_filter_kernel, dim3((H * W + 255) / 256, num), dim3(256), 0, s, poses,
                     disps, intr, ix, thresh, counter, nbuf, H, W);
}

}  // namespace droid
