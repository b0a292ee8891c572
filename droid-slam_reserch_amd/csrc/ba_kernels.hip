// ba_kernels.hip -- dense bundle adjustment on gfx950: structure prep, linearisation, reduced
// camera system assembly (Schur complement on fp32 MFMA), depth back-substitution, retraction.
//
// What is computed follows /root/reference/src/droid_kernels.cu (ba_cuda :1314-1434 and the
// kernels it launches); how it is computed does not:
//   * everything is organised by SOURCE frame ("depth slot"): one workgroup owns (slot, pixel
//     chunk) and walks that frame's outgoing edges (CSR built on the device once per call), so the
//     reference's five host-side accum_cuda passes (:948-998) disappear: C, w and the self row
//     Ei are register sums that are stored once.
//   * per edge only Hjj (21) and vj (6) are reduced; Hii/Hij/Hji/vi follow from the per-edge
//     constant adjoint Ji = -Adj^T Jj (:325-326) in the assemble kernel, in fp64.
//   * the reduced camera system is a dense fp64 (6P+1)^2 matrix on the device (row 6P = rhs);
//     no host round trip, no sparse assembly (:1117-1219, :1222-1311).
//   * S = E C^-1 E^T is a batched SYRK on v_mfma_f32_16x16x4_f32, one wave per 16x16 tile.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ba_internal.hpp"
#include "se3.hpp"

namespace droid {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------

// Exclusive scan of data[0..L) in place by one workgroup; returns the total in *total (LDS).
// lds: blockDim.x ints.
__device__ void block_exscan(int* data, int L, int* lds, int* total) {
  const int T = blockDim.x, t = threadIdx.x;
  const int per = (L + T - 1) / T;
  const int b = t * per;
  int s = 0;
  for (int k = 0; k < per; k++)
    if (b + k < L) s += data[b + k];
  lds[t] = s;
  __syncthreads();
  for (int off = 1; off < T; off <<= 1) {  // Hillis-Steele inclusive scan
    int v = (t >= off) ? lds[t - off] : 0;
    __syncthreads();
    lds[t] += v;
    __syncthreads();
  }
  int run = lds[t] - s;
  for (int k = 0; k < per; k++)
    if (b + k < L) {
      int v = data[b + k];
      data[b + k] = run;
      run += v;
    }
  if (t == T - 1) *total = lds[t];
  __syncthreads();
}

// Sum 32 per-lane values over the 64 lanes of a wave with a halving butterfly (31 exchanges
// instead of 32 x 6).  On return lane l holds the wave total of value
// idx = bit5 + 2*bit4 + 4*bit3 + 8*bit2 + 16*bit1 of l (both lanes of a bit0 pair hold it).
__device__ __forceinline__ float wave_reduce32(float (&v)[32]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const bool hi = lane & 32;
    float keep = hi ? v[2 * i + 1] : v[2 * i];
    float send = hi ? v[2 * i] : v[2 * i + 1];
    v[i] = keep + __shfl_xor(send, 32);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const bool hi = lane & 16;
    float keep = hi ? v[2 * i + 1] : v[2 * i];
    float send = hi ? v[2 * i] : v[2 * i + 1];
    v[i] = keep + __shfl_xor(send, 16);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const bool hi = lane & 8;
    float keep = hi ? v[2 * i + 1] : v[2 * i];
    float send = hi ? v[2 * i] : v[2 * i + 1];
    v[i] = keep + __shfl_xor(send, 8);
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const bool hi = lane & 4;
    float keep = hi ? v[2 * i + 1] : v[2 * i];
    float send = hi ? v[2 * i] : v[2 * i + 1];
    v[i] = keep + __shfl_xor(send, 4);
  }
  {
    const bool hi = lane & 2;
    float keep = hi ? v[1] : v[0];
    float send = hi ? v[0] : v[1];
    v[0] = keep + __shfl_xor(send, 2);
  }
  return v[0] + __shfl_xor(v[0], 1);
}

__device__ __forceinline__ int reduce32_index(int lane) {
  return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) |
         (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4);
}

// ------------------------------------------------------------------------------------------
// prep: depth slots, CSR of edges by source frame, Schur entry lists and tile work list.
// One workgroup; O(E + nbuf) work, run once per ba call (the graph is fixed across iterations).
// Restates the index bookkeeping of ba_cuda :1336-1344 and schur_block :1232-1272.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ba_prep_kernel(BaView v, const int64_t* __restrict__ ii,
                                                        const int64_t* __restrict__ jj) {
  __shared__ int lds[1024];
  __shared__ int tot;
  const int t = threadIdx.x, T = blockDim.x;
  const int E = v.E, nbuf = v.nbuf;
  const int w0 = max(v.t0, v.own0), w1 = min(v.t1, v.own1);

  if (t == 0) {
    v.hdr[HDR_STATUS] = 0;
    v.hdr[HDR_CHOL_FAIL] = 0;
  }
  for (int f = t; f < nbuf; f += T) v.slot_of[f] = 0;
  __syncthreads();
  for (int e = t; e < E; e += T) {
    const int64_t i = ii[e], j = jj[e];
    if (i < 0 || i >= nbuf || j < 0 || j >= nbuf)
      atomicOr(&v.hdr[HDR_STATUS], STATUS_BAD_INDEX);
    else
      v.slot_of[i] = 1;
  }
  if (v.motion_only) {  // no depth slots (:1385-1392); only the index check above is needed
    if (t == 0) {
      v.hdr[HDR_M] = 0;
      v.hdr[HDR_NENT] = 0;
      v.hdr[HDR_NWORK] = 0;
    }
    return;
  }
  for (int f = w0 + t; f < w1; f += T)
    if (f >= 0 && f < nbuf) v.slot_of[f] = 1;
  __syncthreads();
  // slot numbers = exclusive scan of the presence flags (sorted unique, like torch::_unique)
  for (int f = t; f < nbuf; f += T) v.cursor[f] = v.slot_of[f];
  __syncthreads();
  block_exscan(v.cursor, nbuf, lds, &tot);
  const int M = tot;
  for (int f = t; f < nbuf; f += T) {
    if (v.slot_of[f]) {
      v.slot_of[f] = v.cursor[f];
      v.kx[v.cursor[f]] = f;
    } else {
      v.slot_of[f] = -1;
    }
  }
  __syncthreads();
  if (t == 0) {
    v.hdr[HDR_M] = M;
    if (M != v.M) atomicOr(&v.hdr[HDR_STATUS], STATUS_ETA_ROWS);
  }
  const int Ms = min(M, v.M);  // never index past what the host sized the workspace for
  // edges per slot -> seg_ptr
  for (int m = t; m <= nbuf; m += T) v.seg_ptr[m] = 0;
  __syncthreads();
  for (int e = t; e < E; e += T) {
    const int64_t i = ii[e], j = jj[e];
    if (i >= 0 && i < nbuf && j >= 0 && j < nbuf) atomicAdd(&v.seg_ptr[v.slot_of[i]], 1);
  }
  __syncthreads();
  block_exscan(v.seg_ptr, nbuf + 1, lds, &tot);
  for (int m = t; m < nbuf; m += T) v.cursor[m] = 0;
  __syncthreads();
  for (int e = t; e < E; e += T) {
    const int64_t i = ii[e], j = jj[e];
    if (i >= 0 && i < nbuf && j >= 0 && j < nbuf) {
      const int m = v.slot_of[i];
      const int pos = atomicAdd(&v.cursor[m], 1);
      v.seg_edge[v.seg_ptr[m] + pos] = e;
    }
  }
  __syncthreads();
  // deterministic order inside a segment: ascending edge index (insertion sort, segments are short)
  for (int m = t; m < Ms; m += T) {
    const int a = v.seg_ptr[m], b = v.seg_ptr[m + 1];
    for (int x = a + 1; x < b; x++) {
      const int key = v.seg_edge[x];
      int y = x - 1;
      while (y >= a && v.seg_edge[y] > key) {
        v.seg_edge[y + 1] = v.seg_edge[y];
        y--;
      }
      v.seg_edge[y + 1] = key;
    }
  }
  __syncthreads();
  // Schur entries of a slot: the self row Ei (frame in the owned window) followed by the Eij
  // row of every outgoing edge whose target pose is in the window (schur_block :1244-1253).
  for (int m = t; m <= nbuf; m += T) v.ent_ptr[m] = 0;
  __syncthreads();
  for (int m = t; m < Ms; m += T) {
    const int f = v.kx[m];
    int c = (f >= w0 && f < w1) ? 1 : 0;
    for (int x = v.seg_ptr[m]; x < v.seg_ptr[m + 1]; x++) {
      const int p = (int)jj[v.seg_edge[x]] - v.t0;
      if (p >= 0 && p < v.P) c++;
    }
    v.ent_ptr[m] = c;
  }
  __syncthreads();
  block_exscan(v.ent_ptr, nbuf + 1, lds, &tot);
  if (t == 0) v.hdr[HDR_NENT] = tot;
  for (int m = t; m < Ms; m += T) {
    const int f = v.kx[m];
    int o = v.ent_ptr[m];
    if (f >= w0 && f < w1) {
      v.ent_row[o] = m;  // rows [0,M) of Erows are the self rows
      v.ent_pose[o] = f - v.t0;
      o++;
    }
    for (int x = v.seg_ptr[m]; x < v.seg_ptr[m + 1]; x++) {
      const int e = v.seg_edge[x];
      const int p = (int)jj[e] - v.t0;
      if (p >= 0 && p < v.P) {
        v.ent_row[o] = v.M + e;  // rows [M, M+E) are the Eij rows
        v.ent_pose[o] = p;
        o++;
      }
    }
  }
  __syncthreads();
  // SYRK work list: lower-triangular 16x16 tile pairs of every slot's (6 r_m)^2 block
  for (int m = t; m <= nbuf; m += T) v.wk_ptr[m] = 0;
  __syncthreads();
  for (int m = t; m < Ms; m += T) {
    const int R = 6 * (v.ent_ptr[m + 1] - v.ent_ptr[m]);
    const int RT = (R + 15) / 16;
    v.wk_ptr[m] = RT * (RT + 1) / 2;
  }
  __syncthreads();
  block_exscan(v.wk_ptr, nbuf + 1, lds, &tot);
  if (t == 0) v.hdr[HDR_NWORK] = tot;
}

// ------------------------------------------------------------------------------------------
// linearisation.  grid (slots or edges, pixel chunks), 256 threads, LIN_PPT pixels per thread.
// DEPTH = true : blockIdx.x = depth slot; walks the slot's edges; writes per-(edge,chunk) Hjj/vj
//                partial sums, Q = 1/C, w and the E rows (projective_transform_kernel :176-424
//                + accum of C, w, Ei :1397-1402 fused).
// DEPTH = false: motion-only (:1385-1392): blockIdx.x = edge, only the Hjj/vj partials.
// ------------------------------------------------------------------------------------------
template <bool DEPTH>
__global__ __launch_bounds__(LIN_THREADS) void ba_lin_kernel(
    BaView v, const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intrinsics, const float* __restrict__ disps_sens,
    const float* __restrict__ targets, const float* __restrict__ weights,
    const float* __restrict__ eta, const int64_t* __restrict__ ii, const int64_t* __restrict__ jj) {
  __shared__ float red[2][LIN_THREADS / 64][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HW = v.HW, W = v.W;
  const int chunk = blockIdx.y;
  int xb, xe, f = -1, m = -1;
  if (DEPTH) {
    m = blockIdx.x;
    if (m >= min(v.hdr[HDR_M], v.M)) return;
    f = v.kx[m];
    xb = v.seg_ptr[m];
    xe = v.seg_ptr[m + 1];
  } else {
    xb = blockIdx.x;
    xe = xb + 1;
    const int64_t i0 = ii[xb], j0 = jj[xb];
    if (i0 < 0 || i0 >= v.nbuf || j0 < 0 || j0 >= v.nbuf) return;
  }
  const Intr K = {intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};

  int pix[LIN_PPT];
  float disp[LIN_PPT], Cacc[LIN_PPT], wacc[LIN_PPT], ei[LIN_PPT][6];
#pragma unroll
  for (int p = 0; p < LIN_PPT; p++) {
    pix[p] = chunk * LIN_CP + p * LIN_THREADS + tid;
    Cacc[p] = 0.f;
    wacc[p] = 0.f;
#pragma unroll
    for (int n = 0; n < 6; n++) ei[p][n] = 0.f;
    disp[p] = 0.f;
    if (DEPTH && pix[p] < HW) disp[p] = disps[(size_t)f * HW + pix[p]];
  }

  int buf = 0;
  for (int x = xb; x < xe; x++) {
    const int e = DEPTH ? v.seg_edge[x] : x;
    const int ix = (int)ii[e], jx = (int)jj[e];
    const Rel T = rel_pose<true>(poses, ix, jx);
    const bool stereo = (ix == jx);
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; k++) acc[k] = 0.f;
    const float* tg = targets + (size_t)e * 2 * HW;
    const float* wg = weights + (size_t)e * 2 * HW;
#pragma unroll
    for (int p = 0; p < LIN_PPT; p++) {
      if (pix[p] < HW) {
        const int k = pix[p];
        const float d = DEPTH ? disp[p] : disps[(size_t)ix * HW + k];
        const float u = (float)(k % W), vv = (float)(k / W);
        const PixLin L = linearize_pixel(K, T, u, vv, d, tg[k], tg[HW + k]);
        float wu = L.valid * (0.001f * wg[k]);        // dk:305-306
        float wv = L.valid * (0.001f * wg[HW + k]);
        if (DEPTH) {
          Cacc[p] += wu * L.Jzu * L.Jzu + wv * L.Jzv * L.Jzv;         // dk:320, :353
          wacc[p] += wu * L.ru * L.Jzu + wv * L.rv * L.Jzv;           // dk:321, :354
        }
        if (stereo) {  // dk:323, :356: pose terms of a stereo pair carry no weight
          wu = 0.f;
          wv = 0.f;
        }
        // Hjj lower triangle (21) and vj (6); Ju[1] = Jv[0] = 0 structurally
        int l = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
          const float wa_u = wu * L.Ju[a], wa_v = wv * L.Jv[a];
#pragma unroll
          for (int b = 0; b <= a; b++) {
            acc[l] += wa_u * L.Ju[b] + wa_v * L.Jv[b];
            l++;
          }
          acc[21 + a] += wa_u * L.ru + wa_v * L.rv;
        }
        if (DEPTH) {
          float eij[6], eii[6];
          const float su = wu * L.Jzu, sv = wv * L.Jzv;
#pragma unroll
          for (int n = 0; n < 6; n++) eij[n] = su * L.Ju[n] + sv * L.Jv[n];  // dk:341, :374
          float* er = v.Erows + ((size_t)(v.M + e) * 6) * HW + k;
#pragma unroll
          for (int n = 0; n < 6; n++) er[(size_t)n * HW] = eij[n];
          adj_se3(T.t, T.q, eij, eii);  // Eii = -Adj^T Eij (linear in J), dk:325-326, :340
#pragma unroll
          for (int n = 0; n < 6; n++) ei[p][n] -= eii[n];
        }
      }
    }
    // workgroup sum of the 27 block entries -> Hpart[e][chunk][0..31]
    const float tot = wave_reduce32(acc);
    if ((lane & 1) == 0) red[buf][wave][reduce32_index(lane)] = tot;
    __syncthreads();
    if (tid < 32) {
      float s = 0.f;
#pragma unroll
      for (int wv_ = 0; wv_ < LIN_THREADS / 64; wv_++) s += red[buf][wv_][tid];
      v.Hpart[((size_t)e * v.nch + chunk) * 32 + tid] = s;
    }
    buf ^= 1;  // double buffer: one barrier per edge
  }

  if (DEPTH) {
    const float alpha = 0.05f;  // dk:1396
#pragma unroll
    for (int p = 0; p < LIN_PPT; p++) {
      if (pix[p] < HW) {
        const int k = pix[p];
        const size_t o = (size_t)m * HW + k;
        const float sens = disps_sens[(size_t)f * HW + k];
        const float ms = sens > 0.f ? 1.f : 0.f;
        const float C = Cacc[p] + ms * alpha + (1.f - ms) * eta[o];           // dk:1398
        const float w = wacc[p] - ms * alpha * (disp[p] - sens);               // dk:1399
        v.Q[o] = 1.0f / C;                                                     // dk:1400
        v.w[o] = w;
        float* er = v.Erows + ((size_t)m * 6) * HW + k;                        // dk:1402
#pragma unroll
        for (int n = 0; n < 6; n++) er[(size_t)n * HW] = ei[p][n];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// assemble: per edge, chunk partials -> Hjj, vj (fp64); Hii = A Hjj A^T, Hij = -A Hjj,
// vi = -A vj with A = Adj(Tij)^T; scatter-add into the lower triangle of the dense system.
// Restates Hs/vs layout :400-423 and SparseBlock::update_lhs/update_rhs :1131-1173 (blocks of
// frames before t0 are dropped; indices >= P, undefined in the reference, are dropped too).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ba_assemble_kernel(BaView v, const float* __restrict__ poses,
                                                          const int64_t* __restrict__ ii,
                                                          const int64_t* __restrict__ jj) {
  __shared__ double hj[6][6], vj[6], A[6][6], hij[6][6];
  const int e = blockIdx.x, t = threadIdx.x;
  const int64_t i64 = ii[e], j64 = jj[e];
  if (i64 < 0 || i64 >= v.nbuf || j64 < 0 || j64 >= v.nbuf) return;
  const int ix = (int)i64, jx = (int)j64;
  if (ix == jx) return;  // stereo pair: all pose blocks are zero (dk:323, :356)
  const int pi = ix - v.t0, pj = jx - v.t0;
  const bool vi_ok = pi >= 0 && pi < v.P, vj_ok = pj >= 0 && pj < v.P;
  if (!vi_ok && !vj_ok) return;
  if (t < 27) {
    double s = 0.0;
    for (int c = 0; c < v.nch; c++) s += (double)v.Hpart[((size_t)e * v.nch + c) * 32 + t];
    if (t < 21) {
      int a = 0, rem = t;
      while (rem > a) {
        rem -= a + 1;
        a++;
      }
      hj[a][rem] = s;
      hj[rem][a] = s;
    } else {
      vj[t - 21] = s;
    }
  }
  if (t >= 32 && t < 38) {  // column k of A: Adj^T applied to the k-th unit vector
    const int k = t - 32;
    const Rel T = rel_pose<true>(poses, ix, jx);
    float X[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, Y[6];
    X[k] = 1.f;
    adj_se3(T.t, T.q, X, Y);
    for (int r = 0; r < 6; r++) A[r][k] = (double)Y[r];
  }
  __syncthreads();
  const int n = v.n, ld = v.ld;
  double* S = v.sys;
  if (t < 36) {
    const int r = t / 6, c = t % 6;
    double s = 0.0;
    for (int k = 0; k < 6; k++) s -= A[r][k] * hj[k][c];  // Hij = -A Hjj
    hij[r][c] = s;
  }
  __syncthreads();
  if (t < 36) {
    const int r = t / 6, c = t % 6;
    if (vj_ok && r >= c) atomicAdd(&S[(size_t)(6 * pj + r) * ld + 6 * pj + c], hj[r][c]);
    if (vi_ok && r >= c) {
      double s = 0.0;
      for (int k = 0; k < 6; k++) s -= hij[r][k] * A[c][k];  // Hii = A Hjj A^T = -Hij A^T
      atomicAdd(&S[(size_t)(6 * pi + r) * ld + 6 * pi + c], s);
    }
    if (vi_ok && vj_ok) {
      if (pi > pj)
        atomicAdd(&S[(size_t)(6 * pi + r) * ld + 6 * pj + c], hij[r][c]);
      else  // Hji = Hij^T lands in the lower triangle
        atomicAdd(&S[(size_t)(6 * pj + c) * ld + 6 * pi + r], hij[r][c]);
    }
  } else if (t >= 40 && t < 46) {
    const int r = t - 40;
    if (vj_ok) atomicAdd(&S[(size_t)n * ld + 6 * pj + r], vj[r]);
    if (vi_ok) {
      double s = 0.0;
      for (int k = 0; k < 6; k++) s -= A[r][k] * vj[k];  // vi = -A vj
      atomicAdd(&S[(size_t)n * ld + 6 * pi + r], s);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Schur complement S = sum_slots B Q B^T on fp32 MFMA (EEt6x6_kernel :1001-1056 + the triple
// enumeration of schur_block :1257-1272, as a batched SYRK).  One wave per work item =
// (slot, lower 16x16 tile pair, K split); operands are read straight from the E rows in the
// MFMA's own layout: lane l feeds row (l & 15), pixels 4*(l >> 4) .. +3 of each 16-pixel step
// (one 16-byte load per operand per four v_mfma_f32_16x16x4_f32).
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 load4_guard(const float* p, int k, int kend) {
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
  if (p == nullptr) return r;
  if (k + 3 < kend) {
    r = *reinterpret_cast<const f32x4*>(p + k);
  } else {
    if (k < kend) r[0] = p[k];
    if (k + 1 < kend) r[1] = p[k + 1];
    if (k + 2 < kend) r[2] = p[k + 2];
  }
  return r;
}

__global__ __launch_bounds__(256) void ba_schur_kernel(BaView v) {
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int M = min(v.hdr[HDR_M], v.M);
  const int nwork = v.hdr[HDR_NWORK] * SCHUR_KSPLIT;
  const int HW = v.HW, ld = v.ld;
  const int kper = (((HW + SCHUR_KSPLIT - 1) / SCHUR_KSPLIT) + 15) & ~15;
  for (int item = wave_global; item < nwork; item += nwaves) {
    const int pair_g = item / SCHUR_KSPLIT, ks = item % SCHUR_KSPLIT;
    // slot of this work item: last m with wk_ptr[m] <= pair_g
    int lo = 0, hi = M;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (v.wk_ptr[mid] <= pair_g) lo = mid; else hi = mid;
    }
    const int m = lo;
    int pr = pair_g - v.wk_ptr[m];
    int ta = 0;
    while (pr > ta) {
      pr -= ta + 1;
      ta++;
    }
    const int tb = pr;  // ta >= tb
    const int e0 = v.ent_ptr[m];
    const int R = 6 * (v.ent_ptr[m + 1] - e0);
    const int ra = 16 * ta + (lane & 15), rb = 16 * tb + (lane & 15);
    const float* pa = nullptr;
    const float* pb = nullptr;
    if (ra < R) pa = v.Erows + ((size_t)v.ent_row[e0 + ra / 6] * 6 + ra % 6) * HW;
    if (rb < R) pb = v.Erows + ((size_t)v.ent_row[e0 + rb / 6] * 6 + rb % 6) * HW;
    const float* q = v.Q + (size_t)m * HW;
    const int kbeg = ks * kper, kend = min(HW, kbeg + kper);
    if (kbeg >= kend) continue;
    const int kg = 4 * (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = kbeg; k < kend; k += 16) {
      const f32x4 a4 = load4_guard(pa, k + kg, kend);
      const f32x4 b4 = load4_guard(pb, k + kg, kend);
      const f32x4 q4 = load4_guard(q, k + kg, kend);
#pragma unroll
      for (int s = 0; s < 4; s++)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s] * q4[s], b4[s], acc, 0, 0, 0);  // dk:1030-1038
    }
    // D[i][j]: j = lane & 15, i = 4*(lane >> 4) + reg.  (A - S): subtract, lower triangle only.
    const int lj = 16 * tb + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int li = 16 * ta + 4 * (lane >> 4) + r;
      if (li < R && lj < R) {
        const int gi = 6 * v.ent_pose[e0 + li / 6] + li % 6;
        const int gj = 6 * v.ent_pose[e0 + lj / 6] + lj % 6;
        const double val = -(double)acc[r];
        if (ta == tb) {  // both (li,lj) and (lj,li) are computed in a diagonal tile
          if (gi >= gj) atomicAdd(&v.sys[(size_t)gi * ld + gj], val);
        } else {  // the mirror element is not computed: fold it into the lower triangle
          if (gi > gj) atomicAdd(&v.sys[(size_t)gi * ld + gj], val);
          else if (gi < gj) atomicAdd(&v.sys[(size_t)gj * ld + gi], val);
          else atomicAdd(&v.sys[(size_t)gi * ld + gj], 2.0 * val);
        }
      }
    }
  }
}

// rhs of the reduced system: b -= E Q w  (Ev6x1_kernel :1059-1093, update_rhs :1308).
// One workgroup per Schur entry.
__global__ __launch_bounds__(256) void ba_ev_kernel(BaView v) {
  __shared__ float red[4][8];
  const int M = min(v.hdr[HDR_M], v.M);
  const int nent = v.hdr[HDR_NENT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int ent = blockIdx.x; ent < nent; ent += gridDim.x) {
    int lo = 0, hi = M;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (v.ent_ptr[mid] <= ent) lo = mid; else hi = mid;
    }
    const int m = lo;
    const float* er = v.Erows + (size_t)v.ent_row[ent] * 6 * v.HW;
    const float* q = v.Q + (size_t)m * v.HW;
    const float* w = v.w + (size_t)m * v.HW;
    float b[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = tid; k < v.HW; k += 256) {
      const float qw = q[k] * w[k];
#pragma unroll
      for (int n = 0; n < 6; n++) b[n] += qw * er[(size_t)n * v.HW + k];
    }
#pragma unroll
    for (int n = 0; n < 6; n++) {
      float s = b[n];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (lane == 0) red[wave][n] = s;
    }
    __syncthreads();
    if (tid < 6) {
      const double s = (double)red[0][tid] + (double)red[1][tid] + (double)red[2][tid] + (double)red[3][tid];
      atomicAdd(&v.sys[(size_t)v.n * v.ld + 6 * v.ent_pose[ent] + tid], -s);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// depth back-substitution + disparity retraction: dz = Q (w - sum_entries E^T dx), disps += dz
// (EvT6x1_kernel :1095-1115 incl. its `p <= 0` early return, accum + :1417, disp_retr :933-946).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_backsub_kernel(BaView v, float* __restrict__ disps,
                                                          const float* __restrict__ dx,
                                                          float* __restrict__ dz_out) {
  const int m = blockIdx.x;
  if (m >= min(v.hdr[HDR_M], v.M)) return;
  const int k = blockIdx.y * 256 + threadIdx.x;
  if (k >= v.HW) return;
  const int f = v.kx[m];
  float acc = 0.f;
  for (int ent = v.ent_ptr[m]; ent < v.ent_ptr[m + 1]; ent++) {
    const int p = v.ent_pose[ent];
    if (p <= 0) continue;  // dk:1105: the first window pose never feeds back into dz
    const float* er = v.Erows + (size_t)v.ent_row[ent] * 6 * v.HW + k;
    float dw = 0.f;
#pragma unroll
    for (int n = 0; n < 6; n++) dw += er[(size_t)n * v.HW] * dx[6 * p + n];
    acc += dw;
  }
  const size_t o = (size_t)m * v.HW + k;
  const float dz = v.Q[o] * (v.w[o] - acc);
  if (dz_out) dz_out[o] = dz;
  disps[(size_t)f * v.HW + k] += dz;
}

// pose retraction T <- exp(dx) T for the window (pose_retr_kernel :898-931)
__global__ void ba_pose_retr_kernel(float* __restrict__ poses, const float* __restrict__ dx, int t0,
                                    int t1) {
  const int k = t0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= t1) return;
  float xi[6], t[3], q[4], tn[3], qn[4];
  for (int n = 0; n < 6; n++) xi[n] = dx[6 * (k - t0) + n];
  for (int n = 0; n < 3; n++) t[n] = poses[7 * k + n];
  for (int n = 0; n < 4; n++) q[n] = poses[7 * k + 3 + n];
  retr_se3(xi, t, q, tn, qn);
  for (int n = 0; n < 3; n++) poses[7 * k + n] = tn[n];
  for (int n = 0; n < 4; n++) poses[7 * k + 3 + n] = qn[n];
}

// x (fp64 solution) -> dx (fp32), zeros when the factorisation failed (:1202-1210)
__global__ void ba_finish_dx_kernel(BaView v, const double* __restrict__ x, float* __restrict__ dx,
                                    float* __restrict__ dx_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && v.hdr[HDR_CHOL_FAIL]) atomicOr(&v.hdr[HDR_STATUS], STATUS_CHOL_FAIL);
  if (i >= v.n) return;
  const float val = v.hdr[HDR_CHOL_FAIL] ? 0.f : (float)x[i];
  dx[i] = val;
  if (dx_out) dx_out[i] = val;
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
void launch_prep(const BaView& v, const int64_t* ii, const int64_t* jj, hipStream_t s) {
  hipLaunchKernelGGL(ba_prep_kernel, dim3(1), dim3(1024), 0, s, v, ii, jj);
}

void launch_build_stage(const BaView& v, const float* poses, const float* disps, const float* intr,
                        const float* sens, const float* targets, const float* weights,
                        const float* eta, const int64_t* ii, const int64_t* jj, bool motion_only,
                        int stage, hipStream_t s) {
  const bool depth = !motion_only && v.M > 0;
  switch (stage) {
    case 0:
      (void)hipMemsetAsync(v.sys, 0, sizeof(double) * (size_t)(v.n + 1) * v.ld, s);
      if (depth)
        hipLaunchKernelGGL(ba_lin_kernel<true>, dim3(v.M, v.nch), dim3(LIN_THREADS), 0, s, v, poses,
                           disps, intr, sens, targets, weights, eta, ii, jj);
      else if (v.E > 0)
        hipLaunchKernelGGL(ba_lin_kernel<false>, dim3(v.E, v.nch), dim3(LIN_THREADS), 0, s, v, poses,
                           disps, intr, sens, targets, weights, eta, ii, jj);
      break;
    case 1:
      if (v.E > 0)
        hipLaunchKernelGGL(ba_assemble_kernel, dim3(v.E), dim3(64), 0, s, v, poses, ii, jj);
      break;
    case 2:
      if (depth) hipLaunchKernelGGL(ba_schur_kernel, dim3(SCHUR_GRID), dim3(256), 0, s, v);
      break;
    case 3:
      if (depth) hipLaunchKernelGGL(ba_ev_kernel, dim3(1024), dim3(256), 0, s, v);
      break;
  }
}

void launch_build(const BaView& v, const float* poses, const float* disps, const float* intr,
                  const float* sens, const float* targets, const float* weights, const float* eta,
                  const int64_t* ii, const int64_t* jj, bool motion_only, hipStream_t s) {
  for (int stage = 0; stage < 4; stage++)
    launch_build_stage(v, poses, disps, intr, sens, targets, weights, eta, ii, jj, motion_only, stage, s);
}

void launch_update(const BaView& v, float* poses, float* disps, const double* x, float* dx_out,
                   float* dz_out, bool motion_only, hipStream_t s) {
  hipLaunchKernelGGL(ba_finish_dx_kernel, dim3((v.n + 255) / 256), dim3(256), 0, s, v, x, v.dx,
                     dx_out);
  if (!motion_only && v.M > 0)
    hipLaunchKernelGGL(ba_backsub_kernel, dim3(v.M, (v.HW + 255) / 256), dim3(256), 0, s, v, disps,
                       v.dx, dz_out);
  hipLaunchKernelGGL(ba_pose_retr_kernel, dim3((v.P + 63) / 64), dim3(64), 0, s, poses, v.dx, v.t0,
                     v.t1);
}

}  // namespace droid
