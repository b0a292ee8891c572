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
#include <stdlib.h>

#include "ba_internal.hpp"
#include "se3.hpp"

namespace droid {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------

// Exclusive scan of data[0..L) in place by one workgroup; returns the total in *total (LDS).
// lds: at least blockDim.x / 64 ints.  Per-thread serial chunks, a shuffle scan inside every wave and one more over
// the wave totals: three barriers (the Hillis-Steele version it replaces took 22 of them, four times per `ba` call).
__device__ void block_exscan(int* data, int L, int* lds, int* total) {
  const int T = blockDim.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = T >> 6;
  const int per = (L + T - 1) / T;
  const int b = t * per;
  int s = 0;
  for (int k = 0; k < per; k++)
    if (b + k < L) s += data[b + k];
  int incl = s;  // inclusive scan of the per-thread sums within the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  if (wave == 0) {  // exclusive scan of the wave totals
    int w = (lane < nw) ? lds[lane] : 0;
    int wi = w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(wi, off);
      if (lane >= off) wi += v;
    }
    if (lane < nw) lds[lane] = wi - w;
    if (lane == nw - 1) *total = wi;
  }
  __syncthreads();
  int run = lds[wave] + incl - s;
  for (int k = 0; k < per; k++)
    if (b + k < L) {
      int v = data[b + k];
      data[b + k] = run;
      run += v;
    }
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

// Address of entry (gi, gj), gj <= gi, of the reduced camera system (row n = rhs) for the build kernels' atomics
__device__ __forceinline__ double* sys_at(const BaView& v, int gi, int gj) {
  if (v.packed) {   // multi-GPU build (wave-uniform kernel argument).  A real branch: evaluated speculatively next to the
                    // pitched address (select), the packed index slowed the dense-slot folds of one-GPU runs by 40 %
    unsigned idx = pk_index32(v.n, gi, gj);
    asm volatile("" : "+v"(idx));
    return v.psys + idx;
  }
  return v.sys + (size_t)gi * v.ld + gj;
}

// Per-edge metadata of one depth slot, resident in LDS for the lifetime of a workgroup, so that
// the per-pixel loops never chase seg_edge -> jj -> poses through dependent global loads.
constexpr int SLOT_MAXE = 128;  // edges per metadata chunk (slots with more are processed in chunks)
// S = double in the linearisation (fp64 reprojection / residual), float in the Schur and back-substitution
// kernels (Jacobians only).  The relative pose itself is always evaluated in fp64 (se3.hpp: rel_pose_mat).
template <typename S>
struct SlotMetaT {
  int e[SLOT_MAXE];     // edge index
  int pj[SLOT_MAXE];    // target pose index jj - t0 (may be outside [0,P))
  int ent[SLOT_MAXE];   // Schur entry index of the edge within the slot, -1 if its target is not in the window
  int flag[SLOT_MAXE];  // 1 = stereo pair (ii == jj)
  S T[SLOT_MAXE][12];   // relative pose: R (row-major 3x3), t
};
typedef SlotMetaT<float> SlotMeta;

constexpr int SF_TP = 64;             // pixels per LDS tile; 4 threads per pixel split the edges
constexpr int SF_RB = 96;             // rows per block (16 entries)
// Dense slots (more than 16 entries) are served by the SYRK-only kernel further down.
constexpr int SW_MID = 256;   // rows (incl. the w row) served by the two-workgroups-per-CU variant (42 entries; two staging rounds of 128 rows)
constexpr int SW_BIG = 512;   // rows served by the one-per-CU variant (85 entries)
constexpr int SY_MAXSPLIT = 16;  // pixel ranges per slot of the SYRK kernels (each adds one fp64 atomic per output entry)

// which kernel serves a slot: 0 = single 96-row block, 1/2 = SYRK-only kernel (E rows from v.Ebuf),
// 3 = block pairs.
// `wide` is a host-side decision (mean out-degree of the graph): sparse graphs skip the two wide
// launches altogether and leave their few dense slots to the block-pair kernel
constexpr int S2_MAXE = 16;   // edges per slot served by ba_schur2_kernel (class 0)
__device__ __forceinline__ int schur_class(int rows, int nedges, int wide) {
  if (nedges <= S2_MAXE) return 0;
  if (!wide || nedges > SLOT_MAXE || rows > SW_BIG) return 3;
  return rows <= SW_MID ? 1 : 2;
}
// v.Ebuf (dense graphs): the E rows of a slot, TILED by 32-pixel stage -- float (stage s, row, pixel o) of the slot whose
// first entry is e0 and which has `rows` = 6 nent E rows sits at 6 e0 HW + (s rows + row) 32 + o.  One stage of all rows is
// one contiguous block (rows x 128 B): the SYRK kernels stream it; row-major rows (12 KB apart at 48x64) made every
// 128-byte request open another DRAM page and held the stream at 1.6 TB/s.
__device__ __forceinline__ size_t ebuf_index(int e0, int rows, int HW, int row, int k) {
  return (size_t)6 * e0 * HW + ((size_t)(k >> 5) * rows + row) * 32 + (k & 31);
}
// 256-double units a (class-0 slot, pixel range) leaves in v.Gpart: the 16x16 Gram tiles of its 6 nedges E rows
// (lower triangle of row tiles) + one unit for the 6 nedges sums of E Q w
__device__ __forceinline__ int s2_tiles(int nedges) {
  const int ntr = (6 * nedges + 15) / 16;
  return ntr * (ntr + 1) / 2 + 1;
}

// Loads edges [x0, x0+cnt) of slot m (cnt <= SLOT_MAXE) by the first cnt threads of the workgroup.
// ent_base = entry index of the first window edge of this chunk.  Ends with a barrier.
template <typename S>
__device__ __forceinline__ void load_slot_meta(SlotMetaT<S>& sm, const BaView& v, const float* __restrict__ poses,
                                               const int64_t* __restrict__ jj, int f, int x0, int cnt,
                                               int ent_base) {
  const int t = threadIdx.x;
  if (t < cnt) {
    const int e = v.seg_edge[x0 + t];
    const int jx = (int)jj[e];
    const RelMat<S> T = rel_pose_mat<S, true>(poses, f, jx);
    sm.e[t] = e;
    sm.pj[t] = jx - v.t0;
    sm.flag[t] = (jx == f) ? 1 : 0;
#pragma unroll
    for (int n = 0; n < 9; n++) sm.T[t][n] = T.R[n];
#pragma unroll
    for (int n = 0; n < 3; n++) sm.T[t][9 + n] = T.t[n];
  }
  __syncthreads();
  if (t < cnt) {
    int a = ent_base;
    for (int u = 0; u < t; u++) a += (sm.pj[u] >= 0 && sm.pj[u] < v.P) ? 1 : 0;
    sm.ent[t] = (sm.pj[t] >= 0 && sm.pj[t] < v.P) ? a : -1;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// prep: depth slots, CSR of edges by source frame, Schur entry lists and tile work list.
// One workgroup; O(E + nbuf) work, run once per ba call (the graph is fixed across iterations).
// Restates the index bookkeeping of ba_cuda :1336-1344 and schur_block :1232-1272.
// ------------------------------------------------------------------------------------------
// INLDS: the work arrays live in (dynamic) LDS while the tables are built and are copied to the workspace at the
// end -- the kernel is a chain of ~20 barrier-separated phases of a single workgroup, each of which costs a global
// memory round trip otherwise (39 us per call at 256 frames / 2000 edges; two iterations per call in production).
__host__ __device__ inline size_t prep_lds_ints(int nbuf, int E) { return 9 * ((size_t)nbuf + 2) + 4 * ((size_t)E + 1); }

template <bool INLDS>
__global__ __launch_bounds__(1024) void ba_prep_kernel(BaView vg, const int64_t* __restrict__ ii,
                                                        const int64_t* __restrict__ jj) {
  __shared__ int lds[1024];
  __shared__ int tot;
  extern __shared__ int dyn[];
  const int t = threadIdx.x, T = blockDim.x;
  BaView v = vg;  // same header / entry tables; the work arrays may be redirected to LDS
  if (INLDS) {
    int* q = dyn;
    const int nb2 = vg.nbuf + 2, e1 = vg.E + 1;
    v.slot_of = q; q += nb2;
    v.kx = q; q += nb2;
    v.seg_ptr = q; q += nb2;
    v.cursor = q; q += nb2;
    v.ent_ptr = q; q += nb2;
    v.wk_ptr = q; q += nb2;
    v.order = q; q += nb2;
    v.gt_ptr = q; q += nb2;
    q += nb2;  // spare
    v.seg_edge = q; q += e1;
    v.xtmp = q;
  }
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
  // Every step below is edge-parallel with INDEPENDENT loads per thread (the per-slot loops they replace -- an
  // insertion sort and two passes over seg_edge -> jj -- were chains of dependent global accesses: 30 of the 47 us).
  int* utmp = v.xtmp;                // unsorted fill of the segments
  int* xflag = v.xtmp + (E + 1);     // per sorted position: target pose in the window?
  int* xslot = v.xtmp + 2 * (E + 1); // per sorted position: its slot
  for (int e = t; e < E; e += T) {
    const int64_t i = ii[e], j = jj[e];
    if (i >= 0 && i < nbuf && j >= 0 && j < nbuf) {
      const int m = v.slot_of[i];
      const int pos = atomicAdd(&v.cursor[m], 1);
      utmp[v.seg_ptr[m] + pos] = e;
    }
  }
  __syncthreads();
  // deterministic order inside a segment: ascending edge index; position = number of smaller edges of the segment
  for (int e = t; e < E; e += T) {
    const int64_t i = ii[e], j = jj[e];
    if (i >= 0 && i < nbuf && j >= 0 && j < nbuf) {
      const int m = v.slot_of[i];
      const int a = v.seg_ptr[m], b = v.seg_ptr[m + 1];
      int rank = 0;
      for (int x = a; x < b; x++) rank += (utmp[x] < e) ? 1 : 0;
      const int p = (int)j - v.t0;
      v.seg_edge[a + rank] = e;
      xflag[a + rank] = (p >= 0 && p < v.P) ? 1 : 0;
      xslot[a + rank] = m;
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
    for (int x = v.seg_ptr[m]; x < v.seg_ptr[m + 1]; x++) c += xflag[x];
    v.ent_ptr[m] = c;
  }
  __syncthreads();
  block_exscan(v.ent_ptr, nbuf + 1, lds, &tot);
  if (t == 0) v.hdr[HDR_NENT] = tot;
  for (int m = t; m < Ms; m += T) {
    const int f = v.kx[m];
    if (f >= w0 && f < w1) {
      v.ent_row[v.ent_ptr[m]] = m;  // rows [0,M) of Erows are the self rows
      v.ent_pose[v.ent_ptr[m]] = f - v.t0;
    }
  }
  for (int x = t; x < v.seg_ptr[Ms]; x += T) {
    if (!xflag[x]) continue;
    const int m = xslot[x];
    const int f = v.kx[m];
    int o = v.ent_ptr[m] + ((f >= w0 && f < w1) ? 1 : 0);
    for (int y = v.seg_ptr[m]; y < x; y++) o += xflag[y];
    const int e = v.seg_edge[x];
    v.ent_row[o] = v.M + e;  // rows [M, M+E) are the Eij rows
    v.ent_pose[o] = (int)jj[e] - v.t0;
  }
  __syncthreads();
  // partial-sum tiles of the slots ba_schur2_kernel serves (class 0: at most S2_MAXE edges)
  for (int m = t; m <= nbuf; m += T) v.gt_ptr[m] = 0;
  __syncthreads();
  for (int m = t; m < Ms; m += T) {
    const int ne = v.seg_ptr[m + 1] - v.seg_ptr[m];
    v.gt_ptr[m] = (ne > 0 && ne <= S2_MAXE) ? s2_tiles(ne) : 0;
  }
  __syncthreads();
  block_exscan(v.gt_ptr, nbuf + 1, lds, &tot);
  // dispatch order: slots by descending edge count (ties by slot index), so that the heavy
  // workgroups of the slot-parallel kernels start first (longest-processing-time-first packing)
  for (int m = t; m < Ms; m += T) v.wk_ptr[m] = v.seg_ptr[m + 1] - v.seg_ptr[m];
  __syncthreads();
  for (int m = t; m < Ms; m += T) {
    const int c = v.wk_ptr[m];
    int rank = 0;
    for (int u = 0; u < Ms; u++) {
      const int cu = v.wk_ptr[u];
      rank += (cu > c || (cu == c && u < m)) ? 1 : 0;
    }
    v.order[rank] = m;
  }
  if (t == 0) v.hdr[HDR_NWORK] = 0;
  // compact lists of the slots each SYRK variant serves (ascending slot index), so that ba_syrk3_kernel can deal
  // its workgroups over the slots that exist instead of leaving most of the grid to exit early
  if (t < 2) v.hdr[HDR_NC1 + t] = 0;
  __syncthreads();  // the rank loop above reads wk_ptr
  int my3 = 0;
  for (int m = t; m < Ms; m += T) {
    const int nent = v.ent_ptr[m + 1] - v.ent_ptr[m];
    const int c = nent > 0 ? schur_class(6 * nent + 1, v.seg_ptr[m + 1] - v.seg_ptr[m], v.wide) : 0;
    v.wk_ptr[m] = c;
    my3 += (c == 3) ? 1 : 0;
  }
  const int n3 = __syncthreads_count(my3);  // (threads, not slots: only "none" matters)
  if (t == 0 && vg.hint) {
    // launch hint for the host: with no class-3 slot the block-pair launch of every iteration of this call is empty
    // and the host may leave it out -- if this store has arrived by the time it enqueues the iteration (it never waits)
    __hip_atomic_store(vg.hint + 1, n3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(vg.hint, vg.hint_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  for (int m = t; m < Ms; m += T) {
    const int c = v.wk_ptr[m];
    if (c != 1 && c != 2) continue;
    int pos = 0, after = 0;
    for (int u = 0; u < Ms; u++) {
      const int same = v.wk_ptr[u] == c;
      pos += (same && u < m) ? 1 : 0;
      after += (same && u > m) ? 1 : 0;
    }
    vg.cls_list[(c - 1) * (nbuf + 2) + pos] = m;
    if (after == 0) v.hdr[HDR_NC1 + c - 1] = pos + 1;
  }
  if (INLDS) {  // the tables the other kernels read
    __syncthreads();
    for (int f = t; f <= nbuf; f += T) {
      vg.slot_of[f] = (f < nbuf) ? v.slot_of[f] : -1;
      vg.kx[f] = v.kx[f];
      vg.seg_ptr[f] = v.seg_ptr[f];
      vg.ent_ptr[f] = v.ent_ptr[f];
      vg.gt_ptr[f] = v.gt_ptr[f];
      vg.order[f] = v.order[f];
    }
    for (int x = t; x < E; x += T) vg.seg_edge[x] = v.seg_edge[x];
  }
}

// ------------------------------------------------------------------------------------------
// linearisation.  grid (slots or edges, pixel chunks), 256 threads, LIN_PPT pixels per thread.
// DEPTH = true : blockIdx.x = depth slot; walks the slot's edges; writes per-(edge,chunk) Hjj/vj
//                partial sums, Q = 1/C, w and the E rows (projective_transform_kernel :176-424
//                + accum of C, w, Ei :1397-1402 fused).
// DEPTH = false: motion-only (:1385-1392): blockIdx.x = edge, only the Hjj/vj partials.
// ------------------------------------------------------------------------------------------
// EROWS (dense graphs): slots that the SYRK-only Schur kernel will serve also get their UNSCALED E rows
// written to v.Ebuf (row 6*entry + n; the self row Ei = -sum_e Adj^T Eij is accumulated per pixel).
// ZSPLIT: gridDim.z workgroups share a (slot, chunk), each taking a range of the slot's edges.
template <bool DEPTH, bool EROWS, bool ZSPLIT>
__global__ __launch_bounds__(LIN_THREADS, 3) void ba_lin_kernel(
    BaView v, const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intrinsics, const float* __restrict__ disps_sens,
    const float* __restrict__ targets, const float* __restrict__ weights,
    const float* __restrict__ eta, const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, int zero_base) {
  __shared__ float red[2][LIN_THREADS / 64][32];
  __shared__ SlotMetaT<double> sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HW = v.HW, W = v.W;
  const int chunk = blockIdx.y;
  if ((int)blockIdx.x >= zero_base) {
    // spare workgroups clear the reduced camera system for this iteration (lower triangle + rhs row: what the
    // assemble / Schur atomics and the solve touch) -- the linearisation is VALU bound, the stores ride along,
    // and the fill launch in front of it is gone
    if (blockIdx.y != 0 || blockIdx.z != 0) return;
    const int nz = (int)gridDim.x - zero_base;
    if (v.packed) {
      double2* p = reinterpret_cast<double2*>(v.psys);
      const size_t n2 = pk_total(v.n) / 2;
      for (size_t c = (size_t)((int)blockIdx.x - zero_base) * LIN_THREADS + tid; c < n2; c += (size_t)nz * LIN_THREADS)
        p[c] = make_double2(0.0, 0.0);
      return;
    }
    for (int row = (int)blockIdx.x - zero_base; row <= v.n; row += nz) {
      double2* p = reinterpret_cast<double2*>(v.sys + (size_t)row * v.ld);
      const int n2 = (row == v.n) ? v.ld / 2 : (row + 2) / 2;   // 16-byte units covering columns 0..row
      for (int c = tid; c < n2; c += LIN_THREADS) p[c] = make_double2(0.0, 0.0);
    }
    return;
  }
  int xb, xe, f = -1, m = -1;
  if (DEPTH) {
    if ((int)blockIdx.x >= min(v.hdr[HDR_M], v.M)) return;
    m = v.order[blockIdx.x];
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
  // this workgroup's range of the slot's edges (gridDim.z workgroups per (slot, chunk))
  int xs = xb, xend = xe;
  if (DEPTH && ZSPLIT) {
    const int per = (xe - xb + (int)gridDim.z - 1) / (int)gridDim.z;
    xs = min(xe, xb + (int)blockIdx.z * per);
    xend = min(xe, xs + per);
  }

  // E-row emission for this slot?
  bool emit = false, has_self = false;
  int e0 = 0, erows = 0;
  if (DEPTH && EROWS) {
    e0 = v.ent_ptr[m];
    const int nent = v.ent_ptr[m + 1] - e0;
    erows = 6 * nent;
    const int cls = schur_class(6 * nent + 1, xe - xb, 1);
    emit = (cls == 1 || cls == 2);
    has_self = nent > 0 && (v.ent_row[e0] < v.M);
  }
  float selfacc[LIN_PPT][6];
#pragma unroll
  for (int p = 0; p < LIN_PPT; p++)
#pragma unroll
    for (int n = 0; n < 6; n++) selfacc[p][n] = 0.f;

  int pix[LIN_PPT];
  float disp[LIN_PPT], Cacc[LIN_PPT], wacc[LIN_PPT];
  double X0d[LIN_PPT], X1d[LIN_PPT];  // back-projected pixel, fp64 (se3.hpp: linearize_pixel_d)
#pragma unroll
  for (int p = 0; p < LIN_PPT; p++) {
    pix[p] = chunk * LIN_CP + p * LIN_THREADS + tid;
    Cacc[p] = 0.f;
    wacc[p] = 0.f;
    disp[p] = 0.f;
    if (DEPTH && pix[p] < HW) disp[p] = disps[(size_t)f * HW + pix[p]];
    const int kk = pix[p] < HW ? pix[p] : 0;
    X0d[p] = ((double)(kk % W) - (double)K.cx) / (double)K.fx;
    X1d[p] = ((double)(kk / W) - (double)K.cy) / (double)K.fy;
  }

  float in_cur[LIN_PPT][4], in_nxt[LIN_PPT][4];  // target u,v and weight u,v of this thread's pixels
  bool have_next = false;
  auto fetch_edge = [&](int e, float (&dst)[LIN_PPT][4]) {
    const float* tg = targets + (size_t)e * 2 * HW;
    const float* wg = weights + (size_t)e * 2 * HW;
#pragma unroll
    for (int p = 0; p < LIN_PPT; p++) {
      const bool ok = pix[p] < HW;
      const int k = ok ? pix[p] : 0;
      dst[p][0] = ok ? tg[k] : 0.f;
      dst[p][1] = ok ? tg[HW + k] : 0.f;
      dst[p][2] = ok ? wg[k] : 0.f;
      dst[p][3] = ok ? wg[HW + k] : 0.f;
    }
  };
  int buf = 0;
  for (int x = xs; x < xend; x++) {
    if (DEPTH && (x == xs || ((x - xb) % SLOT_MAXE) == 0)) {  // (re)load the metadata chunk holding edge x
      if (x > xs) __syncthreads();
      const int cb = xb + ((x - xb) / SLOT_MAXE) * SLOT_MAXE;
      load_slot_meta(sm, v, poses, jj, f, cb, min(SLOT_MAXE, xe - cb), (EROWS && has_self) ? 1 : 0);
    }
    const int xl = DEPTH ? (x - xb) % SLOT_MAXE : 0;
    const int e = DEPTH ? sm.e[xl] : x;
    const int ix = DEPTH ? f : (int)ii[e];
    double Rt[12];   // relative pose of the edge: R (row-major), t
    bool stereo;
    if (DEPTH) {
#pragma unroll
      for (int n = 0; n < 12; n++) Rt[n] = sm.T[xl][n];
      stereo = sm.flag[xl] != 0;
    } else {
      const int jx = (int)jj[e];
      const RelMat<double> T = rel_pose_mat<double, true>(poses, ix, jx);
#pragma unroll
      for (int n = 0; n < 9; n++) Rt[n] = T.R[n];
#pragma unroll
      for (int n = 0; n < 3; n++) Rt[9 + n] = T.t[n];
      stereo = (ix == jx);
    }
    // software pipeline: the next edge's targets / weights are in flight while this one is reduced
    if (!have_next) fetch_edge(e, in_cur);
    else {
#pragma unroll
      for (int p = 0; p < LIN_PPT; p++)
#pragma unroll
        for (int c = 0; c < 4; c++) in_cur[p][c] = in_nxt[p][c];
    }
    have_next = false;
    if (DEPTH && x + 1 < xend && xl + 1 < SLOT_MAXE) {
      fetch_edge(sm.e[xl + 1], in_nxt);
      have_next = true;
    }
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; k++) acc[k] = 0.f;
#pragma unroll
    for (int p = 0; p < LIN_PPT; p++) {
      if (pix[p] < HW) {
        const int k = pix[p];
        const float d = DEPTH ? disp[p] : disps[(size_t)ix * HW + k];
        const PixLin L = linearize_pixel_d(K, Rt, Rt + 9, X0d[p], X1d[p], d, in_cur[p][0], in_cur[p][1]);
        float wu = L.valid * (0.001f * in_cur[p][2]);        // dk:305-306
        float wv = L.valid * (0.001f * in_cur[p][3]);
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
        if (DEPTH && EROWS && emit) {  // E row of this (edge, pixel), dk:341, :374 (weights after the stereo rule)
          const float su = wu * L.Jzu, sv = wv * L.Jzv;
          float eij[6];
#pragma unroll
          for (int n = 0; n < 6; n++) eij[n] = su * L.Ju[n] + sv * L.Jv[n];
          const int a = sm.ent[xl];
          if (a >= 0) {
#pragma unroll
            for (int n = 0; n < 6; n++) v.Ebuf[ebuf_index(e0, erows, HW, 6 * a + n, k)] = eij[n];
          }
          if (has_self) {
            float eii[6];
            adjT_mat(Rt, Rt + 9, eij, eii);
#pragma unroll
            for (int n = 0; n < 6; n++) selfacc[p][n] -= eii[n];
          }
        }
      }
    }
    // workgroup sum of the 27 block entries -> Hpart[e][chunk][0..31]
    const float tot = wave_reduce32(acc);
    if ((lane & 1) == 0) red[buf][wave][reduce32_index(lane)] = tot;
    __syncthreads();  // (s_waitcnt lgkmcnt(0); s_barrier on this target: the next edge's loads and this edge's stores stay in flight)
    if (tid < 32) {
      float s = 0.f;
#pragma unroll
      for (int wv_ = 0; wv_ < LIN_THREADS / 64; wv_++) s += red[buf][wv_][tid];
      v.Hpart[((size_t)e * v.nch + chunk) * 32 + tid] = s;
    }
    buf ^= 1;  // double buffer: one barrier per edge
  }

  if (DEPTH && ZSPLIT) {
    // partial sums of this edge range; ba_lin_finish_kernel adds them up in a fixed order
#pragma unroll
    for (int p = 0; p < LIN_PPT; p++) {
      if (pix[p] < HW) {
        float* zp = v.zpart + (((size_t)blockIdx.z * v.M + m) * 8) * HW + pix[p];
        zp[0] = Cacc[p];
        zp[(size_t)HW] = wacc[p];
#pragma unroll
        for (int n = 0; n < 6; n++) zp[(size_t)(2 + n) * HW] = selfacc[p][n];
      }
    }
    return;
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
        if (EROWS && emit && has_self) {
#pragma unroll
          for (int n = 0; n < 6; n++) v.Ebuf[ebuf_index(e0, erows, HW, n, k)] = selfacc[p][n];
        }
      }
    }
  }
}

// zsplit > 1: C, w (-> Q, w) and the self rows from the partial sums of the edge ranges
__global__ __launch_bounds__(256) void ba_lin_finish_kernel(BaView v, const float* __restrict__ disps,
                                                            const float* __restrict__ disps_sens,
                                                            const float* __restrict__ eta) {
  if ((int)blockIdx.x >= min(v.hdr[HDR_M], v.M)) return;
  const int m = blockIdx.x, HW = v.HW;
  const int k = blockIdx.y * 256 + threadIdx.x;
  if (k >= HW) return;
  const int f = v.kx[m];
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < v.zsplit; z++) {
    const float* zp = v.zpart + (((size_t)z * v.M + m) * 8) * HW + k;
#pragma unroll
    for (int q = 0; q < 8; q++) s[q] += zp[(size_t)q * HW];
  }
  const float alpha = 0.05f;  // dk:1396
  const size_t o = (size_t)m * HW + k;
  const float sens = disps_sens[(size_t)f * HW + k], disp = disps[(size_t)f * HW + k];
  const float ms = sens > 0.f ? 1.f : 0.f;
  v.Q[o] = 1.0f / (s[0] + ms * alpha + (1.f - ms) * eta[o]);  // dk:1398, :1400
  v.w[o] = s[1] - ms * alpha * (disp - sens);                  // dk:1399
  if (v.wide) {
    const int e0 = v.ent_ptr[m], nent = v.ent_ptr[m + 1] - e0;
    const int cls = schur_class(6 * nent + 1, v.seg_ptr[m + 1] - v.seg_ptr[m], 1);
    if ((cls == 1 || cls == 2) && nent > 0 && v.ent_row[e0] < v.M) {
#pragma unroll
      for (int n = 0; n < 6; n++) v.Ebuf[ebuf_index(e0, 6 * nent, HW, n, k)] = s[2 + n];
    }
  }
}

// ------------------------------------------------------------------------------------------
// assemble: per edge, chunk partials -> Hjj, vj (fp64); Hii = A Hjj A^T, Hij = -A Hjj,
// vi = -A vj with A = Adj(Tij)^T; scatter-add into the lower triangle of the dense system.
// Restates Hs/vs layout :400-423 and SparseBlock::update_lhs/update_rhs :1131-1173 (blocks of
// frames before t0 are dropped; indices >= P, undefined in the reference, are dropped too).
// ------------------------------------------------------------------------------------------
struct AssembleScratch {  // per WAVE
  double hj[6][6], vj[6], A[6][6], hij[6][6];
};
// One virtual workgroup (= one wave) of the assemble step: edge e (or a solver-preset unit for e >= v.E), lane t.
// Wave-level only (no workgroup barrier), so that it can run as a stand-alone 64-thread kernel (motion-only BA) or in
// spare workgroups of ba_schur2_kernel, four virtual workgroups per 256-thread workgroup.
__device__ __forceinline__ void assemble_unit(const BaView& v, const float* __restrict__ poses, const int64_t* __restrict__ ii,
                                              const int64_t* __restrict__ jj, int e, int t, AssembleScratch& sh) {
  auto& hj = sh.hj; auto& vj = sh.vj; auto& A = sh.A; auto& hij = sh.hij;
  if (e >= v.E) {
    // spare workgroups preset the solver's scratch for this iteration (saves two fill launches in front of the
    // factorisation): solution vector and hand-off flags to 0xFF bytes, the failure flag to 0
    unsigned* w = reinterpret_cast<unsigned*>(v.xsol);
    const int words = solver_preset_words(v);
    const int nw = (words + 1023) / 1024;
    if (e - v.E < nw) {
      for (int i = (e - v.E) * 1024 + t; i < min(words, (e - v.E + 1) * 1024); i += 64) w[i] = 0xFFFFFFFFu;
      if (e == v.E && t == 0) v.hdr[HDR_CHOL_FAIL] = 0;
    } else {  // one 64x64 hand-over slot of the panel tiles per workgroup
      uint4* q = reinterpret_cast<uint4*>(v.ldiag + chol_lfin_offset(v.n) + (size_t)(e - v.E - nw) * CHOL_NB * CHOL_NB);
      for (int i = t; i < CHOL_NB * CHOL_NB / 2; i += 64) q[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    return;
  }
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
    const RelMat<double> T = rel_pose_mat<double, true>(poses, ix, jx);
    double X[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, Y[6];
    X[k] = 1.0;
    adjT_mat(T.R, T.t, X, Y);
    for (int r = 0; r < 6; r++) A[r][k] = Y[r];
  }
  __builtin_amdgcn_wave_barrier();  // one wave: its LDS operations are ordered
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int n = v.n;
  if (t < 36) {
    const int r = t / 6, c = t % 6;
    double s = 0.0;
    for (int k = 0; k < 6; k++) s -= A[r][k] * hj[k][c];  // Hij = -A Hjj
    hij[r][c] = s;
  }
  __builtin_amdgcn_wave_barrier();  // one wave: its LDS operations are ordered
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (t < 36) {
    const int r = t / 6, c = t % 6;
    if (vj_ok && r >= c) atomicAdd(sys_at(v, 6 * pj + r, 6 * pj + c), hj[r][c]);
    if (vi_ok && r >= c) {
      double s = 0.0;
      for (int k = 0; k < 6; k++) s -= hij[r][k] * A[c][k];  // Hii = A Hjj A^T = -Hij A^T
      atomicAdd(sys_at(v, 6 * pi + r, 6 * pi + c), s);
    }
    if (vi_ok && vj_ok) {
      if (pi > pj)
        atomicAdd(sys_at(v, 6 * pi + r, 6 * pj + c), hij[r][c]);
      else  // Hji = Hij^T lands in the lower triangle
        atomicAdd(sys_at(v, 6 * pj + c, 6 * pi + r), hij[r][c]);
    }
  } else if (t >= 40 && t < 46) {
    const int r = t - 40;
    if (vj_ok) atomicAdd(sys_at(v, n, 6 * pj + r), vj[r]);
    if (vi_ok) {
      double s = 0.0;
      for (int k = 0; k < 6; k++) s -= A[r][k] * vj[k];  // vi = -A vj
      atomicAdd(sys_at(v, n, 6 * pi + r), s);
    }
  }
}
__global__ __launch_bounds__(64) void ba_assemble_kernel(BaView v, const float* __restrict__ poses,
                                                          const int64_t* __restrict__ ii,
                                                          const int64_t* __restrict__ jj) {
  __shared__ AssembleScratch sh;
  assemble_unit(v, poses, ii, jj, (int)blockIdx.x, (int)threadIdx.x, sh);
}

// ------------------------------------------------------------------------------------------
// Schur complement + reduced rhs, fused with the E rows:
//   S = sum_slots B Q B^T   (EEt6x6_kernel :1001-1056 + triple enumeration :1257-1272)
//   b -= E Q w              (Ev6x1_kernel :1059-1093, update_rhs :1308)
// The reference materialises E = [Ei; Eij] (:1402-1403, 166 MB at 2000 edges) and re-reads it per
// triple.  Here one workgroup owns (depth slot, pixel range): for each 128-pixel tile it RECOMPUTES
// the slot's E rows from weights / disparities / poses (the Jacobians are cheaper than the HBM
// round trip), scales them by sqrt(Q) and parks them in LDS as B~ = B sqrt(Q) (so that
// S = B~ B~^T needs no per-operand scaling), with one extra row w sqrt(Q) whose products with the
// E rows are exactly E Q w.  The SYRK runs on v_mfma_f32_16x16x4_f32 out of LDS; accumulators
// stay in registers over all pixel tiles of the range and are folded into the lower triangle of
// the dense fp64 system with one atomic per entry.
// Slots with more than 16 entries are processed in 96-row blocks (all block pairs), recomputing
// the rows per pair, so any out-degree is handled with bounded LDS and registers.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef SCHUR_STAMPS
__device__ unsigned long long g_schur_stamps[8 * 16];
#define SSTAMP(i) do { if (stamp_on) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_prev; st_prev = now_; } } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif

constexpr int SF_PITCH = SF_TP + 4;   // 16-byte aligned rows, conflict-free 16-byte MFMA operand reads
constexpr int SF_MAXT = 11;           // max 16x16 output tiles per wave: ceil(7*6/4)

// E row (6 values) of one (edge, pixel); T = the edge's relative pose (R row-major, t)
__device__ __forceinline__ void e_row(const Intr& K, const float* T, bool stereo, float X0, float X1, float disp,
                                      float wu_raw, float wv_raw, float* eij) {
  const PixLin L = jacobians_pixel(K, T, T + 9, X0, X1, disp);
  float wu = L.valid * (0.001f * wu_raw), wv = L.valid * (0.001f * wv_raw);
  if (stereo) {
    wu = 0.f;
    wv = 0.f;
  }
  const float su = wu * L.Jzu, sv = wv * L.Jzv;
#pragma unroll
  for (int n = 0; n < 6; n++) eij[n] = su * L.Ju[n] + sv * L.Jv[n];  // dk:341, :374
}

// MULTI = false: slots of at most 16 entries (one row block; no second LDS block, so three
// workgroups fit a CU and one's VALU staging overlaps another's MFMA phase); MULTI = true: the
// rest.  Both variants are launched; a workgroup whose slot belongs to the other one exits.
template <bool MULTI>
__global__ __launch_bounds__(256, MULTI ? 2 : 3) void ba_schur_fused_kernel(
    BaView v, const float* __restrict__ poses, const float* __restrict__ disps,
    const float* __restrict__ intrinsics, const float* __restrict__ weights,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, int wide) {
  __shared__ __attribute__((aligned(16))) float EA[(SF_RB + 16) * SF_PITCH];  // row block A (+ the w row, padded to a full tile)
  __shared__ __attribute__((aligned(16))) float EB[MULTI ? SF_RB * SF_PITCH : 4];  // row block B (only for off-diagonal block pairs)
  __shared__ float SP[4 * 6 * SF_TP];            // partial self rows of the four edge subsets
  __shared__ SlotMeta sm;
  if ((int)blockIdx.x >= min(v.hdr[HDR_M], v.M)) return;
  const int m = v.order[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HW = v.HW, W = v.W;
  const int e0 = v.ent_ptr[m], nent = v.ent_ptr[m + 1] - e0;
  if (nent == 0) return;
  const int R = 6 * nent;  // E rows; global row R is the w row
  const int f = v.kx[m];
  const Intr K = {intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  // pixel range of this workgroup
  const int tiles_total = (HW + SF_TP - 1) / SF_TP;
  const int tpw = (tiles_total + gridDim.y - 1) / gridDim.y;
  const int tile_beg = blockIdx.y * tpw, tile_end = min(tiles_total, tile_beg + tpw);
  if (tile_beg >= tile_end) return;
  const bool has_self = (v.ent_row[e0] < v.M);  // the self row, when present, is entry 0
  const int nblk = (R + 1 + SF_RB - 1) / SF_RB;
  if (schur_class(R + 1, v.seg_ptr[m + 1] - v.seg_ptr[m], wide) != (MULTI ? 3 : 0)) return;  // wide kernels serve 1, 2
  const int pixl = tid & (SF_TP - 1), part = tid >> 6;  // four threads per pixel split the edges
  const int x_beg = v.seg_ptr[m], nedges = v.seg_ptr[m + 1] - x_beg;
  const bool resident = nedges <= SLOT_MAXE;  // the usual case: metadata loaded once
  // pose index of every Schur entry of the slot for the fold at the end (single-block variant: at most 16 entries):
  // two dependent global loads per folded element made the fold 15 % of the workgroup's time
  __shared__ int s_pose[MULTI ? 1 : 17];
  if (!MULTI && tid < nent) s_pose[tid] = v.ent_pose[e0 + tid];
  if (resident) load_slot_meta(sm, v, poses, jj, f, x_beg, nedges, has_self ? 1 : 0);
  else __syncthreads();

  // software prefetch of the next tile's per-pixel inputs (Q, disparity, the weights of this
  // thread's first four edges): issued before the MFMA phase of the current tile
  constexpr int SF_PF = 4;
  float pf_q = 0.f, pf_d = 0.f, pf_wr = 0.f, pf_w[2 * SF_PF];
  auto prefetch = [&](int tile) {
    const int k = tile * SF_TP + pixl;
    const bool ok = (tile < tile_end) && (k < HW);
    pf_q = ok ? v.Q[(size_t)m * HW + k] : 0.f;
    pf_d = ok ? disps[(size_t)f * HW + k] : 0.f;
    pf_wr = (ok && part == 0) ? v.w[(size_t)m * HW + k] : 0.f;
#pragma unroll
    for (int u = 0; u < SF_PF; u++) {
      const int x = part + 4 * u;
      const bool okx = ok && resident && x < nedges;
      const float* wg = weights + (size_t)(okx ? sm.e[x] : 0) * 2 * HW;
      pf_w[2 * u] = okx ? wg[k] : 0.f;
      pf_w[2 * u + 1] = okx ? wg[HW + k] : 0.f;
    }
  };

#ifdef SCHUR_STAMPS
  const bool stamp_on = (lane == 0) && (blockIdx.x == 100) && (blockIdx.y == 1);
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#endif
  for (int ba = 0; ba < nblk; ba++) {
    for (int bb = 0; bb <= ba; bb++) {
      const int ra0 = ba * SF_RB, rb0 = bb * SF_RB;
      const int na = min(SF_RB + ((ba == nblk - 1) ? 1 : 0), R + 1 - ra0);  // rows of block A (may include w)
      const int nb = min(SF_RB, R - rb0);                                  // rows of block B (E rows only)
      const int ta_n = (na + 15) / 16, tb_n = (nb + 15) / 16;
      const int ntiles = (ba == bb) ? ta_n * (ta_n + 1) / 2 : ta_n * tb_n;
      constexpr int MAXT = MULTI ? SF_MAXT : 7;  // one block: at most 7*8/2 = 28 tiles over 4 waves
      // fp32 MFMA chains are kept to ONE 64-pixel tile (16 k-steps): the tile result is added to an fp64 total.
      // A chain over the whole pixel range (512 products) leaves 3e-7 of the block in every partial sum, which
      // the ill-conditioned reduced system (cond 1e6 on the 256-keyframe graph) turns into 3e-5 of pose error.
      double tot[MAXT][4];
#pragma unroll
      for (int t = 0; t < MAXT; t++)
#pragma unroll
        for (int x = 0; x < 4; x++) tot[t][x] = 0.0;

      prefetch(tile_beg);
      for (int tile = tile_beg; tile < tile_end; tile++) {
        const int k = tile * SF_TP + pixl;
        const bool pok = k < HW;
        const float cur_q = pf_q, cur_d = pf_d, cur_wr = pf_wr;
        float cur_w[2 * SF_PF];
#pragma unroll
        for (int u = 0; u < 2 * SF_PF; u++) cur_w[u] = pf_w[u];
        SSTAMP(0);
        __syncthreads();  // previous tile consumed
        SSTAMP(1);
        // ---- stage: recompute the E rows of both blocks for 128 pixels
        {
          const float sq = sqrtf(cur_q);
          const float disp = cur_d;
          for (int side = 0; side < ((ba == bb) ? 1 : 2); side++) {
            float* buf = side ? EB : EA;
            const int r0 = side ? rb0 : ra0;
            const int nrow = side ? nb : min(na, R - ra0);  // E rows in this block
            const int a_beg = r0 / 6, a_end = (r0 + nrow + 5) / 6;  // entries touching the block
            float selfacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const bool self_here = has_self && r0 == 0;
            // edge entries of the block; when the self row lives here, every edge of the slot
            // contributes to it (Ei = -sum_e Adj^T Eij, dk:325-326, :1402)
            int ent_base = has_self ? 1 : 0;
            for (int c0 = 0; c0 < nedges; c0 += SLOT_MAXE) {
              const int cnt = min(SLOT_MAXE, nedges - c0);
              if (!resident) {  // very high out-degree: stream the metadata in chunks
                __syncthreads();
                load_slot_meta(sm, v, poses, jj, f, x_beg + c0, cnt, ent_base);
                for (int u = 0; u < cnt; u++) ent_base += (sm.ent[u] >= 0) ? 1 : 0;
              }
              for (int x = part; x < cnt; x += 4) {  // the four waves take the edges round-robin
                const int a = sm.ent[x];
                const bool in_blk = a >= a_beg && a < a_end;
                if (!(in_blk || self_here)) continue;
                float T[12];
#pragma unroll
                for (int n = 0; n < 12; n++) T[n] = sm.T[x][n];
                float eij[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (pok) {
                  float wu_raw, wv_raw;
                  const int u = x >> 2;  // this thread's u-th edge
                  if (resident && u < SF_PF) {
                    wu_raw = cur_w[0];
                    wv_raw = cur_w[1];
#pragma unroll
                    for (int uu = 1; uu < SF_PF; uu++)
                      if (u == uu) {
                        wu_raw = cur_w[2 * uu];
                        wv_raw = cur_w[2 * uu + 1];
                      }
                  } else {
                    const float* wg = weights + (size_t)sm.e[x] * 2 * HW;
                    wu_raw = wg[k];
                    wv_raw = wg[HW + k];
                  }
                  e_row(K, T, sm.flag[x] != 0, ((float)(k % W) - K.cx) / K.fx, ((float)(k / W) - K.cy) / K.fy, disp, wu_raw, wv_raw, eij);
                }
                if (self_here) {
                  float eii[6];
                  adjT_mat(T, T + 9, eij, eii);
#pragma unroll
                  for (int n = 0; n < 6; n++) selfacc[n] -= eii[n];
                }
                if (in_blk) {
#pragma unroll
                  for (int n = 0; n < 6; n++) {
                    const int row = 6 * a + n - r0;
                    if (row >= 0 && row < nrow) buf[row * SF_PITCH + pixl] = eij[n] * sq;
                  }
                }
              }
            }
            SSTAMP(2);
            if (self_here) {  // four partial sums per pixel, combined in a fixed order
#pragma unroll
              for (int n = 0; n < 6; n++) SP[(part * 6 + n) * SF_TP + pixl] = selfacc[n];
              __syncthreads();
              for (int n = part; n < 6; n += 4) {
                const float sum = (SP[(0 * 6 + n) * SF_TP + pixl] + SP[(1 * 6 + n) * SF_TP + pixl]) +
                                  (SP[(2 * 6 + n) * SF_TP + pixl] + SP[(3 * 6 + n) * SF_TP + pixl]);
                buf[n * SF_PITCH + pixl] = sum * sq;
              }
              if (ba != bb) __syncthreads();  // SP is reused by the second side
            }
            // zero the padding rows up to the next multiple of 16 (and the w row slot)
            const int rows_pad = ((side ? nb : na) + 15) & ~15;
            for (int row = nrow + part; row < rows_pad; row += 4) buf[row * SF_PITCH + pixl] = 0.f;
          }
          if (ba == nblk - 1 && part == 0) {  // the w row: w sqrt(Q), global row R
            EA[(R - ra0) * SF_PITCH + pixl] = cur_wr * sq;
          }
        }
        SSTAMP(3);
        __syncthreads();
        SSTAMP(4);
        prefetch(tile + 1);
        // ---- SYRK of the tile: wave w owns output tiles w, w+4, ...
        const int r = lane & 15, g = lane >> 4;
        const float* Bs = (ba == bb) ? EA : EB;
#pragma unroll
        for (int t = 0; t < MAXT; t++) {
          const int ti = wave + 4 * t;
          if (ti < ntiles) {
            int ta, tb;
            if (ba == bb) {
              ta = 0;
              int rem = ti;
              while (rem > ta) {
                rem -= ta + 1;
                ta++;
              }
              tb = rem;
            } else {
              ta = ti / tb_n;
              tb = ti % tb_n;
            }
            // 16-byte operand reads; K runs in a permuted order (k-step (s,e) of lane group g = pixel
            // 16s+4g+e), identical for both operands
            const float* pa = &EA[(16 * ta + r) * SF_PITCH + 4 * g];
            const float* pb = &Bs[(16 * tb + r) * SF_PITCH + 4 * g];
            f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < SF_TP; s4 += 16) {
              const f32x4 av = *reinterpret_cast<const f32x4*>(pa + s4);
              const f32x4 bv = *reinterpret_cast<const f32x4*>(pb + s4);
#pragma unroll
              for (int e = 0; e < 4; e++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], c, 0, 0, 0);
            }
#pragma unroll
            for (int x = 0; x < 4; x++) tot[t][x] += (double)c[x];
          }
        }
      }
      SSTAMP(5);
      // ---- fold the accumulators into the dense system (A - S): lower triangle, fp64 atomics
      {
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int t = 0; t < MAXT; t++) {
          const int ti = wave + 4 * t;
          if (ti >= ntiles) continue;
          int ta, tb;
          if (ba == bb) {
            ta = 0;
            int rem = ti;
            while (rem > ta) {
              rem -= ta + 1;
              ta++;
            }
            tb = rem;
          } else {
            ta = ti / tb_n;
            tb = ti % tb_n;
          }
#pragma unroll
          for (int x = 0; x < 4; x++) {
            const int li = ra0 + 16 * ta + 4 * g + x;  // global row index within the slot (A side)
            const int lj = rb0 + 16 * tb + r;          // B side, always an E row
            if (lj >= R || li > R) continue;
            const double val = -tot[t][x];
            const int gj = 6 * (MULTI ? v.ent_pose[e0 + lj / 6] : s_pose[lj / 6]) + lj % 6;
            if (li == R) {  // w row: reduced rhs
              atomicAdd(sys_at(v, v.n, gj), val);
              continue;
            }
            const int gi = 6 * (MULTI ? v.ent_pose[e0 + li / 6] : s_pose[li / 6]) + li % 6;
            const bool diag_tile = (ba == bb) && (ta == tb);
            if (diag_tile) {  // both (li,lj) and (lj,li) are computed
              if (gi >= gj) atomicAdd(sys_at(v, gi, gj), val);
            } else {  // the mirror element is not computed: fold it into the lower triangle
              if (gi > gj) atomicAdd(sys_at(v, gi, gj), val);
              else if (gi < gj) atomicAdd(sys_at(v, gj, gi), val);
              else atomicAdd(sys_at(v, gi, gj), 2.0 * val);
            }
          }
        }
      }
      SSTAMP(6);
    }
  }
#ifdef SCHUR_STAMPS
  if (stamp_on)
    for (int i = 0; i < 8; i++) g_schur_stamps[wave * 8 + i] = st_acc[i];
#endif
}


// ------------------------------------------------------------------------------------------
// Schur complement of SPARSE slots (at most S2_MAXE edges leave the frame; the sparse global / local BA
// graphs consist of nothing else).  Same contraction as above, restructured:
//   * only the EDGE rows E_e and the w row enter the SYRK.  The self row of the slot is a fixed linear
//     combination of them, E_i = -sum_e Adj(T_e)^T E_e (dk:325-326, :1402), so its blocks follow from the edge
//     blocks afterwards, per slot and in fp64:  S_ib = -sum_e A_e S_eb,  S_ii = sum_ee' A_e S_ee' A_e'^T,
//     (E Q w)_i = -sum_e A_e (E Q w)_e  (ba_schur_fold_kernel).  The per-pixel adjoint action, the four-way
//     combine of its partial sums through LDS and one barrier per tile disappear from the staging;
//   * the relative pose is a rotation matrix + translation (12 FMAs per pixel instead of the quaternion
//     sandwich), Jacobians only (jacobians_pixel);
//   * fp32 MFMA chains are one 64-pixel tile long and are summed in fp64 registers (tile totals);
//   * no atomics here: each (slot, pixel range) workgroup writes its fp64 Gram tiles to v.Gpart, the fold kernel
//     sums the ranges, forms the self blocks and adds ONE atomic per system entry and slot (6x fewer than one per
//     entry and pixel range).
// Rows of a slot: 6x + n for its x-th edge (every edge of the slot, window target or not: all of them feed the
// self row), row R = 6 nedges = w sqrt(Q); padded with zero rows to a multiple of 16.
// ------------------------------------------------------------------------------------------
constexpr int S2_MAXT = 6;                 // 6 row tiles of 16 at 16 edges: 21 Gram tiles over 4 waves
struct Schur2Meta {
  int e[S2_MAXE];
  int flag[S2_MAXE];
  float T[S2_MAXE][12];
};

// One workgroup = 4 waves that alternate between staging (VALU) and the SYRK (matrix pipe); three workgroups per
// CU.  A producer / consumer split of the waves (tried: 8 waves, double-buffered LDS) gains nothing on gfx950:
// fp32 and fp64 MFMAs do not overlap with VALU work of other waves of the same SIMD (tools/micro/
// mfma_valu_overlap.hip: together = sum of the two alone), so the kernel's time is staging + SYRK whatever the
// arrangement, and both are minimised instead: the w row is NOT part of the SYRK (one more row tile for a single
// useful row: 10 instead of 6 Gram tiles at 8 edges): E Q w is accumulated by the staging threads (6 FMAs per
// edge and pixel) and reduced over the pixels once per workgroup.
// ds_read_b128 serves a wave in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): a group
// mixes operand rows 0-3,12-15 at k-chunk g with rows 4-11 at chunk g+1 (or g-1), so with a linear row pitch of 68 floats rows 11
// and 12 of every tile meet on one bank quad (PMC: 29 % of the LDS cycles of this kernel were conflicts).  Rows 4..11 (mod 16)
// therefore keep their 4-float chunks pairwise swapped (column ^ 4); the staging stores apply the same swizzle.
__device__ __forceinline__ int s2_swz(int row) { return (((row + 4) >> 3) & 1) << 2; }
__global__ __launch_bounds__(256, 3) void ba_schur2_kernel(BaView v, const float* __restrict__ poses,
                                                           const float* __restrict__ disps,
                                                           const float* __restrict__ intrinsics,
                                                           const float* __restrict__ weights,
                                                           const int64_t* __restrict__ ii,
                                                           const int64_t* __restrict__ jj, int wide, int asm_units) {
  __shared__ __attribute__((aligned(16))) float EB[6 * S2_MAXE * SF_PITCH];
  __shared__ Schur2Meta sm;
  if ((int)blockIdx.x >= v.M) {
    // spare workgroups (x >= M): the assemble step of this iteration (per-edge pose blocks -> reduced system, solver scratch
    // presets), four wave-sized units per workgroup.  Independent of the Schur products (both only add into the system), so it
    // hides inside this launch instead of holding a 9 us launch of its own in front of it.
    AssembleScratch* sh = reinterpret_cast<AssembleScratch*>(EB);
    const int unit = (((int)blockIdx.x - v.M) * (int)gridDim.y + (int)blockIdx.y) * 4 + (int)(threadIdx.x >> 6);
    if (unit < asm_units) assemble_unit(v, poses, ii, jj, unit, (int)(threadIdx.x & 63), sh[threadIdx.x >> 6]);
    return;
  }
  if ((int)blockIdx.x >= min(v.hdr[HDR_M], v.M)) return;
  const int m = v.order[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HW = v.HW, W = v.W;
  const int x_beg = v.seg_ptr[m], nedges = v.seg_ptr[m + 1] - x_beg;
  if (nedges == 0 || nedges > S2_MAXE) return;
  const int R = 6 * nedges;
  const int ntr = (R + 15) / 16, ntiles = ntr * (ntr + 1) / 2;
  const int f = v.kx[m];
  const Intr K = {intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  const int tiles_total = (HW + SF_TP - 1) / SF_TP;
  const int tpw = (tiles_total + gridDim.y - 1) / gridDim.y;
  const int tile_beg = blockIdx.y * tpw, tile_end = min(tiles_total, tile_beg + tpw);
  if (tile_beg >= tile_end) return;
  const int pixl = tid & (SF_TP - 1), part = tid >> 6;  // four threads per pixel split the edges
  if (tid < nedges) {
    const int e = v.seg_edge[x_beg + tid];
    const int jx = (int)jj[e];
    const RelMat<float> T = rel_pose_mat<float, true>(poses, f, jx);
    sm.e[tid] = e;
    sm.flag[tid] = (jx == f) ? 1 : 0;
#pragma unroll
    for (int n = 0; n < 9; n++) sm.T[tid][n] = T.R[n];
#pragma unroll
    for (int n = 0; n < 3; n++) sm.T[tid][9 + n] = T.t[n];
  }
  // the zero rows behind the last E row never change
  for (int row = R + part; row < 16 * ntr; row += 4) EB[row * SF_PITCH + pixl] = 0.f;   // (zeros: no swizzle needed)
  __syncthreads();

  constexpr int NU = S2_MAXE / 4;  // edges per thread
  float pf_q = 0.f, pf_d = 0.f, pf_wr = 0.f, pf_w[2 * NU];
  auto prefetch = [&](int tile) {
    const int k = tile * SF_TP + pixl;
    const bool ok = (tile < tile_end) && (k < HW);
    pf_q = ok ? v.Q[(size_t)m * HW + k] : 0.f;
    pf_d = ok ? disps[(size_t)f * HW + k] : 0.f;
    pf_wr = ok ? v.w[(size_t)m * HW + k] : 0.f;
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int x = part + 4 * u;
      const bool okx = ok && x < nedges;
      const float* wg = weights + (size_t)(okx ? sm.e[x] : 0) * 2 * HW;
      pf_w[2 * u] = okx ? wg[k] : 0.f;
      pf_w[2 * u + 1] = okx ? wg[HW + k] : 0.f;
    }
  };

  double tot[S2_MAXT][4];
#pragma unroll
  for (int t = 0; t < S2_MAXT; t++)
#pragma unroll
    for (int x = 0; x < 4; x++) tot[t][x] = 0.0;
  float uacc[NU][6];  // (E Q w) of this thread's edges over its pixels
#pragma unroll
  for (int u = 0; u < NU; u++)
#pragma unroll
    for (int n = 0; n < 6; n++) uacc[u][n] = 0.f;
  // LDS offsets of this lane's operand rows: tile wave + 4 t = (ta, tb) of the row-major lower triangle
  int toff_a[S2_MAXT], toff_b[S2_MAXT];
#pragma unroll
  for (int t = 0; t < S2_MAXT; t++) {
    int ta = 0, rem = min(wave + 4 * t, ntiles - 1);
    while (rem > ta) {
      rem -= ta + 1;
      ta++;
    }
    toff_a[t] = (16 * ta + (lane & 15)) * SF_PITCH + ((4 * (lane >> 4)) ^ s2_swz(lane & 15));
    toff_b[t] = (16 * rem + (lane & 15)) * SF_PITCH + ((4 * (lane >> 4)) ^ s2_swz(lane & 15));
  }

  prefetch(tile_beg);
  for (int tile = tile_beg; tile < tile_end; tile++) {
    const int k = tile * SF_TP + pixl;
    const bool pok = k < HW;
    const float sq = sqrtf(pf_q), disp = pf_d, wq = pf_wr * sq;
    float cur_w[2 * NU];
#pragma unroll
    for (int u = 0; u < 2 * NU; u++) cur_w[u] = pf_w[u];
    if (tile > tile_beg) __syncthreads();  // the previous tile has been multiplied
    // ---- stage B~ = E sqrt(Q) of 64 pixels
    const float X0 = ((float)(k % W) - K.cx) / K.fx, X1 = ((float)(k / W) - K.cy) / K.fy;
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int x = part + 4 * u;
      if (x < nedges) {
        float eij[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (pok && sm.flag[x] == 0) {  // stereo pairs carry no weight in the pose terms (dk:323, :356): zero rows
          const float* T = sm.T[x];
          const float px = T[0] * X0 + T[1] * X1 + T[2] + disp * T[9];
          const float py = T[3] * X0 + T[4] * X1 + T[5] + disp * T[10];
          const float pz = T[6] * X0 + T[7] * X1 + T[8] + disp * T[11];
          const bool bad = pz < DROID_MIN_DEPTH;
          const float d = bad ? 0.f : 1.0f / pz;
          PixLin L;
          pix_jacobians(K, px, py, d, disp, T[9], T[10], T[11], L);
          const float val = bad ? 0.f : 0.001f * sq;   // weights are scaled by 0.001 (dk:305-306), rows by sqrt(Q)
          const float su = (val * cur_w[2 * u]) * L.Jzu, sv = (val * cur_w[2 * u + 1]) * L.Jzv;
#pragma unroll
          for (int n = 0; n < 6; n++) eij[n] = su * L.Ju[n] + sv * L.Jv[n];  // dk:341, :374
        }
#pragma unroll
        for (int n = 0; n < 6; n++) {
          EB[(6 * x + n) * SF_PITCH + (pixl ^ s2_swz(6 * x + n))] = eij[n];
          uacc[u][n] = fmaf(eij[n], wq, uacc[u][n]);   // (E sqrt Q)(w sqrt Q): Ev6x1_kernel dk:1059-1093
        }
      }
    }
    __syncthreads();
    prefetch(tile + 1);
    // ---- SYRK of the tile: wave w owns Gram tiles w, w+4, ...; one fp32 chain per 64-pixel tile, fp64 totals
    // Two Gram tiles of a wave at a time: each keeps its ONE fp32 chain (same sums as before, bit for bit), but the two chains
    // alternate on the matrix pipe and share the wait for their operand reads -- one tile after the other left the pipe
    // idle for the LDS latency twice per tile and for the latency of every dependent MFMA.
#pragma unroll
    for (int t = 0; t < S2_MAXT; t += 2) {
      const bool two = (t + 1 < S2_MAXT) && (wave + 4 * (t + 1) < ntiles);  // wave-uniform
      if (two) {
        const float* pa0 = &EB[toff_a[t]];
        const float* pb0 = &EB[toff_b[t]];
        const float* pa1 = &EB[toff_a[t + 1 < S2_MAXT ? t + 1 : t]];
        const float* pb1 = &EB[toff_b[t + 1 < S2_MAXT ? t + 1 : t]];
        f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < SF_TP; s4 += 16) {
          const f32x4 av0 = *reinterpret_cast<const f32x4*>(pa0 + s4), bv0 = *reinterpret_cast<const f32x4*>(pb0 + s4);
          const f32x4 av1 = *reinterpret_cast<const f32x4*>(pa1 + s4), bv1 = *reinterpret_cast<const f32x4*>(pb1 + s4);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[e], bv0[e], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[e], bv1[e], c1, 0, 0, 0);
          }
        }
#pragma unroll
        for (int x = 0; x < 4; x++) {
          tot[t][x] += (double)c0[x];
          tot[t + 1 < S2_MAXT ? t + 1 : t][x] += (double)c1[x];
        }
      } else if (wave + 4 * t < ntiles) {
        // 16-byte operand reads; K runs in a permuted order (k-step (s,e) of lane group g = pixel 16s+4g+e),
        // identical for both operands
        const float* pa = &EB[toff_a[t]];
        const float* pb = &EB[toff_b[t]];
        f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < SF_TP; s4 += 16) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(pa + s4);
          const f32x4 bv = *reinterpret_cast<const f32x4*>(pb + s4);
#pragma unroll
          for (int e = 0; e < 4; e++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], c, 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < 4; x++) tot[t][x] += (double)c[x];
      }
    }
  }
  // ---- this pixel range's Gram tiles: [tile][lane][4], 32 bytes per lane; behind them the 6 nedges sums of E Q w
  double* gp = v.Gpart + ((size_t)v.gt_ptr[m] * gridDim.y + (size_t)blockIdx.y * (ntiles + 1)) * 256;
#pragma unroll
  for (int t = 0; t < S2_MAXT; t++) {
    const int ti = wave + 4 * t;
    if (ti < ntiles) {
      double* q = gp + (size_t)ti * 256 + 4 * lane;
      *reinterpret_cast<double2*>(q) = make_double2(tot[t][0], tot[t][1]);
      *reinterpret_cast<double2*>(q + 2) = make_double2(tot[t][2], tot[t][3]);
    }
  }
  // E Q w: the per-thread sums go through LDS (the operand tile is free now), row 6x+n, column rotated by the row
  // so that the 6 nedges summing threads hit different banks; fp64 from here on
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NU; u++) {
    const int x = part + 4 * u;
    if (x < nedges) {
#pragma unroll
      for (int n = 0; n < 6; n++) EB[(6 * x + n) * SF_TP + ((pixl + 6 * x + n) & (SF_TP - 1))] = uacc[u][n];
    }
  }
  __syncthreads();
  if (tid < R) {
    double sacc = 0.0;
#pragma unroll 8
    for (int c = 0; c < SF_TP; c++) sacc += (double)EB[tid * SF_TP + ((c + tid) & (SF_TP - 1))];
    gp[(size_t)ntiles * 256 + tid] = sacc;
  }
}

// Per slot: sum the pixel ranges' Gram tiles, derive the self-row blocks (see above) and subtract everything
// from the dense system: S -= E Q E^T on the lower triangle, rhs -= E Q w.  fp64 throughout.  16 waves: every Gram
// tile has its own wave (all partial-sum loads of the slot are in flight together), the last wave builds the
// adjoints meanwhile.
constexpr int S2_GP = 6 * S2_MAXE + 1;  // pitch of the dense Gram matrix in LDS (97: odd, conflict-free columns)
constexpr int S2_MAXC = 8;              // pixel ranges summed with all loads in flight (more: a second round)
__global__ __launch_bounds__(1024) void ba_schur_fold_kernel(BaView v, const float* __restrict__ poses,
                                                             const int64_t* __restrict__ jj, int nsplit) {
  __shared__ double G[(6 * S2_MAXE + 1) * S2_GP];  // rows 0..R-1: E Q E^T (both triangles), row R: E Q w
  __shared__ double As[S2_MAXE][6][6];             // A_e = Adj(T_e)^T
  __shared__ double Vs[S2_MAXE][6][6];             // V_b = -sum_e A_e S_eb  (= S_ib, the self row against edge b)
  __shared__ double Sss[6][6], us[6];
  __shared__ int s_pose[S2_MAXE];                  // window pose of an edge's target, or -1
  if ((int)blockIdx.x >= min(v.hdr[HDR_M], v.M)) return;
  const int m = v.order[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x_beg = v.seg_ptr[m], nedges = v.seg_ptr[m + 1] - x_beg;
  if (nedges == 0 || nedges > S2_MAXE) return;
  const int R = 6 * nedges;
  const int ntr = (R + 15) / 16, ntiles = ntr * (ntr + 1) / 2;
  const int f = v.kx[m];
  const int tiles_total = (v.HW + SF_TP - 1) / SF_TP;
  const int tpw = (tiles_total + nsplit - 1) / nsplit;
  const int nchunk = (tiles_total + tpw - 1) / tpw;  // pixel ranges that hold pixels
  const int e0 = v.ent_ptr[m];
  const bool has_self = (v.ent_ptr[m + 1] > e0) && (v.ent_row[e0] < v.M);
  const int pf = f - v.t0;
  const double* gp = v.Gpart + (size_t)v.gt_ptr[m] * nsplit * 256;
  const size_t cstride = (size_t)(ntiles + 1) * 256;
  if (wave == 15) {
    if (lane < nedges) {
      const int e = v.seg_edge[x_beg + lane];
      const int jx = (int)jj[e];
      const int pj = jx - v.t0;
      s_pose[lane] = (pj >= 0 && pj < v.P) ? pj : -1;
      const RelMat<double> T = rel_pose_mat<double, true>(poses, f, jx);
#pragma unroll
      for (int kk = 0; kk < 6; kk++) {
        double X[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, Y[6];
        X[kk] = 1.0;
        adjT_mat(T.R, T.t, X, Y);
#pragma unroll
        for (int r = 0; r < 6; r++) As[lane][r][kk] = Y[r];
      }
    }
  } else if (wave == 14) {  // E Q w: 6 nedges sums per pixel range
    for (int o = lane; o < R; o += 64) {
      double sacc = 0.0;
      for (int c = 0; c < nchunk; c++) sacc += gp[(size_t)c * cstride + (size_t)ntiles * 256 + o];
      G[R * S2_GP + o] = sacc;
    }
  } else {
    // 1. dense Gram matrix: tile ti = (ta, tb), element (lane, x) = row 16 ta + 4 (lane >> 4) + x, column 16 tb + (lane & 15)
    for (int ti = wave; ti < ntiles; ti += 14) {
      int ta = 0, tb = ti;
      while (tb > ta) {
        tb -= ta + 1;
        ta++;
      }
      double a[4] = {0.0, 0.0, 0.0, 0.0};
      for (int c0 = 0; c0 < nchunk; c0 += S2_MAXC) {
        double2 lo[S2_MAXC], hi[S2_MAXC];
#pragma unroll
        for (int c = 0; c < S2_MAXC; c++) {
          const bool ok = c0 + c < nchunk;
          const double* q = gp + (size_t)(ok ? c0 + c : 0) * cstride + (size_t)ti * 256 + 4 * lane;
          lo[c] = *reinterpret_cast<const double2*>(q);
          hi[c] = *reinterpret_cast<const double2*>(q + 2);
          if (!ok) lo[c] = hi[c] = make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int c = 0; c < S2_MAXC; c++) {
          a[0] += lo[c].x; a[1] += lo[c].y; a[2] += hi[c].x; a[3] += hi[c].y;
        }
      }
      const int r = lane & 15, g = lane >> 4;
#pragma unroll
      for (int x = 0; x < 4; x++) {
        const int li = 16 * ta + 4 * g + x, lj = 16 * tb + r;
        if (li >= R || lj >= R) continue;
        G[li * S2_GP + lj] = a[x];
        if (ta != tb) G[lj * S2_GP + li] = a[x];
      }
    }
  }
  __syncthreads();
  // 2. self-row blocks
  if (has_self) {
    for (int o = tid; o < nedges * 36; o += 1024) {
      const int b = o / 36, r = (o / 6) % 6, c = o % 6;
      double sacc = 0.0;
      for (int e = 0; e < nedges; e++)
#pragma unroll
        for (int kk = 0; kk < 6; kk++) sacc -= As[e][r][kk] * G[(6 * e + kk) * S2_GP + 6 * b + c];
      Vs[b][r][c] = sacc;
    }
    if (tid >= 960 && tid < 966) {
      const int r = tid - 960;
      double sacc = 0.0;
      for (int e = 0; e < nedges; e++)
#pragma unroll
        for (int kk = 0; kk < 6; kk++) sacc -= As[e][r][kk] * G[R * S2_GP + 6 * e + kk];
      us[r] = sacc;
    }
    __syncthreads();
    if (tid < 36) {
      const int r = tid / 6, c = tid % 6;
      double sacc = 0.0;
      for (int b = 0; b < nedges; b++)
#pragma unroll
        for (int kk = 0; kk < 6; kk++) sacc -= Vs[b][r][kk] * As[b][c][kk];
      Sss[r][c] = sacc;
    }
    __syncthreads();
  }
  // 3. scatter: entries = window edges (+ the self entry, index nedges)
  const int nrow = R + (has_self ? 6 : 0);
  const int n = v.n;
  for (int o = tid; o < nrow * nrow; o += 1024) {
    const int ia = o / nrow, ib = o % nrow;
    const int a = ia / 6, r = ia % 6, b = ib / 6, c = ib % 6;
    const int pa = (a == nedges) ? pf : s_pose[a], pb = (b == nedges) ? pf : s_pose[b];
    if (pa < 0 || pb < 0) continue;
    const int gi = 6 * pa + r, gj = 6 * pb + c;
    if (gi < gj) continue;
    double val;
    if (a < nedges && b < nedges) val = G[ia * S2_GP + ib];
    else if (a == nedges && b == nedges) val = Sss[r][c];
    else if (a == nedges) val = Vs[b][r][c];
    else val = Vs[a][c][r];
    atomicAdd(sys_at(v, gi, gj), -val);
  }
  for (int o = tid; o < nrow; o += 1024) {
    const int a = o / 6, r = o % 6;
    const int pa = (a == nedges) ? pf : s_pose[a];
    if (pa < 0) continue;
    const double val = (a == nedges) ? us[r] : G[R * S2_GP + o];
    atomicAdd(sys_at(v, n, 6 * pa + r), -val);
  }
}

// (ta, tb), tb <= ta, of index ti in the row-major enumeration of a lower triangle
__device__ __forceinline__ void tri_coords(int ti, int& ta, int& tb) {
  int a = (int)((sqrtf(8.0f * (float)ti + 1.0f) - 1.0f) * 0.5f);
  if (a * (a + 1) / 2 > ti) a--;
  if ((a + 1) * (a + 2) / 2 <= ti) a++;
  ta = a;
  tb = ti - a * (a + 1) / 2;
}

// ------------------------------------------------------------------------------------------
// Dense slots, SYRK only.  With 17..85 entries per slot the accumulators of S = B Q B^T no longer
// fit one wave's share of a small workgroup, and every workgroup that shares the output tiles of a (slot, pixel range)
// needs ALL rows of the slot: recomputing the E rows per share repeats the whole staging arithmetic and alternates VALU and
// MFMA phases (an earlier kernel of this file did exactly that: 0.50 ms for the Schur complement of an 8-way edge shard).
// For such graphs the linearisation writes the unscaled E rows once (v.Ebuf, tiled by 32-pixel stage: ebuf_index) and the
// kernel below only streams them.  Output in 32x32 super-tiles (2x2 MFMA tiles): one A fragment serves two B fragments.
// Rounds 1-2 ran this on v_mfma_f32_16x16x4_f32 with LDS-DMA staging and folded with fp64 atomics (528 us for the 256
// slots of BASELINE configs[3], of which the 12 M atomics were ~135 us: profiles/r03_dense_syrk_ab.txt).
// ------------------------------------------------------------------------------------------
constexpr int SY_TPX = 32;  // pixels per stage
constexpr int SY_T1 = 16 * 17 / 2, SY_T2 = 32 * 33 / 2;  // 16x16 tiles of a class-1 / class-2 slot's lower triangle (ba_internal.hpp sizes v.sy_part with the same numbers)

// ------------------------------------------------------------------------------------------
// S = B~ B~^T on the bf16 matrix pipe at fp32 accuracy (round 3).  Every fp32 operand B~ = E sqrt(Q) is split into three
// bf16 terms, x = b0 + b1 + b2 (round to nearest: exact to 2^-27 |x|), and a 32-pixel k step becomes the six products of
// order <= 2 on v_mfma_f32_16x16x32_bf16 -- b0b0, b0b1, b1b0, b0b2, b1b1, b2b0; a bf16 x bf16 product is exact in fp32 and
// the 32 products of a step are summed inside the instruction, so the chain has FEWER fp32 roundings than the
// v_mfma_f32_16x16x4_f32 chain it replaces (tools/micro/mfma_acc_bf16.hip: unbiased, 0.7x its rms error); dropped terms
// 2^-26.  6 x 16 cycles instead of 8 x 32 per tile and step.
// Staging: no fp32 image in LDS.  A thread owns 16-byte k groups (row, g): it loads its 8 pixels of the E row straight into
// registers (the loads run two steps ahead), scales them by sqrt(Q), splits, and writes three 16-byte chunks; chunk (plane p,
// k group g, row) sits at ((4p + g) NR + row) x 16 B.  Two task maps (template LANEMAP), same-box A/B in
// profiles/r03_dense_syrk_pmc.txt:
//   * MFMA lane map (row 16 w + (l & 15), g = l >> 4; NR a multiple of 16): operand reads AND staging writes are conflict-free
//     -- a 16-byte LDS access runs in 16-lane groups that mix rows 0-3, 12-15 of k group g with rows 4-11 of g +- 1
//     (MI355X_MICROARCH.md), and with the k groups a multiple of 256 B apart those rows cover all 64 banks once.  But as a
//     GLOBAL access pattern the lane map puts a row's four 32-byte pieces 16 lanes apart: four times the line requests.
//     Used by the class-1 launch (every row staged once per step: 251-254 us against 255).
//   * row-major tasks (row t >> 2, g = t & 3; NR = 4 mod 16): four adjacent lanes per 128-byte row segment, conflict-free
//     writes, but every operand read is a 2-way conflict (rows 8-11 of one k group against rows 12-15 of its neighbour:
//     SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE).  Used by the class-2 launch, whose four workgroups per
//     (slot, pixel range) each stage every row: 38.7 us against 42.6-43.1.
// Row R = w sqrt(Q) gives E Q w.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  const bf16x2 h = __builtin_convertvector((f32x2){a, b}, bf16x2);
  return __builtin_bit_cast(unsigned, h);
}

// x[0..7] -> three planes of 8 bf16 (k order = element order)
__device__ __forceinline__ void split3_bf16(const float (&x)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const float xa = x[2 * e], xb = x[2 * e + 1];
    const unsigned h0 = cvt_pk_bf16(xa, xb);
    const float ra = xa - __uint_as_float(h0 << 16), rb = xb - __uint_as_float(h0 & 0xffff0000u);
    const unsigned h1 = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(h1 << 16), sb = rb - __uint_as_float(h1 & 0xffff0000u);
    p0[e] = h0;
    p1[e] = h1;
    p2[e] = cvt_pk_bf16(sa, sb);
  }
}

// NWV waves per workgroup; NSHARE workgroups share the super-tiles of a (slot, pixel range) -- each of them stages ALL rows,
// so the dense class runs ONE 16-wave workgroup per CU (every row converted once) with two plane buffers (one barrier per
// step, the conversion of step s+1 beside the products of step s) and the loads two steps ahead.
template <int ROWS, int NWV, int CLS, int NSHARE, bool DBUF, bool LANEMAP>
__global__ __launch_bounds__(64 * NWV) void ba_syrk3_kernel(BaView v) {
  constexpr int NW = NWV, NTH = 64 * NWV, NST = (ROWS / 16 + 1) / 2;
  constexpr int MAXS = ((NST * (NST + 1) / 2 + NSHARE - 1) / NSHARE + NW - 1) / NW;  // super-tiles per wave
  // rows of a (plane, k group) panel (+ one tile row of slack): a multiple of 16 with the MFMA lane map, = 4 mod 16 without
  constexpr int NR = ((ROWS + 15) & ~15) + 16 + (LANEMAP ? 0 : 4);
  constexpr int MAXT = (4 * ROWS + NTH - 1) / NTH;  // k groups a thread stages per step
  constexpr int PD = DBUF ? 2 : 1;                  // steps the global loads run ahead
  __shared__ u32x4 PL[(DBUF ? 2 : 1) * 12 * NR];
  const int count = min(v.hdr[HDR_NC1 + CLS - 1], v.M);
  if (count <= 0) return;
  const int HW = v.HW;
  const int stages_total = HW / SY_TPX;  // v.wide guarantees HW % 32 == 0
  const int wgs = (int)(gridDim.x * gridDim.y), wlin = (int)(blockIdx.x + gridDim.x * blockIdx.y);
  const int nrange = max(1, min(min(wgs / count, SY_MAXSPLIT), stages_total));
  if (wlin >= count * nrange) return;
  const int m = v.cls_list[(CLS - 1) * (v.nbuf + 2) + wlin % count];
  const int range = wlin / count;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int e0 = v.ent_ptr[m], nent = v.ent_ptr[m + 1] - e0;
  const int R = 6 * nent;  // E rows; row R is the w row
  const int spw = (stages_total + nrange - 1) / nrange;
  const int st_beg = range * spw, st_end = min(stages_total, st_beg + spw);
  if (st_beg >= st_end) return;
  const int ntr = (R + 1 + 15) / 16;              // 16-row tiles
  const int nst = (ntr + 1) / 2;                  // 32-row super-tile rows
  const int nsup_all = nst * (nst + 1) / 2;       // lower triangle of super-tiles
  // Super-tiles differ in cost: a diagonal one has no use for its strict upper tile (3 of 4 tiles), the last block row of
  // an odd tile count has one tile row only (2 of 4; 1 in the corner).  They are sorted by cost and dealt out in that
  // order -- position p to share p % NSHARE, there to the waves back and forth (0..NW-1, NW-1..0, ..) -- so that shares, the
  // four SIMDs and the waves all carry nearly the same number of MFMAs (13 row tiles: 91 useful tiles of 112).
  __shared__ unsigned char s_cost[NST * (NST + 1) / 2];
  __shared__ short s_sorted[NST * (NST + 1) / 2];
  for (int i = tid; i < nsup_all; i += (int)blockDim.x) {
    int sa, sb;
    tri_coords(i, sa, sb);
    const int rows2 = (2 * sa + 1 < ntr) ? 2 : 1;
    s_cost[i] = (unsigned char)(sa == sb ? (rows2 == 2 ? 3 : 1) : 2 * rows2);
  }
  __syncthreads();
  for (int i = tid; i < nsup_all; i += (int)blockDim.x) {
    const int c = s_cost[i];
    int pos = 0;
    for (int o = 0; o < nsup_all; o++) {
      const int co = s_cost[o];
      pos += (co > c || (co == c && o < i)) ? 1 : 0;
    }
    if (pos % NSHARE == (int)blockIdx.z) s_sorted[pos / NSHARE] = (short)i;
  }
  __syncthreads();
  const int nsup = (nsup_all - (int)blockIdx.z + NSHARE - 1) / NSHARE;  // super-tiles of this workgroup
  if (nsup <= 0) return;
  int mysup[MAXS];
#pragma unroll
  for (int u = 0; u < MAXS; u++) {
    const int kpos = NW * u + ((u & 1) ? NW - 1 - wave : wave);
    mysup[u] = __builtin_amdgcn_readfirstlane(kpos < nsup ? (int)s_sorted[kpos] : -1);
  }

  // staging plan.  No branch around a global load anywhere below (the waitcnt pass counts loads only through straight
  // code; a predicated load made it drain everything at the loop head): every lane loads in every round, from a
  // clamped row; a round whose rows lie past the slot is skipped per WAVE around the conversion only, lanes past the
  // slot inside a needed round write their chunks to the slack row NR - 1.
  const int nrows = R + 1;
  const int sg = LANEMAP ? (lane >> 4) : (tid & 3);  // this thread's k group
  const float* src[MAXT];
  int sstep[MAXT];
  int dst[MAXT];      // chunk index of plane 0
  bool needed[MAXT];  // wave-uniform
#pragma unroll
  for (int it = 0; it < MAXT; it++) {
    const int row = (LANEMAP ? 16 * wave + (lane & 15) : (tid >> 2)) + (NTH / 4) * it;  // (either way wave w: rows 16 w .. + 15)
    const int rr = min(row, nrows - 1);
    src[it] = (rr < R ? v.Ebuf + ebuf_index(e0, R, HW, rr, 0) : v.w + (size_t)m * HW) + 8 * sg;
    sstep[it] = rr < R ? 32 * R : 32;  // the E rows are tiled by stage: one stage of all rows is one contiguous block
    dst[it] = sg * NR + (row < nrows ? row : NR - 1);
    needed[it] = (NTH / 4) * it + 16 * wave < nrows;
  }
  const float* qsrc = v.Q + (size_t)m * HW + 8 * sg;
  f32x4 pf[PD][MAXT][2], pq[2];  // (the Q row is hot in the caches: its loads run one step ahead only)
  auto load_q = [&](int st) {
    const size_t o = (size_t)min(st, st_end - 1) * SY_TPX;
    pq[0] = *reinterpret_cast<const f32x4*>(qsrc + o);
    pq[1] = *reinterpret_cast<const f32x4*>(qsrc + o + 4);
  };
  auto load_stage = [&](int st, int k) {  // past the end: the last stage again (never consumed)
    const size_t sc = (size_t)min(st, st_end - 1);
#pragma unroll
    for (int it = 0; it < MAXT; it++) {
      const float* g = src[it] + sc * sstep[it];
      pf[k][it][0] = *reinterpret_cast<const f32x4*>(g);
      pf[k][it][1] = *reinterpret_cast<const f32x4*>(g + 4);
    }
  };
  auto store_stage = [&](int k, int buf) {
    float sq[8];
#pragma unroll
    for (int e = 0; e < 8; e++) sq[e] = __builtin_amdgcn_sqrtf(pq[e >> 2][e & 3]);
#pragma unroll
    for (int it = 0; it < MAXT; it++) {
      if (needed[it]) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = pf[k][it][e >> 2][e & 3] * sq[e];
        u32x4 p0, p1, p2;
#ifdef SY3_NO_CONV
        p0 = __builtin_bit_cast(u32x4, pf[k][it][0]); p1 = __builtin_bit_cast(u32x4, pf[k][it][1]); p2 = p0;
#else
        split3_bf16(x, p0, p1, p2);
#endif
        u32x4* P = PL + buf * 12 * NR + dst[it];
        P[0] = p0;
        P[4 * NR] = p1;
        P[8 * NR] = p2;
      }
    }
  };

  const int r = lane & 15, g = lane >> 4;
  f32x4 acc[MAXS][4];  // [super-tile][2*ia + ib]: tile (2*sa + ia, 2*sb + ib)
#pragma unroll
  for (int u = 0; u < MAXS; u++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[u][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // plane-0 chunk of this lane's operand row of the first tile row / column of a super-tile: lane part + a scalar
  // (kept apart: six lane registers less across the stage loop)
  const int offl = g * NR + r;
  int offa[MAXS], offb[MAXS];
  bool upper[MAXS], second[MAXS];
#pragma unroll
  for (int u = 0; u < MAXS; u++) {
    int sa, sb;
    tri_coords(max(mysup[u], 0), sa, sb);
    offa[u] = __builtin_amdgcn_readfirstlane(32 * sa);
    offb[u] = __builtin_amdgcn_readfirstlane(32 * sb);
    upper[u] = (sa != sb);
    second[u] = (2 * sa + 1 < ntr);
  }
  // D = sum over the six products of order <= 2 of two split operands, two tiles interleaved.  The products of a step
  // start from zero and join the running total on the VALU: the instruction adds its 32 products and C with
  // truncation-like alignment when C is large (tools/micro/mfma_acc_bf16.hip, "wide range": bias -5e-8 .. -2e-7 with a
  // running accumulator, none with a fresh one).
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto mm2 = [&](const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4& t, const bf16x8 (&a2)[3], const bf16x8 (&b2)[3], f32x4& t2) {
    f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], zero4, 0, 0, 0);
    f32x4 c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[0], b2[2], zero4, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[2], b2[0], c2, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[1], b2[1], c2, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[0], b2[1], c2, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[1], b2[0], c2, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[0], b2[0], c2, 0, 0, 0);
    t += c;
    t2 += c2;
  };
  auto mm1 = [&](const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4& t) {
    f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], zero4, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
    t += c;
  };
  auto multiply = [&](int buf) {
#ifdef SY3_NO_MM
    return;
#endif
    const u32x4* P = PL + buf * 12 * NR;
    auto frag = [&](int chunk) { return __builtin_bit_cast(bf16x8, P[chunk]); };
#pragma unroll
    for (int u = 0; u < MAXS; u++) {
      if (mysup[u] >= 0) {  // wave-uniform
        bf16x8 a0[3], a1[3], b0[3], b1[3];
#pragma unroll
        for (int p = 0; p < 3; p++) {
          a0[p] = frag(offl + offa[u] + 4 * p * NR);
          b0[p] = frag(offl + offb[u] + 4 * p * NR);
        }
        if (second[u]) {
#pragma unroll
          for (int p = 0; p < 3; p++) a1[p] = frag(offl + offa[u] + 16 + 4 * p * NR);
          mm2(a0, b0, acc[u][0], a1, b0, acc[u][2]);
#pragma unroll
          for (int p = 0; p < 3; p++) b1[p] = frag(offl + offb[u] + 16 + 4 * p * NR);
          if (upper[u]) mm2(a0, b1, acc[u][1], a1, b1, acc[u][3]);
          else mm1(a1, b1, acc[u][3]);
        } else if (upper[u]) {  // last block row of an odd tile count: one tile row against two tile columns
#pragma unroll
          for (int p = 0; p < 3; p++) b1[p] = frag(offl + offb[u] + 16 + 4 * p * NR);
          mm2(a0, b0, acc[u][0], a0, b1, acc[u][1]);
        } else {  // its corner
          mm1(a0, b0, acc[u][0]);
        }
      }
    }
  };

  // (__syncthreads() is s_waitcnt lgkmcnt(0); s_barrier on gfx950 -- it orders the LDS traffic and leaves the prefetched global
  // loads in flight; they are waited for, with counted vmcnt, where their registers are used)
  auto lds_barrier = [&]() { __syncthreads(); };
  if constexpr (!DBUF) {
    load_q(st_beg);
    load_stage(st_beg, 0);
    for (int st = st_beg; st < st_end; st++) {
      lds_barrier();  // the planes of step st-1 are consumed
      store_stage(0, 0);
      lds_barrier();  // the planes of step st are complete
      load_q(st + 1);
      load_stage(st + 1, 0);
      multiply(0);
    }
  } else {
    // steps in pairs so that the register sets and the plane buffers have compile-time indices; in step s the planes of
    // s+1 are written (their readers finished before the last barrier) and the loads of s+3 issued
    load_q(st_beg);
    load_stage(st_beg, 0);
    load_stage(st_beg + 1, 1);
    store_stage(0, 0);
    load_q(st_beg + 1);
    load_stage(st_beg + 2, 0);
    lds_barrier();
    // Within a step the conversion (VALU, LDS writes into the other buffer) and the products (LDS reads, matrix pipe) are
    // independent: half of the waves run them in the opposite order, so that behind every barrier one group reads LDS and
    // multiplies while the other converts, instead of all twelve waves queueing for the LDS and then for the matrix pipe
    const bool mul_first = (wave & 1) != 0;  // (two of a SIMD's three waves in one group, one in the other)
    for (int st = st_beg; st < st_end; st += 2) {
      if (mul_first) multiply(0);
      store_stage(1, 1);  // step st+1
      load_q(st + 2);
      load_stage(st + 3, 1);
      if (!mul_first) multiply(0);
      lds_barrier();
      if (mul_first && st + 1 < st_end) multiply(1);
      store_stage(0, 0);  // step st+2
      load_q(st + 3);
      load_stage(st + 4, 0);
      if (!mul_first && st + 1 < st_end) multiply(1);
      lds_barrier();
    }
  }
  // ---- leave the tiles to the fold kernel: no atomics here.  Pair (slot position, pixel range) owns SY_T tiles of 256
  // floats; a lane stores its four values of tile ti at 4 * lane (row 16 ta + 4 g + x, column 16 tb + r).
  float* part = v.sy_part[CLS - 1] + ((size_t)(wlin % count) * nrange + range) * (size_t)((CLS == 1 ? SY_T1 : SY_T2) * 256);
#pragma unroll
  for (int u = 0; u < MAXS; u++) {
    if (mysup[u] >= 0) {
      int sa, sb;
      tri_coords(mysup[u], sa, sb);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int ta = 2 * sa + (q >> 1), tb = 2 * sb + (q & 1);
        if (ta >= ntr || tb > ta) continue;  // outside the slot / strict upper tile of a diagonal super-tile
        *reinterpret_cast<f32x4*>(part + (size_t)(ta * (ta + 1) / 2 + tb) * 256 + 4 * lane) = acc[u][q];
      }
    }
  }
}

// Fold of the dense classes: sums a slot's tiles over its pixel ranges in fp64 and adds ONE fp64 atomic per system entry
// and slot.  The kernels of rounds 1-2 added one per entry and pixel range from inside the SYRK: 12 M atomics at one rank,
// and every extra pixel range -- what a shard with 32 slots on 256 CUs needs -- cost another 0.05 ms.  (A device-scope
// fp64 atomic is a read-modify-write of its line at the memory side, the XCDs' L2s not being coherent: ~128 B of HBM
// traffic each.  An owner-computes fold -- workgroup p gathers block row p of the system from the slots that hold pose p,
// LDS accumulation, plain stores -- was built and measured: 135 us against 136 at one rank, 64 against 29 on an
// 8-way shard, a chain of dependent look-ups per workgroup; profiles/r03_dense_syrk_ab.txt.)
// grid (M, tile groups, 2 classes); wgs1 / wgs2 = workgroups (x * y) of the two SYRK launches.
__global__ __launch_bounds__(256) void ba_syrk_fold_kernel(BaView v, int wgs1, int wgs2) {
  const int cls = (int)blockIdx.z + 1;
  const int count = min(v.hdr[HDR_NC1 + cls - 1], v.M);
  const int pos = (int)blockIdx.x;
  if (pos >= count) return;
  const int tmax = cls == 1 ? SY_T1 : SY_T2;
  const int stages_total = v.HW / SY_TPX;
  const int nrange = max(1, min(min((cls == 1 ? wgs1 : wgs2) / count, SY_MAXSPLIT), stages_total));
  const int spw = (stages_total + nrange - 1) / nrange;
  const int nused = (stages_total + spw - 1) / spw;  // ranges that have stages (the others wrote nothing)
  const int m = v.cls_list[(cls - 1) * (v.nbuf + 2) + pos];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e0 = v.ent_ptr[m], nent = v.ent_ptr[m + 1] - e0;
  const int R = 6 * nent;
  const int ntr = (R + 1 + 15) / 16, ntiles = ntr * (ntr + 1) / 2;
  __shared__ int s_pose[SLOT_MAXE + 2];
  for (int i = tid; i < nent; i += (int)blockDim.x) s_pose[i] = v.ent_pose[e0 + i];
  __syncthreads();
  const float* part = v.sy_part[cls - 1] + (size_t)pos * nrange * tmax * 256;
  const int r = lane & 15, g = lane >> 4;
  for (int ti = (int)blockIdx.y * 4 + wave; ti < ntiles; ti += 4 * (int)gridDim.y) {
    int ta, tb;
    tri_coords(ti, ta, tb);
    double sum[4] = {0.0, 0.0, 0.0, 0.0};
    for (int rg = 0; rg < nused; rg++) {
      const f32x4 pv = *reinterpret_cast<const f32x4*>(part + ((size_t)rg * tmax + ti) * 256 + 4 * lane);
#pragma unroll
      for (int x = 0; x < 4; x++) sum[x] += (double)pv[x];
    }
    const int lj = 16 * tb + r;  // B side, always an E row
    if (lj >= R) continue;
    const int gj = 6 * s_pose[lj / 6] + lj % 6;
#pragma unroll
    for (int x = 0; x < 4; x++) {
      const int li = 16 * ta + 4 * g + x;  // row of the slot (A side)
      if (li > R) continue;
      const double val = -sum[x];
      if (li == R) {  // w row: reduced rhs
        atomicAdd(sys_at(v, v.n, gj), val);
        continue;
      }
      const int gi = 6 * s_pose[li / 6] + li % 6;
      if (ta == tb) {  // diagonal tile: both (li,lj) and (lj,li) are computed
        if (gi >= gj) atomicAdd(sys_at(v, gi, gj), val);
      } else {  // the mirror element is not computed: fold it into the lower triangle
        if (gi > gj) atomicAdd(sys_at(v, gi, gj), val);
        else if (gi < gj) atomicAdd(sys_at(v, gj, gi), val);
        else atomicAdd(sys_at(v, gi, gj), 2.0 * val);
      }
    }
  }
}

#ifdef SCHUR_STAMPS
extern "C" int droid_debug_schur_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_schur_stamps), sizeof(unsigned long long) * 8 * 16);
}
#endif

// ------------------------------------------------------------------------------------------
// depth back-substitution + disparity retraction: dz = Q (w - sum_entries E^T dx), disps += dz
// (EvT6x1_kernel :1095-1115 incl. its `p <= 0` early return, accum + :1417, disp_retr :933-946).
// ------------------------------------------------------------------------------------------
constexpr int BSUB_PPT = 1;  // pixels per thread: amortises the per-workgroup edge metadata

__global__ __launch_bounds__(256) void ba_backsub_kernel(
    BaView v, const float* __restrict__ poses, float* __restrict__ disps,
    const float* __restrict__ intrinsics, const float* __restrict__ weights,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, const double* __restrict__ xsol,
    float* __restrict__ dz_out) {
  __shared__ SlotMeta sm;
  __shared__ float dxs[SLOT_MAXE][6];  // dx of each edge's target pose (0 when it does not feed back)
  if ((int)blockIdx.x >= min(v.hdr[HDR_M], v.M)) return;
  const int m = v.order[blockIdx.x];
  const int HW = v.HW;
  const int f = v.kx[m];
  const Intr K = {intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  int kpix[BSUB_PPT];
  float disp[BSUB_PPT], acc[BSUB_PPT];
  float bx0[BSUB_PPT], bx1[BSUB_PPT];  // back-projected pixel, once per pixel
#pragma unroll
  for (int p = 0; p < BSUB_PPT; p++) {
    kpix[p] = (blockIdx.y * BSUB_PPT + p) * 256 + threadIdx.x;
    disp[p] = kpix[p] < HW ? disps[(size_t)f * HW + kpix[p]] : 0.f;
    acc[p] = 0.f;
    const int kk = kpix[p] < HW ? kpix[p] : 0;
    bx0[p] = ((float)(kk % v.W) - K.cx) / K.fx;
    bx1[p] = ((float)(kk / v.W) - K.cy) / K.fy;
  }
  const int pf = f - v.t0;
  // the self row exists for frames of the owned window (entry 0 of the slot) and feeds back
  // only when its pose index is > 0 (dk:1105)
  const int e0 = v.ent_ptr[m];
  const bool has_self = (v.ent_ptr[m + 1] > e0) && (v.ent_row[e0] < v.M);
  const bool self_on = has_self && pf > 0;
  // dx = fp32(solution), or 0 when the factorisation failed (dk:1202-1210)
  const bool failed = v.hdr[HDR_CHOL_FAIL] != 0;
  float dxi[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (self_on && !failed)
    for (int n = 0; n < 6; n++) dxi[n] = (float)xsol[6 * pf + n];
  const int x_beg = v.seg_ptr[m], nedges = v.seg_ptr[m + 1] - x_beg;
  for (int c0 = 0; c0 < nedges; c0 += SLOT_MAXE) {
    const int cnt = min(SLOT_MAXE, nedges - c0);
    if (c0 > 0) __syncthreads();
    load_slot_meta(sm, v, poses, jj, f, x_beg + c0, cnt, 0);
    if (threadIdx.x < cnt) {
      const int pj = sm.pj[threadIdx.x];
      const bool on = pj > 0 && pj < v.P;  // entries with p <= 0 or p >= P do not feed back
      // per edge: c = dx_j - Adj(T_ij) dx_i, so that E_ij . dx_j + E_i . dx_i (E_i = -Adj^T E_ij per edge,
      // dk:325-326) becomes ONE dot product per (edge, pixel): (Adj^T e) . d = e . (Adj d)
      float c[6];
      for (int n = 0; n < 6; n++) c[n] = (on && !failed) ? (float)xsol[6 * pj + n] : 0.f;
      if (self_on) {
        const float* T = sm.T[threadIdx.x];
        for (int k = 0; k < 6; k++) {  // (Adj d)_k = (Adj^T e_k) . d
          float ek[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, col[6];
          ek[k] = 1.f;
          adjT_mat(T, T + 9, ek, col);
          float z = 0.f;
          for (int n = 0; n < 6; n++) z += col[n] * dxi[n];
          c[k] -= z;
        }
      }
      for (int n = 0; n < 6; n++) dxs[threadIdx.x][n] = c[n];
      sm.ent[threadIdx.x] = on ? 1 : 0;
    }
    __syncthreads();
    // software pipeline: the weights of edge x+1 are in flight while edge x is evaluated (clamped index: no branch
    // around the loads; an edge that does not feed back costs two loads it does not use)
    float wnext[BSUB_PPT][2];
    auto fetch_w = [&](int x) {
      const float* wg = weights + (size_t)sm.e[min(x, cnt - 1)] * 2 * HW;
#pragma unroll
      for (int p = 0; p < BSUB_PPT; p++) {
        const int kk = kpix[p] < HW ? kpix[p] : 0;
        wnext[p][0] = wg[kk];
        wnext[p][1] = wg[HW + kk];
      }
    };
    fetch_w(0);
    for (int x = 0; x < cnt; x++) {
      float wraw[BSUB_PPT][2];
#pragma unroll
      for (int p = 0; p < BSUB_PPT; p++) {
        const bool ok = kpix[p] < HW;
        wraw[p][0] = ok ? wnext[p][0] : 0.f;
        wraw[p][1] = ok ? wnext[p][1] : 0.f;
      }
      fetch_w(x + 1);
      const bool edge_on = sm.ent[x] != 0;
      if (!(edge_on || self_on)) continue;
      float T[12];
#pragma unroll
      for (int n = 0; n < 12; n++) T[n] = sm.T[x][n];
#pragma unroll
      for (int p = 0; p < BSUB_PPT; p++) {
        float eij[6];
        e_row(K, T, sm.flag[x] != 0, bx0[p], bx1[p], disp[p], wraw[p][0], wraw[p][1], eij);
        float dw = 0.f;
#pragma unroll
        for (int n = 0; n < 6; n++) dw += eij[n] * dxs[x][n];
        acc[p] += dw;
      }
    }
  }
#pragma unroll
  for (int p = 0; p < BSUB_PPT; p++) {
    if (kpix[p] >= HW) continue;
    const size_t o = (size_t)m * HW + kpix[p];
    const float dz = v.Q[o] * (v.w[o] - acc[p]);
    if (dz_out) dz_out[o] = dz;
    disps[(size_t)f * HW + kpix[p]] = disp[p] + dz;
  }
}

// pose retraction T <- exp(dx) T for the window (pose_retr_kernel :898-931); dx = fp32 of the fp64
// solution, zeros when the factorisation failed (:1202-1210)
__global__ void ba_pose_retr_kernel(BaView v, float* __restrict__ poses, const double* __restrict__ xsol,
                                    float* __restrict__ dx_out, int* __restrict__ status_mirror) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int fl = v.hdr[HDR_CHOL_FAIL];  // 1 = not positive definite (dx = 0 like dk:1202-1210), 2 = stalled grid
  const bool failed = fl != 0;
  if (p == 0) {
    int st = v.hdr[HDR_STATUS];
    if (failed) {
      st |= (fl >= 2) ? STATUS_CHOL_STALL : STATUS_CHOL_FAIL;
      atomicOr(&v.hdr[HDR_STATUS], (fl >= 2) ? STATUS_CHOL_STALL : STATUS_CHOL_FAIL);
    }
    if (status_mirror) {  // host-visible copy: the wrapper reads it at its next call without synchronising
      __hip_atomic_store(status_mirror + 1, v.hdr[HDR_M], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(status_mirror, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // Errors are STICKY: words 2 / 3 count the iterations that ended with a contract violation or a stalled solve
      // and OR their bits.  Only the device writes them (one kernel at a time per workspace), the host only reads
      // and remembers the count it has reported, so a violation of call k survives call k+1 being enqueued (and
      // resetting the workspace status) before the host looks.
      if (st & (STATUS_BAD_INDEX | STATUS_ETA_ROWS | STATUS_CHOL_STALL)) {
        const int cnt = __hip_atomic_load(status_mirror + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const int bits = __hip_atomic_load(status_mirror + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(status_mirror + 3, bits | st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(status_mirror + 2, cnt + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if (p >= v.P) return;
  const int k = v.t0 + p;
  float xi[6], t[3], q[4], tn[3], qn[4];
  for (int n = 0; n < 6; n++) {
    xi[n] = failed ? 0.f : (float)xsol[6 * p + n];
    v.dx[6 * p + n] = xi[n];
    if (dx_out) dx_out[6 * p + n] = xi[n];
  }
  for (int n = 0; n < 3; n++) t[n] = poses[7 * k + n];
  for (int n = 0; n < 4; n++) q[n] = poses[7 * k + 3 + n];
  retr_se3(xi, t, q, tn, qn);
  for (int n = 0; n < 3; n++) poses[7 * k + n] = tn[n];
  for (int n = 0; n < 4; n++) poses[7 * k + 3 + n] = qn[n];
}

// multi-GPU: psys (packed block-column major, all-reduced over the ranks) -> sys (pitched, what the solver factors in
// place).  grid (rows, block columns J0..J1-1); DAMP (overlap mode): diag += ep + lm diag (dk:1197) here, because the
// factorisation is already running and must not touch columns that are not there yet.
// 256 threads = 8 rows x 32 16-byte units (w_J is even; the strict upper halves of the diagonal tiles are zeros in
// the packed tensor and unread in the pitched one, so whole rows are copied).
template <bool DAMP>
__global__ __launch_bounds__(256) void ba_unpack_kernel(BaView v, int J0, double lm, double ep) {
  const int J = J0 + (int)blockIdx.y, w = pk_width(v.n, J), c0 = CHOL_NB * J;
  const int c = 2 * ((int)threadIdx.x & 31);
  for (int row = c0 + 8 * (int)blockIdx.x + ((int)threadIdx.x >> 5); row <= v.n; row += 8 * (int)gridDim.x) {
    if (c < w) {
      double2 x = *reinterpret_cast<const double2*>(v.psys + pk_colbase(v.n, J) + (size_t)(row - c0) * w + c);
      if (DAMP && row < v.n) {
        if (c0 + c == row) x.x += ep + lm * x.x;
        else if (c0 + c + 1 == row) x.y += ep + lm * x.y;
      }
      *reinterpret_cast<double2*>(v.sys + (size_t)row * v.ld + c0 + c) = x;
    }
  }
}
// runs behind the unpack in stream order: its columns are in memory (end-of-kernel release)
__global__ void ba_set_ready_kernel(int* __restrict__ ready, int b0, int b1, int epoch) {
  const int b = b0 + (int)threadIdx.x;
  if (b < b1) __hip_atomic_store(ready + b, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
void launch_unpack_system(const BaView& v, hipStream_t s) {
  if (v.n <= 0) return;
  const int nb = (v.n + CHOL_NB - 1) / CHOL_NB;
  hipLaunchKernelGGL(ba_unpack_kernel<false>, dim3(min((v.n + 8) / 8, 64), nb), dim3(256), 0, s, v, 0, 0.0, 0.0);
}

void launch_unpack_cols(const BaView& v, int J0, int J1, double lm, double ep, int epoch, hipStream_t s) {
  if (J1 <= J0) return;
  hipLaunchKernelGGL(ba_unpack_kernel<true>, dim3(min((v.n + 8 - CHOL_NB * J0) / 8, 64), J1 - J0), dim3(256), 0, s, v, J0, lm, ep);
  hipLaunchKernelGGL(ba_set_ready_kernel, dim3(1), dim3(64), 0, s, v.ov_ready, J0, J1, epoch);
}

void launch_prep(const BaView& v, const int64_t* ii, const int64_t* jj, hipStream_t s) {
  const size_t lds_bytes = sizeof(int) * prep_lds_ints(v.nbuf, v.E);
  static bool attr_set = false;
  if (lds_bytes <= 150 * 1024) {
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)ba_prep_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      attr_set = true;
    }
    hipLaunchKernelGGL(ba_prep_kernel<true>, dim3(1), dim3(1024), lds_bytes, s, v, ii, jj);
  } else {
    hipLaunchKernelGGL(ba_prep_kernel<false>, dim3(1), dim3(1024), 0, s, v, ii, jj);
  }
}

void launch_build_stage(const BaView& v, const float* poses, const float* disps, const float* intr,
                        const float* sens, const float* targets, const float* weights,
                        const float* eta, const int64_t* ii, const int64_t* jj, bool motion_only,
                        int stage, hipStream_t s) {
  const bool depth = !motion_only && v.M > 0;
  switch (stage) {
    case 0: {
      constexpr int ZB = 192;  // workgroups of the linearisation launch that clear the system instead
      if (depth) {
        const dim3 g(v.M + ZB, v.nch, v.zsplit), b(LIN_THREADS);
        if (v.wide && v.zsplit > 1)
          hipLaunchKernelGGL((ba_lin_kernel<true, true, true>), g, b, 0, s, v, poses, disps, intr, sens, targets, weights, eta, ii, jj, v.M);
        else if (v.wide)
          hipLaunchKernelGGL((ba_lin_kernel<true, true, false>), g, b, 0, s, v, poses, disps, intr, sens, targets, weights, eta, ii, jj, v.M);
        else if (v.zsplit > 1)
          hipLaunchKernelGGL((ba_lin_kernel<true, false, true>), g, b, 0, s, v, poses, disps, intr, sens, targets, weights, eta, ii, jj, v.M);
        else
          hipLaunchKernelGGL((ba_lin_kernel<true, false, false>), g, b, 0, s, v, poses, disps, intr, sens, targets, weights, eta, ii, jj, v.M);
        if (v.zsplit > 1)
          hipLaunchKernelGGL(ba_lin_finish_kernel, dim3(v.M, (v.HW + 255) / 256), dim3(256), 0, s, v, disps, sens, eta);
      } else if (v.E > 0) {
        hipLaunchKernelGGL((ba_lin_kernel<false, false, false>), dim3(v.E + ZB, v.nch), dim3(LIN_THREADS), 0, s, v, poses,
                           disps, intr, sens, targets, weights, eta, ii, jj, v.E);
      } else {
        if (v.packed) (void)hipMemsetAsync(v.psys, 0, sizeof(double) * pk_total(v.n), s);
        else (void)hipMemsetAsync(v.sys, 0, sizeof(double) * (size_t)(v.n + 1) * v.ld, s);
      }
      break;
    }
    case 1:  // (with depth slots the assemble step rides in spare workgroups of the Schur launch, stage 2)
      if (v.E > 0 && !depth)
        hipLaunchKernelGGL(ba_assemble_kernel, dim3(v.E + (solver_preset_words(v) + 1023) / 1024 + solver_preset_tiles(v)), dim3(64), 0, s, v, poses, ii, jj);
      break;
    case 2:
      if (depth) {
        // pixel split so that small graphs still fill the chip (atomics grow with the split)
        const int tiles = (v.HW + SF_TP - 1) / SF_TP;
        int nsplit = 1536 / (v.M > 0 ? v.M : 1);
        nsplit = nsplit < 1 ? 1 : (nsplit > tiles ? tiles : nsplit);
        // dense graphs (mean out-degree >= 12: global BA of a well connected graph, edge-sharded
        // ranks) send their slots of 17..85 entries to the wide kernels
        const int wide = v.wide;
        // sparse slots (<= S2_MAXE edges): Gram tiles per pixel range, then the per-slot fold
        const int asm_units = v.E > 0 ? v.E + (solver_preset_words(v) + 1023) / 1024 + solver_preset_tiles(v) : 0;
        const int asm_x = (asm_units + 4 * v.s2_split - 1) / (4 * v.s2_split);  // spare columns of the grid
        hipLaunchKernelGGL(ba_schur2_kernel, dim3(v.M + asm_x, v.s2_split), dim3(256), 0, s, v, poses, disps, intr, weights, ii, jj,
                           wide, asm_units);
        hipLaunchKernelGGL(ba_schur_fold_kernel, dim3(v.M), dim3(1024), 0, s, v, poses, jj, v.s2_split);
        // slots with more edges than the two kernels above / the SYRK classes below take: block pairs.  Most graphs have
        // none, and the empty launch costs 5 us of every iteration: left out when ba_prep_kernel's hint says so
        // (tag = this call's prepare; read without waiting: not there yet = launch)
        bool none3 = false;
        if (v.hint) {
          const int tag = __atomic_load_n(v.hint, __ATOMIC_ACQUIRE);
          none3 = tag == v.hint_tag && __atomic_load_n(v.hint + 1, __ATOMIC_RELAXED) == 0;
        }
        if (!none3)
          hipLaunchKernelGGL(ba_schur_fused_kernel<true>, dim3(v.M, nsplit), dim3(256), 0, s, v, poses, disps,
                             intr, weights, ii, jj, wide);
        if (wide) {  // dense slots: SYRK straight from the E rows the linearisation wrote (v.Ebuf)
          hipLaunchKernelGGL((ba_syrk3_kernel<SW_MID, 12, 1, 1, true, true>), dim3(v.M, v.sy_ns[0], 1), dim3(768), 0, s, v);
          hipLaunchKernelGGL((ba_syrk3_kernel<SW_BIG, 8, 2, 4, false, false>), dim3(v.M, v.sy_ns[1], 4), dim3(512), 0, s, v);
          hipLaunchKernelGGL(ba_syrk_fold_kernel, dim3(v.M, 34, 2), dim3(256), 0, s, v, v.M * v.sy_ns[0], v.M * v.sy_ns[1]);
        }
      }
      break;
    case 3:
      break;
  }
}

void launch_build(const BaView& v, const float* poses, const float* disps, const float* intr,
                  const float* sens, const float* targets, const float* weights, const float* eta,
                  const int64_t* ii, const int64_t* jj, bool motion_only, hipStream_t s) {
  for (int stage = 0; stage < 4; stage++)
    launch_build_stage(v, poses, disps, intr, sens, targets, weights, eta, ii, jj, motion_only, stage, s);
}

void launch_update(const BaView& v, float* poses, float* disps, const float* intr, const float* weights,
                   const int64_t* ii, const int64_t* jj, const double* x, float* dx_out, float* dz_out,
                   bool motion_only, hipStream_t s, int* status_mirror) {
  if (!motion_only && v.M > 0)
    hipLaunchKernelGGL(ba_backsub_kernel, dim3(v.M, (v.HW + 256 * BSUB_PPT - 1) / (256 * BSUB_PPT)), dim3(256), 0, s, v, poses, disps,
                       intr, weights, ii, jj, x, dz_out);
  hipLaunchKernelGGL(ba_pose_retr_kernel, dim3((v.P + 63) / 64), dim3(64), 0, s, v, poses, x, dx_out, status_mirror);
}

}  // namespace droid
