// ba_internal.hpp -- workspace layout of one `ba` call, shared by the host API and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

namespace droid {

constexpr int LIN_THREADS = 256;
#ifdef LIN_PPT_AB
constexpr int LIN_PPT = LIN_PPT_AB;              // A/B builds (tools/ab_test.sh)
#else
constexpr int LIN_PPT = 2;                       // pixels per thread per chunk
#endif
constexpr int LIN_CP = LIN_THREADS * LIN_PPT;    // pixels per workgroup chunk
constexpr int CHOL_NB = 64;                      // Cholesky block size
// row pitch of the augmented system in doubles: 128-byte rows, so 64-column tiles never share a cache line
__host__ __device__ inline int chol_ld(int n) { return (n + 1 + 15) & ~15; }
// ints of hand-off flags (done[], dver[], abort) of an n x n solve
__host__ __device__ inline size_t chol_flag_words(int n) { return 2 * ((size_t)(n + 1 + CHOL_NB - 1) / CHOL_NB) + 8; }
// lower-triangle tiles of the augmented system in column-major order: tile (bi, bj), bi >= bj
__host__ __device__ inline int chol_tile_index(int nrb, int bi, int bj) { return bj * nrb - bj * (bj - 1) / 2 + (bi - bj); }
__host__ __device__ inline size_t chol_tiles(int n) {
  const int nb = (n + CHOL_NB - 1) / CHOL_NB, nrb = (n + 1 + CHOL_NB - 1) / CHOL_NB;
  return (size_t)chol_tile_index(nrb, nb, nb);  // = sum_{j<nb} (nrb - j)
}
// `ldiag` scratch of the single-launch factorisation, in 64x64 tiles:
//   [nb]         factored diagonal tiles L_jj (+ the inverses of their 16x16 diagonal blocks in the upper part),
//   [nb]         M_j = L[j+1,j] L_jj^-1 (the back-substitution multiplies by them instead of solving),
//   [chol_tiles] hand-over slots `lfin` of the final panel tiles (data-tagged: preset to 0xFF bytes, a 16-column
//                strip is there when its bytes are not)
__host__ __device__ inline size_t chol_mbuf_offset(int n) { return ((size_t)(n + CHOL_NB - 1) / CHOL_NB) * CHOL_NB * CHOL_NB; }
__host__ __device__ inline size_t chol_lfin_offset(int n) { return 2 * chol_mbuf_offset(n); }
__host__ __device__ inline size_t chol_ldiag_doubles(int n) { return chol_lfin_offset(n) + chol_tiles(n) * CHOL_NB * CHOL_NB; }

// Packed lower triangle + rhs row, BLOCK-COLUMN major (what the ranks all-reduce): block column J = columns
// 64 J .. 64 J + w_J - 1 (w_J = 64, the last one n - 64 J) holds rows 64 J .. n (row n = rhs) as rows of w_J doubles.
// The factorisation consumes block columns left to right, so a PREFIX of this tensor is all its first block columns
// need: the collective can be cut into column chunks that arrive in the order the solver wants them (overlap mode).
// (The strict upper halves of the diagonal tiles are carried along as zeros: 4 % of the bytes.)
__host__ __device__ inline size_t pk_colbase(int n, int J) {
  return (size_t)CHOL_NB * ((size_t)J * (size_t)(n + 1) - (size_t)CHOL_NB * (size_t)J * (size_t)(J - 1) / 2);
}
__host__ __device__ inline int pk_width(int n, int J) { return (n - CHOL_NB * J) < CHOL_NB ? (n - CHOL_NB * J) : CHOL_NB; }
__host__ __device__ inline size_t pk_total(int n) {
  if (n <= 0) return 0;
  const int nb = (n + CHOL_NB - 1) / CHOL_NB;
  return pk_colbase(n, nb - 1) + (size_t)(n + 1 - CHOL_NB * (nb - 1)) * (size_t)pk_width(n, nb - 1);
}
__host__ __device__ inline size_t pk_index(int n, int gi, int gj) {   // gi >= 64 (gj / 64); gi = n: rhs row
  const int J = gj / CHOL_NB;
  return pk_colbase(n, J) + (size_t)(gi - CHOL_NB * J) * (size_t)pk_width(n, J) + (size_t)(gj - CHOL_NB * J);
}
// the same in 32-bit arithmetic for the kernels' atomics (the packed tensor has < 2^31 elements up to n = 65 000)
__host__ __device__ inline unsigned pk_index32(int n, int gi, int gj) {
  const int J = gj >> 6, c0 = J << 6;
  const int w = (n - c0) < CHOL_NB ? (n - c0) : CHOL_NB;
  return (unsigned)(CHOL_NB * (J * (n + 1) - (CHOL_NB / 2) * J * (J - 1)) + (gi - c0) * w + (gj - c0));
}

enum { HDR_STATUS = 0, HDR_M = 1, HDR_CHOL_FAIL = 2, HDR_NENT = 3, HDR_NWORK = 4,
       HDR_NC1 = 5, HDR_NC2 = 6,  // slots served by the two SYRK variants (classes 1, 2)
       HDR_WORDS = 16 };
enum { STATUS_BAD_INDEX = 1, STATUS_ETA_ROWS = 2, STATUS_CHOL_FAIL = 4, STATUS_CHOL_STALL = 8 };

// Plain-old-data view of the workspace, passed to kernels by value.
struct BaView {
  int E, nbuf, H, W, HW;
  int t0, t1, P, M;        // window, poses in it, depth slots the host sized the buffers for
  int n, ld;               // n = 6P unknowns; system is (n+1) x ld, ld = roundup16(n+1) (128-byte rows), row n = rhs
  int nch;                 // pixel chunks of the linearisation kernel
  int own0, own1;          // frames whose depth this rank owns
  int motion_only;         // prep: validate indices only, no depth slots
  int* hdr;                // [HDR_WORDS] status, slot count, ...
  int* slot_of;            // [nbuf] frame -> depth slot or -1
  int* kx;                 // [nbuf] slot -> frame (sorted)
  int* seg_ptr;            // [nbuf+1] CSR of edges by slot
  int* seg_edge;           // [E]
  int* cursor;             // [nbuf+1] scratch
  int* ent_ptr;            // [nbuf+1] CSR of Schur entries by slot
  int* ent_row;            // [M+E] row of Erows
  int* ent_pose;           // [M+E] pose index in the window
  int* wk_ptr;             // [nbuf+1] scratch (slot sizes while sorting)
  int* order;              // [nbuf+1] slots sorted by descending edge count: big slots are dispatched first
  int* xtmp;               // [3(E+1)] prep scratch: unsorted segment fill, window flag and slot of every sorted position
  int* cls_list;           // [2][nbuf+2] the slots of SYRK class 1 / class 2, ascending (ba_syrk3_kernel)
  int* gt_ptr;             // [nbuf+2] first partial-sum tile of a slot served by ba_schur2_kernel (exclusive scan)
  double* Gpart;           // [tiles][s2_split][256] fp64 partial sums of those slots' Gram tiles, one per pixel range
  int s2_split;            // pixel ranges per slot of ba_schur2_kernel
  float* Hpart;            // [E][nch][32] per-(edge,chunk) partial Hjj (21) + vj (6)
  float* Q;                // [M][HW] 1/C
  float* w;                // [M][HW]
  float* Erows;            // unused (nullptr): sparse slots recompute their E rows where they are consumed
  float* Ebuf;             // dense graphs only (`wide`): unscaled E rows, 6*(M+E)*HW floats, per slot tiled by 32-pixel stage
                           // (ba_kernels.hip::ebuf_index), written by the linearisation and consumed by the SYRK-only Schur kernels
  float* sy_part[2];       // dense graphs: fp32 tiles of S per (slot of SYRK class 1 / 2, pixel range), summed by ba_syrk_fold_kernel
  int sy_ns[2];            // pixel splits (grid y) of the two SYRK launches
  int wide;                // host decision: mean out-degree >= 12 (dense global BA, edge-sharded ranks)
  int zsplit;              // linearisation: workgroups per (slot, pixel chunk), each takes a range of the slot's edges
  float* zpart;            // zsplit > 1: [zsplit][M][8][HW] partial C, w and self-row sums of those workgroups
  double* sys;             // [n+1][ld] reduced camera system, lower triangle, row n = rhs
  double* psys;            // the same entries packed block-column major (pk_index): what a rank's build phase
                           // accumulates into when the systems are all-reduced
  int packed;              // 1: the build kernels add into psys (multi-GPU), 0: straight into sys
  double* xsol;            // [ld] solve scratch / solution
  float* dx;               // [P][6]
  int* bs_flags;           // [chol_flag_words(n)] hand-off flags of the single-launch factorisation
  double* ldiag;           // [ceil(n/64)][64][64] factored diagonal tiles, then the hand-over slots of the panel tiles
  int* hint;               // host-side launch hints (droid_ba_attach_launch_hints): 2 ints of page-locked host memory {tag, slots of
                           // Schur class 3}, written by ba_prep_kernel; nullptr: none.  hint_tag = tag of this workspace's last prepare
  int hint_tag;
  int* ov_ready;           // [block columns of the system] overlap mode: epoch of the last iteration whose reduced columns are in `sys`
};

struct BaSizes {
  size_t total;
};

// Carves `ws` (may be null: size query only) into the view.  Host-only arithmetic.
#ifndef SY_WGS
#define SY_WGS 1024  // (slot, pixel range) workgroups of the class-1 SYRK launch: four per CU balance the uneven slots (A/B 512 / 768 / 1024 / 1536 / 2048: 270 / 265 / 253 / 262 / 270 us at 256 slots)
#endif
inline size_t ba_carve(BaView& v, void* ws, int E, int nbuf, int H, int W, int t0, int t1, int M) {
  v.E = E; v.nbuf = nbuf; v.H = H; v.W = W; v.HW = H * W;
  v.t0 = t0; v.t1 = t1; v.P = t1 - t0; v.M = M;
  v.n = 6 * v.P; v.ld = chol_ld(v.n);
  v.nch = (v.HW + LIN_CP - 1) / LIN_CP;
  v.own0 = 0; v.own1 = nbuf; v.motion_only = 0;
  v.hint = nullptr; v.hint_tag = 0;
  size_t off = 0;
  char* base = static_cast<char*>(ws);
  auto take = [&](size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += (bytes + 255) & ~size_t(255);
    return p;
  };
  v.hdr = static_cast<int*>(take(sizeof(int) * HDR_WORDS));
  v.slot_of = static_cast<int*>(take(sizeof(int) * (nbuf + 1)));
  v.kx = static_cast<int*>(take(sizeof(int) * (nbuf + 1)));
  v.seg_ptr = static_cast<int*>(take(sizeof(int) * (nbuf + 2)));
  v.seg_edge = static_cast<int*>(take(sizeof(int) * (E + 1)));
  v.cursor = static_cast<int*>(take(sizeof(int) * (nbuf + 2)));
  v.ent_ptr = static_cast<int*>(take(sizeof(int) * (nbuf + 2)));
  v.ent_row = static_cast<int*>(take(sizeof(int) * ((size_t)M + E + 1)));
  v.ent_pose = static_cast<int*>(take(sizeof(int) * ((size_t)M + E + 1)));
  v.wk_ptr = static_cast<int*>(take(sizeof(int) * (nbuf + 2)));
  v.order = static_cast<int*>(take(sizeof(int) * (nbuf + 2)));
  v.gt_ptr = static_cast<int*>(take(sizeof(int) * (nbuf + 2)));
  v.cls_list = static_cast<int*>(take(sizeof(int) * 2 * (nbuf + 2)));
  v.xtmp = static_cast<int*>(take(sizeof(int) * 3 * ((size_t)E + 1)));
  v.Hpart = static_cast<float*>(take(sizeof(float) * ((size_t)E * v.nch * 32 + 32)));
  v.Q = static_cast<float*>(take(sizeof(float) * ((size_t)M * v.HW + 4)));
  v.w = static_cast<float*>(take(sizeof(float) * ((size_t)M * v.HW + 4)));
  v.Erows = nullptr;  // E rows are recomputed where they are consumed, never stored ...
  v.wide = (M > 0 && (long)E >= 12l * M && (v.HW % 32) == 0) ? 1 : 0;
  v.Ebuf = nullptr;   // ... except for dense graphs, where the recomputation would be repeated per output share
  if (v.wide) v.Ebuf = static_cast<float*>(take(sizeof(float) * (6 * ((size_t)M + E) * v.HW + 64)));
  v.sy_part[0] = v.sy_part[1] = nullptr;
  v.sy_ns[0] = v.sy_ns[1] = 0;
  if (v.wide) {
    // pixel splits of the SYRK launches: enough (slot, range) workgroups to fill 256 CUs four times over (class 1: one 12-wave workgroup per
    // CU; class 2: four shares per pair); a split costs a set of partial tiles, no atomics.  A launch has M * ns pairs at most:
    // 136 (class 1: <= 256 rows) or 528 (class 2: <= 512 rows) tiles of 1 KB each
    const int stages = v.HW / 32;
    int nsw = (SY_WGS + M - 1) / M, nsb = (256 + 4 * M - 1) / (4 * M);
    nsw = nsw < 2 ? 2 : (nsw > stages ? stages : nsw);
    nsb = nsb < 2 ? 2 : (nsb > stages ? stages : nsb);
    if (nsw > 16) nsw = 16;  // SY_MAXSPLIT (ba_kernels.hip): the kernels never use more ranges per slot
    if (nsb > 16) nsb = 16;
    v.sy_ns[0] = nsw; v.sy_ns[1] = nsb;
    v.sy_part[0] = static_cast<float*>(take(sizeof(float) * ((size_t)M * nsw * 136 * 256 + 64)));
    v.sy_part[1] = static_cast<float*>(take(sizeof(float) * ((size_t)M * nsb * 528 * 256 + 64)));
  }
  // few depth slots (edge-sharded ranks, small windows): split every slot's edges over several
  // workgroups so that the linearisation still fills the chip
  v.zsplit = 1;
  if (M > 0) {
    const int z = 1536 / (M * v.nch);
    v.zsplit = z < 1 ? 1 : (z > 8 ? 8 : z);
  }
  v.zpart = nullptr;
  if (v.zsplit > 1) v.zpart = static_cast<float*>(take(sizeof(float) * ((size_t)v.zsplit * M * 8 * v.HW + 64)));
  // Schur complement of sparse slots (<= S2_MAXE edges): every (slot, pixel range) workgroup leaves the fp64
  // Gram tiles of its E rows here; tiles(n edges) <= 1.75 n + 1 (7 row tiles of 16 at 16 edges)
  v.s2_split = 1;
  v.Gpart = nullptr;
  if (M > 0) {
    const int ptiles = (v.HW + 63) / 64;
    static const int wg_target = getenv("DROID_S2_WGS") ? atoi(getenv("DROID_S2_WGS")) : 1536;  // diagnostics
    int ns = wg_target / M;
    v.s2_split = ns < 1 ? 1 : (ns > ptiles ? ptiles : ns);
    size_t tiles = (size_t)E * 7 / 4 + (size_t)M + 1;
    if (tiles > (size_t)28 * M) tiles = (size_t)28 * M;
    v.Gpart = static_cast<double*>(take(sizeof(double) * (tiles * v.s2_split * 256 + 64)));
  }
  v.sys = static_cast<double*>(take(sizeof(double) * ((size_t)(v.n + 1) * v.ld + 8)));
  v.psys = static_cast<double*>(take(sizeof(double) * (pk_total(v.n) + 8)));
  v.packed = 0;
  v.xsol = static_cast<double*>(take(sizeof(double) * ((size_t)v.ld + 1)));
  v.bs_flags = static_cast<int*>(take(sizeof(int) * chol_flag_words(v.n)));   // directly after xsol: one fill presets both
  v.ldiag = static_cast<double*>(take(sizeof(double) * chol_ldiag_doubles(v.n)));
  v.dx = static_cast<float*>(take(sizeof(float) * ((size_t)v.n + 8)));
  v.ov_ready = static_cast<int*>(take(sizeof(int) * ((size_t)(v.n + 1 + CHOL_NB - 1) / CHOL_NB + 8)));
  return off;
}

// 4-byte words from xsol through the end of bs_flags (contiguous in the workspace): what one iteration's solve
// expects preset to 0xFF bytes.  With edges, the assemble kernel's spare workgroups do it (and zero the failure
// flag); without, droid_ba_solve_update falls back to fills.
__host__ __device__ inline int solver_preset_words(const BaView& v) {
  return (int)((reinterpret_cast<const char*>(v.bs_flags) - reinterpret_cast<const char*>(v.xsol)) / 4 +
               (long)chol_flag_words(v.n));
}
// 32 KB blocks of the panel-tile hand-over slots (ldiag + chol_lfin_offset), preset to 0xFF bytes the same way
__host__ __device__ inline int solver_preset_tiles(const BaView& v) { return v.n > 0 ? (int)chol_tiles(v.n) : 0; }

// kernels' launchers (ba_kernels.hip / chol.hip)
void launch_prep(const BaView& v, const int64_t* ii, const int64_t* jj, hipStream_t s);
// multi-GPU: expand the (all-reduced) packed system into the pitched matrix the solver factors in place
void launch_unpack_system(const BaView& v, hipStream_t s);
void launch_build(const BaView& v, const float* poses, const float* disps, const float* intr,
                  const float* sens, const float* targets, const float* weights, const float* eta,
                  const int64_t* ii, const int64_t* jj, bool motion_only, hipStream_t s);
// stage: 0 memset+linearise, 1 assemble, 2 schur, 3 ev (measurement support)
void launch_build_stage(const BaView& v, const float* poses, const float* disps, const float* intr,
                        const float* sens, const float* targets, const float* weights,
                        const float* eta, const int64_t* ii, const int64_t* jj, bool motion_only,
                        int stage, hipStream_t s);
// status_mirror: optional device-visible host words {status, depth slots} written at the end of the iteration
void launch_update(const BaView& v, float* poses, float* disps, const float* intr, const float* weights,
                   const int64_t* ii, const int64_t* jj, const double* x, float* dx_out, float* dz_out,
                   bool motion_only, hipStream_t s, int* status_mirror = nullptr);
// In-place damped Cholesky of the lower triangle of sys ((n+1) x ld, row n = rhs) + solve -> x [n].
// flags [chol_flag_words(n)] and ldiag [chol_ldiag_doubles(n)]: scratch of the single-launch factorisation
// (null: one launch per block column).  launch_chol_solve presets x and flags itself unless told that it has
// been done; callers of the two halves preset them with 0xFF bytes before launch_chol_factor.
void launch_chol_solve(double* sys, int n, int ld, double lm, double ep, double* x, int* fail_flag,
                       int* flags, double* ldiag, hipStream_t s, bool preset_done = false);

// Overlap mode (multi-GPU): block columns [J0, J1) of the all-reduced packed system -> pitched matrix, damped; then
// ready[J0..J1) = epoch for the factorisation that is already running (launch_chol_factor_overlap).
void launch_unpack_cols(const BaView& v, int J0, int J1, double lm, double ep, int epoch, hipStream_t s);
bool launch_chol_factor_overlap(double* sys, int n, int ld, int* fail_flag, int* flags, double* ldiag,
                                const int* ready, int epoch, hipStream_t s);
// launch_chol_factor returns whether the single-launch kernel ran; pass that to launch_chol_backsolve
bool launch_chol_factor(double* sys, int n, int ld, double lm, double ep, int* fail_flag, int* flags,
                        double* ldiag, hipStream_t s);
void launch_chol_backsolve(double* sys, int n, int ld, double* x, int* flags, double* ldiag, int* err,
                           hipStream_t s, bool factor_single);

}  // namespace droid
