// api.hip -- the extern "C" boundary (include/droid_backends_hip.h).  Host-side only: argument
// checks, workspace carving, kernel launches on the caller's stream.  No synchronisation except
// in droid_ba_status.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>

#include "../../include/droid_backends_hip.h"
#include "ba_internal.hpp"

namespace droid {
// corr.hip
int launch_corr_index_forward(const void* volume, const float* coords, void* corr, int B, int H1,
                              int W1, int H2, int W2, int r, int dtype, hipStream_t s);
int launch_corr_pyramid_forward(const void* const* volumes, const float* coords, void* corr, int B, int H1, int W1,
                                int r, int levels, int dtype, hipStream_t s);
int launch_corr_index_backward(const float* coords, const void* corr_grad, void* volume_grad, int B,
                               int H1, int W1, int H2, int W2, int r, int dtype, hipStream_t s);
int launch_altcorr_pyramid_forward(const void* const* levels_dev, const int64_t* ii, const int64_t* jj,
                                   const float* coords, float* corr, int E, int frames, int H, int W, int C,
                                   int r, int nlevels, int dtype, hipStream_t s);
int launch_altcorr_forward(const void* f1, const void* f2, const float* coords, void* corr, int B,
                           int N, int H1, int W1, int H2, int W2, int C, int r, int dtype,
                           hipStream_t s);
int launch_altcorr_backward(const float* f1, const float* f2, const float* coords,
                            const float* corr_grad, float* f1g, float* f2g, int B, int N, int H1,
                            int W1, int H2, int W2, int C, int r, hipStream_t s);
// geom.hip
void launch_frame_distance(const float* poses, const float* disps, const float* intr,
                           const int64_t* ii, const int64_t* jj, int E, int nbuf, int H, int W,
                           float beta, float* dist, hipStream_t s);
void launch_frame_distance_matrix(const float* poses, const float* disps, const float* intr, int n, int H, int W,
                                  float beta, float* dist, hipStream_t s);
void launch_projmap(const float* poses, const float* disps, const float* intr, const int64_t* ii,
                    const int64_t* jj, int E, int nbuf, int H, int W, float* coords, float* valid,
                    hipStream_t s);
void launch_iproj(const float* poses, const float* disps, const float* intr, int nm, int H, int W,
                  float* points, hipStream_t s);
void launch_depth_filter(const float* poses, const float* disps, const float* intr,
                         const int64_t* ix, const float* thresh, int num, int nbuf, int H, int W,
                         float* counter, hipStream_t s);
// chol.hip
void launch_chol_pack(const double* A, const double* b, double* S, int n, int ld, hipStream_t s);
void launch_reproject_motion(const float* poses, const float* disps, const float* intr, int intr_stride,
                             const int64_t* ii, const int64_t* jj, const float* target, int E, int nbuf, int H,
                             int W, float* coords, float* valid, float* motn, hipStream_t s);
}  // namespace droid

using namespace droid;

static thread_local char g_err[512] = "";

// workspace -> host-visible status words (droid_ba_attach_status_mirror)
static std::map<const void*, int*> g_mirror;
static std::mutex g_mirror_mu;
// workspace -> launch hints (droid_ba_attach_launch_hints) + tag of the last prepare
struct HintState { int* ptr; int tag; };
static std::map<const void*, HintState> g_hint;
static void hints_of(BaView& v, const void* ws, bool new_call) {
  std::lock_guard<std::mutex> lock(g_mirror_mu);
  auto it = g_hint.find(ws);
  if (it == g_hint.end()) return;
  if (new_call) it->second.tag = it->second.tag >= (1 << 30) ? 1 : it->second.tag + 1;
  v.hint = it->second.ptr;
  v.hint_tag = it->second.tag;
}
static int* mirror_of(const void* ws) {
  std::lock_guard<std::mutex> lock(g_mirror_mu);
  auto it = g_mirror.find(ws);
  return it == g_mirror.end() ? nullptr : it->second;
}

static int fail(int code, const char* fmt, const char* what) {
  snprintf(g_err, sizeof(g_err), fmt, what);
  return code;
}

static int check_hip(const char* where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return DROID_E_HIP;
  }
  return DROID_OK;
}

extern "C" {

int droid_abi_version(void) { return DROID_ABI_VERSION; }
const char* droid_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------- correlation
int droid_corr_index_forward(const void* volume, const float* coords, void* corr, int B, int H1,
                             int W1, int H2, int W2, int radius, int dtype, void* stream) {
  if (B < 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || radius < 0 || radius > 7)
    return fail(DROID_E_ARG, "corr_index_forward: bad %s", "shape/radius");
  if (dtype < DROID_F16 || dtype > DROID_F64) return fail(DROID_E_ARG, "corr_index_forward: bad %s", "dtype");
  if (B == 0) return DROID_OK;
  if (!volume || !coords || !corr) return fail(DROID_E_ARG, "corr_index_forward: null %s", "pointer");
  int rc = launch_corr_index_forward(volume, coords, corr, B, H1, W1, H2, W2, radius, dtype,
                                     (hipStream_t)stream);
  if (rc) return fail(rc, "corr_index_forward: %s", "unsupported configuration");
  return check_hip("corr_index_forward");
}

int droid_corr_pyramid_forward(const void* const* volumes, const float* coords, void* corr, int B, int H1, int W1,
                               int radius, int levels, int dtype, void* stream) {
  if (B < 0 || H1 <= 0 || W1 <= 0 || levels < 1) return fail(DROID_E_ARG, "corr_pyramid_forward: bad %s", "shape");
  if (dtype < DROID_F16 || dtype > DROID_F64) return fail(DROID_E_ARG, "corr_pyramid_forward: bad %s", "dtype");
  if (B == 0) return DROID_OK;
  if (!volumes || !coords || !corr) return fail(DROID_E_ARG, "corr_pyramid_forward: null %s", "pointer");
  for (int l = 0; l < levels; l++)
    if (!volumes[l]) return fail(DROID_E_ARG, "corr_pyramid_forward: null %s", "pyramid level");
  int rc = launch_corr_pyramid_forward(volumes, coords, corr, B, H1, W1, radius, levels, dtype, (hipStream_t)stream);
  if (rc) return fail(rc, "corr_pyramid_forward: %s", "unsupported configuration (radius 3 or 4, levels that divide the map)");
  return check_hip("corr_pyramid_forward");
}

int droid_corr_index_backward(const float* coords, const void* corr_grad, void* volume_grad, int B,
                              int H1, int W1, int H2, int W2, int radius, int dtype, void* stream) {
  if (B < 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || radius < 0 || radius > 7)
    return fail(DROID_E_ARG, "corr_index_backward: bad %s", "shape/radius");
  if (dtype < DROID_F16 || dtype > DROID_F64) return fail(DROID_E_ARG, "corr_index_backward: bad %s", "dtype");
  if (B == 0) return DROID_OK;
  if (!coords || !corr_grad || !volume_grad) return fail(DROID_E_ARG, "corr_index_backward: null %s", "pointer");
  int rc = launch_corr_index_backward(coords, corr_grad, volume_grad, B, H1, W1, H2, W2, radius,
                                      dtype, (hipStream_t)stream);
  if (rc) return fail(rc, "corr_index_backward: %s", "unsupported configuration");
  return check_hip("corr_index_backward");
}

int droid_altcorr_forward(const void* fmap1, const void* fmap2, const float* coords, void* corr,
                          int B, int N, int H1, int W1, int H2, int W2, int C, int radius,
                          int dtype, void* stream) {
  if (B < 0 || N <= 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || C <= 0 || radius < 0 || radius > 7)
    return fail(DROID_E_ARG, "altcorr_forward: bad %s", "shape/radius");
  if (dtype < DROID_F16 || dtype > DROID_F64) return fail(DROID_E_ARG, "altcorr_forward: bad %s", "dtype");
  if (B == 0) return DROID_OK;
  if (!fmap1 || !fmap2 || !coords || !corr) return fail(DROID_E_ARG, "altcorr_forward: null %s", "pointer");
  int rc = launch_altcorr_forward(fmap1, fmap2, coords, corr, B, N, H1, W1, H2, W2, C, radius,
                                  dtype, (hipStream_t)stream);
  if (rc) return fail(rc, "altcorr_forward: %s", "unsupported configuration");
  return check_hip("altcorr_forward");
}

static int altcorr_pyramid_any(const void* const* pyramid, const int64_t* ii, const int64_t* jj,
                               const float* coords, float* corr, int E, int frames, int H, int W,
                               int C, int radius, int levels, int dtype, void* stream) {
  if (E < 0 || frames <= 0 || H <= 0 || W <= 0 || C <= 0 || levels < 1 || levels > 4)
    return fail(DROID_E_ARG, "altcorr_pyramid_forward: bad %s", "shape");
  if (E == 0) return DROID_OK;
  if (!pyramid || !ii || !jj || !coords || !corr) return fail(DROID_E_ARG, "altcorr_pyramid_forward: null %s", "pointer");
  for (int l = 0; l < levels; l++)
    if (!pyramid[l]) return fail(DROID_E_ARG, "altcorr_pyramid_forward: null %s", "pyramid level");
  int rc = launch_altcorr_pyramid_forward(pyramid, ii, jj, coords, corr, E, frames, H, W, C, radius, levels, dtype,
                                          (hipStream_t)stream);
  if (rc) return fail(rc, "altcorr_pyramid_forward: %s", "unsupported configuration (C % 16 (fp32) / 32 (fp16) == 0, C <= 128, r in {3,4})");
  return check_hip("altcorr_pyramid_forward");
}

int droid_altcorr_pyramid_forward(const float* const* pyramid, const int64_t* ii, const int64_t* jj,
                                  const float* coords, float* corr, int E, int frames, int H, int W,
                                  int C, int radius, int levels, void* stream) {
  return altcorr_pyramid_any(reinterpret_cast<const void* const*>(pyramid), ii, jj, coords, corr, E, frames, H, W, C,
                             radius, levels, DROID_F32, stream);
}

int droid_altcorr_pyramid_forward_f16(const void* const* pyramid, const int64_t* ii, const int64_t* jj,
                                      const float* coords, float* corr, int E, int frames, int H, int W,
                                      int C, int radius, int levels, void* stream) {
  return altcorr_pyramid_any(pyramid, ii, jj, coords, corr, E, frames, H, W, C, radius, levels, DROID_F16, stream);
}

int droid_altcorr_backward(const float* fmap1, const float* fmap2, const float* coords,
                           const float* corr_grad, float* fmap1_grad, float* fmap2_grad, int B,
                           int N, int H1, int W1, int H2, int W2, int C, int radius, void* stream) {
  if (B < 0 || N <= 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || C <= 0 || radius < 0 || radius > 7)
    return fail(DROID_E_ARG, "altcorr_backward: bad %s", "shape/radius");
  if (B == 0) return DROID_OK;
  if (!fmap1 || !fmap2 || !coords || !corr_grad || !fmap1_grad || !fmap2_grad)
    return fail(DROID_E_ARG, "altcorr_backward: null %s", "pointer");
  int rc = launch_altcorr_backward(fmap1, fmap2, coords, corr_grad, fmap1_grad, fmap2_grad, B, N,
                                   H1, W1, H2, W2, C, radius, (hipStream_t)stream);
  if (rc) return fail(rc, "altcorr_backward: %s", "unsupported configuration");
  return check_hip("altcorr_backward");
}

// ---------------------------------------------------------------------------- bundle adjustment
static int ba_check(int E, int nbuf, int H, int W, int t0, int t1, int M, int motion_only) {
  if (E < 0 || nbuf <= 0 || H <= 0 || W <= 0) return fail(DROID_E_ARG, "ba: bad %s", "sizes");
  if (t0 < 0 || t1 <= t0 || t1 > nbuf) return fail(DROID_E_ARG, "ba: bad %s", "window [t0,t1)");
  if (M < 0 || M > nbuf) return fail(DROID_E_ARG, "ba: bad %s", "depth-slot count (eta rows)");
  if (!motion_only && M == 0) return fail(DROID_E_ARG, "ba: %s", "eta has no rows");
  return DROID_OK;
}

size_t droid_ba_workspace_bytes(int E, int nbuf, int H, int W, int t0, int t1, int M) {
  BaView v;
  if (E < 0 || nbuf <= 0 || H <= 0 || W <= 0 || t1 <= t0 || M < 0) return 0;
  return ba_carve(v, nullptr, E, nbuf, H, W, t0, t1, M);
}

static int ba_view(BaView& v, void* ws, size_t ws_bytes, int E, int nbuf, int H, int W, int t0,
                   int t1, int M, int motion_only) {
  int rc = ba_check(E, nbuf, H, W, t0, t1, M, motion_only);
  if (rc) return rc;
  if (!ws) return fail(DROID_E_ARG, "ba: null %s", "workspace");
  const size_t need = ba_carve(v, ws, E, nbuf, H, W, t0, t1, motion_only ? 0 : M);
  if (ws_bytes < need) return fail(DROID_E_WORKSPACE, "ba: %s", "workspace too small");
  return DROID_OK;
}

int droid_ba_prepare(const int64_t* ii, const int64_t* jj, int E, int nbuf, int H, int W, int M,
                     int t0, int t1, int own0, int own1, int motion_only, void* workspace,
                     size_t workspace_bytes, void* stream) {
  BaView v;
  int rc = ba_view(v, workspace, workspace_bytes, E, nbuf, H, W, t0, t1, M, motion_only);
  if (rc) return rc;
  if (E > 0 && (!ii || !jj)) return fail(DROID_E_ARG, "ba: null %s", "edge arrays");
  v.own0 = own0 < 0 ? 0 : own0;
  v.own1 = own1 > nbuf ? nbuf : own1;
  v.motion_only = motion_only ? 1 : 0;
  hints_of(v, workspace, true);
  launch_prep(v, ii, jj, (hipStream_t)stream);
  // overlap mode: no reduced rows of any iteration of this call are in `sys` yet (epochs start at 1)
  (void)hipMemsetAsync(v.ov_ready, 0, sizeof(int) * ((size_t)(v.n + 1 + CHOL_NB - 1) / CHOL_NB), (hipStream_t)stream);   // (>= block columns)
  return check_hip("ba_prepare");
}

} // extern "C"
static int ba_build_impl(const float* poses, const float* disps, const float* intrinsics,
                         const float* disps_sens, const float* targets, const float* weights,
                         const float* eta, const int64_t* ii, const int64_t* jj, int E, int nbuf, int H,
                         int W, int M, int t0, int t1, int motion_only, void* workspace,
                         size_t workspace_bytes, void* stream, int packed) {
  BaView v;
  int rc = ba_view(v, workspace, workspace_bytes, E, nbuf, H, W, t0, t1, M, motion_only);
  if (rc) return rc;
  v.packed = packed;
  hints_of(v, workspace, false);
  if (!poses || !disps || !intrinsics) return fail(DROID_E_ARG, "ba: null %s", "state pointer");
  if (E > 0 && (!targets || !weights || !ii || !jj)) return fail(DROID_E_ARG, "ba: null %s", "edge data");
  if (!motion_only && (!eta || !disps_sens)) return fail(DROID_E_ARG, "ba: null %s", "eta/disps_sens");
  launch_build(v, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj,
               motion_only != 0, (hipStream_t)stream);
  return check_hip("ba_build");
}
extern "C" {

int droid_ba_build(const float* poses, const float* disps, const float* intrinsics,
                   const float* disps_sens, const float* targets, const float* weights,
                   const float* eta, const int64_t* ii, const int64_t* jj, int E, int nbuf, int H,
                   int W, int M, int t0, int t1, int motion_only, void* workspace,
                   size_t workspace_bytes, void* stream) {
  return ba_build_impl(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, E, nbuf, H, W, M, t0, t1,
                       motion_only, workspace, workspace_bytes, stream, 0);
}

int droid_ba_build_packed(const float* poses, const float* disps, const float* intrinsics,
                          const float* disps_sens, const float* targets, const float* weights,
                          const float* eta, const int64_t* ii, const int64_t* jj, int E, int nbuf, int H,
                          int W, int M, int t0, int t1, int motion_only, void* workspace,
                          size_t workspace_bytes, void* stream) {
  return ba_build_impl(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, E, nbuf, H, W, M, t0, t1,
                       motion_only, workspace, workspace_bytes, stream, 1);
}

double* droid_ba_packed_system(void* workspace, int E, int nbuf, int H, int W, int t0, int t1, int M,
                               size_t* n_elements) {
  BaView v;
  if (!workspace || t1 <= t0) return nullptr;
  ba_carve(v, workspace, E, nbuf, H, W, t0, t1, M);
  if (n_elements) *n_elements = pk_total(v.n);
  return v.psys;
}

int droid_ba_unpack_system(int E, int nbuf, int H, int W, int M, int t0, int t1, int motion_only, void* workspace,
                           size_t workspace_bytes, void* stream) {
  BaView v;
  int rc = ba_view(v, workspace, workspace_bytes, E, nbuf, H, W, t0, t1, M, motion_only);
  if (rc) return rc;
  launch_unpack_system(v, (hipStream_t)stream);
  return check_hip("ba_unpack_system");
}

// ---- overlap of the all-reduce with the solve (multi-GPU) ------------------------------------------------------
// Chunks of the packed system by block columns of 64: 1, 1, 2, 3, 4, 5, ... block columns.  The first ones are
// small in columns (a block column on the left is the tallest: column 0 alone is 8 % of the bytes) because the
// factorisation starts on them; the later, larger chunks arrive long before the diagonal chain reaches their columns.
static int overlap_bounds(int n, int max_chunks, int* bcol /*[max_chunks + 1]*/) {
  const int nb = (n + CHOL_NB - 1) / CHOL_NB;
  int nc = 0, b = 0, size = 1, ones = 0;
  bcol[0] = 0;
  while (b < nb && nc < max_chunks) {
    b = (nc + 1 == max_chunks || b + size >= nb) ? nb : b + size;
    bcol[++nc] = b;
    if (++ones >= 2) size++;
  }
  return nc;
}

int droid_ba_overlap_plan(int t0, int t1, int max_chunks, int* nchunks_out, size_t* packed_offsets) {
  if (t1 <= t0 || max_chunks < 1 || max_chunks > 32 || !nchunks_out || !packed_offsets)
    return fail(DROID_E_ARG, "ba_overlap_plan: bad %s", "argument");
  const int n = 6 * (t1 - t0);
  int brow[33];
  const int nc = overlap_bounds(n, max_chunks, brow);
  const int nb = (n + CHOL_NB - 1) / CHOL_NB;
  for (int c = 0; c <= nc; c++) packed_offsets[c] = brow[c] >= nb ? pk_total(n) : pk_colbase(n, brow[c]);
  *nchunks_out = nc;
  return DROID_OK;
}

int droid_ba_unpack_chunk(int E, int nbuf, int H, int W, int M, int t0, int t1, int chunk, int max_chunks, float lm,
                          float ep, int epoch, void* workspace, size_t workspace_bytes, void* stream) {
  BaView v;
  int rc = ba_view(v, workspace, workspace_bytes, E, nbuf, H, W, t0, t1, M, 0);
  if (rc) return rc;
  int brow[33];
  if (max_chunks < 1 || max_chunks > 32) return fail(DROID_E_ARG, "ba_unpack_chunk: bad %s", "max_chunks");
  const int nc = overlap_bounds(v.n, max_chunks, brow);
  if (chunk < 0 || chunk >= nc || epoch <= 0) return fail(DROID_E_ARG, "ba_unpack_chunk: bad %s", "chunk / epoch");
  launch_unpack_cols(v, brow[chunk], brow[chunk + 1], (double)lm, (double)ep, epoch, (hipStream_t)stream);
  return check_hip("ba_unpack_chunk");
}

int droid_ba_solve_update_overlap(float* poses, float* disps, const float* intrinsics, const float* weights,
                                  const int64_t* ii, const int64_t* jj, int E, int nbuf, int H, int W, int M,
                                  int t0, int t1, int epoch, int motion_only, float* dx_out, float* dz_out,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  BaView v;
  int rc = ba_view(v, workspace, workspace_bytes, E, nbuf, H, W, t0, t1, M, motion_only);
  if (rc) return rc;
  if (!poses || !disps) return fail(DROID_E_ARG, "ba: null %s", "state pointer");
  if (!motion_only && (!intrinsics || (E > 0 && (!weights || !ii || !jj))))
    return fail(DROID_E_ARG, "ba: null %s", "intrinsics/weights/edges");
  if (E <= 0 || epoch <= 0) return fail(DROID_E_ARG, "ba_solve_update_overlap: needs %s", "edges (the build presets the solver scratch) and epoch > 0");
  hipStream_t s = (hipStream_t)stream;
  if (!launch_chol_factor_overlap(v.sys, v.n, v.ld, v.hdr + HDR_CHOL_FAIL, v.bs_flags, v.ldiag, v.ov_ready, epoch, s))
    return fail(DROID_E_ARG, "ba_solve_update_overlap: %s", "the single-launch solver is not available for this system (unpack and call droid_ba_solve_update)");
  launch_chol_backsolve(v.sys, v.n, v.ld, v.xsol, v.bs_flags, v.ldiag, v.hdr + HDR_CHOL_FAIL, s, true);
  launch_update(v, poses, disps, intrinsics, weights, ii, jj, v.xsol, dx_out, dz_out, motion_only != 0, s,
                mirror_of(workspace));
  return check_hip("ba_solve_update_overlap");
}

int droid_ba_solve_update(float* poses, float* disps, const float* intrinsics, const float* weights,
                          const int64_t* ii, const int64_t* jj, int E, int nbuf, int H, int W, int M,
                          int t0, int t1, float lm, float ep, int motion_only, float* dx_out,
                          float* dz_out, void* workspace, size_t workspace_bytes, void* stream) {
  BaView v;
  int rc = ba_view(v, workspace, workspace_bytes, E, nbuf, H, W, t0, t1, M, motion_only);
  if (rc) return rc;
  if (!poses || !disps) return fail(DROID_E_ARG, "ba: null %s", "state pointer");
  if (!motion_only && (!intrinsics || (E > 0 && (!weights || !ii || !jj))))
    return fail(DROID_E_ARG, "ba: null %s", "intrinsics/weights/edges");
  hipStream_t s = (hipStream_t)stream;
  // with edges, droid_ba_build's assemble kernel has preset the solver scratch and cleared the failure flag
  const bool preset = E > 0;
  if (!preset) (void)hipMemsetAsync(v.hdr + HDR_CHOL_FAIL, 0, sizeof(int), s);
  launch_chol_solve(v.sys, v.n, v.ld, (double)lm, (double)ep, v.xsol, v.hdr + HDR_CHOL_FAIL, v.bs_flags, v.ldiag, s,
                    preset);
  launch_update(v, poses, disps, intrinsics, weights, ii, jj, v.xsol, dx_out, dz_out, motion_only != 0, s,
                mirror_of(workspace));
  return check_hip("ba_solve_update");
}

int droid_ba(float* poses, float* disps, const float* intrinsics, const float* disps_sens,
             const float* targets, const float* weights, const float* eta, const int64_t* ii,
             const int64_t* jj, int E, int nbuf, int H, int W, int M, int t0, int t1,
             int iterations, float lm, float ep, int motion_only, float* dx_out, float* dz_out,
             void* workspace, size_t workspace_bytes, void* stream) {
  int rc = droid_ba_prepare(ii, jj, E, nbuf, H, W, M, t0, t1, 0, nbuf, motion_only, workspace,
                            workspace_bytes, stream);
  if (rc) return rc;
  for (int it = 0; it < iterations; it++) {
    rc = droid_ba_build(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, E,
                        nbuf, H, W, M, t0, t1, motion_only, workspace, workspace_bytes, stream);
    if (rc) return rc;
    rc = droid_ba_solve_update(poses, disps, intrinsics, weights, ii, jj, E, nbuf, H, W, M, t0, t1, lm, ep, motion_only,
                               dx_out, dz_out, workspace, workspace_bytes, stream);
    if (rc) return rc;
  }
  return DROID_OK;
}

// Measurement support: one Gauss-Newton iteration (after droid_ba_prepare) with a HIP event
// between kernel groups on `stream`; blocks until done.  stage_ms[8] = {memset+linearise,
// assemble, fused E-rows + Schur SYRK + rhs, (unused), damp+factor, back-substitution solve, dx/disps/poses update, total}.
int droid_ba_profile_iteration(float* poses, float* disps, const float* intrinsics,
                               const float* disps_sens, const float* targets, const float* weights,
                               const float* eta, const int64_t* ii, const int64_t* jj, int E,
                               int nbuf, int H, int W, int M, int t0, int t1, float lm, float ep,
                               int motion_only, void* workspace, size_t workspace_bytes,
                               void* stream, float* stage_ms) {
  BaView v;
  int rc = ba_view(v, workspace, workspace_bytes, E, nbuf, H, W, t0, t1, M, motion_only);
  if (rc) return rc;
  hints_of(v, workspace, false);
  if (!stage_ms) return fail(DROID_E_ARG, "ba_profile: null %s", "stage_ms");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t ev[8];
  for (auto& e : ev) (void)hipEventCreate(&e);
  (void)hipEventRecord(ev[0], s);
  launch_build_stage(v, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, motion_only != 0, 0, s);
  (void)hipEventRecord(ev[1], s);
  launch_build_stage(v, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, motion_only != 0, 1, s);
  (void)hipEventRecord(ev[2], s);
  launch_build_stage(v, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, motion_only != 0, 2, s);
  (void)hipEventRecord(ev[3], s);
  launch_build_stage(v, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, motion_only != 0, 3, s);
  (void)hipEventRecord(ev[4], s);
  if (E <= 0) (void)hipMemsetAsync(v.hdr + HDR_CHOL_FAIL, 0, sizeof(int), s);
  if (v.n > 0 && E <= 0) {  // presets of x and the hand-off flags (with edges the assemble kernel did it)
    (void)hipMemsetAsync(v.xsol, 0xFF, sizeof(double) * (size_t)v.n, s);
    (void)hipMemsetAsync(v.bs_flags, 0xFF, sizeof(int) * chol_flag_words(v.n), s);
    (void)hipMemsetAsync(v.ldiag + chol_lfin_offset(v.n), 0xFF, sizeof(double) * chol_tiles(v.n) * CHOL_NB * CHOL_NB, s);
  }
  const bool single = launch_chol_factor(v.sys, v.n, v.ld, (double)lm, (double)ep, v.hdr + HDR_CHOL_FAIL, v.bs_flags, v.ldiag, s);
  (void)hipEventRecord(ev[5], s);
  launch_chol_backsolve(v.sys, v.n, v.ld, v.xsol, v.bs_flags, v.ldiag, v.hdr + HDR_CHOL_FAIL, s, single);
  (void)hipEventRecord(ev[6], s);
  launch_update(v, poses, disps, intrinsics, weights, ii, jj, v.xsol, nullptr, nullptr, motion_only != 0, s);
  (void)hipEventRecord(ev[7], s);
  hipError_t e = hipEventSynchronize(ev[7]);
  for (int k = 0; k < 7; k++) (void)hipEventElapsedTime(&stage_ms[k], ev[k], ev[k + 1]);
  (void)hipEventElapsedTime(&stage_ms[7], ev[0], ev[7]);
  for (auto& x : ev) (void)hipEventDestroy(x);
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "ba_profile: %s", hipGetErrorString(e));
    return DROID_E_HIP;
  }
  return check_hip("ba_profile_iteration");
}

double* droid_ba_system(void* workspace, int E, int nbuf, int H, int W, int t0, int t1, int M,
                        size_t* n_elements) {
  BaView v;
  if (!workspace || t1 <= t0) return nullptr;
  ba_carve(v, workspace, E, nbuf, H, W, t0, t1, M);
  if (n_elements) *n_elements = (size_t)(v.n + 1) * v.ld;
  return v.sys;
}

int droid_ba_status(const void* workspace, void* stream, int* status_out, int* depth_slots_out) {
  if (!workspace) return fail(DROID_E_ARG, "ba_status: null %s", "workspace");
  int hdr[HDR_WORDS];
  hipError_t e = hipMemcpyAsync(hdr, workspace, sizeof(hdr), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "ba_status: %s", hipGetErrorString(e));
    return DROID_E_HIP;
  }
  if (status_out) *status_out = hdr[HDR_STATUS];
  if (depth_slots_out) *depth_slots_out = hdr[HDR_M];
  return DROID_OK;
}

int droid_ba_attach_launch_hints(const void* workspace, int* hints) {
  if (!workspace) return fail(DROID_E_ARG, "ba_attach_launch_hints: null %s", "workspace");
  std::lock_guard<std::mutex> lock(g_mirror_mu);
  if (hints) g_hint[workspace] = HintState{hints, 0};
  else g_hint.erase(workspace);
  return DROID_OK;
}

int droid_ba_attach_status_mirror(const void* workspace, int* mirror) {
  if (!workspace) return fail(DROID_E_ARG, "ba_attach_status_mirror: null %s", "workspace");
  std::lock_guard<std::mutex> lock(g_mirror_mu);
  if (mirror) g_mirror[workspace] = mirror;
  else g_mirror.erase(workspace);
  return DROID_OK;
}

size_t droid_chol_scratch_doubles(int n) {
  if (n <= 0) return 0;
  return (size_t)(n + 1) * chol_ld(n) + chol_ldiag_doubles(n) + (chol_flag_words(n) + 1) / 2 + 16;
}

int droid_chol_solve(const double* A, const double* b, double* x, int n, double* scratch,
                     int* fail_flag, void* stream) {
  if (n <= 0 || !A || !b || !x || !scratch || !fail_flag) return fail(DROID_E_ARG, "chol_solve: bad %s", "argument");
  hipStream_t s = (hipStream_t)stream;
  const int ld = chol_ld(n);
  (void)hipMemsetAsync(fail_flag, 0, sizeof(int), s);
  launch_chol_pack(A, b, scratch, n, ld, s);
  double* ldiag = scratch + (size_t)(n + 1) * ld;   // tail of the scratch buffer: diagonal tiles, then flags
  int* flags = reinterpret_cast<int*>(ldiag + chol_ldiag_doubles(n));
  launch_chol_solve(scratch, n, ld, 0.0, 0.0, x, fail_flag, flags, ldiag, s);
  return check_hip("chol_solve");
}

// ---------------------------------------------------------------------------- geometry
int droid_frame_distance(const float* poses, const float* disps, const float* intrinsics,
                         const int64_t* ii, const int64_t* jj, int E, int nbuf, int H, int W,
                         float beta, float* dist, void* stream) {
  if (E < 0 || nbuf <= 0 || H <= 0 || W <= 0) return fail(DROID_E_ARG, "frame_distance: bad %s", "sizes");
  if (E == 0) return DROID_OK;
  if (!poses || !disps || !intrinsics || !ii || !jj || !dist) return fail(DROID_E_ARG, "frame_distance: null %s", "pointer");
  launch_frame_distance(poses, disps, intrinsics, ii, jj, E, nbuf, H, W, beta, dist, (hipStream_t)stream);
  return check_hip("frame_distance");
}

int droid_frame_distance_matrix(const float* poses, const float* disps, const float* intrinsics, int n, int nbuf,
                                int H, int W, float beta, float* dist, void* stream) {
  if (n < 0 || nbuf <= 0 || n > nbuf || H <= 0 || W <= 0) return fail(DROID_E_ARG, "frame_distance_matrix: bad %s", "sizes");
  if (n == 0) return DROID_OK;
  if (!poses || !disps || !intrinsics || !dist) return fail(DROID_E_ARG, "frame_distance_matrix: null %s", "pointer");
  launch_frame_distance_matrix(poses, disps, intrinsics, n, H, W, beta, dist, (hipStream_t)stream);
  return check_hip("frame_distance_matrix");
}

int droid_projmap(const float* poses, const float* disps, const float* intrinsics,
                  const int64_t* ii, const int64_t* jj, int E, int nbuf, int H, int W,
                  float* coords, float* valid, void* stream) {
  if (E < 0 || nbuf <= 0 || H <= 0 || W <= 0) return fail(DROID_E_ARG, "projmap: bad %s", "sizes");
  if (E == 0) return DROID_OK;
  if (!poses || !disps || !intrinsics || !ii || !jj || !coords || !valid) return fail(DROID_E_ARG, "projmap: null %s", "pointer");
  launch_projmap(poses, disps, intrinsics, ii, jj, E, nbuf, H, W, coords, valid, (hipStream_t)stream);
  return check_hip("projmap");
}

int droid_reproject_motion(const float* poses, const float* disps, const float* intrinsics, int intr_stride,
                           const int64_t* ii, const int64_t* jj, const float* target, int E, int nbuf, int H, int W,
                           float* coords, float* valid, float* motn, void* stream) {
  if (E < 0 || nbuf <= 0 || H <= 0 || W <= 0 || (intr_stride != 0 && intr_stride != 4))
    return fail(DROID_E_ARG, "reproject_motion: bad %s", "sizes");
  if (E == 0) return DROID_OK;
  if (!poses || !disps || !intrinsics || !ii || !jj || !coords || !valid) return fail(DROID_E_ARG, "reproject_motion: null %s", "pointer");
  if (motn && !target) return fail(DROID_E_ARG, "reproject_motion: motn needs %s", "target");
  launch_reproject_motion(poses, disps, intrinsics, intr_stride, ii, jj, target, E, nbuf, H, W, coords, valid, motn,
                          (hipStream_t)stream);
  return check_hip("reproject_motion");
}

int droid_iproj(const float* poses, const float* disps, const float* intrinsics, int nm, int H,
                int W, float* points, void* stream) {
  if (nm < 0 || H <= 0 || W <= 0) return fail(DROID_E_ARG, "iproj: bad %s", "sizes");
  if (nm == 0) return DROID_OK;
  if (!poses || !disps || !intrinsics || !points) return fail(DROID_E_ARG, "iproj: null %s", "pointer");
  launch_iproj(poses, disps, intrinsics, nm, H, W, points, (hipStream_t)stream);
  return check_hip("iproj");
}

int droid_depth_filter(const float* poses, const float* disps, const float* intrinsics,
                       const int64_t* ix, const float* thresh, int num, int nbuf, int H, int W,
                       float* counter, void* stream) {
  if (num < 0 || nbuf <= 0 || H <= 0 || W <= 0) return fail(DROID_E_ARG, "depth_filter: bad %s", "sizes");
  if (num == 0) return DROID_OK;
  if (!poses || !disps || !intrinsics || !ix || !thresh || !counter) return fail(DROID_E_ARG, "depth_filter: null %s", "pointer");
  launch_depth_filter(poses, disps, intrinsics, ix, thresh, num, nbuf, H, W, counter, (hipStream_t)stream);
  return check_hip("depth_filter");
}

}  // extern "C"
