/*
 * droid_backends_hip.h -- C ABI of the MI355X (gfx950) implementation of DROID-SLAM's
 * correlation-lookup + dense bundle-adjustment hot path.
 *
 * This is the drop-in boundary: one entry point per operator the reference exports from its
 * `droid_backends` pybind module (/root/reference/src/droid.cpp:237-250).  Signatures carry only
 * raw DEVICE pointers, sizes and a HIP stream (passed as void*, a hipStream_t); no torch types.
 * Every function enqueues on `stream` and returns without synchronising unless stated.
 *
 * Return value: 0 on success, a negative DROID_E_* code on a contract violation detected on the
 * host.  Violations only a kernel can see (index out of range, eta rows != depth slots) are
 * written to the status word in the BA workspace (see droid_ba_status).
 *
 * Tensor layouts are exactly the reference's (row-major, contiguous):
 *   poses  [nbuf,7]  f32  (tx ty tz qx qy qz qw), world->camera     depth_video.py:33
 *   disps  [nbuf,H,W] f32 ; disps_sens [nbuf,H,W] f32 ; intrinsics [4] f32 (fx fy cx cy)
 *   targets, weights [E,2,H,W] f32 ; eta [M,H,W] f32 ; ii, jj [E] int64
 */
#ifndef DROID_BACKENDS_HIP_H
#define DROID_BACKENDS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DROID_ABI_VERSION 1

#define DROID_OK 0
#define DROID_E_ARG (-1)       /* bad size / null pointer / unsupported dtype            */
#define DROID_E_WORKSPACE (-2) /* workspace too small                                    */
#define DROID_E_HIP (-3)       /* a HIP runtime call failed (see droid_last_error)       */

/* element types of the correlation operators (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * correlation_kernels.cu:145, altcorr_kernel.cu:308) */
#define DROID_F16 0
#define DROID_F32 1
#define DROID_F64 2

int droid_abi_version(void);
const char *droid_last_error(void); /* thread-local message of the last failing call */

/* ------------------------------------------------------------------ correlation lookups */

/* corr_index_forward (droid.cpp:170-178 -> correlation_kernels.cu:126-155).
 * volume [B,H1,W1,H2,W2] dtype ; coords [B,2,H1,W1] f32 ; corr [B,2r+1,2r+1,H1,W1] dtype
 * (written completely; the caller need not zero it). */
int droid_corr_index_forward(const void *volume, const float *coords, void *corr, int B, int H1,
                             int W1, int H2, int W2, int radius, int dtype, void *stream);

/* CorrBlock.__call__ over all pyramid levels (droid_slam/modules/corr.py:40-50): level l looks
 * up volumes[l] ([B,H1,W1,H1>>l,W1>>l], dtype) at coords * 2^-l (exact) and writes channels
 * [l (2r+1)^2, (l+1) (2r+1)^2) of corr [B, levels (2r+1)^2, H1, W1] = torch.cat(out_pyramid, dim=2) of the
 * reference (with its leading dimension of 1 dropped), without the cat and the per-level coordinate tensors.
 * volumes: HOST array of `levels` device pointers; coords [B,2,H1,W1] f32 at level-0 scale.  Values are
 * bit-identical to droid_corr_index_forward per level.  radius 3 or 4.  Not one of the reference's nine operators. */
int droid_corr_pyramid_forward(const void *const *volumes, const float *coords, void *corr, int B, int H1,
                               int W1, int radius, int levels, int dtype, void *stream);

/* corr_index_backward (droid.cpp:180-191 -> correlation_kernels.cu:157-185).
 * corr_grad [B,2r+1,2r+1,H1,W1] dtype ; volume_grad [B,H1,W1,H2,W2] dtype (written completely). */
int droid_corr_index_backward(const float *coords, const void *corr_grad, void *volume_grad, int B,
                              int H1, int W1, int H2, int W2, int radius, int dtype, void *stream);

/* altcorr_forward (droid.cpp:193-203 -> altcorr_kernel.cu:290-319).
 * fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C] dtype ; coords [B,N,H1,W1,2] f32 ;
 * corr [B,N,(2r+1)^2,H1,W1] dtype, channel = ix*(2r+1)+iy (written completely). */
int droid_altcorr_forward(const void *fmap1, const void *fmap2, const float *coords, void *corr,
                          int B, int N, int H1, int W1, int H2, int W2, int C, int radius,
                          int dtype, void *stream);

/* AltCorrBlock.corr_fn fused over the pyramid levels (droid_slam/modules/corr.py:105-125; SURVEY.md
 * section 8f row 2): for edge e, level l: altcorr_forward(pyramid[0][ii[e]], pyramid[l][jj[e]],
 * coords[e] / 2^l, r), all levels in one launch and without the per-edge feature-map copies
 * (`pyramid[i][:, jj]`) the Python code makes.
 * pyramid: HOST array of `levels` device pointers, level l = [frames, H>>l, W>>l, C] f32 channels
 * last (AltCorrBlock.pyramid[l] with its leading batch dimension of 1 dropped);
 * ii, jj [E] int64 ; coords [E,H,W,2] f32 (one coordinate set) ;
 * corr [E, levels*(2r+1)^2, H, W] f32 = torch.cat of the per-level results (written completely).
 * fp32, C % 16 == 0, C <= 128, r in {3,4}, levels <= 4; anything else returns DROID_E_ARG and the
 * caller keeps using droid_altcorr_forward per level.  An edge index outside [0,frames) gives zeros. */
int droid_altcorr_pyramid_forward(const float *const *pyramid, const int64_t *ii, const int64_t *jj,
                                  const float *coords, float *corr, int E, int frames, int H, int W,
                                  int C, int radius, int levels, void *stream);

/* The same operator for the pyramid the SLAM path actually holds: `AltCorrBlock(self.video.fmaps[None])`
 * (factor_graph.py:260-261) is built from the HALF feature buffer (depth_video.py:44), `/ 4.0` and `avg_pool2d`
 * stay in half (modules/corr.py:97-104), and `corr_fn` widens the gathered per-edge copies with `.float()`
 * (:120) before the fp32 kernel.  This entry point takes that half pyramid as it is -- level l =
 * [frames, H>>l, W>>l, C] f16 channels last -- and returns the fp32 tensor the reference's
 * `altcorr_forward(fmap1.float(), fmap2.float(), ...)` calls return: products of two halves are exact in fp32 and
 * the dot products are accumulated in fp32 on v_mfma_f32_16x16x32_f16, so only the summation order over channels
 * differs (<= 1e-6 of the output scale; half subnormals are not flushed).  No `.float()` copies, half the feature
 * bytes, 16x the matrix rate of the fp32 path.  C % 32 == 0, C <= 128; otherwise as above. */
int droid_altcorr_pyramid_forward_f16(const void *const *pyramid, const int64_t *ii, const int64_t *jj,
                                      const float *coords, float *corr, int E, int frames, int H, int W,
                                      int C, int radius, int levels, void *stream);

/* altcorr_backward (droid.cpp:205-217 -> altcorr_kernel.cu:322-355), fp32 only like the reference.
 * fmap1_grad/fmap2_grad must be zero-filled by the caller (atomic accumulation);
 * coords_grad is not written (the reference returns zeros). */
int droid_altcorr_backward(const float *fmap1, const float *fmap2, const float *coords,
                           const float *corr_grad, float *fmap1_grad, float *fmap2_grad, int B,
                           int N, int H1, int W1, int H2, int W2, int C, int radius, void *stream);

/* ------------------------------------------------------------------ bundle adjustment */

/* Bytes of device scratch one `ba` call needs (host-only arithmetic, no HIP call).
 * M = rows of eta = number of depth slots |unique(ii) U [t0,t1)| (0 when motion_only). */
size_t droid_ba_workspace_bytes(int E, int nbuf, int H, int W, int t0, int t1, int M);

/* ba (droid.cpp:88-117 -> droid_kernels.cu:1314-1434): `iterations` Gauss-Newton steps, poses
 * and disps updated in place.  dx_out [t1-t0,6] f32 and dz_out [M,H*W] f32 receive the last
 * iteration's updates (dz_out may be NULL when motion_only).  Equivalent to
 * droid_ba_prepare + iterations x (droid_ba_build, droid_ba_solve_update). */
int droid_ba(float *poses, float *disps, const float *intrinsics, const float *disps_sens,
             const float *targets, const float *weights, const float *eta, const int64_t *ii,
             const int64_t *jj, int E, int nbuf, int H, int W, int M, int t0, int t1,
             int iterations, float lm, float ep, int motion_only, float *dx_out, float *dz_out,
             void *workspace, size_t workspace_bytes, void *stream);

/* Phase API (what the multi-GPU driver calls; single-GPU `droid_ba` is built from it).
 *
 * own0/own1: the frames whose depth maps this rank owns.  Edges passed to a rank must all have
 * ii in [own0,own1); window frames outside it create no depth slot here.  Single GPU: 0, nbuf.
 *
 * droid_ba_prepare: once per call -- depth-slot table, CSR of edges by source frame.
 * droid_ba_build:   one linearisation: writes this rank's contribution to the reduced camera
 *                   system into the workspace: S = [ A - E C^-1 E^T ; b^T ] as a dense fp64
 *                   row-major matrix of 6P+1 rows with row pitch ld = roundup16(6P+1), lower
 *                   triangle valid, row 6P = rhs, no damping yet.  droid_ba_system() returns its device address so that the
 *                   caller can all-reduce (sum) it over ranks (RCCL) before the solve.
 * droid_ba_solve_update: damping (diag += ep + lm*diag), Cholesky, solve, depth
 *                   back-substitution for the owned slots (the E rows are recomputed from
 *                   weights / disparities / the not yet retracted poses, never stored), SE3 /
 *                   disparity retraction.
 */
int droid_ba_prepare(const int64_t *ii, const int64_t *jj, int E, int nbuf, int H, int W, int M,
                     int t0, int t1, int own0, int own1, int motion_only, void *workspace,
                     size_t workspace_bytes, void *stream);

int droid_ba_build(const float *poses, const float *disps, const float *intrinsics,
                   const float *disps_sens, const float *targets, const float *weights,
                   const float *eta, const int64_t *ii, const int64_t *jj, int E, int nbuf, int H,
                   int W, int M, int t0, int t1, int motion_only, void *workspace,
                   size_t workspace_bytes, void *stream);

/* Multi-GPU variant of the build phase: the rank's contribution goes to a PACKED copy of the system -- the lower
 * triangle + rhs row by block columns of 64: block column J holds rows 64 J .. 6P (row 6P = rhs) as rows of w_J
 * doubles (w_J = 64, the last one 6P - 64 J), block columns one after the other --
 * which droid_ba_packed_system() exposes as one contiguous fp64 tensor of *n_elements values: all-reduce (sum) it
 * over the ranks as it is (half the bytes of the pitched matrix, no gather / scatter of the triangle), then call
 * droid_ba_unpack_system (one launch: packed -> the pitched matrix the solver factors in place) and
 * droid_ba_solve_update.  Argument meaning as droid_ba_build. */
int droid_ba_build_packed(const float *poses, const float *disps, const float *intrinsics,
                          const float *disps_sens, const float *targets, const float *weights,
                          const float *eta, const int64_t *ii, const int64_t *jj, int E, int nbuf, int H,
                          int W, int M, int t0, int t1, int motion_only, void *workspace,
                          size_t workspace_bytes, void *stream);
double *droid_ba_packed_system(void *workspace, int E, int nbuf, int H, int W, int t0, int t1, int M,
                               size_t *n_elements);
int droid_ba_unpack_system(int E, int nbuf, int H, int W, int M, int t0, int t1, int motion_only,
                           void *workspace, size_t workspace_bytes, void *stream);

/* Overlap of the collective with the solve (multi-GPU, opt-in).  The factorisation consumes
 * block columns left to right and the packed system is block-column major, so a PREFIX of it is all the leading
 * block columns need.  droid_ba_overlap_plan cuts the packed tensor into at most max_chunks contiguous element
 * ranges [packed_offsets[c], packed_offsets[c+1]) of 1, 1, 2, 3, 4, ... block columns.  Per iteration:
 *   main stream : droid_ba_build_packed ... droid_ba_solve_update_overlap(epoch)   -- launched BEFORE the reduction;
 *                 its factorisation waits, tile by tile, for the block columns it is about to read
 *   side stream : (after the build) for c = 0 .. nchunks-1: all-reduce chunk c of the packed tensor in place, then
 *                 droid_ba_unpack_chunk(c, lm, ep, epoch): columns -> pitched matrix with the damping applied, then
 *                 the block columns are published for `epoch`.
 * epoch = 1, 2, ... within one droid_ba_prepare (which resets the published epochs).  The spinning grid leaves
 * DROID_OVERLAP_RESERVE_CUS (default 32) compute units free for the collective's kernels.  Returns DROID_E_ARG
 * when the single-launch solver cannot run this system (then: droid_ba_unpack_system + droid_ba_solve_update).
 * Rehearsed with two ranks on one GPU (gloo) and with eight in-process shards; NOT yet measured over RCCL. */
int droid_ba_overlap_plan(int t0, int t1, int max_chunks, int *nchunks_out, size_t *packed_offsets);
int droid_ba_unpack_chunk(int E, int nbuf, int H, int W, int M, int t0, int t1, int chunk, int max_chunks, float lm,
                          float ep, int epoch, void *workspace, size_t workspace_bytes, void *stream);
int droid_ba_solve_update_overlap(float *poses, float *disps, const float *intrinsics, const float *weights,
                                  const int64_t *ii, const int64_t *jj, int E, int nbuf, int H, int W, int M,
                                  int t0, int t1, int epoch, int motion_only, float *dx_out, float *dz_out,
                                  void *workspace, size_t workspace_bytes, void *stream);

int droid_ba_solve_update(float *poses, float *disps, const float *intrinsics, const float *weights,
                          const int64_t *ii, const int64_t *jj, int E, int nbuf, int H, int W, int M,
                          int t0, int t1, float lm, float ep, int motion_only, float *dx_out,
                          float *dz_out, void *workspace, size_t workspace_bytes, void *stream);

/* Measurement support (bench.py): one Gauss-Newton iteration after droid_ba_prepare with a HIP
 * event between kernel groups on `stream`; synchronises.  stage_ms[8] = {memset+linearise,
 * assemble, fused E-rows + Schur SYRK + rhs, (unused: 0), damp+Cholesky factor, triangular
 * back-solve, state update, total}. */
int droid_ba_profile_iteration(float *poses, float *disps, const float *intrinsics,
                               const float *disps_sens, const float *targets, const float *weights,
                               const float *eta, const int64_t *ii, const int64_t *jj, int E,
                               int nbuf, int H, int W, int M, int t0, int t1, float lm, float ep,
                               int motion_only, void *workspace, size_t workspace_bytes,
                               void *stream, float *stage_ms);

/* Device address / element count of the dense fp64 system inside the workspace. */
double *droid_ba_system(void *workspace, int E, int nbuf, int H, int W, int t0, int t1, int M,
                        size_t *n_elements);

/* Blocking read of the workspace status word: 0 ok; bit0 index out of range; bit1 eta rows !=
 * depth slots; bit2 Cholesky failed in the last solve (not positive definite: dx = 0, like
 * droid_kernels.cu:1207-1210 -- the reference's behaviour, not an error); bit3 the single-launch solver's
 * grid stalled (another spinning grid of a different PROCESS held the CUs; dx = 0; an error: run such
 * deployments with DROID_CHOL_COOPERATIVE=1 or DROID_CHOL_MULTI_LAUNCH=1). */
int droid_ba_status(const void *workspace, void *stream, int *status_out, int *depth_slots_out);

/* Non-blocking error reporting: `mirror` points to 4 ZEROED ints of page-locked host memory that the device can
 * address (hipHostMalloc / torch pin_memory).  Every droid_ba_solve_update on this workspace then ends by
 * writing {status word, depth slots} to words 0 / 1 (system-scope stores by the last kernel of the iteration), and,
 * when the iteration ended with bit0, bit1 or bit3 set, by incrementing word 2 and OR-ing the status into word 3.
 * Words 2 / 3 are STICKY -- only the device writes them, nothing resets them -- so a violation of call k is still
 * there after call k+1 has been enqueued and has reset the workspace's own status word: the host compares word 2
 * with the count it has already reported, whenever it likes, without synchronising the stream.
 * mirror = NULL detaches.  The registration is host-side (keyed by the workspace address).
 *
 * Preconditions of the phase API on one workspace: droid_ba_build and droid_ba_solve_update alternate on ONE
 * stream (build presets the solver scratch that solve_update consumes); concurrent `ba` calls need separate
 * workspaces (droid_backends keeps one per (device, stream)). */
int droid_ba_attach_status_mirror(const void *workspace, int *mirror);

/* Launch hints (optional): `hints` points to 2 ZEROED ints of page-locked host memory that the device can address.
 * droid_ba_prepare's kernel then writes {tag of that prepare, number of depth slots of Schur class 3 (more edges than the
 * regular Schur kernels take: block pairs)} there, and droid_ba_build / droid_ba leave the block-pair launch of an
 * iteration out when the hint of the CURRENT prepare has arrived and says "none" (most graphs: 5 us per iteration).
 * The host never waits for the hint: not there yet = launch.  Results do not depend on it.  hints = NULL detaches. */
int droid_ba_attach_launch_hints(const void *workspace, int *hints);

/* Dense SPD solve used by the BA (exposed for tests): A [n,n] fp64 row-major (lower triangle
 * read, destroyed), b [n] fp64 -> x [n] fp64.  fail_flag (device int) is set to 1 when a pivot
 * is not positive.  scratch (128-byte aligned): >= droid_chol_scratch_doubles(n) doubles (the augmented system
 * with 128-byte rows, the factored diagonal tiles, hand-over slots of the panel tiles, hand-off flags). */
size_t droid_chol_scratch_doubles(int n);
int droid_chol_solve(const double *A, const double *b, double *x, int n, double *scratch,
                     int *fail_flag, void *stream);

/* ------------------------------------------------------------------ geometry operators */

/* frame_distance (droid.cpp:120-136 -> droid_kernels.cu:518-657, :1438-1460): dist [E] f32 */
int droid_frame_distance(const float *poses, const float *disps, const float *intrinsics,
                         const int64_t *ii, const int64_t *jj, int E, int nbuf, int H, int W,
                         float beta, float *dist, void *stream);

/* All ordered pairs of the first n frames in one launch: dist [n,n] f32, dist[i*n + j] = the value
 * droid_frame_distance gives for the edge i -> j.  This is `DepthVideo.distance()` with ii = None
 * (droid_slam/depth_video.py:160-169: meshgrid indices) and the candidate matrix of add_proximity_factors
 * (factor_graph.py:318-326) without index tensors, and with each depth map fetched once per block of 32
 * targets instead of once per pair (SURVEY.md section 8f row 1 "batched all-pairs variant").  Not one of the
 * reference's nine operators.  poses / disps hold at least nbuf >= n frames. */
int droid_frame_distance_matrix(const float *poses, const float *disps, const float *intrinsics, int n,
                                int nbuf, int H, int W, float beta, float *dist, void *stream);

/* projmap (droid.cpp:139-154 -> droid_kernels.cu:427-516, :1463-1488):
 * coords [E,H,W,3] f32 (channel 2 zero), valid [E,H,W,1] f32 */
int droid_projmap(const float *poses, const float *disps, const float *intrinsics,
                  const int64_t *ii, const int64_t *jj, int E, int nbuf, int H, int W,
                  float *coords, float *valid, void *stream);

/* reproject + motion features: what `DepthVideo.reproject` and the three lines after
 * it compute for every update-operator call, in one pass and without lietorch --
 *   droid_slam/depth_video.py:150-158 -> geom/projective_ops.py:96-125 (`projective_transform`: source-frame
 *   intrinsics for the back-projection, target-frame intrinsics for the projection, stereo edges ii == jj use the
 *   baseline (-0.1,0,0), depth < 0.1 projects with depth 1, valid = depth > 0.2), factor_graph.py:203-205
 *   (`motn = cat(coords1 - coords0, target - coords1).permute(0,1,4,2,3).clamp(-64, 64)`).
 * intrinsics: [nbuf,4] with intr_stride = 4, or one [4] for all frames with intr_stride = 0.
 * target [E,H,W,2] f32 and motn [E,4,H,W] f32 may both be null (plain reproject).
 * coords [E,H,W,2] f32, valid [E,H,W,1] f32.  Edges with an index outside [0,nbuf) produce zeros. */
int droid_reproject_motion(const float *poses, const float *disps, const float *intrinsics, int intr_stride,
                           const int64_t *ii, const int64_t *jj, const float *target, int E, int nbuf, int H,
                           int W, float *coords, float *valid, float *motn, void *stream);

/* iproj (droid.cpp:157-166 -> droid_kernels.cu:779-850, :1518-1541): points [nm,H,W,3] f32 */
int droid_iproj(const float *poses, const float *disps, const float *intrinsics, int nm, int H,
                int W, float *points, void *stream);

/* depth_filter (droid.cpp:220-234 -> droid_kernels.cu:661-775, :1491-1515):
 * counter [num,H,W] f32 (written completely) */
int droid_depth_filter(const float *poses, const float *disps, const float *intrinsics,
                       const int64_t *ix, const float *thresh, int num, int nbuf, int H, int W,
                       float *counter, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DROID_BACKENDS_HIP_H */
