"""droid_backends -- MI355X (gfx950) drop-in for the reference's `droid_backends` extension.

Same nine operators, argument order and return structure as the pybind module of
/root/reference/src/droid.cpp:237-250, so `droid_slam/depth_video.py`, `factor_graph.py` and
`modules/corr.py` import and call it unchanged on PyTorch-ROCm.  This file is the host-side
mirror of droid.cpp: contiguity checks + thin calls into the C ABI
(include/droid_backends_hip.h, libdroid_backends_hip.so) on PyTorch's current HIP stream.
PyTorch is plumbing only (device memory, streams); all arithmetic is in the HIP library.
"""
from __future__ import annotations

import ctypes
import os as _os

import torch

from . import _lib
from ._lib import DroidBackendError  # noqa: F401

__all__ = ["ba", "frame_distance", "projmap", "depth_filter", "iproj", "altcorr_forward",
           "altcorr_backward", "corr_index_forward", "corr_index_backward",
           "altcorr_pyramid_forward", "reproject", "motion_features", "frame_distance_matrix",
           "corr_pyramid_forward"]  # the last four are additions

_DT = {torch.float16: _lib.DROID_F16, torch.float32: _lib.DROID_F32, torch.float64: _lib.DROID_F64}
_workspaces = {}   # (device index, stream handle) -> _Workspace

# Contract violations only a kernel can see (edge index outside the buffer, eta rows != depth slots, a stalled
# solver grid) are written to a status word.  DROID_HIP_CHECK=1: read it back after every call (one sync) and
# raise.  Default: no sync -- the last kernel of a call also writes the word to page-locked host memory, and the
# NEXT `ba` call on the same (device, stream) raises if the previous one had reported a violation by then.
_SYNC_CHECK = _os.environ.get("DROID_HIP_CHECK", "0") == "1"


class _Workspace:
    """Grow-only scratch of the `ba` calls of one (device, stream) + the host mirror of its status word."""

    def __init__(self):
        self.buf = None
        # {status of the last iteration, depth slots, number of iterations that ended with a violation / a stalled
        # solve so far, OR of their status bits}: written by the device only (include/droid_backends_hip.h)
        # words 4, 5: launch hints {tag of the call, its slots of Schur class 3}, written by the call's first kernel and read by
        # the library -- never waited for -- when it enqueues an iteration (droid_ba_attach_launch_hints)
        self.mirror = torch.zeros(8, dtype=torch.int32).pin_memory()
        self.seen = 0      # error count already raised / shown to the caller

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes:
            lib = _lib.load()
            if self.buf is not None:
                lib.droid_ba_attach_status_mirror(self.buf.data_ptr(), None)
                lib.droid_ba_attach_launch_hints(self.buf.data_ptr(), None)
            self.buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
            _lib.check(lib.droid_ba_attach_status_mirror(self.buf.data_ptr(), self.mirror.data_ptr()), "ba (status mirror)")
            _lib.check(lib.droid_ba_attach_launch_hints(self.buf.data_ptr(), self.mirror.data_ptr() + 16), "ba (launch hints)")
        return self.buf


def _check_input(x, name):
    # CHECK_CONTIGUOUS, droid.cpp:84-85
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a HIP (cuda) tensor: droid_backends has no CPU path")


def _check_index(x, name):
    """The kernels read edge / frame indices as int64 (torch.long, like the reference's accessors)."""
    _check_input(x, name)
    if x.dtype != torch.int64:
        raise RuntimeError(f"{name} must be int64 (torch.long), got {x.dtype}")


def _check_f32(x, name):
    _check_input(x, name)
    if x.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {x.dtype}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _ws_key(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return (idx, torch.cuda.current_stream(idx).cuda_stream)


def _workspace_obj(device):
    key = _ws_key(device)
    ws = _workspaces.get(key)
    if ws is None:
        ws = _workspaces[key] = _Workspace()
    return ws


def ba_status(workspace=None):
    """(status, depth_slots) of the last `ba` on the current device and stream (blocking read)."""
    import ctypes
    lib = _lib.load()
    obj = None
    if workspace is None:
        dev = torch.cuda.current_device()
        obj = _workspaces.get((dev, torch.cuda.current_stream(dev).cuda_stream))
        workspace = obj.buf if obj is not None else None
    if workspace is None:
        return 0, 0
    st, m = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.droid_ba_status(workspace.data_ptr(), _stream(), ctypes.byref(st), ctypes.byref(m)), "ba_status")
    if obj is not None:
        obj.seen = int(obj.mirror[2])   # droid_ba_status synchronised: the caller has seen everything up to here
    return st.value, m.value


_STATUS_TEXT = {1: "edge index outside the pose buffer", 2: "eta rows != number of depth slots "
                "|unique(ii) U [t0,t1)|", 4: "Cholesky failed (dx = 0)",
                8: "the single-launch solver stalled: another spinning grid held the GPU (dx = 0); "
                   "set DROID_CHOL_COOPERATIVE=1 or DROID_CHOL_MULTI_LAUNCH=1 when several processes share the GPU"}


def _raise_on_status(st, m, when=""):
    # bit 4 (not positive definite => dx = 0) is the reference's silent behaviour (droid_kernels.cu:1207-1210)
    bad = [txt for bit, txt in _STATUS_TEXT.items() if (st & bit) and bit != 4]
    if bad:
        raise RuntimeError(f"droid_backends.ba{when}: " + "; ".join(bad) + f" (device counted {m} depth slots)")


def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations,
       lm, ep, motion_only):
    """Dense bundle adjustment, droid.cpp:88-117 -> ba_cuda droid_kernels.cu:1314-1434.

    poses [nbuf,7] and disps [nbuf,H,W] are updated in place; returns [dx, dz] like the
    reference (dz is an empty tensor when motion_only, where the reference returns an undefined
    one).  eta is not contiguity-checked, as in the reference (droid.cpp:105-112), but it is
    made contiguous here because the kernels index it directly.
    """
    lib = _lib.load()
    for x, n in ((targets, "targets"), (weights, "weights"), (poses, "poses"), (disps, "disps"),
                 (intrinsics, "intrinsics"), (disps_sens, "disps_sens"), (ii, "ii"), (jj, "jj")):
        _check_input(x, n)
    if ii.dtype != torch.int64 or jj.dtype != torch.int64:
        raise RuntimeError("ii and jj must be int64")
    for x, n in ((targets, "targets"), (weights, "weights"), (poses, "poses"), (disps, "disps"),
                 (intrinsics, "intrinsics"), (disps_sens, "disps_sens")):
        if x.dtype != torch.float32:
            raise RuntimeError(f"{n} must be float32")
    t0, t1, iterations = int(t0), int(t1), int(iterations)
    motion_only = bool(motion_only)
    nbuf, H, W = disps.shape
    E = int(ii.shape[0])
    P = t1 - t0
    dev = poses.device
    if motion_only:
        M = 0
        eta_c = None
    else:
        eta_c = eta.contiguous().to(torch.float32).view(-1, H * W)
        M = int(eta_c.shape[0])
    nbytes = lib.droid_ba_workspace_bytes(E, nbuf, H, W, t0, t1, M)
    if nbytes == 0:
        raise RuntimeError("droid_backends.ba: bad sizes / window")
    wso = _workspace_obj(dev)
    # Deferred error report, no sync: the device counts the iterations that ended with a violation or a stalled solve
    # in page-locked host memory and ORs their bits (sticky: a later call cannot overwrite them); raise once per count.
    nerr = int(wso.mirror[2])
    if nerr != wso.seen:
        wso.seen = nerr
        _raise_on_status(int(wso.mirror[3]) & ~4, int(wso.mirror[1]), " (an earlier call on this stream)")
    ws = wso.get(nbytes, dev)
    dx = torch.empty((max(P, 0), 6), dtype=torch.float32, device=dev)
    dz = torch.empty((M, H * W), dtype=torch.float32, device=dev)
    rc = lib.droid_ba(poses.data_ptr(), disps.data_ptr(), intrinsics.data_ptr(), disps_sens.data_ptr(),
                      _ptr(targets), _ptr(weights), _ptr(eta_c), _ptr(ii), _ptr(jj), E, nbuf, H, W, M,
                      t0, t1, iterations, float(lm), float(ep), int(motion_only), dx.data_ptr(),
                      dz.data_ptr() if M > 0 else None, ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "ba")
    if _SYNC_CHECK:
        st = ba_status(ws)
        wso.seen = int(wso.mirror[2])
        _raise_on_status(*st)
    return [dx, dz]


def frame_distance(poses, disps, intrinsics, ii, jj, beta):
    """droid.cpp:120-136 -> frame_distance_cuda droid_kernels.cu:1438-1460."""
    lib = _lib.load()
    for x, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics")):
        _check_f32(x, n)
    _check_index(ii, "ii")
    _check_index(jj, "jj")
    nbuf, H, W = disps.shape
    nbuf = min(int(nbuf), int(poses.shape[0]))
    E = int(ii.shape[0])
    dist = torch.zeros((E,), dtype=torch.float32, device=poses.device)
    _lib.check(lib.droid_frame_distance(poses.data_ptr(), disps.data_ptr(), intrinsics.data_ptr(),
                                        ii.data_ptr(), jj.data_ptr(), E, nbuf, H, W, float(beta),
                                        dist.data_ptr(), _stream()), "frame_distance")
    return dist


def frame_distance_matrix(poses, disps, intrinsics, n, beta, bidirectional=True):
    """`DepthVideo.distance(ii=None)` (droid_slam/depth_video.py:160-190) in one launch: the [n, n] matrix of frame
    distances between the first n frames, d[i, j] = .5 * (frame_distance(i -> j) + frame_distance(j -> i)) when
    `bidirectional` (the reference's default), else frame_distance(i -> j).  No meshgrid index tensors, one kernel
    instead of two, each depth map fetched once per 32 targets.  An addition."""
    lib = _lib.load()
    for x, nm in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics")):
        _check_f32(x, nm)
    nbuf, H, W = disps.shape
    nbuf = min(int(nbuf), int(poses.shape[0]))
    n = int(n)
    d = torch.empty((n, n), dtype=torch.float32, device=poses.device)
    _lib.check(lib.droid_frame_distance_matrix(poses.data_ptr(), disps.data_ptr(), intrinsics.data_ptr(), n, nbuf,
                                               int(H), int(W), float(beta), d.data_ptr(), _stream()),
               "frame_distance_matrix")
    return .5 * (d + d.t()) if bidirectional else d


def projmap(poses, disps, intrinsics, ii, jj):
    """droid.cpp:139-154 -> projmap_cuda droid_kernels.cu:1463-1488."""
    lib = _lib.load()
    for x, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics")):
        _check_f32(x, n)
    _check_index(ii, "ii")
    _check_index(jj, "jj")
    nbuf, H, W = disps.shape
    nbuf = min(int(nbuf), int(poses.shape[0]))
    E = int(ii.shape[0])
    coords = torch.empty((E, H, W, 3), dtype=torch.float32, device=poses.device)
    valid = torch.empty((E, H, W, 1), dtype=torch.float32, device=poses.device)
    _lib.check(lib.droid_projmap(poses.data_ptr(), disps.data_ptr(), intrinsics.data_ptr(), ii.data_ptr(),
                                 jj.data_ptr(), E, nbuf, H, W, coords.data_ptr(), valid.data_ptr(),
                                 _stream()), "projmap")
    return [coords, valid]


def reproject(poses, disps, intrinsics, ii, jj, target=None):
    """`DepthVideo.reproject` (droid_slam/depth_video.py:150-158 -> geom/projective_ops.py:96-125) without
    lietorch: coords [1,E,H,W,2], valid [1,E,H,W,1] for the edges ii -> jj; stereo edges (ii == jj) use the fixed
    baseline.  `intrinsics` is [nbuf,4] (per frame, as `video.intrinsics`) or [4].  With `target` [1,E,H,W,2]
    (or [E,H,W,2]) the motion features of factor_graph.py:203-205 are produced in the same pass and returned as
    a third tensor [1,E,4,H,W]."""
    lib = _lib.load()
    for x, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics"), (ii, "ii"), (jj, "jj")):
        _check_input(x, n)
    if ii.dtype != torch.int64 or jj.dtype != torch.int64:
        raise RuntimeError("ii and jj must be int64")
    for x, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics")) + (((target, "target"),) if target is not None else ()):
        if x.dtype != torch.float32:
            raise RuntimeError(f"{n} must be float32")
    if poses.dim() == 3:  # accept the batched [1,nbuf,...] tensors the caller holds
        poses, disps = poses[0], disps[0]
        if intrinsics.dim() == 3:
            intrinsics = intrinsics[0]
    poses, disps, intrinsics = poses.contiguous(), disps.contiguous(), intrinsics.contiguous()
    ii, jj = ii.reshape(-1).contiguous(), jj.reshape(-1).contiguous()
    nbuf, H, W = disps.shape
    nbuf = min(int(nbuf), int(poses.shape[0]))
    stride = 4 if intrinsics.dim() == 2 else 0
    if stride == 4 and int(intrinsics.shape[0]) < nbuf:
        raise ValueError("reproject: intrinsics has fewer rows than frames")
    E = int(ii.shape[0])
    coords = torch.empty((1, E, H, W, 2), dtype=torch.float32, device=poses.device)
    valid = torch.empty((1, E, H, W, 1), dtype=torch.float32, device=poses.device)
    motn = None
    tptr = None
    if target is not None:
        _check_input(target, "target")
        target = target.reshape(E, H, W, 2).contiguous()
        motn = torch.empty((1, E, 4, H, W), dtype=torch.float32, device=poses.device)
        tptr = target.data_ptr()
    _lib.check(lib.droid_reproject_motion(poses.data_ptr(), disps.data_ptr(), intrinsics.data_ptr(), stride,
                                          ii.data_ptr(), jj.data_ptr(), tptr, E, nbuf, H, W, coords.data_ptr(),
                                          valid.data_ptr(), motn.data_ptr() if motn is not None else None,
                                          _stream()), "reproject")
    return (coords, valid) if motn is None else (coords, valid, motn)


def motion_features(poses, disps, intrinsics, ii, jj, target):
    """factor_graph.py:203-205 in one call: `coords1, mask = video.reproject(ii, jj);
    motn = cat([coords1 - coords0, target - coords1], -1).permute(0,1,4,2,3).clamp(-64, 64)` -> (motn, coords1, mask)."""
    coords, valid, motn = reproject(poses, disps, intrinsics, ii, jj, target)
    return motn, coords, valid


def depth_filter(poses, disps, intrinsics, ix, thresh):
    """droid.cpp:220-234 -> depth_filter_cuda droid_kernels.cu:1491-1515."""
    lib = _lib.load()
    for x, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics"), (thresh, "thresh")):
        _check_f32(x, n)
    _check_index(ix, "ix")
    nbuf, H, W = disps.shape
    nbuf = min(int(nbuf), int(poses.shape[0]))
    num = int(ix.shape[0])
    counter = torch.empty((num, H, W), dtype=torch.float32, device=disps.device)
    _lib.check(lib.droid_depth_filter(poses.data_ptr(), disps.data_ptr(), intrinsics.data_ptr(), ix.data_ptr(),
                                      thresh.data_ptr(), num, nbuf, H, W, counter.data_ptr(), _stream()),
               "depth_filter")
    return counter


def iproj(poses, disps, intrinsics):
    """droid.cpp:157-166 -> iproj_cuda droid_kernels.cu:1518-1541."""
    lib = _lib.load()
    for x, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics")):
        _check_f32(x, n)
    nm, H, W = disps.shape
    points = torch.empty((nm, H, W, 3), dtype=torch.float32, device=disps.device)
    _lib.check(lib.droid_iproj(poses.data_ptr(), disps.data_ptr(), intrinsics.data_ptr(), nm, H, W,
                               points.data_ptr(), _stream()), "iproj")
    return points


def _corr_dtype(t, name):
    if t.dtype not in _DT:
        raise RuntimeError(f"{name}: unsupported dtype {t.dtype} (float16/float32/float64)")
    return _DT[t.dtype]


def corr_index_forward(volume, coords, radius):
    """droid.cpp:170-178 -> corr_index_cuda_forward correlation_kernels.cu:126-155."""
    lib = _lib.load()
    _check_input(volume, "volume")
    _check_input(coords, "coords")
    if coords.dtype != torch.float32:
        raise RuntimeError("coords must be float32")
    B, H1, W1, H2, W2 = volume.shape
    r = int(radius)
    corr = torch.empty((B, 2 * r + 1, 2 * r + 1, H1, W1), dtype=volume.dtype, device=volume.device)
    _lib.check(lib.droid_corr_index_forward(volume.data_ptr(), coords.data_ptr(), corr.data_ptr(), B, H1, W1,
                                            H2, W2, r, _corr_dtype(volume, "volume"), _stream()),
               "corr_index_forward")
    return [corr]


def corr_pyramid_forward(pyramid, coords, radius):
    """CorrBlock.__call__ (droid_slam/modules/corr.py:40-50) without the torch.cat: pyramid = CorrBlock.corr_pyramid
    (list of [B,h,w,h>>l,w>>l] volumes of one dtype), coords [B,2,h,w] float32 at level-0 scale (the tensor __call__
    builds at :43-44).  Returns [corr] with corr [B, levels*(2r+1)^2, h, w] = torch.cat([corr_index_forward(
    pyramid[l], coords / 2**l, r).view(B, -1, h, w) for l], dim=1), bit for bit.  An addition."""
    lib = _lib.load()
    levels = list(pyramid)
    for i, p in enumerate(levels):
        _check_input(p, f"pyramid[{i}]")
        if p.dtype != levels[0].dtype:
            raise RuntimeError("corr_pyramid_forward: pyramid levels must share one dtype")
    _check_f32(coords, "coords")
    B, H1, W1 = int(levels[0].shape[0]), int(levels[0].shape[1]), int(levels[0].shape[2])
    for l, p in enumerate(levels):
        if tuple(p.shape) != (B, H1, W1, H1 >> l, W1 >> l):
            raise RuntimeError(f"corr_pyramid_forward: pyramid[{l}] must be [{B},{H1},{W1},{H1 >> l},{W1 >> l}]")
    if tuple(coords.shape) != (B, 2, H1, W1):
        raise RuntimeError("corr_pyramid_forward: coords must be [B,2,H1,W1]")
    r = int(radius)
    rd2 = (2 * r + 1) ** 2
    corr = torch.empty((B, len(levels) * rd2, H1, W1), dtype=levels[0].dtype, device=levels[0].device)
    ptrs = (ctypes.c_void_p * len(levels))(*[p.data_ptr() for p in levels])
    _lib.check(lib.droid_corr_pyramid_forward(ptrs, coords.data_ptr(), corr.data_ptr(), B, H1, W1, r, len(levels),
                                              _corr_dtype(levels[0], "pyramid"), _stream()), "corr_pyramid_forward")
    return [corr]


def corr_index_backward(volume, coords, corr_grad, radius):
    """droid.cpp:180-191 -> corr_index_cuda_backward correlation_kernels.cu:157-185."""
    lib = _lib.load()
    _check_input(volume, "volume")
    _check_f32(coords, "coords")
    _check_input(corr_grad, "corr_grad")
    B, H1, W1, H2, W2 = volume.shape
    if corr_grad.dtype != volume.dtype:
        corr_grad = corr_grad.to(volume.dtype)
    volume_grad = torch.empty_like(volume)
    _lib.check(lib.droid_corr_index_backward(coords.data_ptr(), corr_grad.data_ptr(), volume_grad.data_ptr(), B,
                                             H1, W1, H2, W2, int(radius), _corr_dtype(volume, "volume"),
                                             _stream()), "corr_index_backward")
    return [volume_grad]


def altcorr_forward(fmap1, fmap2, coords, radius):
    """droid.cpp:193-203 -> altcorr_cuda_forward altcorr_kernel.cu:290-319."""
    lib = _lib.load()
    _check_input(fmap1, "fmap1")
    _check_input(fmap2, "fmap2")
    _check_input(coords, "coords")
    if fmap2.dtype != fmap1.dtype or coords.dtype != torch.float32:
        raise RuntimeError("altcorr_forward: fmap dtypes must match and coords must be float32")
    B, N, H, W, _ = coords.shape
    _, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    r = int(radius)
    rd = 2 * r + 1
    corr = torch.empty((B, N, rd * rd, H, W), dtype=fmap1.dtype, device=fmap1.device)
    _lib.check(lib.droid_altcorr_forward(fmap1.data_ptr(), fmap2.data_ptr(), coords.data_ptr(), corr.data_ptr(),
                                         B, N, H1, W1, H2, W2, C, r, _corr_dtype(fmap1, "fmap1"), _stream()),
               "altcorr_forward")
    return [corr]


def altcorr_pyramid_forward(pyramid, coords, ii, jj, radius):
    """AltCorrBlock.corr_fn (droid_slam/modules/corr.py:105-125) in one launch, without the per-edge
    copies `pyramid[i][:, jj]`: pyramid = AltCorrBlock.pyramid (list of [1, frames, H>>l, W>>l, C]
    or [frames, ...] tensors), coords [E, H, W, 2] (or [1, E, H, W, 2]) float32 at level-0
    scale, ii / jj [E] int64.  Returns [corr] with corr [E, levels*(2r+1)^2, H, W] float32 =
    torch.cat([altcorr_forward(pyramid[0][ii].float(), pyramid[l][jj].float(), coords / 2**l, r) for l], dim=1).
    The pyramid may be float32 or -- what the SLAM path holds, `video.fmaps` is half (depth_video.py:44,
    factor_graph.py:260-261, modules/corr.py:97-104) -- float16: the half pyramid goes to the f16 matrix cores
    as it is (no `.float()` copies; the result is the fp32 evaluation of the widened maps up to summation order).
    Not one of the reference's nine operators."""
    lib = _lib.load()
    levels = [p[0] if p.dim() == 5 else p for p in pyramid]
    if coords.dim() == 5:
        coords = coords[0]
    for i, p in enumerate(levels):
        _check_input(p, f"pyramid[{i}]")
        if p.dtype != levels[0].dtype or p.dtype not in (torch.float32, torch.float16):
            raise RuntimeError("altcorr_pyramid_forward: pyramid levels must all be float32 or all float16")
    _check_input(coords, "coords")
    _check_input(ii, "ii")
    _check_input(jj, "jj")
    if coords.dtype != torch.float32 or ii.dtype != torch.int64 or jj.dtype != torch.int64:
        raise RuntimeError("altcorr_pyramid_forward: coords must be float32 and ii/jj int64")
    frames, H, W, C = levels[0].shape
    for l, p in enumerate(levels):
        if tuple(p.shape) != (frames, H >> l, W >> l, C):
            raise RuntimeError(f"altcorr_pyramid_forward: pyramid[{l}] must be [{frames},{H >> l},{W >> l},{C}]")
    E = int(ii.shape[0])
    if tuple(coords.shape) != (E, H, W, 2) or int(jj.shape[0]) != E:
        raise RuntimeError("altcorr_pyramid_forward: coords must be [E,H,W,2] and ii, jj [E]")
    r = int(radius)
    rd = 2 * r + 1
    corr = torch.empty((E, len(levels) * rd * rd, H, W), dtype=torch.float32, device=coords.device)
    ptrs = (ctypes.c_void_p * len(levels))(*[p.data_ptr() for p in levels])
    fn = lib.droid_altcorr_pyramid_forward_f16 if levels[0].dtype == torch.float16 else lib.droid_altcorr_pyramid_forward
    _lib.check(fn(ptrs, ii.data_ptr(), jj.data_ptr(), coords.data_ptr(), corr.data_ptr(), E, int(frames), int(H),
                  int(W), int(C), r, len(levels), _stream()), "altcorr_pyramid_forward")
    return [corr]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    """droid.cpp:205-217 -> altcorr_cuda_backward altcorr_kernel.cu:322-355 (fp32 only, like the reference)."""
    lib = _lib.load()
    for x, n in ((fmap1, "fmap1"), (fmap2, "fmap2"), (coords, "coords"), (corr_grad, "corr_grad")):
        _check_input(x, n)
        if x.dtype != torch.float32:
            raise RuntimeError(f"altcorr_backward: {n} must be float32")
    B, N, H, W, _ = coords.shape
    _, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    fmap1_grad = torch.zeros_like(fmap1)
    fmap2_grad = torch.zeros_like(fmap2)
    coords_grad = torch.zeros_like(coords)
    _lib.check(lib.droid_altcorr_backward(fmap1.data_ptr(), fmap2.data_ptr(), coords.data_ptr(),
                                          corr_grad.data_ptr(), fmap1_grad.data_ptr(), fmap2_grad.data_ptr(),
                                          B, N, H1, W1, H2, W2, C, int(radius), _stream()), "altcorr_backward")
    return [fmap1_grad, fmap2_grad, coords_grad]
