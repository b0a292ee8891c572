"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy fp64 restatement of the Python-side reprojection and
motion features of the update step.  **parity unpinned**: lietorch is not importable here, so this follows the
reference's formulas by reading, and is pinned only by closed-form cases (tests/test_oracle_geom.py).

  reproject            droid_slam/depth_video.py:150-158 -> droid_slam/geom/projective_ops.py:96-125
  iproj / actp / proj  droid_slam/geom/projective_ops.py:18-37, :75-93 (X1 = Gij * X0), :39-51
  motion features      droid_slam/factor_graph.py:203-205
"""
import numpy as np

MIN_DEPTH = 0.2  # projective_ops.py:6


def _quat_rot(q, X):
    """rotate X [...,3] by unit quaternion q = (x, y, z, w) [...,4]"""
    qv, w = q[..., :3], q[..., 3:4]
    uv = 2.0 * np.cross(qv, X)
    return X + w * uv + np.cross(qv, uv)


def _relative(poses, ii, jj):
    """Gij = Gj * Gi^-1 (projective_ops.py:102) as (t, q); stereo edges get the fixed baseline (:105)."""
    ti, qi = poses[ii, :3], poses[ii, 3:]
    tj, qj = poses[jj, :3], poses[jj, 3:]
    qi_inv = qi * np.array([-1.0, -1.0, -1.0, 1.0])
    # q = qj * qi^-1 (Hamilton product, (x, y, z, w) layout)
    a, b = qj, qi_inv
    q = np.stack([a[:, 3] * b[:, 0] + a[:, 0] * b[:, 3] + a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                  a[:, 3] * b[:, 1] - a[:, 0] * b[:, 2] + a[:, 1] * b[:, 3] + a[:, 2] * b[:, 0],
                  a[:, 3] * b[:, 2] + a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0] + a[:, 2] * b[:, 3],
                  a[:, 3] * b[:, 3] - a[:, 0] * b[:, 0] - a[:, 1] * b[:, 1] - a[:, 2] * b[:, 2]], axis=-1)
    t = tj - _quat_rot(q, ti)
    st = ii == jj
    t[st] = np.array([-0.1, 0.0, 0.0])
    q[st] = np.array([0.0, 0.0, 0.0, 1.0])
    return t, q


def reproject(poses, disps, intrinsics, ii, jj):
    """coords [E,H,W,2], valid [E,H,W,1] (float64).  intrinsics [nbuf,4] or [4]."""
    poses = np.asarray(poses, np.float64)
    disps = np.asarray(disps, np.float64)
    K = np.asarray(intrinsics, np.float64)
    if K.ndim == 1:
        K = np.broadcast_to(K, (disps.shape[0], 4))
    ii, jj = np.asarray(ii, np.int64), np.asarray(jj, np.int64)
    H, W = disps.shape[1:]
    y, x = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    Ki, Kj = K[ii][:, None, None, :], K[jj][:, None, None, :]
    X0 = np.stack([(x[None] - Ki[..., 2]) / Ki[..., 0], (y[None] - Ki[..., 3]) / Ki[..., 1],
                   np.ones((len(ii), H, W))], axis=-1)                       # :27-30 (pts[..., :3]; pts[..., 3] = disp)
    d0 = disps[ii]
    t, q = _relative(poses, ii, jj)
    X1 = _quat_rot(q[:, None, None, :], X0) + t[:, None, None, :] * d0[..., None]   # SE3 acting on a homogeneous point
    Z = X1[..., 2]
    Zc = np.where(Z < 0.5 * MIN_DEPTH, 1.0, Z)                               # :46
    d = 1.0 / Zc
    coords = np.stack([Kj[..., 0] * (X1[..., 0] * d) + Kj[..., 2], Kj[..., 1] * (X1[..., 1] * d) + Kj[..., 3]], axis=-1)
    valid = ((Z > MIN_DEPTH) & (X0[..., 2] > MIN_DEPTH)).astype(np.float64)[..., None]   # :113
    return coords, valid


def motion_features(poses, disps, intrinsics, ii, jj, target):
    """motn [E,4,H,W] = clamp(cat(coords1 - coords0, target - coords1), -64, 64) (factor_graph.py:203-205)."""
    coords, valid = reproject(poses, disps, intrinsics, ii, jj)
    H, W = coords.shape[1:3]
    y, x = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    coords0 = np.stack([x, y], axis=-1)
    motn = np.concatenate([coords - coords0[None], np.asarray(target, np.float64) - coords], axis=-1)
    return np.clip(motn.transpose(0, 3, 1, 2), -64.0, 64.0), coords, valid


def depth_filter(poses, disps, intrinsics, ix, thresh):
    """counter [num,H,W] of droid_kernels.cu:661-775 (depth_filter_kernel): for each selected frame ix and each of
    its six temporal neighbours jx = ix-1, ix-2, ix-3, ix+3, ix+4, ix+5 (:695 -- the kernel's own enumeration),
    every pixel is moved into jx with the relative pose WITHOUT the stereo special case, and the neighbour counts 1
    when the inverse of its transformed disparity is within `thresh` of the inverse disparity of one of the four
    pixels around its projection (:763-767; the comparison is carried out in double precision there, `1.0/dj`).
    Projections whose integer corner is outside [0,W-1) x [0,H-1) do not count (:748)."""
    poses = np.asarray(poses, np.float64)
    disps = np.asarray(disps, np.float64)
    K = np.asarray(intrinsics, np.float64)
    ix = np.asarray(ix, np.int64)
    thresh = np.asarray(thresh, np.float64)
    nbuf, H, W = disps.shape
    y, x = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    out = np.zeros((len(ix), H, W))
    for b, i in enumerate(ix):
        if i < 0 or i >= nbuf:
            continue
        for nb in range(6):
            j = i - nb - 1 if nb < 3 else i + nb
            if j < 0 or j >= nbuf:
                continue
            t, q = _relative(poses, np.array([i]), np.array([j]))   # i != j for the six offsets: no stereo rule
            X0 = np.stack([(x - K[2]) / K[0], (y - K[3]) / K[1], np.ones((H, W))], axis=-1)
            X1 = _quat_rot(q[0][None, None, :], X0) + t[0][None, None, :] * disps[i][..., None]
            uj = K[0] * (X1[..., 0] / X1[..., 2]) + K[2]
            vj = K[1] * (X1[..., 1] / X1[..., 2]) + K[3]
            dj = disps[i] / X1[..., 2]
            u0, v0 = np.floor(uj).astype(np.int64), np.floor(vj).astype(np.int64)
            ok = (u0 >= 0) & (v0 >= 0) & (u0 < W - 1) & (v0 < H - 1)
            u0c, v0c = np.clip(u0, 0, W - 2), np.clip(v0, 0, H - 2)
            hit = np.zeros((H, W), bool)
            with np.errstate(divide="ignore", invalid="ignore"):
                for dv, du in ((0, 0), (0, 1), (1, 0), (1, 1)):
                    hit |= np.abs(1.0 / dj - 1.0 / disps[j][v0c + dv, u0c + du]) < thresh[b]
            out[b] += (hit & ok)
    return out
