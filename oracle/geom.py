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
