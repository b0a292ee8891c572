"""Deterministic synthetic factor graphs for tests and bench.py (recipe: SURVEY.md section 8d).

Pure numpy; shapes and conventions follow the reference's callers:
poses [N,7] = (tx ty tz qx qy qz qw), world->camera (droid_slam/depth_video.py:33);
disps [N,H,W]; intrinsics [fx fy cx cy]; targets/weights [E,2,H,W] (factor_graph.py:237-241);
eta [M,H,W] with M = |unique(ii) U [t0,t1)| (src/droid_kernels.cu:1336-1344, :1398).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# name -> (keyframes, edges, H, W, radius, stereo, lm, ep)   BASELINE.json configs 1..5
CONFIGS = {
    "cfg1": (8, 32, 48, 64, 3, False, 1e-4, 0.1),
    "cfg2": (64, 512, 48, 64, 3, False, 1e-4, 0.1),
    "cfg3": (256, 2000, 48, 64, 3, False, 1e-5, 1e-2),
    "cfg4": (256, 8000, 48, 64, 3, False, 1e-5, 1e-2),
    "cfg5": (128, 1024, 96, 128, 4, True, 1e-5, 1e-2),
}
CONFIG_SEEDS = {"cfg1": 0, "cfg2": 1, "cfg3": 2, "cfg4": 3, "cfg5": 4}


def quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz], -1)


def quat_rot(q, v):
    qv = q[..., :3]
    uv = 2.0 * np.cross(qv, v)
    return v + q[..., 3:4] * uv + np.cross(qv, uv)


def quat_inv(q):
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def so3_exp(phi):
    th = np.linalg.norm(phi, axis=-1, keepdims=True)
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    imag = np.where(small, 0.5 - th * th / 48.0, np.sin(0.5 * ths) / ths)
    return np.concatenate([imag * phi, np.cos(0.5 * th)], -1)


def se3_exp(xi):
    """xi = (tau, phi) -> (t, q); V-matrix form of the SE3 exponential."""
    tau, phi = xi[..., :3], xi[..., 3:]
    q = so3_exp(phi)
    th = np.linalg.norm(phi, axis=-1, keepdims=True)
    th2 = th * th
    small = th < 1e-6
    ths = np.where(small, 1.0, th)
    a = np.where(small, 0.5, (1 - np.cos(ths)) / (ths * ths))
    b = np.where(small, 1.0 / 6.0, (ths - np.sin(ths)) / (ths ** 3))
    c1 = np.cross(phi, tau)
    c2 = np.cross(phi, c1)
    del th2
    return tau + a * c1 + b * c2, q


def se3_mul(ta, qa, tb, qb):
    """(ta,qa) * (tb,qb): x -> Ra(Rb x + tb) + ta."""
    return quat_rot(qa, tb) + ta, quat_mul(qa, qb)


def se3_inv(t, q):
    qi = quat_inv(q)
    return -quat_rot(qi, t), qi


def rel_pose(poses, i, j):
    """Tij = Tj * Ti^-1 for world->camera poses (src/droid_kernels.cu:96-107)."""
    ti, qi = se3_inv(poses[i, :3], poses[i, 3:])
    return se3_mul(poses[j, :3], poses[j, 3:], ti, qi)


def reproject(poses, disps, intr, ii, jj):
    """GT pixel coordinates of frame-ii pixels in frame jj: [E,2,H,W] (x then y) and depth Z."""
    fx, fy, cx, cy = [float(v) for v in intr]
    N, H, W = disps.shape
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    X = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)  # [H,W,3]
    E = len(ii)
    out = np.zeros((E, 2, H, W))
    Z = np.zeros((E, H, W))
    for e in range(E):
        i, j = int(ii[e]), int(jj[e])
        if i == j:  # stereo pair, fixed baseline (src/droid_kernels.cu:219-229)
            t, q = np.array([-0.1, 0.0, 0.0]), np.array([0.0, 0.0, 0.0, 1.0])
        else:
            t, q = rel_pose(poses, i, j)
        Y = quat_rot(q[None, None], X) + disps[i][..., None] * t
        z = Y[..., 2]
        zs = np.where(np.abs(z) < 1e-6, 1e-6, z)
        out[e, 0] = fx * Y[..., 0] / zs + cx
        out[e, 1] = fy * Y[..., 1] / zs + cy
        Z[e] = z
    return out, Z


def make_edges(N, E, stereo, rng, t0=1):
    band = []
    for d in (1, 2, 3):
        for i in range(N):
            for j in (i - d, i + d):
                if 0 <= j < N:
                    band.append((d, i, j))
    band.sort()
    edges = [(i, j) for _, i, j in band]
    if stereo:
        edges = [(i, i) for i in range(N)] + edges
    if len(edges) > E:
        edges = edges[:E]
    have = set(edges)
    guard = 0
    while len(edges) < E:
        i, j = (int(x) for x in rng.integers(0, N, size=2))
        guard += 1
        if guard > 100 * E + 10000:
            raise RuntimeError("cannot place the requested number of long-range edges")
        if abs(i - j) < 4 or (i, j) in have:
            continue
        edges.append((i, j))
        have.add((i, j))
        if len(edges) < E and (j, i) not in have:
            edges.append((j, i))
            have.add((j, i))
    ii = np.array([e[0] for e in edges], dtype=np.int64)
    jj = np.array([e[1] for e in edges], dtype=np.int64)
    missing = sorted(set(range(t0, N)) - set(ii.tolist()))
    assert not missing, f"window frames without outgoing edges: {missing}"
    return ii, jj


@dataclass
class BAProblem:
    poses: np.ndarray          # [nbuf,7] f32 initial (perturbed) state
    disps: np.ndarray          # [nbuf,H,W] f32
    intrinsics: np.ndarray     # [4] f32
    disps_sens: np.ndarray     # [nbuf,H,W] f32
    targets: np.ndarray        # [E,2,H,W] f32
    weights: np.ndarray        # [E,2,H,W] f32
    eta: np.ndarray            # [M,H,W] f32
    ii: np.ndarray             # [E] int64
    jj: np.ndarray             # [E] int64
    t0: int
    t1: int
    lm: float
    ep: float
    radius: int = 3
    gt_poses: np.ndarray = field(default=None, repr=False)
    gt_disps: np.ndarray = field(default=None, repr=False)
    gt_coords: np.ndarray = field(default=None, repr=False)  # [E,2,H,W] noise-free reprojection

    @property
    def n_depth_slots(self):
        return int(self.eta.shape[0])


def box5(x):
    H, W = x.shape[-2:]
    p = np.pad(x, [(0, 0)] * (x.ndim - 2) + [(2, 2), (2, 2)], mode="edge")
    out = np.zeros_like(x)
    for a in range(5):
        for b in range(5):
            out += p[..., a:a + H, b:b + W]
    return out / 25.0


def make_ba_problem(N=8, E=32, H=48, W=64, stereo=False, lm=1e-4, ep=0.1, seed=0, rgbd=False,
                    nbuf=None, t0=1, radius=3, edges=None):
    """SURVEY.md section 8d generator.  Returns float32 arrays shaped like the reference's callers pass.
    edges = (ii, jj): an explicit edge list instead of the band + long-range graph of make_edges (E is ignored)."""
    rng = np.random.default_rng(seed)
    nbuf = N if nbuf is None else nbuf
    intr = np.array([W / 2.0, W / 2.0, W / 2.0, H / 2.0])
    # GT trajectory: smooth random walk, frame 0 = identity
    gt = np.zeros((nbuf, 7))
    gt[:, 6] = 1.0
    for k in range(1, N):
        xi = np.concatenate([rng.normal(0, 0.05, 3) + np.array([0.05, 0, 0]),
                             rng.normal(0, np.deg2rad(1.0), 3)])
        dt, dq = se3_exp(xi)
        t, q = se3_mul(dt, dq, gt[k - 1, :3], gt[k - 1, 3:])
        gt[k, :3], gt[k, 3:] = t, q / np.linalg.norm(q)
    gd = np.ones((nbuf, H, W))
    d = box5(np.exp(rng.normal(0, 0.3, (N, H, W))))
    d = d / d.mean()
    gd[:N] = np.clip(d, 0.1, 4.0)
    # initial state
    poses = gt.copy()
    for k in range(t0, N):
        xi = np.concatenate([rng.normal(0, 0.02, 3), rng.normal(0, np.deg2rad(0.5), 3)])
        dt, dq = se3_exp(xi)
        t, q = se3_mul(dt, dq, gt[k, :3], gt[k, 3:])
        poses[k, :3], poses[k, 3:] = t, q / np.linalg.norm(q)
    disps = gd.copy()
    disps[:N] = gd[:N] * np.exp(rng.normal(0, 0.1, (N, H, W)))
    if edges is None:
        ii, jj = make_edges(N, E, stereo, rng, t0=t0)
    else:
        ii, jj = (np.asarray(x, dtype=np.int64) for x in edges)
    coords, Z = reproject(gt, gd, intr, ii, jj)
    targets = coords + rng.normal(0, 0.25, coords.shape)
    weights = rng.uniform(0, 1, coords.shape)
    ok = (coords[:, 0] >= 0) & (coords[:, 0] <= W - 1) & (coords[:, 1] >= 0) & (coords[:, 1] <= H - 1) & (Z >= 0.25)
    weights *= ok[:, None]
    kx = np.unique(np.concatenate([np.arange(t0, N), ii]))
    eta = 0.2 * rng.uniform(0, 0.01, (len(kx), H, W)) + 1e-7
    sens = np.zeros((nbuf, H, W))
    if rgbd:
        mask = rng.uniform(0, 1, (N, H, W)) < 0.7
        sens[:N] = np.where(mask, gd[:N], 0.0)
    f32 = np.float32
    return BAProblem(poses=poses.astype(f32), disps=disps.astype(f32), intrinsics=intr.astype(f32),
                     disps_sens=sens.astype(f32), targets=targets.astype(f32),
                     weights=weights.astype(f32), eta=eta.astype(f32), ii=ii, jj=jj, t0=t0, t1=N,
                     lm=lm, ep=ep, radius=radius, gt_poses=gt, gt_disps=gd, gt_coords=coords)


def make_config(name, **over):
    N, E, H, W, r, stereo, lm, ep = CONFIGS[name]
    kw = dict(N=N, E=E, H=H, W=W, stereo=stereo, lm=lm, ep=ep, seed=CONFIG_SEEDS[name], radius=r)
    kw.update(over)
    return make_ba_problem(**kw)


def make_corr_inputs(prob: BAProblem, n_edges=None, C=128, seed=0, dtype=np.float16):
    """fmaps [N,C,H,W] ~ N(0,1) (dtype) and per-edge query coords [E,H,W,2] = GT reprojection + U(-1.5,1.5)."""
    rng = np.random.default_rng(1000 + seed)
    N = prob.t1
    _, H, W = prob.disps.shape
    fmaps = rng.normal(0, 1, (N, C, H, W)).astype(dtype)
    E = len(prob.ii) if n_edges is None else n_edges
    c = prob.gt_coords[:E] + rng.uniform(-1.5, 1.5, prob.gt_coords[:E].shape)
    coords = np.ascontiguousarray(np.transpose(c, (0, 2, 3, 1))).astype(np.float32)  # [E,H,W,2]
    return fmaps, coords
