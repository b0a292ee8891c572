"""Second-opinion CPU baseline: one Gauss-Newton iteration of the dense BA as a batched
PyTorch-CPU computation shaped like the reference's Python twin (droid_slam/geom/ba.py:31-107 -- project all edges at
once, scatter the blocks, Schur-eliminate the depths, dense Cholesky), but with the CUDA path's constants and quirks
(src/droid_kernels.cu: MIN_DEPTH 0.25 zeroes weights :302-308, weights x 0.001 :305-306, stereo edges keep their weight in
the depth terms only :323/:356, sensor-depth term alpha = 0.05 :1396-1400, damping on the REDUCED matrix :1197/:1406,
the first window pose never feeds the depth back-substitution :1105).

TEST INFRASTRUCTURE ONLY (like the rest of oracle/): imported by bench.py's cpu_baseline leg and by tests/.  It shares no
code with ba_oracle_impl.h -- poses are 4x4 matrices, Jacobians come from the matrix form dX'/dxi = [w I | -[X']x] with
J_i = -J_j Adj(T_ij) -- so agreement with the C restatement is one more independent pin of the oracle.
"""
from __future__ import annotations

import numpy as np
import torch


def _mat(poses):
    """[n,7] (t, q=xyzw) world->camera -> [n,4,4]."""
    t, q = poses[:, :3], poses[:, 3:]
    x, y, z, w = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).view(-1, 3, 3)
    T = torch.zeros(poses.shape[0], 4, 4, dtype=poses.dtype)
    T[:, :3, :3] = R
    T[:, :3, 3] = t
    T[:, 3, 3] = 1
    return T


def _skew(v):
    z = torch.zeros_like(v[..., 0])
    return torch.stack([z, -v[..., 2], v[..., 1], v[..., 2], z, -v[..., 0], -v[..., 1], v[..., 0], z], -1).view(*v.shape[:-1], 3, 3)


def ba_step(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, lm, ep, dtype=torch.float64):
    """Returns (dx [P,6], dz [M,HW], kx [M]) of ONE iteration; inputs are numpy arrays shaped like the `ba` arguments."""
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    poses, disps, sens, tg, wt, eta = tt(poses), tt(disps), tt(disps_sens), tt(targets), tt(weights), tt(eta)
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    ii = torch.from_numpy(np.asarray(ii, np.int64))
    jj = torch.from_numpy(np.asarray(jj, np.int64))
    nbuf, H, W = disps.shape
    HW, E, P = H * W, ii.shape[0], t1 - t0
    kx = torch.unique(torch.cat([torch.arange(t0, t1), ii]))
    M = kx.shape[0]
    slot = torch.full((nbuf,), -1, dtype=torch.int64)
    slot[kx] = torch.arange(M)
    T = _mat(poses)
    Tij = T[jj] @ torch.linalg.inv(T[ii])                                   # [E,4,4]
    stereo = ii == jj
    Tij[stereo] = torch.eye(4, dtype=dtype)
    Tij[stereo, 0, 3] = -0.1                                                # dk:255-258 baseline of a stereo pair
    R, t = Tij[:, :3, :3], Tij[:, :3, 3]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dtype), torch.arange(W, dtype=dtype), indexing="ij")
    X0 = torch.stack([(xs - cx) / fx, (ys - cy) / fy, torch.ones_like(xs)], -1).view(1, HW, 3)
    d = disps[ii].view(E, HW, 1)
    Xp = X0 @ R.transpose(1, 2) + d * t.view(E, 1, 3)                       # X' = R X + d t (homogeneous weight d)
    x, y, z = Xp.unbind(-1)
    valid = (z >= 0.25).to(dtype)
    zi = torch.where(z >= 0.25, 1.0 / z, torch.zeros_like(z))
    pred = torch.stack([fx * x * zi + cx, fy * y * zi + cy], -1)            # [E,HW,2]
    r = tg.view(E, 2, HW).transpose(1, 2) - pred
    w = 0.001 * wt.view(E, 2, HW).transpose(1, 2) * valid.unsqueeze(-1)
    Jp = torch.zeros(E, HW, 2, 3, dtype=dtype)                              # d(u,v)/dX'
    Jp[..., 0, 0] = fx * zi
    Jp[..., 0, 2] = -fx * x * zi * zi
    Jp[..., 1, 1] = fy * zi
    Jp[..., 1, 2] = -fy * y * zi * zi
    dX = torch.cat([d.unsqueeze(-1) * torch.eye(3, dtype=dtype).view(1, 1, 3, 3).expand(E, HW, 3, 3), -_skew(Xp)], -1)  # [E,HW,3,6]
    Jj = Jp @ dX                                                            # [E,HW,2,6]
    Adj = torch.zeros(E, 6, 6, dtype=dtype)                                 # Adj(T_ij) for xi = (tau, phi)
    Adj[:, :3, :3] = R
    Adj[:, 3:, 3:] = R
    Adj[:, :3, 3:] = _skew(t) @ R
    Ji = -(Jj @ Adj.view(E, 1, 6, 6))
    Jz = (Jp @ t.view(E, 1, 3, 1)).squeeze(-1)                              # [E,HW,2]
    # depth terms keep the weight of stereo edges, pose terms do not (dk:323, :356)
    wz = w
    wp = w * (~stereo).to(dtype).view(E, 1, 1)
    n = 6 * P
    Hpp = torch.zeros(n, n, dtype=dtype)
    bp = torch.zeros(n, dtype=dtype)
    Epz = torch.zeros(P, 6, M, HW, dtype=dtype)                             # pose x depth blocks (dense: config 1 only)
    C = torch.zeros(M, HW, dtype=dtype)
    wv = torch.zeros(M, HW, dtype=dtype)
    ms = slot[ii]
    C.index_add_(0, ms, (wz * Jz * Jz).sum(-1))
    wv.index_add_(0, ms, (wz * r * Jz).sum(-1))
    for (fa, Ja) in ((ii, Ji), (jj, Jj)):
        pa = fa - t0
        oka = (pa >= 0) & (pa < P)
        va = torch.einsum("ekc,ekca,ekc->ea", wp, Ja, r)
        bp.view(P, 6).index_add_(0, pa[oka], va[oka])
        ea = torch.einsum("ekc,ekca,ekc->eak", wp, Ja, Jz)                  # [E,6,HW]
        flat = Epz.permute(0, 2, 1, 3).contiguous().view(P * M, 6, HW)      # block (pose, depth slot)
        flat.index_add_(0, pa[oka] * M + ms[oka], ea[oka])
        Epz = flat.view(P, M, 6, HW).permute(0, 2, 1, 3).contiguous()
        for (fb, Jb) in ((ii, Ji), (jj, Jj)):
            pb = fb - t0
            ok = oka & (pb >= 0) & (pb < P)
            hab = torch.einsum("ekc,ekca,ekcb->eab", wp, Ja, Jb)
            Hb = Hpp.view(P, 6, P, 6).permute(0, 2, 1, 3).contiguous().view(P * P, 6, 6)
            Hb.index_add_(0, pa[ok] * P + pb[ok], hab[ok])
            Hpp = Hb.view(P, P, 6, 6).permute(0, 2, 1, 3).contiguous().view(n, n)
    alpha = 0.05
    s = sens[kx].view(M, HW)
    msk = (s > 0).to(dtype)
    C = C + msk * alpha + (1 - msk) * eta.view(M, HW)
    wv = wv - msk * alpha * (disps[kx].view(M, HW) - s)
    Q = 1.0 / C
    Ef = Epz.view(n, M * HW)
    red = Hpp - (Ef * Q.view(1, -1)) @ Ef.t()
    rb = bp - Ef @ (Q * wv).view(-1)
    red = red + torch.diag(ep + lm * torch.diagonal(red))
    L = torch.linalg.cholesky(red)
    dx = torch.cholesky_solve(rb.view(-1, 1), L).view(-1)
    dxq = dx.clone()
    dxq[:6] = 0.0
    dz = Q * (wv - (Ef.t() @ dxq).view(M, HW))
    return dx.view(P, 6).numpy(), dz.numpy(), kx.numpy()
