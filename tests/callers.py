"""Caller-shaped harness: the CALL SEQUENCES (tensor views, dtypes, index arithmetic) with which the
reference's Python side drives `droid_backends`, restated compactly so that the tests can replay
them through `import droid_backends` (the product) unchanged.  Not product code and not a copy of
the reference's modules: only the shapes/dtypes/ordering at the extension boundary are kept, the
networks are replaced by seeded random tensors.

  VolumeLookup      droid_slam/modules/corr.py:6-21 (CorrSampler), :23-50 (CorrBlock)
  FmapLookup        droid_slam/modules/corr.py:74-90 (CorrLayer), :92-139 (AltCorrBlock)
  Video             droid_slam/depth_video.py:15-60 (buffers), :160-190 (distance), :192-204 (cuda_ba)
  frontend_update   droid_slam/factor_graph.py:196-246 (update)
  backend_update    droid_slam/factor_graph.py:249-300 (update_lowmem)
"""
import torch
import torch.nn.functional as F

import droid_backends


class _Sampler(torch.autograd.Function):
    """modules/corr.py:6-21: forward -> corr_index_forward, backward -> corr_index_backward."""

    @staticmethod
    def forward(ctx, volume, coords, radius):
        ctx.save_for_backward(volume, coords)
        ctx.radius = radius
        out, = droid_backends.corr_index_forward(volume, coords, radius)
        return out

    @staticmethod
    def backward(ctx, g):
        volume, coords = ctx.saved_tensors
        gv, = droid_backends.corr_index_backward(volume, coords, g.contiguous(), ctx.radius)
        return gv, None, None


class _FmapDot(torch.autograd.Function):
    """modules/corr.py:74-90: forward -> altcorr_forward, backward -> altcorr_backward."""

    @staticmethod
    def forward(ctx, f1, f2, coords, r):
        ctx.r = r
        ctx.save_for_backward(f1, f2, coords)
        out, = droid_backends.altcorr_forward(f1, f2, coords, r)
        return out

    @staticmethod
    def backward(ctx, g):
        f1, f2, coords = ctx.saved_tensors
        g1, g2, gc = droid_backends.altcorr_backward(f1, f2, coords, g.contiguous(), ctx.r)
        return g1, g2, gc, None


class VolumeLookup:
    """All-pairs volume pyramid + 4-level lookup (modules/corr.py:23-50).  fmap1/fmap2 [1,E,C,h,w];
    under autocast the matmul yields an fp16 volume, which is what the lookup then sees."""

    def __init__(self, fmap1, fmap2, levels=4, radius=3):
        self.levels, self.radius = levels, radius
        b, n, c, h, w = fmap1.shape
        a = fmap1.reshape(b * n, c, h * w) / 4.0
        bb = fmap2.reshape(b * n, c, h * w) / 4.0
        vol = torch.matmul(a.transpose(1, 2), bb).view(b, n, h, w, h, w)
        vol = vol.reshape(b * n * h * w, 1, h, w)
        self.pyramid = []
        for l in range(levels):
            self.pyramid.append(vol.view(b * n, h, w, h // 2 ** l, w // 2 ** l))
            vol = F.avg_pool2d(vol, 2, stride=2)

    def __call__(self, coords):
        b, n, h, w, _ = coords.shape
        c = coords.permute(0, 1, 4, 2, 3).contiguous().view(b * n, 2, h, w)
        out = [_Sampler.apply(self.pyramid[l], c / 2 ** l, self.radius).view(b, n, -1, h, w)
               for l in range(self.levels)]
        return torch.cat(out, dim=2)


class FmapLookup:
    """Channels-last fmap pyramid + per-level on-the-fly lookup (modules/corr.py:92-139)."""

    def __init__(self, fmaps, levels=4, radius=3):
        self.levels, self.radius = levels, radius
        b, n, c, h, w = fmaps.shape
        x = fmaps.view(b * n, c, h, w) / 4.0
        self.pyramid = []
        for l in range(levels):
            self.pyramid.append(x.permute(0, 2, 3, 1).contiguous().view(b, n, h // 2 ** l, w // 2 ** l, c))
            x = F.avg_pool2d(x, 2, stride=2)

    def __call__(self, coords, ii, jj):
        squeeze = coords.dim() == 5
        if squeeze:
            coords = coords.unsqueeze(-2)
        b, n, h, w, s, _ = coords.shape
        coords = coords.permute(0, 1, 4, 2, 3, 5)
        parts = []
        for l in range(self.levels):
            f1 = self.pyramid[0][:, ii]
            f2 = self.pyramid[l][:, jj]
            cl = (coords / 2 ** l).reshape(b * n, s, h, w, 2).contiguous()
            f1 = f1.reshape((b * n,) + f1.shape[2:])
            f2 = f2.reshape((b * n,) + f2.shape[2:])
            out = _FmapDot.apply(f1.float(), f2.float(), cl, self.radius)
            parts.append(out.view(b, n, s, -1, h, w).permute(0, 1, 3, 4, 5, 2))
        out = torch.cat(parts, dim=2)
        return (out.squeeze(-1) if squeeze else out).contiguous()


class Video:
    """The state buffers `ba` mutates and the two methods that call into the extension
    (depth_video.py:33-45 buffers, :160-190 distance, :192-204 cuda_ba)."""

    def __init__(self, poses, disps, intrinsics, disps_sens, counter):
        self.poses, self.disps, self.intrinsics, self.disps_sens = poses, disps, intrinsics, disps_sens
        self.counter = counter

    def distance(self, ii=None, jj=None, beta=0.3, bidirectional=True):
        matrix = ii is None
        if matrix:
            n = self.counter
            ii, jj = torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")
        ii = ii.to(device="cuda", dtype=torch.long).reshape(-1)      # format_indicies, depth_video.py:113-127
        jj = jj.to(device="cuda", dtype=torch.long).reshape(-1)
        if bidirectional:
            poses = self.poses[:self.counter].clone()
            d1 = droid_backends.frame_distance(poses, self.disps, self.intrinsics[0], ii, jj, beta)
            d2 = droid_backends.frame_distance(poses, self.disps, self.intrinsics[0], jj, ii, beta)
            d = .5 * (d1 + d2)
        else:
            d = droid_backends.frame_distance(self.poses, self.disps, self.intrinsics[0], ii, jj, beta)
        return d.reshape(self.counter, self.counter) if matrix else d

    def cuda_ba(self, target, weight, eta, ii, jj, t0=1, t1=None, itrs=2, lm=1e-4, ep=0.1, motion_only=False):
        if t1 is None:
            t1 = max(ii.max().item(), jj.max().item()) + 1
        droid_backends.ba(self.poses, self.disps, self.intrinsics[0], self.disps_sens, target, weight, eta, ii, jj,
                          t0, t1, itrs, lm, ep, motion_only)
        self.disps.clamp_(min=0.001)


def ba_inputs(target, weight, damping_buf, ii, h, w, EP=1e-7):
    """factor_graph.py:234-238 / :290-293: eta rows and the [E,2,h,w] views handed to cuda_ba."""
    eta = .2 * damping_buf[torch.unique(ii)].contiguous() + EP
    tg = target.view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
    wt = weight.view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
    return tg, wt, eta
