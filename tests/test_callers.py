"""Caller-shaped parity (SURVEY.md section 8b "callers and exact call shapes"): replays the call
sequences of the reference's Python side (tests/callers.py: CorrBlock / AltCorrBlock / their autograd
wrappers / DepthVideo.distance / cuda_ba + clamp_) through `import droid_backends` with the callers'
tensor views and dtypes -- autocast fp16 volume, `.float()` alt path, `intrinsics[0]`, meshgrid indices,
`[E,2,h,w]` permuted targets -- and checks every extension result against the CPU oracle
(parity unpinned, oracle/__init__.py).  One frontend-shaped update (itrs=2, lm=1e-4, ep=0.1,
factor_graph.py:240-241) and one backend-shaped update (t0=1, lm=1e-5, ep=1e-2, chunks of 8 source frames,
factor_graph.py:266-298)."""
import numpy as np
import pytest

from util import quat_angle

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _video(torch, prob, buffer):
    """DepthVideo-like buffers longer than the window (depth_video.py:33-45): per-frame intrinsics rows."""
    import callers
    n = prob.t1
    poses = torch.zeros(buffer, 7, device="cuda")
    poses[:, 6] = 1.0
    poses[:n] = torch.from_numpy(prob.poses).cuda()
    H, W = prob.disps.shape[1:]
    disps = torch.ones(buffer, H, W, device="cuda")
    disps[:n] = torch.from_numpy(prob.disps).cuda()
    sens = torch.zeros(buffer, H, W, device="cuda")
    sens[:n] = torch.from_numpy(prob.disps_sens).cuda()
    intr = torch.from_numpy(prob.intrinsics).cuda()[None].repeat(buffer, 1).contiguous()
    return callers.Video(poses, disps, intr, sens, n)


def _state_err(video, ref, n):
    p = video.poses[:n].cpu().numpy()
    d = video.disps[:n].cpu().numpy()
    et = np.abs(p[:, :3] - ref["poses"][:n, :3]).max()
    er = quat_angle(p[:, 3:].astype(np.float64), ref["poses"][:n, 3:]).max()
    ed = np.abs(d - np.maximum(ref["disps"][:n], 0.001)).max()
    return et, er, ed


def test_frontend_shaped_update(backends, oracle):
    """factor_graph.py:196-246: reproject -> motion features -> CorrBlock lookup (autocast: fp16 volume)
    -> [network replaced by seeded tensors] -> target/weight/damping views -> cuda_ba(itrs=2) -> clamp_."""
    import callers
    from droid_backends import synth
    from oracle import geom as ogeom
    torch = _torch()
    prob = synth.make_ba_problem(N=9, E=26, H=48, W=64, seed=11, lm=1e-4, ep=0.1)
    H, W, n, buffer = 48, 64, prob.t1, 16
    video = _video(torch, prob, buffer)
    rng = np.random.default_rng(5)
    fmaps = torch.from_numpy(rng.normal(0, 1, (buffer, 128, H, W)).astype(np.float16)).cuda()
    ii = torch.from_numpy(prob.ii).cuda()
    jj = torch.from_numpy(prob.jj).cuda()
    E = len(prob.ii)
    with torch.autocast("cuda", enabled=True):          # add_factors runs under autocast (factor_graph.py:85)
        corr_op = callers.VolumeLookup(fmaps[None, ii], fmaps[None, jj])
    assert corr_op.pyramid[0].dtype == torch.float16 and tuple(corr_op.pyramid[3].shape) == (E, H, W, 6, 8)

    y, x = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    coords0 = torch.stack([x, y], dim=-1)
    target0 = torch.from_numpy(np.ascontiguousarray(prob.targets.transpose(0, 2, 3, 1))).cuda()[None]  # [1,E,H,W,2]
    # motion features + reprojection through the fused operator (depth_video.py:150-158 without lietorch)
    motn, coords1, mask = backends.motion_features(video.poses, video.disps, video.intrinsics, ii, jj, target0)
    rc, rv = ogeom.reproject(video.poses.cpu().numpy(), video.disps.cpu().numpy(), video.intrinsics.cpu().numpy(),
                             prob.ii, prob.jj)
    assert np.abs(coords1[0].cpu().numpy() - rc).max() < 5e-3            # pixels, fp32 vs fp64
    ref_motn = np.clip(np.concatenate([rc - coords0.cpu().numpy()[None], target0[0].cpu().numpy() - rc], -1)
                       .transpose(0, 3, 1, 2), -64, 64)
    assert np.abs(motn[0].cpu().numpy() - ref_motn).max() < 5e-3

    corr = corr_op(coords1)                                               # [1,E,4*49,H,W] fp16
    assert corr.dtype == torch.float16 and tuple(corr.shape) == (1, E, 196, H, W)
    cflat = coords1.permute(0, 1, 4, 2, 3).contiguous().view(E, 2, H, W)
    fused, = backends.corr_pyramid_forward(corr_op.pyramid, cflat, 3)
    assert torch.equal(fused, corr[0])                                   # CorrBlock.__call__ without the cat
    for l in range(4):
        ref = oracle.corr_index_forward(corr_op.pyramid[l].cpu().numpy(), (cflat / 2 ** l).cpu().numpy(), 3)
        got = corr[0, :, 49 * l:49 * (l + 1)].reshape(E, 7, 7, H, W).cpu().numpy()
        assert np.array_equal(got, ref), f"level {l}: {(got != ref).sum()} differing half values"

    # stand-ins for the update operator's outputs, then the caller's arithmetic around cuda_ba
    gt = torch.from_numpy(np.ascontiguousarray(prob.gt_coords.transpose(0, 2, 3, 1))).float().cuda()[None]
    delta = (gt - coords1) + torch.from_numpy(rng.normal(0, 0.25, (1, E, H, W, 2))).float().cuda()
    weight = torch.from_numpy(prob.weights.transpose(0, 2, 3, 1).copy()).cuda()[None].half()
    target = coords1 + delta.half().to(dtype=torch.float)
    weight = weight.to(dtype=torch.float)
    damping_buf = torch.full((buffer, H, W), 1e-6, device="cuda")
    uii = torch.unique(ii)
    damping_buf[uii] = torch.from_numpy(rng.uniform(0, 0.01, (len(uii), H, W))).float().cuda()
    t0 = max(1, int(ii.min().item()) + 1)
    tg, wt, eta = callers.ba_inputs(target, weight, damping_buf, ii, H, W)
    assert tuple(tg.shape) == (E, 2, H, W) and tuple(eta.shape) == (len(uii), H, W)
    before = (video.poses.cpu().numpy(), video.disps.cpu().numpy())
    video.cuda_ba(tg, wt, eta, ii, jj, t0, None, itrs=2, lm=1e-4, ep=0.1, motion_only=False)
    torch.cuda.synchronize()
    assert backends.ba_status()[0] & 11 == 0
    t1 = int(max(prob.ii.max(), prob.jj.max())) + 1
    ref = oracle.ba(before[0], before[1], video.intrinsics[0].cpu().numpy(), video.disps_sens.cpu().numpy(),
                    tg.cpu().numpy(), wt.cpu().numpy(), eta.cpu().numpy(), prob.ii, prob.jj, t0, t1, 2, 1e-4, 0.1, False, storage_f32=True)
    et, er, ed = _state_err(video, ref, buffer)
    print(f"[frontend-shaped] max|dt|={et:.2e} angle={er:.2e} max|ddisp|={ed:.2e}")
    assert et < TOL and er < TOL and ed < TOL
    assert float(video.disps.min()) >= 0.001


def test_backend_shaped_update(backends, oracle):
    """factor_graph.py:249-300 (update_lowmem): AltCorrBlock over video.fmaps.view(1, num*rig, ...), edges
    visited in chunks of 8 source frames with frame indices rig*ii / rig*jj + (ii == jj), fp16 fmaps with
    the explicit .float() of corr_fn, then cuda_ba(t0=1, t1=t, itrs=2, lm=1e-5, ep=1e-2) + clamp_."""
    import callers
    from droid_backends import synth
    torch = _torch()
    prob = synth.make_ba_problem(N=20, E=120, H=48, W=64, seed=12, lm=1e-5, ep=1e-2)
    H, W, t, buffer, rig = 48, 64, prob.t1, 24, 1
    video = _video(torch, prob, buffer)
    rng = np.random.default_rng(6)
    vfmaps = torch.from_numpy(rng.normal(0, 1, (buffer, rig, 128, H, W)).astype(np.float16)).cuda()
    num, rig_, ch, ht, wd = vfmaps.shape
    corr_op = callers.FmapLookup(vfmaps.view(1, num * rig_, ch, ht, wd))
    assert corr_op.pyramid[0].dtype == torch.float16
    ii = torch.from_numpy(prob.ii).cuda()
    jj = torch.from_numpy(prob.jj).cuda()
    E = len(prob.ii)
    coords1, _ = backends.reproject(video.poses, video.disps, video.intrinsics, ii, jj)
    target = torch.zeros(1, E, H, W, 2, device="cuda")
    weight = torch.zeros(1, E, H, W, 2, device="cuda")
    damping_buf = torch.full((buffer, H, W), 1e-6, device="cuda")
    gt = torch.from_numpy(np.ascontiguousarray(prob.gt_coords.transpose(0, 2, 3, 1))).float().cuda()[None]
    wsyn = torch.from_numpy(prob.weights.transpose(0, 2, 3, 1).copy()).cuda()[None]
    checked = 0
    s = 8
    for i in range(0, int(jj.max()) + 1, s):
        v = (ii >= i) & (ii < i + s)
        iis, jjs = ii[v], jj[v]
        i1, i2 = rig * iis, rig * jjs + (iis == jjs).long()
        corr1 = corr_op(coords1[:, v], i1, i2)                       # [1,Ev,196,H,W] fp32
        assert corr1.dtype == torch.float32 and tuple(corr1.shape) == (1, int(v.sum()), 196, H, W)
        fused, = backends.altcorr_pyramid_forward([p.float() for p in corr_op.pyramid], coords1[0, v].contiguous(),
                                                  i1, i2, 3)
        assert torch.equal(fused, corr1[0])
        # the half pyramid as AltCorrBlock holds it, straight into the f16 matrix-core entry point: no `.float()` copies
        fused_h, = backends.altcorr_pyramid_forward(corr_op.pyramid, coords1[0, v].contiguous(), i1, i2, 3)
        assert fused_h.dtype == torch.float32
        assert float((fused_h - corr1[0]).abs().max()) <= 2e-6 * float(corr1.abs().max())
        for k in (0, int(v.sum()) - 1):                              # first and last edge of the chunk vs the oracle
            for l in (0, 3):
                f1 = corr_op.pyramid[0][0, i1[k]].float().cpu().numpy()[None]
                f2 = corr_op.pyramid[l][0, i2[k]].float().cpu().numpy()[None]
                cl = (coords1[0, v][k] / 2 ** l).cpu().numpy()[None, None]
                ref = oracle.altcorr_forward(f1, f2, cl, 3, acc_dtype=np.float64)[0, 0]
                got = corr1[0, k, 49 * l:49 * (l + 1)].cpu().numpy()
                assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max()
                checked += 1
        delta = (gt[:, v] - coords1[:, v]) + torch.from_numpy(rng.normal(0, 0.25, (1, int(v.sum()), H, W, 2))).float().cuda()
        target[:, v] = coords1[:, v] + delta.float()
        weight[:, v] = wsyn[:, v].float()
        u = torch.unique(iis)
        damping_buf[u] = torch.from_numpy(rng.uniform(0, 0.01, (len(u), H, W))).float().cuda()
    assert checked >= 8
    tg, wt, eta = callers.ba_inputs(target, weight, damping_buf, ii, H, W)
    before = (video.poses.cpu().numpy(), video.disps.cpu().numpy())
    video.cuda_ba(tg, wt, eta, ii, jj, 1, t, itrs=2, lm=1e-5, ep=1e-2, motion_only=False)
    torch.cuda.synchronize()
    assert backends.ba_status()[0] & 11 == 0
    ref = oracle.ba(before[0], before[1], video.intrinsics[0].cpu().numpy(), video.disps_sens.cpu().numpy(),
                    tg.cpu().numpy(), wt.cpu().numpy(), eta.cpu().numpy(), prob.ii, prob.jj, 1, t, 2, 1e-5, 1e-2, False, storage_f32=True)
    et, er, ed = _state_err(video, ref, buffer)
    print(f"[backend-shaped] max|dt|={et:.2e} angle={er:.2e} max|ddisp|={ed:.2e}")
    assert et < TOL and er < TOL and ed < TOL


def test_motion_only_filler_shaped_call(backends, oracle):
    """trajectory_filler.py:62-73: edges from keyframes to the frames being filled, motion_only=True;
    eta is passed but unused, disps must not change (apart from the caller's clamp)."""
    import callers
    from droid_backends import synth
    torch = _torch()
    prob = synth.make_ba_problem(N=8, E=28, H=48, W=64, seed=13, lm=1e-4, ep=0.1)
    video = _video(torch, prob, 8)
    ii = torch.from_numpy(prob.ii).cuda()
    jj = torch.from_numpy(prob.jj).cuda()
    tg = torch.from_numpy(prob.targets).cuda()
    wt = torch.from_numpy(prob.weights).cuda()
    eta = torch.from_numpy(prob.eta).cuda()
    before = (video.poses.cpu().numpy(), video.disps.cpu().numpy())
    video.cuda_ba(tg, wt, eta, ii, jj, 4, 8, itrs=2, lm=1e-4, ep=0.1, motion_only=True)
    ref = oracle.ba(before[0], before[1], prob.intrinsics, prob.disps_sens, prob.targets, prob.weights, prob.eta,
                    prob.ii, prob.jj, 4, 8, 2, 1e-4, 0.1, True, storage_f32=True)
    et, er, ed = _state_err(video, ref, 8)
    assert et < TOL and er < TOL and ed == 0.0
    assert np.array_equal(video.poses[:4].cpu().numpy(), before[0][:4])      # frames before t0 are fixed


def test_distance_matrix_and_candidate_pairs(backends, oracle):
    """depth_video.py:160-190: all-pairs matrix (meshgrid indices, CPU int64 -> cuda) and the candidate list
    of add_proximity_factors (factor_graph.py:318-326), bidirectional mean of two frame_distance calls."""
    from droid_backends import synth
    torch = _torch()
    prob = synth.make_ba_problem(N=12, E=60, H=48, W=64, seed=14)
    video = _video(torch, prob, 20)
    d = video.distance(beta=0.3)
    assert tuple(d.shape) == (12, 12)
    n = 12
    gi, gj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    gi, gj = gi.reshape(-1), gj.reshape(-1)
    P, D, K = video.poses.cpu().numpy(), video.disps.cpu().numpy(), video.intrinsics[0].cpu().numpy()
    r1 = oracle.frame_distance(P, D, K, gi, gj, 0.3)
    r2 = oracle.frame_distance(P, D, K, gj, gi, 0.3)
    ref = (0.5 * (r1 + r2)).reshape(n, n)
    got = d.cpu().numpy()
    fin = np.isfinite(ref) & (ref < 1e3)
    assert np.array_equal(np.isfinite(got) & (got < 1e3), fin)
    assert np.abs(got[fin] - ref[fin]).max() < 1e-3 * max(1.0, np.abs(ref[fin]).max())
    # the one-launch all-pairs variant (no index tensors) against the meshgrid call sequence above and the oracle
    dm = backends.frame_distance_matrix(video.poses, video.disps, video.intrinsics[0], n, 0.3)
    assert tuple(dm.shape) == (n, n)
    gm = dm.cpu().numpy()
    assert np.array_equal(np.isfinite(gm) & (gm < 1e3), fin)
    assert np.abs(gm[fin] - ref[fin]).max() < 1e-3 * max(1.0, np.abs(ref[fin]).max())
    assert np.abs(gm[fin] - got[fin]).max() < 1e-5 * max(1.0, np.abs(got[fin]).max())
    d1 = backends.frame_distance_matrix(video.poses, video.disps, video.intrinsics[0], n, 0.3, bidirectional=False)
    r1m = r1.reshape(n, n)
    f1m = np.isfinite(r1m) & (r1m < 1e3)
    assert np.abs(d1.cpu().numpy()[f1m] - r1m[f1m]).max() < 1e-3 * max(1.0, np.abs(r1m[f1m]).max())
    ix = torch.arange(0, n)
    jx = torch.arange(2, n)
    ii, jj = torch.meshgrid(ix, jx, indexing="ij")
    d2 = video.distance(ii.reshape(-1), jj.reshape(-1), beta=0.25)
    q1 = oracle.frame_distance(P, D, K, ii.reshape(-1).numpy(), jj.reshape(-1).numpy(), 0.25)
    q2 = oracle.frame_distance(P, D, K, jj.reshape(-1).numpy(), ii.reshape(-1).numpy(), 0.25)
    ref2 = 0.5 * (q1 + q2)
    fin = np.isfinite(ref2) & (ref2 < 1e3)
    assert np.abs(d2.cpu().numpy()[fin] - ref2[fin]).max() < 1e-3 * max(1.0, np.abs(ref2[fin]).max())


def test_autograd_wrappers_forward_and_backward(backends, oracle):
    """CorrSampler / CorrLayer (modules/corr.py:6-21, :74-90) driven by torch.autograd: gradients reach the
    volume / the feature maps through corr_index_backward / altcorr_backward and match the oracle
    element-wise (volume gradient: same rounding order as correlation_kernels.cu:106-117 => exact for fp32)."""
    import callers
    from oracle import corr as oc
    torch = _torch()
    rng = np.random.default_rng(21)
    E, C, H, W = 3, 32, 16, 24
    f1 = torch.from_numpy(rng.normal(0, 1, (1, E, C, H, W)).astype(np.float32)).cuda().requires_grad_(True)
    f2 = torch.from_numpy(rng.normal(0, 1, (1, E, C, H, W)).astype(np.float32)).cuda().requires_grad_(True)
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    c = np.stack([x[None] + rng.uniform(-3, 3, (E, H, W)), y[None] + rng.uniform(-3, 3, (E, H, W))], -1).astype(np.float32)
    coords = torch.from_numpy(c).cuda()[None]
    op = callers.VolumeLookup(f1, f2)
    for p in op.pyramid:
        p.retain_grad()
    out = op(coords)
    g = torch.from_numpy(rng.normal(size=tuple(out.shape)).astype(np.float32)).cuda()
    out.backward(g)
    assert f1.grad is not None and float(f1.grad.abs().max()) > 0
    cflat = coords.permute(0, 1, 4, 2, 3).contiguous().view(E, 2, H, W).cpu().numpy()
    # level-3 volume gradient is what corr_index_backward produced for that level alone
    gl = g[0, :, 49 * 3:].reshape(E, 7, 7, H, W).cpu().numpy()
    ref = oc.corr_index_backward((E, H, W, H >> 3, W >> 3), cflat / 8, gl, 3)
    assert np.array_equal(op.pyramid[3].grad.cpu().numpy(), ref)

    # direct element-wise check of every level (fp32 exact, fp16 exact: same rounding points)
    for l, dt in ((0, np.float32), (1, np.float16), (2, np.float32)):
        vol = torch.zeros((E, H, W, H >> l, W >> l), dtype=torch.float16 if dt == np.float16 else torch.float32, device="cuda")
        cg = rng.normal(size=(E, 7, 7, H, W)).astype(dt)
        gv, = backends.corr_index_backward(vol, torch.from_numpy(cflat / 2 ** l).cuda(), torch.from_numpy(cg).cuda(), 3)
        ref = oc.corr_index_backward(tuple(vol.shape), cflat / 2 ** l, cg, 3)
        assert np.array_equal(gv.cpu().numpy(), ref), (l, dt)

    fm = torch.from_numpy((rng.normal(0, 1, (1, 5, C, H, W)) / 1).astype(np.float32)).cuda().requires_grad_(True)
    aop = callers.FmapLookup(fm)
    ii = torch.tensor([0, 1, 4], device="cuda")
    jj = torch.tensor([1, 3, 4], device="cuda")
    out = aop(coords, ii, jj)
    assert tuple(out.shape) == (1, E, 196, H, W)
    g = torch.from_numpy(rng.normal(size=tuple(out.shape)).astype(np.float32)).cuda()
    out.backward(g)
    # oracle: chain the fp64 alt-corr gradients of the four levels back to the pooled pyramid by autograd on CPU
    fm_cpu = fm.detach().cpu().double().requires_grad_(True)
    import torch.nn.functional as F
    xx = fm_cpu.view(5, C, H, W) / 4.0
    total = 0.0
    for l in range(4):
        lvl = xx.permute(0, 2, 3, 1).contiguous()
        if l == 0:
            lvl0 = lvl
        a, b = lvl0[ii.cpu()], lvl[jj.cpu()]
        cl = (c / 2 ** l)[:, None]
        r1, r2 = oc.altcorr_backward(a.detach().numpy(), b.detach().numpy(), cl,
                                     g[0, :, 49 * l:49 * (l + 1)].cpu().numpy()[:, None], 3)
        total = total + (a * torch.from_numpy(r1)).sum() + (b * torch.from_numpy(r2)).sum()
        xx = F.avg_pool2d(xx, 2, stride=2)
    total.backward()
    ref = fm_cpu.grad.numpy()
    err = np.abs(fm.grad.cpu().numpy() - ref).max() / np.abs(ref).max()
    print(f"AltCorrBlock autograd: rel err {err:.2e}")
    assert err < 2e-5
