"""Correlation-lookup parity AT THE BASELINE.json SHAPES (VERDICT r01 "configs_untested"):

* cfg2 leg: all-pairs volume lookup, 48x64, 4 pyramid levels, r=3, fp16 (autocast, factor_graph.py:85/197)
  and fp32; the volume is built like CorrBlock (modules/corr.py:24-38, :63-71) from synthetic fmaps,
  coords from synth.make_corr_inputs of the cfg2 graph.  Bit-exact against oracle/corr.py.
* cfg3/cfg4 leg: alt-corr 48x64, C=128, r=3, all four levels, per-level `altcorr_forward` and the fused
  `altcorr_pyramid_forward`.
* cfg5 leg: alt-corr 96x128, C=128, r=4, stereo frame indexing rig*jj + (ii == jj)
  (factor_graph.py:277), fmaps [frames*2, ...].
* fp16 `altcorr_forward` (altcorr_kernel.cu:308 dispatches half).

Edge counts are small (the numpy oracle is the clock), map sizes / radii / channel counts / dtypes are
the real ones: the wide row loads, the tensor-start / tensor-end fallbacks, the bounding-box block counts,
the two-round exchange and the XCD tile order of the MFMA kernel are all shape dependent.
The oracle is "parity unpinned" (oracle/__init__.py); tolerances: bit-exact for the volume lookup,
1e-5 of the output scale for fp32 alt-corr."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def synth():
    from droid_backends import synth
    return synth


def _pick_edges(prob, n):
    """half neighbouring edges (|i-j| <= 3), half long-range ones (the tail of the edge list)"""
    E = len(prob.ii)
    return np.concatenate([np.arange(n // 2), np.arange(E - (n - n // 2), E)])


@pytest.fixture(scope="module")
def cfg2_volume(synth):
    """fp32 all-pairs pyramid of 32 cfg2 edges, built on the device with stock PyTorch exactly like
    CorrBlock.__init__ (matmul of fmaps/4, view [B*h*w,1,h2,w2], avg_pool2d), plus query coords."""
    import torch.nn.functional as F
    torch = _torch()
    prob = synth.make_config("cfg2")
    fmaps, coords = synth.make_corr_inputs(prob, C=128, seed=1)
    sel = _pick_edges(prob, 32)
    ii, jj = prob.ii[sel], prob.jj[sel]
    coords = coords[sel].copy()                       # [B,H,W,2]
    H, W = 48, 64
    # windows hanging over the first / last element of the whole tensor (per-element fallback loads)
    coords[0, 0, 0] = [0.3, 0.2]
    coords[0, 0, 1] = [-2.75, 1.5]
    coords[-1, H - 1, W - 1] = [W - 0.6, H - 0.4]
    coords[-1, H - 1, W - 2] = [W + 1.25, H - 1.5]
    coords[5, 7, 9] = [-40.0, 3.0]                    # empty window
    coords[6, 7, 9] = [31.0, 23.999]                  # exactly representable .999 fractions
    fm = torch.from_numpy(fmaps).cuda()
    f1 = fm[torch.from_numpy(ii).cuda()][None].float()    # [1,B,C,H,W]
    f2 = fm[torch.from_numpy(jj).cuda()][None].float()
    B = len(sel)
    a = f1.reshape(B, 128, H * W) / 4.0
    b = f2.reshape(B, 128, H * W) / 4.0
    corr = torch.matmul(a.transpose(1, 2), b).view(1, B, H, W, H, W)
    corr = corr.reshape(B * H * W, 1, H, W)
    pyr = []
    for l in range(4):
        pyr.append(corr.view(B, H, W, H >> l, W >> l))
        corr = F.avg_pool2d(corr, 2, stride=2)
    c = torch.from_numpy(coords).cuda()[None]          # [1,B,H,W,2] like coords1
    c = c.permute(0, 1, 4, 2, 3).contiguous().view(B, 2, H, W)   # CorrBlock.__call__ :43-44
    return pyr, c


@pytest.mark.parametrize("dtype", ["float16", "float32"])
def test_cfg2_volume_lookup_48x64_four_levels_bit_exact(backends, oracle, cfg2_volume, dtype):
    torch = _torch()
    pyr, c = cfg2_volume
    tdt = getattr(torch, dtype)
    outs = []
    for l in range(4):
        vol = pyr[l].to(tdt).contiguous()
        cl = c / 2 ** l                                # modules/corr.py:47
        out, = backends.corr_index_forward(vol, cl, 3)
        ref = oracle.corr_index_forward(vol.cpu().numpy(), cl.cpu().numpy(), 3)
        got = out.cpu().numpy()
        nbad = int(np.sum(got != ref))
        assert got.dtype == ref.dtype and got.shape == (32, 7, 7, 48, 64)
        assert nbad == 0, f"level {l} {dtype}: {nbad} differing elements, max abs " \
                          f"{np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()}"
        assert np.abs(ref.astype(np.float64)).max() > 0.1   # the comparison is not vacuous
        outs.append(out.view(1, 32, -1, 48, 64))
    cat = torch.cat(outs, dim=2)
    assert cat.shape == (1, 32, 4 * 49, 48, 64)
    # the one-call variant (no torch.cat, no per-level coordinate tensors) is the same tensor bit for bit
    fused, = backends.corr_pyramid_forward([p.to(tdt).contiguous() for p in pyr], c, 3)
    assert fused.dtype == tdt and torch.equal(fused, cat[0])


def _alt_pyramid(torch, fmaps_f16, half=False):
    """AltCorrBlock.__init__ (modules/corr.py:92-104): /4, channels-last levels by avg_pool2d.  `half`: the
    arithmetic of the SLAM path -- `video.fmaps` is torch.half (depth_video.py:44), so `/ 4.0` and the pooling
    stay in half and the pyramid IS half (factor_graph.py:260-261); otherwise a float32 pyramid (training)."""
    import torch.nn.functional as F
    x = torch.from_numpy(fmaps_f16).cuda()
    x = (x if half else x.float()) / 4.0
    pyr = []
    for _ in range(4):
        pyr.append(x.permute(0, 2, 3, 1).contiguous()[None])    # [1,frames,h,w,C]
        x = F.avg_pool2d(x, 2, stride=2)
    assert pyr[0].dtype == (torch.float16 if half else torch.float32)
    return pyr


def _check_alt_levels(backends, oracle, torch, pyr, coords, i1, i2, r, tag):
    """per-level altcorr_forward on gathered maps (corr_fn :105-125) + fused pyramid entry point"""
    E, H, W = coords.shape[:3]
    parts = []
    for l in range(4):
        f1 = pyr[0][0][i1].contiguous().float()
        f2 = pyr[l][0][i2].contiguous().float()
        cl = (coords / 2 ** l).reshape(E, 1, H, W, 2).contiguous()
        out, = backends.altcorr_forward(f1, f2, cl, r)
        ref = oracle.altcorr_forward(f1.cpu().numpy(), f2.cpu().numpy(), cl.cpu().numpy(), r, acc_dtype=np.float64)
        got = out.cpu().numpy()
        assert np.isfinite(got).all()
        err = np.abs(got - ref).max() / np.abs(ref).max()
        print(f"[{tag}] level {l}: rel err {err:.2e} (scale {np.abs(ref).max():.2f})")
        assert err < 1e-5, (tag, l, err)
        parts.append(out[:, 0])
    fused, = backends.altcorr_pyramid_forward(pyr, coords, i1, i2, r)
    assert torch.equal(fused, torch.cat(parts, dim=1))


def _check_alt_half_pyramid(backends, oracle, torch, pyr_h, coords, i1, i2, r, tag):
    """The half pyramid of the SLAM path through the f16 matrix-core entry point: against the fp64 oracle evaluated
    on the SAME half values (what `corr_fn`'s `.float()` copies hold, modules/corr.py:120) <= 1e-5 of the output
    scale, against this library's fp32 kernel on the widened maps <= 2e-6 (only the summation order differs:
    products of two halves are exact in fp32), and `altcorr_forward(half, half)` (altcorr_kernel.cu:308) == that
    fp32 result rounded to half within one half ulp."""
    E, H, W = coords.shape[:3]
    assert pyr_h[0].dtype == torch.float16
    fused, = backends.altcorr_pyramid_forward(pyr_h, coords, i1, i2, r)
    assert fused.dtype == torch.float32 and tuple(fused.shape) == (E, 4 * (2 * r + 1) ** 2, H, W)
    got = fused.cpu().numpy()
    assert np.isfinite(got).all()
    pyr_f = [p.float() for p in pyr_h]
    wide, = backends.altcorr_pyramid_forward(pyr_f, coords, i1, i2, r)
    n = (2 * r + 1) ** 2
    for l in range(4):
        f1 = pyr_h[0][0][i1].contiguous()
        f2 = pyr_h[l][0][i2].contiguous()
        cl = (coords / 2 ** l).reshape(E, 1, H, W, 2).contiguous()
        ref = oracle.altcorr_forward(f1.float().cpu().numpy(), f2.float().cpu().numpy(), cl.cpu().numpy(), r,
                                     acc_dtype=np.float64)[:, 0]
        scale = np.abs(ref).max()
        e64 = np.abs(got[:, l * n:(l + 1) * n] - ref).max() / scale
        e32 = float((fused[:, l * n:(l + 1) * n] - wide[:, l * n:(l + 1) * n]).abs().max()) / scale
        print(f"[{tag}] level {l}: f16-MFMA vs fp64 oracle {e64:.2e}, vs fp32 kernel on widened maps {e32:.2e} "
              f"(scale {scale:.2f})")
        assert e64 < 1e-5 and e32 < 2e-6, (tag, l, e64, e32)
        out_h, = backends.altcorr_forward(f1, f2, cl, r)          # half in, half out
        assert out_h.dtype == torch.float16
        d = (out_h[:, 0].float() - fused[:, l * n:(l + 1) * n]).abs()
        ulp = torch.clamp(fused[:, l * n:(l + 1) * n].abs(), min=2.0 ** -14) * 2.0 ** -10
        assert bool((d <= ulp).all()), (tag, l, float((d / ulp).max()))


def test_cfg3_altcorr_48x64_c128_r3(backends, oracle, synth):
    torch = _torch()
    prob = synth.make_config("cfg3")
    rng = np.random.default_rng(33)
    fmaps = rng.normal(0, 1, (prob.t1, 128, 48, 64)).astype(np.float16)
    _, coords = synth.make_corr_inputs(prob, n_edges=None, C=1, seed=2)
    sel = _pick_edges(prob, 4)
    c = coords[sel].copy()
    c[0, 0, 0] = [-2.5, -1.5]
    c[0, 47, 63] = [65.25, 48.75]
    c[1, 10, 10] = [1e7, -1e7]
    pyr = _alt_pyramid(torch, fmaps)
    ii = torch.from_numpy(prob.ii[sel]).cuda()
    jj = torch.from_numpy(prob.jj[sel]).cuda()
    _check_alt_levels(backends, oracle, torch, pyr, torch.from_numpy(c).cuda(), ii, jj, 3, "cfg3 48x64 C128 r3")
    _check_alt_half_pyramid(backends, oracle, torch, _alt_pyramid(torch, fmaps, half=True), torch.from_numpy(c).cuda(),
                            ii, jj, 3, "cfg3 48x64 C128 r3 half pyramid")


def test_cfg5_altcorr_96x128_r4_stereo_indexing(backends, oracle, synth):
    """s_droid_backend path: fmaps [num, rig=2, ...] viewed as [1, num*rig, ...]; frame indices
    rig*ii and rig*jj + (ii == jj) (factor_graph.py:258, :277)."""
    torch = _torch()
    prob = synth.make_ba_problem(N=10, E=48, H=96, W=128, stereo=True, lm=1e-5, ep=1e-2, seed=4, radius=4)
    rng = np.random.default_rng(55)
    rig = 2
    fmaps = rng.normal(0, 1, (prob.t1 * rig, 128, 96, 128)).astype(np.float16)
    _, coords = synth.make_corr_inputs(prob, C=1, seed=4)
    sel = np.array([0, 3, 12, 47])                    # two stereo (ii == jj) edges, one neighbour, one long-range
    assert (prob.ii[sel] == prob.jj[sel]).sum() == 2
    iis = torch.from_numpy(prob.ii[sel]).cuda()
    jjs = torch.from_numpy(prob.jj[sel]).cuda()
    i1 = rig * iis
    i2 = rig * jjs + (iis == jjs).long()
    pyr = _alt_pyramid(torch, fmaps)
    _check_alt_levels(backends, oracle, torch, pyr, torch.from_numpy(coords[sel].copy()).cuda(), i1, i2, 4,
                      "cfg5 96x128 C128 r4")
    _check_alt_half_pyramid(backends, oracle, torch, _alt_pyramid(torch, fmaps, half=True),
                            torch.from_numpy(coords[sel].copy()).cuda(), i1, i2, 4, "cfg5 96x128 C128 r4 half pyramid")


def test_altcorr_half_subnormal_inputs_are_not_flushed(backends, oracle):
    """fp16 subnormals (|x| < 6.1e-5) on the f16 matrix cores: the products must come through like in the
    reference's fp32 evaluation of the widened maps.  fmap1 carries subnormal halves (k * 2^-24), fmap2 values
    around 2^10, so every dot product is O(1) if and only if the subnormal operands are kept."""
    torch = _torch()
    rng = np.random.default_rng(91)
    B, H, W, C, r = 1, 8, 16, 64, 3
    f1 = (rng.integers(1, 1000, (B, H, W, C)) * 2.0 ** -24).astype(np.float16)
    assert (np.abs(f1.astype(np.float64)) < 6.104e-5).all() and (f1 != 0).all()
    f2 = (rng.choice([-1.0, 1.0], (B, H, W, C)) * rng.integers(512, 2048, (B, H, W, C))).astype(np.float16)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    coords = np.stack([xx + 0.25, yy + 0.5], -1)[None, None].astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out, = backends.altcorr_forward(t(f1), t(f2), t(coords), r)
    ref = oracle.altcorr_forward(f1.astype(np.float32), f2.astype(np.float32), coords, r, acc_dtype=np.float64)
    got = out.float().cpu().numpy()
    scale = np.abs(ref).max()
    assert scale > 0.05
    assert np.abs(got - ref).max() / scale < 2e-3          # half output: a half ulp of the scale
    # and through the fp32-output entry point (two single-level pyramids: f1 is level 0 of frame 0, f2 of frame 1)
    pyr = [torch.cat([t(f1), t(f2)], 0)]
    fused, = backends.altcorr_pyramid_forward(pyr, t(coords[0]), torch.tensor([0], device="cuda"),
                                              torch.tensor([1], device="cuda"), r)
    assert np.abs(fused.cpu().numpy() - ref[:, 0]).max() / scale < 1e-5


@pytest.mark.parametrize("lvl", [0, 2])
def test_altcorr_forward_fp16(backends, oracle, lvl):
    """Half dispatch (altcorr_kernel.cu:308).  The reference accumulates the 32-channel chunks and the
    running output in half (:98-142); this kernel accumulates each tap's dot product over all channels in
    fp32 and rounds it to half once, so it is at least as close to the exact value as the reference's own
    arithmetic: checked against the fp64 truth within half precision of the output
    scale, and against the half restatement within the restatement's own distance from the truth."""
    torch = _torch()
    rng = np.random.default_rng(80 + lvl)
    B, H, W, C = 2, 24, 32, 128
    H2, W2 = H >> lvl, W >> lvl
    f1 = (rng.normal(0, 1, (B, H, W, C)) / 4).astype(np.float16)
    f2 = (rng.normal(0, 1, (B, H2, W2, C)) / 4).astype(np.float16)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    c = np.stack([xx[None] + rng.uniform(-2, 2, (B, H, W)), yy[None] + rng.uniform(-2, 2, (B, H, W))], -1)
    coords = (c / (1 << lvl))[:, None].astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out, = backends.altcorr_forward(t(f1), t(f2), t(coords), 3)
    assert out.dtype == torch.float16 and tuple(out.shape) == (B, 1, 49, H, W)
    got = out.float().cpu().numpy().astype(np.float64)
    truth = oracle.altcorr_forward(f1, f2, coords, 3, acc_dtype=np.float64)
    half = oracle.altcorr_forward(f1, f2, coords, 3).astype(np.float64)   # reference arithmetic in half
    scale = np.abs(truth).max()
    e_hip, e_ref = np.abs(got - truth).max() / scale, np.abs(half - truth).max() / scale
    print(f"fp16 altcorr lvl{lvl}: hip vs truth {e_hip:.2e}, half restatement vs truth {e_ref:.2e}")
    assert e_hip < 2e-3                      # a few half ulps of the output scale
    assert e_hip <= 1.5 * e_ref + 1e-3
    assert np.abs(got - half).max() / scale <= 2 * e_ref + 1e-3


@pytest.mark.parametrize("C,r,jitter", [(32, 3, 0.5), (96, 3, 6.0), (96, 4, 1.5), (64, 4, 40.0)])
def test_altcorr_half_wave_kernel_channel_counts_edges_and_diverging_windows(backends, oracle, C, r, jitter):
    """The f16 wave kernel away from the benchmark shape: 1, 2 and 3 k-steps (C = 32, 64, 96), an image that is no multiple of
    the 4x4 query tile (10 x 13: masked queries), windows that diverge inside a tile (larger bounding boxes, up to the
    per-query fallback for incoherent coordinates at jitter 40) and radius 4; fp32 output against the fp64 oracle on the same
    half inputs."""
    torch = _torch()
    rng = np.random.default_rng(7 * C + r)
    F, H, W = 3, 10, 13
    fm = (rng.normal(0, 1, (F, H, W, C)) / 4).astype(np.float16)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    ii = np.array([0, 2, 1, 1], dtype=np.int64)
    jj = np.array([1, 0, 1, 2], dtype=np.int64)
    E = len(ii)
    coords = np.stack([xx[None] + rng.uniform(-jitter, jitter, (E, H, W)), yy[None] + rng.uniform(-jitter, jitter, (E, H, W))],
                      -1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    fused, = backends.altcorr_pyramid_forward([t(fm)], t(coords), t(ii), t(jj), r)
    assert fused.dtype == torch.float32 and tuple(fused.shape) == (E, (2 * r + 1) ** 2, H, W)
    ref = oracle.altcorr_forward(fm[ii].astype(np.float64), fm[jj].astype(np.float64), coords[:, None], r,
                                 acc_dtype=np.float64, chunked=False)[:, 0]
    scale = np.abs(ref).max()
    err = np.abs(fused.cpu().numpy() - ref).max() / scale
    print(f"[half wave kernel C={C} r={r} jitter={jitter}] rel err {err:.2e} (scale {scale:.2f})")
    assert err < 1e-5
