"""GPU parity of the correlation lookups against oracle/corr.py (numpy restatement)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _volume_inputs(B, H, W, lvl, dtype, seed, spread=1.5):
    rng = np.random.default_rng(seed)
    H2, W2 = H >> lvl, W >> lvl
    vol = rng.normal(0, 1, (B, H, W, H2, W2)).astype(dtype)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    cx = (xx[None] + rng.uniform(-4, 4, (B, 1, 1)) + rng.uniform(-spread, spread, (B, H, W))) / (1 << lvl)
    cy = (yy[None] + rng.uniform(-4, 4, (B, 1, 1)) + rng.uniform(-spread, spread, (B, H, W))) / (1 << lvl)
    coords = np.stack([cx, cy], 1).astype(np.float32)
    return vol, coords


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
@pytest.mark.parametrize("lvl", [0, 1, 2, 3])
def test_corr_index_forward_bit_exact(backends, oracle, dtype, lvl):
    """f16/f32: same rounding points as the reference kernel => bit-exact against the oracle."""
    torch = _torch()
    vol, coords = _volume_inputs(3, 24, 32, lvl, dtype, seed=lvl)
    ref = oracle.corr_index_forward(vol, coords, 3)
    out, = backends.corr_index_forward(torch.from_numpy(vol).cuda(), torch.from_numpy(coords).cuda(), 3)
    got = out.cpu().numpy()
    assert got.shape == ref.shape and got.dtype == ref.dtype
    bad = np.sum(got.view(np.uint16 if dtype == np.float16 else np.uint32)
                 != ref.view(np.uint16 if dtype == np.float16 else np.uint32))
    # +0/-0 are the only tolerated bit differences
    assert np.array_equal(got, ref), f"{bad} differing elements, max abs {np.abs(got.astype(np.float64) - ref).max()}"


@pytest.mark.parametrize("radius", [1, 3, 4])
def test_corr_index_forward_radius_and_f64(backends, oracle, radius):
    torch = _torch()
    vol, coords = _volume_inputs(2, 16, 20, 0, np.float64, seed=7)
    ref = oracle.corr_index_forward(vol, coords, radius)
    out, = backends.corr_index_forward(torch.from_numpy(vol).cuda(), torch.from_numpy(coords).cuda(), radius)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_corr_index_forward_out_of_range_coords(backends, oracle):
    """Windows partly / fully outside the plane, negative coordinates, huge values."""
    torch = _torch()
    vol, coords = _volume_inputs(2, 12, 16, 0, np.float32, seed=3)
    coords[0, 0, :, :4] = -2.25
    coords[0, 1, :3, :] = -7.5
    coords[1, 0, 5, 5] = 1e9
    coords[1, 1, 6, 6] = -1e9
    coords[1, :, 7, 7] = 15.999
    ref = oracle.corr_index_forward(vol, coords, 3)
    out, = backends.corr_index_forward(torch.from_numpy(vol).cuda(), torch.from_numpy(coords).cuda(), 3)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_corr_index_forward_empty_batch(backends):
    torch = _torch()
    vol = torch.zeros((0, 8, 8, 8, 8), dtype=torch.float16, device="cuda")
    coords = torch.zeros((0, 2, 8, 8), dtype=torch.float32, device="cuda")
    out, = backends.corr_index_forward(vol, coords, 3)
    assert tuple(out.shape) == (0, 7, 7, 8, 8)


def test_corr_index_backward_matches_autograd_of_forward(backends, oracle):
    """corr is linear in the volume: <corr(V), G> = <V, backward(G)>; check against a dense
    Jacobian-free identity using two random probes, plus the sparsity pattern."""
    torch = _torch()
    vol, coords = _volume_inputs(2, 8, 10, 0, np.float32, seed=5)
    rng = np.random.default_rng(1)
    G = rng.normal(size=(2, 7, 7, 8, 10)).astype(np.float32)
    dv = torch.from_numpy(vol).cuda()
    dc = torch.from_numpy(coords).cuda()
    gv, = backends.corr_index_backward(dv, dc, torch.from_numpy(G).cuda(), 3)
    gv = gv.cpu().numpy().astype(np.float64)
    for seed in (2, 3):
        V = np.random.default_rng(seed).normal(size=vol.shape).astype(np.float32)
        lhs = np.sum(oracle.corr_index_forward(V.astype(np.float64), coords, 3) * G)
        rhs = np.sum(V.astype(np.float64) * gv)
        assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)


def _alt_inputs(B, H, W, lvl, C, seed):
    rng = np.random.default_rng(seed)
    H2, W2 = H >> lvl, W >> lvl
    f1 = (rng.normal(0, 1, (B, H, W, C)).astype(np.float16) / np.float16(4)).astype(np.float32)
    f2 = (rng.normal(0, 1, (B, H2, W2, C)).astype(np.float16) / np.float16(4)).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    cx = (xx[None] + rng.uniform(-3, 3, (B, 1, 1)) + rng.uniform(-1.5, 1.5, (B, H, W))) / (1 << lvl)
    cy = (yy[None] + rng.uniform(-3, 3, (B, 1, 1)) + rng.uniform(-1.5, 1.5, (B, H, W))) / (1 << lvl)
    coords = np.stack([cx, cy], -1)[:, None].astype(np.float32)  # [B,1,H,W,2]
    return f1, f2, coords


@pytest.mark.parametrize("lvl,C,N,r", [(0, 128, 1, 3), (2, 48, 2, 3), (1, 272, 1, 3), (0, 40, 1, 3), (1, 64, 1, 4)])
def test_altcorr_backward_matches_oracle(backends, oracle, lvl, C, N, r):
    """fmap gradients of altcorr (altcorr_kernel.cu:152-286) against the fp64 restatement (itself checked to be
    the adjoint of the forward, tests/test_oracle_corr.py): fp32 atomics, 2e-5 of the gradient scale; the
    coordinate gradient is identically zero like the reference's.  C = 272 takes the path without the register
    accumulation of the query's own gradient, C = 40 a channel count that is not a multiple of 16."""
    torch = _torch()
    from oracle import corr as oc
    f1, f2, coords = _alt_inputs(2, 12, 16, lvl, C, seed=10 + lvl)
    if N == 2:
        coords = np.concatenate([coords, coords + np.float32(0.37)], axis=1)
    coords[0, 0, 0, :3] = [[-9.0, -9.0], [1e4, 3.0], [2.5, -0.5]]   # windows partly or wholly outside fmap2
    rng = np.random.default_rng(5)
    cg = rng.normal(size=(2, N, (2 * r + 1) ** 2, 12, 16)).astype(np.float32)
    r1, r2 = oc.altcorr_backward(f1, f2, coords, cg, r)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    g1, g2, gc = backends.altcorr_backward(t(f1), t(f2), t(coords), t(cg), r)
    e1 = np.abs(g1.cpu().numpy() - r1).max() / np.abs(r1).max()
    e2 = np.abs(g2.cpu().numpy() - r2).max() / np.abs(r2).max()
    print(f"altcorr backward lvl{lvl} C={C} N={N} r={r}: rel err {e1:.2e} {e2:.2e}")
    assert e1 < 2e-5 and e2 < 2e-5
    assert float(gc.abs().max()) == 0.0


@pytest.mark.parametrize("lvl", [0, 1, 2, 3])
def test_altcorr_forward_matches_oracle(backends, oracle, lvl):
    """fp32, tolerance 1e-5 relative to the output scale (SURVEY.md section 8d parity metric)."""
    torch = _torch()
    f1, f2, coords = _alt_inputs(2, 16, 24, lvl, 128, seed=lvl)
    ref = oracle.altcorr_forward(f1, f2, coords, 3, acc_dtype=np.float64)
    out, = backends.altcorr_forward(torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda(),
                                    torch.from_numpy(coords).cuda(), 3)
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print(f"altcorr lvl{lvl}: rel err {err:.2e}")
    assert err < 1e-5


def test_altcorr_equals_volume_lookup(backends):
    """Cross-check of the two operators on the device: alt-corr == lookup in fmap1^T fmap2."""
    torch = _torch()
    f1, f2, coords = _alt_inputs(2, 12, 16, 0, 64, seed=9)
    d1, d2 = torch.from_numpy(f1).cuda().double(), torch.from_numpy(f2).cuda().double()
    vol = torch.einsum("bhwc,bijc->bhwij", d1, d2).contiguous()
    c = torch.from_numpy(coords).cuda()
    a, = backends.altcorr_forward(d1, d2, c, 3)
    v, = backends.corr_index_forward(vol, c[:, 0].permute(0, 3, 1, 2).contiguous(), 3)
    # altcorr channel = ix*7+iy ; corr_index [B, ix, iy, H, W]
    assert torch.allclose(a[:, 0].view(2, 7, 7, 12, 16), v, rtol=1e-9, atol=1e-9)


def test_altcorr_multi_coordinate_sets_and_radius4(backends, oracle):
    torch = _torch()
    f1, f2, coords = _alt_inputs(1, 10, 12, 0, 32, seed=4)
    coords = np.concatenate([coords, coords + 0.37, coords - 5.2], 1)  # N = 3
    ref = oracle.altcorr_forward(f1, f2, coords, 4, acc_dtype=np.float64)
    out, = backends.altcorr_forward(torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda(),
                                    torch.from_numpy(coords).cuda(), 4)
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("jitter,C,H,W", [(2.6, 128, 16, 32), (10.0, 64, 12, 16), (1.0, 48, 9, 19), (1.5, 256, 8, 16)])
def test_altcorr_diverging_windows_and_channel_counts(backends, oracle, jitter, C, H, W):
    """The matrix-core path sizes its work by the bounding box of the windows of 4x4 query
    sub-tiles: jitter 2.6 px forces the two-round exchange (> 12 position blocks per wave),
    10 px the per-query evaluation of incoherent tiles; C=48 runs three 16-channel stages,
    C=256 is served by the LDS-staged vector kernel.  Ragged H x W leaves partial tiles."""
    torch = _torch()
    rng = np.random.default_rng(int(jitter * 10) + C)
    f1 = rng.normal(0, 1, (2, H, W, C)).astype(np.float32)
    f2 = rng.normal(0, 1, (2, H, W, C)).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    cx = xx[None] + rng.uniform(-jitter, jitter, (2, H, W))
    cy = yy[None] + rng.uniform(-jitter, jitter, (2, H, W))
    coords = np.stack([cx, cy], -1)[:, None].astype(np.float32)
    coords[0, 0, 0, 0] = [-1e9, 3.0]            # far-away coordinates give empty windows (zeros)
    coords[1, 0, H - 1, W - 1] = [1e9, -1e9]
    ref = oracle.altcorr_forward(f1, f2, coords, 3, acc_dtype=np.float64)
    out, = backends.altcorr_forward(torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda(),
                                    torch.from_numpy(coords).cuda(), 3)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("r", [3, 4])
def test_altcorr_pyramid_forward_equals_per_level_calls(backends, oracle, r):
    """Fused AltCorrBlock.corr_fn (modules/corr.py:105-125): one launch over (pyramid, ii, jj) must
    equal torch.cat of altcorr_forward on the gathered per-edge maps (bit for bit: same kernel body,
    coordinates scaled by exact powers of two), and the oracle within 1e-5."""
    import torch.nn.functional as F
    torch = _torch()
    rng = np.random.default_rng(40 + r)
    frames, C, H, W, E = 5, 64, 16, 32, 7
    fm = torch.from_numpy((rng.normal(0, 1, (frames, C, H, W)) / 4).astype(np.float32)).cuda()
    pyramid, x = [], fm
    for _ in range(4):
        pyramid.append(x.permute(0, 2, 3, 1).contiguous()[None])  # [1, frames, h, w, C] like AltCorrBlock.pyramid
        x = F.avg_pool2d(x, 2, stride=2)
    ii = torch.from_numpy(rng.integers(0, frames, E)).cuda()
    jj = torch.from_numpy(rng.integers(0, frames, E)).cuda()
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    c = np.stack([xx[None] + rng.uniform(-4, 4, (E, 1, 1)) + rng.uniform(-1.2, 1.2, (E, H, W)),
                  yy[None] + rng.uniform(-4, 4, (E, 1, 1)) + rng.uniform(-1.2, 1.2, (E, H, W))], -1).astype(np.float32)
    coords = torch.from_numpy(c).cuda()
    fused, = backends.altcorr_pyramid_forward(pyramid, coords, ii, jj, r)
    parts = []
    for l in range(4):
        f1 = pyramid[0][0][ii].contiguous()
        f2 = pyramid[l][0][jj].contiguous()
        cl = (coords / 2 ** l)[:, None].contiguous()
        out, = backends.altcorr_forward(f1, f2, cl, r)
        parts.append(out[:, 0])
        ref = oracle.altcorr_forward(f1.cpu().numpy(), f2.cpu().numpy(), cl.cpu().numpy(), r, acc_dtype=np.float64)
        assert np.abs(out.cpu().numpy() - ref).max() < 1e-5 * np.abs(ref).max()
    per_level = torch.cat(parts, dim=1)
    assert fused.shape == per_level.shape == (E, 4 * (2 * r + 1) ** 2, H, W)
    assert torch.equal(fused, per_level)


def test_corr_golden_vectors_on_device(backends):
    """Committed fixtures (tests/golden/corr_golden.npz): device output vs stored expected output."""
    import os
    torch = _torch()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corr_golden.npz"))
    c = torch.from_numpy(g["coords"]).cuda()
    out16, = backends.corr_index_forward(torch.from_numpy(g["volume"]).cuda(), c, 3)
    assert np.array_equal(out16.cpu().numpy(), g["corr_f16"])
    out32, = backends.corr_index_forward(torch.from_numpy(g["volume"].astype(np.float32)).cuda(), c, 3)
    assert np.array_equal(out32.cpu().numpy(), g["corr_f32"])
    alt, = backends.altcorr_forward(torch.from_numpy(g["fmap1"]).cuda(), torch.from_numpy(g["fmap2"]).cuda(),
                                    torch.from_numpy(g["alt_coords"]).cuda(), 3)
    assert np.abs(alt.cpu().numpy() - g["altcorr"]).max() < 1e-5 * np.abs(g["altcorr"]).max()
