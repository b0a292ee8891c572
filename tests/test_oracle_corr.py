"""Pins oracle/corr.py (numpy restatement of the two correlation lookups) by independent
identities: bilinear sampling of the volume (torch grid_sample), alt-corr == lookup in the
explicit all-pairs volume, zero padding outside the plane, and the committed golden vectors."""
import os

import numpy as np
import torch
import torch.nn.functional as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _inputs(B=2, H=10, W=12, seed=0, dtype=np.float64, lvl=0):
    rng = np.random.default_rng(seed)
    H2, W2 = H >> lvl, W >> lvl
    vol = rng.normal(0, 1, (B, H, W, H2, W2)).astype(dtype)
    coords = np.stack([rng.uniform(-3, W2 + 2, (B, H, W)), rng.uniform(-3, H2 + 2, (B, H, W))], 1).astype(np.float32)
    return vol, coords


def test_corr_index_is_bilinear_sampling_with_zero_padding(oracle):
    """corr[b,a,c,y,x] = bilinear sample of volume[b,y,x] at (x0 - r + a, y0 - r + c)."""
    vol, coords = _inputs()
    r = 3
    out = oracle.corr_index_forward(vol, coords, r)
    B, H, W, H2, W2 = vol.shape
    planes = torch.from_numpy(vol.reshape(B * H * W, 1, H2, W2))
    x0 = torch.from_numpy(coords[:, 0].reshape(-1).astype(np.float64))
    y0 = torch.from_numpy(coords[:, 1].reshape(-1).astype(np.float64))
    for a in range(2 * r + 1):
        for c in range(2 * r + 1):
            gx = (x0 - r + a) / (W2 - 1) * 2 - 1
            gy = (y0 - r + c) / (H2 - 1) * 2 - 1
            grid = torch.stack([gx, gy], -1).view(-1, 1, 1, 2)
            s = F.grid_sample(planes, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
            ref = s.view(B, H, W).numpy()
            assert np.abs(out[:, a, c] - ref).max() < 1e-6


def test_corr_index_integer_coordinates_pick_single_taps(oracle):
    rng = np.random.default_rng(1)
    vol = rng.normal(size=(1, 4, 5, 9, 11))
    coords = np.zeros((1, 2, 4, 5), np.float32)
    coords[0, 0] = 5.0
    coords[0, 1] = 4.0
    out = oracle.corr_index_forward(vol, coords, 3)
    for a in range(7):
        for c in range(7):
            assert np.array_equal(out[0, a, c], vol[0, :, :, 4 - 3 + c, 5 - 3 + a])


def test_corr_index_half_rounding_points(oracle):
    """f16: every product and every partial sum is rounded to half (correlation_kernels.cu:55-65)."""
    vol, coords = _inputs(B=1, H=6, W=7, seed=2, dtype=np.float16)
    out = oracle.corr_index_forward(vol, coords, 3)
    assert out.dtype == np.float16
    ref = oracle.corr_index_forward(vol.astype(np.float64), coords, 3)
    err = np.abs(out.astype(np.float64) - ref)
    assert err.max() < 4 * 2.0 ** -11 * max(1.0, np.abs(ref).max())  # a few half ulps
    assert err.max() > 0  # and it is NOT the double-precision result


def test_altcorr_equals_lookup_in_explicit_volume(oracle):
    rng = np.random.default_rng(3)
    B, H, W, C = 2, 6, 8, 16
    for lvl in (0, 1):
        H2, W2 = H >> lvl, W >> lvl
        f1 = rng.normal(size=(B, H, W, C))
        f2 = rng.normal(size=(B, H2, W2, C))
        coords = np.stack([rng.uniform(-2, W2 + 1, (B, H, W)), rng.uniform(-2, H2 + 1, (B, H, W))], -1).astype(np.float32)
        alt = oracle.altcorr_forward(f1, f2, coords[:, None], 3, acc_dtype=np.float64)
        vol = np.einsum("bhwc,bijc->bhwij", f1, f2)
        ref = oracle.corr_index_forward(vol, np.ascontiguousarray(np.transpose(coords, (0, 3, 1, 2))), 3)
        # altcorr channel = ix*7 + iy == corr_index [ix][iy]
        assert np.abs(alt[:, 0].reshape(B, 7, 7, H, W) - ref).max() < 1e-10


def test_altcorr_float32_chunked_accumulation_is_close_to_truth(oracle):
    rng = np.random.default_rng(4)
    f1 = rng.normal(size=(1, 5, 6, 64)).astype(np.float32)
    f2 = rng.normal(size=(1, 5, 6, 64)).astype(np.float32)
    coords = np.stack([rng.uniform(0, 6, (1, 5, 6)), rng.uniform(0, 5, (1, 5, 6))], -1).astype(np.float32)[:, None]
    a32 = oracle.altcorr_forward(f1, f2, coords, 3)
    a64 = oracle.altcorr_forward(f1, f2, coords, 3, acc_dtype=np.float64)
    assert a32.dtype == np.float32
    assert np.abs(a32 - a64).max() < 1e-5 * np.abs(a64).max()


def test_corr_golden_vectors(oracle):
    g = np.load(os.path.join(GOLD, "corr_golden.npz"), allow_pickle=False)
    assert np.array_equal(oracle.corr_index_forward(g["volume"], g["coords"], 3), g["corr_f16"])
    assert np.array_equal(oracle.corr_index_forward(g["volume"].astype(np.float32), g["coords"], 3), g["corr_f32"])
    alt = oracle.altcorr_forward(g["fmap1"], g["fmap2"], g["alt_coords"], 3, acc_dtype=np.float64)
    assert np.abs(alt - g["altcorr"]).max() < 1e-12


def test_altcorr_backward_is_the_adjoint_of_the_forward():
    """altcorr_forward is linear in each feature map, so <corr_grad, F(d1, f2)> = <g1, d1> and
    <corr_grad, F(f1, d2)> = <g2, d2> for any directions d1, d2 (fp64, includes out-of-range taps)."""
    from oracle import corr as oc
    rng = np.random.default_rng(4)
    B, H, W, C, N, r = 2, 6, 7, 8, 2, 2
    f1, f2 = rng.normal(size=(B, H, W, C)), rng.normal(size=(B, H, W, C))
    coords = rng.uniform(-2.5, max(H, W) + 1.5, (B, N, H, W, 2)).astype(np.float32)
    cg = rng.normal(size=(B, N, (2 * r + 1) ** 2, H, W))
    g1, g2 = oc.altcorr_backward(f1, f2, coords, cg, r)
    for _ in range(3):
        d1, d2 = rng.normal(size=f1.shape), rng.normal(size=f2.shape)
        a1 = np.sum(cg * oc.altcorr_forward(d1, f2, coords, r, acc_dtype=np.float64, chunked=False))
        a2 = np.sum(cg * oc.altcorr_forward(f1, d2, coords, r, acc_dtype=np.float64, chunked=False))
        assert abs(a1 - np.sum(g1 * d1)) < 1e-9 * max(1.0, abs(a1))
        assert abs(a2 - np.sum(g2 * d2)) < 1e-9 * max(1.0, abs(a2))


def test_altcorr_backward_golden_vectors():
    """Regression pin of the gradient restatement (tests/golden/make_golden.py)."""
    from oracle import corr as oc
    g = np.load(os.path.join(GOLD, "altcorr_backward_golden.npz"), allow_pickle=False)
    g1, g2 = oc.altcorr_backward(g["fmap1"], g["fmap2"], g["coords"], g["corr_grad"], 3)
    assert np.abs(g1 - g["fmap1_grad"]).max() < 1e-12 and np.abs(g2 - g["fmap2_grad"]).max() < 1e-12
