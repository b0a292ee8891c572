"""CPU tests of oracle/geom.py (numpy restatement of DepthVideo.reproject + the motion features): closed-form
cases, since lietorch (the reference's implementation of the SE3 action) is not importable here."""
import numpy as np

from oracle import geom


def _scene(rng, N=5, H=12, W=16):
    poses = np.zeros((N, 7)); poses[:, 6] = 1.0
    disps = rng.uniform(0.3, 1.5, (N, H, W))
    K = np.array([20.0, 22.0, W / 2 - 0.5, H / 2 - 0.5])
    return poses, disps, K


def test_identity_poses_reproject_to_the_grid():
    rng = np.random.default_rng(0)
    poses, disps, K = _scene(rng)
    ii, jj = np.array([0, 1, 2]), np.array([1, 2, 0])
    coords, valid = geom.reproject(poses, disps, K, ii, jj)
    y, x = np.meshgrid(np.arange(12.0), np.arange(16.0), indexing="ij")
    assert np.abs(coords[..., 0] - x).max() < 1e-12 and np.abs(coords[..., 1] - y).max() < 1e-12
    assert valid.min() == 1.0


def test_pure_translation_shifts_by_fx_tx_disp():
    """Gij = Gj * Gi^-1 with translations only: X1 = X0 + (tj - ti) * disp, Z stays 1 => u' = u + fx * dt_x * disp."""
    rng = np.random.default_rng(1)
    poses, disps, K = _scene(rng)
    poses[1, 0] = 0.25   # world-to-camera translation of frame 1
    coords, _ = geom.reproject(poses, disps, K, np.array([0]), np.array([1]))
    y, x = np.meshgrid(np.arange(12.0), np.arange(16.0), indexing="ij")
    assert np.abs(coords[0, ..., 0] - (x + K[0] * 0.25 * disps[0])).max() < 1e-12
    assert np.abs(coords[0, ..., 1] - y).max() < 1e-12


def test_stereo_edge_uses_fixed_baseline():
    rng = np.random.default_rng(2)
    poses, disps, K = _scene(rng)
    poses[2, :3] = [3.0, -1.0, 0.5]   # must not matter for ii == jj
    coords, _ = geom.reproject(poses, disps, K, np.array([2]), np.array([2]))
    y, x = np.meshgrid(np.arange(12.0), np.arange(16.0), indexing="ij")
    assert np.abs(coords[0, ..., 0] - (x - K[0] * 0.1 * disps[2])).max() < 1e-12


def test_rotation_about_the_optical_axis_rotates_the_image():
    rng = np.random.default_rng(3)
    poses, disps, _ = _scene(rng)
    K = np.array([20.0, 20.0, 7.5, 5.5])
    a = 0.3
    poses[1, 3:] = [0.0, 0.0, np.sin(a / 2), np.cos(a / 2)]
    coords, _ = geom.reproject(poses, disps, K, np.array([0]), np.array([1]))
    y, x = np.meshgrid(np.arange(12.0), np.arange(16.0), indexing="ij")
    xr = np.cos(a) * (x - 7.5) - np.sin(a) * (y - 5.5) + 7.5
    yr = np.sin(a) * (x - 7.5) + np.cos(a) * (y - 5.5) + 5.5
    assert np.abs(coords[0, ..., 0] - xr).max() < 1e-10 and np.abs(coords[0, ..., 1] - yr).max() < 1e-10


def test_points_behind_the_camera_project_with_depth_one_and_are_invalid():
    rng = np.random.default_rng(4)
    poses, disps, K = _scene(rng)
    poses[1, 2] = -5.0   # Z1 = 1 - 5 * disp < 0.1 everywhere
    coords, valid = geom.reproject(poses, disps, K, np.array([0]), np.array([1]))
    y, x = np.meshgrid(np.arange(12.0), np.arange(16.0), indexing="ij")
    assert valid.max() == 0.0
    assert np.abs(coords[0, ..., 0] - x).max() < 1e-12   # X unchanged, divided by 1


def test_per_frame_intrinsics_and_motion_feature_layout():
    rng = np.random.default_rng(5)
    poses, disps, K = _scene(rng)
    Ks = np.stack([K * (1.0 + 0.05 * f) for f in range(5)])
    ii, jj = np.array([0, 3]), np.array([3, 1])
    c, _ = geom.reproject(poses, disps, Ks, ii, jj)
    y, x = np.meshgrid(np.arange(12.0), np.arange(16.0), indexing="ij")
    # identity poses: u' = fx_j * (u - cx_i) / fx_i + cx_j
    assert np.abs(c[0, ..., 0] - (Ks[3, 0] * (x - Ks[0, 2]) / Ks[0, 0] + Ks[3, 2])).max() < 1e-12
    target = c + rng.normal(0, 100.0, c.shape)
    motn, c2, _ = geom.motion_features(poses, disps, Ks, ii, jj, target)
    assert motn.shape == (2, 4, 12, 16) and np.array_equal(c, c2)
    assert motn.max() <= 64.0 and motn.min() >= -64.0
    assert np.allclose(motn[:, 0], np.clip(c[..., 0] - x, -64, 64)) and np.allclose(motn[:, 3], np.clip((target - c)[..., 1], -64, 64))


def test_depth_filter_counts_existing_neighbours_of_a_static_scene():
    """Identity poses, constant disparity: every neighbour that exists agrees, except where the 2x2 corner of the
    (unchanged) projection leaves the image (last row / column, droid_kernels.cu:748)."""
    poses = np.zeros((9, 7)); poses[:, 6] = 1.0
    disps = np.full((9, 10, 12), 0.7)
    K = np.array([20.0, 20.0, 5.5, 4.5])
    ix = np.array([0, 4, 8, 12])
    cnt = geom.depth_filter(poses, disps, K, ix, np.full(4, 0.01))
    # neighbours ix-1, ix-2, ix-3, ix+3, ix+4, ix+5 inside [0, 9)
    assert np.all(cnt[0, :-1, :-1] == 3) and np.all(cnt[1, :-1, :-1] == 5) and np.all(cnt[2, :-1, :-1] == 3)
    assert np.all(cnt[:, -1, :] == 0) and np.all(cnt[:, :, -1] == 0) and np.all(cnt[3] == 0)
    disps[5] = 1.4          # frame 5 disagrees: 1/0.7 - 1/1.4 = 0.71 > threshold (neighbour ix-3 of frame 8)
    assert np.all(geom.depth_filter(poses, disps, K, np.array([8]), np.array([0.01]))[0, :-1, :-1] == 2)


def test_geom_golden_vectors():
    """Regression pin: committed outputs of oracle/geom.py (tests/golden/make_golden.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "geom_golden.npz"), allow_pickle=False)
    motn, coords, valid = geom.motion_features(g["poses"], g["disps"], g["intrinsics"], g["ii"], g["jj"], g["target"])
    assert np.array_equal(valid, g["valid"])
    assert np.abs(coords - g["coords"]).max() < 1e-9 * max(1.0, np.abs(g["coords"]).max())
    assert np.abs(motn - g["motn"]).max() < 1e-9
    cnt = geom.depth_filter(g["df_poses"], g["disps"], g["df_intrinsics"], g["df_ix"], g["df_thresh"])
    assert np.array_equal(cnt, g["df_counter"])
