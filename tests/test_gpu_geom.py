"""GPU parity of frame_distance / projmap / iproj against the fp64 oracle."""
import numpy as np
import pytest

from util import to_dev

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def prob():
    from droid_backends import synth
    return synth.make_config("cfg1")


def test_frame_distance(backends, oracle, prob):
    torch = _torch()
    d = to_dev(prob, torch)
    for beta in (0.3, 0.0, 1.0):
        got = backends.frame_distance(d["poses"], d["disps"], d["intrinsics"], d["ii"], d["jj"], beta).cpu().numpy()
        ref = oracle.frame_distance(prob.poses, prob.disps, prob.intrinsics, prob.ii, prob.jj, beta)
        assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_frame_distance_flags_low_overlap(backends, oracle, prob):
    torch = _torch()
    d = to_dev(prob, torch)
    poses = prob.poses.copy()
    poses[3, :3] += np.array([0, 0, -8.0], np.float32)  # push frame 3 far behind
    got = backends.frame_distance(torch.from_numpy(poses).cuda(), d["disps"], d["intrinsics"], d["ii"], d["jj"], 0.3)
    ref = oracle.frame_distance(poses, prob.disps, prob.intrinsics, prob.ii, prob.jj, 0.3)
    assert (ref == 1000.0).any()
    assert np.array_equal(got.cpu().numpy() == 1000.0, ref == 1000.0)


def test_projmap(backends, oracle, prob):
    torch = _torch()
    d = to_dev(prob, torch)
    coords, valid = backends.projmap(d["poses"], d["disps"], d["intrinsics"], d["ii"], d["jj"])
    rc, rv = oracle.projmap(prob.poses, prob.disps, prob.intrinsics, prob.ii, prob.jj)
    assert np.abs(coords.cpu().numpy() - rc).max() < 2e-3  # pixels, fp32 projection
    assert np.mean(valid.cpu().numpy() != rv) < 1e-4


def test_iproj(backends, oracle, prob):
    torch = _torch()
    d = to_dev(prob, torch)
    pts = backends.iproj(d["poses"], d["disps"], d["intrinsics"]).cpu().numpy()
    ref = oracle.iproj(prob.poses, prob.disps, prob.intrinsics)
    assert np.abs(pts - ref).max() < 1e-4 * np.abs(ref).max()


def test_depth_filter_counts_consistent_neighbours(backends, prob):
    """With GT poses/disparities every interior pixel agrees with its temporal neighbours."""
    torch = _torch()
    poses = torch.from_numpy(prob.gt_poses.astype(np.float32)).cuda()
    disps = torch.from_numpy(prob.gt_disps.astype(np.float32)).cuda()
    intr = torch.from_numpy(prob.intrinsics).cuda()
    ix = torch.tensor([3, 4], dtype=torch.int64, device="cuda")
    thresh = torch.full((2,), 0.2, dtype=torch.float32, device="cuda")
    cnt = backends.depth_filter(poses, disps, intr, ix, thresh).cpu().numpy()
    assert cnt.shape == (2, 48, 64)
    assert cnt.max() <= 6 and cnt[:, 8:-8, 8:-8].mean() > 3.0


def _reproject_case(prob, rng, per_frame_K):
    poses = prob.poses.copy()
    poses[5, :3] += np.array([0, 0, -6.0], np.float32)      # some edges end behind the camera
    nb = prob.disps.shape[0]
    st = np.array([2, nb - 1, nb // 2])
    ii = np.concatenate([prob.ii, st])                       # + three stereo edges (ii == jj)
    jj = np.concatenate([prob.jj, st])
    K = prob.intrinsics.astype(np.float32)
    if per_frame_K:
        K = np.stack([K * np.float32(1.0 + 0.01 * (f % 7)) for f in range(prob.disps.shape[0])]).astype(np.float32)
    H, W = prob.disps.shape[1:]
    target = rng.uniform(-20, 90, (len(ii), H, W, 2)).astype(np.float32)
    return poses, ii, jj, K, target


@pytest.mark.parametrize("per_frame_K", [False, True])
def test_reproject_and_motion_features(backends, prob, per_frame_K):
    """Fused DepthVideo.reproject + motion features against the numpy restatement: coordinates
    within 2e-3 px (fp32 projection), validity identical except on the depth threshold, features clamped to 64."""
    torch = _torch()
    from oracle import geom
    rng = np.random.default_rng(11)
    poses, ii, jj, K, target = _reproject_case(prob, rng, per_frame_K)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    c, v, m = backends.reproject(t(poses), t(prob.disps), t(K), t(ii), t(jj), t(target))
    rm, rc, rv = geom.motion_features(poses, prob.disps, K, ii, jj, target)
    c, v, m = c.cpu().numpy()[0], v.cpu().numpy()[0], m.cpu().numpy()[0]
    assert c.shape == rc.shape and v.shape == rv.shape and m.shape == rm.shape
    ok = np.abs(rc).max(axis=-1) < 1e4                       # pixels near the depth clamp blow up in both
    assert np.abs(c - rc)[ok].max() < 2e-3 * max(1.0, np.abs(rc[ok]).max() / 100)
    assert np.mean(v != rv) < 1e-4
    assert np.abs(m - rm).max() < 5e-3 and np.abs(m).max() <= 64.0
    assert (rv == 0).any() and (np.abs(rm) == 64.0).any()    # the case exercises both branches
    # plain reproject (no target) returns the same coordinates; motion_features() orders (motn, coords, mask)
    c2, v2 = backends.reproject(t(poses)[None], t(prob.disps)[None], t(K)[None] if per_frame_K else t(K), t(ii), t(jj))
    assert torch.equal(c2.cpu(), torch.from_numpy(c)[None]) and torch.equal(v2.cpu(), torch.from_numpy(v)[None])
    m3, c3, _ = backends.motion_features(t(poses), t(prob.disps), t(K), t(ii), t(jj), t(target)[None])
    assert torch.equal(m3.cpu(), torch.from_numpy(m)[None]) and torch.equal(c3.cpu(), torch.from_numpy(c)[None])


def test_reproject_rejects_cpu_tensors_and_bad_indices(backends, prob):
    torch = _torch()
    d = to_dev(prob, torch)
    with pytest.raises(RuntimeError):
        backends.reproject(torch.from_numpy(prob.poses), d["disps"], d["intrinsics"], d["ii"], d["jj"])
    ii = d["ii"].clone()
    ii[0] = 10 ** 6
    c, v = backends.reproject(d["poses"], d["disps"], d["intrinsics"], ii, d["jj"])
    assert float(c[0, 0].abs().max()) == 0.0 and float(v[0, 0].max()) == 0.0   # out-of-range edge: zeros, no fault


def test_depth_filter_matches_oracle(backends, prob):
    """depth_filter counts (droid_kernels.cu:661-775) against the numpy restatement: perturbed poses, frames at
    both ends of the buffer (missing neighbours), an index outside the buffer.  The reference compares in
    double precision, the kernel in fp32: pixels whose error sits on the threshold may differ by one count."""
    torch = _torch()
    from oracle import geom
    d = to_dev(prob, torch)
    nb = prob.disps.shape[0]
    ix = np.array([0, 2, nb // 2, nb - 1, nb + 3], dtype=np.int64)
    thresh = np.array([0.05, 0.1, 0.2, 0.02, 0.1], dtype=np.float32)
    got = backends.depth_filter(d["poses"], d["disps"], d["intrinsics"], torch.from_numpy(ix).cuda(),
                                torch.from_numpy(thresh).cuda()).cpu().numpy()
    ref = geom.depth_filter(prob.poses, prob.disps, prob.intrinsics, ix, thresh)
    assert got.shape == ref.shape and got[-1].max() == 0 and ref[-1].max() == 0
    assert np.abs(got - ref).max() <= 1.0 and np.mean(got != ref) < 2e-3
    assert ref[:4].max() >= 3 and (ref[:4] == 0).any()      # the case is not trivial


def test_geom_golden_vectors_on_device(backends):
    """The committed fixtures (tests/golden/geom_golden.npz, altcorr_backward_golden.npz) through the HIP path."""
    import os
    torch = _torch()
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "geom_golden.npz"), allow_pickle=False)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    c, v, m = backends.reproject(t(g["poses"]), t(g["disps"]), t(g["intrinsics"]), t(g["ii"]), t(g["jj"]), t(g["target"]))
    ok = np.abs(g["coords"]).max(axis=-1) < 1e4
    assert np.abs(c.cpu().numpy()[0] - g["coords"])[ok].max() < 2e-3
    assert np.mean(v.cpu().numpy()[0] != g["valid"]) < 1e-3 and np.abs(m.cpu().numpy()[0] - g["motn"]).max() < 5e-3
    cnt = backends.depth_filter(t(g["df_poses"]), t(g["disps"]), t(g["df_intrinsics"]), t(g["df_ix"]), t(g["df_thresh"]))
    assert np.abs(cnt.cpu().numpy() - g["df_counter"]).max() <= 1.0 and np.mean(cnt.cpu().numpy() != g["df_counter"]) < 5e-3
    a = np.load(os.path.join(gold, "altcorr_backward_golden.npz"), allow_pickle=False)
    g1, g2, _ = backends.altcorr_backward(t(a["fmap1"]), t(a["fmap2"]), t(a["coords"]), t(a["corr_grad"]), 3)
    assert np.abs(g1.cpu().numpy() - a["fmap1_grad"]).max() < 2e-5 * np.abs(a["fmap1_grad"]).max()
    assert np.abs(g2.cpu().numpy() - a["fmap2_grad"]).max() < 2e-5 * np.abs(a["fmap2_grad"]).max()
