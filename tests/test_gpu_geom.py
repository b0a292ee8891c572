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
