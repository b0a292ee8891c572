"""The reference's on-disk keyframe dump (droid.py:92-106 / loop_detect.py:209-220): write / read round trip,
layout checks, and the hand-over of the BA-side arrays."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "droid-slam_reserch_amd"))


def _dump(t=5, ht=64, wd=96, stereo=False, seed=0):
    from droid_backends import keyframes as kf
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(t, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses = np.concatenate([rng.normal(size=(t, 3)), q], axis=1).astype(np.float32)
    c = 2 if stereo else 1
    return kf.KeyframeDump(
        tstamps=np.arange(t, dtype=np.float32) * 3, images=rng.integers(0, 255, (t, 3, ht, wd), dtype=np.uint8),
        disps=rng.uniform(0.1, 2, (t, ht // 8, wd // 8)).astype(np.float32), poses=poses,
        intrinsics=np.tile(np.array([40, 40, 6, 4], np.float32), (t, 1)),
        fmaps=rng.normal(size=(t, c, 128, ht // 8, wd // 8)).astype(np.float16),
        inps=rng.normal(size=(t, 128, ht // 8, wd // 8)).astype(np.float16),
        nets=rng.normal(size=(t, 128, ht // 8, wd // 8)).astype(np.float16))


@pytest.mark.parametrize("stereo", [False, True])
def test_round_trip_is_bit_exact(tmp_path, stereo):
    from droid_backends import keyframes as kf
    d = _dump(stereo=stereo)
    d.backend_finished_poses = d.poses.copy()
    kf.save(str(tmp_path / "rec"), d)
    assert sorted(os.listdir(tmp_path / "rec")) == sorted([f + ".npy" for f in kf.FIELDS] + ["backend_finished_poses.npy"])
    for mmap in (True, False):
        r = kf.load(str(tmp_path / "rec"), mmap=mmap)
        assert r.count == 5 and r.stereo == stereo
        for f in kf.FIELDS + ("backend_finished_poses",):
            a, b = getattr(d, f), np.asarray(getattr(r, f))
            assert a.dtype == b.dtype and np.array_equal(a, b), f
    # the files are plain numpy arrays, readable the way loop_detect.py reads them
    assert np.load(tmp_path / "rec" / "fmaps.npy").shape == d.fmaps.shape


def test_empty_dump_and_missing_file(tmp_path):
    from droid_backends import keyframes as kf
    kf.save(str(tmp_path / "empty"), _dump(t=0))
    assert kf.load(str(tmp_path / "empty")).count == 0
    os.remove(tmp_path / "empty" / "nets.npy")
    with pytest.raises(kf.KeyframeDumpError):
        kf.load(str(tmp_path / "empty"))


def test_validation_rejects_inconsistent_arrays():
    from droid_backends import keyframes as kf
    d = _dump()
    d.disps = d.disps[:, :-1]
    with pytest.raises(kf.KeyframeDumpError):
        d.validate()
    d = _dump()
    d.poses = d.poses.astype(np.float64)
    with pytest.raises(kf.KeyframeDumpError):
        d.validate()
    d = _dump()
    d.poses[2, 3:] *= 2
    with pytest.raises(kf.KeyframeDumpError):
        d.validate()
    d = _dump()
    d.nets = d.nets[:-1]
    with pytest.raises(kf.KeyframeDumpError):
        d.validate()


def test_to_device_pads_like_the_video_buffer():
    from droid_backends import keyframes as kf
    d = _dump()
    s = d.to_device(device="cpu", buffer=12)
    assert s["count"] == 5 and tuple(s["poses"].shape) == (12, 7) and tuple(s["disps"].shape) == (12, 8, 12)
    assert np.array_equal(s["poses"][:5].numpy(), d.poses) and np.array_equal(s["disps"][:5].numpy(), d.disps)
    assert np.array_equal(s["poses"][5:].numpy(), np.tile(np.array([0, 0, 0, 0, 0, 0, 1], np.float32), (7, 1)))
    assert float(s["disps"][5:].min()) == 1.0 and tuple(s["intrinsics"].shape) == (12, 4)


@pytest.mark.gpu
def test_dump_feeds_the_hip_operators(tmp_path, backends):
    """A dump written from a synthetic video, read back and handed to the device gives the same frame distances
    and reprojections as the arrays it was written from."""
    import torch
    from droid_backends import keyframes as kf, synth
    p = synth.make_config("cfg1")
    t, (H, W) = p.disps.shape[0], p.disps.shape[1:]
    rng = np.random.default_rng(2)
    d = kf.KeyframeDump(
        tstamps=np.arange(t, dtype=np.float32), images=rng.integers(0, 255, (t, 3, 8 * H, 8 * W), dtype=np.uint8),
        disps=p.disps, poses=p.poses, intrinsics=np.tile(p.intrinsics, (t, 1)).astype(np.float32),
        fmaps=np.zeros((t, 1, 128, H, W), np.float16), inps=np.zeros((t, 128, H, W), np.float16),
        nets=np.zeros((t, 128, H, W), np.float16))
    kf.save(str(tmp_path / "rec"), d)
    s = kf.load(str(tmp_path / "rec")).to_device("cuda", buffer=t + 4)
    ii, jj = torch.from_numpy(p.ii).cuda(), torch.from_numpy(p.jj).cuda()
    ref_p, ref_d, ref_k = torch.from_numpy(p.poses).cuda(), torch.from_numpy(p.disps).cuda(), torch.from_numpy(p.intrinsics).cuda()
    a = backends.frame_distance(s["poses"], s["disps"], s["intrinsics"][0].contiguous(), ii, jj, 0.3)
    b = backends.frame_distance(ref_p, ref_d, ref_k, ii, jj, 0.3)
    assert torch.equal(a, b)
    ca, va = backends.reproject(s["poses"], s["disps"], s["intrinsics"], ii, jj)
    cb, vb = backends.reproject(ref_p, ref_d, ref_k, ii, jj)
    assert torch.equal(ca, cb) and torch.equal(va, vb)
