"""Generates tests/golden/*.npz from the CPU oracle (the reference cannot run in this pipeline,
SURVEY.md section 8c, so these are regression pins of the restatement, not reference outputs).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "droid-slam_reserch_amd"))

import oracle  # noqa: E402
from droid_backends import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def ba_case(p, its, mo=False):
    o = oracle.ba(p.poses, p.disps, p.intrinsics, p.disps_sens, p.targets, p.weights, p.eta, p.ii, p.jj,
                  p.t0, p.t1, its, p.lm, p.ep, mo)
    return o


def main():
    out = {}
    cases = {
        "tiny": synth.make_ba_problem(N=3, E=4, H=16, W=24, seed=11),
        "cfg1": synth.make_config("cfg1"),
        "cfg1_rgbd": synth.make_config("cfg1", rgbd=True, seed=21),
    }
    for name, p in cases.items():
        o = ba_case(p, 2)
        out[f"{name}_poses"] = o["poses"]
        out[f"{name}_disps"] = o["disps"].astype(np.float64)
        out[f"{name}_dx"] = o["dx"]
    np.savez_compressed(os.path.join(HERE, "ba_golden.npz"), **out)

    # correlation lookups: small inputs + expected outputs
    rng = np.random.default_rng(42)
    vol = rng.normal(0, 1, (2, 6, 8, 6, 8)).astype(np.float16)
    yy, xx = np.meshgrid(np.arange(6, dtype=np.float32), np.arange(8, dtype=np.float32), indexing="ij")
    coords = np.stack([xx[None] + rng.uniform(-2, 2, (2, 6, 8)), yy[None] + rng.uniform(-2, 2, (2, 6, 8))], 1)
    coords = coords.astype(np.float32)
    c16 = oracle.corr_index_forward(vol, coords, 3)
    c32 = oracle.corr_index_forward(vol.astype(np.float32), coords, 3)
    f1 = (rng.normal(0, 1, (2, 6, 8, 32)).astype(np.float16)).astype(np.float32) / 4
    f2 = (rng.normal(0, 1, (2, 6, 8, 32)).astype(np.float16)).astype(np.float32) / 4
    ac = np.ascontiguousarray(np.transpose(coords, (0, 2, 3, 1))[:, None])
    alt = oracle.altcorr_forward(f1, f2, ac, 3, acc_dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "corr_golden.npz"), volume=vol, coords=coords, corr_f16=c16,
                        corr_f32=c32, fmap1=f1, fmap2=f2, alt_coords=ac, altcorr=alt)
    # gradients of the alt-correlation on the same inputs
    cg = rng.normal(0, 1, alt.shape).astype(np.float32)
    from oracle import corr as ocorr, geom as ogeom
    g1, g2 = ocorr.altcorr_backward(f1, f2, ac, cg, 3)
    np.savez_compressed(os.path.join(HERE, "altcorr_backward_golden.npz"), fmap1=f1, fmap2=f2, coords=ac, corr_grad=cg,
                        fmap1_grad=g1, fmap2_grad=g2)

    # reprojection + motion features and depth_filter on a small scene (per-frame intrinsics, one stereo edge,
    # one frame pushed behind the others)
    p = synth.make_ba_problem(N=6, E=14, H=12, W=16, seed=31)
    poses = p.poses.copy()
    poses[4, :3] += np.array([0.0, 0.0, -5.0], np.float32)
    K = np.stack([p.intrinsics * np.float32(1.0 + 0.02 * f) for f in range(p.disps.shape[0])]).astype(np.float32)
    ii = np.concatenate([p.ii, [2]]).astype(np.int64)
    jj = np.concatenate([p.jj, [2]]).astype(np.int64)
    target = rng.uniform(-10, 40, (len(ii), 12, 16, 2)).astype(np.float32)
    motn, coords1, valid = ogeom.motion_features(poses, p.disps, K, ii, jj, target)
    ix = np.array([0, 2, 5], np.int64)
    th = np.array([0.05, 0.2, 0.1], np.float32)
    cnt = ogeom.depth_filter(p.poses, p.disps, p.intrinsics, ix, th)
    np.savez_compressed(os.path.join(HERE, "geom_golden.npz"), poses=poses, disps=p.disps, intrinsics=K, ii=ii, jj=jj,
                        target=target, motn=motn, coords=coords1, valid=valid, df_poses=p.poses,
                        df_intrinsics=p.intrinsics, df_ix=ix, df_thresh=th, df_counter=cnt)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
