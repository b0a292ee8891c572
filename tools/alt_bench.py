"""Alt-corr timing per level and fused, fp32 pyramid and the half pyramid of the SLAM path.
usage: python tools/alt_bench.py [edges] (on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "droid-slam_reserch_amd"))
import numpy as np, torch, torch.nn.functional as F
import droid_backends as db
from droid_backends import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, W, r = 48, 64, 3
prob = synth.make_config("cfg2")
fmaps, coords = synth.make_corr_inputs(prob, n_edges=B, seed=0)
dev = "cuda"
ii = torch.from_numpy(prob.ii[:B]).to(dev); jj = torch.from_numpy(prob.jj[:B]).to(dev)
c = torch.from_numpy(coords).to(dev)


def pyramid(half):
    x = torch.from_numpy(fmaps).to(dev)
    x = (x if half else x.float()) / 4.0          # modules/corr.py:97 (half when video.fmaps is half)
    pyr = []
    for l in range(4):
        pyr.append(x.permute(0, 2, 3, 1).contiguous()); x = F.avg_pool2d(x, 2, stride=2)
    return pyr


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ca = [(c[:, None] / 2 ** l).contiguous() for l in range(4)]
for half in (False, True):
    pyr = pyramid(half)
    a1 = pyr[0][ii].contiguous()
    a2 = [pyr[l][jj].contiguous() for l in range(4)]
    tot = 0.0
    for l in range(4):
        ms = timeit(lambda: db.altcorr_forward(a1, a2[l], ca[l], r))
        tot += ms
        fl = B * H * W * 64 * 128 * 2
        print(f"{'f16' if half else 'f32'} level {l}: {ms*1e3:.0f} us  {fl/ms/1e9:.1f} TFLOP/s useful")
    ms = timeit(lambda: db.altcorr_pyramid_forward(pyr, c, ii, jj, r))
    print(f"{'f16' if half else 'f32'} per-level sum {tot*1e3:.0f} us = {B*H*W/tot/1e6:.3f} Gpix/s; "
          f"fused pyramid {ms*1e3:.0f} us = {B*H*W/ms/1e6:.3f} Gpix/s")
