"""Timing of altcorr_backward (level 0, 48x64, C=128, 32 edges): tiled LDS kernel vs DROID_ALTCORR_BWD_PER_TAP=1."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_reserch_amd")]
import numpy as np, torch
import droid_backends as db
from droid_backends import synth
prob = synth.make_config("cfg2")
B, H, W, C, r = 32, 48, 64, 128, 3
fmaps, coords = synth.make_corr_inputs(prob, n_edges=B, C=C, seed=0)
fm = torch.from_numpy(fmaps).cuda().float() / 4
f1 = fm[torch.from_numpy(prob.ii[:B]).cuda()].permute(0, 2, 3, 1).contiguous()
f2 = fm[torch.from_numpy(prob.jj[:B]).cuda()].permute(0, 2, 3, 1).contiguous()
c = torch.from_numpy(coords).cuda()[:, None].contiguous()
g = torch.randn(B, 1, 49, H, W, device="cuda")
for lvl in (0, 2):
    f2l = f2
    for _ in range(lvl):
        f2l = torch.nn.functional.avg_pool2d(f2l.permute(0, 3, 1, 2), 2, stride=2).permute(0, 2, 3, 1).contiguous()
    cl = (c / 2 ** lvl).contiguous()
    db.altcorr_backward(f1, f2l, cl, g, r); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): db.altcorr_backward(f1, f2l, cl, g, r)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"level {lvl}: altcorr_backward {ms*1e3:.0f} us for {B} edges (incl. 3 zero-fills) = {B*H*W/ms/1e6:.3f} Gpix/s")
