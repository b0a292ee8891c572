import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/droid-slam_reserch_amd")
import numpy as np, torch, torch.nn.functional as F
import droid_backends as db
from droid_backends import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, W, r = 48, 64, 3
prob = synth.make_config("cfg2")
fmaps, coords = synth.make_corr_inputs(prob, n_edges=B, seed=0)
dev = "cuda"
ii = torch.from_numpy(prob.ii[:B]).to(dev); jj = torch.from_numpy(prob.jj[:B]).to(dev)
fm = torch.from_numpy(fmaps).to(dev).float() / 4.0
c = torch.from_numpy(coords).to(dev)
pyr = []; x = fm
for l in range(4):
    pyr.append(x.permute(0, 2, 3, 1).contiguous()); x = F.avg_pool2d(x, 2, stride=2)
a1 = pyr[0][ii].contiguous()
a2 = [pyr[l][jj].contiguous() for l in range(4)]
ca = [(c[:, None] / 2 ** l).contiguous() for l in range(4)]
# bbox statistics per 8x8 tile at level 0
cc = coords  # [B,H,W,2]
fx = np.floor(cc[..., 0]); fy = np.floor(cc[..., 1])
t = fx.reshape(B, H // 8, 8, W // 8, 8); ty = fy.reshape(B, H // 8, 8, W // 8, 8)
rw = (t.max(axis=(2, 4)) - t.min(axis=(2, 4)) + 8); rh = (ty.max(axis=(2, 4)) - ty.min(axis=(2, 4)) + 8)
print("level0 bbox area: mean", (rw * rh).mean(), "max", (rw * rh).max(), "frac > 448:", ((rw * rh) > 448).mean())
for l in range(4):
    db.altcorr_forward(a1, a2[l], ca[l], r); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): db.altcorr_forward(a1, a2[l], ca[l], r)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = B * H * W * 64 * 128 * 2
    print(f"level {l}: {ms*1e3:.0f} us  {fl/ms/1e9:.1f} TFLOP/s useful")
