"""s_memtime phase stamps of altcorr_wave_f16 (diagnostic build: csrc/build.sh -DAM_STAMPS -o tools/libs/lib_amstamps.so,
run with DROID_HIP_LIB=tools/libs/lib_amstamps.so): per level box -> DMA issue -> first block landed -> MFMAs done ->
D written -> combine done, for the first 768 single-wave workgroups."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "droid-slam_reserch_amd"))
import numpy as np, torch, torch.nn.functional as F
import droid_backends as db
from droid_backends import synth
B, H, W, r = 64, 48, 64, 3
prob = synth.make_config("cfg2")
fmaps, coords = synth.make_corr_inputs(prob, n_edges=B, seed=0)
ii = torch.from_numpy(prob.ii[:B]).cuda(); jj = torch.from_numpy(prob.jj[:B]).cuda()
x = torch.from_numpy(fmaps).cuda() / 4.0
c = torch.from_numpy(coords).cuda()
pyr = []
for l in range(4):
    pyr.append(x.permute(0, 2, 3, 1).contiguous()); x = F.avg_pool2d(x, 2, stride=2)
for _ in range(3):
    db.altcorr_pyramid_forward(pyr, c, ii, jj, r); torch.cuda.synchronize()
lib = db._lib.load()
buf = (ctypes.c_ulonglong * (768 * 4 * 32))()
lib.droid_debug_am_stamps(buf)
st = np.array(buf[:], dtype=np.int64).reshape(768, 4, 32)[:, 0, :]
sel = st[:192]        # the sub-tiles of the first edge that ran on... (linear ids 0..191 = all sub-tiles of one unit)
print("waves:", sel.shape[0], " load wait (coords + query rows):", (sel[:, 1] - sel[:, 0]).mean())
names = ["box", "issue", "first block", "blocks", "D write", "combine"]
for l in range(4):
    b = 2 + 7 * l
    d = np.diff(sel[:, b:b + 6], axis=1)
    prev = sel[:, b] - sel[:, b - 1 if l == 0 else b - 2]
    print(f"level {l}: nblk mean {sel[:, b + 6].mean():.1f} max {sel[:, b + 6].max()} | box {prev.mean():.0f} " +
          " ".join(f"{n}={v:.0f}" for n, v in zip(names[1:], d.mean(axis=0))) + f" | level total {(sel[:, b + 5] - sel[:, b - 1 if l == 0 else b - 2]).mean():.0f}")
print("wave total:", (sel[:, 2 + 7 * 3 + 5] - sel[:, 0]).mean(), "cycles (s_memtime)")
interior = sel[sel[:, 8] >= 11]
if len(interior):
    l = 0; b = 2
    d = np.diff(interior[:, b:b + 6], axis=1)
    print(f"interior waves at level 0 ({len(interior)}): " + " ".join(f"{n}={v:.0f}" for n, v in zip(names[1:], d.mean(axis=0))))
