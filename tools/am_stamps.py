"""s_memtime phase stamps of altcorr_forward_mfma (diagnostic build: csrc/build.sh -DAM_STAMPS -o tools/libs/lib_amstamps.so,
run with DROID_HIP_LIB=tools/libs/lib_amstamps.so).  usage: am_stamps.py [level] [half]"""
import sys, ctypes
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "droid-slam_reserch_amd"))
import numpy as np, torch, torch.nn.functional as F
import droid_backends as db
from droid_backends import synth
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
half = len(sys.argv) > 2 and sys.argv[2] == "half"
B, H, W, r = 64, 48, 64, 3
prob = synth.make_config("cfg2")
fmaps, coords = synth.make_corr_inputs(prob, n_edges=B, seed=0)
ii = torch.from_numpy(prob.ii[:B]).cuda(); jj = torch.from_numpy(prob.jj[:B]).cuda()
fm = torch.from_numpy(fmaps).cuda()
fm = (fm if half else fm.float()) / 4.0
c = torch.from_numpy(coords).cuda()
x = fm; pyr = []
for l in range(4):
    pyr.append(x.permute(0, 2, 3, 1).contiguous()); x = F.avg_pool2d(x, 2, stride=2)
a1 = pyr[0][ii].contiguous(); a2 = pyr[lvl][jj].contiguous(); ca = (c[:, None] / 2 ** lvl).contiguous()
for _ in range(3):
    db.altcorr_forward(a1, a2, ca, r); torch.cuda.synchronize()
lib = db._lib.load()
buf = (ctypes.c_ulonglong * (768 * 4 * 32))()
lib.droid_debug_am_stamps(buf)
st = np.array(buf[:], dtype=np.int64).reshape(768, 4, 32)[:48]
names = ["coords+box", "plan", "dma0", "kloop", "dwrite", "combine"]
d = np.diff(st[:, :, :7], axis=2)
print("half" if half else "fp32", "level", lvl, "mean per phase (s_memtime ticks):", " ".join(f"{n}={v:.0f}" for n, v in zip(names, d.mean(axis=(0, 1)))),
      "total", (st[:, :, 6] - st[:, :, 0]).mean())
print("nblk mean", st[:, :, 8].mean(), "npos mean", st[:, :, 9].mean(), "depth mean", st[:, :, 10].mean(), "fast combine frac", st[:, :, 11].mean())
for wg in (0, 5, 20, 40):
    print("wg", wg, "wave0:", d[wg, 0], "start", st[wg, 0, 0] - st[:, :, 0].min(), "nblk", st[wg, :, 8], "npos", st[wg, 0, 9])
for wg in (0, 20):
    for w in range(4):
        t = st[wg, w]
        print("  wg", wg, "wave", w, "barrier-exit rel. to K-loop start:", (t[16:32:2] - t[2]).tolist(), "mfma phase:", (t[17:32:2] - t[16:32:2]).tolist())
