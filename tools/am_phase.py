import sys, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/droid-slam_reserch_amd")
import numpy as np, torch, torch.nn.functional as F
import droid_backends as db
from droid_backends import synth
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, H, W, r = 64, 48, 64, 3
prob = synth.make_config("cfg2")
fmaps, coords = synth.make_corr_inputs(prob, n_edges=B, seed=0)
ii = torch.from_numpy(prob.ii[:B]).cuda(); jj = torch.from_numpy(prob.jj[:B]).cuda()
fm = torch.from_numpy(fmaps).cuda().float() / 4.0
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
st = np.array(buf[:], dtype=np.int64).reshape(768, 4, 32)
hw = st[:, 0, 12]; xcc = st[:, 0, 13] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 16 + cu
print("distinct CUs among the first 768 workgroups:", len(np.unique(key)))
import collections
groups = collections.defaultdict(list)
for i in range(768): groups[key[i]].append(i)
n = 0
for k, ids in groups.items():
    if len(ids) >= 3 and n < 6:
        n += 1
        t0 = min(st[i, 0, 0] for i in ids)
        print("CU", k, "wgs", ids[:4])
        for i in ids[:4]:
            t = st[i, 0]
            print("   wg %4d: start %6d  kloop [%6d, %6d]  end %6d" % (i, t[0] - t0, t[2] - t0, t[4] - t0, t[6] - t0))
