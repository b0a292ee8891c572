import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/droid-slam_reserch_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import droid_backends as db, oracle
from test_gpu_corr import _volume_inputs
vol, coords = _volume_inputs(3, 24, 32, 0, np.float16, seed=0)
ref = oracle.corr_index_forward(vol, coords, 3)
out, = db.corr_index_forward(torch.from_numpy(vol).cuda(), torch.from_numpy(coords).cuda(), 3)
got = out.cpu().numpy()
idx = np.argwhere(got != ref)
print(len(idx))
for (b,a,c,y,x) in idx[:8]:
    x0, y0 = coords[b,0,y,x], coords[b,1,y,x]
    fx, fy = np.floor(x0), np.floor(y0)
    dx, dy = np.float32(x0-fx), np.float32(y0-fy)
    one = np.float32(1)
    ws = [((one-dx)*(one-dy)), ((one-dx)*dy), (dx*(one-dy)), (dx*dy)]
    print("idx", b,a,c,y,x, "got", got[b,a,c,y,x], "ref", ref[b,a,c,y,x], "x0,y0", repr(x0), repr(y0), "dx,dy", repr(dx), repr(dy))
    print("   w32", [repr(w) for w in ws], "w16", [repr(np.float16(w)) for w in ws], "w16 via f64", [repr(np.float16(np.float64(a_)*np.float64(b_))) for a_,b_ in [((one-dx),(one-dy)),((one-dx),dy),(dx,(one-dy)),(dx,dy)]])
    x1, y1 = int(fx)-3+a, int(fy)-3+c
    taps = [vol[b,y,x,yy,xx] if (0<=yy<24 and 0<=xx<32) else None for (xx,yy) in [(x1,y1),(x1,y1+1),(x1+1,y1),(x1+1,y1+1)]]
    print("   taps", taps)
