"""Where does the fp32 noise of one BA iteration enter?  (cfg3-sized graph, default seed 12: the draw on which the
2-iteration state differs from the fp64 oracle by 2.5e-4 in a disparity.)  Builds the reduced camera system on the
device (phase API) with and without the Schur part, compares A = sum of Hessian blocks, S = E C^-1 E^T, the rhs parts
against the fp64 oracle's, and propagates each difference through the damped solve in numpy:
    dx(H, b) = (H + diag(ep + lm diag H))^-1 b
Not a test; the oracle is used as the checker only.  usage: python tools/noise_probe.py [seed] [config]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_reserch_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

import droid_backends as db
import oracle
from droid_backends import synth
from util import ba_args, to_dev

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg3"      # e.g. `noise_probe.py 3 cfg4`: the dense-slot path
p = synth.make_config(cfg, seed=seed)
lib = db._lib.load()
nbuf, H, W = p.disps.shape
E, M, P = len(p.ii), p.eta.shape[0], p.t1 - p.t0
n = 6 * P
ref = oracle.ba(*ba_args(p), 1, p.lm, p.ep, False, debug=True)


def device_system(motion_only):
    d = to_dev(p, torch)
    m = 0 if motion_only else M
    nbytes = lib.droid_ba_workspace_bytes(E, nbuf, H, W, p.t0, p.t1, m)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    assert lib.droid_ba_prepare(d["ii"].data_ptr(), d["jj"].data_ptr(), E, nbuf, H, W, m, p.t0, p.t1, 0, nbuf,
                                int(motion_only), ws.data_ptr(), nbytes, s) == 0
    assert lib.droid_ba_build(d["poses"].data_ptr(), d["disps"].data_ptr(), d["intrinsics"].data_ptr(),
                              d["disps_sens"].data_ptr(), d["targets"].data_ptr(), d["weights"].data_ptr(),
                              d["eta"].data_ptr() if m else None, d["ii"].data_ptr(), d["jj"].data_ptr(), E, nbuf, H, W, m,
                              p.t0, p.t1, int(motion_only), ws.data_ptr(), nbytes, s) == 0
    torch.cuda.synchronize()
    nel = ctypes.c_size_t(0)
    ptr = lib.droid_ba_system(ws.data_ptr(), E, nbuf, H, W, p.t0, p.t1, m, ctypes.byref(nel))
    off = ptr - ws.data_ptr()
    sy = ws[off:off + nel.value * 8].view(torch.float64).view(n + 1, -1).cpu().numpy()
    L = np.tril(sy[:n, :n])
    return L + np.tril(L, -1).T, sy[n, :n].copy()


def solve(Hm, b):
    A = Hm.copy()
    dg = np.diag(A).copy()
    A[np.diag_indices(n)] = dg + p.ep + p.lm * dg
    return np.linalg.solve(A, b)


# truth: A and v from the oracle's per-edge blocks (droid_kernels.cu:1378-1390), S and E Q w by difference
At = np.zeros((n, n))
vt = np.zeros(n)
rows = np.stack([p.ii, p.ii, p.jj, p.jj], 1) - p.t0
cols = np.stack([p.ii, p.jj, p.ii, p.jj], 1) - p.t0
for e in range(E):
    for k in range(4):
        r, c = rows[e, k], cols[e, k]
        if r >= 0 and c >= 0:
            At[6 * r:6 * r + 6, 6 * c:6 * c + 6] += ref["Hs"][e, k]
    for k, fr in enumerate((p.ii[e], p.jj[e])):
        if fr - p.t0 >= 0:
            vt[6 * (fr - p.t0):6 * (fr - p.t0) + 6] += ref["vs"][e, k]
Ht = np.tril(ref["H"]) + np.tril(ref["H"], -1).T
bt = ref["b"]
St, uwt = At - Ht, vt - bt

Hh, bh = device_system(False)
Ah, vh = device_system(True)
Sh, uwh = Ah - Hh, vh - bh
xt = solve(Ht, bt)
rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
print(f"seed {seed}: |A| {np.abs(At).max():.3e} |S| {np.abs(St).max():.3e} |H| {np.abs(Ht).max():.3e} |b| {np.abs(bt).max():.3e} |dx| {np.abs(xt).max():.3e}")
print(f"rel err   A {rel(Ah, At):.2e}  S {rel(Sh, St):.2e}  H {rel(Hh, Ht):.2e}  v {rel(vh, vt):.2e}  EQw {rel(uwh, uwt):.2e}  b {rel(bh, bt):.2e}")
ev = np.linalg.eigvalsh(Ht + np.diag(p.ep + p.lm * np.diag(Ht)))
print(f"damped reduced system: eig min {ev[0]:.3e} second {ev[1]:.3e} max {ev[-1]:.3e}  cond {ev[-1] / ev[0]:.2e}")
for name, Hm, bm in (("all device", Hh, bh), ("only dA", Ht + (Ah - At), bt), ("only dS", Ht - (Sh - St), bt),
                     ("only dv", Ht, bt + (vh - vt)), ("only dEQw", Ht, bt - (uwh - uwt)), ("only db", Ht, bh)):
    x = solve(Hm, bm)
    print(f"  dx error through the solve, {name:11s}: max {np.abs(x - xt).max():.3e}")
print(f"oracle dx vs numpy solve of oracle system: {np.abs(ref['dx'].reshape(-1) - xt).max():.2e}")

for it in (1, 2):
    d = to_dev(p, torch)
    dx, dz = db.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["targets"], d["weights"], d["eta"], d["ii"],
                   d["jj"], p.t0, p.t1, it, p.lm, p.ep, False)
    torch.cuda.synchronize()
    r = ref if it == 1 else oracle.ba(*ba_args(p), it, p.lm, p.ep, False)
    dd = np.abs(d["disps"].cpu().numpy() - r["disps"])
    k = np.unravel_index(dd.argmax(), dd.shape)
    print(f"it {it}: max|ddx| {np.abs(dx.cpu().numpy() - r['dx']).max():.3e}  max|dt| "
          f"{np.abs(d['poses'].cpu().numpy()[:, :3] - r['poses'][:, :3]).max():.3e}  max|ddisp| {dd.max():.3e} at {k} "
          f"(disp {r['disps'][k]:.4f}, init {p.disps[k]:.4f}); pixels > 1e-4: {(dd > 1e-4).sum()}, > 5e-5: {(dd > 5e-5).sum()}")
    if it == 1:
        # how much of the disparity error is the pose error carried through dz = Q (w - E^T dx)?
        dzr = r["dz"]
        dzh = dz.cpu().numpy()
        e = np.abs(dzh - dzr)
        print(f"      dz: max err {e.max():.3e}, |dz| max {np.abs(dzr).max():.3e}")
