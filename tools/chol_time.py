"""Time droid_chol_solve (pack + factor + back-substitution) for n = 1530.  The library comes from DROID_HIP_LIB
(default: the in-tree build); DROID_CHOL_MULTI_LAUNCH=1 selects the one-launch-per-block-column factorisation."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "droid-slam_reserch_amd"))
from droid_backends import _lib
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1530
rng = np.random.default_rng(0)
B = rng.normal(size=(n, n)); A = B @ B.T + n * np.eye(n); b = rng.normal(size=n)
dA0 = torch.from_numpy(A).cuda(); db = torch.from_numpy(b).cuda()
x = torch.zeros(n, dtype=torch.float64, device="cuda"); flag = torch.zeros(1, dtype=torch.int32, device="cuda")
scratch = torch.zeros(lib.droid_chol_scratch_doubles(n), dtype=torch.float64, device="cuda")
def run():
    return lib.droid_chol_solve(dA0.data_ptr(), db.data_ptr(), x.data_ptr(), n, scratch.data_ptr(), flag.data_ptr(), None)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 30
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
xs = np.linalg.solve(A, b)
print("%s n=%d  %.1f us per solve   rel err %.2e  fail=%d" % (os.environ.get("DROID_HIP_LIB", "in-tree"), n,
      e0.elapsed_time(e1) / reps * 1e3, np.abs(x.cpu().numpy() - xs).max() / np.abs(xs).max(), int(flag.item())))
