import sys, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/droid-slam_reserch_amd")
import numpy as np, torch
lib = ctypes.CDLL("/root/repo/tools/libs/lib_stamps.so")
lib.droid_chol_scratch_doubles.argtypes = [ctypes.c_int]
lib.droid_chol_scratch_doubles.restype = ctypes.c_size_t
n = 1530
rng = np.random.default_rng(0)
A = rng.normal(size=(n, n + 8)); A = A @ A.T + n * 0.1 * np.eye(n); b = rng.normal(size=n)
dA, dbb = torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda()
x = torch.zeros(n, dtype=torch.float64, device="cuda")
scratch = torch.zeros(lib.droid_chol_scratch_doubles(n), dtype=torch.float64, device="cuda")
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
vp = ctypes.c_void_p
lib.droid_chol_solve.argtypes = [vp, vp, vp, ctypes.c_int, vp, vp, vp]
for _ in range(3):
    lib.droid_chol_solve(dA.data_ptr(), dbb.data_ptr(), x.data_ptr(), n, scratch.data_ptr(), flag.data_ptr(), None)
    torch.cuda.synchronize()
print("err", np.abs(x.cpu().numpy() - np.linalg.solve(A, b)).max())
buf = (ctypes.c_ulonglong * (64 * 16))()
lib.droid_debug_chol_stamps(buf)
st = np.array(buf[:], dtype=np.int64).reshape(32, 2, 16)
labels = ["load", "updates+store", "potrf0", "barrier", "trsm0", "D11+potrf1", "trail0 barrier", "p=1..3", "store"]
for kk in (1, 5, 12, 20):
    for wg in (0, 1):
        d = np.diff(st[kk, wg, :10])
        print("   trsm0 gemm of wave 0 alone:", st[kk, wg, 10] - st[kk, wg, 4], "repeat1", st[kk, wg, 11] - st[kk, wg, 10], "repeat2", st[kk, wg, 12] - st[kk, wg, 11])
        print(f"panel {kk} wg {wg} ({'diag' if wg == 0 else 'below'}):", " ".join(f"{nm}={v}" for nm, v in zip(labels, d)), "total", st[kk, wg, 9] - st[kk, wg, 0])
