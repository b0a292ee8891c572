"""Wall-clock stamps (10 ns ticks) of the single-launch Cholesky: diagonal workgroup (0) and the one below (1).
Build: -DCHOL_STAMPS into tools/libs/lib_stamps.so (tools/README.md)."""
import sys, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/droid-slam_reserch_amd")
import numpy as np, torch
import os
lib = ctypes.CDLL(os.environ.get("STAMPS_LIB", "/root/repo/tools/libs/lib_stamps.so"))
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
st = np.array(buf[:], dtype=np.uint64).astype(np.int64).reshape(32, 2, 16)
t0 = st[0, 0, 11]
print("columns: wait-begin, inputs-seen, body-start(0), loaded(1), upd(2), potrf0(3), bar(4), trsm0(5), potrf1(6), (7), p123(8), stored(9), published(13); us since start")
for kp in range(0, 24):
    for wg in (0, 1):
        r = st[kp, wg]
        seq = [r[11], r[12], r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[13]]
        print(f"col {kp:2d} wg {wg}: " + " ".join(f"{(v - t0) / 100:7.2f}" for v in seq))
buf2 = (ctypes.c_ulonglong * (64 * 8))()
lib.droid_debug_bs_stamps(buf2)
bs = np.array(buf2[:], dtype=np.int64).reshape(64, 8)
t0 = bs[23, 0]
print("back-substitution, per block column j: start, last x seen, mat-vec done, after sync, solved, published (us)")
for j in range(23, -1, -1):
    print(f"j {j:2d}: " + " ".join(f"{(v - t0) / 100:7.2f}" for v in bs[j, :6]))
print("phase deltas (us), mid columns: strips->loaded | last-strip updates | potrf0 | barrier | solve gemm (wave 0) | barrier wait | D11 update + potrf1 | barrier | p=1..3 | publish")
for kp in (8, 12, 16):
    for wg in (0, 1):
        r = st[kp, wg].astype(float) / 100
        print(f"col {kp:2d} wg {wg}: {r[1]-r[0]:6.2f} {r[2]-r[1]:6.2f} {r[3]-r[2]:6.2f} {r[4]-r[3]:6.2f} {r[10]-r[4]:6.2f} {r[5]-r[10]:6.2f} {r[6]-r[5]:6.2f} {r[7]-r[6]:6.2f} {r[8]-r[7]:6.2f} {r[13]-r[8]:6.2f}")
print("inside wave 0's first solve GEMM (diagnostic build only): barrier -> function entered | operands loaded | 4 MFMAs | stores + return")
for kp in (8, 12, 16):
    for wg in (0, 1):
        r = st[kp, wg].astype(float) / 100
        print(f"col {kp:2d} wg {wg}: {r[9]-r[4]:6.2f} {r[14]-r[9]:6.2f} {r[15]-r[14]:6.2f} {r[10]-r[15]:6.2f}")
