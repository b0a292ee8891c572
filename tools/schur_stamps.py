import sys, ctypes, shutil
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/droid-slam_reserch_amd")
import os; os.environ["DROID_HIP_LIB"] = "/root/repo/tools/libs/lib_sstamps.so"
import numpy as np, torch
import droid_backends as db
from droid_backends import synth
sys.path.insert(0, "/root/repo/tests")
from util import run_hip_ba
p = synth.make_config("cfg3")
for _ in range(2): run_hip_ba(db, p, torch, 2)
lib = db._lib.load()
buf = (ctypes.c_ulonglong * 128)()
lib.droid_debug_schur_stamps(buf)
st = np.array(buf[:], dtype=np.int64).reshape(16, 8)
names = ["prev(mfma)+loopback", "top barrier", "edge loop", "self+pad", "barrier5", "mfma", "atomics"]
for w in range(4):
    print("wave", w, {n: int(v) for n, v in zip(["to_top", "top_barrier", "edge_loop", "self_pad", "barrier5", "mfma(last)", "atomics"], st[w, :7])}, "sum", st[w].sum())
