import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_reserch_amd")]
import numpy as np, torch
from droid_backends import ba_driver, synth
prob = synth.make_config("cfg4")
ranges = ba_driver.partition_frames(prob.ii, prob.t1, 8)
sh = ba_driver.shard_problem(prob, ranges, 0)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
p = ba_driver.BAProblemDev(poses=t(prob.poses), disps=t(prob.disps), intrinsics=t(prob.intrinsics),
                           disps_sens=t(prob.disps_sens), targets=t(sh["targets"]), weights=t(sh["weights"]),
                           eta=t(sh["eta"]), ii=t(sh["ii"]), jj=t(sh["jj"]))
be = ba_driver.HipBackend(); be.prepare(p, prob.t0, prob.t1, sh["own"], False)
def tm(fn, K=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
print("build (pitched) %.3f ms  build_packed %.3f ms  unpack %.3f ms" % (tm(lambda: be.build(p, False)), tm(lambda: be.build_packed(p, False)), tm(lambda: be.unpack(False))))
