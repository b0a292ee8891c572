"""Rank 0 of the W-rank bench (BASELINE configs[3]: 8000 edges in total) on one GPU, for profiling: scale_one.py W [iterations]"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_reserch_amd")]
import numpy as np, torch
from droid_backends import ba_driver, synth
world = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
prob = synth.make_ba_problem(N=256, E=8000, H=48, W=64, lm=1e-5, ep=1e-2, seed=synth.CONFIG_SEEDS["cfg4"])  # BASELINE configs[3]
ranges = ba_driver.partition_frames(prob.ii, 256, world)
sh = ba_driver.shard_problem(prob, ranges, 0)
p = ba_driver.BAProblemDev(poses=t(prob.poses), disps=t(prob.disps), intrinsics=t(prob.intrinsics), disps_sens=t(prob.disps_sens),
                           targets=t(sh["targets"]), weights=t(sh["weights"]), eta=t(sh["eta"]), ii=t(sh["ii"]), jj=t(sh["jj"]))
be = ba_driver.HipBackend()
be.prepare(p, prob.t0, prob.t1, sh["own"], False)
for _ in range(iters):
    be.build(p, False); be.solve_update(p, prob.lm, prob.ep, False)
torch.cuda.synchronize()
print("done")
