"""Per-rank compute of the multi-GPU bench emulated on ONE GPU: for world = 1,2,4,8 build the graph, take the shard of
rank 0 and of a middle rank, and time the local iteration (no collective).  Default: BASELINE configs[3], 8000 edges in
total (what `bench.py --gpus N` runs); `weak`: 2000 edges per rank.  usage: scale_emul.py [weak] [worlds=1,8]"""
import sys, time
WEAK = len(sys.argv) > 1 and sys.argv[1] == "weak"
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_reserch_amd")]
import numpy as np, torch
from droid_backends import ba_driver, synth
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
WORLDS = [int(x) for a in sys.argv[1:] if a.startswith("worlds=") for x in a[7:].split(",")] or [1, 2, 4, 8]
for world in WORLDS:
    prob = synth.make_ba_problem(N=256, E=(2000 * world if WEAK else 8000), H=48, W=64, lm=1e-5, ep=1e-2,
                                 seed=synth.CONFIG_SEEDS["cfg3" if WEAK else "cfg4"])
    ranges = ba_driver.partition_frames(prob.ii, 256, world)
    for rank in sorted({0, world // 2}):
        sh = ba_driver.shard_problem(prob, ranges, rank)
        p = ba_driver.BAProblemDev(poses=t(prob.poses), disps=t(prob.disps), intrinsics=t(prob.intrinsics),
                                   disps_sens=t(prob.disps_sens), targets=t(sh["targets"]), weights=t(sh["weights"]),
                                   eta=t(sh["eta"]), ii=t(sh["ii"]), jj=t(sh["jj"]))
        be = ba_driver.HipBackend()
        be.prepare(p, prob.t0, prob.t1, sh["own"], False)
        for _ in range(2):
            be.build(p, False); be.solve_update(p, prob.lm, prob.ep, False)
        acc = {}
        for _ in range(4):
            s = be.profile_iteration(p, prob.lm, prob.ep, False)
            for k, v in s.items(): acc[k] = acc.get(k, 0.0) + v / 4
        print(f"world {world} rank {rank}: frames {sh['own']} edges {len(sh['ii'])} slots {sh['eta'].shape[0]}  " +
              " ".join(f"{k}={v:.3f}" for k, v in acc.items() if k != "unused"), flush=True)
