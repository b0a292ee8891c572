"""Two ranks sharing the one GPU of the test box (gloo carries the all-reduce of the device
tensor): the HIP phase API with frame ownership (own0/own1) against the unsharded HIP result and
the fp64 oracle.  The 8-GPU RCCL run is the driver's; this covers the device-side slot / entry
bookkeeping of a shard."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "droid-slam_reserch_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from droid_backends import ba_driver, synth
    prob = synth.make_config("cfg1")
    ranges = ba_driver.partition_frames(prob.ii, prob.t1, world)
    sh = ba_driver.shard_problem(prob, ranges, rank)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p = ba_driver.BAProblemDev(poses=t(prob.poses), disps=t(prob.disps), intrinsics=t(prob.intrinsics),
                               disps_sens=t(prob.disps_sens), targets=t(sh["targets"]), weights=t(sh["weights"]),
                               eta=t(sh["eta"]), ii=t(sh["ii"]), jj=t(sh["jj"]))
    solver = ba_driver.ShardedBA()
    solver.run(p, prob.t0, prob.t1, 2, prob.lm, prob.ep, own=sh["own"])
    solver.gather_disps(p.disps, ranges)
    torch.cuda.synchronize()
    st, m = solver.backend.status()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), poses=p.poses.cpu().numpy(), disps=p.disps.cpu().numpy(),
             status=st, M=m, M_expected=sh["eta"].shape[0])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hip_ba_matches_single_and_oracle(tmp_path, backends, oracle):
    import torch
    import torch.multiprocessing as mp
    from droid_backends import synth
    from util import ba_args, run_hip_ba
    assert torch.cuda.is_available()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    prob = synth.make_config("cfg1")
    single = run_hip_ba(backends, prob, torch, 2)
    ref = oracle.ba(*ba_args(prob), 2, prob.lm, prob.ep, False)
    outs = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(2)]
    for o in outs:
        assert int(o["status"]) & 3 == 0 and int(o["M"]) == int(o["M_expected"])
        assert np.abs(o["poses"] - outs[0]["poses"]).max() == 0.0  # replicated solve is bit-identical
        assert np.abs(o["poses"] - single["poses"]).max() < 1e-5
        assert np.abs(o["disps"] - single["disps"]).max() < 1e-4
        assert np.abs(o["poses"] - ref["poses"]).max() < 1e-4
        assert np.abs(o["disps"] - ref["disps"]).max() < 1e-4
