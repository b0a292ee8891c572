"""world_size-2 gloo tests of the edge-sharded BA host logic (droid_backends/ba_driver.py):
partition by source frame, one all-reduce of the dense reduced system per iteration, replicated
solve, local depth back-substitution.  The compute backend here is the CPU oracle's two-phase
API (tests may use the oracle; the product backend is HipBackend) so the collective, the
partition and the phase order -- the parts that are identical on the GPU path -- are exercised
without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """Same interface as ba_driver.HipBackend, CPU tensors, arithmetic by oracle.BAPhases."""

    def __init__(self):
        import oracle
        self.oracle = oracle
        self.ph = None

    def prepare(self, p, t0, t1, own, motion_only):
        self.t0, self.t1, self.own = t0, t1, own
        self.dx = torch.zeros((t1 - t0, 6), dtype=torch.float64)

    def build(self, p, motion_only):
        self.ph = self.oracle.BAPhases("f64")
        H, b = self.ph.build(p.poses.numpy(), p.disps.numpy(), p.intrinsics.numpy(), p.disps_sens.numpy(),
                             p.targets.numpy(), p.weights.numpy(), p.eta.numpy(), p.ii.numpy(), p.jj.numpy(),
                             self.t0, self.t1, self.own[0], self.own[1], motion_only)
        n = H.shape[0]
        self.system = torch.zeros((n + 1, n), dtype=torch.float64)
        self.system[:n] = torch.from_numpy(H)
        self.system[n] = torch.from_numpy(b)
        return self.system

    def solve_update(self, p, lm, ep, motion_only):
        n = self.system.shape[1]
        poses, disps, dx = self.ph.finish(self.system[:n].numpy(), self.system[n].numpy(), lm, ep)
        p.poses.copy_(torch.from_numpy(poses).to(p.poses.dtype))
        p.disps.copy_(torch.from_numpy(disps).to(p.disps.dtype))
        self.dx = torch.from_numpy(dx)
        return self.dx


class PackedOracleBackend(OracleBackend):
    """Like the HIP backend, only the lower triangle + rhs row of `system` are meaningful after
    the collective: exposes reduce_index() and rebuilds the symmetric matrix from the lower part."""

    def reduce_index(self):
        n1, n = self.system.shape
        r = torch.arange(n1).view(-1, 1)
        c = torch.arange(n).view(1, -1)
        return ((c <= r) & (c < n)).flatten().nonzero().squeeze(1)

    def build(self, p, motion_only):
        system = super().build(p, motion_only)
        n = system.shape[1]
        system[:n] += torch.triu(torch.full((n, n), 1e30, dtype=torch.float64), 1)  # poison what must not be read
        return system

    def solve_update(self, p, lm, ep, motion_only):
        n = self.system.shape[1]
        low = torch.tril(self.system[:n])
        self.system[:n] = low + torch.tril(low, -1).T
        return super().solve_update(p, lm, ep, motion_only)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, iterations, packed=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "droid-slam_reserch_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from droid_backends import ba_driver, synth
    prob = synth.make_ba_problem(N=8, E=32, H=12, W=16, seed=0)
    ranges = ba_driver.partition_frames(prob.ii, prob.t1, world)
    sh = ba_driver.shard_problem(prob, ranges, rank)
    f64 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.float64)
    p = ba_driver.BAProblemDev(poses=f64(prob.poses), disps=f64(prob.disps), intrinsics=f64(prob.intrinsics),
                               disps_sens=f64(prob.disps_sens), targets=f64(sh["targets"]),
                               weights=f64(sh["weights"]), eta=f64(sh["eta"]),
                               ii=torch.from_numpy(sh["ii"]), jj=torch.from_numpy(sh["jj"]))
    solver = ba_driver.ShardedBA(backend=PackedOracleBackend() if packed else OracleBackend())
    dx = solver.run(p, prob.t0, prob.t1, iterations, prob.lm, prob.ep, own=sh["own"])
    solver.gather_disps(p.disps, ranges)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), poses=p.poses.numpy(), disps=p.disps.numpy(),
             dx=dx.numpy(), n_local=len(sh["ii"]), f0=ranges[rank][0], f1=ranges[rank][1])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,packed", [(2, False), (3, False), (2, True)])
def test_sharded_ba_equals_single_rank(tmp_path, world, packed, oracle):
    """packed=True: the collective moves only the lower triangle + rhs row (what HipBackend asks for)."""
    from droid_backends import synth
    iterations = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), iterations, packed), nprocs=world, join=True)
    prob = synth.make_ba_problem(N=8, E=32, H=12, W=16, seed=0)
    ref = oracle.ba(prob.poses, prob.disps, prob.intrinsics, prob.disps_sens, prob.targets, prob.weights,
                    prob.eta, prob.ii, prob.jj, prob.t0, prob.t1, iterations, prob.lm, prob.ep, False)
    outs = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    assert sum(int(o["n_local"]) for o in outs) == len(prob.ii)  # every edge on exactly one rank
    for o in outs:
        # the solve is replicated and deterministic: dx and poses identical on every rank and equal to
        # the unsharded result up to the fp64 summation order of the all-reduce
        assert np.abs(o["dx"] - outs[0]["dx"]).max() == 0.0
        assert np.abs(o["poses"] - ref["poses"]).max() < 1e-9
        assert np.abs(o["disps"] - ref["disps"]).max() < 1e-8


def test_partition_frames_is_a_contiguous_cover():
    from droid_backends import ba_driver, synth
    prob = synth.make_config("cfg2")
    for world in (1, 2, 4, 8):
        r = ba_driver.partition_frames(prob.ii, prob.t1, world)
        assert r[0][0] == 0 and r[-1][1] == prob.t1
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        counts = [int(((prob.ii >= a) & (prob.ii < b)).sum()) for a, b in r]
        assert sum(counts) == len(prob.ii)
        assert max(counts) <= 1.5 * len(prob.ii) / world + 16  # balanced by edge count


def test_local_eta_rows_match_device_slot_rule():
    """Depth slots of a rank = unique(ii_local) U (window n owned frames), ascending."""
    from droid_backends import ba_driver, synth
    prob = synth.make_config("cfg1")
    ranges = ba_driver.partition_frames(prob.ii, prob.t1, 2)
    rows = []
    for rank in range(2):
        sh = ba_driver.shard_problem(prob, ranges, rank)
        fr = ba_driver.local_eta_rows(sh["ii"], prob.t0, prob.t1, sh["own"]).numpy()
        assert sh["eta"].shape[0] == len(fr)
        assert np.all((fr >= ranges[rank][0]) & (fr < ranges[rank][1]))
        rows.append(fr)
    allrows = np.concatenate(rows)
    kx = np.unique(np.concatenate([np.arange(prob.t0, prob.t1), prob.ii]))
    assert np.array_equal(np.sort(allrows), kx)  # slots are partitioned, none duplicated
