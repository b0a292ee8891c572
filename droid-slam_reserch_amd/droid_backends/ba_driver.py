"""Phase-level driver of the dense BA: single GPU and edge-sharded multi-GPU (one process per
GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Sharding: the factor graph is partitioned BY SOURCE FRAME.  Rank g owns a
contiguous range of frames [f0,f1), the depth maps of those frames and every edge whose source
`ii` lies in the range.  Per Gauss-Newton iteration each rank linearises its edges and reduces its
depth frames locally (Hii/Hij/Hjj blocks, -E C^-1 E^T, rhs) into the dense (6P+1)^2 fp64 system,
ONE all-reduce(sum) of its lower triangle + rhs row (packed) combines the systems, every rank then runs the identical Cholesky solve (dx
is bit-identical on all ranks), back-substitutes its own depth frames and applies the same pose
retraction.  No other collective is on the data path; `gather_disps` optionally re-assembles the
depth maps at the end of a call.

The compute backend is pluggable so that the host logic (partition, collective, phase order) is
covered by world_size-2 gloo tests on CPU; the product backend is `HipBackend` (C ABI).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch

from . import _lib


@dataclass
class BAProblemDev:
    """Device tensors of one rank, shaped like the reference's `ba` arguments (droid.cpp:88-102)."""
    poses: torch.Tensor       # [nbuf,7] f32, replicated on all ranks
    disps: torch.Tensor       # [nbuf,H,W] f32, valid for owned frames
    intrinsics: torch.Tensor  # [4]
    disps_sens: torch.Tensor  # [nbuf,H,W]
    targets: torch.Tensor     # [E_local,2,H,W]
    weights: torch.Tensor     # [E_local,2,H,W]
    eta: torch.Tensor         # [M_local,H,W]  rows = depth slots of THIS rank, ascending frame
    ii: torch.Tensor          # [E_local] int64
    jj: torch.Tensor          # [E_local] int64


def partition_frames(ii, n_frames, world):
    """Contiguous frame ranges [(f0,f1)] * world balanced by outgoing-edge count (host side)."""
    ii = torch.as_tensor(ii).cpu().to(torch.int64)
    deg = torch.bincount(ii, minlength=n_frames).to(torch.float64) + 1e-3  # every frame costs a little
    csum = torch.cumsum(deg, 0)
    total = float(csum[-1])
    bounds = [0]
    for r in range(1, world):
        f = int(torch.searchsorted(csum, torch.tensor(total * r / world, dtype=torch.float64)).item()) + 1
        bounds.append(min(max(f, bounds[-1]), n_frames))
    bounds.append(n_frames)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def local_eta_rows(ii_local, t0, t1, own):
    """Frames (ascending) that own a depth slot on this rank: unique(ii_local) U ([t0,t1) n own)."""
    w0, w1 = max(t0, own[0]), min(t1, own[1])
    fr = torch.unique(torch.cat([torch.as_tensor(ii_local).cpu().to(torch.int64),
                                 torch.arange(w0, max(w0, w1), dtype=torch.int64)]))
    return fr


class HipBackend:
    """The product compute backend: include/droid_backends_hip.h phase API on the current stream."""

    def __init__(self):
        self.lib = _lib.load()
        self.ws = None

    def __del__(self):
        # the library keys the launch hints by workspace address: drop the registration with the buffers it points to
        try:
            if self.ws is not None:
                self.lib.droid_ba_attach_launch_hints(self.ws.data_ptr(), None)
        except Exception:
            pass

    def _args(self, p: BAProblemDev, t0, t1, motion_only):
        nbuf, H, W = p.disps.shape
        E = int(p.ii.shape[0])
        M = 0 if motion_only else int(p.eta.shape[0])
        return E, int(nbuf), int(H), int(W), M, int(t0), int(t1)

    def prepare(self, p: BAProblemDev, t0, t1, own, motion_only):
        E, nbuf, H, W, M, t0, t1 = self._args(p, t0, t1, motion_only)
        nbytes = self.lib.droid_ba_workspace_bytes(E, nbuf, H, W, t0, t1, M)
        if nbytes == 0:
            raise RuntimeError("ba: bad sizes / window")
        if self.ws is None or self.ws.numel() < nbytes:
            if self.ws is not None:
                self.lib.droid_ba_attach_launch_hints(self.ws.data_ptr(), None)
            self.ws = torch.empty(nbytes + 4096, dtype=torch.uint8, device=p.poses.device)
            self.hints = torch.zeros(2, dtype=torch.int32).pin_memory()   # include/droid_backends_hip.h: launch hints
            _lib.check(self.lib.droid_ba_attach_launch_hints(self.ws.data_ptr(), self.hints.data_ptr()), "ba (launch hints)")
        self._dims = (E, nbuf, H, W, M, t0, t1)
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(self.lib.droid_ba_prepare(p.ii.data_ptr(), p.jj.data_ptr(), E, nbuf, H, W, M, t0, t1,
                                             int(own[0]), int(own[1]), int(motion_only), self.ws.data_ptr(),
                                             self.ws.numel(), s), "ba_prepare")
        nel_sys = ctypes.c_size_t(0)
        ptr = self.lib.droid_ba_system(self.ws.data_ptr(), E, nbuf, H, W, t0, t1, M, ctypes.byref(nel_sys))
        off = ptr - self.ws.data_ptr()
        self.system = self.ws[off:off + nel_sys.value * 8].view(torch.float64)
        nel = ctypes.c_size_t(0)
        ptr = self.lib.droid_ba_packed_system(self.ws.data_ptr(), E, nbuf, H, W, t0, t1, M, ctypes.byref(nel))
        off = ptr - self.ws.data_ptr()
        self.packed = self.ws[off:off + nel.value * 8].view(torch.float64)   # lower triangle + rhs row, contiguous
        self.dx = torch.empty((t1 - t0, 6), dtype=torch.float32, device=p.poses.device)
        self.dz = torch.empty((M, H * W), dtype=torch.float32, device=p.poses.device)
        self._ridx_key = (t1 - t0, nel_sys.value)   # PITCHED element count: reduce_index derives the row pitch from it

    def reduce_index(self):
        """Flat indices of the entries of `system` that the solve reads: the lower triangle of the
        6P x 6P matrix plus the rhs row (include/droid_backends_hip.h: rows of `ld` doubles).  The
        sharded driver all-reduces only these (half the bytes of the dense buffer)."""
        P, nel = self._ridx_key
        cache = getattr(self, "_ridx", None)
        if cache is None or cache[0] != self._ridx_key or cache[1].device != self.system.device:
            n = 6 * P
            ld = nel // (n + 1)
            r = torch.arange(n + 1, device=self.system.device).view(-1, 1)
            c = torch.arange(ld, device=self.system.device).view(1, -1)
            idx = ((c <= r) & (c < n)).flatten().nonzero().squeeze(1)
            self._ridx = cache = (self._ridx_key, idx)
        return cache[1]

    def build(self, p: BAProblemDev, motion_only):
        E, nbuf, H, W, M, t0, t1 = self._dims
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(self.lib.droid_ba_build(p.poses.data_ptr(), p.disps.data_ptr(), p.intrinsics.data_ptr(),
                                           p.disps_sens.data_ptr(), p.targets.data_ptr(), p.weights.data_ptr(),
                                           p.eta.data_ptr() if M > 0 else None, p.ii.data_ptr(), p.jj.data_ptr(),
                                           E, nbuf, H, W, M, t0, t1, int(motion_only), self.ws.data_ptr(),
                                           self.ws.numel(), s), "ba_build")
        return self.system

    def build_packed(self, p: BAProblemDev, motion_only):
        """Multi-GPU build phase: the contribution of this rank lands in `self.packed` (what gets all-reduced)."""
        E, nbuf, H, W, M, t0, t1 = self._dims
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(self.lib.droid_ba_build_packed(p.poses.data_ptr(), p.disps.data_ptr(), p.intrinsics.data_ptr(),
                                                  p.disps_sens.data_ptr(), p.targets.data_ptr(), p.weights.data_ptr(),
                                                  p.eta.data_ptr() if M > 0 else None, p.ii.data_ptr(), p.jj.data_ptr(),
                                                  E, nbuf, H, W, M, t0, t1, int(motion_only), self.ws.data_ptr(),
                                                  self.ws.numel(), s), "ba_build_packed")
        return self.packed

    def unpack(self, motion_only):
        E, nbuf, H, W, M, t0, t1 = self._dims
        _lib.check(self.lib.droid_ba_unpack_system(E, nbuf, H, W, M, t0, t1, int(motion_only), self.ws.data_ptr(),
                                                   self.ws.numel(), torch.cuda.current_stream().cuda_stream),
                   "ba_unpack_system")

    def solve_update(self, p: BAProblemDev, lm, ep, motion_only):
        E, nbuf, H, W, M, t0, t1 = self._dims
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(self.lib.droid_ba_solve_update(p.poses.data_ptr(), p.disps.data_ptr(), p.intrinsics.data_ptr(),
                                                  p.weights.data_ptr(), p.ii.data_ptr(), p.jj.data_ptr(), E, nbuf, H, W, M, t0, t1, float(lm), float(ep),
                                                  int(motion_only), self.dx.data_ptr(),
                                                  self.dz.data_ptr() if M > 0 else None, self.ws.data_ptr(),
                                                  self.ws.numel(), s), "ba_solve_update")
        return self.dx

    # ---- overlap of the collective with the solve (opt-in: ShardedBA(overlap=True)) ----------------------------
    # chunks = collectives per iteration: each costs its own launch + synchronisation latency on the wire, so few
    OVERLAP_MAX_CHUNKS = int(__import__("os").environ.get("DROID_BA_OVERLAP_CHUNKS", "4"))

    def overlap_plan(self):
        """[(first, last)] element ranges of `self.packed`, one per chunk (whole block rows of the system, in order)."""
        E, nbuf, H, W, M, t0, t1 = self._dims
        nc = ctypes.c_int(0)
        offs = (ctypes.c_size_t * (self.OVERLAP_MAX_CHUNKS + 1))()
        _lib.check(self.lib.droid_ba_overlap_plan(t0, t1, self.OVERLAP_MAX_CHUNKS, ctypes.byref(nc), offs), "ba_overlap_plan")
        return [(int(offs[c]), int(offs[c + 1])) for c in range(nc.value)]

    def unpack_chunk(self, chunk, lm, ep, epoch):
        """Chunk `chunk` of the (all-reduced) packed system -> pitched matrix, damped, then published for `epoch`;
        on the current stream (the side stream of the overlap)."""
        E, nbuf, H, W, M, t0, t1 = self._dims
        _lib.check(self.lib.droid_ba_unpack_chunk(E, nbuf, H, W, M, t0, t1, int(chunk), self.OVERLAP_MAX_CHUNKS, float(lm),
                                                  float(ep), int(epoch), self.ws.data_ptr(), self.ws.numel(),
                                                  torch.cuda.current_stream().cuda_stream), "ba_unpack_chunk")

    def solve_update_overlap(self, p: BAProblemDev, epoch, motion_only):
        """Launches the solve of iteration `epoch` BEFORE its system has been reduced: the factorisation waits for the
        block rows it is about to read.  Returns False when the single-launch solver cannot take this system."""
        E, nbuf, H, W, M, t0, t1 = self._dims
        s = torch.cuda.current_stream().cuda_stream
        rc = self.lib.droid_ba_solve_update_overlap(p.poses.data_ptr(), p.disps.data_ptr(), p.intrinsics.data_ptr(),
                                                    p.weights.data_ptr(), p.ii.data_ptr(), p.jj.data_ptr(), E, nbuf, H, W, M,
                                                    t0, t1, int(epoch), int(motion_only), self.dx.data_ptr(),
                                                    self.dz.data_ptr() if M > 0 else None, self.ws.data_ptr(),
                                                    self.ws.numel(), s)
        if rc == -1:      # DROID_E_ARG: not available for this system
            return False
        _lib.check(rc, "ba_solve_update_overlap")
        return True

    def profile_iteration(self, p: BAProblemDev, lm, ep, motion_only):
        """Stage times in ms of one iteration (measurement support, synchronises)."""
        E, nbuf, H, W, M, t0, t1 = self._dims
        ms = (ctypes.c_float * 8)()
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(self.lib.droid_ba_profile_iteration(
            p.poses.data_ptr(), p.disps.data_ptr(), p.intrinsics.data_ptr(), p.disps_sens.data_ptr(),
            p.targets.data_ptr(), p.weights.data_ptr(), p.eta.data_ptr() if M > 0 else None, p.ii.data_ptr(),
            p.jj.data_ptr(), E, nbuf, H, W, M, t0, t1, float(lm), float(ep), int(motion_only),
            self.ws.data_ptr(), self.ws.numel(), s, ms), "ba_profile_iteration")
        names = ["linearize", "assemble", "schur", "unused", "factor", "backsolve", "update", "total"]
        return dict(zip(names, [float(x) for x in ms]))

    def status(self):
        st, m = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self.lib.droid_ba_status(self.ws.data_ptr(), torch.cuda.current_stream().cuda_stream,
                                            ctypes.byref(st), ctypes.byref(m)), "ba_status")
        return st.value, m.value


class ShardedBA:
    """`iterations` Gauss-Newton steps over an edge-sharded graph.  world_size 1 degenerates to
    the single-GPU path (no collective is issued)."""

    def __init__(self, backend=None, group=None, overlap=None):
        """overlap: all-reduce the packed system in row chunks on a side stream while the factorisation, launched
        first, already works on the leading block rows (include/droid_backends_hip.h, droid_ba_overlap_plan).  Opt-in
        (default: the environment variable DROID_BA_OVERLAP=1): rehearsed with gloo and in-process shards, not yet
        measured over RCCL."""
        import os
        self.backend = backend if backend is not None else HipBackend()
        self.group = group
        self.overlap = (os.environ.get("DROID_BA_OVERLAP", "0") == "1") if overlap is None else bool(overlap)
        self._side = None

    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def run(self, p: BAProblemDev, t0, t1, iterations, lm, ep, own=None, motion_only=False):
        import torch.distributed as dist
        world = self._world()
        nbuf = int(p.disps.shape[0])
        own = (0, nbuf) if own is None else own
        be = self.backend
        be.prepare(p, t0, t1, own, motion_only)
        use_overlap = self.overlap and world > 1 and hasattr(be, "solve_update_overlap") and not motion_only \
            and int(p.ii.shape[0]) > 0
        for it in range(int(iterations)):
            if use_overlap:
                packed = be.build_packed(p, motion_only)
                main = torch.cuda.current_stream()
                if self._side is None:
                    self._side = torch.cuda.Stream()
                built = torch.cuda.Event()
                built.record(main)
                if be.solve_update_overlap(p, it + 1, motion_only):      # main stream: waits chunk by chunk on the device
                    with torch.cuda.stream(self._side):
                        self._side.wait_event(built)
                        for c, (a, b) in enumerate(be.overlap_plan()):
                            dist.all_reduce(packed[a:b], op=dist.ReduceOp.SUM, group=self.group)
                            be.unpack_chunk(c, lm, ep, it + 1)
                    main.wait_stream(self._side)     # the next build clears the packed system
                    continue
                use_overlap = False                                      # system too small / solver mode: ordinary path
                dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
                be.unpack(motion_only)
                be.solve_update(p, lm, ep, motion_only)
                continue
            if world > 1 and hasattr(be, "build_packed"):
                # the build kernels add straight into the packed triangle: one collective on a contiguous tensor,
                # then ONE extra launch (unpack) in front of the solve -- no gather / scatter of the triangle
                packed = be.build_packed(p, motion_only)
                dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
                be.unpack(motion_only)
            else:
                system = be.build(p, motion_only)
                if world > 1:
                    ridx = be.reduce_index() if hasattr(be, "reduce_index") else None
                    if ridx is None:
                        dist.all_reduce(system, op=dist.ReduceOp.SUM, group=self.group)
                    else:  # only what the solve reads: lower triangle + rhs row, packed
                        flat = system.view(-1)
                        packed = flat.index_select(0, ridx)
                        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
                        flat.index_copy_(0, ridx, packed)
            be.solve_update(p, lm, ep, motion_only)
        return be.dx

    def gather_disps(self, disps, ranges):
        """All ranks end up with every owner's depth maps (one all-reduce of the masked buffer)."""
        import torch.distributed as dist
        world = self._world()
        if world == 1:
            return disps
        rank = dist.get_rank(self.group)
        buf = torch.zeros_like(disps)
        f0, f1 = ranges[rank]
        buf[f0:f1] = disps[f0:f1]
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        disps.copy_(buf)
        return disps


def shard_problem(prob, ranges, rank):
    """Host-side split of a synthetic BAProblem (numpy) into rank-local arrays."""
    import numpy as np
    f0, f1 = ranges[rank]
    keep = (prob.ii >= f0) & (prob.ii < f1)
    kx_all = np.unique(np.concatenate([np.arange(prob.t0, prob.t1), prob.ii]))
    fr = local_eta_rows(prob.ii[keep], prob.t0, prob.t1, (f0, f1)).numpy()
    rows = np.searchsorted(kx_all, fr)
    return dict(targets=prob.targets[keep], weights=prob.weights[keep], ii=prob.ii[keep], jj=prob.jj[keep],
                eta=prob.eta[rows], own=(f0, f1))
