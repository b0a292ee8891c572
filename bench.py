#!/usr/bin/env python3
"""bench.py -- BA update iterations/s (+ correlation-lookup Gpix/s) of the MI355X hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one Gauss-Newton iteration of `droid_backends.ba` (linearise -> depth-side reduce ->
Schur complement -> Cholesky solve -> retraction) on synthetic data that is resident in HBM when
the clock starts.  K steps are timed as ONE ba call with iterations=K (what
BASELINE.md section 3 defines: iters/s = K / wall(ba(iterations=K))).

N = 1 : BASELINE.json configs[2], the graph the metric is quoted on (256 keyframes / 2000 edges,
        48x64, lm=1e-5, ep=1e-2).  `extra.configs_1gpu` adds configs[1] (64 kf / 512 e), configs[3]
        unsharded (256 kf / 8000 e) and configs[4] (stereo 96x128, 128 kf / 1024 e) on the same GPU.
N > 1 : BASELINE.json configs[3]: 256 keyframes / 8000 edges IN TOTAL (--edges-total), sharded by source
        frame over the N ranks (droid_backends/ba_driver.py), one all-reduce (RCCL) of the packed lower
        triangle of the (6P+1)^2 fp64 reduced camera system per iteration, replicated solve.  Total work is
        fixed as N grows ("scaling": "strong").  `value` is the whole-job rate in 2000-edge equivalents:
        iterations/s x (total edges / 2000), so that it is commensurable with the N = 1 line; the raw
        iterations/s of the 8000-edge graph is config.raw_iters_per_sec.  `extra.weak_scaling` is the second
        figure: 2000 edges PER GPU (2000*N in total) on the same 256 keyframes.

Launching: with --gpus N > 1 and no WORLD_SIZE in the environment this process starts the N ranks itself
(`python -m torch.distributed.run --nproc-per-node N` on 127.0.0.1, like the reference's `mp.spawn`, train.py:184-186)
BEFORE anything touches HIP, stays GPU-free and relays rank 0's JSON line; under a launcher (WORLD_SIZE set) it is
one of the ranks, and a WORLD_SIZE that differs from --gpus is an error.

Rank 0 prints ONE JSON line (see the keys below); everything else goes to stderr.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "droid-slam_reserch_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
FP64_VEC_PEAK_TFLOPS = 78.6  # half of the 157.3 TF fp32 vector peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--edges-total", type=int, default=0, help="0 = 2000 for one GPU, 8000 (BASELINE configs[3]) otherwise")
    ap.add_argument("--edges-per-gpu", type=int, default=2000, help="weak-scaling second figure (N > 1)")
    ap.add_argument("--keyframes", type=int, default=256)
    ap.add_argument("--no-extra", action="store_true", help="skip the other configs / the weak-scaling figure")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-corr", action="store_true")
    ap.add_argument("--corr-edges", type=int, default=256, help="edges per corr-lookup batch")
    ap.add_argument("--stub-backend", action="store_true",
                    help="plumbing test without a GPU (tests/test_bench_launch.py): CPU tensors, gloo, a compute backend "
                         "that only exercises the phase order and the collective; the line is marked invalid")
    return ap.parse_args()


class _StubBackend:
    """No arithmetic: every rank contributes rank + 1 to a tiny 'packed system' so that the all-reduce of
    ShardedBA.run is observable in the result.  Only for --stub-backend."""

    def __init__(self, rank):
        self.rank = rank
        self.reduced = None

    def prepare(self, p, t0, t1, own, motion_only):
        self.packed = torch.zeros(8, dtype=torch.float64)
        self.dx = torch.zeros((t1 - t0, 6), dtype=torch.float32)
        self.M = int(p.eta.shape[0])

    def build_packed(self, p, motion_only):
        self.packed.fill_(float(self.rank + 1))
        return self.packed

    def build(self, p, motion_only):          # world size 1: no collective
        self.reduced = float(self.rank + 1)
        return self.packed

    def unpack(self, motion_only):
        self.reduced = float(self.packed[0])

    def solve_update(self, p, lm, ep, motion_only):
        return self.dx

    def status(self):
        return 0, self.M


def spawn_ranks(args):
    """Parent of a multi-GPU run: start one rank per GPU and relay their output.  No HIP call happens in this process
    (importing torch does not initialise the runtime)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("spawning", args.gpus, "ranks:", " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def bench_ba(args, rank, world, dev, N, E, H=48, W=64, stereo=False, lm=1e-5, ep=1e-2, seed=2, steps=None,
             profile=True):
    """`steps` Gauss-Newton iterations of one `ba` call on an N-keyframe / E-edge graph (E = total over ranks)."""
    import torch.distributed as dist
    from droid_backends import ba_driver, synth

    steps = args.steps if steps is None else steps
    t_gen = time.time()
    prob = synth.make_ba_problem(N=N, E=E, H=H, W=W, stereo=stereo, lm=lm, ep=ep, seed=seed)
    log(f"[rank {rank}] generated {N} kf / {E} edges {H}x{W} in {time.time() - t_gen:.1f}s")
    ranges = ba_driver.partition_frames(prob.ii, N, world)
    sh = ba_driver.shard_problem(prob, ranges, rank)
    p = ba_driver.BAProblemDev(
        poses=to_dev(prob.poses, dev), disps=to_dev(prob.disps, dev), intrinsics=to_dev(prob.intrinsics, dev),
        disps_sens=to_dev(prob.disps_sens, dev), targets=to_dev(sh["targets"], dev),
        weights=to_dev(sh["weights"], dev), eta=to_dev(sh["eta"], dev), ii=to_dev(sh["ii"], dev),
        jj=to_dev(sh["jj"], dev))
    poses0, disps0 = p.poses.clone(), p.disps.clone()
    solver = ba_driver.ShardedBA(backend=_StubBackend(rank)) if args.stub_backend else ba_driver.ShardedBA()
    on_gpu = dev.type == "cuda"

    def reset():
        p.poses.copy_(poses0)
        p.disps.copy_(disps0)

    def barrier():
        if on_gpu:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    if args.warmup > 0:
        solver.run(p, prob.t0, prob.t1, args.warmup, prob.lm, prob.ep, own=sh["own"])
    reset()
    barrier()
    t0 = time.perf_counter()
    solver.run(p, prob.t0, prob.t1, steps, prob.lm, prob.ep, own=sh["own"])
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    st, m = solver.backend.status()
    if st & 11:
        raise RuntimeError(f"BA reported contract violation status={st}")

    info = dict(N=N, E=E, E_local=int(p.ii.shape[0]), M_local=int(p.eta.shape[0]), HW=H * W, P=prob.t1 - prob.t0,
                chol_failed=bool(st & 4), steps=steps)
    if args.stub_backend:
        info["stub_allreduce_sum"] = solver.backend.reduced
    if not profile:
        return dt, {}, info, prob
    # per-stage durations: HIP events on the launch stream, averaged over 5 iterations
    reset()
    solver.backend.prepare(p, prob.t0, prob.t1, sh["own"], False)
    stages = {}
    nprof = 5
    for _ in range(nprof):
        s = solver.backend.profile_iteration(p, prob.lm, prob.ep, False)
        for k, v in s.items():
            stages[k] = stages.get(k, 0.0) + v / nprof
    # production call shape (factor_graph.py:297): iterations=2 per call
    reset()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        solver.run(p, prob.t0, prob.t1, 2, prob.lm, prob.ep, own=sh["own"])
    barrier()
    call2_ms = (time.perf_counter() - t1) / 5 * 1e3
    info["call2_ms"] = call2_ms
    return dt, stages, info, prob


def bench_corr(args, rank, world, dev, prob):
    """Volume lookup (fp16, 4 levels, r=3) and alt-corr (half pyramid of the SLAM path; float32 pyramid) on `corr_edges` edges."""
    import torch.nn.functional as F
    import droid_backends as db
    from droid_backends import synth

    B = args.corr_edges
    H, W, r = 48, 64, 3
    fmaps, coords = synth.make_corr_inputs(prob, n_edges=B, seed=rank)
    ii = torch.from_numpy(prob.ii[:B]).to(dev)
    jj = torch.from_numpy(prob.jj[:B]).to(dev)
    fm = to_dev(fmaps, dev)  # [N,128,H,W] fp16
    c = to_dev(coords, dev)  # [B,H,W,2]
    # pyramid construction is the caller's (stock PyTorch, modules/corr.py:24-38), outside the timed region
    vols = [[] for _ in range(4)]
    for s0 in range(0, B, 32):  # all-pairs volume in chunks of 32 edges (fp32 matmul -> fp16, like autocast)
        f1 = (fm[ii[s0:s0 + 32]].float() / 4.0).reshape(-1, 128, H * W)
        f2 = (fm[jj[s0:s0 + 32]].float() / 4.0).reshape(-1, 128, H * W)
        vol = torch.matmul(f1.transpose(1, 2), f2).half().reshape(-1, 1, H, W)
        for lvl in range(4):
            vols[lvl].append(vol.view(-1, H, W, H >> lvl, W >> lvl))
            vol = F.avg_pool2d(vol.float(), 2, stride=2).half()
    pyramid = [torch.cat(x, 0).contiguous() for x in vols]
    del vols
    cq = c.permute(0, 3, 1, 2).contiguous()  # [B,2,H,W]
    cl = [(cq / 2 ** lvl).contiguous() for lvl in range(4)]

    def run_vol():
        return [db.corr_index_forward(pyramid[lvl], cl[lvl], r)[0] for lvl in range(4)]

    def timeit(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ms_vol = timeit(run_vol, 10)
    pix = B * H * W
    vol_bytes_per_pix = sum(min((2 * r + 2) ** 2, (H >> l) * (W >> l)) * 2 + 8 + (2 * r + 1) ** 2 * 2 for l in range(4))
    out = dict(volume_fp16=dict(gpix_per_s=pix / ms_vol / 1e6, ms=ms_vol, edges=B,
                                algorithmic_bytes_per_pix=vol_bytes_per_pix,
                                hbm_gbs=pix * vol_bytes_per_pix / ms_vol / 1e6,
                                frac_of_8TBs=pix * vol_bytes_per_pix / ms_vol / 1e6 / HBM_PEAK_GBS))
    # the caller's whole CorrBlock.__call__ (4 lookups at coords / 2**l + torch.cat, modules/corr.py:45-50) against the
    # one-call variant that writes the concatenated tensor directly
    def run_call():
        return torch.cat([db.corr_index_forward(pyramid[lvl], cq / 2 ** lvl, r)[0].view(B, -1, H, W) for lvl in range(4)], dim=1)
    ms_call = timeit(run_call, 5)
    ms_fused = timeit(lambda: db.corr_pyramid_forward(pyramid, cq, r), 5)
    out["volume_fp16"]["corrblock_call_ms"] = ms_call
    out["volume_fp16"]["corr_pyramid_forward_ms"] = ms_fused
    out["volume_fp16"]["corr_pyramid_forward_gpix_per_s"] = pix / ms_fused / 1e6
    # alt-corr.  What the SLAM path holds (update_lowmem, factor_graph.py:260-261): AltCorrBlock(video.fmaps[None]) with
    # video.fmaps torch.half (depth_video.py:44) => `/ 4.0` and avg_pool2d run in half and the pyramid IS half
    # (modules/corr.py:97-104); corr_fn widens the gathered per-edge copies with .float() (:120) for the fp32 kernel.
    # `altcorr_pyramid_f16` takes that half pyramid as it is (f16 matrix cores, fp32 accumulation, fp32 output).
    # The fp32 entries are the same work from a float32 pyramid (training without autocast).
    Ba = min(B, 64)
    pixa = Ba * H * W
    alt_flop_per_pix = 4 * (2 * r + 2) ** 2 * 128 * 2
    cf = c[:Ba].contiguous()
    iia_, jja_ = ii[:Ba].contiguous(), jj[:Ba].contiguous()

    def build_pyramid(x):
        pyr_ = []
        for lvl in range(4):
            pyr_.append(x.permute(0, 2, 3, 1).contiguous())
            x = F.avg_pool2d(x, 2, stride=2)
        return pyr_

    pyr_h = build_pyramid(fm / 4.0)                 # half, like AltCorrBlock.__init__ on half fmaps
    assert pyr_h[0].dtype == torch.float16
    ms_h = timeit(lambda: db.altcorr_pyramid_forward(pyr_h, cf, iia_, jja_, r), 5)
    alt16_bytes_per_pix = (128 * 2 + sum(128 * 2 / 4 ** l for l in range(4)) + 8 + 4 * (2 * r + 1) ** 2 * 4)
    out["altcorr_pyramid_f16"] = dict(
        gpix_per_s=pixa / ms_h / 1e6, ms=ms_h, edges=Ba, tflops=pixa * alt_flop_per_pix / ms_h / 1e9,
        frac_of_f16_mfma_peak=pixa * alt_flop_per_pix / ms_h / 1e9 / 2500.0,
        algorithmic_bytes_per_pix=alt16_bytes_per_pix, hbm_gbs=pixa * alt16_bytes_per_pix / ms_h / 1e6,
        frac_of_8TBs=pixa * alt16_bytes_per_pix / ms_h / 1e6 / HBM_PEAK_GBS,
        kernel="altcorr_wave_f16<3, 4, float>: half pyramid (the SLAM path's dtype) -> fp32 corr, one launch over (pyramid, ii, jj); "
               "useful flops only (the box GEMM does ~3x more); bytes = fmap1 row once + fmap2 share + coords + fp32 output")
    # the reference's call sequence on that pyramid: gather + .float() per level, then the fp32 operator (corr.py:113-120)
    def run_ref_sequence():
        return [db.altcorr_forward(pyr_h[0][iia_].float(), pyr_h[lvl][jja_].float(), (cf[:, None] / 2 ** lvl).contiguous(), r)[0]
                for lvl in range(4)]
    ms_seq = timeit(run_ref_sequence, 3)
    out["altcorr_pyramid_f16"]["reference_call_sequence_ms"] = ms_seq
    out["altcorr_pyramid_f16"]["speedup_vs_reference_call_sequence"] = ms_seq / ms_h
    pyr = build_pyramid(fm.float() / 4.0)           # float32 pyramid
    a1 = pyr[0][iia_].contiguous()
    a2 = [pyr[lvl][jja_].contiguous() for lvl in range(4)]
    ca = [(c[:Ba, None] / 2 ** lvl).contiguous() for lvl in range(4)]

    def run_alt():
        return [db.altcorr_forward(a1, a2[lvl], ca[lvl], r)[0] for lvl in range(4)]

    ms_alt = timeit(run_alt, 3)
    alt_bytes_per_pix = (4 * 128 * 4 + sum(128 * 4 / 4 ** l for l in range(4)) + 4 * 8 + 4 * (2 * r + 1) ** 2 * 4)
    out["altcorr_fp32"] = dict(gpix_per_s=pixa / ms_alt / 1e6, ms=ms_alt, edges=Ba,
                               algorithmic_bytes_per_pix=alt_bytes_per_pix,
                               tflops=pixa * alt_flop_per_pix / ms_alt / 1e9,
                               frac_of_fp32_peak=pixa * alt_flop_per_pix / ms_alt / 1e9 / 157.3,
                               hbm_gbs=pixa * alt_bytes_per_pix / ms_alt / 1e6,
                               frac_of_8TBs=pixa * alt_bytes_per_pix / ms_alt / 1e6 / 8000.0,
                               kernel="altcorr_forward_mfma<3, float, float> (fp32 MFMA, float32 maps; useful flops only, the box GEMM does 2-3x more)")
    # the same work through the fused pyramid entry point (no per-edge feature copies, one launch)
    ms_pyr = timeit(lambda: db.altcorr_pyramid_forward(pyr, cf, iia_, jja_, r), 3)
    out["altcorr_pyramid_fp32"] = dict(gpix_per_s=pixa / ms_pyr / 1e6, ms=ms_pyr, edges=Ba,
                                       tflops=pixa * alt_flop_per_pix / ms_pyr / 1e9,
                                       frac_of_fp32_peak=pixa * alt_flop_per_pix / ms_pyr / 1e9 / 157.3,
                                       kernel="altcorr_pyramid_mfma<3, float>: AltCorrBlock.corr_fn in one launch over a float32 (pyramid, ii, jj)")
    # reproject + motion features of the update operator (DepthVideo.reproject + factor_graph.py:203-205), fused:
    # per pixel 4 B disparity + 8 B target in, 8 B coords + 4 B valid + 16 B features out
    Ea = len(prob.ii)
    iia, jja = to_dev(prob.ii, dev), to_dev(prob.jj, dev)
    pz, dz, kz = to_dev(prob.poses, dev), to_dev(prob.disps, dev), to_dev(prob.intrinsics, dev)
    tg = torch.rand((Ea, H, W, 2), device=dev) * 64.0
    ms_rp = timeit(lambda: db.reproject(pz, dz, kz, iia, jja, tg), 5)
    out["reproject_motion"] = dict(gpix_per_s=Ea * H * W / ms_rp / 1e6, ms=ms_rp, edges=Ea, algorithmic_bytes_per_pix=40.0,
                                   hbm_gbs=Ea * H * W * 40.0 / ms_rp / 1e6,
                                   frac_of_8TBs=Ea * H * W * 40.0 / ms_rp / 1e6 / HBM_PEAK_GBS,
                                   kernel="reproject_motion_kernel")
    # frame_distance over ALL ordered pairs of the 256 keyframes (depth_video.py:160-190 with ii=None: the
    # proximity matrix of the global backend), one launch
    Nk = prob.disps.shape[0]
    gi, gj = torch.meshgrid(torch.arange(Nk, device=dev), torch.arange(Nk, device=dev), indexing="ij")
    gi, gj = gi.reshape(-1).contiguous(), gj.reshape(-1).contiguous()
    ms_fd = timeit(lambda: db.frame_distance(pz, dz, kz, gi, gj, 0.3), 3)
    out["frame_distance_all_pairs"] = dict(pairs=int(gi.numel()), ms=ms_fd, gpix_per_s=gi.numel() * H * W / ms_fd / 1e6,
                                           kernel="frame_distance_kernel, 65536 pairs x 3072 pixels (one direction)")
    ms_fm = timeit(lambda: db.frame_distance_matrix(pz, dz, kz, Nk, 0.3), 3)
    out["frame_distance_matrix"] = dict(pairs=int(Nk * Nk), ms=ms_fm, gpix_per_s=Nk * Nk * H * W / ms_fm / 1e6,
                                        kernel="frame_distance_matrix_kernel: DepthVideo.distance(ii=None), bidirectional, "
                                               "one launch, no index tensors (the reference makes two frame_distance calls)")
    # the two training-only operators (modules/corr.py:15-20, :82-88), level 0, a smaller batch: scatter-adds
    Bb = min(B, 32)
    vb, cb = pyramid[0][:Bb].contiguous(), cl[0][:Bb].contiguous()
    gv = torch.randn_like(db.corr_index_forward(vb, cb, r)[0])
    ms_cb = timeit(lambda: db.corr_index_backward(vb, cb, gv, r), 3)
    out["corr_index_backward_fp16_level0"] = dict(gpix_per_s=Bb * H * W / ms_cb / 1e6, ms=ms_cb, edges=Bb)
    f1b, f2b, cab = a1[:Bb].contiguous(), a2[0][:Bb].contiguous(), ca[0][:Bb].contiguous()
    ga = torch.randn_like(db.altcorr_forward(f1b, f2b, cab, r)[0])
    ms_ab = timeit(lambda: db.altcorr_backward(f1b, f2b, cab, ga, r), 3)
    out["altcorr_backward_fp32_level0"] = dict(gpix_per_s=Bb * H * W / ms_ab / 1e6, ms=ms_ab, edges=Bb)
    return out


def cpu_baseline(budget_s=15.0):
    """The oracle (CPU restatement of ba_cuda) timed on this box's host cores on the same 256-keyframe /
    2000-edge graph: one iteration to calibrate, then as many as fit ~budget_s (at most the GPU run's 16)."""
    import oracle
    from droid_backends import synth
    oracle.build()
    p = synth.make_config("cfg3")
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))

    def run(iters):
        t0 = time.perf_counter()
        oracle.ba(p.poses, p.disps, p.intrinsics, p.disps_sens, p.targets, p.weights, p.eta, p.ii, p.jj,
                  p.t0, p.t1, iters, p.lm, p.ep, False)
        return time.perf_counter() - t0

    t1 = run(1)
    iters, dt = 1, t1
    more = int(min(16, budget_s / max(t1, 1e-3)))
    if more >= 2:
        iters, dt = more, run(more)
    out = dict(value=iters / dt, unit="BA iters/s", cores=cores, kind="port", cpu_model=cpu_model(),
               sample=f"{iters} Gauss-Newton iteration(s) of the full 256-keyframe/2000-edge 48x64 graph, "
                      f"fp64 oracle (OpenMP over edges and depth frames), {dt:.1f} s")
    # single thread, bounded: one iteration of the 64-keyframe / 512-edge graph (configs[1])
    try:
        import ctypes
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        p2 = synth.make_config("cfg2")
        t0 = time.perf_counter()
        oracle.ba(p2.poses, p2.disps, p2.intrinsics, p2.disps_sens, p2.targets, p2.weights, p2.eta, p2.ii, p2.jj,
                  p2.t0, p2.t1, 1, p2.lm, p2.ep, False)
        d1 = time.perf_counter() - t0
        gomp.omp_set_num_threads(cores)
        out["single_thread"] = dict(value=1.0 / d1, unit="BA iters/s", cores=1,
                                    sample=f"1 iteration of the 64-keyframe/512-edge 48x64 graph (configs[1]), {d1:.1f} s")
    except Exception as e:  # pragma: no cover
        log("single-thread baseline skipped:", e)
    # second opinion: configs[0] (8 keyframes / 32 edges, 48x64) as a dense batched PyTorch-CPU
    # computation shaped like the reference's geom/ba.py, all host threads; checked against the C restatement
    try:
        from oracle import torch_dense_ba
        tthreads = min(cores, 16)          # beyond that the small batched ops of this formulation only contend (19 s on 256 threads)
        torch.set_num_threads(tthreads)
        pw = synth.make_ba_problem(N=4, E=8, H=8, W=8, seed=1)     # warm-up on a toy graph
        torch_dense_ba.ba_step(pw.poses, pw.disps, pw.intrinsics, pw.disps_sens, pw.targets, pw.weights, pw.eta, pw.ii, pw.jj,
                               pw.t0, pw.t1, pw.lm, pw.ep)
        p1 = synth.make_config("cfg1")
        a1 = (p1.poses, p1.disps, p1.intrinsics, p1.disps_sens, p1.targets, p1.weights, p1.eta, p1.ii, p1.jj, p1.t0, p1.t1)
        t0 = time.perf_counter()
        dx1, dz1, _ = torch_dense_ba.ba_step(*a1, p1.lm, p1.ep)
        dts = time.perf_counter() - t0
        t0 = time.perf_counter()
        o1 = oracle.ba(*a1, 1, p1.lm, p1.ep, False)
        dto = time.perf_counter() - t0
        out["second_opinion_cfg1"] = dict(
            torch_dense_iters_per_s=1.0 / dts, oracle_iters_per_s=1.0 / dto, cores=tthreads, oracle_cores=cores,
            max_abs_dx_difference=float(np.abs(dx1 - o1["dx"]).max()), max_abs_dz_difference=float(np.abs(dz1 - o1["dz"]).max()),
            sample="1 iteration of the 8-keyframe / 32-edge 48x64 graph (BASELINE configs[0]): oracle/torch_dense_ba.py "
                   f"(float64, {tthreads} threads) {dts * 1e3:.0f} ms vs the C restatement ({cores} threads) {dto * 1e3:.0f} ms")
    except Exception as e:  # pragma: no cover
        log("second-opinion baseline skipped:", e)
    # correlation lookups on the CPU: the numpy restatement (one thread), 2 edges, 4 levels each
    try:
        rng = np.random.default_rng(0)
        H, W = 48, 64
        c = np.stack([rng.uniform(0, W, (2, H, W)), rng.uniform(0, H, (2, H, W))], 1).astype(np.float32)
        vols = [rng.normal(0, 1, (2, H, W, H >> l, W >> l)).astype(np.float16) for l in range(4)]
        t0 = time.perf_counter()
        for l in range(4):
            oracle.corr_index_forward(vols[l], c / 2 ** l, 3)
        dv = time.perf_counter() - t0
        f1 = rng.normal(0, 1, (1, H, W, 128)).astype(np.float32)
        f2 = [rng.normal(0, 1, (1, H >> l, W >> l, 128)).astype(np.float32) for l in range(4)]
        ca = np.ascontiguousarray(c[:1].transpose(0, 2, 3, 1))[:, None]
        t0 = time.perf_counter()
        for l in range(4):
            oracle.altcorr_forward(f1, f2[l], ca / 2 ** l, 3, acc_dtype=np.float64)
        da = time.perf_counter() - t0
        out["corr"] = dict(volume_fp16_gpix_per_s=2 * H * W / dv / 1e9, altcorr_gpix_per_s=H * W / da / 1e9, cores=1,
                           kind="port", sample=f"numpy restatement: volume lookup 2 edges x 4 levels {dv:.2f} s, "
                                               f"alt-corr 1 edge x 4 levels {da:.2f} s")
    except Exception as e:  # pragma: no cover
        log("corr cpu baseline skipped:", e)
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main_stub(args, rank, world):
    """--stub-backend: the launch / rendezvous / collective plumbing of a multi-rank run on CPU (gloo)."""
    import torch.distributed as dist
    dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    dt, _, info, _ = bench_ba(args, rank, world, dev, args.keyframes, args.edges_total or 32, H=8, W=8, steps=args.steps,
                              profile=False)
    if rank == 0:
        print(json.dumps({"metric": "BA update iters/sec (STUB BACKEND: plumbing only, not a measurement)", "value": None,
                          "valid": False, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "config": {"workload": "stub", "edges_total": info["E"], "edges_local_rank0": info["E_local"],
                                     "stub_allreduce_sum": info.get("stub_allreduce_sum")}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))      # this process never touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         "(or without a launcher: bench.py starts its own ranks)")
    if args.stub_backend:
        return main_stub(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU fallback)"
    # rehearsal of the multi-rank path on a one-GPU box: all ranks on device 0, gloo instead of RCCL, and the
    # cooperative solver launch (two ranks' spinning grids cannot both be resident on one GPU).  Not a measurement.
    share = os.environ.get("DROID_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local = 0
        os.environ.setdefault("DROID_CHOL_COOPERATIVE", "1")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from droid_backends import synth
    E_total = args.edges_total if args.edges_total > 0 else (2000 if world == 1 else 8000)
    seed = synth.CONFIG_SEEDS["cfg3"] if E_total == 2000 else synth.CONFIG_SEEDS["cfg4"]
    dt, stages, info, prob = bench_ba(args, rank, world, dev, args.keyframes, E_total, seed=seed)
    extra = {}
    if not args.no_extra:
        if world == 1:
            # the other BASELINE configs on this GPU (parity of each: tests/test_gpu_ba*.py, test_gpu_baseline_shapes.py)
            cfgs = {}
            for name, tested in (("cfg2", "test_gpu_ba.py::test_ba_cfg2_64kf_512e"),
                                 ("cfg4", "test_gpu_ba_full.py::test_cfg4_256kf_8000e_matches_oracle (unsharded), "
                                          "test_gpu_sharded.py[cfg4like] (2 ranks)"),
                                 ("cfg5", "test_gpu_ba_full.py::test_cfg5_stereo_96x128_matches_oracle")):
                Nc, Ec, Hc, Wc, rc, stc, lmc, epc = synth.CONFIGS[name]
                d2, _, inf2, _ = bench_ba(args, rank, world, dev, Nc, Ec, Hc, Wc, stc, lmc, epc, synth.CONFIG_SEEDS[name],
                                          steps=8, profile=False)
                cfgs[name] = dict(keyframes=Nc, edges=Ec, H=Hc, W=Wc, stereo=stc, iters_per_sec=8 / d2,
                                  ms_per_iteration=d2 / 8 * 1e3, parity_tested=tested, chol_failed=inf2["chol_failed"])
                torch.cuda.empty_cache()
            extra["configs_1gpu"] = cfgs
        else:
            Ew = args.edges_per_gpu * world
            dw, _, infw, _ = bench_ba(args, rank, world, dev, args.keyframes, Ew, seed=synth.CONFIG_SEEDS["cfg3"], profile=False)
            extra["weak_scaling"] = dict(edges_per_gpu=args.edges_per_gpu, edges_total=Ew, raw_iters_per_sec=args.steps / dw,
                                         value_2000_edge_equivalents=args.steps / dw * Ew / 2000.0,
                                         ms_per_step=dw / args.steps * 1e3, scaling="weak")
            torch.cuda.empty_cache()
    corr = None
    if not args.no_corr:
        try:
            corr = bench_corr(args, rank, world, dev, prob)
        except torch.OutOfMemoryError as e:  # pragma: no cover
            log("corr bench skipped:", e)

    if world > 1:
        import torch.distributed as dist
        if corr is not None:  # edges are independent: aggregate rate = sum over ranks
            t = torch.tensor([corr["volume_fp16"]["gpix_per_s"], corr["altcorr_fp32"]["gpix_per_s"],
                              corr["altcorr_pyramid_f16"]["gpix_per_s"]],
                             dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t)
            corr["volume_fp16"]["gpix_per_s_all_ranks"] = float(t[0])
            corr["altcorr_fp32"]["gpix_per_s_all_ranks"] = float(t[1])
            corr["altcorr_pyramid_f16"]["gpix_per_s_all_ranks"] = float(t[2])

    if rank == 0:
        K = args.steps
        raw = K / dt
        value = raw * (info["E"] / 2000.0)
        # ---- rooflines.  Durations: HIP events on the launch stream (stage_ms).  Algorithmic work:
        # SURVEY.md section 8d.  `traffic`: FETCH_SIZE(x2, gfx950 wide-stream correction)+WRITE_SIZE per
        # launch from the committed rocprofv3 --pmc passes of this same command (profiles/).
        HW, E_l, M_l, Nk, P = info["HW"], info["E_local"], info["M_local"], info["N"], info["P"]
        n = 6 * P
        pmc, pmc_src = {}, None
        for name in ("r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                pmc_src = "profiles/" + name
                break
            except Exception:
                continue

        def traffic(kernel):
            k = pmc.get(kernel) or pmc.get(kernel.replace("<false>", ""))   # (r01 / r02 summaries: not yet a template)
            if not k or "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
                return None
            return (2.0 * k["FETCH_SIZE"]["mean"] + k["WRITE_SIZE"]["mean"]) * 1024.0

        # NOT measured in this run: PMC counters need their own rocprofv3 passes (tools/collect_profiles.sh)
        tsrc = (pmc_src + " (rocprofv3 --pmc passes of this command at the commit that wrote the file; "
                          "FETCH_SIZE x2 per the gfx950 128-byte-request correction, calibrated in "
                          "profiles/r02_gather16_calibration.txt)") if pmc_src else None

        def hbm(kernel, stage, alg_bytes, note):
            a = alg_bytes / (stages[stage] * 1e-3) / 1e9
            return dict(kernel=kernel, bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=a / HBM_PEAK_GBS, traffic=traffic(kernel), traffic_source=tsrc, algorithmic_bytes=alg_bytes, note=note)

        def flop(kernel, stage, flops, peak, note):
            a = flops / (stages[stage] * 1e-3) / 1e12
            return dict(kernel=kernel, bound="mfma", achieved=a, peak=peak, unit="TFLOP/s", frac=a / peak,
                        traffic=traffic(kernel), traffic_source=tsrc, algorithmic_flops=flops, note=note)

        deg = np.bincount(prob.ii, minlength=Nk)  # Schur GEMM, symmetric minimum
        nk = deg + 1
        schur_flops = float(np.sum((6 * nk) * (6 * nk + 1) * HW + 6 * nk * HW)) * (E_l / max(1, len(prob.ii)))
        # kernel names as the library dispatches them (ba_internal.hpp::ba_carve): dense graphs (mean out-degree >= 12) run
        # the linearisation that writes the E rows and the bf16x3 SYRK; few depth slots split a slot's edges over workgroups
        wide = M_l > 0 and E_l >= 12 * M_l and HW % 32 == 0
        zsplit = M_l > 0 and 1536 // (M_l * ((HW + 511) // 512)) > 1
        lin_name = "droid::ba_lin_kernel<true, %s, %s>" % ("true" if wide else "false", "true" if zsplit else "false")
        schur_name = "droid::ba_syrk3_kernel<256, 12, 1, 1, true, true>" if wide else "droid::ba_schur2_kernel"
        schur_note = ("dense-slot Schur SYRK (+ class-2 launch + fold kernel, in the stage time): symmetric-minimum fp32-equivalent flops "
                      "against the fp32 MFMA peak; executed as six v_mfma_f32_16x16x32_bf16 per 32-deep step on three-way split "
                      "operands (DESIGN.md section 7)") if wide else (
                      "Schur SYRK (+ per-slot fold kernel, in the stage time), symmetric-minimum flops, fp32 MFMA peak; the kernel "
                      "also recomputes the E rows; fp32 MFMA and VALU time add up on gfx950 (tools/micro/mfma_valu_overlap.hip)")
        kernels = [
            flop("droid::chol_factor_persistent_kernel<false>", "factor", n ** 3 / 3.0, FP64_VEC_PEAK_TFLOPS,
                 "fp64 Cholesky of the (6P)^2 reduced camera system in one launch, n^3/3 flops over ceil(n/64) "
                 "dependent block columns: a pivot-latency chain, priced against the fp64 vector/MFMA peak"),
            hbm(lin_name, "linearize", 16.0 * E_l * HW + 16.0 * M_l * HW + 56.0 * Nk,
                "compulsory bytes of one BA iteration (16*E*HW + 16*M*HW + 56*N)"),
            flop(schur_name, "schur", schur_flops, 157.3, schur_note),
            hbm("droid::ba_backsub_kernel", "update", 8.0 * E_l * HW + 16.0 * M_l * HW,
                "weights (8*E*HW) + Q, w, disps r/w (16*M*HW)"),
        ]
        dom = max((k for k in stages if k not in ("total", "unused")), key=lambda k: stages[k])
        stage_of = {"factor": 0, "linearize": 1, "schur": 2, "update": 3}
        roof = dict(kernels[stage_of.get(dom, 0)])
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            try:
                cpu = cpu_baseline()
            except Exception as e:  # pragma: no cover
                log("cpu baseline failed:", e)
        line = {
            "metric": "BA update iters/sec (2000-edge equivalents; 256 keyframes, 48x64)",
            "value": value, "unit": "iters/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak" if world == 1 else "strong",
            "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{info['N']}-keyframe / {info['E']}-edge global BA, 48x64, t0=1, lm=1e-5 ep=1e-2"
                                   + (" (BASELINE configs[2])" if (world == 1 and info["E"] == 2000) else
                                      (" (BASELINE configs[3])" if info["E"] == 8000 else "")
                                      + (f", edges sharded by source frame over {world} GPUs, 1 all-reduce/iter" if world > 1 else "")),
                       "raw_iters_per_sec": raw, "edges_total": info["E"], "edges_per_gpu": info["E"] / world,
                       "ms_per_call_iterations2": info["call2_ms"], "stage_ms": stages,
                       "solve": "fp64 dense Cholesky on device", "chol_failed": info["chol_failed"]},
            "roofline": roof,
            "rooflines": kernels,
            "cpu_baseline": cpu,
            "corr": corr,
            "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
