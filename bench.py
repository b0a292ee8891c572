#!/usr/bin/env python3
"""bench.py -- BA update iterations/s (+ correlation-lookup Gpix/s) of the MI355X hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one Gauss-Newton iteration of `droid_backends.ba` (linearise -> depth-side reduce ->
Schur complement -> Cholesky solve -> retraction) on synthetic data that is resident in HBM when
the clock starts.  K steps are timed as ONE ba call with iterations=K (what
BASELINE.md section 3 defines: iters/s = K / wall(ba(iterations=K))).

N = 1 : BASELINE.json configs[2], the graph the metric is quoted on (256 keyframes / 2000 edges,
        48x64, lm=1e-5, ep=1e-2).
N > 1 : weak scaling of the same graph family: 256 keyframes, 2000*N edges, sharded by source
        frame (droid_backends/ba_driver.py), one all-reduce (RCCL) of the dense (6P+1)^2 fp64
        reduced camera system per iteration.  `value` is the whole-job rate in 2000-edge
        equivalents: iterations/s x (total edges / 2000); the raw iterations/s of the big graph
        is reported next to it as config.raw_iters_per_sec.

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
    ap.add_argument("--edges-per-gpu", type=int, default=2000)
    ap.add_argument("--keyframes", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-corr", action="store_true")
    ap.add_argument("--corr-edges", type=int, default=64, help="edges per corr-lookup batch")
    return ap.parse_args()


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def bench_ba(args, rank, world, dev):
    import torch.distributed as dist
    from droid_backends import ba_driver, synth

    N, E = args.keyframes, args.edges_per_gpu * world
    H, W = 48, 64
    t_gen = time.time()
    prob = synth.make_ba_problem(N=N, E=E, H=H, W=W, lm=1e-5, ep=1e-2, seed=synth.CONFIG_SEEDS["cfg3"])
    log(f"[rank {rank}] generated {N} kf / {E} edges in {time.time() - t_gen:.1f}s")
    ranges = ba_driver.partition_frames(prob.ii, N, world)
    sh = ba_driver.shard_problem(prob, ranges, rank)
    p = ba_driver.BAProblemDev(
        poses=to_dev(prob.poses, dev), disps=to_dev(prob.disps, dev), intrinsics=to_dev(prob.intrinsics, dev),
        disps_sens=to_dev(prob.disps_sens, dev), targets=to_dev(sh["targets"], dev),
        weights=to_dev(sh["weights"], dev), eta=to_dev(sh["eta"], dev), ii=to_dev(sh["ii"], dev),
        jj=to_dev(sh["jj"], dev))
    poses0, disps0 = p.poses.clone(), p.disps.clone()
    solver = ba_driver.ShardedBA()

    def reset():
        p.poses.copy_(poses0)
        p.disps.copy_(disps0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        solver.run(p, prob.t0, prob.t1, args.warmup, prob.lm, prob.ep, own=sh["own"])
    reset()
    barrier()
    t0 = time.perf_counter()
    solver.run(p, prob.t0, prob.t1, args.steps, prob.lm, prob.ep, own=sh["own"])
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    st, m = solver.backend.status()
    if st & 3:
        raise RuntimeError(f"BA reported contract violation status={st}")

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
    info = dict(N=N, E=E, E_local=int(p.ii.shape[0]), M_local=int(p.eta.shape[0]), HW=H * W, P=prob.t1 - prob.t0,
                call2_ms=call2_ms, chol_failed=bool(st & 4))
    return dt, stages, info, prob


def bench_corr(args, rank, world, dev, prob):
    """Volume lookup (fp16, 4 levels, r=3) and alt-corr (fp32, 4 levels) on `corr_edges` edges."""
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
    f1 = (fm[ii].float() / 4.0).reshape(B, 128, H * W)
    f2 = (fm[jj].float() / 4.0).reshape(B, 128, H * W)
    vol = torch.matmul(f1.transpose(1, 2), f2).half().reshape(B * H * W, 1, H, W)
    pyramid = []
    for lvl in range(4):
        pyramid.append(vol.view(B, H, W, H >> lvl, W >> lvl).contiguous())
        vol = F.avg_pool2d(vol.float(), 2, stride=2).half()
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
    # alt-corr: channels-last fp32 pyramid of pooled fmaps (modules/corr.py:92-125)
    Ba = min(B, 16)
    fml = fm.float() / 4.0
    pyr = []
    x = fml
    for lvl in range(4):
        pyr.append(x.permute(0, 2, 3, 1).contiguous())
        x = F.avg_pool2d(x, 2, stride=2)
    a1 = pyr[0][ii[:Ba]].contiguous()
    a2 = [pyr[lvl][jj[:Ba]].contiguous() for lvl in range(4)]
    ca = [(c[:Ba, None] / 2 ** lvl).contiguous() for lvl in range(4)]

    def run_alt():
        return [db.altcorr_forward(a1, a2[lvl], ca[lvl], r)[0] for lvl in range(4)]

    ms_alt = timeit(run_alt, 3)
    pixa = Ba * H * W
    alt_bytes_per_pix = (4 * 128 * 4 + sum(128 * 4 / 4 ** l for l in range(4)) + 4 * 8 + 4 * (2 * r + 1) ** 2 * 4)
    alt_flop_per_pix = 4 * (2 * r + 2) ** 2 * 128 * 2
    out["altcorr_fp32"] = dict(gpix_per_s=pixa / ms_alt / 1e6, ms=ms_alt, edges=Ba,
                               algorithmic_bytes_per_pix=alt_bytes_per_pix,
                               tflops=pixa * alt_flop_per_pix / ms_alt / 1e9,
                               frac_of_fp32_vector_peak=pixa * alt_flop_per_pix / ms_alt / 1e9 / 157.3)
    return out


def cpu_baseline(prob_small_iters=1):
    """The oracle (CPU restatement of ba_cuda) timed on this box's host cores: cfg3, one iteration."""
    import oracle
    from droid_backends import synth
    oracle.build()
    p = synth.make_config("cfg3")
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    t0 = time.perf_counter()
    oracle.ba(p.poses, p.disps, p.intrinsics, p.disps_sens, p.targets, p.weights, p.eta, p.ii, p.jj,
              p.t0, p.t1, prob_small_iters, p.lm, p.ep, False)
    dt = time.perf_counter() - t0
    return dict(value=prob_small_iters / dt, unit="BA iters/s", cores=cores, kind="port",
                sample=f"{prob_small_iters} Gauss-Newton iteration(s) of the full 256-keyframe/2000-edge 48x64 "
                       f"graph, fp64 oracle (OpenMP over edges and depth frames), {dt:.1f} s")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    dt, stages, info, prob = bench_ba(args, rank, world, dev)
    corr = None
    if not args.no_corr:
        try:
            corr = bench_corr(args, rank, world, dev, prob)
        except torch.OutOfMemoryError as e:  # pragma: no cover
            log("corr bench skipped:", e)

    if world > 1:
        import torch.distributed as dist
        if corr is not None:  # edges are independent: aggregate rate = sum over ranks
            t = torch.tensor([corr["volume_fp16"]["gpix_per_s"], corr["altcorr_fp32"]["gpix_per_s"]],
                             dtype=torch.float64, device=dev)
            dist.all_reduce(t)
            corr["volume_fp16"]["gpix_per_s_all_ranks"] = float(t[0])
            corr["altcorr_fp32"]["gpix_per_s_all_ranks"] = float(t[1])

    if rank == 0:
        K = args.steps
        raw = K / dt
        value = raw * (info["E"] / 2000.0)
        # dominant kernel group of one iteration (HIP-event stage times)
        dom = max((k for k in stages if k != "total"), key=lambda k: stages[k])
        HW, E_l, M_l, Nk, P = info["HW"], info["E_local"], info["M_local"], info["N"], info["P"]
        n = 6 * P
        if dom in ("linearize", "schur", "rhs", "update", "assemble"):
            # algorithmic (compulsory) bytes of one BA iteration, SURVEY.md section 8d
            alg = 16.0 * E_l * HW + 16.0 * M_l * HW + 56.0 * Nk
            roof = dict(kernel=dom, bound="hbm", achieved=alg / (stages[dom] * 1e-3) / 1e9, peak=HBM_PEAK_GBS,
                        unit="GB/s", traffic=None,
                        note="algorithmic bytes of one BA iteration (16*E*HW + 16*M*HW + 56*N) / duration of the "
                             "dominant kernel group")
        else:
            flops = n ** 3 / 3.0 if dom == "factor" else 2.0 * n * n
            roof = dict(kernel=dom, bound="mfma", achieved=flops / (stages[dom] * 1e-3) / 1e12,
                        peak=FP64_VEC_PEAK_TFLOPS, unit="TFLOP/s", traffic=None,
                        note="fp64 Cholesky of the (6P)^2 reduced camera system: n^3/3 flops / duration; "
                             "latency-bound chain of 2*ceil(n/64) launches, priced against the fp64 vector/MFMA peak")
        roof["frac"] = roof["achieved"] / roof["peak"]
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            try:
                cpu = cpu_baseline(1)
            except Exception as e:  # pragma: no cover
                log("cpu baseline failed:", e)
        line = {
            "metric": "BA update iters/sec (2000-edge equivalents; 256 keyframes, 48x64)",
            "value": value, "unit": "iters/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{info['N']}-keyframe / {info['E']}-edge global BA, 48x64, t0=1, lm=1e-5 ep=1e-2"
                                   + (" (BASELINE configs[2])" if world == 1 else
                                      f", edges sharded by source frame over {world} GPUs, 1 all-reduce/iter"),
                       "raw_iters_per_sec": raw, "edges_total": info["E"], "edges_per_gpu": args.edges_per_gpu,
                       "ms_per_call_iterations2": info["call2_ms"], "stage_ms": stages,
                       "solve": "fp64 dense Cholesky on device", "chol_failed": info["chol_failed"]},
            "roofline": roof,
            "cpu_baseline": cpu,
            "corr": corr,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
