"""What the collective / solve overlap buys per iteration, emulated on ONE GPU: rank 0 of an 8-way shard of BASELINE
configs[3] (256 keyframes / 8000 edges in total) runs its real kernels; the all-reduce is replaced by a delay of T us
(default 150: the ring estimate of DESIGN section 7) -- sequential: build, delay, unpack, solve; overlapped: build, solve
launched, the delay spread over the chunks of a side stream in proportion to their bytes, each followed by its unpack.
usage: python tools/overlap_emul.py [T_us] [world]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "droid-slam_reserch_amd")]
import numpy as np, torch
from droid_backends import ba_driver, synth

T_us = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
prob = synth.make_config("cfg4")
ranges = ba_driver.partition_frames(prob.ii, prob.t1, world)
sh = ba_driver.shard_problem(prob, ranges, 0)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
p = ba_driver.BAProblemDev(poses=t(prob.poses), disps=t(prob.disps), intrinsics=t(prob.intrinsics),
                           disps_sens=t(prob.disps_sens), targets=t(sh["targets"]), weights=t(sh["weights"]),
                           eta=t(sh["eta"]), ii=t(sh["ii"]), jj=t(sh["jj"]))
poses0, disps0 = p.poses.clone(), p.disps.clone()
be = ba_driver.HipBackend()
be.prepare(p, prob.t0, prob.t1, sh["own"], False)

# calibrate torch.cuda._sleep
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(10000000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 10000000 / (e0.elapsed_time(e1) * 1e3)
delay = lambda us: torch.cuda._sleep(int(us * cyc_per_us)) if us > 0 else None

main = torch.cuda.current_stream()
side = torch.cuda.Stream()
plan = be.overlap_plan()
total_el = plan[-1][1]
K = 20


def run(overlap, epoch0):
    p.poses.copy_(poses0); p.disps.copy_(disps0)
    be.prepare(p, prob.t0, prob.t1, sh["own"], False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(K):
        be.build_packed(p, False)
        if not overlap:
            delay(T_us)
            be.unpack(False)
            be.solve_update(p, prob.lm, prob.ep, False)
            continue
        ev = torch.cuda.Event(); ev.record(main)
        assert be.solve_update_overlap(p, it + 1, False)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            for c, (a, b) in enumerate(plan):
                delay(T_us * (b - a) / total_el)
                be.unpack_chunk(c, prob.lm, prob.ep, it + 1)
        main.wait_stream(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


for _ in range(2):
    run(False, 0); run(True, 0)
a = min(run(False, 0) for _ in range(3))
b = min(run(True, 0) for _ in range(3))
st, m = be.status()
print(f"rank 0 of {world}, {len(sh['ii'])} edges, emulated all-reduce {T_us:.0f} us, {len(plan)} chunks "
      f"(elements {[q[1] - q[0] for q in plan]}): sequential {a:.3f} ms/iteration, overlapped {b:.3f} ms/iteration "
      f"(saves {1e3 * (a - b):.0f} us); status {st}")
