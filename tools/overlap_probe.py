import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ.get("GRAFT_REPO_ROOT", ".")
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
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
plan = be.overlap_plan(); tot = plan[-1][1]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(10000000); e1.record(); torch.cuda.synchronize()
cpu = 10000000 / (e0.elapsed_time(e1) * 1e3)
T = 300.0
HOLD = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for rep in range(3):
    be.build_packed(p, False)
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True); start.record(main)
    ev = torch.cuda.Event(); ev.record(main)
    assert be.solve_update_overlap(p, rep + 1, False)
    done_main = torch.cuda.Event(enable_timing=True); done_main.record(main)
    marks = []
    with torch.cuda.stream(side):
        side.wait_event(ev)
        for c, (a, b) in enumerate(plan):
            torch.cuda._sleep(int((2000.0 if c == HOLD else 5.0) * cpu))
            be.unpack_chunk(c, prob.lm, prob.ep, rep + 1)
            m = torch.cuda.Event(enable_timing=True); m.record(side); marks.append(m)
    torch.cuda.synchronize()
    if os.environ.get("DROID_HIP_LIB"):
        import ctypes
        buf = (ctypes.c_ulonglong * 64)()
        be.lib.droid_debug_overlap_stamps(buf)
        st = np.array(buf[:], dtype=np.int64)
        print("diagonal tiles factored at (us after kernel start):", [round((int(v) - int(st[63])) / 100) for v in st[:24]])
    print("chunks done at (us):", [round(start.elapsed_time(m) * 1e3) for m in marks], " solve+update done at", round(start.elapsed_time(done_main) * 1e3), "status", be.status())
