"""Helpers shared by the parity tests."""
import numpy as np


def quat_angle(qa, qb):
    """Rotation angle between unit quaternions (rows), radians."""
    qa = qa / np.linalg.norm(qa, axis=-1, keepdims=True)
    qb = qb / np.linalg.norm(qb, axis=-1, keepdims=True)
    d = np.abs(np.sum(qa * qb, axis=-1)).clip(0, 1)
    return 2 * np.arccos(d)


def ba_args(p):
    return (p.poses, p.disps, p.intrinsics, p.disps_sens, p.targets, p.weights, p.eta, p.ii, p.jj, p.t0, p.t1)


def to_dev(p, torch, device="cuda"):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return dict(poses=t(p.poses), disps=t(p.disps), intrinsics=t(p.intrinsics), disps_sens=t(p.disps_sens),
                targets=t(p.targets), weights=t(p.weights), eta=t(p.eta), ii=t(p.ii), jj=t(p.jj))


def run_hip_ba(backends, p, torch, iterations, motion_only=False):
    d = to_dev(p, torch)
    dx, dz = backends.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["targets"], d["weights"],
                         d["eta"], d["ii"], d["jj"], p.t0, p.t1, iterations, p.lm, p.ep, motion_only)
    torch.cuda.synchronize()
    st, m = backends.ba_status()
    return dict(poses=d["poses"].cpu().numpy(), disps=d["disps"].cpu().numpy(), dx=dx.cpu().numpy(),
                dz=dz.cpu().numpy(), status=st, M=m)


def compare_state(hip, ref, tag=""):
    """max |dt|, max rotation angle, max |ddisp| between two BA results."""
    et = np.abs(hip["poses"][:, :3] - ref["poses"][:, :3]).max()
    er = quat_angle(hip["poses"][:, 3:].astype(np.float64), ref["poses"][:, 3:].astype(np.float64)).max()
    ed = np.abs(hip["disps"] - ref["disps"]).max()
    edx = np.abs(hip["dx"] - ref["dx"]).max() if hip["dx"].size else 0.0
    print(f"[{tag}] max|dt|={et:.3e} max angle={er:.3e} max|ddisp|={ed:.3e} max|ddx|={edx:.3e}")
    return et, er, ed


def stepwise_parity(backends, oracle, p, torch, iterations, tol, tag=""):
    """PER-ITERATION parity from identical inputs, strict max-norm: iteration k of the device starts from the
    device's own float32 state after k-1 iterations, and the oracle evaluates that same iteration from the same
    state.  This is the north star's "match on identical inputs" applied to the map one Gauss-Newton iteration
    computes; a multi-iteration composite additionally carries the PROBLEM's own sensitivity (see
    `assert_composite_parity`).  Returns the device state after `iterations` iterations."""
    import copy
    q = copy.deepcopy(p)
    worst = [0.0, 0.0, 0.0]
    for k in range(iterations):
        hip = run_hip_ba(backends, q, torch, 1)
        ref = oracle.ba(*ba_args(q), 1, q.lm, q.ep, False, storage_f32=True)
        assert hip["status"] & 11 == 0 and hip["M"] == ref["M"]
        et, er, ed = compare_state(hip, ref, f"{tag} iteration {k + 1} from the device state")
        assert et < tol and er < tol and ed < tol, (tag, k, et, er, ed)
        worst = [max(a, b) for a, b in zip(worst, (et, er, ed))]
        q.poses, q.disps = hip["poses"], hip["disps"]
    return q, worst


def sensitive_disparities(oracle, p, iterations, tol, probes=2, seed=0):
    """The ill-conditioned set of a multi-iteration call, measured with the ORACLE ALONE: pixels whose result
    after `iterations` iterations moves by more than tol/4 when nothing but the float32 rounding of the state
    between iterations changes -- (a) fp64 state vs float32 state (`storage_f32`), (b) the float32 state after
    the first iteration nudged by one float32 ulp in a random direction (`probes` draws).  A weakly observed
    depth (C ~ eta, update of 100 % of its value) amplifies a 1-ulp change of the state after iteration 1 by up
    to ~1e3 in iteration 2 (cfg3 seed 12, pixel (238,1,27): 5.4e-5).  Any two float32-state evaluations (two edge
    orders on the device, the reference itself) may differ there by ~1e-4; nowhere else.  Returns a bool mask
    shaped like disps."""
    import copy
    base = oracle.ba(*ba_args(p), iterations, p.lm, p.ep, False, storage_f32=True)
    f64s = oracle.ba(*ba_args(p), iterations, p.lm, p.ep, False, storage_f32=False)
    mask = np.abs(base["disps"] - f64s["disps"]) > tol / 4
    if iterations >= 2:
        first = oracle.ba(*ba_args(p), 1, p.lm, p.ep, False, storage_f32=True)
        rng = np.random.default_rng(seed)
        for _ in range(probes):
            q = copy.deepcopy(p)
            d32 = first["disps"].astype(np.float32)
            p32 = first["poses"].astype(np.float32)
            q.disps = np.where(rng.random(d32.shape) < 0.5, np.nextafter(d32, np.float32(np.inf)),
                               np.nextafter(d32, np.float32(-np.inf))).astype(np.float32)
            q.poses = np.where(rng.random(p32.shape) < 0.5, np.nextafter(p32, np.float32(np.inf)),
                               np.nextafter(p32, np.float32(-np.inf))).astype(np.float32)
            q.poses[:p.t0] = p32[:p.t0]
            r = oracle.ba(*ba_args(q), iterations - 1, p.lm, p.ep, False, storage_f32=True)
            mask |= np.abs(r["disps"] - base["disps"]) > tol / 4
    return mask


def assert_composite_parity(hip, ref, tol, tag="", sensitive=None, hard=2e-3):
    """Composite (multi-iteration) parity: every pose and every pixel strictly within tol, EXCEPT pixels of the
    oracle-measured ill-conditioned set (`sensitive_disparities`: a mask, or a callable that computes it -- only
    called when some pixel exceeds tol), which may differ by up to `hard`.  With `sensitive` None there is no
    exception at all.  Why the set exists: two float32-state evaluations of one problem (device vs oracle with
    float32 storage, two edge orders on the device, the reference itself) round the state after iteration 1
    differently by one ulp, and a weakly observed depth amplifies that by up to ~1e3 in iteration 2; the run-to-run
    order of the device's fp64 atomics alone decides whether cfg3 seed 12 shows 6.3e-5 or 1.3e-4 at its one such
    pixel.  The outliers are printed so that drift is visible; a pixel outside the set fails the test."""
    et, er, ed = compare_state(hip, ref, tag)
    d = np.abs(hip["disps"] - ref["disps"])
    over = d > tol
    n_out = int(over.sum())
    assert et < tol and er < tol, (tag, et, er)
    if n_out == 0:
        print(f"[{tag}] disparities beyond {tol:g}: 0 of {d.size}, max {d.max():.3e}")
        return et, er, ed, 0
    assert sensitive is not None, (tag, n_out, float(d.max()))
    mask = sensitive() if callable(sensitive) else sensitive
    stray = int((over & ~mask).sum())
    where = [tuple(int(v) for v in ix) for ix in np.argwhere(over)[:8]]
    print(f"[{tag}] disparities beyond {tol:g}: {n_out} of {d.size} at {where}, {stray} outside the "
          f"ill-conditioned set ({int(mask.sum())} pixels), max {d.max():.3e}")
    assert mask.sum() <= 2e-4 * mask.size, (tag, int(mask.sum()))   # the set stays tiny
    assert stray == 0 and d.max() < hard, (tag, n_out, stray, float(d.max()))
    return et, er, ed, n_out
