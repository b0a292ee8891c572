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
        assert hip["status"] & 3 == 0 and hip["M"] == ref["M"]
        et, er, ed = compare_state(hip, ref, f"{tag} iteration {k + 1} from the device state")
        assert et < tol and er < tol and ed < tol, (tag, k, et, er, ed)
        worst = [max(a, b) for a, b in zip(worst, (et, er, ed))]
        q.poses, q.disps = hip["poses"], hip["disps"]
    return q, worst


def assert_composite_parity(hip, ref, tol, tag="", per_million=8, hard=2e-3):
    """Composite (multi-iteration) parity: poses strictly within tol; disparities within tol except for at most
    `per_million` pixels per million, none beyond `hard`.  Why an allowance: a weakly observed depth (C ~ eta,
    update of 100 % of its value) amplifies a 1-ulp float32 change of the state after iteration 1 by up to ~1e3 in
    iteration 2 -- measured with the oracle ALONE: rounding its intermediate state to float32, which the
    reference's float tensors do, moves such a pixel by 5.4e-5 (cfg3 seed 12, pixel (238,1,27)).  Any two float32
    evaluations (device vs oracle, two edge orders, the reference itself) differ there by ~1e-4; every other pixel
    and every pose is held to tol."""
    et, er, ed = compare_state(hip, ref, tag)
    d = np.abs(hip["disps"] - ref["disps"])
    n_out = int((d > tol).sum())
    allow = max(2, int(per_million * 1e-6 * d.size))
    print(f"[{tag}] disparities beyond {tol:g}: {n_out} of {d.size} (allowed {allow}), max {d.max():.3e}")
    assert et < tol and er < tol, (tag, et, er)
    assert n_out <= allow and d.max() < hard, (tag, n_out, allow, float(d.max()))
    return et, er, ed, n_out
