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
