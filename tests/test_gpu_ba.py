"""GPU parity of the HIP bundle adjustment (through droid_backends.ba -> C ABI) against the fp64
CPU oracle on identical inputs.  Tolerance from BASELINE.json north_star: < 1e-4 on poses and
disparities (parity vs the oracle; the oracle itself is "parity unpinned", see oracle/__init__.py).
"""
import numpy as np
import pytest

from util import ba_args, compare_state, run_hip_ba, to_dev

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.fixture(scope="module")
def synth():
    from droid_backends import synth
    return synth


def _check(backends, oracle, p, iterations, motion_only=False, tol=TOL, tag=""):
    torch = _torch()
    hip = run_hip_ba(backends, p, torch, iterations, motion_only)
    ref = oracle.ba(*ba_args(p), iterations, p.lm, p.ep, motion_only, storage_f32=True)
    assert hip["status"] & 11 == 0, f"device status {hip['status']}"
    if not motion_only:
        assert hip["M"] == ref["M"]
    et, er, ed = compare_state(hip, ref, tag)
    assert et < tol and er < tol, (et, er)
    assert ed < tol, ed
    if not motion_only:
        assert np.abs(hip["dz"] - ref["dz"]).max() < 10 * tol
    return hip, ref


def test_chol_solve_matches_numpy(backends):
    torch = _torch()
    import ctypes
    lib = backends._lib.load()
    rng = np.random.default_rng(0)
    # 6..65: one block column (per-step kernels); 130..: the single-launch factorisation; 1536: the rhs row is a
    # block row of its own; 2050: 561 tiles on 256 workgroups (several tiles per workgroup and step), odd width
    for n in (6, 42, 64, 65, 130, 378, 700, 1536, 2051):
        A = rng.normal(size=(n, n + 8))
        A = A @ A.T + n * 0.1 * np.eye(n)
        b = rng.normal(size=n)
        dA, db = torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda()
        x = torch.zeros(n, dtype=torch.float64, device="cuda")
        scratch = torch.zeros(lib.droid_chol_scratch_doubles(n), dtype=torch.float64, device="cuda")
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        rc = lib.droid_chol_solve(dA.data_ptr(), db.data_ptr(), x.data_ptr(), n, scratch.data_ptr(),
                                  flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert int(flag.item()) == 0
        ref = np.linalg.solve(A, b)
        err = np.abs(x.cpu().numpy() - ref).max() / np.abs(ref).max()
        print(f"chol n={n} rel err {err:.2e}")
        assert err < 1e-10, (n, err)


def test_chol_failure_flag(backends):
    torch = _torch()
    lib = backends._lib.load()
    n = 12
    A = -np.eye(n)
    dA, db = torch.from_numpy(A).cuda(), torch.ones(n, dtype=torch.float64, device="cuda")
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    scratch = torch.zeros(lib.droid_chol_scratch_doubles(n), dtype=torch.float64, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    lib.droid_chol_solve(dA.data_ptr(), db.data_ptr(), x.data_ptr(), n, scratch.data_ptr(), flag.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1


def test_chol_failure_flag_single_launch(backends):
    """A non-positive pivot deep inside the single-launch factorisation (block column 9 of 12): the flag is
    raised, nothing hangs, and a following well-posed solve on the same scratch is unaffected."""
    torch = _torch()
    lib = backends._lib.load()
    n = 760
    rng = np.random.default_rng(5)
    A = rng.normal(size=(n, n + 8))
    A = A @ A.T + n * 0.1 * np.eye(n)
    b = rng.normal(size=n)
    bad = A.copy()
    bad[600, 600] = -1.0
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    scratch = torch.zeros(lib.droid_chol_scratch_doubles(n), dtype=torch.float64, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    db = torch.from_numpy(b).cuda()
    lib.droid_chol_solve(torch.from_numpy(bad).cuda().data_ptr(), db.data_ptr(), x.data_ptr(), n, scratch.data_ptr(),
                         flag.data_ptr(), s)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    lib.droid_chol_solve(torch.from_numpy(A).cuda().data_ptr(), db.data_ptr(), x.data_ptr(), n, scratch.data_ptr(),
                         flag.data_ptr(), s)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    ref = np.linalg.solve(A, b)
    assert np.abs(x.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-10


def test_reduced_system_matches_oracle(backends, oracle, synth):
    """Phase API: the dense (A - S | b) the device builds equals the oracle's, entry by entry."""
    torch = _torch()
    import ctypes
    lib = backends._lib.load()
    p = synth.make_config("cfg1")
    ref = oracle.ba(*ba_args(p), 1, p.lm, p.ep, False, debug=True)
    d = to_dev(p, torch)
    nbuf, H, W = p.disps.shape
    E, M, P = len(p.ii), p.eta.shape[0], p.t1 - p.t0
    nbytes = lib.droid_ba_workspace_bytes(E, nbuf, H, W, p.t0, p.t1, M)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    rc = lib.droid_ba_prepare(d["ii"].data_ptr(), d["jj"].data_ptr(), E, nbuf, H, W, M, p.t0, p.t1, 0, nbuf, 0,
                              ws.data_ptr(), nbytes, s)
    assert rc == 0
    rc = lib.droid_ba_build(d["poses"].data_ptr(), d["disps"].data_ptr(), d["intrinsics"].data_ptr(),
                            d["disps_sens"].data_ptr(), d["targets"].data_ptr(), d["weights"].data_ptr(),
                            d["eta"].data_ptr(), d["ii"].data_ptr(), d["jj"].data_ptr(), E, nbuf, H, W, M,
                            p.t0, p.t1, 0, ws.data_ptr(), nbytes, s)
    assert rc == 0
    torch.cuda.synchronize()
    nel = ctypes.c_size_t(0)
    ptr = lib.droid_ba_system(ws.data_ptr(), E, nbuf, H, W, p.t0, p.t1, M, ctypes.byref(nel))
    off = ptr - ws.data_ptr()
    n = 6 * P
    sys_ = ws[off:off + nel.value * 8].view(torch.float64).view(n + 1, -1).cpu().numpy()
    Hd = np.tril(sys_[:n, :n])
    Ho = np.tril(ref["H"])
    scale = np.abs(Ho).max()
    eh = np.abs(Hd - Ho).max() / scale
    eb = np.abs(sys_[n, :n] - ref["b"]).max() / np.abs(ref["b"]).max()
    print(f"system: rel err H {eh:.2e}  b {eb:.2e}")
    assert eh < 2e-6 and eb < 2e-6


@pytest.mark.parametrize("iterations", [1, 2])
def test_ba_cfg1_mono(backends, oracle, synth, iterations):
    _check(backends, oracle, synth.make_config("cfg1"), iterations, tag=f"cfg1 it{iterations}")


def test_ba_tiny_3kf_4e(backends, oracle, synth):
    p = synth.make_ba_problem(N=3, E=4, H=16, W=24, seed=11)
    _check(backends, oracle, p, 2, tag="3kf/4e")


def test_ba_rgbd_sensor_depth(backends, oracle, synth):
    p = synth.make_config("cfg1", rgbd=True, seed=21)
    _check(backends, oracle, p, 2, tag="cfg1 rgbd")


def test_ba_stereo_edges(backends, oracle, synth):
    p = synth.make_ba_problem(N=8, E=40, H=48, W=64, stereo=True, seed=5)
    _check(backends, oracle, p, 2, tag="stereo")


def test_ba_motion_only(backends, oracle, synth):
    """trajectory_filler.py:67-72 call shape: new frames [t0,t1) observed from fixed keyframes."""
    p = synth.make_ba_problem(N=10, E=36, H=48, W=64, seed=7)
    keep = (p.ii < 6) & (p.jj >= 6)
    p.ii, p.jj = p.ii[keep], p.jj[keep]
    p.targets, p.weights = p.targets[keep], p.weights[keep]
    p.t0, p.t1 = 6, 10
    assert len(p.ii) > 0
    _check(backends, oracle, p, 3, motion_only=True, tag="motion-only")


def test_ba_window_inside_buffer(backends, oracle, synth):
    """Frontend call shape: buffer longer than the graph, fixed source frames before t0."""
    p = synth.make_ba_problem(N=12, E=50, H=32, W=40, seed=9, nbuf=20, t0=4)
    _check(backends, oracle, p, 2, tag="window t0=4 nbuf=20")


def test_ba_ragged_resolution(backends, oracle, synth):
    """H*W not a multiple of the kernels' tile sizes."""
    p = synth.make_ba_problem(N=6, E=20, H=30, W=37, seed=13)
    _check(backends, oracle, p, 2, tag="30x37")


def test_ba_cfg2_64kf_512e(backends, oracle, synth):
    p = synth.make_config("cfg2")
    _check(backends, oracle, p, 2, tag="cfg2")


def test_ba_eta_rows_mismatch_is_reported(backends, synth):
    torch = _torch()
    p = synth.make_config("cfg1")
    p.eta = p.eta[:-1]
    hip = run_hip_ba(backends, p, torch, 1)
    assert hip["status"] & 2


def test_ba_noncontiguous_input_raises(backends, synth):
    torch = _torch()
    p = synth.make_config("cfg1")
    d = to_dev(p, torch)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        backends.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["targets"].transpose(2, 3),
                    d["weights"], d["eta"], d["ii"], d["jj"], p.t0, p.t1, 1, p.lm, p.ep, False)


def test_ba_zero_weight_edges_change_nothing(backends, synth):
    """Property (size independent): appending edges with zero weight leaves the solution unchanged."""
    torch = _torch()
    p = synth.make_config("cfg1")
    a = run_hip_ba(backends, p, torch, 2)
    q = synth.make_config("cfg1")
    q.ii = np.concatenate([p.ii, p.ii[:5]])
    q.jj = np.concatenate([p.jj, p.jj[:5]])
    q.targets = np.concatenate([p.targets, p.targets[:5] + 3.0])
    q.weights = np.concatenate([p.weights, np.zeros_like(p.weights[:5])])
    b = run_hip_ba(backends, q, torch, 2)
    assert np.abs(a["poses"] - b["poses"]).max() < 1e-6
    assert np.abs(a["disps"] - b["disps"]).max() < 1e-5


def test_ba_golden_vectors_on_device(backends, synth):
    """Committed fixtures (tests/golden/ba_golden.npz): device state after 2 iterations vs stored."""
    import os
    torch = _torch()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ba_golden.npz"))
    cases = {"tiny": synth.make_ba_problem(N=3, E=4, H=16, W=24, seed=11), "cfg1": synth.make_config("cfg1"),
             "cfg1_rgbd": synth.make_config("cfg1", rgbd=True, seed=21)}
    for name, p in cases.items():
        hip = run_hip_ba(backends, p, torch, 2)
        assert np.abs(hip["poses"] - g[f"{name}_poses"]).max() < TOL
        assert np.abs(hip["disps"] - g[f"{name}_disps"]).max() < TOL


def test_chol_single_launch_stress(backends):
    """The single-launch factorisation and back-substitution hand data between workgroups through flags and
    data-tagged slots: repeat the solve many times on varying sizes and check every result (a lost or stale
    hand-off shows up as a wrong solution or a raised failure flag)."""
    torch = _torch()
    lib = backends._lib.load()
    rng = np.random.default_rng(123)
    s = torch.cuda.current_stream().cuda_stream
    cases = []
    for n in (1530, 1531, 1600, 897, 642, 1288):
        B = rng.normal(size=(n, n // 2))
        A = B @ B.T + n * 0.05 * np.eye(n)
        b = rng.normal(size=n)
        cases.append((n, A, np.linalg.solve(A, b), torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda(),
                      torch.zeros(lib.droid_chol_scratch_doubles(n), dtype=torch.float64, device="cuda")))
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    worst = 0.0
    for rep in range(40):
        for n, A, ref, dA, db, scratch in cases:
            x = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
            lib.droid_chol_solve(dA.data_ptr(), db.data_ptr(), x.data_ptr(), n, scratch.data_ptr(), flag.data_ptr(), s)
            got = x.cpu().numpy()   # synchronises
            assert int(flag.item()) == 0, (rep, n)
            err = np.abs(got - ref).max() / np.abs(ref).max()
            worst = max(worst, err)
            assert err < 1e-9, (rep, n, err)
    print(f"240 solves, worst relative error {worst:.2e}")
