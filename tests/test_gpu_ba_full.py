"""Full-size (BASELINE.json configs[2..4]) GPU tests of the HIP bundle adjustment: direct parity
against the fp64 oracle where the oracle finishes in seconds, plus size-independent properties
(edge-permutation invariance, duplicated-edge == doubled weight, cost reduction, empty graph).
Everything goes through droid_backends.ba -> C ABI.  Tolerance 1e-4 (north star)."""
import numpy as np
import pytest

from util import (assert_composite_parity, ba_args, compare_state, run_hip_ba, sensitive_disparities,
                  stepwise_parity, to_dev)

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


@pytest.fixture(scope="module")
def cfg3(synth):
    return synth.make_config("cfg3")


def _copy(p):
    import copy
    return copy.deepcopy(p)


def _parity(backends, oracle, p, iterations, tag):
    """(1) every iteration strictly within TOL of the oracle from identical inputs; (2) the composite call:
    every pose and every pixel strictly within TOL of `oracle.ba(storage_f32=True)` (SURVEY 8d's metric) -- a pixel
    beyond TOL must belong to the ill-conditioned set the oracle alone measures (tests/util.py), anything else
    fails; (3) the composite call equals the chained single-iteration calls."""
    torch = _torch()
    chained, _ = stepwise_parity(backends, oracle, p, torch, iterations, TOL, tag)
    hip = run_hip_ba(backends, p, torch, iterations)
    ref = oracle.ba(*ba_args(p), iterations, p.lm, p.ep, False, storage_f32=True)
    assert hip["status"] & 11 == 0
    assert hip["M"] == ref["M"]
    if iterations == 1:
        et, er, ed = compare_state(hip, ref, tag)
        assert et < TOL and er < TOL and ed < TOL, (et, er, ed)
    else:
        assert_composite_parity(hip, ref, TOL, tag + f" composite x{iterations}",
                                sensitive=lambda: sensitive_disparities(oracle, p, iterations, TOL))
    # atomics order is the only run-to-run freedom (1e-16 of the system), far below float32 resolution of the state
    assert np.abs(hip["poses"] - chained.poses).max() < 1e-6 and np.abs(hip["disps"] - chained.disps).max() < 1e-5


def test_cfg3_256kf_2000e_matches_oracle(backends, oracle, cfg3):
    """The graph the headline metric is quoted on, two Gauss-Newton iterations."""
    _parity(backends, oracle, _copy(cfg3), 2, "cfg3")


@pytest.mark.parametrize("seed", [12, 22, 32])
def test_cfg3_seed_sweep_margin(backends, oracle, synth, seed):
    """Three more draws of the headline graph (256 kf / 2000 e, two iterations): the margin against the 1e-4
    bar must not depend on the one seed of BASELINE configs[2] (VERDICT r01 "thin margin").  The worst of each
    metric is printed; DESIGN.md section 5 tabulates it."""
    _parity(backends, oracle, synth.make_config("cfg3", seed=seed), 2, f"cfg3 seed {seed}")


def test_cfg4_256kf_8000e_matches_oracle(backends, oracle, synth):
    """Dense graph: 31 edges per source frame on average exercises the multi-block Schur path.  Two iterations:
    SURVEY 8d's parity metric is "after ba(iterations=2)"."""
    _parity(backends, oracle, synth.make_config("cfg4"), 2, "cfg4")


def test_cfg5_stereo_96x128_matches_oracle(backends, oracle, synth):
    _parity(backends, oracle, synth.make_config("cfg5"), 2, "cfg5")


def test_dense_graph_syrk_kernels_match_oracle(backends, oracle, synth):
    """36 keyframes with ~37 edges each: every depth slot has more than 34 entries (> 208 E rows), which
    takes the path of edge-sharded ranks: the linearisation writes the E rows once and the SYRK-only
    Schur kernel (512-row variant, output tiles shared by four workgroups) streams them."""
    p = synth.make_ba_problem(N=36, E=1200, H=16, W=32, seed=77, lm=1e-4, ep=0.1)
    deg = np.bincount(p.ii, minlength=36)
    assert deg.min() >= 30 and len(p.ii) >= 12 * p.eta.shape[0]
    _parity(backends, oracle, p, 2, "dense 36kf/1200e")


def test_dense_graph_block_pair_kernel_matches_oracle(backends, oracle, synth):
    """Same kind of graph at a resolution that is not a multiple of 32 pixels: the E-row cache is
    not used and the slots (21-29 entries, up to 175 rows) go through the 96-row block-pair Schur kernel."""
    p = synth.make_ba_problem(N=30, E=720, H=15, W=20, seed=78, lm=1e-4, ep=0.1)
    assert np.bincount(p.ii, minlength=30).min() >= 17
    _parity(backends, oracle, p, 2, "dense 30kf/720e 15x20")


def test_dense_graph_with_fewer_stages_than_pixel_ranges(backends, oracle, synth):
    """Dense slots on a tiny image: 128 pixels are 4 stages of the SYRK kernels, fewer than the pixel ranges their grid
    would deal out (the split is clamped; ranges without stages write nothing and the fold skips them); slots of 17-19
    edges (SYRK class 1) next to a few of 16 or less (the sparse kernel) in one graph."""
    p = synth.make_ba_problem(N=20, E=330, H=8, W=16, seed=5, lm=1e-4, ep=0.1)
    deg = np.bincount(p.ii, minlength=20)
    assert deg.max() > 16 and 330 >= 12 * p.eta.shape[0]
    _parity(backends, oracle, p, 2, "dense 20kf/330e 8x16")


def test_dense_graph_with_every_schur_class(backends, oracle, synth):
    """One dense graph (mean out-degree >= 12) whose depth slots cover every Schur kernel: window-border frames with <= 16
    edges (the sparse kernel), the bulk with 18 (SYRK class 1: <= 256 rows), a hub with 68 edges (class 2: <= 512 rows) and a
    hub connected to every other frame (100 entries = 601 rows: the block-pair kernel, announced by the launch hint)."""
    N = 100
    pairs = set()
    for i in range(N):
        for d in range(1, 10):
            for j in (i - d, i + d):
                if 0 <= j < N:
                    pairs.add((i, j))
    for j in range(0, N, 2):
        if j != 10:
            pairs.add((10, j))
    for j in range(N):
        if j != 20:
            pairs.add((20, j))
    pairs = sorted(pairs)
    ii, jj = [a for a, _ in pairs], [b for _, b in pairs]
    p = synth.make_ba_problem(N=N, H=8, W=16, seed=11, lm=1e-4, ep=0.1, edges=(ii, jj))
    deg = np.bincount(p.ii, minlength=N)
    assert len(ii) >= 12 * p.eta.shape[0] and deg.min() <= 16 and deg[10] + 1 > 42 and 6 * (deg[10] + 1) + 1 <= 512 and deg[20] == N - 1
    _parity(backends, oracle, p, 2, "dense 100kf hubs 8x16")


@pytest.mark.parametrize("variant", ["window_inside_buffer", "rgbd", "stereo_pairs"])
def test_dense_graph_variants(backends, oracle, synth, variant):
    """The dense-slot path under the callers' other conventions: an optimisation window that starts at frame 3 inside a longer
    buffer (fixed frames are sources and targets, their pose blocks are dropped), sensor depth (RGB-D), and stereo pairs
    (edges i -> i keep their weight in the depth terms only, dk:323, :356)."""
    N = 24
    pairs = {(i, j) for i in range(N) for d in range(1, 10) for j in (i - d, i + d) if 0 <= j < N}
    kw = dict(N=N, H=8, W=32, seed=21, lm=1e-4, ep=0.1)
    if variant == "window_inside_buffer":
        kw.update(nbuf=30, t0=3)
    elif variant == "rgbd":
        kw.update(rgbd=True)
    else:
        pairs |= {(i, i) for i in range(N)}
    pairs = sorted(pairs)
    p = synth.make_ba_problem(edges=([a for a, _ in pairs], [b for _, b in pairs]), **kw)
    assert len(pairs) >= 12 * p.eta.shape[0] and np.bincount(p.ii).max() > 16
    _parity(backends, oracle, p, 2, f"dense 24kf 8x32 {variant}")


@pytest.fixture(scope="module")
def cfg3_sensitive(oracle, cfg3):
    """Ill-conditioned disparities of the headline graph over two iterations, measured with the oracle alone."""
    m = sensitive_disparities(oracle, cfg3, 2, TOL)
    print(f"cfg3: {int(m.sum())} of {m.size} disparities are sensitive to the float32 rounding of the state")
    assert m.sum() <= 2e-4 * m.size      # the set is tiny: an allowance inside it cannot hide a regression
    return m


def test_cfg3_edge_permutation_invariance(backends, cfg3, cfg3_sensitive):
    """The solution does not depend on the order of the edge list (only summation order changes)."""
    torch = _torch()
    a = run_hip_ba(backends, _copy(cfg3), torch, 2)
    q = _copy(cfg3)
    perm = np.random.default_rng(3).permutation(len(q.ii))
    q.ii, q.jj, q.targets, q.weights = q.ii[perm], q.jj[perm], q.targets[perm], q.weights[perm]
    b = run_hip_ba(backends, q, torch, 2)
    assert_composite_parity(a, b, TOL, "perm x2", sensitive=cfg3_sensitive)
    a1, b1 = run_hip_ba(backends, _copy(cfg3), torch, 1), run_hip_ba(backends, q, torch, 1)
    et, er, ed = compare_state(a1, b1, "perm x1")
    assert et < 1e-5 and er < 1e-5 and ed < 2e-5   # one iteration: only summation order differs


def test_cfg3_duplicated_edges_equal_doubled_weights(backends, oracle, cfg3):
    """Linearity of the normal equations in the weights: listing an edge twice == doubling its weight."""
    torch = _torch()
    k = 300
    a = _copy(cfg3)
    a.weights[:k] *= 2.0
    ra = run_hip_ba(backends, a, torch, 2)
    b = _copy(cfg3)
    b.ii = np.concatenate([b.ii, b.ii[:k]])
    b.jj = np.concatenate([b.jj, b.jj[:k]])
    b.targets = np.concatenate([b.targets, b.targets[:k]])
    b.weights = np.concatenate([b.weights, b.weights[:k]])
    rb = run_hip_ba(backends, b, torch, 2)
    assert_composite_parity(ra, rb, TOL, "dup x2", sensitive=sensitive_disparities(oracle, a, 2, TOL))
    ra1, rb1 = run_hip_ba(backends, a, torch, 1), run_hip_ba(backends, b, torch, 1)
    et, er, ed = compare_state(ra1, rb1, "dup x1")
    assert et < 1e-5 and er < 1e-5 and ed < 2e-5


def _cost(backends, torch, d):
    """Weighted reprojection cost of the current device state, evaluated with droid_backends.projmap."""
    coords, valid = backends.projmap(d["poses"], d["disps"], d["intrinsics"], d["ii"], d["jj"])
    r = d["targets"].permute(0, 2, 3, 1) - coords[..., :2]
    w = d["weights"].permute(0, 2, 3, 1)
    return float((w * r * r * valid).double().sum().item())   # valid: depth > MIN_DEPTH, as in the solver


def test_cfg3_iterations_reduce_reprojection_cost(backends, cfg3):
    torch = _torch()
    p = _copy(cfg3)
    d = to_dev(p, torch)
    costs = [_cost(backends, torch, d)]
    for _ in range(3):
        backends.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["targets"], d["weights"],
                    d["eta"], d["ii"], d["jj"], p.t0, p.t1, 1, p.lm, p.ep, False)
        costs.append(_cost(backends, torch, d))
    print("cost per iteration:", ["%.4e" % c for c in costs])
    assert costs[1] < 0.5 * costs[0]
    assert costs[3] <= costs[1]


def test_cfg3_k_iterations_equal_k_calls(backends, cfg3):
    """One call with K iterations == K calls with one iteration (same kernels, same order)."""
    torch = _torch()
    a = run_hip_ba(backends, _copy(cfg3), torch, 3)
    p = _copy(cfg3)
    d = to_dev(p, torch)
    for _ in range(3):
        backends.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["targets"], d["weights"],
                    d["eta"], d["ii"], d["jj"], p.t0, p.t1, 1, p.lm, p.ep, False)
    torch.cuda.synchronize()
    assert np.abs(a["poses"] - d["poses"].cpu().numpy()).max() < 1e-5
    assert np.abs(a["disps"] - d["disps"].cpu().numpy()).max() < 1e-4


def test_empty_graph_is_a_noop(backends, synth):
    """E = 0: every frame of the window still owns a depth slot (eta rows = t1 - t0); with no
    observations and disps == disps_sens... the update is exactly zero."""
    torch = _torch()
    p = synth.make_ba_problem(N=5, E=12, H=16, W=24, seed=3)
    p.ii, p.jj = p.ii[:0], p.jj[:0]
    p.targets, p.weights = p.targets[:0], p.weights[:0]
    p.eta = np.ascontiguousarray(p.eta[:p.t1 - p.t0]) if p.eta.shape[0] >= p.t1 - p.t0 else \
        np.full((p.t1 - p.t0,) + p.disps.shape[1:], 1e-3, np.float32)
    poses0, disps0 = p.poses.copy(), p.disps.copy()
    hip = run_hip_ba(backends, p, torch, 2)
    assert hip["status"] & 11 == 0
    assert np.abs(hip["poses"] - poses0).max() == 0
    assert np.abs(hip["disps"] - disps0).max() == 0
    assert np.abs(hip["dx"]).max() == 0


def test_all_weights_zero_leaves_state(backends, synth):
    torch = _torch()
    p = synth.make_config("cfg1")
    p.weights[:] = 0
    poses0, disps0 = p.poses.copy(), p.disps.copy()
    hip = run_hip_ba(backends, p, torch, 2)
    assert np.abs(hip["poses"] - poses0).max() < 1e-7
    assert np.abs(hip["disps"] - disps0).max() < 1e-7


def test_cfg3_repeated_calls_are_reproducible(backends, cfg3):
    """Twenty-four calls of two iterations from the same state: every call ends with a clean status (no failed
    factorisation, i.e. no stalled hand-off in the single-launch solve) and the same result up to the order of
    the fp64 atomics."""
    torch = _torch()
    first = None
    for rep in range(24):
        r = run_hip_ba(backends, _copy(cfg3), torch, 2)
        assert r["status"] == 0, (rep, r["status"])
        if first is None:
            first = r
        else:
            assert np.abs(r["poses"] - first["poses"]).max() < 1e-6 and np.abs(r["disps"] - first["disps"]).max() < 1e-5, rep
