"""CPU oracle for the DROID-SLAM correlation-lookup + dense-BA hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import this package, and only as the checker / timed CPU baseline.  The
product (``droid-slam_reserch_amd``) never imports it and fails loudly without its HIP library.

PARITY UNPINNED: the reference (/root/reference) holds no tests, fixtures or golden vectors for
this path, its CUDA extension cannot be compiled here (no nvcc, no Eigen) and its Python twin
needs the un-vendored ``lietorch``/``torch_scatter``.  The restatement is
therefore pinned by independent derivations (autograd/finite-difference Jacobians, a dense
normal-equation solve, bilinear sampling identities) in ``tests/test_oracle_*.py`` -- not by
outputs of the reference itself.

* ``ba_oracle.c`` / ``ba_oracle_impl.h``: plain-C restatement of ``src/droid_kernels.cu``
  (``ba``, ``frame_distance``, ``projmap``, ``iproj``), fp64 ("truth") and fp32 variants.
* ``corr.py``: numpy restatement of ``src/correlation_kernels.cu`` / ``src/altcorr_kernel.cu``.
* ``torch_dense_ba.py``: one BA iteration as a dense batched PyTorch-CPU computation shaped like ``geom/ba.py`` with the
  CUDA path's constants -- the second-opinion CPU baseline of BASELINE.md section 4 and an independent pin of the C code.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdroid_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("ba_oracle.c", "ba_oracle_impl.h")
    )
    if force or src_newer:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


_DT = {"f64": (np.float64, ctypes.c_double), "f32": (np.float32, ctypes.c_float)}


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _prep(arr, dt):
    return np.ascontiguousarray(np.asarray(arr), dtype=dt)


def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations,
       lm, ep, motion_only, precision="f64", debug=False, storage_f32=False):
    """Restatement of ``ba_cuda`` (src/droid_kernels.cu:1314-1434).

    ``storage_f32``: round ``dx``, ``dz`` and the updated ``poses`` / ``disps`` to float32 after every
    iteration -- the dtypes of the reference's tensors (dk:1202-1212, :1417, :898-946) -- while all
    arithmetic stays in ``precision``.  This is the mode the multi-iteration parity tests use: weakly
    observed depths amplify a 1-ulp (fp32) change of the intermediate state by up to ~1e3, so a restatement
    that carries the state in fp64 between iterations differs from ANY float32-state implementation,
    the reference included, by up to 5e-5 on the 256-keyframe graphs (tests/test_oracle_ba.py).

    Inputs are not modified; returns a dict with the updated ``poses`` / ``disps`` (whole buffers),
    ``dx`` [P,6], ``dz`` [M,HW], ``kx`` [M] and, with ``debug``, the first iteration's dense
    ``H`` (= A - S, no damping), ``b``, per-edge ``Hs`` [E,4,6,6] and ``vs`` [E,2,6].
    """
    npdt, _ = _DT[precision]
    L = lib()
    fn = getattr(L, f"droid_oracle_ba_{precision}")
    fn.restype = ctypes.c_int
    poses = _prep(poses, npdt).copy()
    disps = _prep(disps, npdt).copy()
    nbuf, ht, wd = disps.shape
    intr = _prep(intrinsics, npdt)
    sens = _prep(disps_sens, npdt)
    tg = _prep(targets, npdt)
    wt = _prep(weights, npdt)
    et = _prep(eta, npdt).reshape(-1, ht * wd) if eta is not None and np.size(eta) else np.zeros((0, ht * wd), npdt)
    ii = _prep(ii, np.int64)
    jj = _prep(jj, np.int64)
    E = int(ii.shape[0])
    P = t1 - t0
    dx = np.zeros((max(P, 0), 6), npdt)
    dz = np.zeros((max(et.shape[0], 1), ht * wd), npdt)
    kx = np.zeros((nbuf,), np.int64)
    M = ctypes.c_int(0)
    dH = np.zeros((6 * P, 6 * P), np.float64) if debug else None
    db = np.zeros((6 * P,), np.float64) if debug else None
    dHs = np.zeros((E, 4, 6, 6), npdt) if debug else None
    dvs = np.zeros((E, 2, 6), npdt) if debug else None
    L.droid_oracle_set_storage_f32(ctypes.c_int(1 if storage_f32 else 0))
    try:
        rc = _call_ba(fn, poses, disps, intr, sens, tg, wt, et, ii, jj, E, nbuf, ht, wd, t0, t1, iterations, lm, ep,
                      motion_only, dx, dz, kx, M, dH, db, dHs, dvs)
    finally:
        L.droid_oracle_set_storage_f32(ctypes.c_int(0))
    if rc != 0:
        raise RuntimeError(f"droid_oracle_ba: contract violation rc={rc}")
    out = dict(poses=poses, disps=disps, dx=dx, dz=dz[: M.value], kx=kx[: M.value].copy(), M=M.value)
    if debug:
        out.update(H=dH, b=db, Hs=dHs, vs=dvs)
    return out


def _call_ba(fn, poses, disps, intr, sens, tg, wt, et, ii, jj, E, nbuf, ht, wd, t0, t1, iterations, lm, ep,
             motion_only, dx, dz, kx, M, dH, db, dHs, dvs):
    return fn(_p(poses), _p(disps), _p(intr), _p(sens), _p(tg), _p(wt), _p(et),
            ctypes.c_int(et.shape[0]), _p(ii), _p(jj), ctypes.c_int(E), ctypes.c_int(nbuf),
            ctypes.c_int(ht), ctypes.c_int(wd), ctypes.c_int(t0), ctypes.c_int(t1),
            ctypes.c_int(iterations), ctypes.c_double(lm), ctypes.c_double(ep),
            ctypes.c_int(1 if motion_only else 0), _p(dx), _p(dz), _p(kx), ctypes.byref(M),
            _p(dH), _p(db), _p(dHs), _p(dvs))


class BAPhases:
    """Two-phase form (build -> [sum over ranks] -> finish) used by the world_size-2 tests."""

    def __init__(self, precision="f64"):
        self.precision = precision
        self.npdt = _DT[precision][0]
        self.handle = None

    def build(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1,
              own0, own1, motion_only):
        L = lib()
        fn = getattr(L, f"droid_oracle_ba_build_{self.precision}")
        fn.restype = ctypes.c_void_p
        npdt = self.npdt
        nbuf, ht, wd = np.asarray(disps).shape
        self.poses = _prep(poses, npdt).copy()
        self.disps = _prep(disps, npdt).copy()
        et = _prep(eta, npdt).reshape(-1, ht * wd)
        ii = _prep(ii, np.int64)
        jj = _prep(jj, np.int64)
        P = t1 - t0
        H = np.zeros((6 * P, 6 * P), np.float64)
        b = np.zeros((6 * P,), np.float64)
        rc = ctypes.c_int(0)
        args = [_prep(intrinsics, npdt), _prep(disps_sens, npdt), _prep(targets, npdt), _prep(weights, npdt)]
        self.handle = fn(_p(self.poses), _p(self.disps), _p(args[0]), _p(args[1]), _p(args[2]),
                         _p(args[3]), _p(et), ctypes.c_int(et.shape[0]), _p(ii), _p(jj),
                         ctypes.c_int(ii.shape[0]), ctypes.c_int(nbuf), ctypes.c_int(ht),
                         ctypes.c_int(wd), ctypes.c_int(t0), ctypes.c_int(t1), ctypes.c_int(own0),
                         ctypes.c_int(own1), ctypes.c_int(1 if motion_only else 0), _p(H), _p(b),
                         ctypes.byref(rc))
        if not self.handle:
            raise RuntimeError(f"droid_oracle_ba_build: rc={rc.value}")
        self.P = P
        return H, b

    def finish(self, H, b, lm, ep):
        L = lib()
        fn = getattr(L, f"droid_oracle_ba_finish_{self.precision}")
        fn.restype = ctypes.c_int
        H = np.ascontiguousarray(H, np.float64).copy()
        b = np.ascontiguousarray(b, np.float64)
        dx = np.zeros((self.P, 6), self.npdt)
        fn(ctypes.c_void_p(self.handle), _p(H), _p(b), ctypes.c_double(lm), ctypes.c_double(ep),
           _p(self.poses), _p(self.disps), _p(dx))
        self.handle = None
        return self.poses, self.disps, dx


def linearize_edge(target, weight, poses, disps, intrinsics, ix, jx, precision="f64"):
    """projective_transform_kernel for one edge (src/droid_kernels.cu:176-424)."""
    npdt, _ = _DT[precision]
    fn = getattr(lib(), f"droid_oracle_linearize_edge_{precision}")
    fn.restype = None
    disps = _prep(disps, npdt)
    _, ht, wd = disps.shape
    HW = ht * wd
    Hs = np.zeros((4, 6, 6), npdt)
    vs = np.zeros((2, 6), npdt)
    Eii = np.zeros((6, HW), npdt)
    Eij = np.zeros((6, HW), npdt)
    Cii = np.zeros((HW,), npdt)
    bz = np.zeros((HW,), npdt)
    a = [_prep(target, npdt), _prep(weight, npdt), _prep(poses, npdt), disps, _prep(intrinsics, npdt)]
    fn(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), ctypes.c_int(ix), ctypes.c_int(jx),
       ctypes.c_int(ht), ctypes.c_int(wd), _p(Hs), _p(vs), _p(Eii), _p(Eij), _p(Cii), _p(bz))
    return dict(Hs=Hs, vs=vs, Eii=Eii, Eij=Eij, Cii=Cii, bz=bz)


def frame_distance(poses, disps, intrinsics, ii, jj, beta, precision="f64"):
    npdt, _ = _DT[precision]
    fn = getattr(lib(), f"droid_oracle_frame_distance_{precision}")
    fn.restype = None
    disps = _prep(disps, npdt)
    _, ht, wd = disps.shape
    ii = _prep(ii, np.int64)
    jj = _prep(jj, np.int64)
    dist = np.zeros((ii.shape[0],), npdt)
    a = [_prep(poses, npdt), _prep(intrinsics, npdt)]
    fn(_p(a[0]), _p(disps), _p(a[1]), _p(ii), _p(jj), ctypes.c_int(ii.shape[0]), ctypes.c_int(ht),
       ctypes.c_int(wd), ctypes.c_double(beta), _p(dist))
    return dist


def projmap(poses, disps, intrinsics, ii, jj, precision="f64"):
    npdt, _ = _DT[precision]
    fn = getattr(lib(), f"droid_oracle_projmap_{precision}")
    fn.restype = None
    disps = _prep(disps, npdt)
    _, ht, wd = disps.shape
    ii = _prep(ii, np.int64)
    jj = _prep(jj, np.int64)
    E = ii.shape[0]
    coords = np.zeros((E, ht, wd, 3), npdt)
    valid = np.zeros((E, ht, wd, 1), npdt)
    a = [_prep(poses, npdt), _prep(intrinsics, npdt)]
    fn(_p(a[0]), _p(disps), _p(a[1]), _p(ii), _p(jj), ctypes.c_int(E), ctypes.c_int(ht),
       ctypes.c_int(wd), _p(coords), _p(valid))
    return coords, valid


def iproj(poses, disps, intrinsics, precision="f64"):
    npdt, _ = _DT[precision]
    fn = getattr(lib(), f"droid_oracle_iproj_{precision}")
    fn.restype = None
    disps = _prep(disps, npdt)
    nm, ht, wd = disps.shape
    pts = np.zeros((nm, ht, wd, 3), npdt)
    a = [_prep(poses, npdt), _prep(intrinsics, npdt)]
    fn(_p(a[0]), _p(disps), _p(a[1]), ctypes.c_int(nm), ctypes.c_int(ht), ctypes.c_int(wd), _p(pts))
    return pts


def retr(xi, t, q, precision="f64"):
    """retrSE3 (src/droid_kernels.cu:877-895)."""
    npdt, _ = _DT[precision]
    fn = getattr(lib(), f"droid_oracle_retr_{precision}")
    fn.restype = None
    t1 = np.zeros(3, npdt)
    q1 = np.zeros(4, npdt)
    a = [_prep(xi, npdt), _prep(t, npdt), _prep(q, npdt)]
    fn(_p(a[0]), _p(a[1]), _p(a[2]), _p(t1), _p(q1))
    return t1, q1


from .corr import altcorr_forward, corr_index_backward, corr_index_forward  # noqa: E402,F401
