"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/droid_backends_hip.h declares; the Python mirror exposes the reference's nine
operators (src/droid.cpp:237-250); the product never imports the oracle."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "droid_backends_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(droid_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(backends):
    lib = ctypes.CDLL(backends._lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
    assert sorted(backends._lib.SYMBOLS) == syms


def test_abi_version_and_workspace_query(backends):
    lib = backends._lib.load()
    assert lib.droid_abi_version() == 1
    small = lib.droid_ba_workspace_bytes(32, 8, 48, 64, 1, 8, 8)
    big = lib.droid_ba_workspace_bytes(2000, 256, 48, 64, 1, 256, 256)
    assert 0 < small < big
    assert lib.droid_ba_workspace_bytes(32, 8, 48, 64, 5, 5, 8) == 0  # empty window


def test_host_side_argument_checks_need_no_gpu(backends):
    lib = backends._lib.load()
    rc = lib.droid_corr_index_forward(None, None, None, 1, 8, 8, 8, 8, 3, 7, None)  # bad dtype
    assert rc == -1 and b"dtype" in lib.droid_last_error()
    rc = lib.droid_ba(None, None, None, None, None, None, None, None, None, 4, 8, 8, 8, 8, 5, 3, 1, 1e-4, 0.1, 0,
                      None, None, None, 0, None)  # t1 <= t0
    assert rc == -1 and b"window" in lib.droid_last_error()
    rc = lib.droid_reproject_motion(None, None, None, 3, None, None, None, 4, 8, 8, 8, None, None, None, None)  # bad stride
    assert rc == -1 and b"reproject_motion" in lib.droid_last_error()
    assert lib.droid_reproject_motion(None, None, None, 4, None, None, None, 0, 8, 8, 8, None, None, None, None) == 0  # E = 0
    assert lib.droid_chol_scratch_doubles(0) == 0
    n = 1530   # system rows of 128 bytes + diagonal / M tiles + one hand-over slot per lower-triangle tile + flags
    assert lib.droid_chol_scratch_doubles(n) >= (n + 1) * 1536 + (2 * 24 + 300) * 4096


def test_python_mirror_has_the_reference_operators(backends):
    for name in ["ba", "frame_distance", "projmap", "depth_filter", "iproj", "altcorr_forward",
                 "altcorr_backward", "corr_index_forward", "corr_index_backward"]:
        assert callable(getattr(backends, name))
    for name in ["altcorr_pyramid_forward", "reproject", "motion_features", "frame_distance_matrix", "corr_pyramid_forward"]:   # additions
        assert callable(getattr(backends, name))
    from droid_backends import keyframes
    assert callable(keyframes.load) and callable(keyframes.save)


def test_cpu_tensors_are_refused_not_emulated(backends):
    import pytest
    import torch
    v = torch.zeros((1, 4, 4, 4, 4))
    c = torch.zeros((1, 2, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        backends.corr_index_forward(v, c, 3)


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "droid-slam_reserch_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".sh")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)
                assert "libdroid_oracle" not in txt, os.path.join(dp, f)
