"""droid-slam_reserch_amd: MI355X-native correlation-lookup + dense-BA hot path of DROID-SLAM.

The importable drop-in module is `droid_backends` (this directory on sys.path); this package
init only makes `importlib.import_module("droid-slam_reserch_amd")` work from the repo root.
"""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
if _here not in _sys.path:
    _sys.path.insert(0, _here)

import droid_backends  # noqa: E402,F401
