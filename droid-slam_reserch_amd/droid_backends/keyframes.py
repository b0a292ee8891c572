"""The reference's on-disk keyframe dump: the data format on the input side of
the bundle-adjustment path.

Written by `Droid.save_reconstruction` (droid_slam/droid.py:92-106) and read back by the multi-session tooling
(droid_slam/loop_detect.py:209-220): one directory with eight `.npy` arrays, all cut to the first `t` keyframes,

    tstamps.npy     [t]               float32   frame timestamps                     (depth_video.py:27)
    images.npy      [t,3,ht,wd]       uint8     input images                         (:29)
    disps.npy       [t,ht/8,wd/8]     float32   inverse depths                       (:33)
    poses.npy       [t,7]             float32   world-to-camera (tx,ty,tz,qx,qy,qz,qw) (:32)
    intrinsics.npy  [t,4]             float32   fx,fy,cx,cy at 1/8 resolution        (:36)
    fmaps.npy       [t,c,128,ht/8,wd/8] float16 correlation features, c = 2 for stereo (:43)
    inps.npy        [t,128,ht/8,wd/8] float16   context features                     (:45)
    nets.npy        [t,128,ht/8,wd/8] float16   GRU hidden state                     (:44)

plus the optional `backend_finished_poses.npy` [t,7] (droid.py:108-111).  `load()` maps the big arrays instead of
reading them (a 1000-keyframe dump is ~1.5 GB of features), checks that the arrays agree with each other, and
`to_device()` hands the BA-side subset to the GPU in the layouts `droid_backends.ba` takes.  Only numpy's own
`.npy` reader is used (`allow_pickle=False`).
"""
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

FIELDS = ("tstamps", "images", "disps", "poses", "intrinsics", "fmaps", "inps", "nets")
_DTYPES = {"tstamps": np.float32, "images": np.uint8, "disps": np.float32, "poses": np.float32,
           "intrinsics": np.float32, "fmaps": np.float16, "inps": np.float16, "nets": np.float16}


class KeyframeDumpError(ValueError):
    pass


@dataclass
class KeyframeDump:
    tstamps: np.ndarray
    images: np.ndarray
    disps: np.ndarray
    poses: np.ndarray
    intrinsics: np.ndarray
    fmaps: np.ndarray
    inps: np.ndarray
    nets: np.ndarray
    backend_finished_poses: Optional[np.ndarray] = None

    @property
    def count(self) -> int:
        return int(self.tstamps.shape[0])

    @property
    def stereo(self) -> bool:
        return int(self.fmaps.shape[1]) == 2

    def validate(self):
        """Shapes and dtypes of `DepthVideo` (depth_video.py:27-45) and mutual consistency."""
        t = self.count
        for name in FIELDS:
            a = getattr(self, name)
            if a.dtype != _DTYPES[name]:
                raise KeyframeDumpError(f"{name}: dtype {a.dtype}, expected {np.dtype(_DTYPES[name])}")
            if a.shape[0] != t:
                raise KeyframeDumpError(f"{name}: {a.shape[0]} keyframes, tstamps has {t}")
        if self.tstamps.ndim != 1 or self.poses.shape != (t, 7) or self.intrinsics.shape != (t, 4):
            raise KeyframeDumpError("tstamps [t], poses [t,7], intrinsics [t,4] expected")
        if self.images.ndim != 4 or self.images.shape[1] != 3:
            raise KeyframeDumpError("images [t,3,ht,wd] expected")
        ht, wd = self.images.shape[2:]
        h8, w8 = ht // 8, wd // 8
        if self.disps.shape != (t, h8, w8):
            raise KeyframeDumpError(f"disps {self.disps.shape}, expected {(t, h8, w8)}")
        if self.fmaps.ndim != 5 or self.fmaps.shape[1] not in (1, 2) or self.fmaps.shape[2:] != (128, h8, w8):
            raise KeyframeDumpError(f"fmaps {self.fmaps.shape}, expected [t,1|2,128,{h8},{w8}]")
        for name in ("inps", "nets"):
            if getattr(self, name).shape != (t, 128, h8, w8):
                raise KeyframeDumpError(f"{name} {getattr(self, name).shape}, expected {(t, 128, h8, w8)}")
        if self.backend_finished_poses is not None and self.backend_finished_poses.shape != (t, 7):
            raise KeyframeDumpError("backend_finished_poses [t,7] expected")
        if t:
            qn = np.linalg.norm(np.asarray(self.poses[:, 3:], np.float64), axis=1)
            if not np.all(np.abs(qn - 1.0) < 1e-3):
                raise KeyframeDumpError("poses: quaternions are not normalised")
        return self

    def to_device(self, device="cuda", buffer: Optional[int] = None):
        """The BA-side state as device tensors in `droid_backends.ba` layouts: poses [nbuf,7], disps [nbuf,H,W],
        intrinsics [nbuf,4] (per frame; `intrinsics[0]` is what `ba` takes, depth_video.py:196), tstamps [nbuf];
        `buffer` pads to the video buffer size the way `DepthVideo` initialises it (identity poses, unit
        disparities)."""
        import torch
        t = self.count
        nbuf = max(t, int(buffer or t))
        poses = np.zeros((nbuf, 7), np.float32); poses[:, 6] = 1.0
        disps = np.ones((nbuf,) + self.disps.shape[1:], np.float32)
        intr = np.zeros((nbuf, 4), np.float32)
        ts = np.zeros((nbuf,), np.float32)
        poses[:t], disps[:t], intr[:t], ts[:t] = self.poses, self.disps, self.intrinsics, self.tstamps
        f = lambda a: torch.from_numpy(a).to(device)
        return {"poses": f(poses), "disps": f(disps), "intrinsics": f(intr), "tstamps": f(ts), "count": t}


def save(path, dump: KeyframeDump):
    """`Droid.save_reconstruction` (droid.py:92-106): one `.npy` per field, the first `count` keyframes."""
    dump.validate()
    os.makedirs(path, exist_ok=True)
    for name in FIELDS:
        np.save(os.path.join(path, name + ".npy"), np.ascontiguousarray(getattr(dump, name)))
    if dump.backend_finished_poses is not None:  # droid.py:108-111
        np.save(os.path.join(path, "backend_finished_poses.npy"), np.ascontiguousarray(dump.backend_finished_poses))


def load(path, mmap: bool = True) -> KeyframeDump:
    """`loop_detect.py:209-220`; the feature arrays are memory-mapped unless `mmap=False`."""
    arrays = {}
    for name in FIELDS:
        fn = os.path.join(path, name + ".npy")
        if not os.path.isfile(fn):
            raise KeyframeDumpError(f"missing {fn}")
        big = name in ("images", "fmaps", "inps", "nets")
        arrays[name] = np.load(fn, mmap_mode="r" if (mmap and big) else None, allow_pickle=False)
    fn = os.path.join(path, "backend_finished_poses.npy")
    extra = np.load(fn, allow_pickle=False) if os.path.isfile(fn) else None
    return KeyframeDump(backend_finished_poses=extra, **arrays).validate()
