"""ctypes view of the .klg RGB-D log reader / writer (co_fusion_amd/host/KlgIO.cpp; format of the reference's
GUI/Tools/KlgLogReader.cpp:22-87).  Host-only code: works without a GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _libmod


class KlgError(RuntimeError):
    pass


def _host():
    return _libmod.load_host()


class KlgReader:
    def __init__(self, path, width, height, flip_colors=False):
        self.lib = _host()
        self.h = C.c_void_p()
        n = C.c_int()
        if self.lib.cofusion_klg_open(str(path).encode(), width, height, int(flip_colors), C.byref(self.h), C.byref(n)) != 0:
            raise KlgError(self.lib.cofusion_last_error().decode())
        self.num_frames, self.width, self.height = n.value, width, height

    def __iter__(self):
        return self

    def __next__(self):
        depth = np.empty((self.height, self.width), np.float32)
        rgb = np.empty((self.height, self.width, 3), np.uint8)
        ts = C.c_int64()
        rc = self.lib.cofusion_klg_next(self.h, C.byref(ts), depth.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p))
        if rc == 1:
            raise StopIteration
        if rc != 0:
            raise KlgError(self.lib.cofusion_last_error().decode())
        return ts.value, depth, rgb

    def close(self):
        if self.h:
            self.lib.cofusion_klg_close(self.h)
            self.h = None

    def __del__(self):
        self.close()


class KlgWriter:
    def __init__(self, path, width, height, compress_depth=True):
        self.lib = _host()
        self.h = C.c_void_p()
        if self.lib.cofusion_klg_create(str(path).encode(), width, height, int(compress_depth), C.byref(self.h)) != 0:
            raise KlgError(self.lib.cofusion_last_error().decode())
        self.width, self.height = width, height

    def write(self, timestamp, depth_m, rgb):
        d = np.ascontiguousarray(depth_m, np.float32)
        c = np.ascontiguousarray(rgb, np.uint8)
        assert d.shape == (self.height, self.width) and c.shape == (self.height, self.width, 3)
        if self.lib.cofusion_klg_write(self.h, C.c_int64(int(timestamp)), d.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)) != 0:
            raise KlgError(self.lib.cofusion_last_error().decode())

    def close(self):
        if self.h:
            self.lib.cofusion_klg_finish(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        self.close()
