"""Python face of the node-level host (include/sonde_node.h, csrc/node.cpp in libsonde_rccl.so): ONE process, the GPUs of one
node, one decoder batch per GPU, the IQ of all channels arriving on one ingest device and scattered over xGMI straight into
the rows each decoder reads.  Thin: every call goes through the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FRAME_DTYPE, INPUT_IQ

TEST_SHARED_DEVICES = 0x100        # SONDE_NODE_TEST_SHARED_DEVICES (scatter_mode bit; tests/test_node_fake.py)


class SondeNodeConfig(C.Structure):
    _fields_ = [("n_devices", C.c_uint32), ("devices", C.POINTER(C.c_int32)), ("ingest", C.c_uint32), ("n_channels", C.c_uint32),
                ("types", C.POINTER(C.c_uint8)), ("max_samples", C.c_uint32), ("input_kind", C.c_int32), ("flags", C.c_uint32), ("scatter_mode", C.c_uint32)]


NODE_SYMBOLS = ["sonde_node_create", "sonde_node_shard_range", "sonde_node_destroy", "sonde_node_devices", "sonde_node_range", "sonde_node_batch", "sonde_node_submit", "sonde_node_submit_on", "sonde_node_gather_stats",
                "sonde_node_submit_local", "sonde_node_scatter_done", "sonde_node_sync", "sonde_node_frames", "sonde_node_poll",
                "sonde_node_scatter_stats", "sonde_node_last_error"]


_libs = {}


def lib(path: str | None = None):
    """libsonde_rccl.so (built on demand: csrc/Makefile `rccl`).  path: another build of csrc/node.cpp -- the test build linked against
    tests/cpp/fake_rccl.cpp (tests/test_node_fake.py); the product never passes it."""
    import os
    import subprocess
    _lib.load()
    if path is None:
        path = os.path.join(_lib.PKG_DIR, "libsonde_rccl.so")
        if not os.path.exists(path):                 # only multi-GPU hosts need RCCL
            subprocess.check_call(["make", "-s", "-C", os.path.join(_lib.PKG_DIR, "csrc"), "rccl"])
    if path not in _libs:
        _libs[path] = C.CDLL(path)
    L = _libs[path]
    if not getattr(L, "_node_ready", False):
        vp = C.c_void_p
        L.sonde_node_shard_range.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.sonde_node_shard_range.restype = None
        L.sonde_node_create.argtypes = [C.POINTER(SondeNodeConfig), C.POINTER(vp)]
        L.sonde_node_destroy.argtypes = [vp]
        L.sonde_node_destroy.restype = None
        L.sonde_node_devices.argtypes = [vp]
        L.sonde_node_devices.restype = C.c_uint32
        L.sonde_node_range.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.sonde_node_batch.argtypes = [vp, C.c_uint32]
        L.sonde_node_batch.restype = vp
        L.sonde_node_submit.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.sonde_node_submit_on.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp]
        L.sonde_node_gather_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.sonde_node_submit_local.argtypes = [vp, C.POINTER(vp), C.c_size_t, C.c_size_t]
        L.sonde_node_scatter_done.argtypes = [vp]
        L.sonde_node_sync.argtypes = [vp]
        L.sonde_node_sync.restype = C.c_long
        L.sonde_node_frames.argtypes = [vp, vp, C.c_size_t]
        L.sonde_node_frames.restype = C.c_long
        L.sonde_node_poll.argtypes = [vp, vp, C.POINTER(C.c_uint32), C.c_size_t]
        L.sonde_node_poll.restype = C.c_long
        L.sonde_node_scatter_stats.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.sonde_node_last_error.restype = C.c_char_p
        L._node_ready = True
    return L


class SondeNodeError(RuntimeError):
    pass


class SondeNode:
    """All channels of one node: `devices` (HIP ordinals) each decode a contiguous channel range; submit() takes the IQ of ALL
    channels as a device tensor [C, n, 2] on devices[ingest]."""

    def __init__(self, n_channels: int, max_samples: int, devices=(0,), ingest: int = 0, types=None, input_kind: int = INPUT_IQ, flags: int = 0, scatter_mode: int = 0,
                 lib_path: str | None = None):
        self.L = lib(lib_path)
        self._devs = (C.c_int32 * len(devices))(*devices)
        cfg = SondeNodeConfig()
        cfg.n_devices, cfg.devices, cfg.ingest = len(devices), self._devs, ingest
        cfg.n_channels, cfg.max_samples, cfg.input_kind, cfg.flags = n_channels, max_samples, input_kind, flags
        cfg.scatter_mode = scatter_mode
        self._types = None
        if types is not None:
            self._types = np.ascontiguousarray(types, dtype=np.uint8)
            assert self._types.shape == (n_channels,)
            cfg.types = self._types.ctypes.data_as(C.POINTER(C.c_uint8))
        h = C.c_void_p()
        if self.L.sonde_node_create(C.byref(cfg), C.byref(h)) != 0:
            raise SondeNodeError(self.L.sonde_node_last_error().decode())
        self.h, self.n_channels, self.devices, self.ingest = h, n_channels, tuple(devices), ingest

    def _chk(self, rc):
        if rc < 0:
            raise SondeNodeError(self.L.sonde_node_last_error().decode())
        return rc

    def range(self, d: int):
        f, c = C.c_uint32(), C.c_uint32()
        self._chk(self.L.sonde_node_range(self.h, d, C.byref(f), C.byref(c)))
        return f.value, f.value + c.value

    def submit(self, iq, stream: int | None = None):
        """iq: device tensor [C, n, 2] on the ingest device.  The scatter starts behind the work queued on `stream` (a hipStream_t
        value; default: torch's current stream on the ingest device) -- the stream that produced iq."""
        assert iq.is_cuda and iq.device.index == self.devices[self.ingest] and iq.shape[0] == self.n_channels and (iq.dim() == 2 or iq.stride(1) == 2)
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(iq.device).cuda_stream
        self._keep = (iq, getattr(self, "_keep", (None, None))[0])
        self._chk(self.L.sonde_node_submit_on(self.h, C.c_void_p(iq.data_ptr()), iq.shape[1], iq.stride(0) // (2 if iq.dim() == 3 else 1), C.c_void_p(stream)))

    def submit_local(self, rows):
        """rows[d]: device tensor [count_d, n, 2] on devices[d] (all with the same channel stride)."""
        import torch
        for r in rows:                 # the per-device decoders run on the node's own streams: the rows must be complete
            torch.cuda.current_stream(r.device).synchronize()
        arr = (C.c_void_p * len(rows))(*[r.data_ptr() for r in rows])
        self._keep = (rows, getattr(self, "_keep", (None, None))[0])
        self._chk(self.L.sonde_node_submit_local(self.h, arr, rows[0].shape[1], rows[0].stride(0) // 2))

    def scatter_done(self):
        self._chk(self.L.sonde_node_scatter_done(self.h))

    def sync(self) -> int:
        return self._chk(self.L.sonde_node_sync(self.h))

    def frames(self) -> np.ndarray:
        n = self.sync()
        out = np.zeros(n, dtype=FRAME_DTYPE)
        if n:
            out = out[:self._chk(self.L.sonde_node_frames(self.h, out.ctypes.data_as(C.c_void_p), n))]
        return out

    def poll(self, cap: int = 4096):
        out = (_lib.SondeData * cap)()
        chan = (C.c_uint32 * cap)()
        res = []
        while True:
            n = self._chk(self.L.sonde_node_poll(self.h, out, chan, cap))
            if n == 0:
                return res
            for i in range(n):
                d = _lib.SondeData()
                C.memmove(C.byref(d), C.byref(out[i]), C.sizeof(d))
                res.append((int(chan[i]), d))

    def scatter_stats(self):
        ms, by, ns = C.c_float(), C.c_uint64(), C.c_uint32()
        self._chk(self.L.sonde_node_scatter_stats(self.h, C.byref(ms), C.byref(by), C.byref(ns)))
        return {"ms": ms.value, "bytes_from_ingest": by.value, "sends": ns.value}

    def gather_stats(self):
        ms, by = C.c_double(), C.c_uint64()
        self._chk(self.L.sonde_node_gather_stats(self.h, C.byref(ms), C.byref(by)))
        return {"ms": ms.value, "bytes": by.value}

    def close(self):
        if getattr(self, "h", None):
            self.L.sonde_node_destroy(self.h)
            self.h = None

    __del__ = close
