#!/usr/bin/env python3
"""Latency of the B1 boundary (X_decode of include/sonde_abi.h, one channel, host buffers as SDR++ hands them over):
wall time of one rs41_decode() call series per input buffer, for the buffer sizes an SDR++ stream typically delivers.
Each call that completes a 2048-sample tile does: host->device copy of the tile, one demod+FEC launch, synchronise,
fetch counts (and frames).  usage (GPU box): python tools/b1_latency.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrpp_radiosonde_amd import _lib, synth  # noqa: E402

L = _lib.load()
n = 2048 * 400
x = np.random.default_rng(1).standard_normal(n).astype(np.float32) * 0.3          # FM audio of a silent channel is noise
for name, buf in (("rs41", 2048), ("rs41", 4800), ("rs41", 1000), ("m10", 2048), ("imet4", 16384)):
    dec = getattr(L, f"{name}_decoder_init")(48000)
    sd = _lib.SondeData()
    ts = []
    for off in range(0, n - buf + 1, buf):
        b = x[off: off + buf]
        t0 = time.perf_counter()
        while getattr(L, f"{name}_decode")(dec, C.byref(sd), b.ctypes.data_as(C.c_void_p), buf) != _lib.PROCEED:
            pass
        ts.append(time.perf_counter() - t0)
    getattr(L, f"{name}_decoder_deinit")(dec)
    ts = np.array(ts[20:]) * 1e6
    print(f"{name:6s} buffers of {buf:6d} samples ({buf / 48.0:7.1f} ms of signal): per call median {np.median(ts):7.1f} us, "
          f"p99 {np.percentile(ts, 99):7.1f} us, max {ts.max():7.1f} us  -> {buf / 48e3 / (np.mean(ts) * 1e-6):7.0f} x real time")
