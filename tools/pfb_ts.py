#!/usr/bin/env python3
"""Cycle stamps of one filter-bank workgroup (library built with `make EXTRA=-DP_TS`): where a workgroup's life goes.
usage: python tools/pfb_ts.py [streams]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrpp_radiosonde_amd import _lib
from sdrpp_radiosonde_amd.batch import SondeChannelizer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ch = SondeChannelizer(n_streams=S)
n = ch.samples_per_submit
x = torch.randn((S, n, 2), device="cuda:0")
for k in range(6):
    ch.submit(x)
torch.cuda.synchronize()
L = _lib.load()
ts = (C.c_ulonglong * 128)()
assert L.sonde_debug_pfb_ts(ts) == 0
t = np.array(list(ts), dtype=np.int64).reshape(16, 8)       # [stamp][wave]
names = ["start", "loads issued, taps in LDS", "barrier", "fold done", "rotated + barrier", "FFT done", "atan done", "barrier", "tile + barrier", "stored"]
t0 = t[0].min()
print("streams", S)
for i, nm in enumerate(names):
    print(f"{nm:28s} wave0 {t[i][0] - t0:6d}  waves min {t[i].min() - t0:6d} max {t[i].max() - t0:6d}")
