#!/usr/bin/env python3
"""Cycle stamps of one bins-decoder wave (library built with `make EXTRA=-DBK_TS`): where a wave's 30 us go."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrpp_radiosonde_amd import synth, _lib
from sdrpp_radiosonde_amd.batch import SondeChannelizer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ch = SondeChannelizer(n_streams=S)
n = ch.samples_per_submit
x = torch.randn((S, n, 2), device="cuda:0")
for k in range(6):
    ch.submit(x)
torch.cuda.synchronize()
L = _lib.load()
ts = (C.c_ulonglong * 64)()
assert L.sonde_debug_bins_ts(ts) == 0
t = np.array(list(ts), dtype=np.int64)
names = {0: "start", 1: "prologue issued", 2: "barrier", 3: "pass0", 4: "pass1", 5: "pass2", 6: "tile0 rounds", 7: "roll", 8: "pass3", 9: "pass4", 10: "pass5",
         11: "tile1 rounds", 12: "pass6", 13: "pass7", 14: "tile2 rounds", 15: "state saved", 16: "end"}
prev = t[0]
for i in range(17):
    print(f"{i:2d} {names[i]:16s} +{t[i] - prev:7d}  (at {t[i] - t[0]})")
    prev = t[i]
print("prologue: kernargs", t[30]-t[0], "g_comp", t[31]-t[30], "phases", t[32]-t[31], "state", t[33]-t[32], "hist", t[34]-t[33], "fstate", t[35]-t[34], "rest (taps)", t[1]-t[35])
print("last tile: front", t[20] - t[13], "reduce", t[21] - t[20], "ring", t[22] - t[21], "loop filter", t[23] - t[22], "k4", t[14] - t[23])
