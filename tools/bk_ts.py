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
for i, nm in ((0, "start"), (1, "loads issued, tables"), (5, "passes 0-2"), (2, "history / ring words in"), (6, "tile 0 rounds"), (10, "roll, passes 3-5"), (11, "tile 1 rounds"),
              (13, "roll, passes 6-7"), (14, "tile 2 rounds"), (15, "roll, state saved"), (16, "end")):
    print(f"{nm:26s} at {t[i] - t[0]:6d}")
