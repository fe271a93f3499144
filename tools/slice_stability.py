#!/usr/bin/env python3
"""How steady is a time-sliced launch?  Per-step device times (HIP events around every submit) of a shape over many steps, sliced
(the library's choice or forced) against unsliced: min / median / p99 / max and the slow steps.  usage (GPU box):
python tools/slice_stability.py [channels tiles slices steps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrpp_radiosonde_amd import synth
from sdrpp_radiosonde_amd.batch import SondeBatch, strided_rows

C, tiles, S, steps = (int(x) for x in (sys.argv[1:5] + ["1280", "96", "0", "600"][len(sys.argv) - 1:])[:4])
n = tiles * 2048
blocks = [strided_rows(synth.make_rs41_batch(C, n, seed=1000 + k, ebn0_db=14.0, device="cuda:0").iq) for k in range(3)]
for ts in (1, S):
    b = SondeBatch(C, n, time_slices=ts)
    s = torch.cuda.current_stream()
    for k in range(300):                       # clock ramp
        b.submit(blocks[k % 3], s.cuda_stream)
    b.sync()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record(s)
    for k in range(steps):
        b.submit(blocks[k % 3], s.cuda_stream)
        ev[k + 1].record(s)
    b.sync(); torch.cuda.synchronize()
    t = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(steps)])
    slow = np.nonzero(t > 1.5 * np.median(t))[0]
    print(f"{C} x {tiles} time_slices={ts}: min {t.min():.4f} median {np.median(t):.4f} p99 {np.percentile(t, 99):.4f} max {t.max():.4f} mean {t.mean():.4f} ms; "
          f"{len(slow)} of {steps} steps > 1.5 x median: {[(int(i), round(float(t[i]), 3)) for i in slow[:12]]}", flush=True)
    b.close()
