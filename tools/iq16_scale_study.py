#!/usr/bin/env python3
"""How much of the 16 bits does SONDE_INPUT_IQ16 need?  RS41 / DFM / M10 channels at a working SNR, quantised with the unit-amplitude
signal at `scale` counts (the noise at 10-14 dB Eb/N0 is 2-3 times that): FEC-clean frames against the float path on the unquantised
signal.  usage: python tools/iq16_scale_study.py  (GPU box)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdrpp_radiosonde_amd import _lib, synth  # noqa: E402
from sdrpp_radiosonde_amd.batch import SondeBatch  # noqa: E402

TILE = 2048
C, n = 32, 96 * TILE


def clean(frames):
    return int(((frames["nerr"] >= 0).all(axis=1)).sum())


print("| sonde, Eb/N0 | float | " + " | ".join(f"scale {s}" for s in (8192, 512, 64, 16, 8, 4, 2)) + " |")
print("|---|---|" + "---|" * 7)
for stype, eb in ((0, 10.0), (0, 12.0), (1, 10.0), (3, 14.0)):
    sb = synth.make_batch(stype, C, n, seed=5 + stype, ebn0_db=eb, device="cuda", cfo_max_hz=1000.0)
    types = np.full(C, stype, dtype=np.uint8)
    b = SondeBatch(C, n, types=types)
    b.submit(sb.iq)
    row = [clean(b.frames())]
    b.close()
    for scale in (8192, 512, 64, 16, 8, 4, 2):
        q = torch.clamp(torch.round(sb.iq * float(scale)), -32768, 32767).to(torch.int16)
        b = SondeBatch(C, n, types=types, input_kind=_lib.INPUT_IQ16)
        b.submit(q)
        row.append(clean(b.frames()))
        b.close()
    print(f"| {('RS41', 'DFM09', 'iMS-100', 'M10')[stype]}, {eb:.0f} dB | " + " | ".join(str(x) for x in row) + " |")
