#!/usr/bin/env python3
"""Regenerates tests/golden/others_golden.npz: for each sonde type 1..6 one channel of 8-bit quantised synthetic IQ (like an
RTL-SDR capture), the bits the CPU oracle demodulated and the frames it decoded at the time the fixture was cut.  The
reference ships no vectors for this path: the fixture is the repo's own, it pins the oracle against
silent drift and gives the GPU tests a committed anchor for every decoder.  Inputs and expected outputs only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from sdrpp_radiosonde_amd import synth  # noqa: E402
import oracle_lib  # noqa: E402

TILES = {1: 24, 2: 24, 3: 16, 4: 48, 5: 24, 6: 40}          # iMet / C50: multiples of 8 tiles (16384 samples)
SNR = {1: 10.5, 2: 10.0, 3: 12.5, 4: 7.0, 5: 7.0, 6: 12.5}  # low enough that corrections / rejects occur
out = {}
for t, tiles in TILES.items():
    n = 2048 * tiles
    sb = synth.make_batch(t, 1, n, seed=4040 + t, ebn0_db=SNR[t], amp_range=(0.6, 0.9))
    q = np.clip(np.round(sb.iq.numpy()[0] * 100.0), -127, 127).astype(np.int8)
    iq = q.astype(np.float32) / np.float32(100.0)
    ch = oracle_lib.Channel(t, 0)
    ch.feed(iq)
    fr = ch.frames()
    out[f"iq{t}"] = q
    out[f"bits{t}"] = np.packbits(ch.bits(), bitorder="little")
    out[f"nbits{t}"] = np.array([len(ch.bits())], dtype=np.int64)
    out[f"frames{t}"] = fr.view(np.uint8).reshape(len(fr), -1)
    print("type", t, "frames", len(fr), "nerr", fr["nerr"].tolist()[:8])
path = os.path.join(HERE, "others_golden.npz")
np.savez_compressed(path, **out)
print("bytes", os.path.getsize(path))
