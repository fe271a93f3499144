#!/usr/bin/env python3
"""Regenerates tests/golden/rs41_golden.npz.

The reference ships no golden vectors for this path, so this fixture is the
repo's own: 8-bit quantised synthetic RS41 IQ (like an RTL-SDR capture) plus what the CPU oracle
made of it at the time the fixture was cut.  It pins the oracle against silent drift and gives
the GPU tests a second, committed, anchor.  Inputs and expected outputs only -- no reference code.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from sdrpp_radiosonde_amd import synth  # noqa: E402
import oracle_lib  # noqa: E402

C, TILES = 3, 34
n = 2048 * TILES
sb = synth.make_rs41_batch(C, n, seed=2024, ebn0_db=11.0, amp_range=(0.5, 0.9))
q = np.clip(np.round(sb.iq.numpy() * 100.0), -127, 127).astype(np.int8)       # 8-bit capture
iq = q.astype(np.float32) / np.float32(100.0)
chs, bits, states = [], [], []
for c in range(C):
    ch = oracle_lib.Channel(0, c)
    ch.feed(iq[c])
    chs.append(ch)
    bits.append(np.packbits(ch.bits(), bitorder="little"))
    s = ch.state()
    states.append([s["t_next"], s["period"], np.float32(s["bias"]).view(np.int32), np.float32(s["amp"]).view(np.int32),
                   np.float32(s["yprev"]).view(np.int32), len(ch.bits())])
frames = np.concatenate([ch.frames() for ch in chs])
tx = np.stack([f for lst in sb.frames for (_, f) in lst])
np.savez_compressed(os.path.join(HERE, "rs41_golden.npz"), iq_int8=q, bits=np.stack(bits), states=np.array(states, dtype=np.int64),
                    frames=frames.view(np.uint8).reshape(len(frames), -1), tx_frames=tx)
print("frames", len(frames), "nerr", frames["nerr"].tolist(), "bytes", os.path.getsize(os.path.join(HERE, "rs41_golden.npz")))
