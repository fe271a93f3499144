#!/usr/bin/env python3
"""Cuts tests/golden/tx_<sonde>.npz: TRANSMITTED-frame fixtures.

Unlike rs41_golden.npz / others_golden.npz (8-bit IQ + what the ORACLE made of it: drift pins that are re-cut whenever the
SPEC moves), nothing in these files comes from a decoder: each holds 8-bit synthetic IQ (an RTL-SDR-class capture) of a few
channels at two signal-to-noise ratios and the frame bytes the GENERATOR put on the air (sdrpp_radiosonde_amd/synth.py shares
no code with either decoder).  The test (tests/test_tx_fixtures.py) holds every decoder -- the oracle on CPU, the HIP path on
the GPU -- to a floor: at least `floor` of the transmitted frames come back FEC-clean, and every FEC-clean frame is one of the
transmitted ones.  DO NOT regenerate these files when the SPEC changes: a SPEC change must still meet the floors that were cut
with the decoder of round 5 (floor = 90 % of what that decoder delivered, rounded down).  Run once, in the build container:
    python tests/golden/make_tx_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from sdrpp_radiosonde_amd import synth  # noqa: E402
import oracle_lib  # noqa: E402
import tx_common  # noqa: E402

# (tiles, (high, low) Eb/N0 -- the AFSK sondes: carrier-to-noise in 48 kHz)
CASES = {0: (48, (16.0, 10.5)), 1: (24, (16.0, 10.0)), 2: (24, (16.0, 10.5)), 3: (16, (18.0, 13.0)),
         4: (48, (14.0, 8.0)), 5: (24, (14.0, 8.0)), 6: (48, (16.0, 12.0))}
C = 2
# The AFSK sondes' timing loop closes 3 (iMet) / 6 (C50) times per second: a channel whose symbol clock starts out within a few
# percent of half a symbol off the loop's initial guess takes longer to acquire than these 2-second clips last (found while cutting
# these fixtures).  The clips draw their channels outside those zones -- a property of the generator's timing offset,
# not of any decoder's output.
SLOW = {4: (0.95, 0.05), 5: (0.45, 0.70)}


def slow_zone(t, tau):
    if t not in SLOW:
        return False
    lo, hi = SLOW[t]
    return bool(np.any((tau >= lo) & (tau <= hi)) if lo < hi else np.any((tau >= lo) | (tau <= hi)))


def main():
    for t, (tiles, snrs) in CASES.items():
        n = 2048 * tiles
        out = {"snr": np.array(snrs, dtype=np.float32)}
        for k, snr in enumerate(snrs):
            seed = 7700 + 10 * t + k
            nch = 3 if t == 0 else C
            sb = synth.make_batch(t, nch, n, seed=seed, ebn0_db=snr, **({} if t in (4, 5) else {"amp_range": (0.6, 0.9)}))
            while slow_zone(t, sb.tau):
                seed += 100
                sb = synth.make_batch(t, nch, n, seed=seed, ebn0_db=snr)
            q = np.clip(np.round(sb.iq.numpy() * 100.0), -127, 127).astype(np.int8)
            iq = q.astype(np.float32) / np.float32(100.0)
            txl = [(c, np.asarray(f, dtype=np.uint8)) for c in range(nch) for (_, f) in sb.frames[c]]
            L = max(len(f) for _, f in txl)
            tx = np.zeros((len(txl), L), dtype=np.uint8)
            for i, (_, f) in enumerate(txl):
                tx[i, : len(f)] = f
            fr = oracle_lib.batch_run(t, iq, nthreads=4, cap_per_channel=1000)
            hit, alien = tx_common.score(t, fr, tx, np.array([c for c, _ in txl]), np.array([len(f) for _, f in txl]))
            out[f"iq{k}"] = q
            out[f"tx{k}"] = tx
            out[f"txch{k}"] = np.array([c for c, _ in txl], dtype=np.int32)
            out[f"txlen{k}"] = np.array([len(f) for _, f in txl], dtype=np.int32)
            out[f"floor{k}"] = np.array([int(0.9 * hit)], dtype=np.int32)
            print(f"type {t} snr {snr}: sent {len(txl)} clean-and-transmitted {hit} alien {alien} floor {int(0.9 * hit)}")
        path = os.path.join(HERE, f"tx_{tx_common.NAMES[t]}.npz")
        np.savez_compressed(path, **out)
        print("  ", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
