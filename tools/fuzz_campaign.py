#!/usr/bin/env python3
"""Differential campaign (GPU box): the low-SNR mixed-batch parity test of tests/test_gpu_fuzz_parity.py over many seeds and
SNRs, plus the RS corrector unit test with fresh random patterns.  usage: python tools/fuzz_campaign.py [n_seeds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz_parity as fz  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(2024)
t0 = time.time()
for i in range(n):
    ebn0 = float(rng.uniform(4.0, 16.0))
    seed = int(rng.integers(10, 10_000))
    flags = int(rng.choice([0, 0, 4, 32]))                              # the default (joined at every submit), SONDE_FLAG_PIPELINE, SONDE_FLAG_LATE_JOIN
    cfo = float(rng.choice([500.0, 1500.0, 2500.0]))                    # carrier offsets: the AFC of SPEC 3.0b
    iq16 = (False, False, True, 8)[int(rng.integers(0, 4))]              # a quarter of the batches as 16-bit integer IQ rows (SONDE_INPUT_IQ16), a quarter as 8-bit (IQ8)
    slices = int(rng.choice([0, 0, 2, 3, 4]))                           # SondeBatchConfig.time_slices (round 6): the library's choice, or forced
    total = fz.run_mixed(ebn0, seed, check_coverage=False, flags=flags, cfo_max_hz=cfo, iq16=iq16, time_slices=slices)      # raises on the first differing bit, state or frame
    print(f"[{i + 1}/{n}] Eb/N0 {ebn0:5.2f} dB seed {seed} flags {flags} slices {slices} cfo +-{cfo:.0f} Hz{' int8' if iq16 == 8 else (' int16' if iq16 else '')}: {total} frames, identical to the oracle", flush=True)
for snr in (6.0, 10.0, 20.0):
    fz.test_afsk_low_snr_bit_exact(snr)
    print(f"afsk {snr} dB: identical", flush=True)
print(f"campaign done in {time.time() - t0:.0f} s")
