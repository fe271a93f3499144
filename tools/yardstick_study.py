#!/usr/bin/env python3
"""Yardstick study: the SPEC demodulator (oracle/or_dsp.c = the HIP path, bit for bit) against the conventional per-sample
receiver (oracle/or_yardstick.c) on the same synthetic IQ: FEC-clean frames of each, the overlap, per sonde and Eb/N0, with
carrier offset and symbol-clock offset.  usage: python tools/yardstick_study.py [--gpu] > profiles/r4_yardstick.md
(--gpu: the HIP path instead of its CPU twin; run on the GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from sdrpp_radiosonde_amd import synth  # noqa: E402

NAMES = {0: "RS41", 1: "DFM09", 3: "M10", 2: "iMS-100", 6: "MRZ-N1"}
CORES = len(os.sched_getaffinity(0))


def clean_mask(t, fr):
    if t == 0:
        return (fr["nerr"] >= 0).all(axis=1)
    if t in (1, 2):
        return fr["nerr"][:, 1] == 0
    return fr["nerr"][:, 0] == 0


def keys(t, fr):
    """identity of a FEC-clean frame: (channel, the FEC-covered bytes)"""
    lo = 8 if t == 0 else 0
    return {(int(f["channel"]), bytes(f["data"][lo: f["len"]])) for f in fr[clean_mask(t, fr)]}


def scene(t, C, n, seed, ebn0, cfo_hz=0.0, ppm=0.0, device="cpu"):
    fs = 48000.0 * (1.0 + ppm * 1e-6)            # the sonde's clock is off: its symbols are 1/(1 + ppm) as long in receiver samples
    sb = synth.make_batch(t, C, n, seed=seed, ebn0_db=ebn0, device=device, cfo_max_hz=0.0, fs=fs)
    iq = sb.iq
    if cfo_hz:
        ph = 2.0 * np.pi * cfo_hz / 48000.0 * torch.arange(n, dtype=torch.float64, device=iq.device)
        c, s = torch.cos(ph).float(), torch.sin(ph).float()
        iq = torch.stack([iq[..., 0] * c - iq[..., 1] * s, iq[..., 0] * s + iq[..., 1] * c], dim=-1).contiguous()
    return iq, sb


def decode_spec(t, iq, gpu):
    if gpu:
        from sdrpp_radiosonde_amd.batch import SondeBatch
        b = SondeBatch(iq.shape[0], iq.shape[1], types=np.full(iq.shape[0], t, dtype=np.uint8))
        b.submit(iq.to("cuda:0"))
        fr = b.frames()
        b.close()
        return fr
    return oracle_lib.batch_run(t, iq.cpu().numpy(), nthreads=CORES)


def main():
    gpu = "--gpu" in sys.argv
    C, n = 32, 2048 * 96
    snrs = (8.0, 10.0, 12.0, 14.0, 16.0)
    print("# Yardstick: the SPEC demodulator against a conventional per-sample receiver")
    print()
    print(f"SPEC = {'the HIP path (libsonde_mi355.so)' if gpu else 'oracle/or_dsp.c (the HIP path bit for bit)'}; yardstick = oracle/or_yardstick.c "
          "(VFO channel filter, libm atan2f, AGC, 4-symbol low-pass at cutoff = 1.0 x and 0.65 x symbol rate, per-symbol Gardner PI loop) -> the same framers / FEC.")
    print(f"{C} channels x {n} samples per cell; cells: FEC-clean frames SPEC / yardstick(1.0) / yardstick(0.65) | share of the yardstick(1.0)'s frames the SPEC also has.")
    print()
    for t in (0, 1, 3, 2, 6):
        print(f"## {NAMES[t]}")
        print()
        print("| condition | " + " | ".join(f"{s:g} dB" for s in snrs) + " |")
        print("|---|" + "---|" * len(snrs))
        for label, cfo, ppm in (("CFO 0, clock 0", 0.0, 0.0), ("CFO +1 kHz", 1000.0, 0.0), ("CFO -1 kHz", -1000.0, 0.0), ("CFO +2 kHz", 2000.0, 0.0),
                                ("CFO -2 kHz", -2000.0, 0.0), ("clock +100 ppm", 0.0, 100.0), ("clock -100 ppm", 0.0, -100.0)):
            row = []
            for s in snrs:
                iq, sb = scene(t, C, n, 4000 + t, s, cfo, ppm, device="cuda:0" if gpu else "cpu")
                host = iq.cpu().numpy()
                a = keys(t, decode_spec(t, iq, gpu))
                y1 = keys(t, oracle_lib.yard_run(t, host, nthreads=CORES, cutoff_rel=1.0))
                y2 = keys(t, oracle_lib.yard_run(t, host, nthreads=CORES, cutoff_rel=0.65))
                share = 100.0 * len(a & y1) / len(y1) if y1 else float("nan")
                row.append(f"{len(a)} / {len(y1)} / {len(y2)} \\| {share:.0f} %")
            print(f"| {label} | " + " | ".join(row) + " |")
        print()


if __name__ == "__main__":
    main()
