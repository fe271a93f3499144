"""The SPEC demodulator against a CONVENTIONAL receiver (oracle/or_yardstick.c: VFO channel filter, libm atan2f, AGC,
per-symbol Gardner PI loop -- what SURVEY.md Appendix B.1 records for sondedump's gfsk_demod behind
/root/reference/src/decode/decoder.hpp:22,61 and /root/reference/src/main.cpp:55-60 put in front of it), which shares no
demodulator arithmetic with the SPEC.  It is the independent evidence that a SPEC shaped for 256-lane workgroups decodes the
frames a normal CPU decoder decodes.  The yardstick is test infrastructure like the rest of oracle/.

CPU: the oracle (the HIP path's bit-exact twin) against the yardstick.  GPU: the HIP path itself.
Full table: tools/yardstick_study.py -> profiles/r4_yardstick.md."""
import os
import sys

import numpy as np
import pytest
import torch

from sdrpp_radiosonde_amd import synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import yardstick_study as ys  # noqa: E402

CORES = len(os.sched_getaffinity(0))
N = 2048 * 96


def test_yardstick_is_an_independent_decoder(oracle):
    """At high SNR the yardstick decodes every frame the generator sent, for all five GFSK sondes: two demodulators that share
    no arithmetic agree with the generator (and with each other) on every frame."""
    for t in (0, 1, 2, 3, 6):
        iq, sb = ys.scene(t, 6, N, 500 + t, 25.0)
        y = ys.keys(t, oracle.yard_run(t, iq.numpy(), nthreads=CORES))
        a = ys.keys(t, oracle.batch_run(t, iq.numpy(), nthreads=CORES))
        sent = sum(len(f) for f in sb.frames)
        assert len(y) >= sent - 2 * 6 and len(a) >= sent - 2 * 6, (t, sent, len(y), len(a))      # (the frame cut by the end of the block, the acquisition frame)
        assert len(a & y) >= 0.99 * len(y), t


# (sonde, Eb/N0, carrier offset Hz, clock ppm, least share of the yardstick's FEC-clean frames the SPEC must also deliver, channels, seed)
CELLS = [(t, snr, cfo, ppm, 0.95 if abs(cfo) >= 2000.0 else 0.98, 16, 4100 + t)
         for t, snrs in ((0, (12.0, 14.0)), (1, (12.0, 14.0)), (3, (14.0, 16.0)))
         for snr in snrs
         for cfo, ppm in ((0.0, 0.0), (1000.0, 0.0), (-2000.0, 0.0), (0.0, 100.0), (0.0, -100.0))]
# Round 5: the cells where the SPEC used to lose -- RS41 at its waterfall (Eb/N0 10 dB: 8 dB decodes nothing,
# 12 dB everything, for either receiver) with the carrier on and 1 kHz off the channel centre: 94 % of the yardstick's frames before
# the carrier-following boxcar of SPEC 3.0d, 96.8-98.4 % with it (32 channels: a frame is half a percent); DFM and MRZ-N1 2 kHz off
# at 12-16 dB, iMS-100 at 14 dB (profiles/r5_yardstick.md)
CELLS += [(0, 10.0, 0.0, 0.0, 0.965, 32, 4000), (0, 10.0, 1000.0, 0.0, 0.955, 32, 4000), (0, 10.0, -1000.0, 0.0, 0.955, 32, 4000),
          (0, 12.0, 2000.0, 0.0, 0.99, 16, 4100), (1, 12.0, 2000.0, 0.0, 0.975, 16, 4101), (1, 12.0, -2000.0, 0.0, 0.975, 16, 4101),      # (SPEC 3.2b: 97.5 / 96.8 % -> 98.9 / 98.2 %)
          (6, 14.0, -2000.0, 0.0, 0.99, 16, 4106), (6, 14.0, 2000.0, 0.0, 0.99, 16, 4106),      # (SPEC 3.0e: the threshold follows the rotation; before: 92 % of 32 channels' frames at -2 kHz)
          (6, 16.0, 2000.0, 0.0, 0.99, 16, 4106), (6, 16.0, -2000.0, 0.0, 0.99, 16, 4106), (0, 12.0, -2000.0, 0.0, 0.99, 16, 4100),
          (2, 14.0, 2000.0, 0.0, 0.985, 16, 4102), (2, 14.0, -2000.0, 0.0, 0.985, 16, 4102)]


def _shares(oracle, decode):
    bad = []
    for t, snr, cfo, ppm, least, C, seed in CELLS:
        iq, sb = ys.scene(t, C, N, seed, snr, cfo, ppm)
        host = iq.numpy()
        y = ys.keys(t, oracle.yard_run(t, host, nthreads=CORES))
        a = ys.keys(t, decode(t, iq))
        if len(y) == 0 or len(a & y) < least * len(y) or len(a) < least * len(y):
            bad.append((t, snr, cfo, ppm, len(a), len(y), len(a & y)))
    return bad


def test_spec_decodes_what_a_conventional_receiver_decodes(oracle):
    """RS41, DFM, M10 at and above each sonde's working SNR, with carrier offsets of +1 / -2 kHz (the AFC of SPEC 3.0b) and
    symbol clocks off by +-100 ppm: the SPEC's FEC-clean frames include >= 98 % (95 % at 2 kHz) of the yardstick's, and are at
    least as large a share in number.  Below those SNRs the two differ by design (profiles/r4_yardstick.md: the SPEC is 1-3 dB better for
    DFM / iMS-100 / M10 / MRZ-N1, whose VFO channels are 15-50 kHz wide in the reference, and level with it for RS41)."""
    bad = _shares(oracle, lambda t, iq: oracle.batch_run(t, iq.numpy(), nthreads=CORES))
    assert not bad, bad


def test_afc_pulls_in_a_carrier_offset(oracle):
    """SPEC 3.0b: with the carrier 2 kHz off an RS41 channel at 12 kS/s sits at 4.4 of 6 kHz on one side; the AFC state converges
    to u = tan(pi f / fs_int) (minus the leak's few percent) within a second and the frames decode; without a signal the state
    stays near 0."""
    iq, sb = ys.scene(0, 4, N, 77, 14.0, 2000.0)
    for c in range(4):
        ch = oracle.Channel(0, c)
        ch.feed(iq.numpy()[c, :2048 * 24])
        u1 = ch.state()["yprev"]
        ch.feed(iq.numpy()[c, 2048 * 24:])
        u = ch.state()["yprev"]
        want = np.tan(np.pi * 2000.0 / 12000.0)
        assert 0.80 * want < u1 < 1.02 * want and 0.90 * want < u < 1.02 * want, (c, u1, u, want)
        assert len(ch.frames()) >= len(sb.frames[c]) - 2
    rng = np.random.default_rng(5)
    noise = rng.standard_normal((N, 2)).astype(np.float32)
    ch = oracle.Channel(0, 0)
    ch.feed(noise)
    assert abs(ch.state()["yprev"]) < 0.15


@pytest.mark.gpu
def test_hip_path_decodes_what_a_conventional_receiver_decodes(oracle):
    """The same cells through libsonde_mi355.so."""
    from sdrpp_radiosonde_amd.batch import SondeBatch

    def decode(t, iq):
        b = SondeBatch(iq.shape[0], iq.shape[1], types=np.full(iq.shape[0], t, dtype=np.uint8))
        b.submit(iq.to("cuda:0"))
        fr = b.frames()
        b.close()
        return fr
    bad = _shares(oracle, decode)
    assert not bad, bad
