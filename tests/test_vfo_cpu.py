"""VFO front-end (SURVEY 8 rows a1 + a2: discriminator + rational resampler at the reference's VFO rates,
/root/reference/src/main.cpp:55-60, main.hpp:44-52), CPU side: the oracle's properties, the product's tap tables, and the
oracle chain VFO-rate IQ -> 48 kS/s FM audio -> decoder that the GPU tests compare against."""
import numpy as np
import pytest

import oracle_lib
from sdrpp_radiosonde_amd import _lib, synth

RATES = {10000: (24, 5), 15000: (16, 5), 20000: (12, 5), 40000: (6, 5), 50000: (24, 25)}


@pytest.mark.parametrize("rate", sorted(RATES))
def test_ratio_and_taps(rate):
    v = oracle_lib.Vfo(rate)
    assert (v.up, v.down) == RATES[rate] and v.up * rate == v.down * 48000
    g = v.taps()
    np.testing.assert_allclose(g.sum(axis=1), 1.0, atol=2e-6)                 # every phase: unit DC gain
    # the product builds the same table (host code, no GPU needed)
    L = _lib.load()
    gp = np.zeros_like(g)
    assert L.sonde_vfo_taps(rate, gp.ctypes.data) == 0
    assert np.array_equal(g.view(np.uint32), gp.view(np.uint32))
    up, down = _lib.C.c_int(), _lib.C.c_int()
    assert L.sonde_vfo_ratio(rate, _lib.C.byref(up), _lib.C.byref(down)) == 0 and (up.value, down.value) == RATES[rate]
    assert L.sonde_vfo_ratio(12345, None, None) != 0
    # prototype response: flat in the passband, >= 50 dB down beyond the images' edge
    proto = g.T.reshape(-1).astype(np.float64)                               # proto[up t + p] = g[p][t] (x up)
    H = np.abs(np.fft.rfft(proto, 1 << 16)) / v.up
    f = np.fft.rfftfreq(1 << 16, 1.0 / (rate * v.up))
    lo = min(rate, 48000)
    assert np.all(np.abs(20 * np.log10(H[f < 0.25 * lo])) < 0.6)
    assert np.all(20 * np.log10(H[f > 0.75 * lo] + 1e-12) < -50.0)


def test_twelve_fifths_is_the_channelizer_stage():
    """(round 4: the channelizer's bins run at 20 kS/s; its resampler is the VFO front-end's 20 kS/s stage, 12/5)"""
    L = oracle_lib.lib()
    g = np.zeros((12, 16), dtype=np.float32)
    L.or_chan_resamp_taps(oracle_lib.fptr(g))
    assert np.array_equal(g, oracle_lib.Vfo(20000).taps())


@pytest.mark.parametrize("rate", sorted(RATES))
def test_tone_and_streaming(rate):
    up, down = RATES[rate]
    n = 40 * down * 8
    f0 = 0.11 * rate
    t = np.arange(n)
    iq = np.stack([np.cos(2 * np.pi * f0 * t / rate), np.sin(2 * np.pi * f0 * t / rate)], axis=1).astype(np.float32)
    whole = oracle_lib.Vfo(rate).process(iq)
    assert whole.shape[0] == n * up // down
    # a constant frequency is a constant discriminator output: 4 f0 / rate quadrants per sample, passed with unit gain
    np.testing.assert_allclose(whole[64:], 4.0 * f0 / rate, atol=2e-3)
    # streaming in ragged pieces (multiples of `down`) gives the same samples bit for bit
    v = oracle_lib.Vfo(rate)
    cuts = [0, down, 7 * down, 8 * down, 200 * down, n]
    parts = [v.process(iq[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(parts).view(np.uint32), whole.view(np.uint32))


@pytest.mark.parametrize("stype,rate", [(0, 10000), (1, 15000), (2, 20000), (3, 50000), (6, 20000)])
def test_oracle_chain_decodes(stype, rate):
    """VFO-rate IQ -> or_vfo -> or_channel (real input): the frames the generator sent come out."""
    up, down = RATES[rate]
    n_out = 6144 * 24
    n_in = n_out * down // up
    sb = synth.make_batch(stype, 2, n_in, seed=77 + stype, ebn0_db=22.0, fs=float(rate), cfo_max_hz=300.0)
    x = sb.iq.numpy()
    for c in range(2):
        audio = oracle_lib.Vfo(rate).process(x[c])
        ch = oracle_lib.Channel(stype, c)
        ch.feed(audio, is_iq=False)
        fr = ch.frames()
        want = [f for f in sb.frames[c] if f[0] + 8 * len(f[1]) * (2 if stype in (1, 2, 3, 6) else 1) < n_in * synth.SONDE_BAUD[stype] / rate - 64]
        assert len(fr) >= max(1, len(want) - 1), (len(fr), len(want))
        assert (fr["nerr"] >= 0).all()
