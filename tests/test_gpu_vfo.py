"""VFO front-end on the GPU (vfo.hip) against the oracle (or_chan.c or_vfo_*): the 48 kS/s rows bit for bit at every
ratio of /root/reference/src/main.hpp:44-52, across ragged submits and strided rows; then the whole reference chain
VFO-rate IQ -> FM -> resampler -> decoder (main.cpp:55-68) on the GPU, frames identical to the oracle chain's."""
import numpy as np
import pytest
import torch

import oracle_lib
from sdrpp_radiosonde_amd import synth
from sdrpp_radiosonde_amd._lib import INPUT_REAL
from sdrpp_radiosonde_amd.batch import SondeBatch, SondeVfo, VFO_RATE

pytestmark = pytest.mark.gpu
RATES = {10000: (24, 5), 15000: (16, 5), 20000: (12, 5), 40000: (6, 5), 50000: (24, 25)}


@pytest.mark.parametrize("rate", sorted(RATES))
def test_rows_bit_exact_ragged_submits(rate):
    up, down = RATES[rate]
    C = 3
    pieces = [down, 1600, 1600 + down, 3 * 1600, 7 * down, 4800 - 2 * down, 12800]      # below / at / across the kernel's chunk size
    total = sum(pieces)
    rng = np.random.default_rng(rate)
    ph = np.cumsum(rng.uniform(-1.2, 1.2, size=(C, total)), axis=1)
    amp = rng.uniform(0.2, 1.0, size=(C, 1))
    iq = np.stack([amp * np.cos(ph), amp * np.sin(ph)], axis=2).astype(np.float32)
    iq += 0.01 * rng.standard_normal(iq.shape).astype(np.float32)
    iq[1, 100:140] = 0.0                                                                # a dropout: atan2q(0, 0) = 0
    want = np.stack([oracle_lib.Vfo(rate).process(iq[c]) for c in range(C)])
    v = SondeVfo(C, rate, max(pieces))
    assert (v.up, v.down) == (up, down)
    dev = torch.from_numpy(iq).cuda()
    got, a = [], 0
    for n in pieces:
        rows = dev[:, a:a + n]                                                           # strided rows of the whole recording
        out = v.process(rows)
        assert out.shape == (C, n * up // down)
        got.append(out.cpu().numpy())
        a += n
    got = np.concatenate(got, axis=1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    with pytest.raises(Exception):
        v.process(dev[:, :down + 1])
    v.close()


@pytest.mark.parametrize("stype", [0, 1, 3, 4])
def test_reference_chain_vfo_rate_to_frames(stype):
    """What the reference runs per VFO: IQ at supportedTypes[i].bandwidth -> FM -> resampler -> decoder.  GPU: SondeVfo ->
    SondeBatch(INPUT_REAL).  Frames bit-identical to the oracle chain's."""
    rate = VFO_RATE[stype]
    up, down = RATES[rate]
    C, n_sub = 4, 3
    n_out = 49152                       # per submit: a multiple of 6144 (whole tiles at every ratio) and of 16384 (iMet)
    n_in = n_out * down // up
    if stype == 4:
        sb = synth.make_imet_batch(C, n_in * n_sub, seed=31, snr_db=25.0, fs=float(rate))
    else:
        sb = synth.make_batch(stype, C, n_in * n_sub, seed=31 + stype, ebn0_db=20.0, fs=float(rate), cfo_max_hz=300.0)
    x = sb.iq.numpy()
    want = []
    for c in range(C):
        vo, ch = oracle_lib.Vfo(rate), oracle_lib.Channel(stype, c)
        for k in range(n_sub):
            ch.feed(vo.process(x[c, k * n_in:(k + 1) * n_in]), is_iq=False)
        want.append(ch.frames())
    want = np.concatenate(want)
    assert len(want) >= C * 2
    v = SondeVfo(C, rate, n_in)
    b = SondeBatch(C, n_out, types=np.full(C, stype, dtype=np.uint8), input_kind=INPUT_REAL)
    dev = sb.iq.cuda()
    got = []
    for k in range(n_sub):
        audio = v.process(dev[:, k * n_in:(k + 1) * n_in])
        b.submit(audio)
        b.sync()
        got.append(b.frames())
    got = np.concatenate(got)
    got = got[np.lexsort((got["bitpos"], got["channel"]))]
    want = want[np.lexsort((want["bitpos"], want["channel"]))]
    assert len(got) == len(want)
    for f in ("channel", "len", "bitpos", "nerr"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["data"], want["data"])
    b.close()
    v.close()
