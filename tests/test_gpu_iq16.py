"""SONDE_INPUT_IQ16 (16-bit integer IQ rows: what SDR hardware and WAV recordings hold; the step in front of `vfo->output`,
/root/reference/src/main.cpp:55-57): the kernel converts int16 -> float exactly in its load path and then runs SONDE_INPUT_IQ's
arithmetic, so bits, loop state and frames must equal (a) the float path fed with the same integers as floats and (b) the oracle."""
import numpy as np
import pytest
import torch

import oracle_lib
from sdrpp_radiosonde_amd import _lib, synth
from sdrpp_radiosonde_amd.batch import SondeBatch, strided_rows

pytestmark = pytest.mark.gpu
TILE = 2048


def quantise(iq: torch.Tensor, full_scale: float = 8192.0, bits: int = 16) -> torch.Tensor:
    """bits 16: int16 rows (SONDE_INPUT_IQ16); bits 8: int8 rows (SONDE_INPUT_IQ8), the unit-amplitude signal at 16 counts"""
    if bits == 8:
        return torch.clamp(torch.round(iq * 16.0), -128, 127).to(torch.int8)
    return torch.clamp(torch.round(iq * full_scale), -32768, 32767).to(torch.int16)


KIND = {16: _lib.INPUT_IQ16, 8: _lib.INPUT_IQ8}


def run(kind, x, types, chunks, flags=0, strided=False):
    C, n = x.shape[0], x.shape[1]
    b = SondeBatch(C, max(chunks) * TILE, types=types, input_kind=kind, flags=flags)
    frames, pos = [], 0
    for k in chunks:
        blk = x[:, pos:pos + k * TILE].contiguous()
        if strided:
            blk = strided_rows(blk)
        b.submit(blk)
        frames.append(b.frames().copy())
        pos += k * TILE
    bits = [(b.nbits(c), b.read_bits(c, max(0, b.nbits(c) - 4000), min(4000, b.nbits(c)))) for c in range(C)]      # (the newest 4000: the ring is finite)
    state = [b.state(c) for c in range(C)]
    b.close()
    return np.concatenate(frames), bits, state


@pytest.mark.parametrize("bits", [16, 8])
@pytest.mark.parametrize("stype,flags,chunks", [(0, 0, (8, 16)), (1, 0, (24,)), (3, 0, (5, 7, 12)), (2, 0, (24,)), (6, 0, (24,)), (0, 1, (24,)), (3, 1, (12, 12))])
def test_iq16_equals_float_path_and_oracle(stype, flags, chunks, bits):
    C, n = 6, sum(chunks) * TILE
    sb = synth.make_batch(stype, C, n, seed=300 + stype, ebn0_db=13.0, device="cuda", cfo_max_hz=1500.0)
    x16 = quantise(sb.iq, bits=bits)
    xf = x16.to(torch.float32)
    types = np.full(C, stype, dtype=np.uint8)
    # (flags 1 = SONDE_FLAG_WIDE: the (2:1, 16 taps) and (none, 16 taps) classes: 8-byte (int8: 4-byte) loads of two samples, converted)
    f16, b16, s16 = run(KIND[bits], x16, types, chunks, flags, strided=True)
    ff, bf, sf = run(_lib.INPUT_IQ, xf, types, chunks, flags)
    assert len(f16) > 0 and np.array_equal(f16, ff)
    for c in range(C):
        assert b16[c][0] == bf[c][0] and np.array_equal(b16[c][1], bf[c][1]) and s16[c] == sf[c]
    if flags == 0:
        xh = xf.cpu().numpy()
        for c in range(2):
            ch = oracle_lib.Channel(stype, c)
            ch.feed(xh[c], is_iq=True)
            ref = ch.frames()
            got = f16[f16["channel"] == c]
            got = got[np.argsort(got["bitpos"], kind="stable")]
            ref = ref[np.argsort(ref["bitpos"], kind="stable")]
            assert len(ref) == len(got) and got.tobytes() == ref.tobytes()
            rb = ch.bits()
            assert len(rb) == b16[c][0] and np.array_equal(rb[-len(b16[c][1]):] if len(b16[c][1]) else rb[:0], b16[c][1]) and ch.state() == s16[c]


@pytest.mark.parametrize("bits", [16, 8])
def test_iq16_mixed_types_pipelined(bits):
    """Mixed batch (launch units per type, channel lists) on 16-bit / 8-bit rows."""
    C, n = 48, 24 * TILE
    types = np.array([(0, 3, 1)[c % 3] for c in range(C)], dtype=np.uint8)
    x = torch.empty((C, n, 2), dtype=torch.float32, device="cuda")
    for t in (0, 3, 1):
        idx = np.nonzero(types == t)[0]
        x[torch.from_numpy(idx).cuda()] = synth.make_batch(t, len(idx), n, seed=40 + t, ebn0_db=14.0, device="cuda").iq
    x16 = quantise(x, bits=bits)
    for flags in (0, 4):
        f16, b16, s16 = run(KIND[bits], x16, types, (24,), flags)
        ff, bf, sf = run(_lib.INPUT_IQ, x16.to(torch.float32), types, (24,), flags)
        assert len(f16) >= C and np.array_equal(f16, ff)
        assert all(b16[c][0] == bf[c][0] and np.array_equal(b16[c][1], bf[c][1]) and s16[c] == sf[c] for c in range(C))


def test_iq16_checks_dtype_and_survives_silence():
    b = SondeBatch(2, 8 * TILE, input_kind=_lib.INPUT_IQ16)
    with pytest.raises(Exception):
        b.submit(torch.zeros((2, 8 * TILE, 2), dtype=torch.float32, device="cuda"))
    b.submit(torch.zeros((2, 8 * TILE, 2), dtype=torch.int16, device="cuda"))
    assert b.sync() == 0                               # all-zero input: atan2q(0, 0) = 0, nothing decodes, nothing breaks
    b.close()


@pytest.mark.parametrize("bits", [16, 8])
@pytest.mark.parametrize("snr", [10.0, 20.0])
def test_iq16_tone_demodulated_sondes(snr, bits):
    """iMet-4 and SRS-C50 (AFSK on FM: tone demodulator kernel in front of kernel A) from 16-bit rows, beside RS41 channels in one batch:
    frames, bits and loop state of the float path on the same integers; two submits of 16384-sample granules."""
    per, n = 4, 16384 * 6
    a = synth.make_imet_batch(per, 2 * n, seed=71, snr_db=snr)
    c5 = synth.make_c50_batch(per, 2 * n, seed=72, snr_db=snr)
    r = synth.make_rs41_batch(per, 2 * n, seed=73, ebn0_db=14.0)
    x16 = quantise(torch.cat([a.iq, c5.iq, r.iq]).cuda(), bits=bits)
    types = np.array([4] * per + [5] * per + [0] * per, dtype=np.uint8)
    f16, b16, s16 = run(KIND[bits], x16, types, (48, 48))
    ff, bf, sf = run(_lib.INPUT_IQ, x16.to(torch.float32), types, (48, 48))
    assert len(f16) >= 2 * per and np.array_equal(f16, ff)
    assert {int(t) for t in types[np.unique(f16["channel"])]} == {0, 4, 5}          # every sonde type decoded something
    for c in range(3 * per):
        assert b16[c][0] == bf[c][0] and np.array_equal(b16[c][1], bf[c][1]) and s16[c] == sf[c]


@pytest.mark.parametrize("bits", [16, 8])
def test_iq16_from_host_memory(bits):
    C, n = 8, 24 * TILE
    sb = synth.make_rs41_batch(C, n, seed=9, ebn0_db=14.0, device="cuda")
    x16 = quantise(sb.iq, bits=bits)
    f_dev, _, _ = run(KIND[bits], x16, None, (24,))
    b = SondeBatch(C, n, input_kind=KIND[bits])
    b.submit_host(x16.cpu().numpy())
    f_host = b.frames().copy()
    b.close()
    assert len(f_dev) >= 1 and np.array_equal(f_dev, f_host)
