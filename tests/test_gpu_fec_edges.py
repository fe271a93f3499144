"""-m gpu: the RS(255,231) corrector and the sync detector at their limits, with byte errors injected into the
on-air bit stream of a clean (40 dB) signal so that every codeword's error count is known exactly:
0..12 errors -> corrected, nerr = count, payload = transmitted; 13+ -> rejected (nerr = -1), frame untouched;
errors in the parity bytes, in the shortened part's neighbours, in both codewords at once; header bit errors up
to and beyond the sync threshold.  Every record is also compared byte for byte with the oracle's."""
import numpy as np
import pytest
import torch

from sdrpp_radiosonde_amd import synth
from sdrpp_radiosonde_amd.batch import SondeBatch

pytestmark = pytest.mark.gpu
TILE = 2048
COUNTS = [0, 1, 2, 6, 11, 12, 13, 14, 24]


def _cw_bytes(flen, c):
    """frame byte offsets of codeword c: 24 parity bytes, then the interleaved message bytes"""
    return np.concatenate([np.arange(8 + 24 * c, 32 + 24 * c), np.arange(56 + c, flen, 2)])


def _inject(bits, pos, byte_off, val):
    """xor `val` into frame byte `byte_off` of the frame starting at bit `pos` (bits are LSB first)"""
    for b in range(8):
        if (val >> b) & 1:
            bits[pos + 8 * byte_off + b] ^= 1


@pytest.mark.parametrize("extended", [False, True])
def test_rs_corrector_limits(oracle, extended):
    C, n = 9, TILE * (150 if extended else 100)
    flen = 518 if extended else 320
    nbits = int(n * 4800 / 48000) + 16
    bits, frames = synth.rs41_bitstreams(77 + int(extended), np.arange(C), nbits, extended)
    bits = bits.copy()
    rng = np.random.default_rng(99)
    plan = {}                                  # (channel, tx bit position) -> (e0, e1)
    for c in range(C):
        for k, (pos, _) in enumerate(frames[c]):
            e0 = COUNTS[(c + k) % len(COUNTS)]
            e1 = COUNTS[(c + 2 * k + 3) % len(COUNTS)]
            for cw, e in ((0, e0), (1, e1)):
                offs = _cw_bytes(flen, cw)
                offs = offs[offs != 56]            # byte 56 selects the frame length: keep it clean here
                # the first few go into the parity bytes, the rest anywhere in the codeword
                pick = set(rng.choice(offs[:24], size=min(e, 2), replace=False).tolist()) if e else set()
                while len(pick) < e:
                    pick.add(int(rng.choice(offs)))
                for o in pick:
                    _inject(bits[c], pos, int(o), int(rng.integers(1, 256)))
            plan[(c, pos)] = (e0, e1)
    iq, *_ = synth.gfsk_modulate(bits, n, 4800.0, seed=5, ebn0_db=40.0)
    b = SondeBatch(C, n)
    b.submit(iq.to("cuda:0"))
    got = b.frames()
    ref = oracle.batch_run(0, iq.numpy(), nthreads=4)
    assert got.tobytes() == ref.tobytes()
    checked = {e: 0 for e in COUNTS}
    for f in got:
        c = int(f["channel"])
        cand = [(abs(int(f["bitpos"]) - pos), pos, tx) for pos, tx in frames[c]]
        d, pos, tx = min(cand, key=lambda t: t[0])
        assert d < 64
        e = plan[(c, pos)]
        for cw in (0, 1):
            offs = _cw_bytes(flen, cw)
            if e[cw] <= 12:
                assert f["nerr"][cw] == e[cw], (c, pos, cw, e, f["nerr"])
                assert np.array_equal(f["data"][offs], tx[offs])
            else:
                assert f["nerr"][cw] == -1, (c, pos, cw, e, f["nerr"])
                assert not np.array_equal(f["data"][offs], tx[offs])      # left as received
            checked[e[cw]] += 1
    assert all(v >= 2 for v in checked.values()), checked


def test_sync_header_bit_errors(oracle):
    """0..6 wrong header bits: frame found (threshold 6 of 64); 7 or more: not found.  Inverted polarity alike."""
    C, n = 8, TILE * 60
    nbits = int(n * 4800 / 48000) + 16
    bits, frames = synth.rs41_bitstreams(123, np.arange(C), nbits)
    bits = bits.copy()
    rng = np.random.default_rng(3)
    nflip = {}
    for c in range(C):
        for k, (pos, _) in enumerate(frames[c]):
            m = (3 * c + 5 * k) % 10                       # 0..9 flipped header bits
            for q in rng.choice(64, size=m, replace=False):
                bits[c, pos + int(q)] ^= 1
            nflip[(c, pos)] = m
    for invert in (False, True):
        iq, *_ = synth.gfsk_modulate(bits, n, 4800.0, seed=8, ebn0_db=40.0, invert=invert)
        b = SondeBatch(C, n)
        b.submit(iq.to("cuda:0"))
        got = b.frames()
        ref = oracle.batch_run(0, iq.numpy(), nthreads=4)
        assert got.tobytes() == ref.tobytes()
        found = set()
        for f in got:
            c = int(f["channel"])
            d, pos = min((abs(int(f["bitpos"]) - p), p) for p, _ in frames[c])
            if d < 64:
                found.add((c, pos))
                assert (f["flags"] & 1) == int(invert)
        sent = [(c, pos) for c in range(C) for pos, _ in frames[c][1:]]          # the first frame may fall into acquisition
        assert all(((c, pos) in found) == (nflip[(c, pos)] <= 6) for c, pos in sent)
        assert any(nflip[k] > 6 for k in sent) and any(nflip[k] == 6 for k in sent)
