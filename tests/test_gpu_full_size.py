"""-m gpu: BASELINE.json's configurations at their FULL per-GPU sizes (configs[1], [2] and the per-GPU shard of
configs[4]).  The IQ is synthesised on the GPU, the oracle runs on all host cores over the host copy of the
same buffer, and the comparison is (a) every frame record byte for byte, (b) the per-channel bit counts,
(c) size-independent properties: frames ordered by (channel, bit position), every frame whose FEC succeeded
equals a transmitted frame, and almost every transmitted frame is received."""
import os

import numpy as np
import pytest
import torch

from sdrpp_radiosonde_amd import synth
from sdrpp_radiosonde_amd.batch import SondeBatch

pytestmark = pytest.mark.gpu
TILE = 2048
CORES = os.cpu_count() or 4


def _check_order(fr):
    key = fr["channel"].astype(np.int64) << 40 | fr["bitpos"].astype(np.int64)
    assert (np.diff(key) > 0).all(), "frames must come out ordered by (channel, bit position)"


def _tx_match_fraction(sb, got, skip=8):
    """fraction of the frames with clean FEC that equal a transmitted frame (bytes skip.. are FEC-covered)"""
    good = got[(got["nerr"] >= 0).all(axis=1)]
    hit = 0
    for f in good:
        hit += any(np.array_equal(tx[skip:], f["data"][skip: f["len"]]) for _, tx in sb.frames[f["channel"]])
    return hit, len(good)


def test_config2_1024_rs41_channels_full_size(oracle):
    C, n = 1024, 96 * TILE           # the bench workload: 1.61 GB of IQ
    sb = synth.make_rs41_batch(C, n, seed=2024, ebn0_db=14.0, device="cuda:0")
    b = SondeBatch(C, n)
    b.submit(sb.iq)
    got = b.frames()
    host = sb.iq.cpu().numpy()
    ref = oracle.batch_run(0, host, nthreads=CORES)
    assert len(ref) >= 5 * C
    assert got.tobytes() == ref.tobytes()
    _check_order(got)
    nb = np.array([b.nbits(c) for c in range(C)])
    assert abs(nb - n // 10).max() <= 16                 # 4800 Bd at 48 kS/s, minus the FIR look-ahead
    for c in range(0, C, 97):                            # bit streams of a sample of channels
        ch = oracle.Channel(0, c)
        ch.feed(host[c])
        assert nb[c] == len(ch.bits()) and np.array_equal(b.read_bits(c, 0, int(nb[c])), ch.bits())
    hit, good = _tx_match_fraction(sb, got)
    assert hit == good and good >= 0.97 * len(got)
    sent = sum(len(f) for f in sb.frames)
    assert len(got) >= sent - C                           # at most the acquisition frame of a channel is lost
    # idempotence of the device state reset: a fresh batch over the same buffer gives the same records
    b2 = SondeBatch(C, n)
    b2.submit(sb.iq)
    assert b2.frames().tobytes() == got.tobytes()


@pytest.mark.parametrize("flags", [0, 32])       # the default (every submit joined into the caller's stream) and SONDE_FLAG_LATE_JOIN (launch units joined one submit late)
def test_config3_4096_mixed_channels_full_size(oracle, flags):
    C, n = 4096, 32 * TILE
    order = (0, 3, 1)                                     # RS41, M10, DFM09 by channel % 3
    types = np.array([order[c % 3] for c in range(C)], dtype=np.uint8)
    iq = torch.empty((C, n, 2), dtype=torch.float32, device="cuda:0")
    refs, sbs = [], {}
    for t in order:
        idx = np.nonzero(types == t)[0]
        sb = synth.make_batch(int(t), len(idx), n, seed=300 + int(t), ebn0_db=16.0, device="cuda:0")
        iq[torch.from_numpy(idx).to("cuda:0")] = sb.iq
        r = oracle.batch_run(int(t), sb.iq.cpu().numpy(), nthreads=CORES)
        r["channel"] = idx[r["channel"]]
        refs.append(r)
        sbs[t] = (idx, sb)
        del sb
    ref = np.concatenate(refs)
    ref = ref[np.lexsort((ref["bitpos"], ref["channel"]))]
    b = SondeBatch(C, n, types=types, flags=flags)
    b.submit(iq)
    got = b.frames()
    assert len(ref) >= C
    assert got.tobytes() == ref.tobytes()
    _check_order(got)
    for t in order:
        assert (got["type"] == t).sum() >= (types == t).sum()
    assert (got["type"] == types[got["channel"]]).all()


def test_config5_shard_8192_rs41_channels(oracle):
    """65 536 channels / 8 GPUs = 8192 channels per rank, one second (24 tiles) per submit, two submits."""
    C, n = 8192, 48 * TILE
    sb = synth.make_rs41_batch(C, n, seed=555, ebn0_db=15.0, device="cuda:0")
    b = SondeBatch(C, n // 2)
    parts = []
    for lo in (0, n // 2):
        b.submit(sb.iq[:, lo: lo + n // 2].contiguous())
        parts.append(b.frames())
    got = np.concatenate(parts)
    got = got[np.lexsort((got["bitpos"], got["channel"]))]
    ref = oracle.batch_run(0, sb.iq.cpu().numpy(), nthreads=CORES)
    assert len(ref) >= 2 * C
    assert got.tobytes() == ref.tobytes()
    hit, good = _tx_match_fraction(sb, got)
    assert hit == good and good >= 0.97 * len(got)
