"""-m gpu: TIME-SLICED demod launches (csrc/launch.h SdSlice, round 6).  A channel's submit is cut into segments, each its own
workgroup, segment s waiting in the kernel for segment s - 1 of the same channel; a segment is to the arithmetic what a submit is,
so frames, bits and loop state must be those of the unsliced launch = the oracle's, for every segment count, every class, mixed
batches, 16-bit rows, and submits whose tile count the segment length does not divide.
What a frame stream must equal: /root/reference/src/decode/decoder.hpp:61 (one X_decode call sequence per channel); the oracle
(oracle/, the CPU restatement) stands for it."""
import os

import numpy as np
import pytest
import torch

from sdrpp_radiosonde_amd import synth
from sdrpp_radiosonde_amd.batch import SondeBatch, strided_rows

pytestmark = pytest.mark.gpu
TILE = 2048
CORES = os.cpu_count() or 4


def _key(a):
    return a[np.lexsort((a["bitpos"], a["channel"]))]


@pytest.mark.parametrize("sonde,C,tiles,seg", [(0, 1250, 24, 2), (0, 1250, 24, 4), (0, 300, 20, 3), (0, 2100, 12, 2), (1, 1100, 24, 4), (3, 1100, 24, 3), (2, 600, 24, 4), (6, 600, 24, 2)])
def test_sliced_launch_equals_oracle(oracle, sonde, C, tiles, seg):
    """One sonde type, three consecutive submits (the third shorter: its tile count is not a multiple of the segment length)."""
    n = tiles * TILE
    short = (tiles // 2 + 1) * TILE
    total = 2 * n + short
    sb = synth.make_batch(sonde, C, total, seed=40 + sonde + seg, ebn0_db=13.0 if sonde == 0 else 15.0, device="cuda:0")
    types = None if sonde == 0 else np.full(C, sonde, dtype=np.uint8)
    b = SondeBatch(C, n, types=types, time_slices=seg)
    parts = []
    for lo, ln in ((0, n), (n, n), (2 * n, short)):
        b.submit(strided_rows(sb.iq[:, lo: lo + ln].contiguous()))
        parts.append(b.frames().copy())
    got = _key(np.concatenate(parts))
    host = sb.iq.cpu().numpy()
    ref = _key(oracle.batch_run(sonde, host, nthreads=CORES))
    assert len(ref) >= C and got.tobytes() == ref.tobytes()
    for c in (0, 1, C // 2, C - 1):
        ch = oracle.Channel(sonde, c)
        ch.feed(host[c])
        rs, gs = ch.state(), b.state(c)
        assert (gs["t_next"], gs["period"], gs["bias"], gs["amp"]) == (rs["t_next"], rs["period"], rs["bias"], rs["amp"]), c
        nb = b.nbits(c)
        assert nb == len(ch.bits()) and np.array_equal(b.read_bits(c, nb - 4000, 4000), ch.bits()[-4000:])
    b.close()


@pytest.mark.parametrize("flags,seg,bits", [(32, 2, 0), (32, 3, 0), (4, 2, 0), (4, 4, 16), (0, 0, 0), (0, 0, 16)])
def test_sliced_mixed_batch_equals_oracle(oracle, flags, seg, bits):
    """RS41 / M10 / DFM09 by channel % 3.  Late-joined and never joined: one launch unit per sonde type on its own stream, each sliced
    as told.  Default flags (seg 0): ONE launch over both demodulator classes (sd_demod_mixed_kernel, round 6; never sliced), float and
    16-bit integer rows."""
    from sdrpp_radiosonde_amd import _lib
    C, tiles, NS = 3072, 24, 3
    n = tiles * TILE
    order = (0, 3, 1)
    types = np.array([order[c % 3] for c in range(C)], dtype=np.uint8)
    iq = torch.empty((C, NS * n, 2), dtype=torch.float32, device="cuda:0")
    for t in order:
        idx = np.nonzero(types == t)[0]
        sb = synth.make_batch(int(t), len(idx), NS * n, seed=900 + int(t), ebn0_db=15.0, device="cuda:0")
        iq[torch.from_numpy(idx).to("cuda:0")] = sb.iq
        del sb
    if bits:
        q = torch.clamp(torch.round(iq * 8192.0), -32768, 32767).to(torch.int16)
        iq = q.to(torch.float32)
    dev = q if bits else iq
    host = iq.cpu().numpy()
    refs = []
    for t in order:
        idx = np.nonzero(types == t)[0]
        r = oracle.batch_run(int(t), host[idx], nthreads=CORES)
        r["channel"] = idx[r["channel"]]
        refs.append(r)
    ref = _key(np.concatenate(refs))
    b = SondeBatch(C, n, types=types, flags=flags, input_kind=_lib.INPUT_IQ16 if bits else _lib.INPUT_IQ, time_slices=seg)
    b.ticket()
    st = torch.cuda.current_stream().cuda_stream
    blocks = [strided_rows(dev[:, k * n: (k + 1) * n].contiguous()) for k in range(NS)]
    parts = []
    for k in range(NS):
        b.submit(blocks[k], st)
        if k >= 1:
            parts.append(b.frames_of(k))
    parts.append(b.frames_of(NS))
    got = _key(np.concatenate(parts))
    b.close()
    assert len(ref) >= C and got.tobytes() == ref.tobytes()
