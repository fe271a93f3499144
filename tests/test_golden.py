"""Committed fixture tests/golden/rs41_golden.npz (made by tests/golden/make_golden.py):
CPU: the oracle still produces it.  GPU: the HIP path produces it too."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rs41_golden.npz")


def _load():
    g = np.load(GOLD)
    iq = g["iq_int8"].astype(np.float32) / np.float32(100.0)
    return g, iq


def test_oracle_reproduces_golden(oracle):
    g, iq = _load()
    frames = []
    for c in range(iq.shape[0]):
        ch = oracle.Channel(0, c)
        ch.feed(iq[c])
        assert np.array_equal(np.packbits(ch.bits(), bitorder="little"), g["bits"][c])
        s = ch.state()
        exp = g["states"][c]
        assert (s["t_next"], s["period"], len(ch.bits())) == (exp[0], exp[1], exp[5])
        for k, i in (("bias", 2), ("amp", 3), ("yprev", 4)):
            assert np.float32(s[k]).view(np.int32) == exp[i]
        frames.append(ch.frames())
    frames = np.concatenate(frames)
    assert frames.view(np.uint8).reshape(len(frames), -1).tobytes() == g["frames"].tobytes()
    assert len(frames) >= 3 and (frames["nerr"] > 0).any()      # the fixture exercises the RS corrector
    for f in frames:
        if (f["nerr"] >= 0).all():
            assert any(np.array_equal(t[8:], f["data"][8:320]) for t in g["tx_frames"])


@pytest.mark.gpu
def test_hip_path_reproduces_golden():
    from sdrpp_radiosonde_amd.batch import SondeBatch
    g, iq = _load()
    C, n = iq.shape[0], iq.shape[1]
    b = SondeBatch(C, n)
    b.submit(torch.from_numpy(iq).to("cuda:0"))
    got = b.frames()
    assert got.view(np.uint8).reshape(len(got), -1).tobytes() == g["frames"].tobytes()
    for c in range(C):
        nb = int(g["states"][c][5])
        assert b.nbits(c) == nb
        assert np.array_equal(np.packbits(b.read_bits(c, 0, nb), bitorder="little"), g["bits"][c])
        s = b.state(c)
        assert (s["t_next"], s["period"]) == (g["states"][c][0], g["states"][c][1])
