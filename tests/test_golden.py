"""Committed fixtures tests/golden/rs41_golden.npz (made by tests/golden/make_golden.py) and others_golden.npz (the other six
sonde types, make_others_golden.py): CPU: the oracle still produces them.  GPU: the HIP path produces them too."""
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


OTHERS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "others_golden.npz")


@pytest.mark.parametrize("t", [1, 2, 3, 4, 5, 6])
def test_oracle_reproduces_others_golden(oracle, t):
    g = np.load(OTHERS)
    iq = g[f"iq{t}"].astype(np.float32) / np.float32(100.0)
    ch = oracle.Channel(t, 0)
    ch.feed(iq)
    assert len(ch.bits()) == int(g[f"nbits{t}"][0])
    assert np.array_equal(np.packbits(ch.bits(), bitorder="little"), g[f"bits{t}"])
    fr = ch.frames()
    assert len(fr) >= 1 and fr.view(np.uint8).reshape(len(fr), -1).tobytes() == g[f"frames{t}"].tobytes()


@pytest.mark.gpu
def test_hip_path_reproduces_others_golden():
    """every type through its own one-channel batch: bits and frame records equal the fixture's"""
    from sdrpp_radiosonde_amd.batch import SondeBatch
    g = np.load(OTHERS)
    for group in ((1, 2, 3, 6, 4, 5),):
        for t in group:
            iq = torch.from_numpy(g[f"iq{t}"].astype(np.float32) / np.float32(100.0))[None].to("cuda:0")
            b = SondeBatch(1, iq.shape[1], types=np.array([t], dtype=np.uint8))
            b.submit(iq)
            got = b.frames()
            nb = int(g[f"nbits{t}"][0])
            assert b.nbits(0) == nb, t
            assert np.array_equal(np.packbits(b.read_bits(0, 0, nb), bitorder="little"), g[f"bits{t}"]), t
            assert got.view(np.uint8).reshape(len(got), -1).tobytes() == g[f"frames{t}"].tobytes(), t
            b.close()

