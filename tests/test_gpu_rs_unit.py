"""-m gpu: the RS(255,231) corrector alone (sonde_batch_test_rs255: rs255_decode_pair of sd_rsdec.h on caller-supplied
codeword pairs) against the oracle's textbook decoder (or_rs255_decode) -- thousands of random error patterns of every weight
0..16, both codewords of a pair with different weights (so that the closed forms for one and two errors, the general locator
and the clean path meet inside one wave), errors confined to the parity bytes, and the cases no transmitted frame can
provoke: words whose nearest codeword differs in the PADDING of the shortened code (roots outside [0, n): must be rejected)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from sdrpp_radiosonde_amd import _lib
from sdrpp_radiosonde_amd.batch import SondeBatch

pytestmark = pytest.mark.gpu


def _encode(msg_full):
    """systematic full-length codeword: positions 0..23 parity, 24..254 message (or_rs255_encode's layout)"""
    L = oracle_lib.lib()
    cw = np.zeros(255, dtype=np.uint8)
    cw[24:] = msg_full
    L.or_rs255_encode(oracle_lib.u8ptr(cw), 255)
    return cw


def _oracle(cw, n):
    L = oracle_lib.lib()
    w = np.zeros(255, dtype=np.uint8)
    w[:n] = cw[:n]
    st = L.or_rs255_decode(oracle_lib.u8ptr(w), n)
    return st, w


@pytest.mark.parametrize("n", [24 + 132, 24 + 231])
def test_corrector_against_textbook_decoder(n):
    rng = np.random.default_rng(n)
    words = []
    # random messages, weights 0..16 anywhere / in the parity only / bursts
    for trial in range(1500):
        msg = np.zeros(231, dtype=np.uint8)
        msg[: n - 24] = rng.integers(0, 256, size=n - 24)
        cw = _encode(msg)
        e = int(rng.integers(0, 17)) if trial % 3 else int(rng.integers(0, 4))
        if trial % 7 == 0:
            pos = rng.choice(24, size=min(e, 24), replace=False)
        elif trial % 11 == 0 and e:
            start = int(rng.integers(0, n - e))
            pos = np.arange(start, start + e)
        else:
            pos = rng.choice(n, size=e, replace=False)
        r = cw.copy()
        r[pos] ^= rng.integers(1, 256, size=len(pos)).astype(np.uint8)
        words.append(r)
    if n < 255:
        # the nearest codeword has 1, 2 or 3 non-zero bytes in the padding: the received word (padding zeroed) is at that
        # distance from it, so the locator's roots lie outside [0, n) -- plus up to two ordinary errors inside
        for trial in range(600):
            k = 1 + trial % 3
            msg = np.zeros(231, dtype=np.uint8)
            msg[: n - 24] = rng.integers(0, 256, size=n - 24)
            pad = rng.choice(np.arange(n - 24, 231), size=k, replace=False)
            msg[pad] = rng.integers(1, 256, size=k)
            r = _encode(msg)
            r[n:] = 0
            inside = rng.choice(n, size=trial % 3, replace=False)
            r[inside] ^= rng.integers(1, 256, size=len(inside)).astype(np.uint8)
            words.append(r)
    # pure noise
    for trial in range(200):
        r = np.zeros(255, dtype=np.uint8)
        r[:n] = rng.integers(0, 256, size=n)
        words.append(r)
    if len(words) % 2:
        words.append(words[0])
    order = rng.permutation(len(words))
    words = [words[i] for i in order]
    pairs = np.zeros((len(words) // 2, 2, 256), dtype=np.uint8)
    for i, w in enumerate(words):
        pairs[i // 2, i % 2, :n] = w[:n]
    status = np.zeros((len(pairs), 2), dtype=np.int32)
    b = SondeBatch(1, 2048)
    got = pairs.copy()
    rc = b.L.sonde_batch_test_rs255(b.h, got.ctypes.data_as(C.c_void_p), len(got), n, status.ctypes.data_as(C.c_void_p))
    assert rc == 0, _lib.last_error()
    seen = {}
    for i in range(len(pairs)):
        for c in range(2):
            st, w = _oracle(pairs[i, c], n)
            assert status[i, c] == st, (i, c, status[i, c], st)
            assert np.array_equal(got[i, c, :n], w[:n]), (i, c, st)
            assert not got[i, c, n:].any()
            seen[st] = seen.get(st, 0) + 1
    assert seen.get(-1, 0) > 200 and all(seen.get(k, 0) > 20 for k in range(0, 13)), seen
    b.close()


@pytest.mark.parametrize("n", [24 + 132, 255])
def test_corrector_against_independent_pgz_fixture(n):
    """tests/golden/rs255_pgz.npz: words, decisions and corrected words from an independent Peterson-Gorenstein-Zierler decoder
    (tests/golden/make_rs_fixtures.py: a direct solve of the syndrome matrix in pure Python, no Berlekamp-Massey, no Forney)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rs255_pgz.npz"))
    words, want_st, want = g[f"words{n}"], g[f"status{n}"], g[f"fixed{n}"]
    assert len(words) % 2 == 0
    pairs = np.zeros((len(words) // 2, 2, 256), dtype=np.uint8)
    pairs.reshape(-1, 256)[:, :255] = words
    status = np.zeros((len(pairs), 2), dtype=np.int32)
    b = SondeBatch(1, 2048)
    rc = b.L.sonde_batch_test_rs255(b.h, pairs.ctypes.data_as(C.c_void_p), len(pairs), n, status.ctypes.data_as(C.c_void_p))
    assert rc == 0, _lib.last_error()
    assert np.array_equal(status.reshape(-1), want_st)
    assert np.array_equal(pairs.reshape(-1, 256)[:, :n], want[:, :n])
    b.close()
