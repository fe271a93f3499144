"""CPU tests of the C-ABI library: it loads, exports every symbol include/sonde_abi.h declares,
its host-side tables equal the oracle's, and it fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from sdrpp_radiosonde_amd import _lib
from sdrpp_radiosonde_amd.batch import SondeBatch, SondeError, get_taps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "sonde_abi.h")).read()
    declared = set(re.findall(r"\b(sonde_[a-z0-9_]+)\s*\(", hdr))
    for x in re.findall(r"^SONDE_B1_DECL\(\w+,\s*(\w+)\)", hdr, flags=re.M):
        declared |= {f"{x}_decoder_init", f"{x}_decoder_deinit", f"{x}_decode"}
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.sonde_version()


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.SondeFrame) == 560 and _lib.FRAME_DTYPE.itemsize == 560
    assert _lib.SondeFrame.bitpos.offset == 24 and _lib.SondeFrame.data.offset == 32
    assert C.sizeof(_lib.SondeBatchConfig) == 48 and _lib.SondeBatchConfig.launch_units.offset == 32      # (round 5: + launch_units)
    assert _lib.SondeBatchConfig.struct_size.offset == 36 and _lib.SondeBatchConfig().struct_size == 48  # (round 6: + struct_size, time_slices)
    assert _lib.SondeBatchConfig.time_slices.offset == 40
    assert (_lib.FLAG_JOIN, _lib.FLAG_LATE_JOIN, _lib.FLAG_PIPELINE) == (16, 32, 4)


def test_config_struct_size_is_checked(lib):
    """a caller built against an older header (or one that did not zero the struct) must fail loudly, not hand over
    garbage as launch_units; launch_units itself is range-checked."""
    h = C.c_void_p()
    cfg = _lib.SondeBatchConfig()
    cfg.n_channels, cfg.max_samples = 4, 2048
    for bad in (0, 32, 40, 0xDEADBEEF):
        cfg.struct_size = bad
        assert lib.sonde_batch_create(C.byref(cfg), C.byref(h)) != 0 and b"struct_size" in lib.sonde_last_error()
    cfg.struct_size = C.sizeof(cfg)
    cfg.launch_units = 17
    assert lib.sonde_batch_create(C.byref(cfg), C.byref(h)) != 0 and b"launch_units" in lib.sonde_last_error()
    cfg.launch_units, cfg.time_slices = 0, 99
    assert lib.sonde_batch_create(C.byref(cfg), C.byref(h)) != 0 and b"time_slices" in lib.sonde_last_error()


def test_tap_tables_equal_oracle_bit_for_bit(lib, oracle):
    L = oracle.lib()
    for t in range(4):
        ref = np.zeros((32, 32), dtype=np.float32)
        L.or_make_taps(L.or_modem(t), oracle.fptr(ref.reshape(-1)))
        got = get_taps(t)
        assert got.tobytes() == ref.tobytes()
        assert np.allclose(got.sum(axis=1), 1.0, atol=1e-6)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback(lib):
    with pytest.raises(SondeError):
        SondeBatch(4, 2048)
    assert lib.rs41_decoder_init(48000) is None        # init fails, it does not fall back


def test_argument_validation(lib):
    cfg = _lib.SondeBatchConfig()
    h = C.c_void_p()
    cfg.n_channels, cfg.max_samples = 0, 2048
    assert lib.sonde_batch_create(C.byref(cfg), C.byref(h)) != 0 and b"n_channels" in lib.sonde_last_error()
    cfg.n_channels, cfg.max_samples = 4, 1000
    assert lib.sonde_batch_create(C.byref(cfg), C.byref(h)) != 0 and b"multiple" in lib.sonde_last_error()
    assert lib.rs41_decoder_init(44100) is None        # reference always passes 48000 (main.cpp:16,62-68)
    cfg.n_channels, cfg.max_samples, cfg.input_kind = 4, 2048, 7
    assert lib.sonde_batch_create(C.byref(cfg), C.byref(h)) != 0 and b"input_kind" in lib.sonde_last_error()


def test_row_stride_is_monotonic(lib):
    """buffers sized from sonde_row_stride(max_samples) are written at sonde_row_stride(n_samples), n_samples <= max_samples:
    the stride must never shrink as the row grows (the round-4 rule gave rows of 80 KiB a 192 KiB stride and rows of 96 KiB 128 KiB),
    must hold the row, and stay a multiple of 16 bytes."""
    for kind in (_lib.INPUT_IQ, _lib.INPUT_REAL, _lib.INPUT_IQ16, _lib.INPUT_IQ8):
        eb = lib.sonde_sample_bytes(kind)
        prev = 0
        for tiles in range(1, 400):
            n = 2048 * tiles
            st = lib.sonde_row_stride(n, kind)
            assert st >= n and st >= prev and (st * eb) % 16 == 0, (kind, n, st, prev)
            assert st * eb <= max(2 * n * eb, 65536), (kind, n, st)          # never more than twice the row
            prev = st
    assert lib.sonde_row_stride(10240, _lib.INPUT_IQ) == 16384 and lib.sonde_row_stride(12288, _lib.INPUT_IQ) == 16384


def test_input_kinds_sample_bytes_and_row_stride(lib):
    """complex64, float discriminator samples, 16-bit integer IQ: element sizes, and the recommended row stride in elements
    (the same BYTES for the same row: 1.5 MiB rows go 2 MiB apart whatever they hold)"""
    assert [lib.sonde_sample_bytes(k) for k in (_lib.INPUT_IQ, _lib.INPUT_REAL, _lib.INPUT_IQ16, _lib.INPUT_IQ8)] == [8, 4, 4, 2]
    assert lib.sonde_row_stride(2 * 2048 * 96, _lib.INPUT_IQ8) == 524288        # 0.75 MiB -> 1 MiB
    n = 2048 * 96
    assert lib.sonde_row_stride(n, _lib.INPUT_IQ) == 262144            # 1.5 MiB -> 2 MiB
    assert lib.sonde_row_stride(n, _lib.INPUT_IQ16) == 262144          # 0.75 MiB -> 1 MiB
    assert lib.sonde_row_stride(2 * n, _lib.INPUT_IQ16) == 524288      # 1.5 MiB -> 2 MiB


def test_parse_frame_fields(lib):
    """Frame -> SondeData fragments: seq/serial, time, position+speed of a generated frame."""
    from sdrpp_radiosonde_amd import synth
    fr = synth.rs41_build_frames(3, np.array([42]), np.array([7]))[0]
    f = _lib.SondeFrame()
    f.type, f.len = 0, 320
    C.memmove(f.data, fr.ctypes.data, 320)
    out = (_lib.SondeData * 8)()
    n = lib.sonde_parse_frame(C.byref(f), out, 8)
    frags = {out[i].fields: out[i] for i in range(n)}
    a = frags[_lib.DATA_SEQ | _lib.DATA_SERIAL]
    assert a.seq == 1007 and a.serial == b"S0000042"
    t = frags[_lib.DATA_TIME]
    assert t.time == 315964800 + 2200 * 604800 + (123456000 + 7000) // 1000 - 18
    p = frags[_lib.DATA_POS | _lib.DATA_SPEED]
    assert abs(p.lat - np.degrees(np.radians(47.0) + 42e-5)) < 1e-4 and abs(p.lon - np.degrees(np.radians(8.0) + 7e-6)) < 1e-4
    assert abs(p.alt - 1035.0) < 0.5 and abs(p.climb - 5.0) < 0.02 and abs(p.speed - 12.0) < 0.02 and abs(p.heading - 90.0) < 0.2
    # a corrupted subframe is dropped (CRC), the others survive
    f.data[60] ^= 0xFF
    n2 = lib.sonde_parse_frame(C.byref(f), out, 8)
    assert n2 == n - 1


def test_rs41_ptu_through_the_stateful_parser(lib):
    """RS41 temperature / humidity: the calibration table arrives 16 bytes per frame, so PTU fragments start once
    the fragments holding Rf1, Rf2, the polynomial and calH have been seen; values match what the generator encoded."""
    from sdrpp_radiosonde_amd import synth
    nfr = 70
    frames = synth.rs41_build_frames(3, np.full(nfr, 42), np.arange(nfr))
    Tt, RHt = synth.rs41_true_ptu(np.full(nfr, 42), np.arange(nfr))
    h = lib.sonde_parser_create(0)
    out = (_lib.SondeData * 8)()
    first_ptu, pct, kills = None, [], []
    for k in range(nfr):
        f = _lib.SondeFrame()
        f.type, f.len = 0, 320
        C.memmove(f.data, frames[k].ctypes.data, 320)
        n = lib.sonde_parser_feed(h, C.byref(f), out, 8)
        ptu = [out[i] for i in range(n) if out[i].fields & _lib.DATA_PTU]
        kills += [(k, out[i].shutdown) for i in range(n) if out[i].fields & _lib.DATA_SHUTDOWN]
        if ptu:
            if first_ptu is None:
                first_ptu = k
            assert abs(ptu[0].temp - Tt[k]) < 0.02, (k, ptu[0].temp, Tt[k])
            assert abs(ptu[0].rh - RHt[k]) < 0.05, (k, ptu[0].rh, RHt[k])
            assert ptu[0].pressure == 0.0                      # RS41-SG: the adaptor falls back to the ISA model
            pct.append(ptu[0].calib_percent)
    lib.sonde_parser_destroy(h)
    # channel 42 is even: burst-kill timer not armed -> no shutdown fragment; an odd channel reports its countdown
    assert kills == []
    fr43 = synth.rs41_build_frames(3, np.full(nfr, 43), np.arange(nfr))
    h2 = lib.sonde_parser_create(0)
    k43 = []
    for k in range(nfr):
        f = _lib.SondeFrame()
        f.type, f.len = 0, 320
        C.memmove(f.data, fr43[k].ctypes.data, 320)
        n = lib.sonde_parser_feed(h2, C.byref(f), out, 8)
        k43 += [(k, out[i].shutdown) for i in range(n) if out[i].fields & _lib.DATA_SHUTDOWN]
    lib.sonde_parser_destroy(h2)
    assert k43 == [((0x31 - 1000 % 51) % 51, 3643), ((0x31 - 1000 % 51) % 51 + 51, 3643)]
    # sequence numbers start at 1000 -> fragment index 1000 % 51 = 31; fragments 3..7 are complete 28 frames later
    assert first_ptu == (7 - 1000 % 51) % 51
    assert pct == sorted(pct) and abs(pct[-1] - 100.0) < 1e-3
    # the stateless entry point has no calibration memory: no PTU fragment
    f = _lib.SondeFrame()
    f.type, f.len = 0, 320
    C.memmove(f.data, frames[60].ctypes.data, 320)
    n = lib.sonde_parse_frame(C.byref(f), out, 8)
    assert not any(out[i].fields & _lib.DATA_PTU for i in range(n))


def test_rs41_conversions_equal_oracle_bit_for_bit(lib):
    import oracle_lib
    O = oracle_lib.lib()
    rng = np.random.default_rng(5)
    co = (C.c_float * 3)(-243.911, 0.187654, 8.2e-06)
    cal = (C.c_float * 3)(1.0007, 0.013, -2e-4)
    for _ in range(2000):
        f1 = int(rng.integers(100000, 150000)); f2 = f1 + int(rng.integers(30000, 80000))
        f = int(rng.integers(f1 - 20000, f2 + 60000))
        rf1 = float(rng.uniform(700, 800)); rf2 = float(rng.uniform(1050, 1150))
        a = np.float32(lib.sonde_rs41_temp(f, f1, f2, rf1, rf2, co, cal))
        b = np.float32(O.or_rs41_temp(f, f1, f2, rf1, rf2, co, cal))
        assert a.tobytes() == b.tobytes()
        T = float(rng.uniform(-90, 40))
        a = np.float32(lib.sonde_rs41_rh(f, f1, f2, 45.0, T))
        b = np.float32(O.or_rs41_rh(f, f1, f2, 45.0, T))
        assert a.tobytes() == b.tobytes() and (0.0 <= a <= 100.0)


def test_dfm_temperature_through_the_parser(lib):
    """DFM: the CONF block delivers one measurement channel per frame (24-bit floats); temperature fragments start
    once channels 0, 3 and 4 have been seen and match what the generator encoded; product == oracle bit for bit."""
    import oracle_lib
    from sdrpp_radiosonde_amd import synth
    O = oracle_lib.lib()
    nfr = 30
    cw, _ = synth.dfm_build_frames(5, np.full(nfr, 9), np.arange(nfr))
    Tt = synth.dfm_true_temp(np.full(nfr, 9), np.arange(nfr))
    h = lib.sonde_parser_create(1)
    out = (_lib.SondeData * 8)()
    got = {}
    for k in range(nfr):
        f = _lib.SondeFrame()
        f.type, f.len = 1, 33
        C.memmove(f.data, cw[k].ctypes.data, 33)
        n = lib.sonde_parser_feed(h, C.byref(f), out, 8)
        for i in range(n):
            if out[i].fields & _lib.DATA_PTU:
                got[k] = out[i].temp
    lib.sonde_parser_destroy(h)
    assert sorted(got) == [7, 14, 21, 28]                      # CONF id 0 comes every 7th frame; the first one lacks refs
    for k, T in got.items():
        assert abs(T - Tt[k]) < 0.05, (k, T, Tt[k])
    rng = np.random.default_rng(1)
    for _ in range(500):
        f1 = float(rng.uniform(500, 2000)); f2 = float(rng.uniform(30000, 60000)); f = f1 + float(rng.uniform(-10, 4e5))
        a, b = np.float32(lib.sonde_dfm_temp(f, f1, f2)), np.float32(O.or_dfm_temp(f, f1, f2))
        assert a.tobytes() == b.tobytes()
