"""CPU: the post-FEC field parsers added in round 2 (csrc/parse.cpp) against what the generator encoded
(sdrpp_radiosonde_amd/synth.py shares no code with the parser) and, for the sensor conversions, against independent
double-precision restatements in oracle/or_physics.c -- within stated tolerances, not bit for bit.

These stand where sondedump's per-sonde subframe parsers fill SondeData (fields consumed at
/root/reference/src/decode/decoder.hpp:64-106; README.md:9-19 lists which sonde carries what)."""
import ctypes as C

import numpy as np
import pytest

from sdrpp_radiosonde_amd import _lib, synth


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def _feed(lib, h, stype, data, nerr=(0, 0), length=None):
    f = _lib.SondeFrame()
    f.type, f.len = stype, length if length is not None else len(data)
    f.nerr[0], f.nerr[1] = nerr
    C.memmove(f.data, np.ascontiguousarray(data).ctypes.data, len(data))
    out = (_lib.SondeData * 8)()
    n = lib.sonde_parser_feed(h, C.byref(f), out, 8)
    res = []
    for i in range(n):
        d = _lib.SondeData()
        C.memmove(C.byref(d), C.byref(out[i]), C.sizeof(d))
        res.append(d)
    return res


def test_rs41_sgp_pressure_and_xdata_ozone(lib, oracle):
    """RS41-SGP: fragment.pressure from the pressure sensor once calibration fragments 0x25..0x2A are in (before that,
    and on an RS41-SG, 0 -> the adaptor's ISA fallback, decoder.hpp:108-110); extended frames: DATA_OZONE from XDATA."""
    O = oracle.lib()
    nfr = 60
    ch = np.full(nfr, 17)
    frames = synth.rs41_build_frames(3, ch, np.arange(nfr), extended=True, sgp=True)
    Pt = synth.rs41_true_pressure(np.arange(nfr))
    h = lib.sonde_parser_create(0)
    first_p, n_o3 = None, 0
    for k in range(nfr):
        for d in _feed(lib, h, 0, frames[k]):
            if d.fields & _lib.DATA_PTU:
                if d.pressure > 0:
                    first_p = k if first_p is None else first_p
                    assert abs(d.pressure - Pt[k]) < 0.05, (k, d.pressure, Pt[k])     # count quantisation: 0.02 hPa
            if d.fields & _lib.DATA_OZONE:
                cur, tp = synth.ozone_true(k)
                ref = O.or_ozone_mpa_d(round(float(cur) * 1e4) / 1e4, round(float(tp) * 100) / 100)
                assert abs(d.o3_mpa - ref) < 1e-4 * ref
                n_o3 += 1
    lib.sonde_parser_destroy(h)
    # sequence numbers start at 1000 -> calibration fragment 1000 % 51 = 31: the pressure block (fragments 0x25..0x2A) is
    # complete 11 frames later, the temperature words (fragments 3..7) 27 frames later -- the first PTU fragment carries both
    assert first_p == max((0x2A - 1000 % 51) % 51, (7 - 1000 % 51) % 51) and n_o3 == nfr
    # the sensor-less RS41-SG keeps answering 0 with the same calibration memory
    sg = synth.rs41_build_frames(3, ch, np.arange(nfr), sgp=False)
    h = lib.sonde_parser_create(0)
    ps = [d.pressure for k in range(nfr) for d in _feed(lib, h, 0, sg[k]) if d.fields & _lib.DATA_PTU]
    lib.sonde_parser_destroy(h)
    assert len(ps) > 20 and all(p == 0.0 for p in ps)


def test_rs41_pressure_polynomial_against_independent_double(lib, oracle):
    O = oracle.lib()
    rng = np.random.default_rng(3)
    for _ in range(500):
        cf = rng.normal(0, 1, 25).astype(np.float32) * np.float32(10.0) ** rng.integers(-3, 3, 25).astype(np.float32)
        cf[24] = np.float32(rng.uniform(0.3, 0.9))
        f1 = int(rng.integers(100000, 150000)); f2 = f1 + int(rng.integers(30000, 80000)); f = int(rng.integers(f1 + 5000, f2 + 20000))
        t = float(rng.uniform(-60, 40))
        cp = cf.ctypes.data_as(C.POINTER(C.c_float))
        a, b = lib.sonde_rs41_pressure(f, f1, f2, t, cp), O.or_rs41_pressure_d(f, f1, f2, t, cp)
        scale = sum(abs(float(cf[4 * j + k])) * (float(cf[24]) / ((f - f1) / (f2 - f1))) ** j * abs(t) ** k for j in range(6) for k in range(4))
        assert abs(a - b) <= 2e-5 * scale + 1e-6, (a, b, scale)          # 24 single-precision terms: a few ulp of the largest


def test_dfm_counter_and_time_layout(lib):
    """frame counter = 8 bits at bit 24 of the 48-bit payload, UTC ms of minute = its last 16 bits."""
    nfr = 12
    cw, _ = synth.dfm_build_frames(5, np.full(nfr, 9), np.arange(300, 300 + nfr))
    h = lib.sonde_parser_create(1)
    seqs, times = [], []
    for k in range(nfr):
        for d in _feed(lib, h, 1, cw[k]):
            if d.fields & _lib.DATA_SEQ:
                seqs.append(d.seq)
            if d.fields & _lib.DATA_TIME:
                times.append((300 + k, d.time))
    lib.sonde_parser_destroy(h)
    assert seqs == [(300 + k) & 0xFF for k in range(nfr) if (300 + k) % 3 == 0]
    import calendar
    for fi, t in times:       # date block (id 8) of the previous 3-frame group: 2024-06-15 12:mm, mm = its frame index % 60
        mm = [(fi - j) % 60 for j in range(1, 4) if (fi - j) % 3 == 2][0]
        assert t == calendar.timegm((2024, 6, 15, 12, mm, 0)) + ((fi * 1000 + 123000) % 60000) // 1000


def test_m10_ptu_and_m20_frames(lib, oracle):
    O = oracle.lib()
    nfr = 20
    ch, fi = np.full(nfr, 6), np.arange(nfr)
    m10 = synth.m10_build_frames(7, ch, fi)
    Tt, RHt = synth.m10_true_ptu(ch, fi)
    h = lib.sonde_parser_create(3)
    for k in range(nfr):
        fr = _feed(lib, h, 3, m10[k])
        assert [d.fields for d in fr] == [_lib.DATA_POS | _lib.DATA_SPEED | _lib.DATA_TIME, _lib.DATA_PTU]
        assert abs(fr[1].temp - Tt[k]) < 0.05 and abs(fr[1].rh - RHt[k]) < 0.02, (k, fr[1].temp, Tt[k], fr[1].rh, RHt[k])
    m20 = synth.m20_build_frames(7, ch, fi)
    T2 = synth.m20_true_temp(ch, fi)
    for k in range(nfr):
        fr = _feed(lib, h, 3, m20[k], length=70)
        assert len(fr) == 2 and fr[0].fields == _lib.DATA_POS | _lib.DATA_SPEED | _lib.DATA_TIME and fr[1].fields == _lib.DATA_PTU
        assert abs(fr[0].lat - 47.006) < 1e-4 and abs(fr[0].lon - (8.0 + 1e-5 * k)) < 1e-4 and abs(fr[0].alt - (1000 + 5 * k)) < 0.01
        assert abs(fr[0].speed - 12.0) < 1e-3 and abs(fr[0].heading - 90.0) < 1e-3 and abs(fr[0].climb - 5.0) < 1e-3
        assert fr[0].time == 315964800 + 2200 * 604800 + (k + 123456) - 18
        assert abs(fr[1].temp - T2[k]) < 0.05
    assert _feed(lib, h, 3, m20[0], nerr=(-1, 0), length=70) == []          # checksum failure: nothing
    lib.sonde_parser_destroy(h)
    rng = np.random.default_rng(2)
    for _ in range(300):
        sc, adc = int(rng.integers(0, 3)), int(rng.integers(40, 4050))
        a, b = lib.sonde_m10_temp(sc, adc), O.or_m10_temp_d(sc, adc)
        assert abs(a - b) < 2e-3, (sc, adc, a, b)
        a, b = lib.sonde_m20_temp(adc), O.or_m20_temp_d(adc)
        assert abs(a - b) < 2e-3
        ref = int(rng.integers(50000, 200000)); sen = int(ref * rng.uniform(0.85, 1.15)); T = float(rng.uniform(-80, 40))
        assert abs(lib.sonde_m10_rh(sen, ref, T) - O.or_m10_rh_d(sen, ref, T)) < 2e-2


def test_ims100_fields(lib, oracle):
    """iMS-100 / RS-11G (row a6): seq / serial, time, position + speed, PTU once the three polynomial words have arrived;
    a word with a wrong parity bit voids the fields it belongs to; an uncorrectable BCH block voids the frame.
    SELF-REFERENTIAL: the field layout this parser assumes is the repo's own and the generator (synth.py) shares it;
    this test pins the parser against the generator, not against a recorded sonde -- the parser is marked experimental.
    """
    O = oracle.lib()
    nfr = 12
    ch, fi = np.full(nfr, 21), np.arange(40, 40 + nfr)
    data, _ = synth.ims_build_frames(9, ch, fi)
    Tt, RHt = synth.ims_true_ptu(ch, fi)
    h = lib.sonde_parser_create(2)
    n_ptu = 0
    for k in range(nfr):
        fr = {d.fields: d for d in _feed(lib, h, 2, data[k])}
        seqf = _lib.DATA_SEQ | (_lib.DATA_SERIAL if fi[k] % 4 == 0 else 0)
        assert seqf in fr and fr[seqf].seq == fi[k]
        if fi[k] % 4 == 0:
            assert fr[seqf].serial == b"5000021"
        t = fr[_lib.DATA_TIME]
        assert t.time == 315964800 + 2200 * 604800 + ((fi[k] * 500 + 123456000) % 604800000) // 1000 - 18
        p = fr[_lib.DATA_POS | _lib.DATA_SPEED]
        assert abs(p.lat - 47.021) < 1e-5 and abs(p.lon - (8.0 + 1e-5 * fi[k])) < 1e-5 and abs(p.alt - (1000 + 2.5 * fi[k])) < 0.01
        assert abs(p.speed - 12.0) < 1e-4 and abs(p.heading - 90.0) < 1e-4 and abs(p.climb - 5.0) < 1e-4
        if _lib.DATA_PTU in fr:
            n_ptu += 1
            assert k >= 3 and abs(fr[_lib.DATA_PTU].temp - Tt[k]) < 2e-3 and abs(fr[_lib.DATA_PTU].rh - RHt[k]) < 0.006
        else:
            assert k < 3                                     # calibration words 1..3 arrive with counters 41, 42, 43
    assert n_ptu == nfr - 3
    # parity: flip one data bit of word 9 (latitude): position fragment disappears, the others stay
    bad = data[5].copy()
    bit = 17 * 9 + 4
    bad[bit >> 3] ^= 0x80 >> (bit & 7)
    fr = [d.fields for d in _feed(lib, h, 2, bad)]
    assert _lib.DATA_POS | _lib.DATA_SPEED not in fr and _lib.DATA_TIME in fr
    assert _feed(lib, h, 2, data[6], nerr=(0, 1)) == []
    lib.sonde_parser_destroy(h)
    for f in (0, 1000, 27000, 65535):
        assert abs(lib.sonde_ims100_temp(f, *synth.IMS_CAL) - O.or_ims100_temp_d(f, *synth.IMS_CAL)) < 1e-4


def test_imet_xdata_ozone(lib, oracle):
    O = oracle.lib()
    h = lib.sonde_parser_create(4)
    for k in (0, 17, 300):
        pk = synth.imet_build_packets(5, k, xdata=True)
        assert len(pk) == 3 and len(pk[2]) == 13
        fr = _feed(lib, h, 4, pk[2])
        cur, tp = synth.ozone_true(k)
        ref = O.or_ozone_mpa_d(round(float(cur) * 1000) / 1000, round(float(tp) * 100) / 100)
        assert len(fr) == 1 and fr[0].fields == _lib.DATA_OZONE and abs(fr[0].o3_mpa - ref) < 1e-4 * ref
    lib.sonde_parser_destroy(h)


def test_mrzn1_fields(lib):
    """SELF-REFERENTIAL: the field layout this parser assumes is the repo's own and the generator (synth.py) shares it;
    this test pins the parser against the generator, not against a recorded sonde -- the parser is marked experimental."""
    import calendar
    nfr = 8
    ch, fi = np.full(nfr, 12), np.arange(nfr)
    fr = synth.mrz_build_frames(4, ch, fi)
    h = lib.sonde_parser_create(6)
    for k in range(nfr):
        got = {d.fields: d for d in _feed(lib, h, 6, fr[k])}
        seqf = _lib.DATA_SEQ | (_lib.DATA_SERIAL if k % 4 == 0 else 0)
        assert got[seqf].seq == k and (k % 4 or got[seqf].serial == b"MRZ-7000012")
        assert got[_lib.DATA_TIME].time == calendar.timegm((2024, 6, 15, 12, 34, 56)) + k
        p = got[_lib.DATA_POS | _lib.DATA_SPEED]
        assert abs(p.lat - 47.012) < 1e-5 and abs(p.lon - (8.0 + 1e-5 * k)) < 1e-5 and abs(p.alt - (1000 + 5 * k)) < 0.02
        assert abs(p.speed - 12.0) < 0.01 and abs(p.heading - 90.0) < 0.05 and abs(p.climb - 5.0) < 0.01
        assert abs(got[_lib.DATA_PTU].temp - synth.mrz_true_temp(ch, fi)[k]) < 0.006
    assert _feed(lib, h, 6, fr[0], nerr=(-1, 0)) == []                # CRC failure: nothing
    lib.sonde_parser_destroy(h)
