"""-m gpu: the B1 boundary -- X_decoder_init / X_decode / X_decoder_deinit exactly as
/root/reference/src/decode/decoder.hpp:22,39,61 drives them: real 48 kS/s discriminator samples in
arbitrary buffer sizes, the same buffer re-offered until PROCEED, one SondeData fragment per PARSED."""
import ctypes as C

import numpy as np
import pytest

from sdrpp_radiosonde_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _discriminate(oracle, iq):
    L = oracle.lib()
    n = iq.shape[0]
    d = np.zeros(n, dtype=np.float32)
    last = np.zeros(2, dtype=np.float32)
    L.or_discriminate(oracle.fptr(np.ascontiguousarray(iq).reshape(-1)), n, oracle.fptr(d), oracle.fptr(last))
    return d


def _run_b1(name, d, bufsize):
    L = _lib.load()
    dec = getattr(L, f"{name}_decoder_init")(48000)
    assert dec, _lib.last_error()
    frags, sd = [], _lib.SondeData()
    for off in range(0, len(d), bufsize):
        buf = np.ascontiguousarray(d[off: off + bufsize])
        calls = 0
        while getattr(L, f"{name}_decode")(dec, C.byref(sd), buf.ctypes.data_as(C.c_void_p), len(buf)) != _lib.PROCEED:
            assert sd.fields != 0                      # decoder.hpp:112 would drop it
            frags.append({k: getattr(sd, k) for k, _ in _lib.SondeData._fields_})
            calls += 1
            assert calls < 1000
    getattr(L, f"{name}_decoder_deinit")(dec)
    return frags


@pytest.mark.parametrize("bufsize", [4800, 1000, 50000])
def test_rs41_b1_stream(oracle, bufsize):
    n = 2048 * 80
    sb = synth.make_rs41_batch(1, n, seed=12, ebn0_db=24.0)
    d = _discriminate(oracle, sb.iq.numpy()[0])
    frags = _run_b1("rs41", d, bufsize)
    # reference: the oracle's frames (bit-exact to the GPU's, tests/test_gpu_parity.py) through the parser
    ch = oracle.Channel(0, 0)
    ch.feed(d[: (len(d) // 2048) * 2048], is_iq=False)
    L = _lib.load()
    expect = []
    out = (_lib.SondeData * 8)()
    for f in ch.frames():
        fr = _lib.SondeFrame.from_buffer_copy(f.tobytes())
        for i in range(L.sonde_parse_frame(C.byref(fr), out, 8)):
            expect.append({k: getattr(out[i], k) for k, _ in _lib.SondeData._fields_})
    assert len(expect) >= 9 and len(frags) == len(expect)
    for a, b in zip(frags, expect):
        assert a == b
    seqs = [f["seq"] for f in frags if f["fields"] & _lib.DATA_SEQ]
    assert seqs == list(range(seqs[0], seqs[0] + len(seqs))) and frags[0]["serial"] == b"S0000000"
    pos = [f for f in frags if f["fields"] & _lib.DATA_POS]
    assert all(abs(p["lat"] - 47.0) < 0.01 and abs(p["lon"] - 8.0) < 0.01 and abs(p["speed"] - 12.0) < 0.05 for p in pos)


def test_dfm_and_m10_b1_fields(oracle):
    for name, stype in (("dfm09", 1), ("m10", 3)):
        n = 2048 * 60
        sb = synth.make_batch(stype, 1, n, seed=31, ebn0_db=26.0)
        frags = _run_b1(name, _discriminate(oracle, sb.iq.numpy()[0]), 4096)
        pos = [f for f in frags if f["fields"] & _lib.DATA_POS]
        assert len(pos) >= 3, name
        for p in pos:
            assert abs(p["lat"] - 47.0) < 1e-3 and abs(p["lon"] - 8.0) < 1e-2 and 990.0 < p["alt"] < 1200.0, (name, p)
            assert abs(p["climb"] - 5.0) < 0.05 and abs(p["speed"] - 12.0) < 0.05 and abs(p["heading"] - 90.0) < 0.5, (name, p)
        times = [f["time"] for f in frags if f["fields"] & _lib.DATA_TIME]
        assert len(times) >= 2 and all(t > 1_400_000_000 for t in times), name


def test_silence_always_proceeds():
    """All seven decoders of main.hpp:36-42: a buffer of zeros yields no fragment (decoder.hpp:61 then flushes)."""
    L = _lib.load()
    sd = _lib.SondeData()
    buf = np.zeros(4096, dtype=np.float32)
    for name in ("rs41", "dfm09", "ims100", "m10", "imet4", "c50", "mrzn1"):
        dec = getattr(L, f"{name}_decoder_init")(48000)
        assert dec
        assert getattr(L, f"{name}_decode")(dec, C.byref(sd), buf.ctypes.data_as(C.c_void_p), 4096) == _lib.PROCEED
        getattr(L, f"{name}_decoder_deinit")(dec)


def test_batch_decoder_cpp(tmp_path, oracle):
    """sonde::BatchDecoder: the multi-channel counterpart of radiosonde::Decoder<> on top of sonde_batch_poll --
    one sticky aggregate per channel, dew point and ISA pressure as decoder.hpp:84-110, callback per fragment."""
    import os
    import re
    import subprocess
    from sdrpp_radiosonde_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "batch_decoder_test")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "batch_decoder_test.cpp"), "-o", exe,
                           "-L", libdir, "-l:libsonde_mi355.so", f"-Wl,-rpath,{libdir}"])
    Cn, n = 3, 2048 * 96 * 5                      # ~20 s: enough frames for the calibration fragments 3..7 -> PTU
    sb = synth.make_rs41_batch(Cn, n, seed=808, ebn0_db=30.0)
    path = str(tmp_path / "iq.bin")
    sb.iq.numpy().tofile(path)
    out = subprocess.check_output([exe, path, str(Cn), str(n), "5"], text=True)
    assert "ERROR" not in out, out[-2000:]
    cbs = [l for l in out.splitlines() if l.startswith("CB ")]
    done = [l for l in out.splitlines() if l.startswith("DONE")][0]
    assert int(re.search(r"fired=(\d+)", done).group(1)) == len(cbs) >= 3 * 30 * Cn
    last = {}
    for l in cbs:
        last[int(re.search(r"ch=(\d+)", l).group(1))] = l
    L = oracle.lib()
    for c in range(Cn):
        l = last[c]
        assert re.search(r"serial=(\S+)", l).group(1) == "S%07d" % c
        temp, rh = float(re.search(r"temp=(\S+)", l).group(1)), float(re.search(r"rh=(\S+)", l).group(1))
        alt = float(re.search(r" alt=(\S+)", l).group(1))
        assert -60 < temp < 20 and 0 < rh <= 100                       # PTU arrived (calibration complete enough)
        # RS41-SG carries no pressure: the aggregate holds the ISA value for the altitude at the time of the fall-back
        assert float.fromhex(re.search(r"pressure=(\S+)", l).group(1)) > 0
        dew = float.fromhex(re.search(r"dewpt=(\S+)", l).group(1))
        assert abs(dew - float(L.or_dewpt(np.float32(temp), np.float32(rh)))) < 0.05 and alt > 900


def test_ims100_b1_fields(oracle):
    """ims100_decode (main.hpp:38, row a6): GPS, time, T/RH fragments of an iMS-100 / RS-11G stream (README.md:14-15).
    SELF-REFERENTIAL: the field layout this parser assumes is the repo's own and the generator (synth.py) shares it;
    this test pins the parser against the generator, not against a recorded sonde -- the parser is marked experimental.
    """
    n = 2048 * 60
    sb = synth.make_batch(2, 1, n, seed=33, ebn0_db=26.0)
    frags = _run_b1("ims100", _discriminate(oracle, sb.iq.numpy()[0]), 4096)
    pos = [f for f in frags if f["fields"] & _lib.DATA_POS]
    assert len(pos) >= 3
    for p in pos:
        assert abs(p["lat"] - 47.0) < 1e-3 and abs(p["lon"] - 8.0) < 1e-2 and 990.0 < p["alt"] < 1100.0
        assert abs(p["climb"] - 5.0) < 0.01 and abs(p["speed"] - 12.0) < 0.01 and abs(p["heading"] - 90.0) < 0.01
    seqs = [f["seq"] for f in frags if f["fields"] & _lib.DATA_SEQ]
    assert seqs == list(range(seqs[0], seqs[0] + len(seqs)))
    assert any(f["fields"] & _lib.DATA_SERIAL and f["serial"] == b"5000000" for f in frags)
    ptu = [f for f in frags if f["fields"] & _lib.DATA_PTU]
    assert len(ptu) >= 2 and all(-60.0 < f["temp"] < 30.0 and 40.0 < f["rh"] < 60.0 for f in ptu)
    times = [f["time"] for f in frags if f["fields"] & _lib.DATA_TIME]
    assert len(times) >= 3 and all(t > 1_400_000_000 for t in times)


def test_m20_and_m10_ptu_b1(oracle):
    n = 2048 * 60
    for m20 in (False, True):
        sb = synth.make_batch(3, 1, n, seed=35, ebn0_db=26.0, m20=m20)
        frags = _run_b1("m10", _discriminate(oracle, sb.iq.numpy()[0]), 4096)
        pos = [f for f in frags if f["fields"] & _lib.DATA_POS]
        ptu = [f for f in frags if f["fields"] & _lib.DATA_PTU]
        assert len(pos) >= 3 and len(ptu) >= 3, m20
        assert all(abs(p["lat"] - 47.0) < 1e-3 and 990.0 < p["alt"] < 1200.0 for p in pos)
        assert all(-5.0 < f["temp"] < 15.0 for f in ptu)
        if not m20:
            assert all(30.0 < f["rh"] < 50.0 for f in ptu)


def test_rs41_sgp_pressure_and_ozone_b1(oracle):
    """RS41-SGP with an ozone interface: extended (518-byte) frames through the GPU decoder; the PTU fragments carry
    the sensor's pressure (decoder.hpp:89) and the XDATA subframe yields DATA_OZONE (decoder.hpp:102-106)."""
    n = 2048 * 640                        # 27 s: the calibration words of the pressure block arrive 16 bytes per frame
    sb = synth.make_rs41_batch(1, n, seed=14, ebn0_db=26.0, extended=True, sgp=True)
    frags = _run_b1("rs41", _discriminate(oracle, sb.iq.numpy()[0]), 48000)
    o3 = [f["o3_mpa"] for f in frags if f["fields"] & _lib.DATA_OZONE]
    assert len(o3) >= 20 and all(3.0 < v < 8.0 for v in o3)
    assert any(f["fields"] & _lib.DATA_SERIAL for f in frags)


def test_imet_xdata_b1(oracle):
    G = 16384
    sb = synth.make_imet_batch(1, G * 8, seed=31, snr_db=30.0, xdata=True)
    frags = _run_b1("imet4", _discriminate(oracle, sb.iq.numpy()[0]), 4800)
    o3 = [f["o3_mpa"] for f in frags if f["fields"] & _lib.DATA_OZONE]
    assert len(o3) >= 2 and all(4.0 < v < 6.0 for v in o3)
    assert sum(1 for f in frags if f["fields"] & _lib.DATA_PTU) >= 2


def test_mrzn1_b1_fields(oracle):
    """mrzn1_decode (main.hpp:42; README.md:19: GPS + temperature).
    SELF-REFERENTIAL: the field layout this parser assumes is the repo's own and the generator (synth.py) shares it;
    this test pins the parser against the generator, not against a recorded sonde -- the parser is marked experimental.
    """
    n = 2048 * 120
    sb = synth.make_batch(6, 1, n, seed=37, ebn0_db=26.0)
    frags = _run_b1("mrzn1", _discriminate(oracle, sb.iq.numpy()[0]), 4096)
    pos = [f for f in frags if f["fields"] & _lib.DATA_POS]
    assert len(pos) >= 3
    for p in pos:
        assert abs(p["lat"] - 47.0) < 1e-3 and abs(p["lon"] - 8.0) < 1e-2 and 990.0 < p["alt"] < 1100.0
        assert abs(p["climb"] - 5.0) < 0.02 and abs(p["speed"] - 12.0) < 0.02 and abs(p["heading"] - 90.0) < 0.2
    seqs = [f["seq"] for f in frags if f["fields"] & _lib.DATA_SEQ]
    assert seqs == list(range(seqs[0], seqs[0] + len(seqs)))
    assert any(f["fields"] & _lib.DATA_SERIAL and f["serial"] == b"MRZ-7000000" for f in frags)
    assert all(0.0 < f["temp"] < 12.0 for f in frags if f["fields"] & _lib.DATA_PTU)
    import calendar
    times = [f["time"] for f in frags if f["fields"] & _lib.DATA_TIME]
    assert len(times) >= 3 and all(0 <= t - calendar.timegm((2024, 6, 15, 12, 34, 56)) < 200 for t in times)


@pytest.mark.parametrize("chunk", [4800, 1000])
def test_b3_iq_stream_decoder_cpp(tmp_path, oracle, chunk):
    """sonde::IqStreamDecoder (B3, `dsp::stream<dsp::complex_t>` in): complex IQ at 48 kS/s in arbitrary buffer sizes ->
    the same frames the batch API decodes from the same samples -> merged FullData callbacks."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "iq_stream_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "iq_stream_test.cpp"), "-o", exe,
                           "-L", libdir, "-l:libsonde_mi355.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    n = 2048 * 96
    sb = synth.make_rs41_batch(1, n, seed=77, ebn0_db=25.0)
    path = str(tmp_path / "iq.bin")
    sb.iq.numpy()[0].tofile(path)
    out = subprocess.check_output([exe, path, "0", str(chunk)], text=True)
    assert "ERROR" not in out, out[-1500:]
    cbs = [l for l in out.splitlines() if l.startswith("CB ")]
    # reference: the oracle's frames of the same IQ through the stateful parser: one callback per fragment with fields
    ch = oracle.Channel(0, 0)
    ch.feed(sb.iq.numpy()[0])
    L = _lib.load()
    h = L.sonde_parser_create(0)
    o = (_lib.SondeData * 8)()
    nfrag = 0
    for f in ch.frames():
        fr = _lib.SondeFrame.from_buffer_copy(f.tobytes())
        nfrag += L.sonde_parser_feed(h, C.byref(fr), o, 8)
    L.sonde_parser_destroy(h)
    assert len(cbs) == nfrag >= 12
    assert re.search(r"serial=(\S+)", cbs[-1]).group(1) == "S0000000"
    assert abs(float(re.search(r"lat=(\S+)", cbs[-1]).group(1)) - 47.0) < 0.01


@pytest.mark.parametrize("stype,rate,chunk", [(0, 10000, 1000), (3, 50000, 4096), (1, 15000, 777)])
def test_b3_iq_stream_decoder_vfo_rate_cpp(tmp_path, oracle, stype, rate, chunk):
    """B3 at the reference's own VFO rate (sampleRate = supportedTypes[i].bandwidth, /root/reference/src/main.hpp:44-52,
    main.cpp:55-60): IqStreamDecoder runs the VFO front-end (FM + rational resampler on the GPU) in front of the decoder.
    Callbacks == fragments of the oracle chain or_vfo -> or_channel(real) through the stateful parser."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "iq_stream_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "iq_stream_test.cpp"), "-o", exe,
                           "-L", libdir, "-l:libsonde_mi355.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    up, down = {10000: (24, 5), 15000: (16, 5), 50000: (24, 25)}[rate]
    n_in = 6144 * 40 * down // up
    sb = synth.make_batch(stype, 1, n_in, seed=91 + stype, ebn0_db=24.0, fs=float(rate), cfo_max_hz=300.0)
    x = sb.iq.numpy()[0]
    path = str(tmp_path / "iq.bin")
    x.tofile(path)
    out = subprocess.check_output([exe, path, str(stype), str(chunk), str(rate)], text=True)
    assert "ERROR" not in out, out[-1500:]
    cbs = [l for l in out.splitlines() if l.startswith("CB ")]
    # the adaptor decodes whole input granules only (3 tiles of output); feed the oracle the same prefix
    gran = 6144 * down // up
    n_used = (n_in // gran) * gran
    vo, ch = oracle.Vfo(rate), oracle.Channel(stype, 0)
    ch.feed(vo.process(x[:n_used]), is_iq=False)
    L = _lib.load()
    h = L.sonde_parser_create(stype)
    o = (_lib.SondeData * 8)()
    nfrag = 0
    for f in ch.frames():
        fr = _lib.SondeFrame.from_buffer_copy(f.tobytes())
        nfrag += L.sonde_parser_feed(h, C.byref(fr), o, 8)
    L.sonde_parser_destroy(h)
    assert len(cbs) == nfrag >= 3, (len(cbs), nfrag)



def test_b3_type_switch_on_one_decoder_cpp(tmp_path, oracle):
    """The reference's onTypeSelected (/root/reference/src/main.cpp:366-405): stop the active decoder, new VFO bandwidth,
    resampler.setInSamplerate(bw), start another decoder.  Here: ONE sonde::IqStreamDecoder, RS41 at 10 kS/s, then re-initialised
    for M10 at 50 kS/s: the callbacks of each stream are the oracle chain's fragments for that type; the aggregate starts
    empty after the switch (lastData.init(), main.cpp:376)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "iq_stream_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "iq_stream_test.cpp"), "-o", exe,
                           "-L", libdir, "-l:libsonde_mi355.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    paths, want = [], []
    L = _lib.load()
    for stype, rate in ((0, 10000), (3, 50000)):
        up, down = {10000: (24, 5), 50000: (24, 25)}[rate]
        n_in = 6144 * 40 * down // up
        sb = synth.make_batch(stype, 1, n_in, seed=191 + stype, ebn0_db=24.0, fs=float(rate), cfo_max_hz=300.0)
        x = sb.iq.numpy()[0]
        path = str(tmp_path / f"iq{stype}.bin")
        x.tofile(path)
        paths.append((path, stype, rate))
        gran = 6144 * down // up
        vo, ch = oracle.Vfo(rate), oracle.Channel(stype, 0)
        ch.feed(vo.process(x[:(n_in // gran) * gran]), is_iq=False)
        h = L.sonde_parser_create(stype)
        o = (_lib.SondeData * 8)()
        nfrag = 0
        for f in ch.frames():
            fr = _lib.SondeFrame.from_buffer_copy(f.tobytes())
            nfrag += L.sonde_parser_feed(h, C.byref(fr), o, 8)
        L.sonde_parser_destroy(h)
        want.append(nfrag)
    (pa, ta, ra), (pb, tb, rb) = paths
    out = subprocess.check_output([exe, pa, str(ta), "1000", str(ra), pb, str(tb), str(rb)], text=True)
    assert "ERROR" not in out, out[-1500:]
    first, second = out.split("SWITCH")
    cbs = [[l for l in part.splitlines() if l.startswith("CB ")] for part in (first, second)]
    assert [len(c) for c in cbs] == want and min(want) >= 3, ([len(c) for c in cbs], want)
    assert "serial=S" in cbs[0][-1]                      # RS41 serial numbers of the generator
    assert "serial=S" not in cbs[1][0]                   # nothing of the RS41 aggregate survives the switch
