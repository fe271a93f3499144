"""CPU: the C++ Decoder adaptor (include/sonde_decoder.hpp) against the semantics of
/root/reference/src/decode/decoder.hpp:53-119,132-174, and the product's dewpt / ISA functions against
the oracle's (which are KAT-pinned to the reference's values)."""
import os
import re
import subprocess

import numpy as np
import pytest

from sdrpp_radiosonde_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_physics_equal_oracle_bit_for_bit(oracle):
    L, P = oracle.lib(), _lib.load()
    rng = np.random.default_rng(0)
    for t, rh in zip(rng.uniform(-90, 40, 300), rng.uniform(0.5, 100, 300)):
        a, b = np.float32(P.sonde_dewpt(float(t), float(rh))), np.float32(L.or_dewpt(float(t), float(rh)))
        assert a.tobytes() == b.tobytes()
    for alt in list(rng.uniform(-500, 90000, 400)) + [0.0, 10999.9, 11000.0, 20000.0, 32000.0, 47000.0, 51000.0, 77000.0, 12000.0]:
        a, b = np.float32(P.sonde_altitude_to_pressure(float(alt))), np.float32(L.or_altitude_to_pressure(float(alt)))
        assert a.tobytes() == b.tobytes()
    assert abs(P.sonde_dewpt(-50.0, 30.0) - (-59.7688)) < 1e-4
    assert abs(P.sonde_altitude_to_pressure(12000.0) - 193.3049) < 1e-3


def test_decoder_adaptor_merge_semantics(tmp_path, oracle):
    exe = str(tmp_path / "adaptor_test")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adaptor_test.cpp"), "-o", exe,
                           "-L", libdir, "-l:libsonde_mi355.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([exe], text=True)
    assert "ERROR" not in out, out
    cbs = [l for l in out.splitlines() if l.startswith("CB ")]
    done = [l for l in out.splitlines() if l.startswith("DONE")][0]
    # 6 fragments with fields != 0 -> 6 callbacks (the fields == 0 one is merged silently); one more get() per buffer for PROCEED
    assert "fired=6 per_buffer=2,2,2" in done and "get_calls=10" in done
    L = oracle.lib()

    def field(line, name):
        return re.search(name + r"=(\S*)", line).group(1)

    # 1: seq+serial, nothing else yet; pressure <= 0 -> ISA at alt 0
    assert field(cbs[0], "seq") == "1234" and field(cbs[0], "serial") == "S1234567"
    assert float.fromhex(field(cbs[0], "pressure")) == np.float32(L.or_altitude_to_pressure(0.0))
    # 2: position merged; pressure stays (it was > 0 already: the fallback is sticky, decoder.hpp:108)
    assert field(cbs[1], "alt") == "12000.0" and field(cbs[1], "hdg") == "90.00"
    assert float.fromhex(field(cbs[1], "pressure")) == np.float32(L.or_altitude_to_pressure(0.0))
    # 3: PTU with pressure 0 -> dew point + ISA(12000 m); calibrated at 100 %
    assert float.fromhex(field(cbs[2], "dewpt")) == np.float32(L.or_dewpt(-50.0, 30.0))
    assert float.fromhex(field(cbs[2], "pressure")) == np.float32(L.or_altitude_to_pressure(12000.0))
    assert field(cbs[2], "cal") == "1"
    # 4: time + burst-kill; 5: ozone text; 6: second PTU overrides pressure and un-calibrates
    assert field(cbs[3], "time") == "1700000000" and field(cbs[3], "kill") == "3600"
    assert field(cbs[4], "aux") == "O3=3.14mPa"
    assert float.fromhex(field(cbs[5], "pressure")) == 900.0 and field(cbs[5], "cal") == "0" and field(cbs[5], "aux") == "O3=3.14mPa"
    assert field(cbs[5], "serial") == "S1234567"           # sticky across frames
