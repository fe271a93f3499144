"""-m gpu: examples/sonde_file_decoder.cpp -- the reference module's whole data flow (IQ stream -> decoder -> sondeDataHandler
-> GPX + CSV, /root/reference/src/main.cpp:54-72,320-331) on a synthetic RS41 flight, at 48 kS/s and at the reference's own
VFO rate.  The files must hold the flight the generator modelled."""
import os
import re
import subprocess

import numpy as np
import pytest

from sdrpp_radiosonde_amd import _lib, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("rate", [48000, 10000])
def test_iq_file_to_gpx_and_csv(tmp_path, rate):
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "sonde_file_decoder")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "sonde_file_decoder.cpp"), "-o", exe,
                           "-L", libdir, "-l:libsonde_mi355.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    n = int(2048 * 160 * rate / 48000) // 1280 * 1280
    sb = synth.make_rs41_batch(1, n, seed=5, ebn0_db=25.0, fs=float(rate), cfo_max_hz=300.0)
    iqf, gpx, csv = str(tmp_path / "iq.cf32"), str(tmp_path / "out.gpx"), str(tmp_path / "out.csv")
    sb.iq.numpy()[0].tofile(iqf)
    out = subprocess.check_output([exe, iqf, "0", str(rate), gpx, csv], text=True)
    m = re.search(r"(\d+) samples, (\d+) callbacks; last: serial=(\S+) seq=(\d+) lat=(\S+) lon=(\S+) alt=(\S+)", out)
    assert m, out
    assert int(m.group(1)) == n and int(m.group(2)) >= 10 and m.group(3) == "S0000000"
    g = open(gpx).read()
    assert g.startswith("<?xml") and g.endswith("</trkseg>\n</trk>\n</gpx>\n") and "<name>S0000000</name>" in g
    pts = re.findall(r'<trkpt lat="([-\d.]+)" lon="([-\d.]+)">', g)
    assert len(pts) >= 3
    lat, lon = np.array(pts, dtype=np.float64).T
    assert np.all(np.abs(lat - 47.0) < 0.05) and np.all(np.abs(lon - lon[0]) < 0.05)
    rows = open(csv).read().strip().splitlines()
    assert len(rows) >= 3 and rows[0].lower().startswith("epoch") and all(r.count(",") == rows[0].count(",") for r in rows[1:])
