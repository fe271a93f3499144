"""CPU: the product's GPX / CSV sinks (include/sonde_sinks.hpp) against the REFERENCE's own classes.
Pinned parity: the oracle here is the real reference code (/root/reference/src/gpx.cpp, ptu.cpp compiled
into oracle/_ref/libref_sinks.so where /root/reference exists) and a committed fixture of its output
(tests/golden/sinks_golden.json) for boxes without it.  Byte-identical after every operation."""
import json
import os
import subprocess

import pytest

import sinks_script
from sdrpp_radiosonde_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sinks_equal_committed_reference_output():
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "sinks_golden.json")))
    L = _lib.load()
    assert sinks_script.run_gpx(L, "product") == gold["gpx"]
    assert sinks_script.run_ptu(L, "product") == gold["ptu"]
    assert gold["gpx"][-1].count("<trkpt") == 5 and gold["gpx"][-1].endswith("</trkseg>\n</trk>\n</gpx>\n")


@pytest.mark.gpu
def test_sinks_equal_committed_reference_output_on_the_gpu_box():
    """The same check inside the driver's `-m gpu` run: the one row pinned to the reference's own compiled code (the fixture
    is the output of /root/reference/src/gpx.cpp + ptu.cpp, tests/make_sinks_golden.py) is then part of GPUTEST_rNN."""
    test_sinks_equal_committed_reference_output()


@pytest.mark.skipif(not os.path.exists("/root/reference/src/gpx.cpp"), reason="needs the mounted reference")
def test_sinks_equal_live_reference():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    R, L = sinks_script.ref_lib(), _lib.load()
    ref_g, got_g = sinks_script.run_gpx(R, "ref"), sinks_script.run_gpx(L, "product")
    assert len(ref_g) == len(got_g)
    for i, (a, b) in enumerate(zip(ref_g, got_g)):
        assert a == b, f"GPX differs after operation {i}"
    assert sinks_script.run_ptu(R, "ref") == sinks_script.run_ptu(L, "product")
