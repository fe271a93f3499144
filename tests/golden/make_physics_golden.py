#!/usr/bin/env python3
"""Regenerates tests/golden/physics_golden.json from the REFERENCE's own compiled code: tests/cpp/ref_boundary_test.cpp includes
/root/reference/src/decode/decoder.hpp unmodified, its `physics` mode calls the header's static dewpt() and
altitude_to_pressure() (decoder.hpp:132-174) and prints inputs and outputs as hex floats.  The fixture holds inputs and
expected outputs only; it pins sonde_dewpt / sonde_altitude_to_pressure to the reference on boxes
that do not have /root/reference (the GPU box).  Needs /root/reference; run from the repo root."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from test_ref_boundary import build_ref_boundary  # noqa: E402

out = subprocess.check_output([build_ref_boundary(), "physics"], text=True)
alt, dew = [], []
for line in out.splitlines():
    w = line.split()
    if w[0] == "ALT":
        alt.append([w[1], w[2]])
    elif w[0] == "DEW":
        dew.append([w[1], w[2], w[3]])
path = os.path.join(HERE, "physics_golden.json")
json.dump({"source": "/root/reference/src/decode/decoder.hpp:132-174 compiled by tests/cpp/ref_boundary_test.cpp (mode `physics`)",
           "altitude_to_pressure": alt, "dewpt": dew}, open(path, "w"), indent=0)
print(len(alt), "altitudes,", len(dew), "dew points ->", path)
