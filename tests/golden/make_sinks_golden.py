#!/usr/bin/env python3
"""Regenerates tests/golden/sinks_golden.json by driving the REFERENCE's own GPXWriter / PTUWriter
(oracle/_ref/libref_sinks.so, built by `make -C oracle ref` from /root/reference/src/gpx.cpp and
ptu.cpp) with the script in tests/sinks_script.py.  The fixture holds data only: the operations and
the file contents the reference produced after each one."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import sinks_script  # noqa: E402

out = {"gpx": sinks_script.run_gpx(sinks_script.ref_lib(), "ref"), "ptu": sinks_script.run_ptu(sinks_script.ref_lib(), "ref"),
       "gpx_ops": sinks_script.GPX_OPS, "ptu_ops": sinks_script.PTU_OPS}
json.dump(out, open(os.path.join(HERE, "sinks_golden.json"), "w"), indent=0)
print("checkpoints", len(out["gpx"]), len(out["ptu"]), "final gpx bytes", len(out["gpx"][-1]))
