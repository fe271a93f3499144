"""The drop-in boundary, proven against the reference's OWN header.

/root/reference/src/decode/decoder.hpp and common.hpp are compiled unmodified with `-I include/compat` (this repo's
stand-ins for the absent sondedump headers, decoder.hpp:6-14) and a test-only dsp/block.h skeleton; the seven
radiosonde::Decoder<...> instantiations of /root/reference/src/main.hpp:36-42 link against libsonde_mi355.so.
CPU: compile + link + construct.  GPU: one RS41 stream through the REFERENCE's run() loop (decoder.hpp:53-119) gives
the same callbacks as include/sonde_decoder.hpp.  The binary is built here (where /root/reference exists) into
tests/cpp/_build/ (git-ignored, travels to the GPU box); everything skips when neither source nor binary is there.
This is boundary evidence, not an oracle."""
import os
import subprocess

import numpy as np
import pytest

from sdrpp_radiosonde_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
EXE = os.path.join(BUILD, "ref_boundary_test")


def build_ref_boundary() -> str:
    """Compile tests/cpp/ref_boundary_test.cpp against the reference's decoder.hpp (in place) and libsonde_mi355.so."""
    _lib.load()
    os.makedirs(BUILD, exist_ok=True)
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "tests", "cpp", "refstub"), "-I", REF_SRC,
                           "-I", os.path.join(REF_SRC, "decode"), "-I", os.path.join(ROOT, "include", "compat"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "ref_boundary_test.cpp"),
                           "-o", EXE, "-L", libdir, "-l:libsonde_mi355.so", "-Wl,-rpath,$ORIGIN/../../../sdrpp_radiosonde_amd",
                           "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_SRC, "decode", "decoder.hpp")), reason="/root/reference absent")
def test_reference_decoder_hpp_compiles_and_links_against_this_abi():
    exe = build_ref_boundary()
    out = subprocess.check_output([exe, "link"], text=True)
    assert out.strip() == "LINK OK 7"
    # the seven triples resolve to this library, not to anything else
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", exe], text=True)
    for x in ("rs41", "dfm09", "ims100", "m10", "imet4", "c50", "mrzn1"):
        for fn in ("decoder_init", "decoder_deinit", "decode"):
            assert f"{x}_{fn}" in syms, (x, fn)


@pytest.mark.gpu
def test_reference_run_loop_on_gpu_decoder(oracle, tmp_path):
    if not os.path.exists(EXE):
        if not os.path.exists(os.path.join(REF_SRC, "decode", "decoder.hpp")):
            pytest.skip("neither /root/reference nor a prebuilt tests/cpp/_build/ref_boundary_test")
        build_ref_boundary()
    from sdrpp_radiosonde_amd import synth
    n = 2048 * 80
    sb = synth.make_rs41_batch(1, n, seed=12, ebn0_db=24.0)
    L = oracle.lib()
    d = np.zeros(n, dtype=np.float32)
    last = np.zeros(2, dtype=np.float32)
    L.or_discriminate(oracle.fptr(np.ascontiguousarray(sb.iq.numpy()[0]).reshape(-1)), n, oracle.fptr(d), oracle.fptr(last))
    path = str(tmp_path / "rs41.f32")
    d.tofile(path)
    out = subprocess.check_output([EXE, "pump", path, "4800"], text=True)
    assert "ERROR" not in out and out.strip().endswith("DONE"), out[-2000:]
    ref = [l[4:] for l in out.splitlines() if l.startswith("REF ")]
    own = [l[4:] for l in out.splitlines() if l.startswith("OWN ")]
    assert len(ref) >= 9 and ref == own
    seqs = [int(l.split("seq=")[1].split()[0]) for l in ref]
    assert max(seqs) - min(s for s in seqs if s) >= 2 and "serial=S0000000" in ref[-1]
