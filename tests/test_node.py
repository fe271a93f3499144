"""The node-level host (include/sonde_node.h, csrc/node.cpp in libsonde_rccl.so): one process, the GPUs of one node, one
decoder batch per GPU, RCCL scatter of the ingest device's IQ straight into strided rows (north_star).
CPU: the library exports every declared symbol and refuses bad configurations.  GPU: a node of ONE device decodes the same frames
and fragments as a plain SondeBatch (channel numbers node-wide); a node of TWO devices against the oracle (skips on a one-GPU box:
the N-GPU run is the driver's)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE = 2048


def test_node_abi_exports():
    from sdrpp_radiosonde_amd import node
    L = node.lib()
    hdr = open(os.path.join(ROOT, "include", "sonde_node.h")).read()
    declared = set(re.findall(r"\b(sonde_node_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(node.NODE_SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    cfg = node.SondeNodeConfig()
    cfg.n_devices, cfg.ingest, cfg.n_channels, cfg.max_samples = 2, 5, 8, TILE          # ingest not one of the devices
    h = C.c_void_p()
    assert L.sonde_node_create(C.byref(cfg), C.byref(h)) != 0 and b"ingest" in L.sonde_node_last_error()
    cfg.n_devices, cfg.ingest, cfg.n_channels = 4, 0, 3                                    # fewer channels than devices
    assert L.sonde_node_create(C.byref(cfg), C.byref(h)) != 0 and b"fewer channels" in L.sonde_node_last_error()


@pytest.mark.gpu
def test_node_takes_16_bit_iq():
    """SONDE_INPUT_IQ16 through the node host (4-byte elements in the ingest copy / scatter): frames of the float path on the same integers."""
    import torch
    from sdrpp_radiosonde_amd import _lib, synth
    from sdrpp_radiosonde_amd.batch import SondeBatch
    from sdrpp_radiosonde_amd.node import SondeNode
    C_, n = 21, 24 * TILE
    sb = synth.make_rs41_batch(C_, n, seed=62, ebn0_db=14.0, device="cuda:0")
    x16 = torch.clamp(torch.round(sb.iq * 8192.0), -32768, 32767).to(torch.int16)
    nd = SondeNode(C_, n, devices=(0,), input_kind=_lib.INPUT_IQ16)
    nd.submit(x16)
    a = nd.frames()
    nd.close()
    bt = SondeBatch(C_, n)
    bt.submit(x16.to(torch.float32))
    b = bt.frames()
    bt.close()
    assert len(a) >= C_ // 2 and a.tobytes() == b.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("strided", [False, True])
def test_node_of_one_device_equals_a_batch(strided):
    import torch
    from sdrpp_radiosonde_amd import synth
    from sdrpp_radiosonde_amd.batch import SondeBatch, strided_rows
    from sdrpp_radiosonde_amd.node import SondeNode
    C_, n = 37, 24 * TILE
    sb = synth.make_rs41_batch(C_, 2 * n, seed=61, ebn0_db=14.0, device="cuda:0")
    nd = SondeNode(C_, n, devices=(0,))
    bt = SondeBatch(C_, n)
    for k in range(2):
        blk = sb.iq[:, k * n: (k + 1) * n].contiguous()
        if strided:
            blk = strided_rows(blk)                      # ingest rows already on the recommended stride: the one-run path
        nd.submit(blk)
        nd.scatter_done()
        bt.submit(blk)
        a, b = nd.frames(), bt.frames()
        assert len(a) >= C_ // 2 and a.tobytes() == b.tobytes(), k
        fa, fb = nd.poll(), bt.poll()
        assert len(fa) == len(fb) > 0 and all(x[0] == y[0] and bytes(x[1]) == bytes(y[1]) for x, y in zip(fa, fb))
    st = nd.scatter_stats()
    assert st["sends"] == 0 and st["bytes_from_ingest"] == 0 and st["ms"] > 0.0      # one device: a device copy, nothing leaves it
    nd.close()


@pytest.mark.gpu
def test_node_of_two_devices_equals_oracle(oracle):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU run)")
    from sdrpp_radiosonde_amd import synth
    from sdrpp_radiosonde_amd.node import SondeNode
    C_, n = 75, 24 * TILE                                  # unequal shards: 38 + 37
    sb = synth.make_rs41_batch(C_, n, seed=62, ebn0_db=14.0, device="cuda:1")
    nd = SondeNode(C_, n, devices=(0, 1), ingest=1)
    assert nd.range(0) == (0, 38) and nd.range(1) == (38, 75)
    nd.submit(sb.iq)
    got = nd.frames()
    ref = oracle.batch_run(0, sb.iq.cpu().numpy(), nthreads=4)
    assert len(ref) >= C_ // 2 and got.tobytes() == ref.tobytes()
    st = nd.scatter_stats()
    assert st["sends"] == 1 and st["bytes_from_ingest"] == 38 * n * 8       # rows back to back: ONE send of exactly device 0's 38 rows
    nd.close()
