"""The node host's N > 1 logic, EXECUTED on one GPU.  csrc/node.cpp is built a second time, linked against
tests/cpp/fake_rccl.cpp instead of librccl (every ncclSend paired with its ncclRecv as a device copy, ordered on the streams the
operations were posted with), and sonde_node_create's test hook lets HIP device 0 stand for every device of the node.  What runs is
the product's own code: shard ranges, the three transfer shapes (straight from back-to-back ingest rows; packed chunk by chunk from
any other stride; row by row), the double-buffered row sets, the per-device batches, frame gather with node-wide channel numbers and
the scatter statistics.  What does NOT run: RCCL itself and xGMI (the driver's multi-GPU run).
The reference anchor for independent channels: /root/reference/src/main.cpp:18-24 (any number of module instances, no shared state)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
FAKE = os.path.join(BUILD, "libsonde_rccl_fake.so")
TILE = 2048


def build_fake_node_lib():
    """tests/cpp/_build/libsonde_rccl_fake.so = csrc/node.cpp + tests/cpp/fake_rccl.cpp (no librccl); rebuilt when a source is newer."""
    pkg = os.path.join(ROOT, "sdrpp_radiosonde_amd")
    srcs = [os.path.join(pkg, "csrc", "node.cpp"), os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp")]
    deps = srcs + [os.path.join(ROOT, "include", "sonde_node.h"), os.path.join(ROOT, "include", "sonde_abi.h")]
    if os.path.exists(FAKE) and all(os.path.getmtime(FAKE) >= os.path.getmtime(d) for d in deps):
        return FAKE
    os.makedirs(BUILD, exist_ok=True)
    from sdrpp_radiosonde_amd import _lib
    _lib.load()                                  # (builds libsonde_mi355.so if it is missing)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", FAKE] + srcs +
                          ["-L" + pkg, "-lsonde_mi355", "-Wl,-rpath," + pkg])
    return FAKE


def test_fake_node_library_builds_and_exports():
    from sdrpp_radiosonde_amd import node
    L = node.lib(build_fake_node_lib())
    for s in node.NODE_SYMBOLS + ["fake_rccl_stats"]:
        assert hasattr(L, s), s
    # the range arithmetic the shards follow (pure host code)
    for n in (3, 7, 1024, 65536, 65537):
        for nd in (1, 2, 3, 8):
            spans = []
            for d in range(nd):
                f, c = C.c_uint32(), C.c_uint32()
                L.sonde_node_shard_range(n, nd, d, C.byref(f), C.byref(c))
                spans.append((f.value, f.value + c.value))
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _fake_stats(L):
    a, b, g = C.c_uint64(), C.c_uint64(), C.c_uint64()
    L.fake_rccl_stats(C.byref(a), C.byref(b), C.byref(g))
    return a.value, b.value, g.value


@pytest.mark.gpu
@pytest.mark.parametrize("nd,ingest,layout,mode", [(2, 0, "contiguous", 0), (2, 1, "reco", 0), (3, 0, "reco", 0), (3, 2, "odd", 0), (3, 1, "contiguous", 0),
                                                   (3, 0, "reco", 1), (5, 3, "reco", 0)])
def test_node_of_several_shards_on_one_gpu(oracle, nd, ingest, layout, mode):
    """nd shards with UNEQUAL channel counts, three consecutive submits (both row sets and the staging buffers are reused), every ingest
    layout: frames of all shards = the oracle's, node-wide channel numbers; the bytes that 'left the ingest device' are exactly the
    peers' rows; sends as the header promises."""
    import torch
    from sdrpp_radiosonde_amd import node, synth
    from sdrpp_radiosonde_amd.batch import row_stride, strided_rows
    Cn, tiles, NS = 16 * nd + 1, 12, 3                  # 16 * nd + 1: unequal shards (the first device takes the extra channel)
    n = tiles * TILE
    sb = synth.make_rs41_batch(Cn, NS * n, seed=70 + nd, ebn0_db=14.0, device="cuda:0")
    nd_obj = node.SondeNode(Cn, n, devices=(0,) * nd, ingest=ingest, scatter_mode=mode | node.TEST_SHARED_DEVICES, lib_path=build_fake_node_lib())
    L = nd_obj.L
    spans = [nd_obj.range(d) for d in range(nd)]
    assert spans[0][0] == 0 and spans[-1][1] == Cn and len({b - a for a, b in spans}) == 2
    peers_rows = Cn - (spans[ingest][1] - spans[ingest][0])
    _fake_stats(L)
    parts, frags = [], []
    for k in range(NS):
        blk = sb.iq[:, k * n: (k + 1) * n].contiguous()
        if layout == "reco":
            blk = strided_rows(blk)
        elif layout == "odd":
            blk = strided_rows(blk, n + 6144)
        nd_obj.submit(blk)
        parts.append(nd_obj.frames().copy())
        frags += nd_obj.poll()
        st = nd_obj.scatter_stats()
        sends, byts, groups = _fake_stats(L)
        assert st["bytes_from_ingest"] == peers_rows * n * 8 == byts, (k, st, byts)      # exactly the rows' bytes: no padding travels
        assert st["sends"] == sends
        if mode == 1:
            assert sends == peers_rows
        elif layout == "contiguous":
            assert sends == nd - 1 and groups == 1
        else:
            assert nd - 1 < sends <= 4 * (nd - 1) and groups == 4                         # packed: four chunks
    got = np.concatenate(parts)
    got = got[np.lexsort((got["bitpos"], got["channel"]))]
    ref = oracle.batch_run(0, sb.iq.cpu().numpy(), nthreads=os.cpu_count() or 4)
    assert len(ref) >= Cn and got.tobytes() == ref.tobytes()
    assert len(frags) > 0 and max(c for c, _ in frags) >= spans[-1][0]                    # fragments carry node-wide channel numbers
    nd_obj.close()


@pytest.mark.gpu
def test_node_of_several_shards_mixed_types_and_16_bit_rows(oracle):
    """Types differ from shard to shard (each device's batch gets ITS slice of the type list) and the rows are 16-bit integers
    (4-byte elements through packing and transfer)."""
    import torch
    from sdrpp_radiosonde_amd import _lib, node, synth
    from sdrpp_radiosonde_amd.batch import strided_rows
    nd, Cn, tiles = 3, 50, 16
    n = tiles * TILE
    order = (0, 1, 3)
    types = np.array([order[(c // 7) % 3] for c in range(Cn)], dtype=np.uint8)
    iq = torch.empty((Cn, 2 * n, 2), dtype=torch.float32, device="cuda:0")
    for t in order:
        idx = np.nonzero(types == t)[0]
        iq[torch.from_numpy(idx).to("cuda:0")] = synth.make_batch(int(t), len(idx), 2 * n, seed=5 + t, ebn0_db=16.0, device="cuda:0").iq
    q = torch.clamp(torch.round(iq * 8192.0), -32768, 32767).to(torch.int16)
    host = q.to(torch.float32).cpu().numpy()
    refs = []
    for t in order:
        idx = np.nonzero(types == t)[0]
        r = oracle.batch_run(int(t), host[idx], nthreads=os.cpu_count() or 4)
        r["channel"] = idx[r["channel"]]
        refs.append(r)
    ref = np.concatenate(refs)
    ref = ref[np.lexsort((ref["bitpos"], ref["channel"]))]
    nd_obj = node.SondeNode(Cn, n, devices=(0,) * nd, ingest=1, types=types, input_kind=_lib.INPUT_IQ16, scatter_mode=node.TEST_SHARED_DEVICES,
                            lib_path=build_fake_node_lib())
    parts = []
    for k in range(2):
        nd_obj.submit(strided_rows(q[:, k * n: (k + 1) * n].contiguous()))
        parts.append(nd_obj.frames().copy())
        assert nd_obj.scatter_stats()["bytes_from_ingest"] == (Cn - (nd_obj.range(1)[1] - nd_obj.range(1)[0])) * n * 4
    got = np.concatenate(parts)
    got = got[np.lexsort((got["bitpos"], got["channel"]))]
    nd_obj.close()
    assert len(ref) >= Cn and got.tobytes() == ref.tobytes()
