"""-m gpu: every number of the driver's bench line has a parity test at ITS shape and ITS flags.

bench.py entry                      test here
other_configs.mix4096 / _late_join  test_mix4096_five_blocks_back_to_back[0 / 32]     (default flags: joined at every submit; [32]: SONDE_FLAG_LATE_JOIN; [4]: SONDE_FLAG_PIPELINE)
low_snr                             test_headline_shape_at_9_db                        (1024 x 96 tiles, Eb/N0 9 dB)
other_configs.rt1250, ch1280x96     test_part_filled_last_generation[1250-24-0 / 1280-96-0] and [..-32] (not a multiple of one residency; default and late-joined)
other_configs.config5_full_1gpu     tests/test_gpu_full_size.py::test_config5_all_65536_channels_on_one_gpu            (65 536 channels x 24 tiles through sonde_node, float and 16-bit rows)
other_configs.config1_cpu_plumbing  tests/test_gpu_full_size.py::test_config1_single_channel_cpu_plumbing_equals_the_gpu_decoder
other_configs.wideband8x4           tests/test_channelizer.py::test_fused_channelizer_frames_equal_oracle[4-8]
other_configs.rt1250_host_e2e       test_host_path_at_the_target_shape                 (1250 channels x 1 s from host memory to SondeData fragments)
other_configs.cs16_1024x96, cs16_8192x24   test_16_bit_rows_at_the_bench_shapes[1024-96 / 8192-24]  (SONDE_INPUT_IQ16 rows on the recommended stride)
other_configs.rt1250_host_e2e_cs16  test_16_bit_rows_at_the_bench_shapes[1250-24] (+ host staging: tests/test_gpu_iq16.py::test_iq16_from_host_memory)
other_configs.cs8_1024x96, rt1250_host_e2e_cs8   test_16_bit_rows_at_the_bench_shapes[1024-96-8 / 1250-24-8]  (SONDE_INPUT_IQ8)
other_configs.wideband8_cs16        tests/test_channelizer.py::test_channelizer_takes_16_bit_wideband_blocks (two streams; the 8-stream grid: [4-8] above)

What a frame stream must equal: /root/reference/src/decode/decoder.hpp:61 (one X_decode call sequence per channel); the oracle
(oracle/, the CPU restatement) stands for it."""
import os

import numpy as np
import pytest
import torch

from sdrpp_radiosonde_amd import synth
from sdrpp_radiosonde_amd._lib import FLAG_JOIN, FLAG_LATE_JOIN, FLAG_PIPELINE
from sdrpp_radiosonde_amd.batch import SondeBatch, strided_rows

pytestmark = pytest.mark.gpu
TILE = 2048
CORES = os.cpu_count() or 4


def _key(a):
    return a[np.lexsort((a["bitpos"], a["channel"]))]


def _join_of(flags):
    return 2 if flags & FLAG_PIPELINE else (1 if flags & FLAG_LATE_JOIN else 0)


@pytest.mark.parametrize("flags", [0, FLAG_LATE_JOIN, FLAG_PIPELINE])
def test_mix4096_five_blocks_back_to_back(oracle, flags):
    """BASELINE configs[2] as bench.py measures it: RS41 / M10 / DFM09 by channel % 3, 4096 channels x 24 tiles per submit,
    five consecutive blocks, at the DEFAULT flags (round 6: ordinary stream semantics, every submit joined into the caller's stream),
    with SONDE_FLAG_LATE_JOIN (every type's kernels keep their own stream, the caller's stream is joined
    one submit late) and with SONDE_FLAG_PIPELINE (never joined), rows on
    the recommended stride.  (a) two submits in flight, frames per ticket: all frames byte for byte the oracle's over the
    whole 120-tile signal; (b) all five submits queued with NO host interaction in between (the bench's loop): the frames of
    the last two tickets and every channel's loop state and newest bits equal run (a)'s."""
    C, tiles, NB = 4096, 24, 5
    n = tiles * TILE
    order = (0, 3, 1)
    types = np.array([order[c % 3] for c in range(C)], dtype=np.uint8)
    iq = torch.empty((C, NB * n, 2), dtype=torch.float32, device="cuda:0")
    refs = []
    for t in order:
        idx = np.nonzero(types == t)[0]
        sb = synth.make_batch(int(t), len(idx), NB * n, seed=700 + int(t), ebn0_db=16.0, device="cuda:0")
        iq[torch.from_numpy(idx).to("cuda:0")] = sb.iq
        r = oracle.batch_run(int(t), sb.iq.cpu().numpy(), nthreads=CORES)
        r["channel"] = idx[r["channel"]]
        refs.append(r)
        del sb
    ref = _key(np.concatenate(refs))
    blocks = [strided_rows(iq[:, k * n: (k + 1) * n].contiguous()) for k in range(NB)]
    del iq
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream

    # (a) tickets, two submits in flight
    b = SondeBatch(C, n, types=types, flags=flags)
    info = b.launch_info()
    # (default flags: ONE launch over both classes, sd_demod_mixed_kernel; the opt-in modes: one unit per sonde type on its own stream)
    assert info["join"] == _join_of(flags) and (info["units"] == 3 if flags else info["units"] == 1)
    b.ticket()
    per_ticket = []
    for k in range(NB):
        b.submit(blocks[k], st)
        if k >= 1:
            per_ticket.append(b.frames_of(k))
    per_ticket.append(b.frames_of(NB))
    got = _key(np.concatenate(per_ticket))
    assert len(ref) >= 3 * C
    assert got.tobytes() == ref.tobytes()
    probe = list(range(0, C, 61))
    state_a = [b.state(c) for c in probe]
    nb_a = [b.nbits(c) for c in probe]
    bits_a = [b.read_bits(c, nb_a[i] - 2000, 2000) for i, c in enumerate(probe)]
    b.close()

    # (b) the bench's loop: five submits, nothing in between
    b = SondeBatch(C, n, types=types, flags=flags)
    for k in range(NB):
        b.submit(blocks[k], st)
    f5 = b.frames_of(NB)
    f4 = b.frames_of(NB - 1)
    assert f5.tobytes() == per_ticket[NB - 1].tobytes() and f4.tobytes() == per_ticket[NB - 2].tobytes()
    for i, c in enumerate(probe):
        assert b.state(c) == state_a[i] and b.nbits(c) == nb_a[i], c
        assert np.array_equal(b.read_bits(c, nb_a[i] - 2000, 2000), bits_a[i]), c
    b.close()


def test_headline_shape_at_9_db(oracle):
    """bench.py's low_snr entry: the headline shape (1024 RS41 channels x 96 tiles, one residency, in-loop clean-frame
    decoder active) at Eb/N0 9 dB, where most frames need the Reed-Solomon corrector's general path in the epilogue and
    some are beyond it: every record -- corrected, rejected (nerr -1) -- equals the oracle's."""
    C, n = 1024, 96 * TILE
    sb = synth.make_rs41_batch(C, n, seed=909, ebn0_db=9.0, device="cuda:0")
    b = SondeBatch(C, n)
    b.submit(strided_rows(sb.iq))
    got = b.frames()
    ref = oracle.batch_run(0, sb.iq.cpu().numpy(), nthreads=CORES)
    assert len(ref) >= 4 * C
    assert got.tobytes() == ref.tobytes()
    touched = (got["nerr"] != 0).any(axis=1).sum()                # corrected or given up (-1): the corrector's general path ran
    assert touched >= 0.9 * len(got), (touched, len(got))
    good = got[(got["nerr"] >= 0).all(axis=1)]
    assert (got["nerr"] > 0).any(axis=1).sum() >= 0.2 * len(got) and len(good) >= 100          # (at 9 dB most frames lose one codeword)
    for f in good[:: max(1, len(good) // 400)]:                   # a sample: FEC-clean frames are transmitted frames
        assert any(np.array_equal(tx[8:], f["data"][8: f["len"]]) for _, tx in sb.frames[f["channel"]])


@pytest.mark.parametrize("C,tiles,flags", [(1250, 24, 0), (1280, 96, 0), (1250, 24, FLAG_LATE_JOIN), (1280, 96, FLAG_LATE_JOIN), (1250, 24, FLAG_PIPELINE),
                                           (1250, 48, FLAG_JOIN), (1537, 24, 0), (1537, 24, FLAG_LATE_JOIN)])
def test_part_filled_last_generation(oracle, C, tiles, flags):
    """Channel counts that are not a multiple of one residency (4 workgroups x 256 CUs): 1250 x 24 tiles = bench.py's rt1250 (the
    north_star's per-GPU share of 10^4 channels, one second per submit), 1280 x 96 = ch1280x96, both as the bench runs them (round 6):
    at the DEFAULT flags (ordinary stream semantics) and with SONDE_FLAG_LATE_JOIN (two launch units on their own streams joined into
    the caller's stream one submit late), submits queued back to back (frames per ticket, two in flight); the same never joined
    (SONDE_FLAG_PIPELINE); SONDE_FLAG_JOIN (round 5's opt-out) is the default spelled out; 1537 = 1.5 residencies + 1.  Four
    consecutive submits; frames, bit counts and loop state against the oracle."""
    n, NS = tiles * TILE, 4
    sb = synth.make_rs41_batch(C, NS * n, seed=1250 + C, ebn0_db=13.0, device="cuda:0")
    blocks = [strided_rows(sb.iq[:, k * n: (k + 1) * n].contiguous()) for k in range(NS)]
    b = SondeBatch(C, n, flags=flags)
    info = b.launch_info()
    assert info["join"] == _join_of(flags) and (info["units"] == 2 if _join_of(flags) else info["units"] >= 1)
    st = torch.cuda.current_stream().cuda_stream
    parts = []
    if not flags & FLAG_JOIN:
        b.ticket()
        for k in range(NS):
            b.submit(blocks[k], st)
            if k >= 1:
                parts.append(b.frames_of(k))
        parts.append(b.frames_of(NS))
    else:
        for k in range(NS):
            b.submit(blocks[k], st)
            parts.append(b.frames())
    got = _key(np.concatenate(parts))
    host = sb.iq.cpu().numpy()
    ref = oracle.batch_run(0, host, nthreads=CORES)
    assert len(ref) >= C * (NS * n // 48000 - 1)
    assert got.tobytes() == ref.tobytes()
    for c in (0, 1023, 1024, 1025, C - 1):
        ch = oracle.Channel(0, c)
        ch.feed(host[c])
        rs, gs = ch.state(), b.state(c)
        assert (gs["t_next"], gs["period"], gs["bias"], gs["amp"]) == (rs["t_next"], rs["period"], rs["bias"], rs["amp"]), c
        assert b.nbits(c) == len(ch.bits())


def test_completion_contract_depends_on_the_flags_only():
    """include/sonde_abi.h, "how a submit completes on the caller's stream" (ADVICE r5): flags 0 = ordinary stream
    semantics WHATEVER the channel count, the mix of types or the device's CU count; the lagging join and the never-joined mode are
    opt-in flags.  sonde_batch_launch_info reports the mode the flags ask for."""
    n = 24 * TILE
    mixed = np.array([(0, 3, 1)[c % 3] for c in range(96)], dtype=np.uint8)
    for flags in (0, FLAG_JOIN, FLAG_LATE_JOIN, FLAG_PIPELINE):
        for C, types in ((4, None), (1024, None), (1250, None), (2048, None), (96, mixed)):
            b = SondeBatch(C, n, types=types, flags=flags)
            assert b.launch_info()["join"] == _join_of(flags), (flags, C)
            b.close()


@pytest.mark.parametrize("flags", [0, FLAG_LATE_JOIN, FLAG_PIPELINE])
@pytest.mark.parametrize("shape", ["rs41_1250", "mixed_1536"])
def test_sample_buffer_may_be_rewritten_on_the_callers_stream(oracle, flags, shape):
    """VERDICT r5 item 3's acceptance test.  ONE sample buffer, refilled on the caller's stream right behind every submit (an
    asynchronous device copy -- what a streaming host's DMA does; /root/reference/src/decode/decoder.hpp:59-117 flushes the stream
    buffer right after X_decode returns).  Default flags: nothing else to do (stream order).  SONDE_FLAG_LATE_JOIN /
    SONDE_FLAG_PIPELINE: sonde_batch_wait_input on that stream before the copy.  Shapes that are cut into launch units (1250 RS41
    channels: a part-filled generation; a mixed batch: one unit per type).  Frames of all submits: the oracle's, byte for byte."""
    tiles, NS = 12, 6
    n = tiles * TILE
    if shape == "rs41_1250":
        C, types, order = 1250, None, (0,)
    else:
        C, order = 1536, (0, 3, 1)
        types = np.array([order[c % 3] for c in range(C)], dtype=np.uint8)
    iq = torch.empty((C, NS * n, 2), dtype=torch.float32, device="cuda:0")
    refs = []
    for t in order:
        idx = np.arange(C) if types is None else np.nonzero(types == t)[0]
        sb = synth.make_batch(int(t), len(idx), NS * n, seed=3300 + int(t), ebn0_db=15.0, device="cuda:0")
        iq[torch.from_numpy(idx).to("cuda:0")] = sb.iq
        r = oracle.batch_run(int(t), sb.iq.cpu().numpy(), nthreads=CORES)
        r["channel"] = idx[r["channel"]]
        refs.append(r)
        del sb
    ref = _key(np.concatenate(refs))
    blocks = [iq[:, k * n: (k + 1) * n].contiguous() for k in range(NS)]
    del iq
    buf = strided_rows(blocks[0].clone())                 # THE buffer: every submit reads it, every copy overwrites it
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    b = SondeBatch(C, n, types=types, flags=flags)
    assert b.launch_info()["join"] == _join_of(flags)
    b.ticket()
    parts = []
    with torch.cuda.stream(s):
        for k in range(NS):
            b.submit(buf, s.cuda_stream)
            if flags:
                b.wait_input(s.cuda_stream)               # device-side; the host does not wait
            if k + 1 < NS:
                buf.copy_(blocks[k + 1], non_blocking=True)      # queued on s right behind the submit
            if k >= 1:
                parts.append(b.frames_of(k))
        parts.append(b.frames_of(NS))
    got = _key(np.concatenate(parts))
    b.close()
    assert len(ref) >= C and got.tobytes() == ref.tobytes()


def test_wait_input_releases_the_buffer_on_another_stream(oracle):
    """sonde_batch_wait_input with a stream that is NOT the submit's: a copy engine's stream may refill the buffer behind it (one
    plain launch, 1024 channels: the event is recorded on the submit's stream at the call)."""
    C, tiles, NS = 1024, 12, 4
    n = tiles * TILE
    sb = synth.make_rs41_batch(C, NS * n, seed=515, ebn0_db=15.0, device="cuda:0")
    ref = _key(oracle.batch_run(0, sb.iq.cpu().numpy(), nthreads=CORES))
    blocks = [sb.iq[:, k * n: (k + 1) * n].contiguous() for k in range(NS)]
    buf = strided_rows(blocks[0].clone())
    torch.cuda.synchronize()
    s, cp = torch.cuda.Stream(), torch.cuda.Stream()
    b = SondeBatch(C, n)
    parts = []
    for k in range(NS):
        s.wait_stream(cp)                                  # the submit reads what the copy stream wrote
        b.submit(buf, s.cuda_stream)
        b.wait_input(cp.cuda_stream)
        if k + 1 < NS:
            with torch.cuda.stream(cp):
                buf.copy_(blocks[k + 1], non_blocking=True)
        parts.append(b.frames().copy())
    got = _key(np.concatenate(parts))
    b.close()
    assert got.tobytes() == ref.tobytes()


def test_host_path_at_the_target_shape(oracle):
    """bench.py's rt1250_host_e2e: 1250 RS41 channels, one second at a time, HOST memory in (sonde_batch_submit_host: PCIe + staging
    into strided rows), SondeData fragments out (sonde_batch_poll).  Frames of every submit equal the oracle's; every FEC-clean frame
    yields its sequence fragment with the right channel and serial; nothing is lost between submits."""
    from sdrpp_radiosonde_amd import _lib
    C, n, NS = 1250, 24 * TILE, 3
    sb = synth.make_rs41_batch(C, NS * n, seed=4242, ebn0_db=15.0, device="cuda:0")
    host = sb.iq.cpu().numpy()
    b = SondeBatch(C, n)
    frames, frags = [], []
    for k in range(NS):
        b.submit_host(np.ascontiguousarray(host[:, k * n: (k + 1) * n]))
        frames.append(b.frames())
        frags += b.poll(cap=4096)
    got = _key(np.concatenate(frames))
    ref = oracle.batch_run(0, host, nthreads=CORES)
    assert len(ref) >= C and got.tobytes() == ref.tobytes()
    seqs = {}
    for c, d in frags:
        if d.fields & _lib.DATA_SEQ:
            assert d.serial == ("S%07d" % c).encode()
            seqs.setdefault(c, []).append(d.seq)
    clean = got[(got["nerr"] >= 0).all(axis=1)]
    for c in range(0, C, 53):
        want = [int(f["data"][59]) | (int(f["data"][60]) << 8) for f in clean if f["channel"] == c]
        assert seqs.get(c, []) == want, c
    assert sum(len(v) for v in seqs.values()) == len(clean)


@pytest.mark.parametrize("C,tiles,bits", [(1024, 96, 16), (8192, 24, 16), (1250, 24, 16), (1024, 96, 8), (1250, 24, 8)])
def test_16_bit_rows_at_the_bench_shapes(oracle, C, tiles, bits):
    """SONDE_INPUT_IQ16 at the shapes of other_configs.cs16_*: the bench's own quantisation (full scale 8192 per unit amplitude), rows on the
    recommended stride, two consecutive submits.  Every frame byte for byte the oracle's on the same integers as floats."""
    from sdrpp_radiosonde_amd import _lib
    n = tiles * TILE
    sb = synth.make_rs41_batch(C, 2 * n, seed=910 + tiles, ebn0_db=14.0, device="cuda:0")
    x16 = torch.clamp(torch.round(sb.iq * 8192.0), -32768, 32767).to(torch.int16) if bits == 16 else torch.clamp(torch.round(sb.iq * 16.0), -128, 127).to(torch.int8)
    del sb
    b = SondeBatch(C, n, input_kind=_lib.INPUT_IQ16 if bits == 16 else _lib.INPUT_IQ8)
    got = []
    for k in range(2):
        b.submit(strided_rows(x16[:, k * n: (k + 1) * n].contiguous()))
        got.append(b.frames().copy())
    b.close()
    got = _key(np.concatenate(got))
    ref = _key(oracle.batch_run(0, x16.to(torch.float32).cpu().numpy(), nthreads=CORES))
    assert len(ref) >= C * (2 * tiles) // 32 and got.tobytes() == ref.tobytes()
