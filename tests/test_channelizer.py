"""Wideband front-end (BASELINE config 4): 10 MS/s -> 512-bin polyphase channelizer (20 kS/s per bin, one PHASE sample per
step) -> per-bin discriminator (wrapped phase difference) -> 12/5 resampler -> decoder.  CPU: the oracle's building blocks and an end-to-end decode.  GPU: the HIP
front-end against the oracle, bit-exact at both intermediate products and in the decoded frames."""
import ctypes as C

import numpy as np
import pytest
import torch

from sdrpp_radiosonde_amd import synth

BLOCK = 1_280_000          # wideband samples per block = 2560 steps = 6144 samples at 48 kS/s per bin
STEPS = 2560


def test_fft512_and_tables(oracle):
    L = oracle.lib()
    tw = np.zeros(512, dtype=np.float32)
    L.or_chan_twiddles(oracle.fptr(tw))
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(512) + 1j * rng.standard_normal(512)).astype(np.complex64)
    re, im = np.ascontiguousarray(x.real), np.ascontiguousarray(x.imag)
    L.or_fft512(oracle.fptr(re), oracle.fptr(im), oracle.fptr(tw))
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.max(np.abs((re + 1j * im) - ref)) < 2e-4
    h = np.zeros(8192, dtype=np.float32)
    L.or_chan_proto(oracle.fptr(h))
    assert abs(h.sum() - 1.0) < 1e-5 and np.allclose(h, h[::-1], atol=1e-9)
    g = np.zeros(192, dtype=np.float32)
    L.or_chan_resamp_taps(oracle.fptr(g))
    assert np.allclose(g.reshape(12, 16).sum(axis=1), 1.0, atol=1e-6)


def test_prototype_is_symmetric_bit_for_bit(oracle):
    """The filter-bank kernel keeps only the first half of the prototype in LDS and reads tap u >= 4096 at 8191 - u (round 5:
    three workgroups per CU); sonde_chan_create refuses a prototype that is not symmetric bit for bit.  The SPEC's is."""
    L = oracle.lib()
    h = np.zeros(16 * 512, dtype=np.float32)
    L.or_chan_proto(oracle.fptr(h))
    assert h.tobytes() == h[::-1].tobytes()


def test_channelizer_isolates_a_tone(oracle):
    """A tone 1.2 kHz above the centre of bin 77 comes out of bin 77 as a phase ramp of that slope at 20 kS/s; the resampled
    discriminator reads its frequency."""
    L = oracle.lib()
    k, df = 77, 1200.0
    n = np.arange(BLOCK, dtype=np.float64)
    ph = 2 * np.pi * (k * 10e6 / 512 + df) / 10e6 * n
    iq = np.stack([np.cos(ph), np.sin(ph)], axis=1).astype(np.float32)
    ch = L.or_chan_new()
    bins = np.zeros((512, STEPS), dtype=np.float32)
    out48 = np.zeros((512, STEPS * 12 // 5), dtype=np.float32)
    L.or_chan_block(ch, oracle.fptr(iq.reshape(-1)), STEPS, oracle.fptr(bins.reshape(-1)), oracle.fptr(out48.reshape(-1)))
    L.or_chan_free(ch)
    ph = bins[k, 100:].astype(np.float64)                   # quadrants
    dph = np.diff(ph)
    dph -= 4.0 * np.round(dph / 4.0)
    assert np.allclose(dph * 20000 / 4.0, df, atol=12.0)    # atan2q: |error| <= 4.5e-4 quadrant per phase (SPEC 3.1)
    # discriminator gain 2/pi at 20 kS/s: d = 2*pi*df/20000 * 2/pi
    assert np.allclose(out48[k, 500:], 4 * df / 20000, atol=2e-3)


def _oracle_decode_wideband(oracle, iq_np, bins_active, types=None, composite=False, odd=False):
    """composite=False: 48 kS/s rows (SPEC 3.5) fed to the channels' real-input path (what the product's unfused mode does);
    composite=True: the decimated rows of SPEC 3.5b (resampler + boxcar as one filter: the product's default, fused mode)."""
    L = oracle.lib()
    nblk = iq_np.shape[0] // BLOCK
    ch = L.or_chan_new_odd() if odd else L.or_chan_new()      # odd: the half-bin-shifted bank (SPEC 3.5c)
    dec = {k: oracle.Channel(int(types[k]) if types is not None else 0, k) for k in bins_active}
    n_out = STEPS * 12 // 5
    out48 = np.zeros((512, n_out), dtype=np.float32)
    decs = np.zeros(512, dtype=np.uint8)
    for k in bins_active:
        decs[k] = 4                                       # the 12 kS/s sondes (SPEC 3.0); M10 (50 kHz wide) does not fit a 19.5 kHz bin
    outdec = np.zeros((512, n_out // 2), dtype=np.float32)
    first = None
    for b in range(nblk):
        blk = np.ascontiguousarray(iq_np[b * BLOCK: (b + 1) * BLOCK]).reshape(-1)
        bins = np.zeros((512, STEPS), dtype=np.float32) if b == 0 else None
        L.or_chan_block2(ch, oracle.fptr(blk), STEPS, oracle.fptr(bins.reshape(-1)) if b == 0 else None, oracle.fptr(out48.reshape(-1)),
                         decs.ctypes.data, oracle.fptr(outdec.reshape(-1)))
        if b == 0:
            first = (bins, out48.copy(), outdec.copy())
        for k in bins_active:
            if composite:
                dec[k].feed_decimated(outdec[k, :n_out // int(decs[k])], int(decs[k]))
            else:
                dec[k].feed(out48[k], is_iq=False)
    L.or_chan_free(ch)
    return dec, first


def test_composite_rows_are_resample_then_average(oracle):
    """SPEC 3.5b against SPEC 3.5 + 3.0: the composite filter's decimated rows equal the boxcar average of the 48 kS/s rows up
    to rounding (both are sums of the same 64 / 32 products), the composite taps sum to one per row, and a scene decodes
    through either path to the same frames."""
    bins_active = [100, 333]
    iq, truth = synth.make_wideband_rs41(bins_active, 10 * BLOCK, seed=6, ebn0_db=32.0)
    types = np.zeros(512, dtype=np.uint8)
    types[333] = 1                                          # a DFM bin too (no DFM signal there: the arithmetic is what is compared)
    a, first = _oracle_decode_wideband(oracle, iq.numpy(), bins_active, types=types, composite=False)
    c, _ = _oracle_decode_wideband(oracle, iq.numpy(), bins_active, types=types, composite=True)
    _, out48, outdec = first
    n_out = out48.shape[1]
    box4 = out48[100].astype(np.float64).reshape(-1, 4).mean(axis=1)
    box4b = out48[333].astype(np.float64).reshape(-1, 4).mean(axis=1)
    assert np.abs(outdec[100, :n_out // 4] - box4).max() < 4e-6 and np.abs(outdec[333, :n_out // 4] - box4b).max() < 4e-6
    L = oracle.lib()
    g = np.zeros(192, dtype=np.float32)
    L.or_chan_resamp_taps(oracle.fptr(g))
    G = np.zeros(60, dtype=np.float32)
    L.or_chan_composite_taps(oracle.fptr(g), 4, oracle.fptr(G))
    assert np.abs(G.reshape(3, 20).astype(np.float64).sum(axis=1) - 1.0).max() < 1e-6
    assert (G.reshape(3, 20)[:, 17:] == 0).all() and (G.reshape(3, 20)[:, 16] != 0).any()
    fa, fc = a[100].frames(), c[100].frames()
    assert len(fa) >= 1 and np.array_equal(fa["data"], fc["data"]) and np.array_equal(fa["bitpos"], fc["bitpos"])


def test_odd_stacked_bank_covers_the_gaps_between_bins(oracle):
    """SPEC 3.5c: a transmitter half a bin (9.77 kHz) above the centre of bin 100 is out of every even bin's reach and in the middle of
    odd bin 100; one 4 kHz above an even centre decodes in the even bank (the reaches overlap: at 30 dB the odd bank may have it too,
    5.8 kHz off): the two banks together cover the whole band
    (the reference's VFO sits anywhere on a 1 kHz raster, /root/reference/src/main.cpp:14,55-56)."""
    half = 10e6 / 512 / 2
    iq, truth = synth.make_wideband_rs41([100, 333], 10 * BLOCK, seed=6, ebn0_db=30.0, offset_hz=half)
    even, _ = _oracle_decode_wideband(oracle, iq.numpy(), [100, 101, 333, 334], composite=True)
    odd, _ = _oracle_decode_wideband(oracle, iq.numpy(), [100, 333], composite=True, odd=True)
    for k in (100, 333):
        assert len(even[k].frames()) == 0 and len(even[k + 1].frames()) == 0
        fr = odd[k].frames()
        assert len(fr) >= 1 and (fr["nerr"] >= 0).all()
        for f in fr:
            assert any(np.array_equal(tx[8:], f["data"][8:320]) for _, tx in truth[k])
    iq, truth = synth.make_wideband_rs41([100, 333], 10 * BLOCK, seed=6, ebn0_db=30.0, offset_hz=4000.0)
    even, _ = _oracle_decode_wideband(oracle, iq.numpy(), [100, 333], composite=True)
    assert all(len(even[k].frames()) >= 1 for k in (100, 333))
    # the ramp: 256 steps are a whole number of turns, every value lies in [-2, 2]
    L = oracle.lib()
    r = np.array([L.or_chan_ramp(m) for m in range(512)])
    assert np.array_equal(r[:256], r[256:]) and np.abs(r).max() <= 2.0 and r[0] == 0.0 and abs(r[1] - 1.953125) < 1e-7


def test_oracle_decodes_rs41_out_of_a_wideband_scene(oracle):
    bins_active = [100, 333]
    iq, truth = synth.make_wideband_rs41(bins_active, 10 * BLOCK, seed=6, ebn0_db=32.0)
    dec, _ = _oracle_decode_wideband(oracle, iq.numpy(), bins_active)
    for k in bins_active:
        fr = dec[k].frames()
        assert len(fr) >= 1, k
        for f in fr:
            assert (f["nerr"] >= 0).all()
            assert any(np.array_equal(tx[8:], f["data"][8:320]) for _, tx in truth[k])


@pytest.mark.gpu
def test_hip_channelizer_bit_exact_and_decodes(oracle):
    from sdrpp_radiosonde_amd.batch import SondeChannelizer
    bins_active = [5, 100, 333, 511]
    iq, truth = synth.make_wideband_rs41(bins_active, 10 * BLOCK, seed=8, ebn0_db=33.0, device="cuda:0")
    chz = SondeChannelizer(fused=False)            # keep the 48 kS/s rows: the intermediate products are compared below
    assert chz.samples_per_submit == BLOCK and not chz.fused
    got_frames, first = [], None
    for b in range(10):
        chz.submit(iq[b * BLOCK: (b + 1) * BLOCK].contiguous())
        if b == 0:
            first = chz.read()
        got_frames.append(chz.frames())
    got = np.concatenate(got_frames)
    dec, ofirst = _oracle_decode_wideband(oracle, iq.cpu().numpy(), bins_active)
    # intermediate products of the first block, every bin, bit for bit
    assert first[0].tobytes() == ofirst[0].tobytes(), "PFB/FFT output differs"
    assert first[1].tobytes() == ofirst[1].tobytes(), "discriminator/resampler output differs"
    ref = np.concatenate([dec[k].frames() for k in bins_active])
    act = got[np.isin(got["channel"], bins_active)]
    act = act[np.lexsort((act["bitpos"], act["channel"]))]     # per-submit batches -> (bin, time) order
    assert len(ref) >= len(bins_active) and act.tobytes() == ref.tobytes()
    assert len(got) == len(act)                       # silent bins produce no frames
    for f in act:
        assert any(np.array_equal(tx[8:], f["data"][8:320]) for _, tx in truth[int(f["channel"])])


@pytest.mark.gpu
def test_hip_channelizer_two_blocks_per_submit_equals_oracle(oracle):
    """blocks_per_submit = 2 (10 240 steps: two rounds of filter-bank workgroups, the history ping-pong across submits, strided
    views of the caller's buffer read in place): frames identical to the oracle fed block by block."""
    from sdrpp_radiosonde_amd.batch import SondeChannelizer
    bins_active = [17, 256, 480]
    iq, truth = synth.make_wideband_rs41(bins_active, 10 * BLOCK, seed=21, ebn0_db=33.0, device="cuda:0")
    chz = SondeChannelizer(blocks_per_submit=2)
    assert chz.samples_per_submit == 2 * BLOCK
    got = []
    for b in range(5):
        chz.submit(iq[2 * b * BLOCK: 2 * (b + 1) * BLOCK])             # a view: no copy on either side
        got.append(chz.frames())
    got = np.concatenate(got)
    dec, _ = _oracle_decode_wideband(oracle, iq.cpu().numpy(), bins_active, composite=True)
    ref = np.concatenate([dec[k].frames() for k in bins_active])
    act = got[np.lexsort((got["bitpos"], got["channel"]))]
    assert len(ref) >= len(bins_active) and act.tobytes() == ref.tobytes()



@pytest.mark.gpu
def test_hip_channelizer_three_streams_in_one_object(oracle, monkeypatch):
    """sonde_chan_create_multi: S wideband streams per submit, one launch of every stage over all of them (grid.y = stream).
    Stream s carries its own scene: bins, 48 kS/s rows and frames of stream s equal a single-stream object's for that scene
    (which test_hip_channelizer_bit_exact_and_decodes ties to the oracle), channel = 512 s + bin; alternating the submit
    stream exercises the front-end's own cross-stream ordering."""
    import torch
    from sdrpp_radiosonde_amd.batch import SondeChannelizer
    S, NBLK = 3, 10
    scenes = [synth.make_wideband_rs41([40 + 100 * s, 300 + s], NBLK * BLOCK, seed=30 + s, ebn0_db=33.0, device="cuda:0")[0] for s in range(S)]
    multi = SondeChannelizer(n_streams=S, fused=False)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    got, first = [], None
    for b in range(NBLK):
        blk = torch.stack([sc[b * BLOCK: (b + 1) * BLOCK] for sc in scenes]).contiguous()
        torch.cuda.synchronize()
        multi.submit(blk, streams[b & 1].cuda_stream)
        if b == 0:
            first = multi.read()
        got.append(multi.frames())
    got = np.concatenate(got)
    for s in range(S):
        one = SondeChannelizer(fused=False)
        ref = []
        for b in range(NBLK):
            one.submit(scenes[s][b * BLOCK: (b + 1) * BLOCK].contiguous())
            if b == 0:
                rb, ro = one.read()
                assert first[0][512 * s: 512 * (s + 1)].tobytes() == rb.tobytes(), s
                assert first[1][512 * s: 512 * (s + 1)].tobytes() == ro.tobytes(), s
            ref.append(one.frames())
        ref = np.concatenate(ref)
        ref["channel"] += 512 * s
        mine = got[(got["channel"] >= 512 * s) & (got["channel"] < 512 * (s + 1))]
        key = lambda a: a[np.lexsort((a["bitpos"], a["channel"]))]
        assert len(ref) >= 1 and key(mine).tobytes() == key(ref).tobytes(), s


@pytest.mark.gpu
def test_unfused_and_fused_modes_decode_the_same_frames(oracle):
    """The unfused mode (phases -> discriminator + 12/5 resampler kernel -> 48 kS/s rows -> real-input decoder: SPEC 3.5 + 3.0) and
    the fused default (composite filter inside the decoder kernel, SPEC 3.5b) agree in frames (not in the last bit of the loop
    state: the composite filter sums the same products in another order)."""
    from sdrpp_radiosonde_amd.batch import SondeChannelizer
    iq, _ = synth.make_wideband_rs41([77, 400], 10 * BLOCK, seed=41, ebn0_db=30.0, device="cuda:0")
    outs = []
    for fused in (False, True):
        chz = SondeChannelizer(fused=fused)
        assert chz.fused == fused
        fr = []
        for b in range(10):
            chz.submit(iq[b * BLOCK: (b + 1) * BLOCK].contiguous())
            fr.append(chz.frames())
        outs.append(np.concatenate(fr))
        chz.close()
    assert len(outs[0]) >= 2 and np.array_equal(outs[0]["data"], outs[1]["data"]) and np.array_equal(outs[0]["bitpos"], outs[1]["bitpos"])


def test_m10_bins_are_refused():
    """An M10 channel is 50 kHz wide in the reference (main.hpp:48): it does not fit a 19.5 kHz bin, and the library says so
    instead of decoding garbage (on a box without a GPU the create fails anyway: both are errors)."""
    from sdrpp_radiosonde_amd.batch import SondeChannelizer, SondeError
    types = np.zeros(512, dtype=np.uint8)
    types[5] = 3
    with pytest.raises(SondeError):
        SondeChannelizer(types=types, blocks_per_submit=4)


@pytest.mark.gpu
@pytest.mark.parametrize("bps,streams", [(1, 1), (2, 1), (1, 2), (5, 1), (4, 8)])
def test_fused_channelizer_frames_equal_oracle(oracle, bps, streams):
    """The default mode: the per-bin discriminator and the composite resampler + decimator (SPEC 3.5b) run inside the decoder
    kernel (two launches per submit, the 48 kS/s rows never exist).  Frames of every bin == the oracle's (or_chan.c
    or_chan_block2 -> or_channel, pre-decimated input), for
    RS41 and (silent) DFM bins, one and two blocks per submit, one and two streams per object; (4, 8) = the shape
    of bench.py's other_configs.wideband8x4: eight streams, four blocks per submit (4096 bins x 12 tiles per decoder launch)."""
    import torch
    from sdrpp_radiosonde_amd.batch import SondeChannelizer
    bins_active = [9, 130, 257, 500]
    nblk = 12 if bps == 4 else (10 // bps) * bps   # whole submits only: the oracle sees what the channelizer sees
    scenes = [synth.make_wideband_rs41(bins_active, nblk * BLOCK, seed=50 + s, ebn0_db=33.0, device="cuda:0")[0] for s in range(streams)]
    types = np.zeros(512 * streams, dtype=np.uint8)
    m10_bins = [7, 23]                             # two silent bins of another sonde type (DFM: other sync search and frame decoder)
    for s_ in range(streams):
        types[[512 * s_ + k for k in m10_bins]] = 1
    chz = SondeChannelizer(types=types, blocks_per_submit=bps, n_streams=streams)
    assert chz.fused
    got = []
    for b in range(nblk // bps):
        blk = [sc[b * bps * BLOCK: (b + 1) * bps * BLOCK] for sc in scenes]
        chz.submit(torch.stack(blk).contiguous() if streams > 1 else blk[0].contiguous())
        got.append(chz.frames())
    got = np.concatenate(got)
    key = lambda a: a[np.lexsort((a["bitpos"], a["channel"]))]
    refs = []
    for s_, sc in enumerate(scenes):
        dec, _ = _oracle_decode_wideband(oracle, sc.cpu().numpy(), bins_active + m10_bins, types=types[:512], composite=True)
        for k in bins_active + m10_bins:
            r = dec[k].frames().copy()
            r["channel"] = 512 * s_ + k
            refs.append(r)
    ref = np.concatenate(refs)
    assert len(ref) >= len(bins_active) * streams and key(got).tobytes() == key(ref).tobytes()
    # bits and loop state of a silent bin too (no frames there: the frame comparison alone would not see it)
    rb = dec[7].bits()                             # (dec: the last stream's oracle channels)
    c7 = 512 * (streams - 1) + 7
    assert chz.batch.nbits(c7) == len(rb) > 3000 and np.array_equal(chz.batch.read_bits(c7, len(rb) - 3000, 3000), rb[-3000:])
    st, rs = chz.batch.state(c7), dec[7].state()
    assert (st["t_next"], st["period"]) == (rs["t_next"], rs["period"])


@pytest.mark.gpu
def test_dense_scene_frames_equal_oracle(oracle):
    """bench.py's other_configs.wideband8_dense scene (the sparse scene decodes 2 frames per step, the FEC stage is
    idle): an RS41 transmitter in every other bin (256 of 512), built exactly as bench.py builds it (per-transmitter Eb/N0 raised with
    their number, the sum scaled into 16 bits' range), one stream, the 1.024 s scene block by block.  Every bin's frames == the
    oracle's (187 frames: a bin's 1.024 s hold one whole frame or none), none in the empty bins."""
    import torch
    from sdrpp_radiosonde_amd.batch import SondeChannelizer
    bins_active = list(range(1, 512, 2))
    nblk = 8
    scene = synth.make_wideband_rs41(bins_active, nblk * BLOCK, seed=7, ebn0_db=30.0 + 10.0 * np.log10(len(bins_active) / 16.0), device="cuda:0")[0]
    scene *= min(1.0, 4.0 / np.sqrt(len(bins_active)))
    chz = SondeChannelizer(blocks_per_submit=1)
    assert chz.fused
    got = []
    for b in range(nblk):
        chz.submit(scene[b * BLOCK: (b + 1) * BLOCK].contiguous())
        got.append(chz.frames())
    got = np.concatenate(got)
    key = lambda a: a[np.lexsort((a["bitpos"], a["channel"]))]
    dec, _ = _oracle_decode_wideband(oracle, scene.cpu().numpy(), bins_active, composite=True)
    refs = []
    for k in bins_active:
        r = dec[k].frames().copy()
        r["channel"] = k
        refs.append(r)
    ref = np.concatenate(refs)
    assert key(got).tobytes() == key(ref).tobytes()
    per_bin = np.bincount(ref["channel"], minlength=512)
    assert len(ref) >= 150 and per_bin[0::2].sum() == 0, len(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("fused,bits", [(True, 16), (False, 16), (True, 8), (False, 8)])
def test_channelizer_takes_16_bit_wideband_blocks(oracle, fused, bits):
    """sonde_chan_set_input(SONDE_INPUT_IQ16): the wideband block as int16 I, Q pairs (what a 10 MS/s receiver delivers).  Converted
    exactly on the way into the filter bank's window: phases (every bin), frames, bits and loop state equal those of the float
    block holding the same integers, and the frames equal the oracle's on those floats.  Two streams, two blocks per submit, five
    submits (the carried window is kept as integers from submit to submit)."""
    import torch
    from sdrpp_radiosonde_amd import _lib
    from sdrpp_radiosonde_amd.batch import SondeChannelizer
    bins_active, streams, bps, nblk = [9, 130, 257, 500], 2, 2, 10
    scenes = [synth.make_wideband_rs41(bins_active, nblk * BLOCK, seed=70 + s, ebn0_db=33.0, device="cuda:0")[0] for s in range(streams)]
    if bits == 8:       # (SONDE_INPUT_IQ8: four unit-amplitude carriers + noise at 12 counts per unit: inside +-127)
        x16 = [torch.clamp(torch.round(sc * 12.0), -128, 127).to(torch.int8) for sc in scenes]
    else:
        x16 = [torch.clamp(torch.round(sc * 2048.0), -32768, 32767).to(torch.int16) for sc in scenes]
    chz_i = SondeChannelizer(blocks_per_submit=bps, n_streams=streams, fused=fused, input_kind=_lib.INPUT_IQ16 if bits == 16 else _lib.INPUT_IQ8)
    chz_f = SondeChannelizer(blocks_per_submit=bps, n_streams=streams, fused=fused)
    with pytest.raises(Exception):
        chz_i.submit(torch.stack([x[:bps * BLOCK].to(torch.float32) for x in x16]).contiguous())      # a float block into the integer object
    gi, gf = [], []
    for b in range(nblk // bps):
        chz_i.submit(torch.stack([x[b * bps * BLOCK: (b + 1) * bps * BLOCK] for x in x16]).contiguous())
        chz_f.submit(torch.stack([x[b * bps * BLOCK: (b + 1) * bps * BLOCK].to(torch.float32) for x in x16]).contiguous())
        gi.append(chz_i.frames())
        gf.append(chz_f.frames())
        pi, oi = chz_i.read()
        pf, of = chz_f.read()
        assert np.array_equal(pi.view(np.uint32), pf.view(np.uint32)), b            # the phases of every bin and step
        assert fused or np.array_equal(oi.view(np.uint32), of.view(np.uint32))
    gi, gf = np.concatenate(gi), np.concatenate(gf)
    assert len(gi) >= len(bins_active) * streams and gi.tobytes() == gf.tobytes()
    for c in (9, 512 + 257, 7):
        assert chz_i.batch.nbits(c) == chz_f.batch.nbits(c) and chz_i.batch.state(c) == chz_f.batch.state(c)
    key = lambda a: a[np.lexsort((a["bitpos"], a["channel"]))]
    refs = []
    for s_ in range(streams):
        dec, _ = _oracle_decode_wideband(oracle, x16[s_].to(torch.float32).cpu().numpy(), bins_active, composite=fused)
        for k in bins_active:
            r = dec[k].frames().copy()
            r["channel"] = 512 * s_ + k
            refs.append(r)
    assert key(gi).tobytes() == key(np.concatenate(refs)).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("fused,streams", [(True, 4), (False, 2)])
def test_dual_stacking_equals_oracle_and_covers_the_band(oracle, fused, streams):
    """sonde_chan_create_dual (SPEC 3.5c): the even and the odd-stacked bank over the same samples, two streams.  Stream 0 carries
    transmitters half a bin above bins 100 / 333 (decoded by odd bins 100 / 333 only), stream 1 transmitters 3 kHz above bins 40 / 200
    (even bins); (True, 4) = the shape of bench.py's other_configs.wideband4_dual (streams 2 and 3: two more scenes).  Phases of every
    bin of both banks, frames, bits and loop state equal the oracle's two banks."""
    import torch
    from sdrpp_radiosonde_amd.batch import SondeChannelizer
    half = 10e6 / 512 / 2
    NBLK = 10
    sc = [synth.make_wideband_rs41([100, 333], NBLK * BLOCK, seed=6, ebn0_db=30.0, offset_hz=half, device="cuda:0")[0],
          synth.make_wideband_rs41([40, 200], NBLK * BLOCK, seed=9, ebn0_db=30.0, offset_hz=3000.0, device="cuda:0")[0]]
    watch = {0: [100, 333, 101], 1: [40, 200]}
    for extra in range(2, streams):
        sc.append(synth.make_wideband_rs41([7 + extra, 450], NBLK * BLOCK, seed=20 + extra, ebn0_db=30.0, offset_hz=-half if extra & 1 else 0.0, device="cuda:0")[0])
        watch[extra] = [7 + extra, 450, 6 + extra, 449]
    chz = SondeChannelizer(n_streams=streams, dual=True, fused=fused)
    assert chz.n_channels == 1024 * streams and chz.fused == fused
    got, first = [], None
    for b in range(NBLK):
        chz.submit(torch.stack([s[b * BLOCK: (b + 1) * BLOCK] for s in sc]).contiguous())
        if b == 0:
            first = chz.read()
        got.append(chz.frames())
    got = np.concatenate(got)
    key = lambda a: a[np.lexsort((a["bitpos"], a["channel"]))]
    refs = []
    for p in range(streams):
        host = sc[p].cpu().numpy()
        for odd in (False, True):
            dec, ofirst = _oracle_decode_wideband(oracle, host, watch[p], composite=fused, odd=odd)
            base = 1024 * p + (512 if odd else 0)
            assert first[0][base: base + 512].tobytes() == ofirst[0].tobytes(), (p, odd)      # phases of all 512 bins, first block
            for k in watch[p]:
                r = dec[k].frames().copy()
                r["channel"] = base + k
                refs.append(r)
                st, rs = chz.batch.state(base + k), dec[k].state()
                assert (st["t_next"], st["period"]) == (rs["t_next"], rs["period"]) and chz.batch.nbits(base + k) == len(dec[k].bits())
    ref = np.concatenate(refs)
    want_ch = {1024 * 0 + 512 + 100, 1024 * 0 + 512 + 333, 1024 * 1 + 40, 1024 * 1 + 200}
    assert want_ch <= set(ref["channel"].tolist())
    watched = {1024 * p + 512 * o + k for p in range(streams) for o in (0, 1) for k in watch[p]}
    sel = got[np.isin(got["channel"], sorted(watched))]
    assert key(sel).tobytes() == key(ref).tobytes()
    assert not (set(got["channel"].tolist()) & {100, 333})          # the even bank of stream 0 hears nothing of the half-bin transmitters


@pytest.mark.gpu
@pytest.mark.parametrize("streams", [1, 2])
def test_overlapped_submits_equal_serial(streams):
    """The fused channelizer can overlap consecutive submits (option: filter bank on one internal stream, decoder on another,
    two bins buffers).  A host that keeps two submits in flight, and refills ONE staging block on its stream right behind
    each submit (the library makes that stream wait for the filter bank, the block's last reader), gets the frames of the
    serial mode (both kernels in the caller's stream, frames fetched after every submit: what the oracle test above checks)."""
    import torch
    from sdrpp_radiosonde_amd.batch import SondeChannelizer
    bins_active = [9, 130, 257, 500]
    NBLK = 10
    scenes = [synth.make_wideband_rs41(bins_active, NBLK * BLOCK, seed=60 + s, ebn0_db=33.0, device="cuda:0")[0] for s in range(streams)]
    blocks = [torch.stack([sc[b * BLOCK: (b + 1) * BLOCK] for sc in scenes]).contiguous() if streams > 1 else scenes[0][b * BLOCK: (b + 1) * BLOCK].contiguous()
              for b in range(NBLK)]
    ser = SondeChannelizer(n_streams=streams)
    assert ser.fused and not ser.overlap
    want = []
    for blk in blocks:
        ser.submit(blk)
        want.append(ser.frames())
    ser.close()
    assert sum(len(w) for w in want) >= len(bins_active) * streams
    ovl = SondeChannelizer(n_streams=streams, overlap=True)
    assert ovl.fused and ovl.overlap
    stage = torch.empty_like(blocks[0])
    st = torch.cuda.current_stream().cuda_stream
    ovl.batch.ticket()                                  # per-submit completion events from here on
    got = []
    for b, blk in enumerate(blocks):
        stage.copy_(blk)                                # overwrites the block of the previous submit, in stream order
        ovl.submit(stage, st)
        if b >= 1:
            got.append(ovl.batch.frames_of(b))          # the frames of the submit before this one (tickets count from 1)
    got.append(ovl.batch.frames_of(NBLK))
    ovl.close()
    for b in range(NBLK):
        assert got[b].tobytes() == want[b].tobytes(), b


# ---- scenes of tools/wb_campaign.py's kind, fixed and in the driver's run: LOW signal-to-noise ratios, carriers
# up to +-2 kHz off the bin centre, both stackings, integer blocks, a dense band.  (The campaign tool draws hundreds of such scenes at
# random; its logs are under profiles/.  These ten are what every `pytest -m gpu` holds the front-end to.)
WB_SCENES = [
    # (name, seed, Eb/N0 dB, streams, blocks per submit, dual, carrier offset Hz, input bits, occupied bins)
    ("snr8_1stream", 811, 8.0, 1, 1, False, 0.0, 0, 4),
    ("snr10_plus2khz", 812, 10.0, 1, 2, False, 2000.0, 0, 4),
    ("snr10_minus2khz", 813, 10.0, 2, 1, False, -2000.0, 0, 4),
    ("snr12_3streams", 814, 12.0, 3, 2, False, 1300.0, 0, 5),
    ("snr9_dual_halfbin", 815, 9.0, 1, 1, True, 9765.625, 0, 4),
    ("snr11_dual_minus3khz", 816, 11.0, 2, 2, True, -3000.0, 0, 4),
    ("snr9_int16", 817, 9.0, 1, 1, False, -1500.0, 16, 4),
    ("snr12_int8", 818, 12.0, 2, 1, False, 800.0, 8, 4),
    ("snr14_int16_dual", 819, 14.0, 1, 2, True, 2000.0, 16, 4),
    ("snr13_dense256", 820, 13.0, 1, 1, False, 0.0, 0, 256),
]


@pytest.mark.gpu
@pytest.mark.parametrize("scene", WB_SCENES, ids=[s[0] for s in WB_SCENES])
def test_wideband_scenes_low_snr_offsets_dual_integer_blocks(oracle, scene):
    """Fused mode against oracle/or_chan.c (composite path, SPEC 3.5b): the PHASES of the first block of every bin, and for every bin that
    carries a signal (plus two silent DFM bins): frames, bit counts, the newest 4000 ring bits and the timing-loop state.
    Reference chain: /root/reference/src/main.cpp:55-68 (VFO -> FM -> resampler -> decoder, per channel)."""
    import torch
    from sdrpp_radiosonde_amd.batch import SondeChannelizer
    name, seed, ebn0, streams, bps, dual, offset, bits, occ = scene
    nblk = 8
    if occ > 64:
        active = list(range(1, 512, 2))[:occ]
        eb = ebn0 + 10.0 * np.log10(len(active))                 # every transmitter brings its own white noise over the 10 MHz
        dfm = []
    else:
        rng = np.random.default_rng(seed)
        active = sorted(int(x) for x in rng.choice(np.arange(2, 510), size=occ, replace=False))
        dfm = [int(x) for x in rng.choice([b for b in range(2, 510) if b not in active], size=2, replace=False)]
        eb = ebn0 + 10.0 * np.log10(len(active))                 # (the generator adds white noise once PER transmitter: this makes ebn0 the ratio a bin sees)
    scenes = [synth.make_wideband_rs41(active, nblk * BLOCK, seed=seed + s, ebn0_db=eb, device="cuda:0", offset_hz=offset)[0] for s in range(streams)]
    if occ > 64:
        scenes = [sc * min(1.0, 4.0 / np.sqrt(len(active))) for sc in scenes]
    q = None
    if bits == 8:
        q = [torch.clamp(torch.round(sc * 12.0), -128, 127).to(torch.int8) for sc in scenes]
    elif bits == 16:
        q = [torch.clamp(torch.round(sc * 2048.0), -32768, 32767).to(torch.int16) for sc in scenes]
    if q is not None:
        scenes = [x.to(torch.float32) for x in q]              # (the oracle sees the same integers as floats)
    per = 1024 if dual else 512
    types = np.zeros(per * streams, dtype=np.uint8)
    for s_ in range(streams):
        types[[per * s_ + k for k in dfm]] = 1
        if dual:
            types[[per * s_ + 512 + k for k in dfm]] = 1
    chz = SondeChannelizer(types=types, blocks_per_submit=bps, n_streams=streams, dual=dual, input_kind=(3 if bits == 8 else 2) if bits else 0)
    assert chz.fused
    got, first_phases = [], None
    for b in range(nblk // bps):
        blk = [sc[b * bps * BLOCK: (b + 1) * bps * BLOCK] for sc in (q if q is not None else scenes)]
        chz.submit(torch.stack(blk).contiguous() if streams > 1 else blk[0].contiguous())
        if b == 0:
            first_phases = chz.read()[0]
        got.append(chz.frames())
    got = np.concatenate(got)
    key = lambda a: a[np.lexsort((a["bitpos"], a["channel"]))]
    watch = active + dfm
    refs = []
    for s_, sc in enumerate(scenes):
        for odd in ((False, True) if dual else (False,)):
            dec, first = _oracle_decode_wideband(oracle, sc.cpu().numpy(), watch, types=types[:512], composite=True, odd=odd)
            base = per * s_ + (512 if odd else 0)
            # the filter bank's product, every bin of this stream and stacking, the first block (quadrants, bit for bit)
            assert first_phases[base: base + 512, :STEPS].tobytes() == first[0].tobytes(), (name, s_, odd)
            for k in watch:
                r = dec[k].frames().copy()
                r["channel"] = base + k
                refs.append(r)
                rb = dec[k].bits()
                assert chz.batch.nbits(base + k) == len(rb), (name, s_, k)
                tail = min(len(rb), 4000)
                assert np.array_equal(chz.batch.read_bits(base + k, len(rb) - tail, tail), rb[-tail:]), (name, s_, k)
                st, rs = chz.batch.state(base + k), dec[k].state()
                assert (st["t_next"], st["period"], st["bias"], st["amp"]) == (rs["t_next"], rs["period"], rs["bias"], rs["amp"]), (name, s_, k)
    ref = np.concatenate(refs)
    sel = got[np.isin(got["channel"], [per * s_ + o + k for s_ in range(streams) for o in ((0, 512) if dual else (0,)) for k in watch])]
    assert key(sel).tobytes() == key(ref).tobytes(), name
    chz.close()
    # the scenes are not vacuous: something decodes wherever the stacking can hear the carrier, and at these ratios the FEC has work
    if not dual and abs(offset) <= 2000.0:
        assert len(ref) >= (len(active) * streams) // 2, (name, len(ref))
    if ebn0 <= 10.0 and len(ref):
        assert (ref["nerr"] != 0).any(), name
