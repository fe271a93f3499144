#!/usr/bin/env python3
"""bench_configs.py -- everything bench.py measures that is NOT the headline (round 6 split; VERDICT r5 hygiene): the shared measuring
helpers (resident blocks, the K-step timing loop), the non-headline configurations of the default line (other_configs: BASELINE
configs[2], [3], [4]'s shard and whole, the part-filled shapes in both completion modes, integer rows, the host path), the wideband
run (configs[3]) and the multi-GPU runs (the one-process node host; --multiproc ranks).  bench.py keeps the contract: arguments, the
headline workload (configs[1]), roofline, traffic, cpu_baseline, the ONE JSON line."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

FLAG_PIPELINE = 4
FLAG_JOIN = 16          # (accepted and ignored since round 6: joining at every submit is the default)
FLAG_LATE_JOIN = 32     # launch units joined into the caller's stream one submit late (opt-in; round 5's default)
JOIN_NAMES = ("every submit (default: ordinary stream semantics)", "one submit late (SONDE_FLAG_LATE_JOIN)", "never (SONDE_FLAG_PIPELINE)")
CLASS_NAMES = ("dec1_nt16", "dec2_nt16", "dec4_nt8 (RS41/DFM/iMS-100/MRZ-N1)", "dec2_nt8 (M10)")


def ramp_and_time(submit, sync, args, barrier, reset=None):
    """Clock ramp (untimed), W warmup steps, then EXACTLY K timed steps bracketed by barrier + synchronize.
    reset(): called between warmup and the timed region (empties the library's kernel-event ring)."""
    t_r = time.perf_counter()
    while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:
        for _ in range(32):
            submit()
        sync()
    for _ in range(args.warmup):
        submit()
    sync()
    if reset is not None:
        reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        submit()
    sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0     # this rank's K steps; the caller takes the MAX over ranks, which is when the closing
    barrier()                         # barrier would release -- without charging the barrier's own latency to the steps
    return dt


def cyclic_ok(n, NB):
    """NB blocks of n samples hold a whole number of RS41 frame periods (320 + 64 bytes at 4800 Bd)?"""
    nb = NB * n * 4800.0 / 48000.0
    return abs(nb - round(nb)) < 1e-9 and int(round(nb)) % (8 * 384) == 0


def make_blocks(kind, C, tiles, NB, ebn0, dev, seed, first_channel=0):
    """NB consecutive blocks [C, n, 2] of ONE continuous signal per channel, each its own allocation in HBM, plus the
    per-channel sonde types (None = all RS41).  RS41 channels: the bit stream repeats seamlessly after NB blocks when NB
    blocks hold a whole number of frame periods (5 x 96 or 5 x 24 tiles do); the other sondes are continuous over the NB
    blocks with one discontinuity at the wrap.
    (One allocation per block: a [C, NB * n] view would put the channels 15 x 512 KiB apart, which costs 8 %: HBM channel
    aliasing, profiles/r2_notes.md.)"""
    from sdrpp_radiosonde_amd import synth
    n = tiles * 2048
    types = None

    def rs41(c, fc, sd, eb):
        if NB > 1 and cyclic_ok(n, NB):
            return synth.make_rs41_cyclic(c, n, NB, seed=sd, ebn0_db=eb, device=dev, first_channel=fc, chunk=128).iq
        return synth.make_rs41_batch(c, NB * n, seed=sd, ebn0_db=eb, device=dev, first_channel=fc).iq

    if kind == "rs41":
        full = rs41(C, first_channel, seed, ebn0)
    elif kind == "mix":
        order = (0, 3, 1)
        types = np.array([order[c % 3] for c in range(C)], dtype=np.uint8)
        full = torch.empty((C, NB * n, 2), dtype=torch.float32, device=dev)
        for t in order:
            idx = np.nonzero(types == t)[0]
            part = rs41(len(idx), 0, seed, ebn0 + 2.0) if t == 0 else \
                synth.make_batch(int(t), len(idx), NB * n, seed=seed + 10 * t, ebn0_db=ebn0 + 2.0, device=dev).iq
            full[torch.from_numpy(idx).to(dev)] = part
            del part
    else:
        types = np.full(C, int(kind), dtype=np.uint8)
        full = synth.make_batch(int(kind), C, NB * n, seed=seed, ebn0_db=ebn0 + 2.0, device=dev).iq
    blocks = [full[:, k * n: (k + 1) * n].contiguous() for k in range(NB)] if NB > 1 else [full]
    del full
    torch.cuda.synchronize()
    return blocks, types


def measure(blocks, types, flags, args, local_rank, barrier, stream, input_kind=0):
    """Time args.steps submits cycling through `blocks` (W warmup, clock ramp first); returns the raw figures of this rank."""
    from sdrpp_radiosonde_amd.batch import SondeBatch
    C, n = blocks[0].shape[0], blocks[0].shape[1]
    # frames of a FIRST submit from a fresh decoder: the quantity the CPU baseline's `frames_per_pass` counts
    ts = int(getattr(args, "time_slices", 0) or 0)
    fresh = SondeBatch(C, n, device=local_rank, types=types, flags=flags, input_kind=input_kind, time_slices=ts)
    fresh.submit(blocks[0], stream)
    nfr_first = int(fresh.sync())
    fresh.close()
    batch = SondeBatch(C, n, device=local_rank, types=types, flags=flags, input_kind=input_kind, time_slices=ts)
    launch = batch.launch_info()                   # launch units per submit and how they are joined (the library's choice at these flags)
    turn = [0]

    def submit():
        batch.submit(blocks[turn[0] % len(blocks)], stream)
        turn[0] += 1
    # kernel times: HIP events recorded by the library around the launches of every --time-every-th timed step
    # (an event record is a few microseconds of bubble in the command stream)
    dt = ramp_and_time(submit, batch.sync, args, barrier, reset=lambda: batch.set_timing(args.time_every))
    demod_ms, framer_ms, class_ms = 0.0, 0.0, {}
    if args.time_every:
        demod_ms, framer_ms = batch.kernel_ms()
        class_ms = batch.class_ms()
    nfr_step = 0                                   # frames of one more pass over the cycle, per step
    for _ in range(0 if args.pmc_child else len(blocks)):      # (profiled sub-runs end with the timed steps)
        submit()
        nfr_step += batch.sync()
    nfr_step /= len(blocks)
    batch.close()
    return {"dt": dt, "demod_ms": demod_ms, "framer_ms": framer_ms, "class_ms": class_ms, "nfr_first": nfr_first, "nfr_step": nfr_step, "launch": launch}


def alg_bytes_of(C, n, sample_bytes=8):
    """algorithmic bytes of one step: 8 B per complex64 sample (4 B per 16-bit IQ sample) read once + bits written"""
    return C * n * sample_bytes + C * (n * 4800 // 48000) // 8


def restride(blocks, args):
    """The resident blocks on the channel stride asked for (--row-stride / --stride-pad): views [C, n, 2] of padded allocations."""
    from sdrpp_radiosonde_amd.batch import strided_rows
    n = blocks[0].shape[1]
    if args.stride_pad:
        st = n + args.stride_pad
    elif getattr(args, "row_stride", "pow2") == "pow2":
        st = None                                   # the library's recommendation
    else:
        return blocks
    out = []
    for i in range(len(blocks)):
        out.append(strided_rows(blocks[i], st))
        blocks[i] = None                            # free the contiguous copy before the next block is padded
    return out


def small_run(kind, C, tiles, NB, flags, args, local_rank, dev, barrier, stream, ebn0=None, steps=None, warmup=None, iq16=False, iq8=False):
    """One of the non-headline configurations, measured in this process: a compact record for other_configs / low_snr.
    iq16: the same signal as 16-bit integer IQ rows (SONDE_INPUT_IQ16: full scale 8192 per unit amplitude), 4 bytes per sample."""
    import copy
    a = copy.copy(args)
    # (their own step counts, stated in the record: the pipelined class streams need a few steps to fill and one to drain,
    # which a 20-step region would charge at 3-5 %)
    a.steps = steps or 100
    a.warmup = warmup or 20
    a.ramp_ms = min(args.ramp_ms, 100.0)
    blocks, types = make_blocks(kind, C, tiles, NB, args.ebn0 if ebn0 is None else ebn0, dev, seed=1000)
    if iq16:
        for i in range(len(blocks)):
            blocks[i] = torch.clamp(torch.round(blocks[i] * 8192.0), -32768, 32767).to(torch.int16)
    if iq8:                                                    # (SONDE_INPUT_IQ8: the unit-amplitude signal at 16 counts)
        for i in range(len(blocks)):
            blocks[i] = torch.clamp(torch.round(blocks[i] * 16.0), -128, 127).to(torch.int8)
    blocks = restride(blocks, args)
    m = measure(blocks, types, flags, a, local_rank, barrier, stream, input_kind=3 if iq8 else (2 if iq16 else 0))
    stride_samples = int(blocks[0].stride(0) // 2)
    del blocks
    torch.cuda.empty_cache()
    n = tiles * 2048
    ms = m["dt"] / a.steps * 1e3
    rec = {"channels": C, "samples_per_channel": n, "channel_stride_samples": stride_samples, "blocks_cycled": NB, "flags": flags, "launch_units": m["launch"]["units"],
           "join": JOIN_NAMES[m["launch"]["join"]] if m["launch"]["units"] > 1 else "one launch on the caller's stream",
           "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(ms, 4), "value": round(C * n / (ms * 1e-3) / 1e6, 3), "unit": "Msamples/s",
           "step_frac": round(alg_bytes_of(C, n, 2 if iq8 else (4 if iq16 else 8)) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "frames_per_step_steady": round(m["nfr_step"], 2)}
    if iq16:
        rec["input"] = "int16 IQ (SONDE_INPUT_IQ16): 4 bytes per sample; step_frac counts those"
    if iq8:
        rec["input"] = "int8 IQ (SONDE_INPUT_IQ8): 2 bytes per sample; step_frac counts those"
    if m["class_ms"]:
        rec["kernel_ms"] = {CLASS_NAMES[k]: round(v, 4) for k, v in m["class_ms"].items()}
    elif m["launch"]["units"] > 1:
        rec["kernel_ms"] = {"fork_to_completion": round(m["demod_ms"], 4),
                            "note": "launch units on their own streams: HIP events from a submit's fork to its completion, overlapping the neighbouring submits -- not a kernel duration"}
    else:
        rec["kernel_ms"] = {"demod": round(m["demod_ms"], 4), "framer_fec": round(m["framer_ms"], 4)}
    return rec


def host_e2e_run(C, tiles, NB, args, local_rank, dev, steps=20, iq16=False, iq8=False):
    """The boundary's whole host path at the north_star's per-GPU shape: C channels, one second (24 tiles) at a time, from HOST memory
    (pinned) through sonde_batch_submit_host (PCIe + staging into strided rows), the kernels, and sonde_batch_poll down to
    SondeData fragments with their channel numbers (the reference's callback input, decoder.hpp:59-117, main.cpp:320-331).
    PCIe-inclusive by construction: reported as a real-time factor, never as `value`."""
    import ctypes
    from sdrpp_radiosonde_amd import _lib
    from sdrpp_radiosonde_amd.batch import SondeBatch
    n = tiles * 2048
    blocks, _ = make_blocks("rs41", C, tiles, NB, args.ebn0, dev, seed=1000)
    if iq16:                                  # (16-bit integer IQ in host memory: half the bytes over PCIe)
        blocks = [torch.clamp(torch.round(b * 8192.0), -32768, 32767).to(torch.int16) for b in blocks]
    if iq8:
        blocks = [torch.clamp(torch.round(b * 16.0), -128, 127).to(torch.int8) for b in blocks]
    host = [b.cpu().pin_memory().numpy() for b in blocks]
    del blocks
    torch.cuda.empty_cache()
    batch = SondeBatch(C, n, device=local_rank, input_kind=3 if iq8 else (2 if iq16 else 0))
    L = batch.L
    cap = 65536
    out = (_lib.SondeData * cap)()
    chan = (ctypes.c_uint32 * cap)()

    def step(k):
        batch.submit_host(host[k % NB])
        nfr = batch.sync()
        nfrag = 0
        while True:
            got = L.sonde_batch_poll(batch.h, out, chan, cap)
            if got <= 0:
                break
            nfrag += got
        return nfr, nfrag
    for k in range(NB):                       # warm: staging buffer, parsers, clocks
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fr = fg = 0
    for k in range(steps):
        a, b_ = step(NB + k)
        fr += a
        fg += b_
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    batch.close()
    sig_s = n / 48000.0
    return {"channels": C, "samples_per_channel": n, "steps": steps, "ms_per_step": round(dt * 1e3, 3),
            "value": round(C * n / dt / 1e6, 3), "unit": "Msamples/s (PCIe-inclusive, host to SondeData)",
            "realtime_factor": round(sig_s / dt, 1), "realtime_channels": round(C * sig_s / dt, 0),
            "frames_per_step": round(fr / steps, 1), "fragments_per_step": round(fg / steps, 1),
            "note": "pinned host IQ -> sonde_batch_submit_host (PCIe, strided staging) -> kernels -> sonde_batch_poll -> SondeData fragments, synchronously, "
                    "one step = one second of signal of every channel; realtime_channels = how many 48 kS/s channels this one GPU keeps up with through "
                    "the whole host path"}


def config5_full_run(args, local_rank, dev, steps=20, warmup=5):
    """BASELINE configs[4] WHOLE on one GPU: 65 536 RS41 channels x 49 152 samples (T = 1 s) = 25.8 GB resident (rows on the recommended
    stride: 34 GB), decoded through the node host with devices = (this one,) -- the object that shards the same block over 8 GPUs.
    Parity at this size: tests/test_gpu_full_size.py::test_config5_all_65536_channels_on_one_gpu."""
    from sdrpp_radiosonde_amd import synth
    from sdrpp_radiosonde_amd.batch import row_stride
    from sdrpp_radiosonde_amd.node import SondeNode
    C, tiles, CH = 65536, 24, 8192
    n = tiles * 2048
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 120 * 2 ** 30:
        return {"skipped": f"needs ~110 GB of free HBM, {free / 2 ** 30:.0f} GB are free"}
    t_gen = time.perf_counter()
    block = torch.empty((C, row_stride(n), 2), dtype=torch.float32, device=dev)[:, :n]
    for c0 in range(0, C, CH):                          # 8192 channels at a time (every channel its own signal)
        block[c0: c0 + CH] = synth.make_rs41_batch(CH, n, seed=1000, ebn0_db=args.ebn0, device=dev, first_channel=c0).iq
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    node = SondeNode(C, n, devices=(local_rank,))
    t_r = time.perf_counter()                       # clock ramp first (an idle MI355X sits at 157 MHz and needs ~0.1 s of load), then the warmup steps
    while (time.perf_counter() - t_r) * 1e3 < min(args.ramp_ms, 250.0):
        for _ in range(8):
            node.submit_local([block])
        node.sync()
    for _ in range(warmup):
        node.submit_local([block])
    nfr = node.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        node.submit_local([block])
    node.sync()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    node.close()
    del block
    torch.cuda.empty_cache()
    return {"workload": "BASELINE configs[4] WHOLE on ONE MI355X: 65536 RS41 channels x 49152 samples (T = 1 s) per step, 25.8 GB resident, through sonde_node with one device "
                        "(one block re-submitted: a discontinuity per step; the 8-GPU split of this block is the driver's --gpus 8 run)",
            "channels": C, "samples_per_channel": n, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4),
            "value": round(C * n / (ms * 1e-3) / 1e6, 3), "unit": "Msamples/s", "step_frac": round(alg_bytes_of(C, n) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "realtime_factor": round(n / 48000.0 / (ms * 1e-3), 1), "frames_per_step": int(nfr), "generation_s": round(t_gen, 1)}


def config1_run(args):
    """BASELINE configs[0]: ONE RS41 channel, 10 s of synthetic IQ (481 280 samples = whole tiles) through the CPU restatement on one
    thread (the plumbing case, no GPU) -- and, beside it, the same discriminator stream through the B1 triple (rs41_decode, 0.1 s
    buffers) on the GPU with its per-call latency.  Parity: tests/test_gpu_full_size.py::test_config1_single_channel_cpu_plumbing_equals_the_gpu_decoder."""
    import ctypes
    import oracle_lib
    from sdrpp_radiosonde_amd import _lib, synth
    n = 480000 // 2048 * 2048 + 2048
    sb = synth.make_rs41_batch(1, n, seed=4100, ebn0_db=22.0)
    iq = sb.iq.numpy()
    oracle_lib.batch_run(0, iq[:, :8192], nthreads=1)
    t0 = time.perf_counter()
    ref = oracle_lib.batch_run(0, iq, nthreads=1)
    t_cpu = time.perf_counter() - t0
    L = oracle_lib.lib()
    d = np.zeros(n, dtype=np.float32)
    last = np.zeros(2, dtype=np.float32)
    L.or_discriminate(oracle_lib.fptr(np.ascontiguousarray(iq[0]).reshape(-1)), n, oracle_lib.fptr(d), oracle_lib.fptr(last))
    G = _lib.load()
    dec = G.rs41_decoder_init(48000)
    sd = _lib.SondeData()
    ts, nfrag = [], 0
    for off in range(0, n - 4800 + 1, 4800):
        b = np.ascontiguousarray(d[off: off + 4800])
        t1 = time.perf_counter()
        while G.rs41_decode(dec, ctypes.byref(sd), b.ctypes.data_as(ctypes.c_void_p), 4800) != _lib.PROCEED:
            nfrag += 1
        ts.append(time.perf_counter() - t1)
    G.rs41_decoder_deinit(dec)
    ts = np.array(ts[5:]) * 1e6
    return {"workload": "BASELINE configs[0]: RS41-SG single channel, 481280 samples (10 s) of synthetic 4800 Bd GFSK IQ @ 48 kS/s",
            "cpu": {"kind": "port", "cores": 1, "ms": round(t_cpu * 1e3, 2), "value": round(n / t_cpu / 1e6, 3), "unit": "Msamples/s",
                    "frames": int(len(ref)), "realtime_factor": round(n / 48000.0 / t_cpu, 1)},
            "gpu_b1": {"path": "rs41_decoder_init / rs41_decode (the reference's X_decode slot, decoder.hpp:22,61), real 48 kS/s input, 0.1 s host buffers",
                       "per_call_us_median": round(float(np.median(ts)), 1), "per_call_us_p99": round(float(np.percentile(ts, 99)), 1),
                       "fragments": nfrag, "realtime_factor": round(0.1 / (float(np.mean(ts)) * 1e-6), 1)}}


def other_configs(args, rank, local_rank, world, dev, barrier, reduce_max_sum, stream, C):
    """Everything in the default bench line that is NOT the headline: the other BASELINE configurations (other_configs), the headline
    workload with its rows back to back (contiguous_layout) and at Eb/N0 9 dB (low_snr); each entry has a same-shape, same-flags parity
    test (tests/test_gpu_bench_shapes.py, test_gpu_full_size.py, test_channelizer.py).  Returns the keys to merge into the line."""
    out = {}
    # ---- the other BASELINE configurations and the low-SNR point, measured in this same process
    others = {}
    # every shape that is cut into launch units appears TWICE, labelled with its completion mode: at the default
    # flags (ordinary stream semantics: every submit joined into the caller's stream) and with SONDE_FLAG_LATE_JOIN (opt-in)
    others["mix4096"] = small_run("mix", 4096, 24, 5, 0, args, local_rank, dev, barrier, stream)
    others["mix4096"]["workload"] = ("BASELINE configs[2]: RS41 / M10 / DFM09 by channel % 3, 4096 channels x 49152 samples per step, DEFAULT flags: "
                                     "every submit joined into the caller's stream (ordinary stream semantics)")
    others["mix4096_joined"] = dict(others["mix4096"], workload="= mix4096 (the default IS the joined mode since round 6; the key rounds 4-5 reported SONDE_FLAG_JOIN under)")
    others["mix4096_late_join"] = small_run("mix", 4096, 24, 5, FLAG_LATE_JOIN, args, local_rank, dev, barrier, stream)
    others["mix4096_late_join"]["workload"] = ("the same with SONDE_FLAG_LATE_JOIN (opt-in; round 5's default): one launch unit per sonde type on its own stream, the "
                                               "caller's stream joined one submit late -- the host double-buffers or calls sonde_batch_wait_input")
    others["shard8192"] = small_run("rs41", 8192, 24, 5, 0, args, local_rank, dev, barrier, stream)
    others["shard8192"]["workload"] = "BASELINE configs[4], one GPU's shard: 8192 RS41 channels x 49152 samples (T = 1 s) per step"
    others["rt1250"] = small_run("rs41", 1250, 24, 5, 0, args, local_rank, dev, barrier, stream)
    others["rt1250"]["workload"] = ("north_star's per-GPU share of 10^4 channels on 8 GPUs: 1250 RS41 channels x 49152 samples (T = 1 s) per step "
                                    "(1.22 residencies of 4 workgroups x 256 CUs); DEFAULT flags (ordinary stream semantics): one launch, two time slices")
    others["rt1250_late_join"] = small_run("rs41", 1250, 24, 5, FLAG_LATE_JOIN, args, local_rank, dev, barrier, stream)
    others["rt1250_late_join"]["workload"] = "the same with SONDE_FLAG_LATE_JOIN: two launch units on their own streams joined one submit late, the tail of one overlaps the next submit of the other"
    others["ch1280x96"] = small_run("rs41", 1280, 96, 5, 0, args, local_rank, dev, barrier, stream)
    others["ch1280x96"]["workload"] = "1280 RS41 channels x 196608 samples per step: the headline's rows, 1.25 residencies; DEFAULT flags (ordinary stream semantics): one launch, time-sliced"
    others["ch1280x96_late_join"] = small_run("rs41", 1280, 96, 5, FLAG_LATE_JOIN, args, local_rank, dev, barrier, stream)
    others["ch1280x96_late_join"]["workload"] = "the same with SONDE_FLAG_LATE_JOIN (two launch units, joined one submit late)"
    others["cs16_1024x96"] = small_run("rs41", 1024, 96, 5, 0, args, local_rank, dev, barrier, stream, iq16=True)
    others["cs16_1024x96"]["workload"] = ("the headline's signal as 16-bit integer IQ rows (SONDE_INPUT_IQ16, what SDR hardware delivers): 1024 RS41 channels x 196608 "
                                          "samples per step, 4 bytes per sample; frames identical to the float path on the same integers")
    others["cs16_8192x24"] = small_run("rs41", 8192, 24, 5, 0, args, local_rank, dev, barrier, stream, iq16=True)
    others["cs16_8192x24"]["workload"] = "BASELINE configs[4]'s per-GPU shard as 16-bit integer IQ rows: 8192 RS41 channels x 49152 samples per step"
    others["cs8_1024x96"] = small_run("rs41", 1024, 96, 5, 0, args, local_rank, dev, barrier, stream, iq8=True)
    others["cs8_1024x96"]["workload"] = "the headline's signal as 8-bit integer IQ rows (SONDE_INPUT_IQ8: 2 bytes per sample, the signal at 16 counts)"
    try:
        others["rt1250_host_e2e"] = host_e2e_run(1250, 24, 5, args, local_rank, dev)
        others["rt1250_host_e2e"]["workload"] = ("the north_star's per-GPU share, end to end through the boundary: 1250 RS41 channels, one second at a time, host "
                                                 "memory in, SondeData fragments out")
        others["rt1250_host_e2e_cs16"] = host_e2e_run(1250, 24, 5, args, local_rank, dev, iq16=True)
        others["rt1250_host_e2e_cs16"]["workload"] = "the same from 16-bit integer IQ in host memory (SONDE_INPUT_IQ16)"
        others["rt1250_host_e2e_cs8"] = host_e2e_run(1250, 24, 5, args, local_rank, dev, iq8=True)
        others["rt1250_host_e2e_cs8"]["workload"] = "the same from 8-bit integer IQ in host memory (SONDE_INPUT_IQ8)"
    except Exception as e:                    # (never lets the line fail: the headline above does not depend on it)
        others["rt1250_host_e2e"] = {"error": f"{type(e).__name__}: {e}"}
    for name, S, B in (("wideband", 1, 1), ("wideband1x8", 1, 8), ("wideband8", 8, 1), ("wideband8_dense", 8, 1), ("wideband8x4", 8, 4), ("wideband4_dual", 4, 1), ("wideband8_cs16", 8, 1)):
        import copy
        a = copy.copy(args)
        a.wb_streams, a.wb_blocks = S, B
        a.wb_occupied = 256 if name.endswith("_dense") else 16
        a.wb_dual = name.endswith("_dual")
        a.wb_iq16 = name.endswith("_cs16")
        a.steps, a.warmup, a.ramp_ms = max(40, min(args.steps, 50)), max(8, min(args.warmup, 10)), min(args.ramp_ms, 100.0)
        w = run_wideband(a, rank, local_rank, world, dev, barrier, reduce_max_sum)
        others[name] = {
            "workload": "BASELINE configs[3]: " + w["config"]["workload"], "ms_per_step": w["ms_per_step"], "value": w["value"],
            "unit": w["unit"], "realtime_streams": w["realtime_streams"], "us_per_stream_block": round(w["ms_per_step"] * 1e3 / (S * B), 2),
            "step_frac": w["roofline"]["step_frac"], "kernel_ms": w["kernel_ms"], "frames_per_step": w["frames_per_step"],
            "occupied_bins_per_stream": w["config"]["occupied_bins_per_stream"], "steps": a.steps, "warmup": a.warmup}
    for name, fn in (("config5_full_1gpu", lambda: config5_full_run(args, local_rank, dev)), ("config1_cpu_plumbing", lambda: config1_run(args))):
        try:
            others[name] = fn()
        except Exception as e:                    # (never lets the line fail)
            others[name] = {"error": f"{type(e).__name__}: {e}"}
    out["other_configs"] = others
    if getattr(args, "row_stride", "pow2") == "pow2" and not args.stride_pad:
        # the same workload with the rows back to back, measured in this run: what the layout is worth
        import copy
        ac = copy.copy(args)
        ac.row_stride = "contiguous"
        cl = small_run("rs41", C, args.tiles, args.blocks, args.flags, ac, local_rank, dev, barrier, stream)
        out["contiguous_layout"] = {"channel_stride_samples": cl["channel_stride_samples"], "steps": cl["steps"], "ms_per_step": cl["ms_per_step"],
                                    "step_frac": cl["step_frac"], "kernel_ms": cl["kernel_ms"],
                                    "note": "the headline workload with its rows back to back (--row-stride contiguous)"}
    ls = small_run("rs41", C, args.tiles, args.blocks, args.flags, args, local_rank, dev, barrier, stream, ebn0=9.0)
    out["low_snr"] = {"ebn0": 9.0, "steps": ls["steps"], "ms_per_step": ls["ms_per_step"], "step_frac": ls["step_frac"], "kernel_ms": ls["kernel_ms"],
                      "frames_per_step_steady": ls["frames_per_step_steady"],
                      "note": "the headline workload at Eb/N0 9 dB: most frames need the Reed-Solomon corrector's general path"}
    return out


def scattered_blocks(args, rank, local_rank, world, dev, dist, barrier):
    """--multiproc (one process per GPU): rank 0 holds the IQ of ALL channels of one block at a time (generated block by block:
    synth.make_rs41_cyclic_block), scatters it with torch.distributed.scatter (RCCL over xGMI with backend nccl) and frees it.  The
    scatters are outside the timed region (inputs are resident when timing starts); their time and rate are reported beside the 7-link
    xGMI egress bound.  (The native scatter lives in the ONE-process node host, run_node below: round 6 retired the second, rank-per-GPU
    native stack.)"""
    from sdrpp_radiosonde_amd import synth
    from sdrpp_radiosonde_amd.shard import scatter_iq
    C, n, NB = args.channels, args.tiles * 2048, args.blocks
    cyc = NB > 1 and cyclic_ok(n, NB)
    if not cyc:
        NB = 1                                   # no seamless cycle of this shape: one block, re-submitted
    blocks, ms = [], 0.0
    for k in range(NB):
        full = None
        if rank == 0:
            full = synth.make_rs41_cyclic_block(world * C, n, NB, k, seed=1000, ebn0_db=args.ebn0, device=dev, chunk=128) if cyc else \
                synth.make_rs41_batch(world * C, n, seed=1000, ebn0_db=args.ebn0, device=dev).iq
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        blk = scatter_iq(full, C, n, dev, src=0)
        torch.cuda.synchronize()
        ms += (time.perf_counter() - t0) * 1e3
        blocks.append(blk)
        del full
    torch.cuda.empty_cache()
    sent = (world - 1) * C * n * 8 * NB                      # bytes that left the root
    gbs = sent / (ms * 1e-3) / 1e9
    bound = 7 * 153.0                                        # GB/s: all seven xGMI links of the root at once
    return blocks, None, {
        "ingest": "scatter from rank 0: torch.distributed.scatter",
        "rows_delivered_strided": False,
        "ms": round(ms, 3), "blocks": NB, "bytes_from_root": sent, "gbs": round(gbs, 2),
        "root_egress_bound_gbs": bound, "frac_of_bound": round(gbs / (bound * min(1.0, (world - 1) / 7.0)), 4),
        "note": "root holds one block of all ranks at a time; outside the timed region"}


def run_node(args, launched_ranks=1):
    """--gpus N through the native node-level host, ONE process (include/sonde_node.h, csrc/node.cpp in libsonde_rccl.so).
    Timed region (the contract's): K steps with every GPU's shard RESIDENT in its HBM (sonde_node_submit_local: one
    sonde_batch_submit per device on the node's streams), bracketed by a synchronise of every device; value = samples of all GPUs /
    that time.  Beside it, per step and outside the timed region: the ingest path north_star names -- the IQ of ALL channels on
    GPU 0, scattered over xGMI by sonde_node_submit (scatter_ms: device time on the ingest GPU's stream; bytes; sends; fraction of
    the 7-link egress bound) -- and the return path (gather_ms: frame records of every device to host memory)."""
    from sdrpp_radiosonde_amd.node import SondeNode
    from sdrpp_radiosonde_amd.batch import row_stride, strided_rows
    N = args.gpus
    if args.channels is None:
        args.channels = 1024
    if args.time_every is None:
        args.time_every = 8 if args.steps >= 16 else 4
    if args.flags is None:
        args.flags = 0
    C, n, NB = args.channels, args.tiles * 2048, args.blocks
    devs = [torch.device("cuda", d) for d in range(N)]
    shards = []                                    # [device][block]: the device's channels, rows on the stride asked for
    for d in range(N):
        torch.cuda.set_device(d)
        blocks, _ = make_blocks("rs41", C, args.tiles, NB, args.ebn0, devs[d], seed=1000 + d, first_channel=d * C)
        shards.append(restride(blocks, args))
    NB = len(shards[0])
    torch.cuda.set_device(0)
    node = SondeNode(N * C, n, devices=list(range(N)), ingest=0, flags=args.flags)
    turn = [0]

    def submit():
        node.submit_local([shards[d][turn[0] % NB] for d in range(N)])
        turn[0] += 1

    def sync_all():
        node.sync()
        for d in range(N):
            torch.cuda.synchronize(d)

    for d in range(N):
        node_batch_timing(node, d, 0)
    dt = ramp_and_time(submit, sync_all, args, sync_all, reset=lambda: [node_batch_timing(node, d, args.time_every) for d in range(N)])
    kern = [node_batch_kernel_ms(node, d) for d in range(N)]
    nfr = 0
    for _ in range(NB):
        submit()
        nfr += node.sync()
    nfr /= NB
    samples_per_step = N * C * n
    msps = samples_per_step * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3
    alg = alg_bytes_of(C, n)                        # per GPU
    demod_ms = max((k[0] for k in kern if k), default=0.0)
    demod_ms = min(demod_ms, ms_per_step) if demod_ms > 0 else ms_per_step
    # ---- the ingest path: all channels on GPU 0 (one block), scattered per step; then the frame gather.  The ingest block lies BACK TO
    # BACK (include/sonde_node.h's recommendation: one send per peer straight from the buffer, exactly the shard's bytes, no packing pass)
    scatter, with_scatter = None, None
    try:
        st_full = n
        full = torch.empty((N * C, st_full, 2), dtype=torch.float32, device=devs[0])[:, :n]
        for d in range(N):
            full[d * C: (d + 1) * C] = shards[d][0].to(devs[0])
        torch.cuda.synchronize(0)
        # (a) the SAME K steps with the scatter INSIDE the timed region: every step = sonde_node_submit of the ingest block (scatter over
        # xGMI + the ingest GPU's own strided copy + every device's decode), bracketed like the headline
        def submit_ingest():
            node.submit(full)
        dt_s = ramp_and_time(submit_ingest, sync_all, args, sync_all)
        with_scatter = {"value": round(samples_per_step * args.steps / dt_s / 1e6, 3), "ms_per_step": round(dt_s / args.steps * 1e3, 4)}
        ms, gms, nby, nsend, gby = [], [], 0, 0, 0
        for k in range(6):
            node.submit(full)
            fr = node.frames()
            sst, gst = node.scatter_stats(), node.gather_stats()
            if k:                                   # (the first scatter pays RCCL's connection set-up)
                ms.append(sst["ms"]); gms.append(gst["ms"])
            nby, nsend, gby = sst["bytes_from_ingest"], sst["sends"], gst["bytes"]
        bound = 7 * 153.0 * min(1.0, (N - 1) / 7.0)             # GB/s: the ingest GPU's xGMI links towards its N - 1 peers
        sms = sum(ms) / len(ms)
        scatter = {"ingest": "sonde_node_submit: IQ of all channels on GPU 0 (rows back to back) -> grouped ncclSend / ncclRecv straight into every peer's decoder rows: "
                             "one send per peer of exactly the shard's bytes",
                   "ms": round(sms, 3), "bytes_from_ingest": nby, "sends": nsend, "gbs": round(nby / (sms * 1e-3) / 1e9, 2) if sms > 0 else None,
                   "ingest_egress_bound_gbs": round(bound, 1), "frac_of_bound": round(nby / (sms * 1e-3) / 1e9 / bound, 4) if (sms > 0 and N > 1) else None,
                   "gather_ms": round(sum(gms) / len(gms), 3), "gather_bytes": gby, "frames_gathered": int(len(fr)),
                   "note": "per-step device time of the scatter alone and host time of the gather alone: averages of 5 steps after RCCL's first-call set-up; "
                           "gather = frame records of every device copied to host memory (one process: nothing travels back over xGMI); "
                           "value_with_scatter (top level) is the whole step with this scatter inside the timed region"}
    except Exception as e:                          # (never lets the line fail)
        scatter = {"error": f"{type(e).__name__}: {e}"}
    node.close()
    out = {
        "metric": "IQ Msamples/s through demod+FEC @ 48 kS/s/ch", "value": round(msps, 3), "unit": "Msamples/s", "n_gpus": N,
        "steps": args.steps, "warmup": args.warmup, "ramp_ms": args.ramp_ms, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"RS41-SG x {C} channels/GPU x {n} samples per step (4800 Bd GFSK, 48 kS/s, Eb/N0 {args.ebn0} dB); {NB} consecutive blocks of a continuous signal resident in every GPU's HBM, cycled",
                   "channels_per_gpu": C, "samples_per_channel": n, "sharding": f"channels/{N} (contiguous ranges, sonde_node_shard_range)", "flags": args.flags,
                   "channel_stride_samples": int(shards[0][0].stride(0) // 2),
                   "host": f"ONE process, sonde_node_* (libsonde_rccl.so: ncclCommInitAll over {N} device(s), one SondeBatch per device); ranks launched by the caller: {launched_ranks}",
                   "ingest": "value: resident shards (sonde_node_submit_local) in the timed region; value_with_scatter: ingest on GPU 0, scatter inside the timed region"},
        "frames_per_s": round(nfr * args.steps / dt, 1), "frames_per_step_steady": round(nfr, 2),
        "realtime_channels": round(msps * 1e6 / 48000.0, 1),
        "kernel_ms": {"demod_per_device": [round(k[0], 4) if k else None for k in kern], "framer_fec_per_device": [round(k[1], 4) if k else None for k in kern],
                      "note": "HIP events around every %dth launch on each device (capped at the step for the roofline)" % args.time_every},
        "roofline": {"bound": "hbm", "achieved": round(alg / (demod_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / (demod_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "step_frac": round(alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes": alg,
                     "kernel": "sd_demod_kernel, per GPU (slowest device's HIP-event time, capped at the step); step_frac = one GPU's bytes / the whole step",
                     "kernel_time_source": "HIP events (no nested rocprofv3 pass in node mode)"},
        "nccl_ranks": {"backend": "rccl (ncclCommInitAll, one process)" if N > 1 else "none (one device)", "world": N, "distinct_devices": N},
        "scatter": scatter,
    }
    if scatter and "ms" in scatter:
        out["scatter_ms"], out["gather_ms"] = scatter["ms"], scatter["gather_ms"]
    # BOTH figures, each labelled: `value` = the contract's (inputs resident in every GPU's HBM when the timed region
    # starts); `value_with_scatter` = the same K steps with the IQ of all channels arriving on GPU 0 and scattered over xGMI inside the region
    out["value_label"] = "shards RESIDENT in every GPU's HBM (sonde_node_submit_local): scatter outside the timed region"
    if with_scatter is not None:
        out["value_with_scatter"] = with_scatter["value"]
        out["ms_per_step_with_scatter"] = with_scatter["ms_per_step"]
        out["value_with_scatter_label"] = ("the same K steps, every step = sonde_node_submit of the ingest block on GPU 0: RCCL scatter over xGMI (exactly the "
                                           "shards' bytes) + every device's decode INSIDE the timed region")
    return out


def node_batch_timing(node, d, every):
    import ctypes
    from sdrpp_radiosonde_amd import _lib
    L = _lib.load()
    L.sonde_batch_set_timing(ctypes.c_void_p(node.L.sonde_node_batch(node.h, d)), int(every))


def node_batch_kernel_ms(node, d):
    import ctypes
    from sdrpp_radiosonde_amd import _lib
    L = _lib.load()
    a, b = ctypes.c_float(), ctypes.c_float()
    if L.sonde_batch_kernel_ms(ctypes.c_void_p(node.L.sonde_node_batch(node.h, d)), ctypes.byref(a), ctypes.byref(b)) != 0:
        return None
    return a.value, b.value


def run_wideband(args, rank, local_rank, world, dev, barrier, reduce_max_sum):
    """BASELINE configs[3]: S independent 10 MS/s complex streams -> 512-bin channelizer -> per-bin demod+FEC.
    One step = one block of 1 280 000 wideband samples (0.128 s of signal) per stream."""
    from sdrpp_radiosonde_amd import synth
    from sdrpp_radiosonde_amd.batch import SondeChannelizer

    S = args.wb_streams
    dual = bool(getattr(args, "wb_dual", False))
    iq16 = bool(getattr(args, "wb_iq16", False))                       # the wideband blocks as int16 I, Q pairs (sonde_chan_set_input)
    chan = SondeChannelizer(blocks_per_submit=args.wb_blocks, device=local_rank, n_streams=S, overlap=getattr(args, "wb_overlap", False), dual=dual,
                            input_kind=2 if iq16 else 0)      # ONE object: every stage is one launch over all S streams
    nwb = chan.samples_per_submit
    occ = int(getattr(args, "wb_occupied", 16))
    bins_active = list(range(8, 504, 8))[:occ] if occ <= 62 else list(range(1, 512, 2))[:occ]
    # a 1.024 s scene (8 blocks of 0.128 s) with `occ` RS41 transmitters (16: a sparse band; 256: every other bin, so that the sync
    # search collects frames and the FEC stage decodes them in half the bins), cycled block by block so that the per-bin streams
    # are continuous (one discontinuity per wrap) and frames really decode; stream s runs s blocks ahead of stream 0.  Every
    # transmitter brings its own white noise over the 10 MHz: the per-transmitter Eb/N0 is raised with their number
    NB = 8 // args.wb_blocks
    scene, _ = synth.make_wideband_rs41(bins_active, NB * nwb, seed=7 + rank, ebn0_db=30.0 + 10.0 * np.log10(max(1.0, len(bins_active) / 16.0)), device=dev)
    scene *= min(1.0, 4.0 / np.sqrt(len(bins_active)))               # (the sum of many carriers stays inside 16 bits for --wb-iq16)
    if iq16:
        scene = torch.clamp(torch.round(scene * 1024.0), -32768, 32767).to(torch.int16)      # (16 carriers of unit amplitude + noise: well inside 16 bits)
    one = [scene[i * nwb: (i + 1) * nwb] for i in range(NB)]
    blocks = [torch.stack([one[(i + s) % NB] for s in range(S)]).contiguous() if S > 1 else one[i] for i in range(NB)]
    del scene
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    counter = [0]

    def submit():
        chan.submit(blocks[counter[0] % NB], st)
        counter[0] += 1

    def sync():
        chan.batch.sync()

    chans = [chan]
    dt = ramp_and_time(submit, sync, args, barrier, reset=chans[0].kernel_ms)
    pfb_ms, rs_ms, dem_ms, fr_ms = chans[0].kernel_ms()
    nfr = 0                                                       # frames of one more pass over the scene, per block
    for i in range(NB):
        submit()
        nfr += int(chan.batch.sync())
    dt, nfr_total = reduce_max_sum(dt, nfr / NB)
    samples_per_step = S * nwb * world
    msps = samples_per_step * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3
    alg_bytes = S * nwb * (4 if iq16 else 8)              # every stream's block read once (the filter-bank launch covers all S streams)
    achieved = alg_bytes / (pfb_ms * 1e-3) / 1e9
    return {
        "metric": "wideband IQ Msamples/s through channelizer+demod+FEC @ 10 MS/s/stream",
        "value": round(msps, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ramp_ms": args.ramp_ms, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{S} x 10 MS/s complex IQ -> 512-bin polyphase channelizer (20 kS/s/bin, one phase sample per step) -> FM discriminator (wrapped phase difference) -> 12/5 resampler "
                               f"-> {S} x {1024 if dual else 512} x 48 kS/s RS41 demod+FEC" + (" (both stackings: bins every 9.77 kHz)" if dual else "") +
                               f", one launch per stage over all streams; {nwb} wideband samples per stream per step; {len(bins_active)} of the 512 bins of every stream carry a transmitter",
                   "streams_per_gpu": S, "wideband_samples_per_step": nwb, "occupied_bins_per_stream": len(bins_active)},
        "realtime_factor": round(msps * 1e6 / (S * world * 10e6) , 2),
        "realtime_streams": round(msps / 10.0, 1),
        "narrowband_msps": round(512 * (nwb * 12 // 5 // 500) * S * world * args.steps / dt / 1e6, 3),
        "frames_per_step": round(nfr_total, 2),
        "kernel_ms": dict({"pfb_fft": round(pfb_ms, 4), "demod": round(dem_ms, 4), "framer_fec": round(fr_ms, 4)},
                          **({} if chan.fused else {"disc_resample": round(rs_ms, 4)})),      # (fused: discriminator + resampler run inside the bins decoder; no such kernel)
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": None, "algorithmic_bytes": alg_bytes, "kernel": "sd_pfb_kernel (8 B per wideband sample read once)", "step_frac": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "written_bytes": S * (nwb // 500) * 512 * 2,
                     "note": "algorithmic bytes = the wideband samples read once; the launch also writes 2 B per bin and step (a 16-bit phase, "
                             "re-read once by the decoder) and re-reads its overlapping windows through the L2"},
    }
