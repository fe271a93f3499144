#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s through demod+FEC at 48 kS/s per channel (BASELINE.json metric).

One "step" = one pass of the hot path (demodulator + framer/FEC) over one batch of synthetic RS41 IQ that is
already resident in HBM.  N=1 workload = BASELINE.json configs[1]: 1024 synthetic RS41-SG channels on one MI355X.
With N>1 every rank owns its own shard of channels (channels are independent: no data-path collective, weak
scaling).  `python bench.py --gpus N` works as typed: without a torchrun environment it re-executes itself
under torch.distributed.run with N ranks on 127.0.0.1.

Prints ONE JSON line on rank 0 (DESIGN.md section 6 has the roofline accounting).
Other modes (not the headline): --mix (BASELINE configs[2]), --wideband (configs[3]).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

# the CPU baseline's OpenMP threads must sleep, not spin, between parallel regions: a spinning 256-thread
# team starves the single-thread measurement that follows it (has to be set before any OpenMP runtime loads)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def effective_cpus() -> int:
    """CPUs this process may actually use: the scheduler affinity mask, cut down by a cgroup CPU quota if there is one
    (os.cpu_count() is the machine's, not the container's)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--ramp-ms", type=float, default=250.0,
                    help="untimed submits for this long before the warmup steps: the GPU idles at 157 MHz and needs ~0.1 s of load "
                         "to reach its 2.35 GHz working clock; a 3-step warmup (1 ms) measures the ramp, not the kernel")
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (default: 1024, BASELINE configs[1]; with --mix 4096, configs[2])")
    ap.add_argument("--blocks", type=int, default=5, help="headline workload: consecutive blocks of a continuous, seamlessly repeating signal "
                    "held in HBM and cycled through (1: the same block every step; 5 x 96 tiles = 32 frame periods)")
    ap.add_argument("--tiles", type=int, default=96, help="2048-sample tiles per channel per step (96 = 4.096 s)")
    ap.add_argument("--ebn0", type=float, default=14.0)
    ap.add_argument("--cpu-channels", type=int, default=0, help="channels of the CPU baseline sample (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall time spent on the all-thread CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--mix", action="store_true", help="BASELINE configs[2]: sonde type = (RS41, M10, DFM09)[channel %% 3] (not the headline workload)")
    ap.add_argument("--sonde-type", type=int, default=0, help="all channels of this SONDE_* type (1 DFM09, 2 iMS-100, 3 M10; not the headline workload)")
    ap.add_argument("--wideband", action="store_true", help="BASELINE configs[3]: 10 MS/s IQ -> 512-bin channelizer -> per-bin demod+FEC")
    ap.add_argument("--wb-streams", type=int, default=1, help="--wideband: independent 10 MS/s streams processed per step")
    ap.add_argument("--wb-blocks", type=int, default=1, choices=(1, 2), help="--wideband: blocks of 1 280 000 samples (0.128 s) per submit")
    ap.add_argument("--time-every", type=int, default=None, help="kernel-timing HIP events on every n-th timed step (1: all; default 8, "
                    "4 for runs of fewer than 64 steps).  A timed step carries two event records of 6.4 us of command-stream bubble each "
                    "(profiles/r2_notes.md), inside the timed region: every 8th costs 0.6 %% of the step")
    ap.add_argument("--flags", type=int, default=0, help="SondeBatchConfig.flags (1: RS41 wide, 2: FEC as its own kernel)")
    ap.add_argument("--stride-pad", type=int, default=0, help="experiment: extra samples between channels in HBM")
    ap.add_argument("--scatter", action="store_true", help="ingest on rank 0 and scatter IQ shards over RCCL before timing")
    ap.add_argument("--scatter-torch", action="store_true", help="--scatter through torch.distributed instead of libsonde_rccl.so")
    return ap.parse_args()


def cpu_baseline(iq, C, n, args):
    """The oracle (plain-C restatement, OpenMP over channels) on this host's cores, on a bounded sample of the same
    channels.  Single thread first (>= 1 s of work, nothing else running), then a sweep over thread counts on a short
    sample, then whole passes with the best count until --cpu-seconds of wall time are spent."""
    import oracle_lib
    cores = effective_cpus()
    cc = args.cpu_channels or C
    host_iq = iq[:cc].cpu().numpy()
    # ---- one thread: channels one at a time until >= 1.2 s have been spent
    oracle_lib.batch_run(0, host_iq[:1, :2048 * 4], nthreads=1)            # load the library, touch the code
    t1, c1 = 0.0, 0
    while t1 < 1.2 and c1 < cc:
        t0 = time.perf_counter()
        oracle_lib.batch_run(0, host_iq[c1:c1 + 1], nthreads=1)
        t1 += time.perf_counter() - t0
        c1 += 1
    single = c1 * n / t1 / 1e6
    # ---- thread-count sweep on a short sample (about 1 s each at the single-thread rate x threads)
    cands = sorted({t for t in (1, 2, 4, 8, 16, 32, 64, 128, cores) if t <= cores})
    sweep = {}
    for t in cands:
        if t == 1:
            sweep[1] = round(single, 3)
            continue
        k = int(min(cc, max(t, min(4 * t, t * single * 1e6 / n))))     # >= one channel per thread, about 1 s
        t0 = time.perf_counter()
        oracle_lib.batch_run(0, host_iq[:k], nthreads=t)
        sweep[t] = round(k * n / (time.perf_counter() - t0) / 1e6, 3)
    best_t = max(sweep, key=lambda t: sweep[t])
    # ---- the reported figure: whole passes over the sample with the best thread count
    passes, cdt, nref = 0, 0.0, 0
    while cdt < args.cpu_seconds:
        t0 = time.perf_counter()
        ref = oracle_lib.batch_run(0, host_iq, nthreads=best_t)
        cdt += time.perf_counter() - t0
        passes += 1
        nref = int(len(ref))
    return {"value": round(passes * cc * n / cdt / 1e6, 3), "unit": "Msamples/s", "cores": best_t, "kind": "port",
            "sample": f"{passes} passes over {cc} of the same channels x {n} samples ({cdt:.1f} s wall), oracle/ (plain C, "
                      f"OpenMP over channels); thread count = best of the sweep",
            "host_cpus": {"affinity": len(os.sched_getaffinity(0)), "effective": cores, "os_cpu_count": os.cpu_count()},
            "thread_sweep_msps": {str(k): v for k, v in sweep.items()},
            "single_thread_msps": round(single, 3),
            "single_thread_sample": f"{c1} channels x {n} samples, {t1:.2f} s, before any multi-thread run",
            "frames_per_pass": nref}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: spawn the ranks ourselves (one process per GPU, rendezvous on 127.0.0.1)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback)")
    # SONDE_BENCH_BACKEND=gloo is a test hook: it lets the N>1 code path run on a box with fewer GPUs than
    # ranks (ranks share devices, scalars are reduced on the host).  The driver's runs use RCCL ("nccl").
    backend = os.environ.get("SONDE_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "gloo":
        local_rank %= ndev
    elif ndev < world:
        sys.exit(f"bench.py: --gpus {world} needs {world} visible GPUs, this node has {ndev} (SONDE_BENCH_BACKEND=gloo shares devices for tests)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    nccl_ranks = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # which device every rank sits on, as the collective backend sees it (all-gather of device identities)
        props = torch.cuda.get_device_properties(local_rank)
        ident = f"{socket.gethostname()}:{local_rank}:{getattr(props, 'uuid', props.name)}"
        objs = [None] * world
        dist.all_gather_object(objs, ident)
        nccl_ranks = {"backend": "rccl" if backend == "nccl" else backend, "world": world, "distinct_devices": len(set(objs))}
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max_sum(dt, count):
        if dist is None:
            return dt, float(count)
        t = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([count], device=red_dev, dtype=torch.float64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        return float(t.item()), float(c.item())

    if args.wideband:
        out = run_wideband(args, rank, local_rank, world, dev, barrier, reduce_max_sum)
    else:
        out = run_channels(args, rank, local_rank, world, dev, dist, barrier, reduce_max_sum)
    if nccl_ranks is not None:
        out["nccl_ranks"] = nccl_ranks
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


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


def run_channels(args, rank, local_rank, world, dev, dist, barrier, reduce_max_sum):
    from sdrpp_radiosonde_amd import synth, _lib
    from sdrpp_radiosonde_amd.batch import SondeBatch
    from sdrpp_radiosonde_amd.shard import scatter_iq
    import ctypes

    if args.channels is None:
        args.channels = 4096 if args.mix else 1024
    if args.time_every is None:
        args.time_every = 8 if args.steps >= 64 else 4
    C, n = args.channels, args.tiles * 2048
    scatter_ms = None
    types = None
    blocks = None
    if args.scatter and world > 1:
        # rank 0 ingests the IQ of ALL channels (the same seamless NB-block signal the rank-local mode generates) and scatters
        # it block by block: the native scatter (csrc/shard_rccl.cpp: grouped ncclSend / ncclRecv), or
        # --scatter-torch: dist.scatter.  scatter_ms = the time of all NB scatters, outside the timed region.
        NB = args.blocks
        shards = None
        if rank == 0:
            shards = []
            for r in range(world):
                if NB > 1:
                    f = synth.make_rs41_cyclic(C, n, NB, seed=1000 + r, ebn0_db=args.ebn0, device=dev, first_channel=r * C, chunk=128).iq
                    shards.append([f[:, k * n: (k + 1) * n].contiguous() for k in range(NB)])
                    del f
                else:
                    shards.append([synth.make_rs41_batch(C, n, seed=1000 + r, ebn0_db=args.ebn0, device=dev, first_channel=r * C).iq])
        ns = None
        if not args.scatter_torch:
            from sdrpp_radiosonde_amd.shard import NativeShard
            ns = NativeShard(local_rank)
        blocks, scatter_ms = [], 0.0
        for k in range(NB):
            full = torch.cat([shards[r][k] for r in range(world)]) if rank == 0 else None
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            blk = ns.scatter_iq(full, (C, n, 2), root=0) if ns is not None else scatter_iq(full, C, n, dev, src=0)
            torch.cuda.synchronize()
            scatter_ms += (time.perf_counter() - t0) * 1e3
            blocks.append(blk)
            del full
        del shards
        iq = blocks[0]
        if NB == 1:
            blocks = None
    elif args.mix:
        order = (0, 3, 1)
        types = np.array([order[c % 3] for c in range(C)], dtype=np.uint8)
        iq = torch.empty((C, n, 2), dtype=torch.float32, device=dev)
        for t in order:
            idx = np.nonzero(types == t)[0]
            iq[torch.from_numpy(idx).to(dev)] = synth.make_batch(int(t), len(idx), n, seed=1000 + rank + 10 * t, ebn0_db=args.ebn0 + 2.0, device=dev).iq
    elif args.sonde_type:
        types = np.full(C, args.sonde_type, dtype=np.uint8)
        iq = synth.make_batch(args.sonde_type, C, n, seed=1000 + rank, ebn0_db=args.ebn0 + 2.0, device=dev).iq
    else:
        # headline workload: NB consecutive blocks of one continuous signal per channel, cycled, so that every step decodes
        # NEW samples of a seamless stream (re-submitting one block makes a junk frame per channel and step at the seam, which
        # costs the RS corrector's full 24 iterations: an artefact of the bench, not of the signal)
        NB = args.blocks
        if NB > 1:
            full = synth.make_rs41_cyclic(C, n, NB, seed=1000 + rank, ebn0_db=args.ebn0, device=dev, first_channel=rank * C, chunk=128).iq
            # one allocation per block: a [C, NB * n] view would put the channels 15 x 512 KiB apart, which costs 8 % (HBM channel
            # aliasing; 3 x 512 KiB, the contiguous block, does not)
            blocks = [full[:, k * n: (k + 1) * n].contiguous() for k in range(NB)]
            del full
            iq = blocks[0]
        else:
            iq = synth.make_rs41_batch(C, n, seed=1000 + rank, ebn0_db=args.ebn0, device=dev, first_channel=rank * C).iq
    if args.stride_pad:
        padded = torch.empty((C, n + args.stride_pad, 2), dtype=torch.float32, device=dev)
        padded[:, :n] = iq
        iq = padded[:, :n]
    torch.cuda.synchronize()

    stream = torch.cuda.current_stream().cuda_stream
    # frames of a FIRST submit from a fresh decoder: the quantity the CPU baseline's `frames_per_pass` counts
    fresh = SondeBatch(C, n, device=local_rank, types=types, flags=args.flags)
    fresh.submit(iq, stream)
    nfr_first = int(fresh.sync())
    fresh.close()

    batch = SondeBatch(C, n, device=local_rank, types=types, flags=args.flags)

    # kernel times: HIP events recorded by the library on the submit stream around the launches of every
    # --time-every-th timed step (an event record is a few microseconds of bubble in the command stream)
    def reset():
        if hasattr(batch.L, "sonde_batch_set_timing"):
            batch.set_timing(args.time_every)
        else:
            batch.kernel_ms()
    if blocks is None:
        blocks = [iq]
    turn = [0]

    def submit():
        batch.submit(blocks[turn[0] % len(blocks)], stream)
        turn[0] += 1
    dt = ramp_and_time(submit, batch.sync, args, barrier, reset=reset)
    demod_ms, framer_ms = batch.kernel_ms()
    nfr_step = 0                                   # frames of one more pass over the cycle, per step
    for _ in range(len(blocks)):
        submit()
        nfr_step += batch.sync()
    nfr_step /= len(blocks)
    dt, nfr_total = reduce_max_sum(dt, nfr_step)

    # read-only streaming kernel over the same IQ buffer: what this GPU's HBM delivers to a pure read
    gbs = ctypes.c_float(0.0)
    if _lib.load().sonde_hbm_read_probe(ctypes.c_void_p(iq.data_ptr()), C * n * 8, 10, ctypes.byref(gbs)) != 0:
        raise RuntimeError(_lib.last_error())
    achievable = float(gbs.value)

    samples_per_step = C * n * world
    msps = samples_per_step * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3
    # roofline of the dominant kernel (the demodulator): algorithmic bytes = 8 B per complex64 sample read once
    # + bits written (n/sps/8 bytes per channel) -- DESIGN.md section 6
    alg_bytes = C * n * 8 + C * (n * 4800 // 48000) // 8
    achieved = alg_bytes / (demod_ms * 1e-3) / 1e9
    step_achieved = alg_bytes / (ms_per_step * 1e-3) / 1e9
    # HBM traffic per launch: PMC counters cannot be read from inside this process; the figure is REPLAYED from the
    # committed rocprofv3 --pmc passes of this same command (profiles/*_traffic.json) when the workload matches
    traffic, traffic_source = None, None
    for name in ("r2_traffic.json", "r1_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
            if tj["channels_per_gpu"] == C and tj["samples_per_channel"] == n and not args.mix:
                traffic = tj["fetch_bytes"] + tj["write_bytes"]
                traffic_source = f"replayed from profiles/{name} (separate rocprofv3 --pmc passes of this command on the builder's box, FETCH_SIZE x2 gfx950 correction); not measured in this run"
                break
        except (OSError, KeyError, ValueError):
            continue

    out = {
        "metric": "IQ Msamples/s through demod+FEC @ 48 kS/s/ch",
        "value": round(msps, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ramp_ms": args.ramp_ms,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": (f"RS41/M10/DFM09 by channel % 3 x {C} channels/GPU x {n} samples (48 kS/s)" if args.mix else
                                f"SONDE type {args.sonde_type} x {C} channels/GPU x {n} samples (48 kS/s)" if args.sonde_type else
                                f"RS41-SG x {C} channels/GPU x {n} samples per step (4800 Bd GFSK, 48 kS/s, Eb/N0 {args.ebn0} dB)"
                                + (f"; {len(blocks)} consecutive blocks of a continuous signal resident in HBM, cycled" if len(blocks) > 1 else "")),
                   "channels_per_gpu": C, "samples_per_channel": n, "sharding": f"channels/{world}",
                   "ingest": ("rank-local" if scatter_ms is None else
                              "scatter from rank 0: torch.distributed" if args.scatter_torch else "scatter from rank 0: libsonde_rccl (grouped ncclSend/ncclRecv)")},
        "frames_per_s": round(nfr_total * args.steps / dt, 1),
        "frames_per_step_steady": round(nfr_total, 2),
        "frames_first_submit": nfr_first,
        "realtime_channels": round(msps * 1e6 / 48000.0, 1),
        "kernel_ms": {"demod": round(demod_ms, 4), "framer_fec": round(framer_ms, 4),
                      "note": f"HIP events on every {args.time_every}th timed step; for RS41 the demod kernel includes sync search and FEC"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "step_achieved": round(step_achieved, 2), "step_frac": round(step_achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_source,
                     "achievable_read": round(achievable, 1), "frac_of_achievable": round(achieved / achievable, 4),
                     "algorithmic_bytes": alg_bytes, "kernel": "sd_demod_kernel (dominant kernel of the step; frac = its HIP-event time, step_frac = whole step by the wall clock)"},
    }
    if scatter_ms is not None:
        out["scatter_ms"] = round(scatter_ms, 3)
    if rank == 0 and world == 1 and not args.no_cpu and not args.mix and not args.sonde_type:
        out["cpu_baseline"] = cpu_baseline(iq, C, n, args)
        out["cpu_baseline"]["frames_match_gpu_first_submit"] = bool(out["cpu_baseline"]["frames_per_pass"] == nfr_first) \
            if (args.cpu_channels or C) == C else None
    return out


def run_wideband(args, rank, local_rank, world, dev, barrier, reduce_max_sum):
    """BASELINE configs[3]: S independent 10 MS/s complex streams -> 512-bin channelizer -> per-bin demod+FEC.
    One step = one block of 1 280 000 wideband samples (0.128 s of signal) per stream."""
    from sdrpp_radiosonde_amd import synth
    from sdrpp_radiosonde_amd.batch import SondeChannelizer

    S = args.wb_streams
    chans = [SondeChannelizer(blocks_per_submit=args.wb_blocks, device=local_rank) for _ in range(S)]
    nwb = chans[0].samples_per_submit
    bins_active = list(range(8, 504, 8))
    # a 1.024 s scene (8 blocks of 0.128 s) with 16 RS41 transmitters, cycled block by block so that the per-bin streams
    # are continuous (one discontinuity per wrap) and frames really decode
    NB = 8 // args.wb_blocks
    scene, _ = synth.make_wideband_rs41(bins_active[:16], NB * nwb, seed=7 + rank, ebn0_db=30.0, device=dev)
    blocks = [scene[i * nwb: (i + 1) * nwb] for i in range(NB)]
    torch.cuda.synchronize()
    # one HIP stream per wideband stream: their (small) kernels overlap on the GPU
    hip_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(S - 1)]
    counter = [0]

    def submit():
        blk = blocks[counter[0] % NB]
        counter[0] += 1
        for c, st in zip(chans, hip_streams):
            c.submit(blk, st.cuda_stream)

    def sync():
        for c in chans:
            c.batch.sync()

    dt = ramp_and_time(submit, sync, args, barrier, reset=chans[0].kernel_ms)
    pfb_ms, rs_ms, dem_ms, fr_ms = chans[0].kernel_ms()
    nfr = 0                                                       # frames of one more pass over the scene, per block
    for i in range(NB):
        submit()
        nfr += sum(int(c.batch.sync()) for c in chans)
    dt, nfr_total = reduce_max_sum(dt, nfr / NB)
    samples_per_step = S * nwb * world
    msps = samples_per_step * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3
    alg_bytes = nwb * 8                                   # one stream's block read once (PFB kernel)
    achieved = alg_bytes / (pfb_ms * 1e-3) / 1e9
    return {
        "metric": "wideband IQ Msamples/s through channelizer+demod+FEC @ 10 MS/s/stream",
        "value": round(msps, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ramp_ms": args.ramp_ms, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{S} x 10 MS/s complex IQ -> 512-bin polyphase channelizer (40 kS/s/bin) -> FM discriminator -> 6/5 resampler "
                               f"-> 512 x 48 kS/s RS41 demod+FEC; {nwb} wideband samples per stream per step", "streams_per_gpu": S,
                   "wideband_samples_per_step": nwb},
        "realtime_factor": round(msps * 1e6 / (S * world * 10e6) , 2),
        "realtime_streams": round(msps / 10.0, 1),
        "narrowband_msps": round(512 * (nwb * 6 // 5 // 250) * S * world * args.steps / dt / 1e6, 3),
        "frames_per_step": round(nfr_total, 2),
        "kernel_ms": {"pfb_fft": round(pfb_ms, 4), "disc_resample": round(rs_ms, 4), "demod": round(dem_ms, 4), "framer_fec": round(fr_ms, 4)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": None, "algorithmic_bytes": alg_bytes, "kernel": "sd_pfb_kernel (8 B per wideband sample read once)",
                     "note": "one 80 MB/s stream is latency-bound, nowhere near the HBM roofline; frac is reported for completeness"},
    }


if __name__ == "__main__":
    main()
