#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s through demod+FEC at 48 kS/s per channel (BASELINE.json metric).

One "step" = one pass of the hot path (kernel A demod + kernel B framer/FEC) over one batch of
synthetic RS41 IQ that is already resident in HBM.  N=1 workload = BASELINE.json configs[1]:
1024 synthetic RS41-SG channels on one MI355X.  With N>1 every rank owns its own shard of
channels (channels are independent: no data-path collective, weak scaling).

Prints ONE JSON line on rank 0 (see DESIGN.md section 6 for the roofline accounting).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--ramp-ms", type=float, default=250.0,
                    help="untimed submits for this long before the warmup steps: the GPU idles at 157 MHz and needs ~0.1 s of load "
                         "to reach its 2.35 GHz working clock; a 3-step warmup (1 ms) measures the ramp, not the kernel")
    ap.add_argument("--channels", type=int, default=1024, help="channels per GPU")
    ap.add_argument("--tiles", type=int, default=96, help="2048-sample tiles per channel per step (96 = 4.096 s)")
    ap.add_argument("--ebn0", type=float, default=14.0)
    ap.add_argument("--cpu-channels", type=int, default=0, help="channels of the CPU baseline sample (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall time spent on the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--mix", action="store_true", help="BASELINE configs[2]: sonde type = (RS41, M10, DFM09)[channel % 3] (not the headline workload)")
    ap.add_argument("--stride-pad", type=int, default=0, help="experiment: extra samples between channels in HBM")
    ap.add_argument("--scatter", action="store_true", help="ingest on rank 0 and scatter IQ shards over RCCL before timing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # SONDE_BENCH_BACKEND=gloo is a test hook: it lets the N>1 code path run on a box with fewer GPUs than
    # ranks (ranks share devices, scalars are reduced on the host).  The driver's runs use RCCL ("nccl").
    backend = os.environ.get("SONDE_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    from sdrpp_radiosonde_amd import synth
    from sdrpp_radiosonde_amd.batch import SondeBatch
    from sdrpp_radiosonde_amd.shard import scatter_iq

    C, n = args.channels, args.tiles * 2048
    scatter_ms = None
    if args.scatter and world > 1:
        full = None
        if rank == 0:
            full = torch.cat([synth.make_rs41_batch(C, n, seed=1000 + r, ebn0_db=args.ebn0, device=dev, first_channel=r * C).iq
                              for r in range(world)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iq = scatter_iq(full, C, n, dev, src=0)
        torch.cuda.synchronize()
        scatter_ms = (time.perf_counter() - t0) * 1e3
        del full
    elif args.mix:
        order = (0, 3, 1)
        types = np.array([order[c % 3] for c in range(C)], dtype=np.uint8)
        iq = torch.empty((C, n, 2), dtype=torch.float32, device=dev)
        for t in order:
            idx = np.nonzero(types == t)[0]
            iq[torch.from_numpy(idx).to(dev)] = synth.make_batch(int(t), len(idx), n, seed=1000 + rank + 10 * t, ebn0_db=args.ebn0 + 2.0, device=dev).iq
    else:
        iq = synth.make_rs41_batch(C, n, seed=1000 + rank, ebn0_db=args.ebn0, device=dev, first_channel=rank * C).iq
    if args.stride_pad:
        padded = torch.empty((C, n + args.stride_pad, 2), dtype=torch.float32, device=dev)
        padded[:, :n] = iq
        iq = padded[:, :n]
    torch.cuda.synchronize()

    batch = SondeBatch(C, n, device=local_rank, types=types if args.mix else None)
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t_r = time.perf_counter()
    while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:       # clock ramp (untimed, see --ramp-ms)
        for _ in range(32):
            batch.submit(iq, stream)
        batch.sync()
    for _ in range(args.warmup):
        batch.submit(iq, stream)
    nfr_step = batch.sync()
    if args.warmup:
        batch.kernel_ms()   # reset the event ring
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.submit(iq, stream)
    batch.sync()
    barrier()
    dt = time.perf_counter() - t0
    demod_ms, framer_ms = batch.kernel_ms()    # HIP events on the submit stream, averaged over the timed steps
    nfr_step = batch.sync()

    if dist is not None:
        t = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        fr = torch.tensor([nfr_step], device=red_dev, dtype=torch.float64)
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        nfr_total = float(fr.item())
    else:
        nfr_total = float(nfr_step)

    # read-only streaming kernel over the same IQ buffer: what this GPU's HBM delivers to a pure read
    import ctypes
    from sdrpp_radiosonde_amd import _lib
    gbs = ctypes.c_float(0.0)
    if _lib.load().sonde_hbm_read_probe(ctypes.c_void_p(iq.data_ptr()), C * n * 8, 10, ctypes.byref(gbs)) != 0:
        raise RuntimeError(_lib.last_error())
    achievable = float(gbs.value)

    samples_per_step = C * n * world
    msps = samples_per_step * args.steps / dt / 1e6
    # roofline of the dominant kernel (kernel A): algorithmic bytes = 8 B per complex64 sample read once
    # + bits written (n/sps/8 bytes per channel) -- DESIGN.md section 6
    alg_bytes = C * n * 8 + C * (n * 4800 // 48000) // 8
    achieved = alg_bytes / (demod_ms * 1e-3) / 1e9
    # HBM traffic per launch: PMC counters cannot be read from inside this process; the figure comes from the
    # committed rocprofv3 --pmc passes of this same command (profiles/r1_traffic.json) when the workload matches
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
        if tj["channels_per_gpu"] == C and tj["samples_per_channel"] == n:
            traffic = tj["fetch_bytes"] + tj["write_bytes"]
    except (OSError, KeyError, ValueError):
        pass

    out = {
        "metric": "IQ Msamples/s through demod+FEC @ 48 kS/s/ch",
        "value": round(msps, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ramp_ms": args.ramp_ms,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": (f"RS41/M10/DFM09 by channel % 3 x {C} channels/GPU x {n} samples (48 kS/s)" if args.mix else
                                f"RS41-SG x {C} channels/GPU x {n} samples (4800 Bd GFSK, 48 kS/s, Eb/N0 {args.ebn0} dB)"),
                   "channels_per_gpu": C, "samples_per_channel": n, "sharding": f"channels/{world}",
                   "ingest": "rccl-scatter" if scatter_ms is not None else "rank-local"},
        "frames_per_s": round(nfr_total * args.steps / dt, 1),
        "frames_per_step": nfr_total,
        "realtime_channels": round(msps * 1e6 / 48000.0, 1),
        "kernel_ms": {"demod": round(demod_ms, 4), "framer_fec": round(framer_ms, 4)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "achievable_read": round(achievable, 1), "frac_of_achievable": round(achieved / achievable, 4),
                     "algorithmic_bytes": alg_bytes, "kernel": "sd_demod_kernel<true, false, 4>"},
    }
    if scatter_ms is not None:
        out["scatter_ms"] = round(scatter_ms, 3)

    if rank == 0 and world == 1 and not args.no_cpu and not args.mix:
        # CPU baseline: the oracle (plain-C restatement, OpenMP over channels) on this host's cores, on a bounded
        # sample: whole passes over the same channels until >= --cpu-seconds of wall time have been spent
        import oracle_lib
        cores = os.cpu_count() or 1
        cc = args.cpu_channels or C
        host_iq = iq[:cc].cpu().numpy()
        oracle_lib.batch_run(0, host_iq[:min(cc, cores), :2048 * 4], nthreads=cores)   # warm the library and the thread pool
        passes, cdt, nref = 0, 0.0, 0
        while cdt < args.cpu_seconds:
            t0 = time.perf_counter()
            ref = oracle_lib.batch_run(0, host_iq, nthreads=cores)
            cdt += time.perf_counter() - t0
            passes += 1
            nref = int(len(ref))
        c1 = max(1, min(cc, 8))
        t0 = time.perf_counter()
        oracle_lib.batch_run(0, host_iq[:c1], nthreads=1)
        cdt1 = time.perf_counter() - t0
        # parity of full outputs is the job of tests/; here only a sanity figure (frames per pass)
        out["cpu_baseline"] = {"value": round(passes * cc * n / cdt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
                               "sample": f"{passes} passes over {cc} of the same channels x {n} samples "
                                         f"({cdt:.1f} s wall), oracle/ OpenMP over channels",
                               "single_thread_msps": round(c1 * n / cdt1 / 1e6, 3),
                               "frames_per_pass": nref}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
