#!/usr/bin/env python3
"""DESIGN.md section 6's table, from the files of ONE evidence run (tools/evidence.sh on one box).

  python tools/design_table.py profiles/r6           # prints the markdown
  python tools/design_table.py profiles/r6 --write   # replaces the region between the table markers of DESIGN.md

Reads <prefix>_bench.json (python bench.py), <prefix>_bench_driver_args.json (--steps 20 --warmup 5), <prefix>_rocprof.csv,
<prefix>_wb_rocprof.csv, <prefix>_counters.csv, <prefix>_wb_counters.csv, <prefix>_pytest_gpu.log.  Nothing is typed in by hand."""
import json
import os
import re
import sys

PARITY = {
    "headline": "`test_config2_1024_rs41_channels_full_size`",
    "contiguous_layout": "`test_rows_on_the_recommended_stride_and_host_staging`",
    "low_snr": "`test_headline_shape_at_9_db`",
    "mix4096": "`test_mix4096_five_blocks_back_to_back[0]`, `test_config3_4096_mixed_channels_full_size[0]`",
    "mix4096_late_join": "`test_mix4096_five_blocks_back_to_back[32]`, `test_config3_4096_mixed_channels_full_size[32]`",
    "rt1250_late_join": "`test_part_filled_last_generation[1250-24-32]`",
    "ch1280x96_late_join": "`test_part_filled_last_generation[1280-96-32]`",
    "wideband1x8": "`test_fused_channelizer_frames_equal_oracle[5-1]` (5 blocks per submit; 8: the bench's)",
    "config5_full_1gpu": "`test_config5_all_65536_channels_on_one_gpu`",
    "config1_cpu_plumbing": "`test_config1_single_channel_cpu_plumbing_equals_the_gpu_decoder`",
    "shard8192": "`test_config5_shard_8192_rs41_channels`",
    "rt1250": "`test_part_filled_last_generation[1250-24-0]`",
    "ch1280x96": "`test_part_filled_last_generation[1280-96-0]`",
    "cs16_1024x96": "`test_iq16_equals_float_path_and_oracle`",
    "cs16_8192x24": "`test_16_bit_rows_at_the_bench_shapes`",
    "cs8_1024x96": "`test_16_bit_rows_at_the_bench_shapes[1024-96-8]`",
    "wideband": "`test_fused_channelizer_frames_equal_oracle[1-1]`",
    "wideband8": "`test_fused_channelizer_frames_equal_oracle[1-2]`, `[4-8]`",
    "wideband8_dense": "`test_dense_scene_frames_equal_oracle`",
    "wideband8x4": "`test_fused_channelizer_frames_equal_oracle[4-8]`",
    "wideband4_dual": "`test_dual_stacking_equals_oracle_and_covers_the_band[True-4]`",
    "wideband8_cs16": "`test_channelizer_takes_16_bit_wideband_blocks[True-16]`",
}


def load(p):
    with open(p) as f:
        txt = f.read()
    i = txt.index('{"metric"')
    return json.loads(txt[i:txt.index("\n", i)] if "\n" in txt[i:] else txt[i:])


def rows_of(p):
    """tools/rocprof_summary.py's lines: db,kernel,calls,avg_us,min_us,max_us,counter,counter_avg,corrected_bytes -- the kernel name may hold commas"""
    with open(p) as f:
        for ln in f:
            ln = ln.rstrip("\n")
            if not ln or ln.startswith("db,"):
                continue
            db, rest = ln.split(",", 1)
            kernel, calls, avg, mn, mx, ctr, cavg, cb = rest.rsplit(",", 7)
            yield dict(db=db, kernel=kernel, calls=calls, avg_us=avg, min_us=mn, max_us=mx, counter=ctr, counter_avg=cavg, corrected_bytes=cb)


def kernels(p):
    out = {}
    if not os.path.exists(p):
        return out
    for row in rows_of(p):
        out.setdefault(row["db"], {})[row["kernel"]] = row
    return out


def counters(p):
    out = {}
    if not os.path.exists(p):
        return out
    if True:
        for row in rows_of(p):
            if row.get("counter"):
                out.setdefault(row["kernel"], {})[row["counter"]] = (float(row["counter_avg"]), row.get("corrected_bytes") or "")
    return out


def main():
    pre = sys.argv[1]
    d = load(pre + "_bench.json")
    dd = load(pre + "_bench_driver_args.json")
    rows = []
    r = d["roofline"]
    rows.append(("headline (`python bench.py`, %d steps)" % d["steps"], "1024 RS41 x 96 tiles, flags 0: one launch", d["ms_per_step"], d["value"] / 1e3,
                 "**%.3f** (kernel %.1f us by the trace in the run: **%.3f**)" % (r["step_frac"], r["kernel_us_trace"], r["frac"]), PARITY["headline"]))
    r2 = dd["roofline"]
    rows.append(("headline, the driver's arguments (%d steps)" % dd["steps"], "the same", dd["ms_per_step"], dd["value"] / 1e3,
                 "%.3f (kernel %.1f us: %.3f)" % (r2["step_frac"], r2["kernel_us_trace"], r2["frac"]), "the same"))
    samples = 1024 * 196608
    for k, shape in (("contiguous_layout", "rows back to back"), ("low_snr", "the headline at Eb/N0 9 dB")):
        v = d[k]
        rows.append(("`%s`" % k, shape, v["ms_per_step"], samples / v["ms_per_step"] / 1e6, "%.3f" % v["step_frac"], PARITY[k]))
    oc = d["other_configs"]
    for k, v in oc.items():
        if k.startswith("rt1250_host_e2e") or k == "mix4096_joined" or "error" in v or "skipped" in v:
            continue
        if k == "config1_cpu_plumbing":
            rows.append(("`%s`" % k, "1 RS41 channel x 481280 samples: the CPU port on one thread; beside it the B1 triple on the GPU", "%.2f (CPU)" % v["cpu"]["ms"],
                         "%.3f (CPU)" % (v["cpu"]["value"] / 1e3), "B1: %.1f us per 0.1 s call" % v["gpu_b1"]["per_call_us_median"], PARITY[k]))
            continue
        if k == "config5_full_1gpu":
            rows.append(("`%s`" % k, "65536 ch x 24 tiles through sonde_node, one device", v["ms_per_step"], v["value"] / 1e3, "%.3f" % v["step_frac"], PARITY[k]))
            continue
        if "channels" in v:
            shape = "%d ch x %d tiles, flags %d: %d unit%s, %s" % (v["channels"], v["samples_per_channel"] // 2048, v["flags"], v["launch_units"],
                                                                  "" if v["launch_units"] == 1 else "s", v["join"])
        else:
            m = re.search(r"(\d+) x 10 MS/s", v["workload"])
            nb = 4 if k.endswith("x4") else (8 if k.endswith("x8") else 1)
            shape = "%s stream%s x %d block%s, %d of 512 bins occupied" % (m.group(1), "" if m.group(1) == "1" else "s", nb, "s" if nb > 1 else "", v["occupied_bins_per_stream"])
            km = v["kernel_ms"]
            shape += "; filter bank %.1f + decoder %.1f us" % (1e3 * km.get("pfb_fft", 0), 1e3 * km.get("demod", 0))
        rows.append(("`%s`" % k, shape, v["ms_per_step"], v["value"] / 1e3, "%.3f" % v["step_frac"], PARITY.get(k, "")))
    e2e = [oc[k] for k in ("rt1250_host_e2e", "rt1250_host_e2e_cs16", "rt1250_host_e2e_cs8") if k in oc]
    if e2e:
        rows.append(("`rt1250_host_e2e` / `_cs16` / `_cs8`", "1250 channels x 1 s from pinned host memory to SondeData fragments", " / ".join("%.2f" % v["ms_per_step"] for v in e2e),
                     " / ".join("%.1f" % (v["value"] / 1e3) for v in e2e) + " (PCIe-inclusive)", " / ".join("%.0f x" % v["realtime_factor"] for v in e2e) + " real time",
                     "`test_host_path_at_the_target_shape`, `test_iq16_from_host_memory`"))
    out = ["| entry of the bench line | shape, flags, launch units | ms per step | Gsample/s | step_frac | parity test at this shape |", "|---|---|---|---|---|---|"]
    for name, shape, ms, gs, frac, par in rows:
        out.append("| %s | %s | %s | %s | %s | %s |" % (name, shape, ms if isinstance(ms, str) else "%.4f" % ms, gs if isinstance(gs, str) else "%.1f" % gs, frac, par))
    out.append("")
    # kernels by the stand-alone traces, counters
    ks = kernels(pre + "_rocprof.csv")
    for db, kk in ks.items():
        for name, row in kk.items():
            if "demod" in name:
                out.append("Stand-alone trace of the headline command (`%s_rocprof.csv`): `%s` %s launches, avg **%s us** (min %s, max %s) = %.3f of the roofline."
                           % (os.path.basename(pre), name, row["calls"], row["avg_us"], row["min_us"], row["max_us"], r["algorithmic_bytes"] / float(row["avg_us"]) / 8e6))
    cs = counters(pre + "_counters.csv")
    for name, c in cs.items():
        if "demod" in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            tr = float(c["FETCH_SIZE"][1]) + float(c["WRITE_SIZE"][1])
            out.append("HBM traffic of that kernel by the counters (`%s_counters.csv`; FETCH_SIZE x 2 x 1024 B + WRITE_SIZE x 1024 B per launch): %.4f GB = **%.3f x** algorithmic; "
                       "VALU %.1f M instructions, LDS bank conflicts %.1f M of %.1f M LDS cycles." % (os.path.basename(pre), tr / 1e9, tr / r["algorithmic_bytes"],
                       c.get("SQ_INSTS_VALU", (0,))[0] / 1e6, c.get("SQ_LDS_BANK_CONFLICT", (0,))[0] / 1e6, c.get("SQ_LDS_IDX_ACTIVE", (0,))[0] / 1e6))
    wk = kernels(pre + "_wb_rocprof.csv")
    for db, label in (("p_wb", "1 stream"), ("p_wb8", "8 streams")):
        if db in wk:
            out.append("Config 4, %s, stand-alone trace (`%s_wb_rocprof.csv`): " % (label, os.path.basename(pre))
                       + ", ".join("`%s` %s us" % (n.split("<")[0], row["avg_us"]) for n, row in sorted(wk[db].items(), reverse=True)) + ".")
    wc = counters(pre + "_wb_counters.csv")
    tot = 0.0
    parts = []
    for name, c in wc.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f, w = float(c["FETCH_SIZE"][1]), float(c["WRITE_SIZE"][1])
            tot += f + w
            parts.append("`%s` %.1f MB fetched + %.1f MB written" % (name.split("<")[0], f / 1e6, w / 1e6))
    if parts:
        out.append("Config 4, 8 streams, HBM traffic per step (`%s_wb_counters.csv`): %s = **%.1f MB = %.2f x** the 81.9 MB of input." % (os.path.basename(pre), "; ".join(parts), tot / 1e6, tot / 81.92e6))
    cb = d["cpu_baseline"]
    out.append("CPU baselines on the box's %d usable cores: oracle port %.2f Gsample/s (%.3f on one thread), conventional receiver %.2f Gsample/s; `frames_first_submit` (GPU) %d = the oracle's `frames_per_pass` %d."
               % (cb["cores"], cb["value"] / 1e3, cb.get("single_thread_msps", 0) / 1e3, cb["conventional"]["value"] / 1e3, d["frames_first_submit"], cb.get("frames_per_pass", -1)))
    lg = pre + "_pytest_gpu.log"
    if os.path.exists(lg):
        with open(lg) as f:
            m = re.findall(r"^(?:=+ )?((?:\d+ failed, )?\d+ passed.*?) in ([\d.]+)s", f.read(), re.M)
        if m:
            out.append("`pytest -m gpu` on the same box: %s in %s s." % m[-1])
    text = "\n".join(out)
    if "--write" in sys.argv:
        p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "DESIGN.md")
        with open(p) as f:
            s = f.read()
        a, b = "<!-- r6-table-begin -->", "<!-- r6-table-end -->"
        i, j = s.index(a) + len(a), s.index(b)
        with open(p, "w") as f:
            f.write(s[:i] + "\n" + text + "\n" + s[j:])
    else:
        print(text)


if __name__ == "__main__":
    main()
