#!/usr/bin/env python3
"""Differential campaign for the wideband front-end's fused mode (GPU box): random scenes (seed, Eb/N0, bins, streams, blocks
per submit, sonde-type map with 4:1 and 2:1 bins), every bin that carries a signal or runs the 2:1 class compared with the
oracle's composite path (SPEC 3.5b): frames, every bit of the bit ring, timing-loop state.  usage: python tools/wb_campaign.py [n]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import oracle_lib as oracle  # noqa: E402
import test_channelizer as tc  # noqa: E402
from sdrpp_radiosonde_amd import synth  # noqa: E402
from sdrpp_radiosonde_amd.batch import SondeChannelizer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(77)
t0 = time.time()
for it in range(n):
    streams = int(rng.choice([1, 1, 2, 3, 8]))
    bps = int(rng.choice([1, 2, 5, 4]))
    nblk = 12 if bps == 4 else 10
    dual = bool(rng.integers(0, 4) == 0) and streams <= 3           # a quarter of the scenes: both stackings (SPEC 3.5c)
    offset = float(rng.choice([0.0, 9765.625, -3000.0])) if dual else 0.0
    ebn0 = float(rng.uniform(9.0, 30.0))
    seed = int(rng.integers(10, 10_000))
    active = sorted(int(x) for x in rng.choice(np.arange(2, 510), size=4, replace=False))
    m10 = [int(x) for x in rng.choice([b for b in range(2, 510) if b not in active], size=2, replace=False)]
    scenes = [synth.make_wideband_rs41(active, nblk * tc.BLOCK, seed=seed + s, ebn0_db=ebn0, device="cuda:0", offset_hz=offset)[0] for s in range(streams)]
    iq16 = (False, False, True, 8)[int(rng.integers(0, 4))]   # a quarter of the scenes as int16 I, Q blocks, a quarter as int8 (sonde_chan_set_input)
    if iq16 == 8:
        q16 = [torch.clamp(torch.round(sc * 12.0), -128, 127).to(torch.int8) for sc in scenes]
        scenes = [q.to(torch.float32) for q in q16]
    elif iq16:
        q16 = [torch.clamp(torch.round(sc * 2048.0), -32768, 32767).to(torch.int16) for sc in scenes]
        scenes = [q.to(torch.float32) for q in q16]      # (the oracle sees the same integers as floats)
    per = 1024 if dual else 512
    types = np.zeros(per * streams, dtype=np.uint8)
    for s_ in range(streams):
        types[[per * s_ + k for k in m10]] = 1          # silent bins of another sonde type (DFM)
        if dual:
            types[[per * s_ + 512 + k for k in m10]] = 1
    chz = SondeChannelizer(types=types, blocks_per_submit=bps, n_streams=streams, dual=dual, input_kind=(3 if iq16 == 8 else 2) if iq16 else 0)
    assert chz.fused
    got = []
    for b in range(nblk // bps):
        blk = [sc[b * bps * tc.BLOCK: (b + 1) * bps * tc.BLOCK] for sc in (q16 if iq16 else scenes)]
        chz.submit(torch.stack(blk).contiguous() if streams > 1 else blk[0].contiguous())
        got.append(chz.frames())
    got = np.concatenate(got)
    key = lambda a: a[np.lexsort((a["bitpos"], a["channel"]))]
    refs, nbits = [], 0
    for s_, sc in enumerate(scenes):
      for odd in ((False, True) if dual else (False,)):
        dec, _ = tc._oracle_decode_wideband(oracle, sc.cpu().numpy(), active + m10, types=types[:512], composite=True, odd=odd)
        for k in active + m10:
            r = dec[k].frames().copy()
            r["channel"] = per * s_ + (512 if odd else 0) + k
            refs.append(r)
            rb = dec[k].bits()
            c = per * s_ + (512 if odd else 0) + k
            assert chz.batch.nbits(c) == len(rb), (it, s_, k)
            tail = min(len(rb), 4000)
            assert np.array_equal(chz.batch.read_bits(c, len(rb) - tail, tail), rb[-tail:]), (it, s_, k)
            st, rs = chz.batch.state(c), dec[k].state()
            assert (st["t_next"], st["period"]) == (rs["t_next"], rs["period"]), (it, s_, k)
            nbits += tail
    ref = np.concatenate(refs)
    watched = sorted(set(ref["channel"].tolist())) if len(ref) else []
    sel = got[np.isin(got["channel"], [per * s_ + o + k for s_ in range(streams) for o in ((0, 512) if dual else (0,)) for k in active + m10])]
    assert key(sel).tobytes() == key(ref).tobytes(), it
    chz.close()
    print(f"[{it + 1}/{n}] seed {seed} Eb/N0 {ebn0:5.1f} dB streams {streams} blocks/submit {bps} dual {int(dual)}{' int8' if iq16 == 8 else (' int16' if iq16 else '')} offset {offset:.0f} Hz bins {active} + DFM {m10}: "
          f"{len(ref)} frames, {nbits} ring bits, loop state of {len(active + m10) * streams} bins identical to the oracle", flush=True)
print(f"wideband campaign done in {time.time() - t0:.0f} s")
