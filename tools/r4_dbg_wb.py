import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
oracle.build()
from sdrpp_radiosonde_amd import synth
from sdrpp_radiosonde_amd.batch import SondeChannelizer
from test_channelizer import _oracle_decode_wideband, BLOCK
bps, streams = int(sys.argv[1]), int(sys.argv[2])
nblkreq = int(sys.argv[3]) if len(sys.argv) > 3 else 10
bins_active = [9, 130, 257, 500]
nblk = (nblkreq // bps) * bps
scenes = [synth.make_wideband_rs41(bins_active, nblk * BLOCK, seed=50 + s, ebn0_db=33.0, device="cuda:0")[0] for s in range(streams)]
types = np.zeros(512 * streams, dtype=np.uint8)
m10_bins = [7, 23]
for s_ in range(streams):
    types[[512 * s_ + k for k in m10_bins]] = 3
chz = SondeChannelizer(types=types, blocks_per_submit=bps, n_streams=streams)
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
        r = dec[k].frames().copy(); r["channel"] = 512 * s_ + k; refs.append(r)
    for k in bins_active + m10_bins:
        c = 512 * s_ + k
        rb = dec[k].bits()
        nb = chz.batch.nbits(c)
        same = nb == len(rb) and np.array_equal(chz.batch.read_bits(c, max(0, nb - 3000), min(nb, 3000)), rb[-3000:])
        st, rs = chz.batch.state(c), dec[k].state()
        print("stream", s_, "bin", k, "nbits", nb, len(rb), "bits_same", same, "state_same", (st["t_next"], st["period"]) == (rs["t_next"], rs["period"]))
ref = np.concatenate(refs)
g, r = key(got), key(ref)
print("frames got", len(g), "ref", len(r), "equal", g.tobytes() == r.tobytes())
print("got  ", [(int(f["channel"]), int(f["bitpos"]), f["nerr"].tolist()) for f in g])
print("ref  ", [(int(f["channel"]), int(f["bitpos"]), f["nerr"].tolist()) for f in r])
