import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from sdrpp_radiosonde_amd import synth
from sdrpp_radiosonde_amd.batch import SondeBatch
C, n = 1024, 96 * 2048
import sys as _s
EB = float(_s.argv[1]) if len(_s.argv) > 1 else 14.0
full = synth.make_rs41_cyclic(C, n, 5, seed=1000, ebn0_db=EB, device="cuda:0", chunk=128).iq       # the bench's seamless signal
blocks = [full[:, k * n:(k + 1) * n].contiguous() for k in range(5)]
del full
b = SondeBatch(C, n)
for k in range(200):
    b.submit(blocks[k % 5])
b.sync()
fr = []
for k in range(20):
    b.submit(blocks[k % 5])
    b.sync()
    fr.append(b.frames())
fr = np.concatenate(fr)
d = fr["data"].view(np.uint32).reshape(len(fr), -1)
dbg = d[:, 124:132].astype(np.int64)
names = ["extract", "cwbuild", "decode_pair", "writeback+record", "  syndromes", "  fast paths", "  locator (RiBM)", "  roots+values"]
for i, nm in enumerate(names):
    v = dbg[:, i]
    print(f"{nm:18s} cycles: median {np.median(v):8.0f}  p90 {np.percentile(v,90):8.0f}  max {v.max():8.0f}")
dirty = (fr["nerr"] > 0).any(axis=1)
print("dirty frames:", dirty.mean(), " decode_pair median clean/dirty:", np.median(dbg[~dirty, 2]), np.median(dbg[dirty, 2]) if dirty.any() else None)
dp = dbg[:, 2]
print("decode_pair cycles histogram:", {f"<{hi}": int(((dp >= lo) & (dp < hi)).sum()) for lo, hi in ((0, 3000), (3000, 6000), (6000, 12000), (12000, 20000), (20000, 10**9))})
print("frames", len(fr), "failed", int((fr["nerr"] < 0).any(axis=1).sum()), "nerr histogram", np.bincount(np.clip(fr["nerr"].reshape(-1), 0, 12)))

