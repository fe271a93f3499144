import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sdrpp_radiosonde_amd import synth
from sdrpp_radiosonde_amd.batch import SondeBatch
C, tiles = int(sys.argv[1]), int(sys.argv[2])
n = tiles * 2048
iq = synth.make_rs41_batch(C, n, seed=1000, ebn0_db=14.0, device="cuda:0", chunk=128).iq
b = SondeBatch(C, n)
for _ in range(100): b.submit(iq)
b.sync()
fr = []
for k in range(10):
    b.submit(iq); b.sync(); fr.append(b.frames())
fr = np.concatenate(fr)
d = fr["data"].view(np.uint32).reshape(len(fr), -1)[:, 128:132].astype(np.int64)
for i, nm in enumerate(["prologue (entry -> first round)", "tile loop", "epilogue (to this frame's end)"]):
    print(f"{nm:34s} cycles median {np.median(d[:, i]):9.0f} p90 {np.percentile(d[:, i], 90):9.0f}")
tot = d[:, 0] + d[:, 1] + d[:, 2]
print("per tile (loop / tiles):", np.median(d[:, 1]) / tiles, " total median", np.median(tot), " prologue share", np.median(d[:,0])/np.median(tot), " epilogue share", np.median(d[:,2])/np.median(tot))
