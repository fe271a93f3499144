import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sdrpp_radiosonde_amd import synth
from sdrpp_radiosonde_amd.batch import SondeBatch
C, n = 1024, 2048*24
sb = synth.make_rs41_batch(C, n, seed=3, ebn0_db=20.0, device="cuda:0")
host = sb.iq.cpu().numpy()
b = SondeBatch(C, n)
b.submit_host(host); b.sync()
t0=time.perf_counter()
for _ in range(5):
    b.submit_host(host); b.sync()
dt=(time.perf_counter()-t0)/5
print(f"submit_host (pageable host memory -> device -> decode): {C*n/dt/1e6:.1f} Msamples/s, {C*n*8/dt/1e9:.2f} GB/s, {dt*1e3:.1f} ms per {C*n*8/1e6:.0f} MB")
