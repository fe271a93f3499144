#!/usr/bin/env python3
"""Why is a 20-step timed region slower per step than a 200-step one?  Per-step device times (HIP events on the submit
stream) of the headline workload right behind a synchronize, after the clock ramp; with and without idle time before."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sdrpp_radiosonde_amd.batch import SondeBatch

dev = torch.device("cuda:0")
blocks, types = bench.make_blocks("rs41", 1024, 96, 5, 14.0, dev, seed=1000)
b = SondeBatch(1024, 96 * 2048)
st = torch.cuda.current_stream().cuda_stream
i = [0]
def submit():
    b.submit(blocks[i[0] % 5], st); i[0] += 1
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.25:
    for _ in range(32): submit()
    b.sync()
for idle_us in (0, 100, 1000, 10000):
    for rep in range(2):
        for _ in range(5): submit()
        b.sync(); torch.cuda.synchronize()
        if idle_us: time.sleep(idle_us * 1e-6)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
        tw = time.perf_counter()
        ev[0].record()
        for k in range(40):
            submit(); ev[k + 1].record()
        b.sync(); torch.cuda.synchronize()
        wall = (time.perf_counter() - tw) * 1e3
        d = [ev[k].elapsed_time(ev[k + 1]) for k in range(40)]
        print(f"idle {idle_us:6d} us: wall/40 {wall/40:.4f} ms | steps 0-3 {' '.join(f'{x:.3f}' for x in d[:4])} | mean 4-19 {sum(d[4:20])/16:.4f} | mean 20-39 {sum(d[20:])/20:.4f}")
