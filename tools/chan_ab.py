import torch, numpy as np, time, sys
sys.path.insert(0, ".")
from sdrpp_radiosonde_amd.batch import SondeChannelizer
for fused in (True, False):
    for S, B in ((1, 1), (8, 1), (8, 2)):
        c = SondeChannelizer(n_streams=S, blocks_per_submit=B, fused=fused)
        x = torch.randn((S, c.samples_per_submit, 2), device="cuda:0") * 0.1
        if S == 1: x = x[0].contiguous()
        for _ in range(30): c.submit(x)
        c.batch.sync(); c.kernel_ms()
        t0 = time.perf_counter()
        for _ in range(80): c.submit(x)
        c.batch.sync(); dt = (time.perf_counter() - t0) / 80 * 1e6
        print("fused", fused, "S", S, "blocks", B, "us/step", round(dt, 1), [round(v * 1e3, 1) for v in c.kernel_ms()], flush=True)
