"""Step time of the headline launch from a cold start (idle clocks, no power history): groups of G steps against wall time.
If the launch is bound by the shader clock (issue / chain), the rate follows the clock (2.33 GHz for the first half second, 2.21
after the package reaches its power limit); if it is bound by HBM, it does not."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from sdrpp_radiosonde_amd.batch import SondeBatch

C, tiles = int(sys.argv[1]), int(sys.argv[2])
G = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda:0")
blocks, types = bench.make_blocks("rs41", C, tiles, 1, 14.0, dev, seed=1000)
stream = torch.cuda.current_stream().cuda_stream
b = SondeBatch(C, tiles * 2048, device=0, types=types, flags=0)
b.submit(blocks[0], stream); b.sync()
time.sleep(4.0)
t00 = time.perf_counter()
rows = []
for g in range(int(3.5 / (G * 0.00027 * C * tiles / 98304)) + 1):
    t0 = time.perf_counter()
    for _ in range(G):
        b.submit(blocks[0], stream)
    b.sync()
    t1 = time.perf_counter()
    rows.append((t1 - t00, (t1 - t0) / G * 1e3))
for k, (t, ms) in enumerate(rows):
    if k < 30 or k % 8 == 0:
        print(f"t={t:6.3f} s  {ms:.4f} ms/step")
b.close()
