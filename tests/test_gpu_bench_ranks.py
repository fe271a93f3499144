"""-m gpu: bench.py's N>1 code path with the HIP decoder, on a one-GPU box.

`python bench.py --gpus 2` as typed (no torchrun environment) must spawn its own ranks; SONDE_BENCH_BACKEND=gloo lets the
two ranks share the one device (scalars are reduced on the host).  What runs per rank is the real thing: synthetic IQ on
the GPU, SondeBatch through the C ABI, barrier + max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_under_gloo():
    env = dict(os.environ, SONDE_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu", "--channels", "64",
                          "--tiles", "12", "--steps", "4", "--warmup", "2", "--ramp-ms", "10"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                    # ONE line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["sharding"] == "channels/2"
    assert j["nccl_ranks"]["world"] == 2 and j["nccl_ranks"]["backend"] == "gloo"
    assert j["frames_per_step_steady"] > 0 and j["value"] > 0
    # whole-job aggregate: both ranks' samples over the slowest rank's time
    assert abs(j["value"] - 2 * 64 * 12 * 2048 / (j["ms_per_step"] * 1e-3) / 1e6) < 0.02 * j["value"]


@pytest.mark.gpu
def test_bench_scatter_ingest_two_ranks_under_gloo():
    """--scatter: rank 0 generates every rank's blocks of the seamless signal and scatters them block by block (here through
    torch.distributed, the two ranks sharing the one GPU; the native RCCL scatter needs one device per rank)."""
    env = dict(os.environ, SONDE_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scatter", "--scatter-torch", "--no-cpu",
                          "--channels", "64", "--tiles", "12", "--steps", "4", "--warmup", "2", "--ramp-ms", "10"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["config"]["ingest"].startswith("scatter from rank 0") and j["scatter_ms"] > 0
    assert j["frames_per_step_steady"] == pytest.approx(2 * 64 * 12 * 2048 / 10 / 3072, rel=0.02)      # one frame per 3072 symbols and channel


def test_bench_refuses_more_ranks_than_gpus_without_the_hook():
    """CPU: --gpus N with a torchrun environment of another size is an error with a clear message, not an assert."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=3" in (out.stderr + out.stdout)
