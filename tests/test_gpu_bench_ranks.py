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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rank-local", "--no-cpu", "--channels", "64",
                          "--tiles", "12", "--steps", "4", "--warmup", "2", "--ramp-ms", "10"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                    # ONE line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["sharding"] == "channels/2" and j["config"]["ingest"] == "rank-local"
    assert j["nccl_ranks"]["world"] == 2 and j["nccl_ranks"]["backend"] == "gloo"
    assert j["frames_per_step_steady"] > 0 and j["value"] > 0
    # whole-job aggregate: both ranks' samples over the slowest rank's time
    assert abs(j["value"] - 2 * 64 * 12 * 2048 / (j["ms_per_step"] * 1e-3) / 1e6) < 0.02 * j["value"]


@pytest.mark.gpu
def test_bench_scatter_ingest_two_ranks_under_gloo():
    """`bench.py --gpus 2` as typed ingests the way north_star names it: rank 0 generates ONE block of every rank's channels at a
    time and scatters it (here through torch.distributed, the two ranks sharing the one GPU under the gloo test hook; the
    native RCCL scatter needs one device per rank: test_bench_native_scatter_two_gpus)."""
    env = dict(os.environ, SONDE_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu",
                          "--channels", "64", "--tiles", "12", "--steps", "4", "--warmup", "2", "--ramp-ms", "10"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["config"]["ingest"].startswith("scatter from rank 0") and j["scatter_ms"] > 0
    assert j["scatter"]["blocks"] == 5 and j["scatter"]["bytes_from_root"] == 5 * 64 * 12 * 2048 * 8
    assert j["frames_per_step_steady"] == pytest.approx(2 * 64 * 12 * 2048 / 10 / 3072, rel=0.02)      # one frame per 3072 symbols and channel


@pytest.mark.gpu
def test_bench_native_scatter_two_gpus():
    """`python bench.py --gpus 2` exactly as the driver types it, when the box has two GPUs: ONE process, the node host (csrc/node.cpp,
    ncclCommInitAll, grouped ncclSend / ncclRecv scatter); both figures in the line, each labelled."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the one-GPU boxes run the gloo variant above)")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SONDE_BENCH_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu", "--channels", "256",
                          "--tiles", "24", "--steps", "10", "--warmup", "3", "--ramp-ms", "50"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert j["config"]["host"].startswith("ONE process, sonde_node_") and j["nccl_ranks"]["world"] == 2
    assert j["value_with_scatter"] > 0 and j["value"] >= j["value_with_scatter"] and "value_label" in j and "value_with_scatter_label" in j
    assert j["scatter"]["bytes_from_ingest"] == 256 * 24 * 2048 * 8 and j["scatter"]["sends"] == 1
    assert j["frames_per_step_steady"] == pytest.approx(2 * 256 * 24 * 2048 / 10 / 3072, rel=0.02)


def test_bench_refuses_more_ranks_than_gpus_without_the_hook():
    """CPU: --gpus N with a torchrun environment of another size is an error with a clear message, not an assert."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=3" in (out.stderr + out.stdout)
