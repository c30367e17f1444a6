"""bench.py's multi-rank control flow on a one-GPU box: two ranks launched the way the driver launches
them (torch.distributed.run), both on GPU 0 over gloo (GSW_BENCH_REHEARSAL=1; RCCL refuses two ranks on
one device).  Checks the contract of the printed line, not its numbers."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("workload_args,expect_config", [
    (["--workload", "fp_n10", "--batch", "64"], None),
    (["--config", "3", "--batch", "4", "--pool", "8"], 3),  # the 8-GPU configuration's preset, cut down
    (["--config", "5", "--batch", "4", "--pool", "8"], 5),  # the power-of-two ring configuration's preset, cut down
])
def test_two_rank_rehearsal_prints_one_contract_line(workload_args, expect_config):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GSW_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1"] + workload_args
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout  # rank 0 only
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["value"] > 0
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    # rank 0 compared what it gathered (every rank's digests and sampled product) with the stored products
    assert out["config"]["exchange_verified"] is True and out["config"]["benchmark_config"] == expect_config
    assert out["roofline"]["unit"] == "TOP/s"
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


@pytest.mark.gpu
def test_bare_gpus_2_makes_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it starts the two ranks itself (rehearsal mode: both on GPU 0
    over gloo) and the line says how many ranks the process group saw."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GSW_BENCH_REHEARSAL"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "fp_n10", "--batch", "64"]
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world_size_seen"] == 2 and out["config"]["backend"] == "gloo"
    assert out["config"]["exchange_verified"] is True
    # without the rehearsal switch a one-GPU box has too few devices: refused, non-zero, nothing printed as a result
    env.pop("GSW_BENCH_REHEARSAL")
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    import torch
    if torch.cuda.device_count() < 2:
        assert proc.returncode != 0 and "GPU(s) visible" in proc.stderr
        assert not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_result_drain_on_the_gpu_without_peers():
    """The device side of dist.ResultDrain (receive buffers, side stream, pinned host memory) with one rank."""
    import torch

    sys.path.insert(0, ROOT)
    from tfhe_amd import dist as gdist

    dev = torch.device("cuda", 0)
    entries, shape = 7, (33, 20)
    data = torch.arange(entries * 33 * 20, dtype=torch.int32, device=dev).reshape(entries, *shape)
    drain = gdist.ResultDrain(shape, torch.int32, world=1, rank=0, entries=entries, chunk=3, device=dev)
    for scale in (1, 2):  # second call reuses the buffers
        host = drain.run(data * scale)
        assert host.is_pinned() and host.shape == (1, entries) + shape
        assert torch.equal(host[0], (data * scale).cpu())
