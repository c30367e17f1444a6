"""The multi-rank path on CPU (gloo, world_size 2): batch sharding, the digest/sample gather to rank 0
and the max-over-ranks timing reduction that bench.py uses.  The per-rank products are stood in for by
the CPU oracle (this is a test of the plumbing; the GPU path is covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import gsw_oracle as O
    from tfhe_amd import dist as gdist

    r, lr, w = gdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    ring = O.fp()
    gsw = O.GSW(ring, 2, 62)
    per = ring.big_n(2) * 3
    pool = O.fill_uniform(ring, 5, 0, 0, ring.q - 1, 2 * batch * per).reshape(2 * batch, ring.big_n(2), 3)
    lo, hi = gdist.shard_range(batch, rank, world)
    mine = gsw.mul_gadget_batch(pool[2 * lo:2 * hi:2], pool[2 * lo + 1:2 * hi:2])
    digests = torch.zeros(batch, dtype=torch.int64)  # equal shapes on every rank: own slice filled
    for k in range(hi - lo):
        digests[lo + k] = np.uint64(O.digest(mine[k])).astype(np.int64)
    gdist.barrier()
    gathered = gdist.gather_to_root(digests)
    sample = gdist.gather_to_root(torch.from_numpy(mine[0].astype(np.int64)))
    slowest = gdist.max_over_ranks(1.0 + rank)
    assert slowest == float(world)
    if rank == 0:
        total = torch.stack(gathered).sum(0).numpy().astype(np.uint64)
        np.savez(out_path, digests=total, sample1=sample[1].numpy())
    else:
        assert gathered is None and sample is None
    gdist.barrier()
    gdist.shutdown()


def test_shard_range_partitions_the_batch():
    sys.path.insert(0, ROOT)
    from tfhe_amd import dist as gdist

    for total in (0, 1, 7, 8, 1024, 16384):
        for world in (1, 2, 3, 8):
            spans = [gdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        gdist.shard_range(8, 2, 2)


def test_two_rank_gather_matches_single_process(tmp_path, oracle):
    batch, world = 5, 2
    out_path = str(tmp_path / "root.npz")
    mp.spawn(_worker, args=(world, _free_port(), batch, out_path), nprocs=world, join=True)
    got = np.load(out_path)
    ring = oracle.fp()
    gsw = oracle.GSW(ring, 2, 62)
    per = ring.big_n(2) * 3
    pool = oracle.fill_uniform(ring, 5, 0, 0, ring.q - 1, 2 * batch * per).reshape(2 * batch, ring.big_n(2), 3)
    want = gsw.mul_gadget_batch(pool[0::2], pool[1::2])
    assert got["digests"].tolist() == [oracle.digest(want[b]) for b in range(batch)]
    lo1, _ = (3, 5)  # rank 1 of 2 owns entries [3, 5) of 5
    assert np.array_equal(got["sample1"].astype(np.uint32), want[lo1])


def _drain_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from tfhe_amd import dist as gdist

    gdist.init(backend="gloo")
    entries, shape = 7, (5, 3)
    mine = (torch.arange(entries * 15, dtype=torch.int32).reshape(entries, *shape) + 1000 * rank)
    drain = gdist.ResultDrain(shape, torch.int32, world, rank, entries, chunk=3, device="cpu")  # 3 + 3 + 1 (padded)
    for _ in range(2):  # reusable: the second batch overwrites the first
        host = drain.run(mine)
    if rank == 0:
        np.save(out_path, host.numpy())
    else:
        assert host is None
    gdist.barrier()
    gdist.shutdown()


def test_result_drain_gathers_every_entry_in_chunks(tmp_path):
    world = 2
    out_path = str(tmp_path / "drain.npy")
    mp.spawn(_drain_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
    got = np.load(out_path)
    base = np.arange(7 * 15, dtype=np.int32).reshape(7, 5, 3)
    assert got.shape == (2, 7, 5, 3)
    assert np.array_equal(got[0], base) and np.array_equal(got[1], base + 1000)


def test_bench_never_reports_fewer_ranks_than_asked_for():
    """`bench.py --gpus N` either runs N ranks or fails: bare on a box with fewer GPUs it refuses before touching a
    device; under a launcher whose WORLD_SIZE disagrees it refuses too.  No JSON line in either case."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box would really run the two ranks")
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GSW_BENCH_REHEARSAL")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"]
    bare = subprocess.run(cmd, cwd=root, env=base, capture_output=True, text=True, timeout=120)
    assert bare.returncode != 0 and "GPU(s) visible" in bare.stderr and "{" not in bare.stdout
    lone = subprocess.run(cmd, cwd=root, env=dict(base, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True,
                          text=True, timeout=120)
    assert lone.returncode != 0 and "WORLD_SIZE=1" in lone.stderr and "{" not in lone.stdout
