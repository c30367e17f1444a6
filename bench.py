#!/usr/bin/env python
"""Headline benchmark: GSW ciphertext mults/s (whole job) on N MI355X GPUs.

A step is one pass of the hot path over one batch of synthetic input resident in HBM:
`batch` independent products Pc = G^-1(C1c) * C2c mod q (right-operand preparation + int8-MFMA
contraction, whose epilogue also accumulates a 64-bit digest of every product).  Default workload: the headline configuration,
batch 1024, n = 256, p = 2^31 - 1, on one GPU.  With N > 1 every rank runs its own batch (weak
scaling, no collective inside the contraction) and the digests plus one sampled product per rank
are gathered to rank 0 over RCCL inside the timed region.

`--config K` selects configuration K of the benchmark list at its real per-GPU size: 2 = the default;
3 = 16 384 products of n = 512 over Fp sharded over 8 GPUs, i.e. 2 048 per GPU; 5 = 4 096 products of
n = 512 over Z/2^32.

The engine and every torch operation of this file run on ONE explicit (non-default) HIP stream, so the
operand fill, the contraction, the gather and the checks are ordered by the stream itself.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import tfhe_amd  # noqa: E402
from tfhe_amd import _lib, dist as gdist  # noqa: E402

INT8_DENSE_PEAK_TOPS = 5033.0  # 256 CU x 4 SIMD x 2048 int8 ops/clk x 2.4 GHz (2x the bf16 dense MFMA peak)
HBM_PEAK_GBS = 8000.0
# the contraction kernels of the shipped library (a -DGSW_DIAGNOSTIC_BUILDS library also accepts the superseded ones)
KERNEL_NAMES = {2: "binlimb_gemm_kernel<dma>", 3: "binlimb_gemm_kernel<regs>", 9: "binlimb_gemm16_skew"}

WORKLOADS = {
    # name: (ring, n, default batch)
    "fp_n256": (lambda: tfhe_amd.Fp(), 256, 1024),      # the headline configuration: 1024 products, n = 256, p < 2^31
    "fp_n512": (lambda: tfhe_amd.Fp(), 512, 256),       # the 8-GPU configuration's shape (n = 512), a short batch
    "z32_n512": (lambda: tfhe_amd.Zpow2(32), 512, 256),  # the power-of-two ring configuration's shape (q = 2^32, n = 512)
    "fp_n10": (lambda: tfhe_amd.Fp(), 10, 1024),        # the reference's own test parameters (plumbing)
    # diagnostic shapes without any padding (n + 1 a multiple of 32, rows a multiple of 128): tile time vs K depth
    "fp_n127": (lambda: tfhe_amd.Fp(), 127, 4096),
    "fp_n255": (lambda: tfhe_amd.Fp(), 255, 1024),
    "fp_n511": (lambda: tfhe_amd.Fp(), 511, 256),
}
# --config K: configuration K (1-based) of the benchmark list at its real per-GPU size -> (workload, batch per GPU, pool)
CONFIGS = {
    2: ("fp_n256", 1024, 2048),   # 1024 products, n = 256, one GPU (the default)
    3: ("fp_n512", 2048, 1024),   # 16 384 products over 8 GPUs = 2 048 per GPU; pool of 1024 ciphertexts (34 GB)
    5: ("z32_n512", 4096, 1024),  # 4 096 products, q = 2^32, n = 512; outputs 142 GB + pool 36 GB of 288 GB
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=0, choices=[0] + sorted(CONFIGS),
                    help="benchmark configuration (1-based) at its real per-GPU size; overrides --workload, and "
                         "gives the defaults of --batch / --pool")
    ap.add_argument("--workload", default="fp_n256", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="products per GPU per step (default: the workload's)")
    ap.add_argument("--pool", type=int, default=0, help="distinct resident ciphertexts (default 2 * batch, capped)")
    ap.add_argument("--gather", choices=["digest", "full"], default="digest",
                    help="what rank 0 collects each step: 64-bit digests of every product plus one sampled product "
                         "per rank (default), or every product, drained to pinned host memory in chunks (bounded "
                         "by the root's xGMI / PCIe ingress, not by the kernels)")
    ap.add_argument("--diag-pool", choices=["uniform", "zero", "sparse"], default="uniform",
                    help="diagnostic: operand data pattern (anything but uniform is not the benchmark)")
    ap.add_argument("--variant", type=int, default=None, help="contraction kernel variant: 9 (default), 2 or 3 (the simple cross-check kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="target CPU work for the baseline sample")
    return ap.parse_args()


def algorithmic_ops(big_n: int, n: int) -> float:
    """int8 ops of one product: binary left operand, 4 byte limbs of the right one, no padding credit."""
    return 2.0 * big_n * big_n * (n + 1) * 4


def algorithmic_bytes(big_n: int, n: int) -> float:
    return 3.0 * big_n * (n + 1) * 4


def load_traffic(workload: str, variant: int, products_per_launch: float):
    """HBM-side bytes per contraction launch: the per-product figure of the committed rocprofv3 PMC
    passes (profiles/pmc_traffic.json, corrected as described there) times the launch size."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(workload)
        if entry is None or entry.get("kernel_variant") != variant:
            return None  # no PMC passes committed for this kernel on this workload
        return entry["bytes_per_product"] * products_per_launch
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(engine, ring, n, pool_t, lhs_idx, rhs_idx, out_t, seconds: float):
    """Times the CPU oracle (a port of the reference algorithm; the Rust crate cannot be built here) on a
    bounded sample of the same workload and checks the GPU's products against it bit for bit."""
    from oracle import gsw_oracle as O  # the checker / CPU baseline: the only use of the oracle in this file

    # every core this process may use: OMP_NUM_THREADS when the box sets it (its share of a shared machine),
    # else the scheduler's affinity mask
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or avail
    ring_o = O.fp() if ring.kind == _lib.RING_FP_M31 else O.zpow2(ring.log2q)
    og = O.GSW(ring_o, n, n * ring_o.ell)
    width = n + 1

    def fetch(t, idx):
        return np.ascontiguousarray(t[idx].cpu().numpy().view(np.uint32)[:, :, :width])

    # calibrate on one product per thread, then size the sample to about `seconds` of wall time
    first = min(threads, len(lhs_idx))
    c1, c2 = fetch(pool_t, lhs_idx[:first]), fetch(pool_t, rhs_idx[:first])
    t0 = time.perf_counter()
    got = og.mul_gadget_batch(c1, c2, threads=threads)
    dt = time.perf_counter() - t0
    verified = bool(np.array_equal(got, fetch(out_t, list(range(first)))))
    count, total = first, dt
    extra = int(min(len(lhs_idx) - first, max(0, (seconds - dt) / max(dt, 1e-6)) * first))
    extra -= extra % threads if extra >= threads else 0
    if extra > 0:
        c1, c2 = fetch(pool_t, lhs_idx[first:first + extra]), fetch(pool_t, rhs_idx[first:first + extra])
        t0 = time.perf_counter()
        got = og.mul_gadget_batch(c1, c2, threads=threads)
        total += time.perf_counter() - t0
        verified = verified and bool(np.array_equal(got, fetch(out_t, list(range(first, first + extra)))))
        count += extra
    # the reference's own algorithm (dense N^3 product, one thread: gsw_impl.rs:107) at its test size
    small = O.GSW(O.fp(), 10, 310)
    a = O.fill_uniform(O.fp(), 1, 1, 0, 1, 341 * 341).reshape(341, 341)
    b = O.fill_uniform(O.fp(), 1, 2, 0, O.FP_P - 1, 341 * 341).reshape(341, 341)
    t0 = time.perf_counter()
    small.mult(a, b)
    dense_s = time.perf_counter() - t0
    big_n = ring_o.big_n(n)
    return {
        "value": count / total, "unit": "mults/s", "cores": threads, "host_cores": os.cpu_count(),
        "cores_available_to_this_process": avail, "kind": "port",
        "sample": f"{count} products of the same pool (n={n}, N={big_n}) through the compact-form CPU port, "
                  f"OpenMP over the batch, {total:.1f} s",
        "gpu_matches_cpu_bit_exact": verified,
        "reference_dense_algorithm_mults_per_s_1core_extrapolated": 1.0 / (dense_s * (big_n / 341.0) ** 3),
    }


def spawn_ranks(n_ranks: int, rehearsal: bool) -> int:
    """`bench.py --gpus N` started bare (no torch.distributed.run around it): this process touches no GPU, starts one
    child per rank with the rendezvous environment set, lets them print (rank 0 prints the JSON line) and returns
    the first non-zero exit status, ending the other ranks when one fails."""
    import socket
    import subprocess

    visible = torch.cuda.device_count()  # counts devices without initialising one
    if visible < n_ranks and not rehearsal:
        print(f"bench.py: --gpus {n_ranks} but {visible} GPU(s) visible; one rank per GPU is the only mode "
              "(GSW_BENCH_REHEARSAL=1 puts every rank on GPU 0 over gloo to rehearse the control flow)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    children = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    status = 0
    pending = set(range(n_ranks))
    while pending:
        for r in sorted(pending):
            rc = children[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                print(f"bench.py: rank {r} exited with status {rc}; stopping the other ranks", file=sys.stderr)
                for o in pending:
                    children[o].terminate()  # the exact processes started above
        time.sleep(0.05)
    return status


def main():
    args = parse_args()
    # GSW_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo, to exercise the multi-rank control flow on a
    # one-GPU box (RCCL refuses two ranks on one device); its numbers are not a measurement
    rehearsal = os.environ.get("GSW_BENCH_REHEARSAL") == "1"
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: make the N ranks here, before anything in this process touches a GPU
        raise SystemExit(spawn_ranks(args.gpus, rehearsal))
    rank, local_rank, world = gdist.init("gloo" if rehearsal else None)
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pool_default = 0
    if args.config:
        args.workload, default_batch, pool_default = CONFIGS[args.config]
        make_ring, n, _ = WORKLOADS[args.workload]
    else:
        make_ring, n, default_batch = WORKLOADS[args.workload]
    ring = make_ring()
    batch = args.batch or default_batch
    if args.variant is not None:
        os.environ["GSW_GEMM_VARIANT"] = str(args.variant)
    gsw = tfhe_amd.GSW(ring, n, n * ring.ell, seed=0x6A5C0000 + rank, device=local_rank)
    eng, L = gsw.engine, _lib.lib()
    variant = L.gsw_ctx_gemm_variant(eng.handle)  # chosen by shape unless GSW_GEMM_VARIANT says otherwise
    # One explicit stream for the engine and for torch.  Torch's default stream has the handle NULL, which
    # the engine reads as "use your own stream": work on the two would then be unordered.
    stream = torch.cuda.Stream(dev)
    eng.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        result = run(args, gsw, ring, n, batch, pool_default, variant, rank, world, dev, rehearsal)
    if rank == 0:
        print(json.dumps(result), flush=True)
    gdist.barrier()
    gdist.shutdown()


def run(args, gsw, ring, n, batch, pool_default, variant, rank, world, dev, rehearsal):
    eng, L = gsw.engine, _lib.lib()
    big_n, ld, ct = eng.big_n, eng.ld, eng.compact_elems
    # room for two pipeline chunks of 128 prepared right operands (8.6 GB at n = 512), and never less than 256 MiB:
    # small shapes then run as one preparation and one contraction launch instead of a launch-bound pipeline
    _lib.check(L.gsw_ctx_set_workspace_limit(eng.handle, max(min(batch, 256) * eng.prepared_bytes, 256 << 20)))

    # resident pool of synthetic compact ciphertexts: i.i.d. uniform in [0, q), padding columns zero
    pool = args.pool or pool_default or min(2 * batch, 2048)
    pool -= pool % 2
    pool_t = torch.empty((pool, big_n, ld), dtype=torch.int32, device=dev)
    _lib.check(L.gsw_fill_uniform(eng.handle, 0x6A5C0000 + 1, rank, 0, ring.max_u64(), 1, pool_t.data_ptr(),
                                  pool * ct))
    pool_t[:, :, n + 1:] = 0
    if args.diag_pool == "zero":      # diagnostic only: how much of the kernel's time is clock (power), not cycles
        pool_t.zero_()
    elif args.diag_pool == "sparse":  # one bit set per element
        pool_t.copy_(torch.ones_like(pool_t) << (pool_t & 7))
        pool_t[:, :, n + 1:] = 0
    bad = C.c_uint64(0)
    _lib.check(L.gsw_validate_compact(eng.handle, pool_t.data_ptr(), ct, pool, C.byref(bad)))
    if bad.value:
        raise SystemExit(f"operand pool breaks the contraction's invariants in {bad.value} words")
    out_t = torch.empty((batch, big_n, ld), dtype=torch.int32, device=dev)
    digests = torch.zeros(batch, dtype=torch.int64, device=dev)
    lhs_idx = [(2 * b) % pool for b in range(batch)]
    rhs_idx = [(2 * b + 1) % pool for b in range(batch)]
    # entry b multiplies pool items (2b mod pool, 2b+1 mod pool): with pool = 2 * batch that is one
    # strided call; a smaller pool wraps around in `pool / 2`-sized calls
    per_call = min(batch, pool // 2)

    drain = None
    if args.gather == "full":
        drain = gdist.ResultDrain((big_n, ld), torch.int32, world, rank, batch, chunk=32, device=dev)
    gathered = {}

    def step():
        for b0 in range(0, batch, per_call):
            cur = min(per_call, batch - b0)
            _lib.check(L.gsw_mul_gadget_batch_digest(eng.handle, pool_t.data_ptr(), 2 * ct,
                                                     pool_t.data_ptr() + 4 * ct, 2 * ct,
                                                     out_t.data_ptr() + 4 * ct * b0, ct, cur, 0,
                                                     digests.data_ptr() + 8 * b0))
        # the exchange step follows the contraction on the same stream
        if drain is not None:
            gathered["full"] = drain.run(out_t)  # every product of every rank to rank 0's host memory
        elif world > 1:
            # digests of every product + one sampled product per rank to rank 0
            gathered["digests"] = gdist.gather_to_root(digests)
            gathered["sample"] = gdist.gather_to_root(out_t[0])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    gdist.barrier()
    eng.enable_timing(True)
    eng.reset_timing()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    gdist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = gdist.max_over_ranks(elapsed, device=dev)
    gemm_ms, gemm_launches = eng.kernel_time(0)
    prep_ms, prep_launches = eng.kernel_time(1)
    eng.enable_timing(False)

    # Untimed check of what the last step left behind: the fused digests equal the digest kernel's over the
    # stored products, and what rank 0 gathered is what the ranks hold (its own digests entry for entry; for
    # every rank, the digest of the gathered sample product equals that rank's gathered digest 0).
    check = torch.zeros(batch, dtype=torch.int64, device=dev)
    _lib.check(L.gsw_digest_batch(eng.handle, out_t.data_ptr(), ct, ct, batch, check.data_ptr()))
    exchange_ok = bool(torch.equal(check, digests))
    if rank == 0 and gathered.get("digests") is not None:
        got = gathered["digests"]
        exchange_ok = exchange_ok and bool(torch.equal(got[0].to(dev), digests))
        one = torch.zeros(1, dtype=torch.int64, device=dev)
        for r, sample in enumerate(gathered["sample"]):
            sample = sample.to(dev).contiguous()
            _lib.check(L.gsw_digest_batch(eng.handle, sample.data_ptr(), ct, ct, 1, one.data_ptr()))
            exchange_ok = exchange_ok and int(one.item()) == int(got[r][0].item())
    if rank == 0 and gathered.get("full") is not None:
        exchange_ok = exchange_ok and bool(torch.equal(gathered["full"][0].to(dev), out_t))
    torch.cuda.synchronize(dev)

    if rank != 0:
        return None
    total_mults = world * batch * args.steps
    mults_per_launch = batch * args.steps / max(gemm_launches, 1)
    gemm_avg_s = gemm_ms / 1e3 / max(gemm_launches, 1)
    achieved_tops = algorithmic_ops(big_n, n) * mults_per_launch / gemm_avg_s / 1e12 if gemm_avg_s > 0 else 0.0
    result = {
        "metric": "GSW ciphertext mults/sec (node)", "value": total_mults / elapsed, "unit": "mults/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int8",  # int8 MFMA, exact int32 accumulation, u32 ring elements
        "data": "synthetic" if args.diag_pool == "uniform" else f"DIAGNOSTIC pool pattern {args.diag_pool}: not the benchmark",
        "config": {"workload": f"{args.workload}: batch {batch} per GPU of independent GSW products, n={n}, "
                               f"ell={ring.ell}, N={big_n}, q={'2^31-1' if ring.kind == 0 else '2^%d' % ring.log2q}",
                   "benchmark_config": args.config or (2 if args.workload == "fp_n256" and batch == 1024 else None),
                   "batch_per_gpu": batch, "pool": pool, "gather": args.gather,
                   "exchange_verified": exchange_ok,
                   "world_size_seen": gdist.world_size_seen(), "backend": gdist.backend_name(),
                   "parallelism": f"batch-sharded x{world}" + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else "")},
        "roofline": {
            "bound": "mfma", "achieved": achieved_tops, "peak": INT8_DENSE_PEAK_TOPS, "unit": "TOP/s",
            "frac": achieved_tops / INT8_DENSE_PEAK_TOPS, "traffic": load_traffic(args.workload, variant, mults_per_launch),
            "algorithmic_bytes_per_launch": algorithmic_bytes(big_n, n) * mults_per_launch,
            "op_kind": "int8 MFMA multiply-accumulate (exact int32 accumulation, u32 ring elements) counted as "
                       "2 ops; algorithmic ops 2*N^2*(n+1)*4 per product",
            "kernel": KERNEL_NAMES.get(variant, f"diagnostic build {variant} of a -DGSW_DIAGNOSTIC_BUILDS library (wrong results by design)"),
            "kernel_variant": variant, "kernel_avg_ms": gemm_avg_s * 1e3, "launches": gemm_launches,
            "products_per_launch": mults_per_launch,
            "prepare_kernel_avg_ms": prep_ms / max(prep_launches, 1),
            "algorithmic_bytes_per_product": algorithmic_bytes(big_n, n),
            "hbm_gbs_algorithmic": algorithmic_bytes(big_n, n) * total_mults / world / elapsed / 1e9,
            "hbm_peak_gbs": HBM_PEAK_GBS,
        },
    }
    if not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline(eng, ring, n, pool_t, lhs_idx, rhs_idx, out_t, args.cpu_seconds)
    if not exchange_ok and os.environ.get("GSW_BENCH_DIAGNOSTIC") != "1":  # timing-only experiment builds compute nothing valid
        raise SystemExit("the digests or gathered results do not match the stored products: " + json.dumps(result))
    return result


if __name__ == "__main__":
    main()
