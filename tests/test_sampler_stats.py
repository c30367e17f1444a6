"""Statistical validation of the seeded samplers against the distributions error_sampling.rs draws from:
DiscrGaussianSampler (error_sampling.rs:32-71: discrete Gaussian, sigma = 1 by default, negative samples
mapped to -R), NaiveSampler (error_sampling.rs:73-84: uniform on [0, 65535]) and rnd_ring_elm
(error_sampling.rs:20-24: uniform on [min, max] inclusive).  The reference draws from an OS-seeded
generator, so its streams cannot be reproduced; moments, range, tails and a chi-square fit are what
can be pinned.  The CPU tests check the oracle's generator, the GPU test checks that the device
generator produces the identical stream (so the same statistics hold for it).
"""
import math

import numpy as np
import pytest

P = 2147483647
COUNT = 1_000_000


def _signed(x, q):
    x = x.astype(np.int64)
    return np.where(x > q // 2, x - q, x)


def _chi2(observed, expected):
    return float(((observed - expected) ** 2 / expected).sum())


def _check_gaussian(samples_signed, sigma):
    kmax = int(math.ceil(13.0 * sigma)) + 1
    assert np.abs(samples_signed).max() <= kmax
    ks = np.arange(-kmax, kmax + 1)
    pmf = np.exp(-ks.astype(np.float64) ** 2 / (2 * sigma * sigma))
    pmf /= pmf.sum()
    n = samples_signed.size
    keep = pmf * n >= 20  # cells with a usable expected count; the rest are pooled into one tail cell
    obs = np.array([(samples_signed == k).sum() for k in ks], dtype=np.float64)
    o = np.append(obs[keep], obs[~keep].sum())
    e = np.append(pmf[keep] * n, max(pmf[~keep].sum() * n, 1e-9))
    dof = len(o) - 1
    assert _chi2(o, e) < dof + 6 * math.sqrt(2 * dof), (_chi2(o, e), dof)
    mean, var = samples_signed.mean(), samples_signed.var()
    true_var = float((pmf * ks ** 2).sum())
    assert abs(mean) < 5 * sigma / math.sqrt(n)
    assert abs(var - true_var) < 5 * true_var * math.sqrt(2.0 / n)
    assert abs((samples_signed > 0).mean() - (samples_signed < 0).mean()) < 5 / math.sqrt(n)


@pytest.mark.parametrize("sigma", [1.0, 3.2])
def test_oracle_gaussian_matches_the_discrete_gaussian(oracle, sigma):
    for ring in (oracle.fp(), oracle.zpow2(32), oracle.zpow2(8)):
        x = oracle.sample_gaussian(ring, 41, 3, sigma, COUNT)
        assert x.max() < ring.q
        _check_gaussian(_signed(x, ring.q), sigma)
    # the mapping to the ring is the sign rule of error_sampling.rs:57-70: k >= 0 -> k, k < 0 -> -(|k|)
    raw = oracle.sample_gaussian_i32(41, 3, sigma, 1000)
    fp = oracle.sample_gaussian(oracle.fp(), 41, 3, sigma, 1000)
    assert np.array_equal(fp.astype(np.int64), np.where(raw < 0, P + raw.astype(np.int64), raw))


def test_oracle_naive_and_uniform_ranges(oracle):
    ring = oracle.fp()
    x = oracle.sample_naive(ring, 5, 1, COUNT).astype(np.int64)
    assert x.min() >= 0 and x.max() <= 65535 and x.min() < 16 and x.max() > 65520
    cells = np.bincount(x >> 12, minlength=16).astype(np.float64)
    assert _chi2(cells, np.full(16, COUNT / 16)) < 15 + 6 * math.sqrt(30)
    lo, hi = 5, P - 1
    u = oracle.fill_uniform(ring, 5, 2, lo, hi, COUNT).astype(np.int64)
    assert u.min() >= lo and u.max() <= hi
    cells = np.bincount(((u - lo) * 32 // (hi - lo + 1)), minlength=32).astype(np.float64)
    assert _chi2(cells, np.full(32, COUNT / 32)) < 31 + 6 * math.sqrt(62)
    # inclusive bounds (random_range(min..=max), error_sampling.rs:22-23)
    tiny = oracle.fill_uniform(ring, 5, 3, 7, 9, 3000)
    assert set(tiny.tolist()) == {7, 8, 9}
    bits = oracle.fill_uniform(ring, 5, 4, 0, 1, 100000)  # rnd_dmatrix(N, m, 0, 1) of gsw_impl.rs:30
    assert abs(bits.mean() - 0.5) < 0.01


@pytest.mark.gpu
def test_device_samplers_produce_the_oracle_streams(oracle):
    import tfhe_amd
    from tfhe_amd import _lib

    L = _lib.lib()  # raises if libgsw_hip.so is missing: there is no fallback
    for ring_o, ring_d in ((oracle.fp(), tfhe_amd.Fp()), (oracle.zpow2(32), tfhe_amd.Zpow2(32)),
                           (oracle.zpow2(8), tfhe_amd.Zpow2(8))):
        eng = tfhe_amd.Engine(ring_d, 2, 2 * ring_d.ell)
        buf = eng.alloc_u32(COUNT)
        for sigma in (1.0, 3.2):
            _lib.check(L.gsw_sample_gaussian(eng.handle, 41, 3, sigma, buf.ptr, COUNT))
            got = buf.download(np.uint32, COUNT)
            assert np.array_equal(got, oracle.sample_gaussian(ring_o, 41, 3, sigma, COUNT))
        _check_gaussian(_signed(got, ring_o.q), 3.2)
        _lib.check(L.gsw_sample_naive(eng.handle, 5, 1, buf.ptr, COUNT))
        assert np.array_equal(buf.download(np.uint32, COUNT), oracle.sample_naive(ring_o, 5, 1, COUNT))
        hi = min(ring_o.q - 1, P - 1)
        _lib.check(L.gsw_fill_uniform(eng.handle, 5, 2, 0, hi, 0, buf.ptr, COUNT))
        assert np.array_equal(buf.download(np.uint32, COUNT), oracle.fill_uniform(ring_o, 5, 2, 0, hi, COUNT))
