"""Host-side mirror of the reference's `FheScheme<R>` for `GSW<R, T>` over the C-ABI engine.

Same method names, argument meaning and error behaviour as src/gsw/mod.rs:12-26 and
src/gsw/gsw_impl.rs:12-125 of n0900/tfhe; the arithmetic runs in the HIP library
(`libgsw_hip.so`) and nowhere else -- there is no CPU fallback.

Ciphertexts live on the GPU in compact form Cc = BitDecomp^-1_linear(C) (N x (n+1), padded
leading dimension); `Ciphertext.to_dense()` gives the reference's N x N matrix
BitDecomp(Cc) (the flattened ciphertext).  The reference's own dense operations on
arbitrary N x N matrices are the `*_dense` methods.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import FLAG_NAND, GswError, check

FP_P = 2147483647  # src/field.rs:19


@dataclass(frozen=True)
class Ring:
    """`R: RingElement` (src/lib.rs:15-29): Fp (src/field.rs) or Zpow2<M> (src/pow2_ring.rs)."""

    kind: int
    log2q: int

    @property
    def ell(self) -> int:  # R::Num_Bits
        return 31 if self.kind == _lib.RING_FP_M31 else self.log2q

    @property
    def q(self) -> int:
        return FP_P if self.kind == _lib.RING_FP_M31 else 1 << self.log2q

    def max_u64(self) -> int:  # field.rs:36-38, pow2_ring.rs:31-33
        return self.q - 1


def Fp() -> Ring:
    return Ring(_lib.RING_FP_M31, 31)


def Zpow2(m: int) -> Ring:
    return Ring(_lib.RING_ZPOW2, m)


class DeviceBuffer:
    """A hipMalloc'ed block owned by an engine context."""

    def __init__(self, engine: "Engine", nbytes: int):
        self.engine = engine
        self.nbytes = int(nbytes)
        ptr = C.c_void_p()
        check(_lib.lib().gsw_malloc(engine.handle, self.nbytes, C.byref(ptr)))
        self.ptr = ptr.value

    def free(self):
        if self.ptr and self.engine.handle:
            _lib.lib().gsw_free(self.engine.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, arr: np.ndarray, offset: int = 0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        check(_lib.lib().gsw_memcpy_h2d(self.engine.handle, self.ptr + offset, arr.ctypes.data, arr.nbytes))

    def download(self, dtype, count: int, offset: int = 0) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        check(_lib.lib().gsw_memcpy_d2h(self.engine.handle, out.ctypes.data, self.ptr + offset, out.nbytes))
        return out


class Engine:
    """One `gsw_ctx`: ring, (n, m), GPU and stream."""

    def __init__(self, ring: Ring, n: int, m: int, device: int = 0):
        self.ring, self.n, self.m, self.device = ring, int(n), int(m), int(device)
        self.handle = None
        handle = C.c_void_p()
        check(_lib.lib().gsw_ctx_create(C.byref(handle), ring.kind, ring.log2q, self.n, self.m, self.device))
        self.handle = handle.value
        L = _lib.lib()
        self.ell = L.gsw_ctx_ell(self.handle)
        self.big_n = L.gsw_ctx_big_n(self.handle)
        self.ld = L.gsw_ctx_ld(self.handle)
        self.compact_elems = L.gsw_ctx_compact_elems(self.handle)
        self.prepared_bytes = L.gsw_ctx_prepared_bytes(self.handle)
        self.rbits_elems = L.gsw_ctx_rbits_elems(self.handle)
        self.pk_prepared_bytes = L.gsw_ctx_pk_prepared_bytes(self.handle)

    def close(self):
        if self.handle:
            _lib.lib().gsw_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(_lib.lib().gsw_sync(self.handle))

    def set_stream(self, hip_stream: Optional[int]):
        check(_lib.lib().gsw_ctx_set_stream(self.handle, hip_stream))

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def alloc_u32(self, count: int) -> DeviceBuffer:
        return DeviceBuffer(self, 4 * count)

    # ---- timing of the two hot kernels (HIP events on the stream they run on) ----
    def enable_timing(self, on: bool = True):
        check(_lib.lib().gsw_ctx_enable_timing(self.handle, int(on)))

    def reset_timing(self):
        check(_lib.lib().gsw_ctx_reset_timing(self.handle))

    def counters(self):
        """(right operands prepared, products contracted) since the context was created."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(_lib.lib().gsw_ctx_counters(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_time(self, which: int):
        ms, count = C.c_double(0), C.c_long(0)
        check(_lib.lib().gsw_ctx_kernel_time(self.handle, which, C.byref(ms), C.byref(count)))
        return ms.value, count.value


class Ciphertext:
    """`FheScheme::Ciphertext` held on the GPU in compact form (one or a batch of them)."""

    def __init__(self, engine: Engine, batch: int = 1, buf: Optional[DeviceBuffer] = None):
        self.engine = engine
        self.batch = int(batch)
        self.buf = buf if buf is not None else engine.alloc_u32(engine.compact_elems * self.batch)
        self._cache = None  # weak reference to the PreparedCache bound to this batch, if any

    def invalidate(self, slot: int = 0, count: Optional[int] = None) -> None:
        """Entries slot .. slot + count - 1 are about to change: a PreparedCache bound to this batch forgets their
        prepared forms.  Every method of this module that writes into an existing batch calls it; code that
        writes through `buf` or a raw pointer must call it itself."""
        cache = self._cache() if self._cache is not None else None
        if cache is not None:
            count = self.batch - slot if count is None else count
            cache.valid[slot:slot + count] = 0

    @property
    def ptr(self) -> int:
        return self.buf.ptr

    @property
    def stride(self) -> int:
        return self.engine.compact_elems

    def entry_ptr(self, b: int) -> int:
        return self.buf.ptr + 4 * self.stride * b

    @classmethod
    def from_compact(cls, engine: Engine, host) -> "Ciphertext":
        """Upload N x (n+1) (or batch x N x (n+1)) canonical uint32 values."""
        host = np.ascontiguousarray(host, dtype=np.uint32)
        if host.ndim == 2:
            host = host[None]
        if host.shape[1:] != (engine.big_n, engine.n + 1):
            raise ValueError(f"compact ciphertext must be {engine.big_n} x {engine.n + 1}, got {host.shape[1:]}")
        ct = cls(engine, host.shape[0])
        if ct.batch * engine.compact_elems * 4 <= cls._BULK_BYTES and engine.compact_elems == engine.big_n * engine.ld:
            # small entries: one copy of the padded image instead of a pitched copy per entry
            padded = np.zeros((ct.batch, engine.big_n, engine.ld), dtype=np.uint32)
            padded[:, :, : engine.n + 1] = host
            ct.buf.upload(padded)
            return ct
        for b in range(ct.batch):
            check(_lib.lib().gsw_upload_compact(engine.handle, ct.entry_ptr(b), host[b].ctypes.data))
        return ct

    _BULK_BYTES = 256 << 20

    def to_compact(self) -> np.ndarray:
        e = self.engine
        if self.batch > 1 and self.batch * e.compact_elems * 4 <= self._BULK_BYTES and e.compact_elems == e.big_n * e.ld:
            raw = self.buf.download(np.uint32, self.batch * e.compact_elems).reshape(self.batch, e.big_n, e.ld)
            return np.ascontiguousarray(raw[:, :, : e.n + 1])
        out = np.empty((self.batch, e.big_n, e.n + 1), dtype=np.uint32)
        for b in range(self.batch):
            check(_lib.lib().gsw_download_compact(e.handle, out[b].ctypes.data, self.entry_ptr(b)))
        return out[0] if self.batch == 1 else out

    def to_bytes(self) -> bytes:
        """The batch as one wire object (include/gsw_hip.h, "wire format")."""
        e, L = self.engine, _lib.lib()
        size = L.gsw_wire_bytes(e.handle, _lib.WIRE_CIPHERTEXT, self.batch)
        out = np.empty(size, dtype=np.uint8)
        check(L.gsw_wire_write_compact(e.handle, self.ptr, self.stride, self.batch, out.ctypes.data, size))
        return out.tobytes()

    @classmethod
    def from_bytes(cls, engine: Engine, data: bytes) -> "Ciphertext":
        """Validate a wire object against the engine's ring and (n, m) and upload its ciphertexts."""
        L = _lib.lib()
        raw = np.frombuffer(data, dtype=np.uint8)
        count = C.c_int(0)
        check(L.gsw_wire_read_compact(engine.handle, raw.ctypes.data, raw.size, None, 0, 0, C.byref(count)))
        ct = cls(engine, max(count.value, 1))
        ct.batch = count.value
        check(L.gsw_wire_read_compact(engine.handle, raw.ctypes.data, raw.size, ct.ptr, ct.stride, count.value,
                                      C.byref(count)))
        return ct

    def to_dense(self, b: int = 0) -> np.ndarray:
        """BitDecomp(Cc): the reference's N x N matrix form (helper.rs:17-21), computed on the GPU."""
        e = self.engine
        dense = e.alloc_u32(e.big_n * e.big_n)
        check(_lib.lib().gsw_bit_decomp(e.handle, self.entry_ptr(b), e.big_n, e.n + 1, e.ld, dense.ptr))
        return dense.download(np.uint32, e.big_n * e.big_n).reshape(e.big_n, e.big_n)


class PreparedCache:
    """Prepared right operands beside a pool of ciphertexts (gsw_eval_circuit_cached): entry s is the prepared form
    of pool slot s when valid[s] is set.  A cache belongs to one pool: it is bound to the first batch it is used
    with (or to `pool` here), every writer of this module into that batch clears the flags of the entries it
    changes, and using the cache with another batch is refused."""

    def __init__(self, engine: Engine, slots: int, pool: Optional["Ciphertext"] = None):
        self.engine, self.slots = engine, int(slots)
        self.buf = engine.alloc(engine.prepared_bytes * self.slots)
        self.valid = np.zeros(self.slots, dtype=np.uint8)
        self._pool = None
        if pool is not None:
            self.bind(pool)

    def bind(self, pool: "Ciphertext") -> None:
        import weakref

        bound = self._pool() if self._pool is not None else None
        if bound is pool:
            return
        if bound is not None:
            raise ValueError("this prepared-operand cache belongs to another pool")
        if self.slots != pool.batch:
            raise ValueError("the prepared-operand cache must have one entry per pool slot")
        other = pool._cache() if pool._cache is not None else None
        if other is not None and other is not self:
            raise ValueError("the pool already has a prepared-operand cache")
        self._pool = weakref.ref(pool)
        pool._cache = weakref.ref(self)

    def entry_ptr(self, slot: int) -> int:
        return self.buf.ptr + self.engine.prepared_bytes * slot


@dataclass
class SecretKey:
    """GswSk (src/gsw/sk.rs:13-17): t, s = (1, -t), v = PowersOf2(s)."""

    t: np.ndarray
    s: np.ndarray
    v: np.ndarray


@dataclass
class PublicKey:
    """GswPk (src/gsw/pk.rs:13-16) plus its device-side prepared form."""

    b: np.ndarray
    pk_matrix: np.ndarray
    prepared: Optional[DeviceBuffer] = None


class GSW:
    """`GSW<R, T>` (src/gsw/mod.rs:29-38) with `impl FheScheme<R>` (src/gsw/gsw_impl.rs:12-125).

    sampler: "gaussian" (DiscrGaussianSampler::default, sigma = 1), "naive" (NaiveSampler) or a
    float sigma.  All randomness comes from the engine's seeded counter-based device generator
    (the reference uses the OS-seeded thread rng; its streams cannot be reproduced).
    `check_max=False` lifts the reference's `assert!(max <= P)` (error_sampling.rs:21), without
    which `keygen` over Zpow2<32> cannot run.
    `use_flatten=True` stands for the reference's non-default cargo feature of that name (Cargo.toml:16-17): the dense
    operations then apply flatten_matrix to their result (gsw_impl.rs:91-92,99-100,108-109,120-121).
    """

    def __init__(self, ring: Ring, n: int, m: int, sampler="gaussian", seed: int = 0, device: int = 0,
                 check_max: bool = True, use_flatten: bool = False):
        self.use_flatten = bool(use_flatten)
        self.ring, self.n, self.m = ring, int(n), int(m)
        self.sampler, self.seed, self.check_max = sampler, int(seed), check_max
        self.engine = Engine(ring, n, m, device)
        self._draw = 0

    # ---- helpers ----
    def _stream(self) -> int:
        self._draw += 1
        return self._draw

    def _uniform(self, lo: int, hi: int, count: int) -> DeviceBuffer:
        buf = self.engine.alloc_u32(count)
        check(_lib.lib().gsw_fill_uniform(self.engine.handle, self.seed, self._stream(), lo, hi,
                                          int(not self.check_max), buf.ptr, count))
        return buf

    def _errors(self, count: int) -> DeviceBuffer:
        buf = self.engine.alloc_u32(count)
        L, h = _lib.lib(), self.engine.handle
        if self.sampler == "naive":
            check(L.gsw_sample_naive(h, self.seed, self._stream(), buf.ptr, count))
        else:
            sigma = 1.0 if self.sampler == "gaussian" else float(self.sampler)
            check(L.gsw_sample_gaussian(h, self.seed, self._stream(), sigma, buf.ptr, count))
        return buf

    def _neg(self, x: np.ndarray) -> np.ndarray:
        q = self.ring.q
        return ((q - x.astype(np.uint64)) % q).astype(np.uint32)

    def make_sk(self, t) -> SecretKey:
        """GswSk::new (src/gsw/sk.rs:20-28).  Key assembly is host logic on n + 1 values."""
        t = np.ascontiguousarray(t, dtype=np.uint32)
        q, ell = self.ring.q, self.ring.ell
        s = np.concatenate([np.array([1 % q], dtype=np.uint32), self._neg(t)])
        g = np.array([(1 << k) % q for k in range(ell)], dtype=np.uint64)  # gsw/mod.rs:58-62
        v = ((s.astype(np.uint64)[:, None] * g[None, :]) % q).astype(np.uint32).reshape(-1)  # helper.rs:65-77
        return SecretKey(t, s, v)

    def make_pk(self, rand_b, e, t) -> PublicKey:
        """GswPk::new (src/gsw/pk.rs:23-39): b = B t + e on the GPU, pk_matrix = [b | B]."""
        rand_b = np.ascontiguousarray(rand_b, dtype=np.uint32)
        e = np.ascontiguousarray(e, dtype=np.uint32)
        t = np.ascontiguousarray(t, dtype=np.uint32)
        if rand_b.shape[1] != t.size:
            raise ValueError(f"Dimension Mismatch ncols {rand_b.shape[1]}, {t.size}")  # pk.rs:24
        if rand_b.shape[0] != e.size:
            raise ValueError(f"Dimension Mismatch nrows {rand_b.shape[0]}, {e.size}")  # pk.rs:25
        m, n = rand_b.shape
        eng, L = self.engine, _lib.lib()
        d_b, d_e, d_bt = eng.alloc_u32(m * n), eng.alloc_u32(m), eng.alloc_u32(m)
        d_b.upload(rand_b)
        d_e.upload(e)
        check(L.gsw_matvec(eng.handle, d_b.ptr, m, n, n, t.ctypes.data, d_bt.ptr))
        check(L.gsw_add(eng.handle, d_bt.ptr, d_e.ptr, d_bt.ptr, m))
        b = d_bt.download(np.uint32, m)
        pk_matrix = np.concatenate([b[:, None], rand_b], axis=1)
        pk = PublicKey(b, pk_matrix)
        if (m, n) == (self.m, self.n):
            pk.prepared = eng.alloc(eng.pk_prepared_bytes)
            check(L.gsw_prepare_pk(eng.handle, np.ascontiguousarray(pk_matrix).ctypes.data, pk.prepared.ptr))
        return pk

    # ---- keys on the wire ----
    def _host_object(self, kind: int, values: np.ndarray) -> bytes:
        L, h = _lib.lib(), self.engine.handle
        size = L.gsw_wire_bytes(h, kind, 1)
        out = np.empty(size, dtype=np.uint8)
        values = np.ascontiguousarray(values, dtype=np.uint32)
        check(L.gsw_wire_write_host(h, kind, values.ctypes.data, out.ctypes.data, size))
        return out.tobytes()

    def pk_to_bytes(self, pk: PublicKey) -> bytes:
        if pk.pk_matrix.shape != (self.m, self.n + 1):
            raise ValueError("public key was built for different (n, m)")
        return self._host_object(_lib.WIRE_PUBLIC_KEY, pk.pk_matrix)

    def sk_to_bytes(self, sk: SecretKey) -> bytes:
        return self._host_object(_lib.WIRE_SECRET_KEY, sk.t)

    def pk_from_bytes(self, data: bytes) -> PublicKey:
        eng, L = self.engine, _lib.lib()
        raw = np.frombuffer(data, dtype=np.uint8)
        mat = np.empty((self.m, self.n + 1), dtype=np.uint32)
        check(L.gsw_wire_read_host(eng.handle, _lib.WIRE_PUBLIC_KEY, raw.ctypes.data, raw.size, mat.ctypes.data))
        pk = PublicKey(mat[:, 0].copy(), mat)
        pk.prepared = eng.alloc(eng.pk_prepared_bytes)
        check(L.gsw_prepare_pk(eng.handle, mat.ctypes.data, pk.prepared.ptr))
        return pk

    def sk_from_bytes(self, data: bytes) -> SecretKey:
        raw = np.frombuffer(data, dtype=np.uint8)
        t = np.empty(self.n, dtype=np.uint32)
        check(_lib.lib().gsw_wire_read_host(self.engine.handle, _lib.WIRE_SECRET_KEY, raw.ctypes.data, raw.size,
                                            t.ctypes.data))
        return self.make_sk(t)

    # ---- FheScheme ----
    def keygen(self, download: bool = True):
        """src/gsw/gsw_impl.rs:17-25, on the device (gsw_keygen_device): only the secret t crosses to the host.
        download=False skips fetching pk_matrix for the host-side PublicKey (it stays usable for encryption)."""
        eng, L = self.engine, _lib.lib()
        t = np.empty(self.n, dtype=np.uint32)
        prepared = eng.alloc(eng.pk_prepared_bytes)
        d_pk = eng.alloc_u32(self.m * eng.ld) if download else None
        naive = self.sampler == "naive"
        sigma = 1.0 if self.sampler in ("naive", "gaussian") else float(self.sampler)
        first = self._draw + 1
        self._draw += 3  # t, e, B
        try:
            check(L.gsw_keygen_device(eng.handle, self.seed, first, _lib.SAMPLER_NAIVE if naive else _lib.SAMPLER_GAUSSIAN,
                                      sigma, int(not self.check_max), t.ctypes.data, d_pk.ptr if download else None,
                                      prepared.ptr))
        except GswError as exc:
            if exc.code == _lib.E_INVALID and "max <= P" in exc.message:
                raise AssertionError(exc.message) from exc  # the reference panics here (error_sampling.rs:21)
            raise
        sk = self.make_sk(t)
        if download:
            mat = d_pk.download(np.uint32, self.m * eng.ld).reshape(self.m, eng.ld)[:, :self.n + 1].copy()
            pk = PublicKey(mat[:, 0].copy(), mat, prepared)
        else:
            pk = PublicKey(None, None, prepared)
        return sk, pk

    def keygen_host_assembled(self):
        """The same key generation with the samples drawn on the device but assembled through the host
        (GswPk::new as make_pk does it): a cross-check of keygen, draw for draw."""
        mx = self.ring.max_u64()
        try:
            t = self._uniform(0, mx, self.n).download(np.uint32, self.n)
            err = self._errors(self.m).download(np.uint32, self.m)
            rand_b = self._uniform(0, mx, self.m * self.n).download(np.uint32, self.m * self.n)
        except GswError as exc:
            if exc.code == _lib.E_INVALID and "max <= P" in exc.message:
                raise AssertionError(exc.message) from exc  # the reference panics here (error_sampling.rs:21)
            raise
        sk = self.make_sk(t)
        return sk, self.make_pk(rand_b.reshape(self.m, self.n), err, t)

    def encrypt_batch(self, pk: PublicKey, messages: Sequence[int], rbits=None, out: Optional[Ciphertext] = None,
                      slot: int = 0, cache: Optional[PreparedCache] = None) -> Ciphertext:
        """Batched src/gsw/gsw_impl.rs:27-42 in compact form: R * pk_matrix + message * G.

        rbits: optional (batch, N, m) 0/1 bytes standing for rnd_dmatrix(N, m, 0, 1); when absent the
        seeded device generator draws them.  out / slot: write the ciphertexts into entries slot, slot + 1, ..
        of an existing batch (a pool for eval_circuit) instead of a new one.  cache: also leave the prepared
        form of every fresh ciphertext in the cache entries of those slots (gsw_encrypt_batch_prepared).
        """
        eng, L = self.engine, _lib.lib()
        if pk.prepared is None:
            raise ValueError("public key was built for different (n, m)")
        msgs = np.ascontiguousarray(messages, dtype=np.uint32)
        batch = msgs.size
        d_m = eng.alloc_u32(batch)
        d_m.upload(msgs)
        if out is None:
            ct, dst = Ciphertext(eng, batch), None
        else:
            if slot < 0 or slot + batch > out.batch:
                raise ValueError(f"entries {slot}..{slot + batch - 1} do not fit a batch of {out.batch}")
            ct, dst = out, out.entry_ptr(slot)
        target = ct.ptr if dst is None else dst
        if cache is not None:
            cache.bind(ct)  # one cache per pool; refuses a cache that belongs to another batch
        ct.invalidate(slot, batch)  # whatever cache the batch has: these entries change (flags set again on success)
        prepared_out = None if cache is None else cache.entry_ptr(slot)
        if rbits is None:
            # the bits are drawn inside the call, chunk by chunk beside the contraction: no N x m matrix per entry
            check(L.gsw_encrypt_batch_seeded(eng.handle, pk.prepared.ptr, self.seed, self._stream(), d_m.ptr, target,
                                             ct.stride, prepared_out, batch))
        else:
            d_r = eng.alloc_u32(eng.rbits_elems * batch)
            rbits = np.ascontiguousarray(rbits, dtype=np.uint8).reshape(batch, eng.big_n, self.m)
            for b in range(batch):
                check(L.gsw_pack_rbits(eng.handle, rbits[b].ctypes.data, d_r.ptr + 4 * eng.rbits_elems * b))
            if cache is None:
                check(L.gsw_encrypt_batch(eng.handle, pk.prepared.ptr, d_r.ptr, eng.rbits_elems, d_m.ptr, target,
                                          ct.stride, batch))
            else:
                check(L.gsw_encrypt_batch_prepared(eng.handle, pk.prepared.ptr, d_r.ptr, eng.rbits_elems, d_m.ptr,
                                                   target, ct.stride, prepared_out, batch))
        if cache is not None:
            cache.valid[slot:slot + batch] = 1
        eng.sync()
        return ct

    def encrypt(self, pk: PublicKey, message: int, rbits=None) -> Ciphertext:
        """src/gsw/gsw_impl.rs:27-42."""
        return self.encrypt_batch(pk, [message], None if rbits is None else np.asarray(rbits)[None])

    def decrypt_batch(self, sk: SecretKey, ct: Ciphertext, raw: bool = False, slot: int = 0,
                      count: Optional[int] = None):
        """Batched src/gsw/gsw_impl.rs:48-52; slot / count select a range of entries (default: all)."""
        eng = self.engine
        count = ct.batch - slot if count is None else count
        if slot < 0 or count < 0 or slot + count > ct.batch:
            raise ValueError(f"entries {slot}..{slot + count - 1} are outside a batch of {ct.batch}")
        bits = np.empty(count, dtype=np.uint32)
        rawv = np.empty(count, dtype=np.uint32)
        s = np.ascontiguousarray(sk.s, dtype=np.uint32)
        check(_lib.lib().gsw_decrypt_batch(eng.handle, s.ctypes.data, ct.entry_ptr(slot), ct.stride, count,
                                           bits.ctypes.data, rawv.ctypes.data))
        return (bits, rawv) if raw else bits

    def decrypt(self, sk: SecretKey, ct: Ciphertext) -> int:
        """src/gsw/gsw_impl.rs:48-52."""
        return int(self.decrypt_batch(sk, ct)[0])

    def share_secret(self, sk: SecretKey, parties: int, seed: int = 0) -> np.ndarray:
        """Additive shares of sk.s (what `setup` of src/tfhe_gsw_fp.rs:14-18 hands to a sharing scheme):
        parties x (n + 1) values whose column sums are s modulo q."""
        q = self.ring.q
        rng = np.random.default_rng(seed)
        shares = rng.integers(0, q, size=(parties, self.n + 1), dtype=np.uint64)
        shares[-1] = (sk.s.astype(np.uint64) + (q - shares[:-1].sum(axis=0) % q)) % q
        return shares.astype(np.uint32)

    def part_dec(self, share: np.ndarray, ct: Ciphertext) -> np.ndarray:
        """One party's partial decryption (part_dec of src/tfhe_gsw_fp.rs:27-29, a todo!() there)."""
        share = np.ascontiguousarray(share, dtype=np.uint32)
        if share.size != self.n + 1:
            raise ValueError(f"a share of sk.s has n + 1 = {self.n + 1} entries, got {share.size}")
        out = np.empty(ct.batch, dtype=np.uint32)
        check(_lib.lib().gsw_partial_decrypt_batch(self.engine.handle, share.ctypes.data, ct.ptr, ct.stride, ct.batch,
                                                   out.ctypes.data))
        return out

    def combine_partials(self, partials) -> np.ndarray:
        partials = np.ascontiguousarray(partials, dtype=np.uint32)
        parties, batch = partials.shape
        out = np.empty(batch, dtype=np.uint32)
        check(_lib.lib().gsw_combine_partials(self.engine.handle, partials.ctypes.data, parties, batch, out.ctypes.data))
        return out

    def mp_decrypt_batch(self, sk: SecretKey, ct: Ciphertext) -> np.ndarray:
        eng = self.engine
        out = np.empty(ct.batch, dtype=np.uint32)
        s = np.ascontiguousarray(sk.s, dtype=np.uint32)
        try:
            check(_lib.lib().gsw_mp_decrypt_batch(eng.handle, s.ctypes.data, ct.ptr, ct.stride, ct.batch,
                                                  out.ctypes.data))
        except GswError as exc:
            if exc.code == _lib.E_UNSUPPORTED:
                raise RuntimeError(exc.message) from exc  # "Only supported for pow2 rings!" (gsw_impl.rs:69)
            raise
        return out

    def mp_decrypt(self, sk: SecretKey, ct: Ciphertext) -> int:
        """src/gsw/gsw_impl.rs:67-84."""
        return int(self.mp_decrypt_batch(sk, ct)[0])

    def _same_size(self, c1: Ciphertext, c2: Ciphertext):
        if c1.batch != c2.batch or c1.engine.compact_elems != c2.engine.compact_elems:
            # the reference asserts equal nrows/ncols (gsw_impl.rs:88-89,105-106)
            raise ValueError("Cannot add Ciphertexts because they are different sizes")

    def add(self, c1: Ciphertext, c2: Ciphertext) -> Ciphertext:
        """src/gsw/gsw_impl.rs:87-94 (compact form is linear, so this is the same sum)."""
        self._same_size(c1, c2)
        out = Ciphertext(self.engine, c1.batch)
        check(_lib.lib().gsw_add(self.engine.handle, c1.ptr, c2.ptr, out.ptr, c1.stride * c1.batch))
        return out

    def mult_const(self, ct: Ciphertext, constant: int) -> None:
        """src/gsw/gsw_impl.rs:97-101: in place, like the reference."""
        ct.invalidate()
        check(_lib.lib().gsw_mult_const(self.engine.handle, ct.ptr, int(constant), ct.stride * ct.batch))

    def _product(self, c1: Ciphertext, c2: Ciphertext, flags: int) -> Ciphertext:
        self._same_size(c1, c2)
        out = Ciphertext(self.engine, c1.batch)
        check(_lib.lib().gsw_mul_gadget_batch(self.engine.handle, c1.ptr, c1.stride, c2.ptr, c2.stride, out.ptr,
                                              out.stride, c1.batch, flags))
        return out

    def mult(self, c1: Ciphertext, c2: Ciphertext) -> Ciphertext:
        """src/gsw/gsw_impl.rs:104-111 on flattened ciphertexts: Pc = G^-1(C1c) * C2c."""
        return self._product(c1, c2, 0)

    def nand(self, c1: Ciphertext, c2: Ciphertext) -> Ciphertext:
        """src/gsw/gsw_impl.rs:114-123: I - C1 * C2, i.e. G - Pc in compact form."""
        return self._product(c1, c2, FLAG_NAND)

    GATE_OPS = {"mult": 0, "nand": 1, "add": 2, "mult_const": 3}

    @classmethod
    def encode_gates(cls, gates) -> np.ndarray:
        """(op, a, b, out) tuples -> the (len, 4) uint32 array that is the C ABI's gsw_gate list; a list that is
        evaluated many times is encoded once."""
        if isinstance(gates, np.ndarray):
            if gates.dtype != np.uint32 or gates.ndim != 2 or gates.shape[1] != 4:
                raise ValueError("an encoded gate list is a (len, 4) uint32 array")
            return np.ascontiguousarray(gates)
        gates = list(gates)
        arr = np.empty((len(gates), 4), dtype=np.uint32)
        for i, (op, ga, gb, out) in enumerate(gates):
            if op not in cls.GATE_OPS:
                raise ValueError(f"unknown gate op {op!r}")
            arr[i] = (cls.GATE_OPS[op], ga, gb, out)
        return arr

    def eval_circuit(self, pool: Ciphertext, gates, cache: Optional[PreparedCache] = None) -> None:
        """Apply (op, a, b, out) gates over the slots of `pool` (a batch of resident ciphertexts), in list
        order: op is "mult" / "nand" / "add" / "mult_const" (src/gsw/gsw_impl.rs:87-123; for mult_const
        b is the constant).  `gates` is a sequence of such tuples or the array encode_gates made of one.
        Independent consecutive products run as one batched contraction.  With a PreparedCache every slot is
        put into prepared form at most once until it is overwritten."""
        arr = self.encode_gates(gates)
        gate_ptr = arr.ctypes.data_as(C.POINTER(_lib.Gate))
        if cache is None:
            pool.invalidate()  # gates write slots behind the back of a cache the pool may have
            check(_lib.lib().gsw_eval_circuit(self.engine.handle, pool.ptr, pool.stride, pool.batch, gate_ptr, len(arr)))
            return
        cache.bind(pool)
        check(_lib.lib().gsw_eval_circuit_cached(self.engine.handle, pool.ptr, pool.stride, pool.batch, cache.buf.ptr,
                                                 cache.valid.ctypes.data, gate_ptr, len(arr)))

    # ---- the reference's dense operations on arbitrary matrices ----
    def _dense_binary(self, fn, a, b) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        if a.shape != b.shape or a.ndim != 2 or a.shape[0] != a.shape[1]:
            raise ValueError("Cannot add Ciphertexts because they are different sizes")  # sic: gsw_impl.rs:105
        eng = self.engine
        dim = a.shape[0]
        d_a, d_b, d_o = eng.alloc_u32(a.size), eng.alloc_u32(a.size), eng.alloc_u32(a.size)
        d_a.upload(a)
        d_b.upload(b)
        check(fn(eng.handle, d_a.ptr, d_b.ptr, d_o.ptr, dim))
        if self.use_flatten:  # gsw_impl.rs:108-109,120-121
            check(_lib.lib().gsw_flatten(eng.handle, d_o.ptr, dim, dim // eng.ell))
        return d_o.download(np.uint32, a.size).reshape(dim, dim)

    def set_dense_fast(self, mode) -> None:
        """MFMA path of the dense products (see mult_dense): 0 / False never, 1 / True from 1024 rows up (the
        default), 2 whenever the left operand has the form."""
        check(_lib.lib().gsw_ctx_set_dense_fast(self.engine.handle, int(mode)))

    def last_dense_path(self) -> int:
        """Which kernel the last dense product used: 0 = the plain N^3 kernel, 1 = the binary x digit MFMA contraction
        (flattened left operand), 2 = the digit x digit MFMA products (arbitrary entries)."""
        return int(_lib.lib().gsw_ctx_last_dense_path(self.engine.handle))

    def last_combine(self) -> int:
        """How the last contraction launch recombined its limb sums: 1 = 32-bit form (branch-free epilogue on interior
        tiles), 0 = the general epilogue with the 64-bit form on every tile, -1 = no launch yet."""
        return int(_lib.lib().gsw_ctx_last_combine(self.engine.handle))

    def mult_dense(self, c1, c2) -> np.ndarray:
        """src/gsw/gsw_impl.rs:104-111 (default features): the N x N x N product on the reference's own
        matrices.  A left operand in the reference's ciphertext format (0 / 1 off the diagonal) runs on the binary x
        digit MFMA contraction, any other on sixteen digit x digit MFMA products, small matrices on the plain
        kernel; the results are the same."""
        return self._dense_binary(_lib.lib().gsw_mul_dense, c1, c2)

    def nand_dense(self, c1, c2) -> np.ndarray:
        """src/gsw/gsw_impl.rs:114-123."""
        return self._dense_binary(_lib.lib().gsw_nand_dense, c1, c2)

    def add_dense(self, c1, c2) -> np.ndarray:
        """src/gsw/gsw_impl.rs:87-94."""
        a = np.ascontiguousarray(c1, dtype=np.uint32)
        b = np.ascontiguousarray(c2, dtype=np.uint32)
        if a.shape != b.shape:
            raise ValueError("Cannot add Ciphertexts because they are different sizes")
        eng = self.engine
        d_a, d_b = eng.alloc_u32(a.size), eng.alloc_u32(a.size)
        d_a.upload(a)
        d_b.upload(b)
        check(_lib.lib().gsw_add(eng.handle, d_a.ptr, d_b.ptr, d_a.ptr, a.size))
        if self.use_flatten:  # gsw_impl.rs:91-92
            check(_lib.lib().gsw_flatten(eng.handle, d_a.ptr, a.shape[0], a.shape[1] // eng.ell))
        return d_a.download(np.uint32, a.size).reshape(a.shape)

    def mult_const_dense(self, c, constant: int) -> np.ndarray:
        """src/gsw/gsw_impl.rs:97-101."""
        a = np.ascontiguousarray(c, dtype=np.uint32)
        d_a = self.engine.alloc_u32(a.size)
        d_a.upload(a)
        check(_lib.lib().gsw_mult_const(self.engine.handle, d_a.ptr, int(constant), a.size))
        if self.use_flatten:  # gsw_impl.rs:99-100
            check(_lib.lib().gsw_flatten(self.engine.handle, d_a.ptr, a.shape[0], a.shape[1] // self.engine.ell))
        return d_a.download(np.uint32, a.size).reshape(a.shape)

    def decrypt_dense(self, sk: SecretKey, c) -> int:
        """src/gsw/gsw_impl.rs:48-52 on the reference's N x N matrix."""
        a = np.ascontiguousarray(c, dtype=np.uint32)
        eng = self.engine
        if a.shape != (eng.big_n, eng.big_n):
            raise ValueError(f"dense ciphertext must be {eng.big_n} x {eng.big_n}")
        d_a = eng.alloc_u32(a.size)
        d_a.upload(a)
        bit = C.c_uint32(0)
        v = np.ascontiguousarray(sk.v, dtype=np.uint32)
        check(_lib.lib().gsw_decrypt_dense(eng.handle, v.ctypes.data, d_a.ptr, C.byref(bit)))
        return int(bit.value)

    # ---- gadget helpers (src/gsw/helper.rs) on host arrays, computed on the GPU ----
    def bit_decomp_matrix(self, a) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint32)
        rows, cols = a.shape
        eng = self.engine
        d_in, d_out = eng.alloc_u32(a.size), eng.alloc_u32(a.size * eng.ell)
        d_in.upload(a)
        check(_lib.lib().gsw_bit_decomp(eng.handle, d_in.ptr, rows, cols, cols, d_out.ptr))
        return d_out.download(np.uint32, a.size * eng.ell).reshape(rows, cols * eng.ell)

    def bit_decomp_as(self, a, fmt: int, ld_in: Optional[int] = None) -> np.ndarray:
        """bit_decomp_matrix in another storage format: _lib.DECOMP_U8 gives one byte per bit, _lib.DECOMP_BITS the
        row's bits as a little-endian bit string packed 32 per uint32 (rows of ceil(cols * ell / 32) words)."""
        a = np.ascontiguousarray(a, dtype=np.uint32)
        rows, cols = a.shape
        eng, L = self.engine, _lib.lib()
        ld_in = cols if ld_in is None else ld_in
        padded = np.zeros((rows, ld_in), dtype=np.uint32)
        padded[:, :cols] = a
        per_row = L.gsw_bit_decomp_row_elems(eng.handle, cols, fmt)
        dtype = np.uint8 if fmt == _lib.DECOMP_U8 else np.uint32
        d_in = eng.alloc_u32(padded.size)
        d_out = eng.alloc(max(rows * per_row * np.dtype(dtype).itemsize, 16))
        d_in.upload(padded)
        check(L.gsw_bit_decomp_as(eng.handle, d_in.ptr, rows, cols, ld_in, d_out.ptr, fmt))
        return d_out.download(dtype, rows * per_row).reshape(rows, per_row)

    def bit_decomp_inv_matrix(self, a, quirk: bool = True) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint32)
        rows, wide = a.shape
        eng = self.engine
        if wide % eng.ell:
            raise ValueError("column count must be a multiple of Num_Bits")
        cols = wide // eng.ell
        d_in, d_out = eng.alloc_u32(a.size), eng.alloc_u32(rows * cols)
        d_in.upload(a)
        check(_lib.lib().gsw_bit_decomp_inv(eng.handle, d_in.ptr, rows, cols, d_out.ptr, cols, int(quirk)))
        return d_out.download(np.uint32, rows * cols).reshape(rows, cols)

    def flatten_matrix(self, a) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint32)
        rows, wide = a.shape
        eng = self.engine
        d = eng.alloc_u32(a.size)
        d.upload(a)
        check(_lib.lib().gsw_flatten(eng.handle, d.ptr, rows, wide // eng.ell))
        return d.download(np.uint32, a.size).reshape(rows, wide)
