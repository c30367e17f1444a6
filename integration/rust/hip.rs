//! `src/gsw/hip.rs` -- `FheScheme<R>` (src/gsw/mod.rs:12-26) implemented on the MI355X engine through
//! `include/gsw_hip.h`.  Drop this file into the crate, add `pub mod hip;` to src/gsw/mod.rs:7-10 and the
//! `build.rs` next to it; `GswHip<R, T>` then stands wherever `GSW<R, T>` does (same associated types,
//! same panics).  Not compiled in the engine's own build environment (no Rust toolchain there); the same
//! call sequences are exercised by the C++ and Python mirrors of this binding.
//!
//! Element convention: every `R` crosses the boundary as its canonical value (`Into<u64>`,
//! field.rs:52-56 / pow2_ring.rs:148-152) in a `u32`; dense matrices cross ROW-major (nalgebra is
//! column-major, field.rs:107-112).
use std::ffi::CStr;
use std::marker::PhantomData;
use std::os::raw::{c_char, c_int, c_uint, c_void};

use nalgebra::{DMatrix, DVector};
use rand::Rng;

use crate::{
    error_sampling::{rnd_dmatrix, rnd_dvec, ErrorSampling},
    gsw::{pk::GswPk, sk::GswSk, FheScheme},
    RingElement,
};

#[repr(C)]
pub struct gsw_ctx {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gsw_gate {
    pub op: u32,
    pub a: u32,
    pub b: u32,
    pub out: u32,
}
pub const GSW_GATE_MULT: u32 = 0;
pub const GSW_GATE_NAND: u32 = 1;
pub const GSW_GATE_ADD: u32 = 2;
pub const GSW_GATE_MULT_CONST: u32 = 3;
pub const GSW_FLAG_NAND: c_uint = 1;

#[rustfmt::skip]
extern "C" {
    fn gsw_ctx_create(out: *mut *mut gsw_ctx, ring_kind: c_int, log2q: c_int, n: c_int, m: c_int, device: c_int) -> c_int;
    fn gsw_ctx_destroy(ctx: *mut gsw_ctx);
    fn gsw_last_error() -> *const c_char;
    fn gsw_sync(ctx: *mut gsw_ctx) -> c_int;
    fn gsw_ctx_big_n(ctx: *const gsw_ctx) -> c_int;
    fn gsw_ctx_ld(ctx: *const gsw_ctx) -> c_int;
    fn gsw_ctx_compact_elems(ctx: *const gsw_ctx) -> usize;
    fn gsw_ctx_pk_prepared_bytes(ctx: *const gsw_ctx) -> usize;
    fn gsw_malloc(ctx: *mut gsw_ctx, bytes: usize, d_ptr: *mut *mut c_void) -> c_int;
    fn gsw_free(ctx: *mut gsw_ctx, d_ptr: *mut c_void) -> c_int;
    fn gsw_memcpy_h2d(ctx: *mut gsw_ctx, d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> c_int;
    fn gsw_memcpy_d2h(ctx: *mut gsw_ctx, h_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    fn gsw_memset_zero(ctx: *mut gsw_ctx, d_dst: *mut c_void, bytes: usize) -> c_int;
    fn gsw_bit_decomp(ctx: *mut gsw_ctx, d_in: *const u32, rows: c_int, cols: c_int, ld_in: c_int, d_out: *mut u32) -> c_int;
    fn gsw_bit_decomp_inv(ctx: *mut gsw_ctx, d_in: *const u32, rows: c_int, cols: c_int, d_out: *mut u32, ld_out: c_int, quirk: c_int) -> c_int;
    fn gsw_mul_dense(ctx: *mut gsw_ctx, c1: *const u32, c2: *const u32, out: *mut u32, dim: c_int) -> c_int;
    fn gsw_nand_dense(ctx: *mut gsw_ctx, c1: *const u32, c2: *const u32, out: *mut u32, dim: c_int) -> c_int;
    fn gsw_mul_gadget_batch(ctx: *mut gsw_ctx, c1: *const u32, s1: usize, c2: *const u32, s2: usize, out: *mut u32, so: usize, batch: c_int, flags: c_uint) -> c_int;
    fn gsw_eval_circuit(ctx: *mut gsw_ctx, d_pool: *mut u32, slot_stride: usize, n_slots: u32, gates: *const gsw_gate, n_gates: usize) -> c_int;
    fn gsw_prepare_pk(ctx: *mut gsw_ctx, h_pk: *const u32, d_prepared: *mut c_void) -> c_int;
    fn gsw_encrypt_batch_seeded(ctx: *mut gsw_ctx, pk: *const c_void, seed: u64, stream: u64, msgs: *const u32, out: *mut u32, stride_out: usize, d_prepared_out: *mut c_void, batch: c_int) -> c_int;
    fn gsw_decrypt_dense(ctx: *mut gsw_ctx, h_v: *const u32, d_c: *const u32, bit: *mut u32) -> c_int;
    fn gsw_mp_decrypt_batch(ctx: *mut gsw_ctx, h_s: *const u32, d_cc: *const u32, stride: usize, batch: c_int, out: *mut u32) -> c_int;
    fn gsw_matvec(ctx: *mut gsw_ctx, d_mat: *const u32, rows: c_int, cols: c_int, ld: c_int, h_vec: *const u32, d_out: *mut u32) -> c_int;
    fn gsw_add(ctx: *mut gsw_ctx, a: *const u32, b: *const u32, out: *mut u32, count: usize) -> c_int;
    fn gsw_ctx_set_dense_fast(ctx: *mut gsw_ctx, enable: c_int) -> c_int;
    fn gsw_ctx_last_dense_path(ctx: *const gsw_ctx) -> c_int;
    fn gsw_ctx_last_combine(ctx: *const gsw_ctx) -> c_int;
    fn gsw_flatten(ctx: *mut gsw_ctx, d_mat: *mut u32, rows: c_int, cols: c_int) -> c_int;
    fn gsw_ctx_prepared_bytes(ctx: *const gsw_ctx) -> usize;
    fn gsw_eval_circuit_cached(ctx: *mut gsw_ctx, d_pool: *mut u32, slot_stride: usize, n_slots: u32, d_prepared_pool: *mut c_void, h_valid: *mut u8, gates: *const gsw_gate, n_gates: usize) -> c_int;
    fn gsw_keygen_device(ctx: *mut gsw_ctx, seed: u64, first_stream: u64, sampler: c_int, sigma: f64, unchecked: c_int, h_t_out: *mut u32, d_pk_matrix_out: *mut u32, d_pk_prepared: *mut c_void) -> c_int;
    fn gsw_mul_gadget_batch_static(ctx: *mut gsw_ctx, c1: *const u32, s1: usize, c2: *const u32, s2: usize, out: *mut u32, so: usize, batch: c_int, flags: c_uint) -> c_int;
}

/// Non-zero status -> the panic the reference would have raised (the library's message carries its text).
fn ok(rc: c_int) {
    if rc != 0 {
        panic!("{}", unsafe { CStr::from_ptr(gsw_last_error()) }.to_string_lossy());
    }
}

/// Which `gsw_ctx_create` ring a `RingElement` maps to.
pub trait HipRing: RingElement {
    const KIND: c_int;
    const LOG2Q: c_int;
}
impl HipRing for crate::field::Fp {
    const KIND: c_int = 0;
    const LOG2Q: c_int = 31;
}
impl<const M: u64> HipRing for crate::pow2_ring::Zpow2<M> {
    const KIND: c_int = 1;
    const LOG2Q: c_int = M as c_int;
}

/// A device allocation freed on drop.
struct Dev {
    ctx: *mut gsw_ctx,
    ptr: *mut c_void,
}
impl Dev {
    fn new(ctx: *mut gsw_ctx, bytes: usize) -> Self {
        let mut ptr = std::ptr::null_mut();
        ok(unsafe { gsw_malloc(ctx, bytes.max(16), &mut ptr) });
        Dev { ctx, ptr }
    }
    fn from_host(ctx: *mut gsw_ctx, v: &[u32]) -> Self {
        let d = Dev::new(ctx, v.len() * 4);
        ok(unsafe { gsw_memcpy_h2d(ctx, d.ptr, v.as_ptr() as _, v.len() * 4) });
        d
    }
    fn to_host(&self, len: usize) -> Vec<u32> {
        let mut v = vec![0u32; len];
        ok(unsafe { gsw_memcpy_d2h(self.ctx, v.as_mut_ptr() as _, self.ptr, len * 4) });
        v
    }
    fn u32(&self) -> *mut u32 {
        self.ptr as *mut u32
    }
}
impl Drop for Dev {
    fn drop(&mut self) {
        unsafe { gsw_free(self.ctx, self.ptr) };
    }
}

pub struct GswHip<R: HipRing, T: ErrorSampling<R>> {
    ctx: *mut gsw_ctx,
    n: usize,
    m: usize,
    err_sampling: T,
    _marker: PhantomData<R>,
}

fn canon<R: RingElement>(x: R) -> u32 {
    let v: u64 = x.into();
    v as u32
}

impl<R: HipRing + 'static, T: ErrorSampling<R>> GswHip<R, T> {
    pub fn new(n: usize, m: usize, err_sampling: T) -> Self {
        let mut ctx = std::ptr::null_mut();
        ok(unsafe { gsw_ctx_create(&mut ctx, R::KIND, R::LOG2Q, n as c_int, m as c_int, 0) });
        Self { ctx, n, m, err_sampling, _marker: PhantomData }
    }
    fn big_n(&self) -> usize {
        unsafe { gsw_ctx_big_n(self.ctx) as usize }
    }
    /// column-major `DMatrix<R>` -> row-major canonical u32
    fn to_rows(m: &DMatrix<R>) -> Vec<u32> {
        let mut v = Vec::with_capacity(m.len());
        for r in 0..m.nrows() {
            for c in 0..m.ncols() {
                v.push(canon(m[(r, c)]));
            }
        }
        v
    }
    fn from_rows(rows: usize, cols: usize, v: &[u32]) -> DMatrix<R> {
        DMatrix::from_fn(rows, cols, |r, c| R::from(v[r * cols + c] as u64))
    }
    fn same_size(a: &DMatrix<R>, b: &DMatrix<R>) {
        assert_eq!(a.nrows(), b.nrows(), "Cannot add Ciphertexts because they are different sizes");
        assert_eq!(a.ncols(), b.ncols(), "Cannot add Ciphertexts because they are different sizes");
    }
    fn dense_binary(
        &self,
        f: unsafe extern "C" fn(*mut gsw_ctx, *const u32, *const u32, *mut u32, c_int) -> c_int,
        a: &DMatrix<R>,
        b: &DMatrix<R>,
    ) -> DMatrix<R> {
        assert_eq!(a.nrows(), a.ncols(), "ciphertexts are square");
        let (da, db) = (Dev::from_host(self.ctx, &Self::to_rows(a)), Dev::from_host(self.ctx, &Self::to_rows(b)));
        let out = Dev::new(self.ctx, a.len() * 4);
        ok(unsafe { f(self.ctx, da.u32(), db.u32(), out.u32(), a.nrows() as c_int) });
        Self::from_rows(a.nrows(), a.ncols(), &out.to_host(a.len()))
    }
    /// Dense N x N ciphertext -> device-resident compact form (the linear BitDecomp^-1).
    fn upload_compact(&self, c: &DMatrix<R>) -> Dev {
        let big_n = self.big_n();
        assert_eq!((c.nrows(), c.ncols()), (big_n, big_n), "ciphertext is N x N");
        let dense = Dev::from_host(self.ctx, &Self::to_rows(c));
        let elems = unsafe { gsw_ctx_compact_elems(self.ctx) };
        let cc = Dev::new(self.ctx, elems * 4);
        unsafe {
            ok(gsw_memset_zero(self.ctx, cc.ptr, elems * 4));
            ok(gsw_bit_decomp_inv(self.ctx, dense.u32(), big_n as c_int, (self.n + 1) as c_int, cc.u32(), gsw_ctx_ld(self.ctx), 0));
        }
        cc
    }

    /// The throughput path for flattened ciphertexts that stay on the device: a list of gates over a pool
    /// of compact ciphertexts (`slots` of them, `gsw_ctx_compact_elems` u32 apart).
    pub fn eval_circuit(&self, d_pool: *mut u32, slots: u32, gates: &[gsw_gate]) {
        let stride = unsafe { gsw_ctx_compact_elems(self.ctx) };
        ok(unsafe { gsw_eval_circuit(self.ctx, d_pool, stride, slots, gates.as_ptr(), gates.len()) });
    }
    /// Batched product of device-resident compact ciphertexts: `out[b] = mult(c1[b], c2[b])` (or nand).
    pub fn mul_gadget_batch(&self, c1: *const u32, c2: *const u32, out: *mut u32, batch: usize, nand: bool) {
        let stride = unsafe { gsw_ctx_compact_elems(self.ctx) };
        let flags = if nand { GSW_FLAG_NAND } else { 0 };
        ok(unsafe { gsw_mul_gadget_batch(self.ctx, c1, stride, c2, stride, out, stride, batch as c_int, flags) });
    }
    pub fn sync(&self) {
        ok(unsafe { gsw_sync(self.ctx) });
    }
    /// The same gate list with a cache of prepared right operands beside the pool (`gsw_ctx_prepared_bytes`
    /// per slot, one validity flag per slot): a slot is prepared once until a gate overwrites it.
    pub fn eval_circuit_cached(&self, d_pool: *mut u32, slots: u32, d_prepared: *mut c_void, valid: &mut [u8], gates: &[gsw_gate]) {
        assert_eq!(valid.len(), slots as usize, "one flag per slot");
        let stride = unsafe { gsw_ctx_compact_elems(self.ctx) };
        ok(unsafe { gsw_eval_circuit_cached(self.ctx, d_pool, stride, slots, d_prepared, valid.as_mut_ptr(), gates.as_ptr(), gates.len()) });
    }
    pub fn prepared_bytes(&self) -> usize {
        unsafe { gsw_ctx_prepared_bytes(self.ctx) }
    }
    /// `mul_gadget_batch` for callers that may not allocate device memory (static scratch of the library).
    pub fn mul_gadget_batch_static(&self, c1: *const u32, c2: *const u32, out: *mut u32, batch: usize, nand: bool) {
        let stride = unsafe { gsw_ctx_compact_elems(self.ctx) };
        let flags = if nand { GSW_FLAG_NAND } else { 0 };
        ok(unsafe { gsw_mul_gadget_batch_static(self.ctx, c1, stride, c2, stride, out, stride, batch as c_int, flags) });
    }
    /// Which kernel served the last `mult` / `nand`: 1 = the int8-MFMA contraction (left operand in the reference's
    /// ciphertext format), 0 = the plain N^3 kernel.  `set_dense_mode(2)` takes the MFMA path at any size.
    pub fn last_dense_path(&self) -> i32 {
        unsafe { gsw_ctx_last_dense_path(self.ctx) }
    }
    /// How the last contraction launch recombined its limb sums (1 = 32-bit form, 0 = 64-bit form, -1 = none yet).
    pub fn last_combine(&self) -> i32 {
        unsafe { gsw_ctx_last_combine(self.ctx) }
    }
    /// flatten_matrix (helper.rs:58-62) on a host matrix, computed on the GPU: what the reference applies to the
    /// result of encrypt / add / mult_const / mult / nand under its `use_flatten` feature (Cargo.toml:16-17).
    #[cfg(feature = "use_flatten")]
    fn flatten_matrix(&self, c: &mut DMatrix<R>) {
        let (rows, cols) = (c.nrows(), c.ncols());
        let ell = R::Num_Bits as usize;
        assert_eq!(cols % ell, 0, "flatten needs a column count that is a multiple of Num_Bits");
        let d = Dev::from_host(self.ctx, &Self::to_rows(c));
        ok(unsafe { gsw_flatten(self.ctx, d.u32(), rows as c_int, (cols / ell) as c_int) });
        *c = Self::from_rows(rows, cols, &d.to_host(rows * cols));
    }
    pub fn set_dense_mode(&self, mode: i32) {
        ok(unsafe { gsw_ctx_set_dense_fast(self.ctx, mode) });
    }
    /// keygen (gsw_impl.rs:17-25) with the samples drawn by the library's seeded device generator and everything
    /// but the secret kept on the GPU: returns t and the m x ld public-key matrix as canonical values.
    pub fn keygen_on_device(&self, seed: u64, gaussian: bool) -> (Vec<u32>, Vec<u32>) {
        let ld = unsafe { gsw_ctx_ld(self.ctx) } as usize;
        let mut t = vec![0u32; self.n];
        let d_pk = Dev::new(self.ctx, self.m * ld * 4);
        let prepared = Dev::new(self.ctx, unsafe { gsw_ctx_pk_prepared_bytes(self.ctx) });
        ok(unsafe { gsw_keygen_device(self.ctx, seed, 1, gaussian as c_int, 1.0, 0, t.as_mut_ptr(), d_pk.u32(), prepared.ptr) });
        (t, d_pk.to_host(self.m * ld))
    }
}

impl<R: HipRing + 'static, T: ErrorSampling<R>> FheScheme<R> for GswHip<R, T> {
    type SecretKey = GswSk<R>;
    type PublicKey = GswPk<R>;
    type Ciphertext = DMatrix<R>;

    // gsw_impl.rs:17-25 -- the samples come from the host generator and `self.err_sampling` exactly as in
    // the reference; b = B t + e (pk.rs:26) runs on the GPU.
    fn keygen(&self) -> (Self::SecretKey, Self::PublicKey) {
        let sk = GswSk::new(rnd_dvec(self.n, 0, R::max_u64()));
        let err: DVector<R> = self.err_sampling.rnd_error_dvec(self.m);
        let random_matrix: DMatrix<R> = rnd_dmatrix(self.m, self.n, 0, R::max_u64());
        let (m, n) = (self.m, self.n);
        let d_b = Dev::from_host(self.ctx, &Self::to_rows(&random_matrix));
        let d_e = Dev::from_host(self.ctx, &err.iter().map(|x| canon(*x)).collect::<Vec<u32>>());
        let d_bt = Dev::new(self.ctx, m * 4);
        let t: Vec<u32> = sk.t.iter().map(|x| canon(*x)).collect();
        unsafe {
            ok(gsw_matvec(self.ctx, d_b.u32(), m as c_int, n as c_int, n as c_int, t.as_ptr(), d_bt.u32()));
            ok(gsw_add(self.ctx, d_bt.u32(), d_e.u32(), d_bt.u32(), m));
        }
        let b = DVector::from_iterator(m, d_bt.to_host(m).into_iter().map(|x| R::from(x as u64)));
        let mut pk_matrix = DMatrix::zeros(m, n + 1); // pk.rs:28-36
        pk_matrix.set_column(0, &b);
        pk_matrix.view_mut((0, 1), (m, n)).copy_from(&random_matrix);
        (sk, GswPk { b, pk_matrix })
    }

    // gsw_impl.rs:27-42 -- R * pk_matrix on the int8 MFMA path with R drawn by the seeded device generator
    // (the seed comes from the host generator the reference uses), BitDecomp on the device, message on
    // the diagonal on the host as :34-36 does.
    fn encrypt(&self, pk: &Self::PublicKey, message: R) -> Self::Ciphertext {
        let big_n = self.big_n();
        let prepared = Dev::new(self.ctx, unsafe { gsw_ctx_pk_prepared_bytes(self.ctx) });
        ok(unsafe { gsw_prepare_pk(self.ctx, Self::to_rows(&pk.pk_matrix).as_ptr(), prepared.ptr) });
        let seed: u64 = rand::rng().random();
        let zero = Dev::from_host(self.ctx, &[0u32]);
        let elems = unsafe { gsw_ctx_compact_elems(self.ctx) };
        let cc = Dev::new(self.ctx, elems * 4);
        let dense = Dev::new(self.ctx, big_n * big_n * 4);
        unsafe {
            ok(gsw_encrypt_batch_seeded(self.ctx, prepared.ptr, seed, 1, zero.u32(), cc.u32(), elems, std::ptr::null_mut(), 1));
            ok(gsw_bit_decomp(self.ctx, cc.u32(), big_n as c_int, (self.n + 1) as c_int, gsw_ctx_ld(self.ctx), dense.u32()));
        }
        let mut product = Self::from_rows(big_n, big_n, &dense.to_host(big_n * big_n));
        for i in 0..product.ncols() {
            product[(i, i)] += message;
        }
        #[cfg(feature = "use_flatten")] // gsw_impl.rs:38-39
        self.flatten_matrix(&mut product);
        product
    }

    // gsw_impl.rs:48-52
    fn decrypt(&self, sk: &Self::SecretKey, c: &Self::Ciphertext) -> R {
        let dc = Dev::from_host(self.ctx, &Self::to_rows(c));
        let v: Vec<u32> = sk.v.iter().map(|x| canon(*x)).collect();
        let mut bit = 0u32;
        ok(unsafe { gsw_decrypt_dense(self.ctx, v.as_ptr(), dc.u32(), &mut bit) });
        R::from(bit as u64)
    }

    // gsw_impl.rs:67-84 -- <C[i, :], v> = <BitDecomp^-1(C)[i, :], s>; the library answers
    // "Only supported for pow2 rings!" for Fp and `ok` panics with it, as :69 does.
    fn mp_decrypt(&self, sk: &Self::SecretKey, c: &Self::Ciphertext) -> R {
        let cc = self.upload_compact(c);
        let s: Vec<u32> = sk.s.iter().map(|x| canon(*x)).collect();
        let mut message = 0u32;
        ok(unsafe { gsw_mp_decrypt_batch(self.ctx, s.as_ptr(), cc.u32(), 0, 1, &mut message) });
        R::from(message as u64)
    }

    // gsw_impl.rs:87-94 and :97-101: O(N^2) on data that lives on the host -- not worth a PCIe round trip
    // (gsw_add / gsw_mult_const serve device-resident ciphertexts)
    fn add(&self, a: &Self::Ciphertext, b: &Self::Ciphertext) -> Self::Ciphertext {
        Self::same_size(a, b);
        #[allow(unused_mut)]
        let mut res = a + b;
        #[cfg(feature = "use_flatten")] // gsw_impl.rs:91-92
        self.flatten_matrix(&mut res);
        res
    }
    fn mult_const(&self, c: &mut Self::Ciphertext, constant: R) {
        *c *= constant;
        #[cfg(feature = "use_flatten")] // gsw_impl.rs:99-100
        self.flatten_matrix(c);
    }

    // gsw_impl.rs:104-111 (default features).  A left operand in the reference's own ciphertext format --
    // BitDecomp(R * A) + mu * I: 0 / 1 off the diagonal -- runs on the int8-MFMA contraction inside
    // gsw_mul_dense (3.3 ms against 78 ms at N = 7967); any other entries take the plain N^3 kernel.
    fn mult(&self, a: &Self::Ciphertext, b: &Self::Ciphertext) -> Self::Ciphertext {
        Self::same_size(a, b);
        #[allow(unused_mut)]
        let mut res = self.dense_binary(gsw_mul_dense, a, b);
        #[cfg(feature = "use_flatten")] // gsw_impl.rs:108-109
        self.flatten_matrix(&mut res);
        res
    }
    // gsw_impl.rs:114-123
    fn nand(&self, a: &Self::Ciphertext, b: &Self::Ciphertext) -> Self::Ciphertext {
        #[allow(unused_mut)]
        let mut res = self.dense_binary(gsw_nand_dense, a, b);
        #[cfg(feature = "use_flatten")] // gsw_impl.rs:120-121
        self.flatten_matrix(&mut res);
        res
    }
}

impl<R: HipRing, T: ErrorSampling<R>> Drop for GswHip<R, T> {
    fn drop(&mut self) {
        unsafe { gsw_ctx_destroy(self.ctx) }
    }
}

#[cfg(test)]
mod tests {
    // The crate's own scheme tests (gsw_impl.rs:165-219) with `GSW { .. }` replaced by `GswHip::new(..)`.
    use super::*;
    use crate::{error_sampling::{DiscrGaussianSampler, NaiveSampler}, field::Fp, pow2_ring::Zpow2};

    #[test]
    fn encryption_decryption_naive() {
        test_inputs(GswHip::<Fp, NaiveSampler>::new(10, 10 * Fp::Num_Bits, NaiveSampler));
    }

    #[test]
    fn encryption_decryption_discr_gaussian() {
        test_inputs(GswHip::<Zpow2<30>, DiscrGaussianSampler>::new(10, 10 * Zpow2::<30>::Num_Bits, DiscrGaussianSampler::default()));
    }

    #[test]
    fn encryption_decryption_pow_of_two() {
        let fhe = GswHip::<Zpow2<31>, DiscrGaussianSampler>::new(10, 10 * Zpow2::<31>::Num_Bits, DiscrGaussianSampler::default());
        let (sk, pk) = fhe.keygen();
        let mut rng = rand::rng();
        for _ in 0..20 {
            let msg = Zpow2::<31>::from(rng.random_range(0..Zpow2::<31>::max_u64()));
            let encr = fhe.encrypt(&pk, msg);
            let decr = fhe.mp_decrypt(&sk, &encr);
            assert_eq!(decr, msg, "{:064b} vs {:064b}", decr.value(), msg.value());
        }
    }

    #[test]
    fn mult_and_nand_decrypt() {
        let fhe = GswHip::<Fp, DiscrGaussianSampler>::new(10, 10 * Fp::Num_Bits, DiscrGaussianSampler::default());
        let (sk, pk) = fhe.keygen();
        for (a, b) in [(0u64, 0u64), (0, 1), (1, 0), (1, 1)] {
            let (ca, cb) = (fhe.encrypt(&pk, Fp::from(a)), fhe.encrypt(&pk, Fp::from(b)));
            assert_eq!(fhe.decrypt(&sk, &fhe.mult(&ca, &cb)), Fp::from(a * b));
            assert_eq!(fhe.decrypt(&sk, &fhe.nand(&ca, &cb)), Fp::from(1 - a * b));
        }
    }

    fn test_inputs<R: RingElement + 'static, T: FheScheme<R>>(fhe: T)
    where
        T::Ciphertext: Sized,
    {
        let (sk, pk) = fhe.keygen();
        let encr = fhe.encrypt(&pk, R::zero());
        assert_eq!(fhe.decrypt(&sk, &encr), R::zero());
        let encr = fhe.encrypt(&pk, R::one());
        assert_eq!(fhe.decrypt(&sk, &encr), R::one());
    }
}
