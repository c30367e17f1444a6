/*
 * gsw_hip.h -- C ABI of the MI355X (gfx950) engine for the GSW hot path of n0900/tfhe.
 *
 * The reference crate has no FFI; the boundary these entry points replace is the Rust
 * trait `FheScheme<R>` as implemented for `GSW<R, T>` (src/gsw/mod.rs:12-26,
 * src/gsw/gsw_impl.rs:12-125).  A Rust shim implements that trait on top of this header
 * (integration/rust/hip.rs).  Every entry point below cites the reference lines it stands for.
 *
 * Element type.  The reference's `R: RingElement` (src/lib.rs:15-29) is `Fp`
 * (src/field.rs:7-19, p = 2^31 - 1, stored in Montgomery form) or `Zpow2<M>`
 * (src/pow2_ring.rs:14-16).  Across this ABI every element is its CANONICAL value in
 * [0, q) as a uint32_t (`Into<u64>` of field.rs:52-56 / pow2_ring.rs:148-152, truncated).
 * Zpow2<M> is supported for 1 <= M <= 32.
 *
 * Matrix forms.
 *  - dense:    the reference's `Ciphertext = DMatrix<R>` (gsw_impl.rs:15), N x N with
 *              N = ell * (n + 1), here ROW-MAJOR uint32 (nalgebra is column-major).
 *  - compact:  Cc = BitDecomp^-1_linear(C), N rows x (n + 1) columns, row-major uint32
 *              with leading dimension ld = gsw_ctx_ld() (n + 1 rounded up to 16); the
 *              padding columns must be zero (every routine here keeps them zero).
 *              BitDecomp(Cc) is the flattened ciphertext; the product of gsw_impl.rs:107
 *              satisfies BitDecomp^-1_linear(C1 * C2) = C1 * C2c, which for a flattened
 *              left operand is the binary x byte-limb contraction the MFMA kernel computes.
 *  - prepared: the right operand re-laid out as signed byte-limb planes in MFMA fragment
 *              order (opaque; size gsw_ctx_prepared_bytes()).
 *
 * All pointers named d_* are DEVICE pointers on the context's GPU.  Calls enqueue work on
 * the context's stream and return; gsw_sync() waits.  Return value: 0 on success, else a
 * GSW_E_* code with a message in gsw_last_error() (the shim turns these into the panics of
 * the reference: gsw_impl.rs:69,88-89,105-106; error_sampling.rs:21; pk.rs:24-25).
 * A context is thread-compatible (one thread at a time).
 */
#ifndef GSW_HIP_H
#define GSW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSW_RING_FP_M31 0 /* Fp, field.rs:7-19                  */
#define GSW_RING_ZPOW2 1  /* Zpow2<M>, pow2_ring.rs:14-16       */

#define GSW_OK 0
#define GSW_E_INVALID 1     /* bad argument / dimension mismatch */
#define GSW_E_UNSUPPORTED 2 /* ring or operation not supported   */
#define GSW_E_HIP 3         /* HIP runtime error                 */
#define GSW_E_NOMEM 4       /* device allocation failed          */

#define GSW_FLAG_NAND 1u /* out = G - G^-1(C1c) * C2c  (gsw_impl.rs:114-123 in compact form) */

typedef struct gsw_ctx gsw_ctx;

/* ---- context: stands for the `GSW<R, T> { n, m, .. }` value (gsw/mod.rs:29-38) ---- */
int gsw_ctx_create(gsw_ctx **out, int ring_kind, int log2q, int n, int m, int device);
void gsw_ctx_destroy(gsw_ctx *ctx);
const char *gsw_last_error(void);
const char *gsw_version(void);
/* use the caller's hipStream_t (e.g. the framework's current stream); NULL restores the context's own
 * stream.  The legacy default stream has the handle NULL and therefore cannot be selected: a caller whose
 * other work runs there either passes an explicit stream it also uses for that work, or brackets the
 * calls with gsw_sync() / its own device synchronisation. */
int gsw_ctx_set_stream(gsw_ctx *ctx, void *hip_stream);
int gsw_sync(gsw_ctx *ctx);
int gsw_ctx_ell(const gsw_ctx *ctx);               /* R::Num_Bits (field.rs:39, pow2_ring.rs:19)   */
int gsw_ctx_big_n(const gsw_ctx *ctx);             /* N = ell * (n + 1) (gsw_impl.rs:28)           */
int gsw_ctx_ld(const gsw_ctx *ctx);                /* leading dimension of compact ciphertexts      */
int gsw_ctx_gemm_variant(const gsw_ctx *ctx);      /* contraction kernel in use (GSW_GEMM_VARIANT or chosen by shape) */
size_t gsw_ctx_compact_elems(const gsw_ctx *ctx);  /* N * ld uint32 per compact ciphertext          */
size_t gsw_ctx_prepared_bytes(const gsw_ctx *ctx); /* bytes of one prepared right operand           */
size_t gsw_ctx_rbits_elems(const gsw_ctx *ctx);    /* N * ld_r uint32 of packed encryption bits     */
size_t gsw_ctx_pk_prepared_bytes(const gsw_ctx *ctx);
/* upper bound, in bytes, of the scratch the batch entry points may allocate (default 2 GiB) */
int gsw_ctx_set_workspace_limit(gsw_ctx *ctx, size_t bytes);
/* per-kernel timing with HIP events on the context's stream: enable, then read the totals */
int gsw_ctx_enable_timing(gsw_ctx *ctx, int enable);
/* which: 0 = contraction kernel, 1 = right-operand preparation.  Waits for the stream. */
int gsw_ctx_kernel_time(gsw_ctx *ctx, int which, double *total_ms, long *launches);
int gsw_ctx_reset_timing(gsw_ctx *ctx);
/* work counters since the context was created: right operands put into prepared form (the public key of
 * gsw_prepare_pk included) and batch entries handed to the contraction kernel */
int gsw_ctx_counters(const gsw_ctx *ctx, uint64_t *prepared_operands, uint64_t *contracted_products);

/* ---- device memory helpers (plain hipMalloc / hipMemcpy on the context's GPU) ---- */
int gsw_malloc(gsw_ctx *ctx, size_t bytes, void **d_ptr);
int gsw_free(gsw_ctx *ctx, void *d_ptr);
int gsw_memcpy_h2d(gsw_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int gsw_memcpy_d2h(gsw_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int gsw_memset_zero(gsw_ctx *ctx, void *d_dst, size_t bytes);
/* host (rows x cols, row-major, tight) <-> device (rows x ld): pads / strips the leading dimension */
int gsw_upload_compact(gsw_ctx *ctx, uint32_t *d_dst, const uint32_t *h_src);
int gsw_download_compact(gsw_ctx *ctx, uint32_t *h_dst, const uint32_t *d_src);

/* ---- gadget helpers (gsw/helper.rs) ---- */
/* bit_decomp_matrix (helper.rs:7-21): d_in rows x cols (leading dim ld_in) -> d_out rows x cols*ell, tight */
int gsw_bit_decomp(gsw_ctx *ctx, const uint32_t *d_in, int rows, int cols, int ld_in, uint32_t *d_out);
/* The same decomposition in the format its consumer reads (helper.rs:7-21 fixes the bit order, not the storage):
 * GSW_DECOMP_U32 as above; GSW_DECOMP_U8 one byte per bit (rows of cols*ell bytes); GSW_DECOMP_BITS the row's
 * cols*ell bits as one little-endian bit string, 32 per uint32, rows of ceil(cols*ell / 32) words.
 * gsw_bit_decomp_row_elems gives the elements (uint32 / bytes / words) of one output row; rows are tight. */
#define GSW_DECOMP_U32 0
#define GSW_DECOMP_U8 1
#define GSW_DECOMP_BITS 2
size_t gsw_bit_decomp_row_elems(const gsw_ctx *ctx, int cols, int format);
int gsw_bit_decomp_as(gsw_ctx *ctx, const uint32_t *d_in, int rows, int cols, int ld_in, void *d_out, int format);
/* bit_decomp_inv_matrix (helper.rs:24-49): d_in rows x cols*ell tight -> d_out rows x cols (leading dim ld_out).
 * quirk != 0 reproduces the reference (bit set iff entry == 1, then R::from); quirk == 0 is the linear inverse. */
int gsw_bit_decomp_inv(gsw_ctx *ctx, const uint32_t *d_in, int rows, int cols, uint32_t *d_out, int ld_out,
                       int quirk);
/* flatten_matrix (helper.rs:51-62), in place on a rows x cols*ell tight matrix */
int gsw_flatten(gsw_ctx *ctx, uint32_t *d_mat, int rows, int cols);

/* ---- FheScheme operations on dense N x N matrices (the reference's own form) ---- */
/* add (gsw_impl.rs:87-94) and mult_const (gsw_impl.rs:97-101), elementwise over `count` elements;
 * also valid on compact ciphertexts (count = gsw_ctx_compact_elems) */
int gsw_add(gsw_ctx *ctx, const uint32_t *d_a, const uint32_t *d_b, uint32_t *d_out, size_t count);
int gsw_mult_const(gsw_ctx *ctx, uint32_t *d_inout, uint32_t constant, size_t count);
/* mult (gsw_impl.rs:104-111, default features) and nand (gsw_impl.rs:114-123): the dim x dim x dim product
 * over R for arbitrary entries.  Three kernels, identical results:
 *  1. dim is the context's N and the left operand has the reference's own ciphertext format -- BitDecomp(R * A) +
 *     mu * I (gsw_impl.rs:27-42): entries 0 / 1 everywhere off the diagonal -- : B * C2 + diag(d) * C2 on the binary x
 *     digit int8-MFMA contraction (B the binary part, d the diagonal entries other than 0 / 1).  The form is detected
 *     on the device (one pass over C1 and a host synchronisation per call).
 *  2. any other left operand (what mult(mult(a, b), c) multiplies by in the reference's default build, which never
 *     flattens: Cargo.toml:16-17): both operands split into four signed base-256 digits, sixteen digit x digit
 *     int8-MFMA products with exact int32 accumulation, recombined mod q.  Needs 4 * pad^2 bytes of workspace for the
 *     right operand (pad = dim rounded up to 64) plus 4 * pad bytes per left-operand row it takes at a time, within the
 *     workspace limit; dim < 32768.
 *  3. the plain dim^3 kernel: small matrices (below 1024 rows by default), or when the workspace limit rules out 2. */
int gsw_mul_dense(gsw_ctx *ctx, const uint32_t *d_c1, const uint32_t *d_c2, uint32_t *d_out, int dim);
int gsw_nand_dense(gsw_ctx *ctx, const uint32_t *d_c1, const uint32_t *d_c2, uint32_t *d_out, int dim);
/* 0: every dense product on the plain kernel (A/B timing, cross-checks); 1 (default): the MFMA kernels from
 * 1024 rows up, where they pay (N = 341: 0.06 ms against 0.02 ms plain; N = 7967: 3.2 ms against 78 ms);
 * 2: the MFMA kernels at any size */
int gsw_ctx_set_dense_fast(gsw_ctx *ctx, int enable);
/* which kernel the last dense product of this context used: 0 = plain dim^3, 1 = binary x digit contraction
 * (flattened left operand), 2 = digit x digit products (arbitrary entries) */
int gsw_ctx_last_dense_path(const gsw_ctx *ctx);
/* Test aid, no counterpart in the reference.  How the last contraction launch of this context recombined its four
 * limb sums: 1 = the 32-bit form with the branch-free epilogue on interior tiles (every benchmark shape), 0 = the
 * general epilogue with the 64-bit form on every tile (contractions too long for the 32-bit bound, or
 * GSW_FORCE_WIDE_COMBINE=1 in the environment), -1 = no launch yet. */
int gsw_ctx_last_combine(const gsw_ctx *ctx);

/* ---- the hot path: Pc = G^-1(C1c) * C2c mod q on int8 MFMA (gsw_impl.rs:104-123 via the
 *      identity above).  Strides are in uint32 elements between consecutive batch entries;
 *      a stride of 0 reuses one operand for the whole batch.  The output may not share memory with
 *      an operand (GSW_E_INVALID): the reference returns a fresh matrix (gsw_impl.rs:107). ---- */
/* Invariants of the operands, relied upon but not tested by the contraction (gsw_validate_compact tests
 * them): every word is a canonical ring value (no bit at or above ell), and every padding word of a row
 * (columns n + 1 .. ld - 1) is zero.  The right operand is split into signed base-256 digits on the
 * assumption that its words are canonical (an element of Fp at or above p comes out as another element).  In
 * the left operand a bit at or above ell, or a padding word, only meets K slots whose digits are 0.
 * Everything this library writes keeps the invariant (products write zeros into the padding of their
 * output); a caller that fills a buffer itself should zero the padding. */
/* *h_bad_out = number of words, over `batch` compact ciphertexts, that are not canonical or are non-zero
 * padding.  Waits for the stream. */
int gsw_validate_compact(gsw_ctx *ctx, const uint32_t *d_cc, size_t stride, int batch, uint64_t *h_bad_out);
/* limb-split one or more compact right operands into prepared form */
int gsw_prepare_rhs(gsw_ctx *ctx, const uint32_t *d_c2c, size_t stride_c2c, void *d_prepared, int batch);
int gsw_mul_gadget_prepared(gsw_ctx *ctx, const uint32_t *d_c1c, size_t stride1, const void *d_prepared,
                            size_t stride_prepared_bytes, uint32_t *d_out, size_t stride_out, int batch,
                            unsigned flags);
/* same, preparing the right operands internally in workspace chunks */
int gsw_mul_gadget_batch(gsw_ctx *ctx, const uint32_t *d_c1c, size_t stride1, const uint32_t *d_c2c, size_t stride2,
                         uint32_t *d_out, size_t stride_out, int batch, unsigned flags);
/* same, and d_digests[b] receives the gsw_digest_batch() digest of product b (computed in the
 * contraction's epilogue, so the products are not read back) */
int gsw_mul_gadget_batch_digest(gsw_ctx *ctx, const uint32_t *d_c1c, size_t stride1, const uint32_t *d_c2c,
                                size_t stride2, uint32_t *d_out, size_t stride_out, int batch, unsigned flags,
                                uint64_t *d_digests);

/* No counterpart in the reference.  gsw_mul_gadget_batch for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the prepared right
 * operands are staged in a fixed device array of the library, GSW_STATIC_SCRATCH_BYTES per GPU, held from
 * the library's load to the end of the process, instead of the context's growable workspace.  The call
 * allocates and frees nothing on the device or through the HIP runtime.  Products go through the array
 * in chunks of floor(GSW_STATIC_SCRATCH_BYTES / gsw_ctx_prepared_bytes()) on the context's stream, one
 * after another, so any batch length is accepted; a context whose single prepared operand exceeds the
 * array (n above 512 for Fp) is refused with GSW_E_UNSUPPORTED.  Kernel timing (gsw_ctx_enable_timing)
 * does not cover this call.  Every context on a GPU shares that GPU's array: the library orders its users -- a
 * call's stream first waits (hipStreamWaitEvent on a per-GPU event made at gsw_ctx_create) for the last contraction
 * of the call before it, whichever context or stream that was -- so calls from several contexts or threads queue
 * up on the array instead of racing on it; they do not overlap on the GPU. */
#define GSW_STATIC_SCRATCH_BYTES ((size_t)64 << 20)
int gsw_mul_gadget_batch_static(gsw_ctx *ctx, const uint32_t *d_c1c, size_t stride1, const uint32_t *d_c2c,
                                size_t stride2, uint32_t *d_out, size_t stride_out, int batch, unsigned flags);

/* ---- gate lists over resident ciphertexts: what a caller of mult / nand / add / mult_const
 *      (gsw_impl.rs:87-123) does next -- chains and NAND circuits -- without the operands leaving
 *      HBM.  The pool holds n_slots compact ciphertexts slot_stride uint32 apart. ---- */
#define GSW_GATE_MULT 0u       /* pool[out] = mult(pool[a], pool[b])         gsw_impl.rs:104-111 */
#define GSW_GATE_NAND 1u       /* pool[out] = nand(pool[a], pool[b])         gsw_impl.rs:114-123 */
#define GSW_GATE_ADD 2u        /* pool[out] = add(pool[a], pool[b])          gsw_impl.rs:87-94   */
#define GSW_GATE_MULT_CONST 3u /* pool[out] = b * pool[a], b a ring element  gsw_impl.rs:97-101  */
typedef struct gsw_gate {
    uint32_t op, a, b, out;
} gsw_gate;
/* Applies the gates with the meaning of running them one after another in list order.  Consecutive
 * product gates of one kind that do not depend on each other are issued as one batched contraction,
 * so a levelised circuit costs one launch sequence per level.  A product gate must not name one of
 * its own operands as its output (GSW_E_INVALID); add and mult_const may work in place. */
int gsw_eval_circuit(gsw_ctx *ctx, uint32_t *d_pool, size_t slot_stride, uint32_t n_slots, const gsw_gate *h_gates,
                     size_t n_gates);

/* Same, with a cache of prepared right operands beside the pool, so that a slot is put into prepared form once
 * and not once per gate that multiplies by it: d_prepared_pool holds n_slots entries of gsw_ctx_prepared_bytes()
 * each (entry s belongs to slot s); h_valid[s] != 0 says entry s currently is the prepared form of slot s.  A
 * product whose right operand is valid runs without any preparation; one whose operand is stale prepares it into
 * its entry first and sets the flag; every gate clears the flag of the slot it writes.  The flags are the
 * caller's array (in/out): set them for entries filled otherwise (gsw_encrypt_batch_prepared, gsw_prepare_rhs) and
 * clear them for slots changed behind the library's back.  A run of independent products is one contraction
 * launch straight out of the cache.  A list the checks reject leaves flags and pool untouched; a failure after
 * launches have begun clears every flag.  The flags are only as good as the caller's bookkeeping: ANY other write
 * into a pool slot (gsw_encrypt_batch*, gsw_upload_compact, gsw_add / gsw_mult_const, a product or gsw_eval_circuit
 * with the pool as output, a copy) must clear that slot's flag, or later gates multiply by the stale prepared
 * operand without any error.  (The Python mirror, tfhe_amd/gsw.py, binds a cache to its pool and does this itself.) */
int gsw_eval_circuit_cached(gsw_ctx *ctx, uint32_t *d_pool, size_t slot_stride, uint32_t n_slots, void *d_prepared_pool,
                            uint8_t *h_valid, const gsw_gate *h_gates, size_t n_gates);

/* ---- encrypt (gsw_impl.rs:27-42): compact form R * pk + message * G, the same contraction
 *      with the N x m random bit matrix as the binary operand ---- */
/* pk_matrix (pk.rs:28-38), m x (n + 1) row-major tight on the HOST -> prepared device form */
int gsw_prepare_pk(gsw_ctx *ctx, const uint32_t *h_pk_matrix, void *d_pk_prepared);
/* rnd_dmatrix(N, m, 0, 1) of gsw_impl.rs:30 as packed words (ell bits per word, word j of a row
 * holds the bits for pk rows j*ell .. j*ell + ell - 1, little endian); host 0/1 bytes -> packed device */
int gsw_pack_rbits(gsw_ctx *ctx, const uint8_t *h_bits /* N x m */, uint32_t *d_rbits);
/* seeded device generator for the same matrix (uniform bits; the reference draws from rand::rng()) */
int gsw_random_rbits(gsw_ctx *ctx, uint64_t seed, uint64_t stream, uint32_t *d_rbits, int batch);
int gsw_encrypt_batch(gsw_ctx *ctx, const void *d_pk_prepared, const uint32_t *d_rbits, size_t stride_rbits,
                      const uint32_t *d_messages /* batch values */, uint32_t *d_out, size_t stride_out, int batch);

/* gsw_encrypt_batch that also leaves every fresh ciphertext in prepared form (entry b at d_prepared_out +
 * b * gsw_ctx_prepared_bytes()), the form in which a product reads its right operand: acc <- mult(acc, fresh)
 * chains then never prepare anything (gsw_mul_gadget_prepared, gsw_eval_circuit_cached) */
int gsw_encrypt_batch_prepared(gsw_ctx *ctx, const void *d_pk_prepared, const uint32_t *d_rbits, size_t stride_rbits,
                               const uint32_t *d_messages, uint32_t *d_out, size_t stride_out, void *d_prepared_out,
                               int batch);

/* encrypt (gsw_impl.rs:27-42) with rnd_dmatrix(N, m, 0, 1) of gsw_impl.rs:30 drawn inside the call: the same
 * ciphertexts, bit for bit, as gsw_random_rbits(seed, stream, R, batch) followed by gsw_encrypt_batch on R, but
 * the bit matrices (N * ld_r words each) exist only chunk by chunk in the context's workspace, drawn beside the
 * contraction of the chunk before.  d_prepared_out may be NULL; otherwise it receives the prepared form of
 * every ciphertext as in gsw_encrypt_batch_prepared. */
int gsw_encrypt_batch_seeded(gsw_ctx *ctx, const void *d_pk_prepared, uint64_t seed, uint64_t stream,
                             const uint32_t *d_messages, uint32_t *d_out, size_t stride_out, void *d_prepared_out,
                             int batch);

/* ---- decrypt (gsw_impl.rs:48-52) and mp_decrypt (gsw_impl.rs:67-84) on compact ciphertexts.
 *      h_s is sk.s = (1, -t) (sk.rs:20-24) on the host; <C[i,:], v> = <Cc[i,:], s>. ---- */
int gsw_decrypt_batch(gsw_ctx *ctx, const uint32_t *h_s, const uint32_t *d_cc, size_t stride, int batch,
                      uint32_t *h_bits_out, uint32_t *h_raw_out /* may be NULL: <row ell-1, s> */);
/* GSW_E_UNSUPPORTED for Fp, as the reference panics (gsw_impl.rs:68-70) */
int gsw_mp_decrypt_batch(gsw_ctx *ctx, const uint32_t *h_s, const uint32_t *d_cc, size_t stride, int batch,
                         uint32_t *h_messages_out);
/* Threshold hooks.  The reference's threshold layer shares sk.s among parties (tfhe_gsw_fp.rs:14-18) and
 * leaves part_dec as todo!() (tfhe_gsw_fp.rs:27-29).  With additive shares s = s_1 + .. + s_k mod q,
 * party i publishes <row ell-1 of Cc, s_i> (n + 1 share values on the host); the sum of all partials
 * is the value decrypt applies its quarter-window test to (gsw_impl.rs:48-52,127-131). */
int gsw_partial_decrypt_batch(gsw_ctx *ctx, const uint32_t *h_share, const uint32_t *d_cc, size_t stride, int batch,
                              uint32_t *h_partial_out);
/* h_partials is parties x batch, row per party */
int gsw_combine_partials(gsw_ctx *ctx, const uint32_t *h_partials, int parties, int batch, uint32_t *h_bits_out);
/* d_out[i] = <d_mat[i, 0..cols), h_vec> mod q for i < rows (leading dimension ld): the matrix-vector
 * products of keygen (b = B t, pk.rs:26) and of the key invariant pk_matrix * s = e (gsw_impl.rs:159) */
int gsw_matvec(gsw_ctx *ctx, const uint32_t *d_mat, int rows, int cols, int ld, const uint32_t *h_vec,
               uint32_t *d_out);
/* dense-form decrypt: row ell-1 of C dotted with sk.v (gsw_impl.rs:48-52) */
int gsw_decrypt_dense(gsw_ctx *ctx, const uint32_t *h_v, const uint32_t *d_c, uint32_t *h_bit_out);

/* ---- sampling (error_sampling.rs) and digests ---- */
/* rnd_ring_elm (error_sampling.rs:20-24): uniform in [min, max]; fails with GSW_E_INVALID when
 * max > P unless `unchecked` (the reference asserts max <= P, which makes Zpow2<32> keygen panic) */
int gsw_fill_uniform(gsw_ctx *ctx, uint64_t seed, uint64_t stream, uint64_t min, uint64_t max, int unchecked,
                     uint32_t *d_out, size_t count);
/* NaiveSampler (error_sampling.rs:73-84) and DiscrGaussianSampler (error_sampling.rs:32-71) */
int gsw_sample_naive(gsw_ctx *ctx, uint64_t seed, uint64_t stream, uint32_t *d_out, size_t count);
int gsw_sample_gaussian(gsw_ctx *ctx, uint64_t seed, uint64_t stream, double sigma, uint32_t *d_out, size_t count);
/* keygen (gsw_impl.rs:17-25) with nothing but the secret crossing to the host: t <- U[0, max]^n, e from the sampler
 * (m values), B <- U[0, max]^(m x n) on generator streams first_stream, +1, +2 (the order of gsw_impl.rs:18-21);
 * b = B t + e, pk_matrix = [b | B] (pk.rs:23-39) and its prepared form are made on the device.  h_t_out receives t
 * (GswSk::new, sk.rs:20-28, is host logic on n + 1 values); d_pk_matrix_out (m rows x gsw_ctx_ld() uint32, may be
 * NULL) the matrix; d_pk_prepared (gsw_ctx_pk_prepared_bytes()) what gsw_encrypt_batch reads.  Fails with
 * GSW_E_INVALID when max > P unless `unchecked`, as the reference asserts (error_sampling.rs:21). */
#define GSW_SAMPLER_NAIVE 0    /* NaiveSampler, error_sampling.rs:73-84 */
#define GSW_SAMPLER_GAUSSIAN 1 /* DiscrGaussianSampler, error_sampling.rs:32-71 */
int gsw_keygen_device(gsw_ctx *ctx, uint64_t seed, uint64_t first_stream, int sampler, double sigma, int unchecked,
                      uint32_t *h_t_out, uint32_t *d_pk_matrix_out, void *d_pk_prepared);
/* 64-bit position-weighted digest of each batch entry (count uint32 each) */
int gsw_digest_batch(gsw_ctx *ctx, const uint32_t *d_x, size_t stride, size_t count, int batch,
                     uint64_t *d_digests);

/* ---- wire format.  The reference has no serialisation (its keys and ciphertexts are in-memory
 *      nalgebra values: pk.rs:13-16, sk.rs:13-17, gsw_impl.rs:15); this is the canonical form in which
 *      they travel between hosts or are stored.  All integers little endian:
 *        0  "GSW1"            4  kind (GSW_WIRE_*)   5  ring kind (GSW_RING_*)   6  bits per element   7  0
 *        8  n   12  m   16  rows   20  cols   24  entry count   28  0          (u32 each)
 *       32  count * rows * cols canonical u32 values, row-major, no padding
 *      end  u64 digest of the payload taken as one entry of gsw_digest_batch
 *      rows x cols: ciphertext N x (n+1) in compact form (BitDecomp^-1 of the reference's matrix,
 *      helper.rs:24-49 without the == 1 test); public key m x (n+1) = pk_matrix (pk.rs:28-38);
 *      secret key 1 x n = t (sk.rs:20-28; s and v are recomputed from it).
 *      Readers reject a wrong magic, kind, ring, shape or size, non-canonical elements and a digest
 *      mismatch with GSW_E_INVALID. ---- */
#define GSW_WIRE_CIPHERTEXT 1
#define GSW_WIRE_PUBLIC_KEY 2
#define GSW_WIRE_SECRET_KEY 3
/* bytes of a wire object of `count` entries (keys: count = 1); 0 for a bad argument */
size_t gsw_wire_bytes(const gsw_ctx *ctx, int kind, int count);
/* keys live on the host: pk_matrix (m x (n+1) row-major) or t (n values) */
int gsw_wire_write_host(gsw_ctx *ctx, int kind, const uint32_t *h_values, void *h_out, size_t out_bytes);
int gsw_wire_read_host(gsw_ctx *ctx, int kind, const void *h_in, size_t in_bytes, uint32_t *h_values);
/* device-resident compact ciphertexts <-> wire bytes on the host; with d_cc == NULL the reader only
 * validates and reports the entry count */
int gsw_wire_write_compact(gsw_ctx *ctx, const uint32_t *d_cc, size_t stride, int batch, void *h_out,
                           size_t out_bytes);
int gsw_wire_read_compact(gsw_ctx *ctx, const void *h_in, size_t in_bytes, uint32_t *d_cc, size_t stride,
                          int max_batch, int *batch_out);

/* Profiling aid, no counterpart in the reference: streams `bytes` (a multiple of 64 KiB) of d_src once
 * through the contraction's own LDS-DMA load instruction -- mode 0: 1 KiB contiguous per wave
 * instruction (right-operand pattern); mode 1: 16 rows x 64 B at `pitch` bytes (left-operand pattern,
 * which spans bytes / 64 * pitch bytes of d_src) -- so that rocprofv3's FETCH_SIZE can be calibrated
 * on a known byte count in those patterns.  src_bytes is the size of the buffer behind d_src. */
int gsw_diag_stream_read(gsw_ctx *ctx, const void *d_src, size_t src_bytes, size_t bytes, int mode, size_t pitch);

/* Test aid, no counterpart in the reference, host only (no context, no GPU).  The right operand of the
 * contraction is held as four signed base-256 digits per element, so a limb sum over k_slots K slots lies in
 * [-128 k_slots, 127 k_slots]; the accumulators of limb l start at h_start4[l] instead of 0.  For Fp these keep
 * every sum non-negative and satisfy sum_l 2^(8 l) h_start4[l] = 0 mod p, so the recombined value is unchanged;
 * for Z/2^M they vanish mod 2^32.  *lean_out: the sums stay small enough for the 32-bit recombination
 * (`wide` != 0 asks for the values of the 64-bit one).  `bits` is M for GSW_RING_ZPOW2 and ignored for Fp. */
int gsw_diag_limb_start_values(int ring_kind, int bits, int k_slots, int wide, uint32_t *h_start4, int *lean_out);
/* Test aid, host only.  Over Fp the default kernel keeps bit i of a left-operand byte in place (byte values 0 / 2^c
 * instead of 0 / 1: a cheaper fragment) and the preparation scales the matching right-operand rows by 2^-c mod p.
 * Returns the mode for a contraction over k_slots K slots: 0 = no scaling (always for Z/2^M), 1 = c in {0, 1}, 2 = c in
 * {0..3}; -1 for a bad argument.  A limb sum then lies in [-128 u, 127 u] with u = k_slots * w / 32, w = 32 / 48 / 120 for
 * modes 0 / 1 / 2, and gsw_diag_limb_start_values returns the start values for that bound. */
int gsw_diag_bit_scale_mode(int ring_kind, int bits, int k_slots);

#ifdef __cplusplus
}
#endif
#endif
