/*
 * gsw_oracle.h -- CPU restatement of the GSW arithmetic of n0900/tfhe.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker the HIP engine is
 * compared against and the CPU baseline that is timed beside it.  Nothing in
 * the product path (tfhe_amd/, the C-ABI library) may link, load or call it.
 *
 * PARITY STATUS: the Rust reference cannot be compiled in this environment
 * (no cargo/rustc), so this restatement is pinned only by the known-answer
 * tests and invariants the reference's own unit tests hold (field.rs:97-122,
 * pow2_ring.rs:199-261, helper.rs:111-164, sk.rs:37-44, gsw_impl.rs:150-219).
 * The reference has no test, vector or fixture for mult/add/mult_const/nand:
 * for those four operations parity is UNPINNED; they are checked here by the
 * definition (exact modular matrix product), by an independent Python big-int
 * restatement, and by decrypt-after-operation.
 *
 * Conventions: every ring element is a canonical uint32_t in [0, q); every
 * matrix is dense row-major.  (The reference stores nalgebra column-major
 * matrices of 8-byte elements, Fp in Montgomery form: field.rs:18,107-112.)
 */
#ifndef GSW_ORACLE_H
#define GSW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSWO_RING_FP_M31 0 /* F_p, p = 2^31 - 1        (field.rs:7-19)      */
#define GSWO_RING_ZPOW2 1  /* Z / 2^M, 1 <= M <= 32    (pow2_ring.rs:14-16) */

#define GSWO_FP_P 2147483647u

typedef struct {
    int kind;      /* GSWO_RING_*                                         */
    int ell;       /* R::Num_Bits: 31 for Fp, M for Zpow2<M>              */
    uint32_t mask; /* 2^M - 1 for Zpow2, p for Fp (p is all-ones too)     */
} gswo_ring;

/* ---- ring element arithmetic (field.rs:7-56, pow2_ring.rs:36-152) ---- */
int gswo_ring_init(gswo_ring *r, int kind, int log2q);
uint32_t gswo_from_u64(const gswo_ring *r, uint64_t x);
uint32_t gswo_add(const gswo_ring *r, uint32_t a, uint32_t b);
uint32_t gswo_sub(const gswo_ring *r, uint32_t a, uint32_t b);
uint32_t gswo_mul(const gswo_ring *r, uint32_t a, uint32_t b);
uint32_t gswo_neg(const gswo_ring *r, uint32_t a);
uint64_t gswo_max_u64(const gswo_ring *r);
/* Fp only: multiplicative inverse (ff::Field::invert, used by field.rs:68-78) */
uint32_t gswo_fp_inv(uint32_t a);
/* gsw_impl.rs:127-131 */
uint32_t gswo_is_zero_one(const gswo_ring *r, uint32_t x);

/* ---- linear algebra over the ring (nalgebra DMatrix ops) ---- */
/* out[rows x cols] = a[rows x inner] * b[inner x cols]  (gsw_impl.rs:31,107) */
void gswo_matmul(const gswo_ring *r, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t rows,
                 size_t inner, size_t cols);
uint32_t gswo_dot(const gswo_ring *r, const uint32_t *a, const uint32_t *b, size_t len);

/* ---- gadget helpers (gsw/helper.rs, gsw/mod.rs:58-62) ---- */
void gswo_gadget_vector(const gswo_ring *r, uint32_t *g /* ell */);
/* helper.rs:7-21: in[rows x cols] -> out[rows x cols*ell], bit k of column j at j*ell+k */
void gswo_bit_decomp_matrix(const gswo_ring *r, const uint32_t *in, uint32_t *out, size_t rows, size_t cols);
/* helper.rs:24-49: in[rows x cols*ell] -> out[rows x cols].  quirk != 0 follows the
 * reference exactly (bit i is set iff the entry == 1, then R::from(repr));
 * quirk == 0 is the linear inverse sum_k 2^k * entry mod q. */
void gswo_bit_decomp_inv_matrix(const gswo_ring *r, const uint32_t *in, uint32_t *out, size_t rows, size_t cols,
                                int quirk);
/* helper.rs:51-62 (the reference's flatten, quirk included), in place */
void gswo_flatten_matrix(const gswo_ring *r, uint32_t *mat, size_t rows, size_t cols_times_ell);
/* helper.rs:65-77 */
void gswo_powers_of_2(const gswo_ring *r, const uint32_t *b, size_t len, uint32_t *out /* len*ell */);

/* ---- keys (gsw/sk.rs:20-28, gsw/pk.rs:23-39) ---- */
void gswo_sk_new(const gswo_ring *r, const uint32_t *t, size_t n, uint32_t *s /* n+1 */, uint32_t *v /* (n+1)*ell */);
void gswo_pk_new(const gswo_ring *r, const uint32_t *rand_b /* m x n */, const uint32_t *e /* m */,
                 const uint32_t *t /* n */, size_t m, size_t n, uint32_t *b /* m */, uint32_t *pk /* m x (n+1) */);

/* ---- scheme, dense N x N form, N = ell*(n+1) (gsw/gsw_impl.rs:27-123) ---- */
/* rbits: N x m matrix of 0/1 bytes (the reference draws it with rnd_dmatrix(N, m, 0, 1)) */
void gswo_encrypt(const gswo_ring *r, size_t n, size_t m, const uint32_t *pk, const uint8_t *rbits, uint32_t msg,
                  uint32_t *out /* N x N */);
uint32_t gswo_decrypt(const gswo_ring *r, size_t n, const uint32_t *v, const uint32_t *c);
/* returns 0 and writes *out, or -1 for Fp (the reference panics: gsw_impl.rs:68-70) */
int gswo_mp_decrypt(const gswo_ring *r, size_t n, const uint32_t *v, const uint32_t *c, uint32_t *out);
void gswo_ct_add(const gswo_ring *r, const uint32_t *c1, const uint32_t *c2, uint32_t *out, size_t count);
void gswo_ct_mult_const(const gswo_ring *r, uint32_t *c, uint32_t k, size_t count);
void gswo_ct_mult(const gswo_ring *r, const uint32_t *c1, const uint32_t *c2, uint32_t *out, size_t big_n);
void gswo_ct_nand(const gswo_ring *r, const uint32_t *c1, const uint32_t *c2, uint32_t *out, size_t big_n);

/* ---- compact form: Cc = BitDecomp^-1_linear(C), N x (n+1) ----
 * For any C1, C2:  BitDecomp^-1_linear(C1*C2) = C1 * C2c, so with
 * C1 = BitDecomp(C1c) the product of gsw_impl.rs:107 collapses to
 * Pc = BitDecomp(C1c) * C2c, an (N x N binary) x (N x (n+1)) contraction. */
#define GSWO_FLAG_NAND 1u
void gswo_encrypt_compact(const gswo_ring *r, size_t n, size_t m, const uint32_t *pk, const uint8_t *rbits,
                          uint32_t msg, uint32_t *out /* N x (n+1) */);
void gswo_mul_gadget(const gswo_ring *r, size_t n, const uint32_t *c1c, const uint32_t *c2c, uint32_t *out,
                     unsigned flags);
/* rows [row0, row0+nrows) of the same product; out holds nrows x (n+1) */
void gswo_mul_gadget_rows(const gswo_ring *r, size_t n, const uint32_t *c1c, const uint32_t *c2c, uint32_t *out,
                          size_t row0, size_t nrows, unsigned flags);
/* batch of independent products, OpenMP over the batch (the CPU baseline) */
void gswo_mul_gadget_batch(const gswo_ring *r, size_t n, const uint32_t *c1c, size_t stride1, const uint32_t *c2c,
                           size_t stride2, uint32_t *out, size_t stride_out, size_t batch, unsigned flags,
                           int threads);
uint32_t gswo_decrypt_compact(const gswo_ring *r, size_t n, const uint32_t *s, const uint32_t *cc);
int gswo_mp_decrypt_compact(const gswo_ring *r, size_t n, const uint32_t *s, const uint32_t *cc, uint32_t *out);
/* <row, s> for one row of a compact ciphertext (noise measurement in tests) */
uint32_t gswo_compact_row_dot_s(const gswo_ring *r, size_t n, const uint32_t *s, const uint32_t *cc, size_t row);

/* ---- sampling (error_sampling.rs) on a counter-based generator ----
 * The reference draws from rand::rng() (thread-local, OS seeded), so no
 * stream can be reproduced; only ranges and distributions are restated. */
uint64_t gswo_rng_u64(uint64_t seed, uint64_t stream, uint64_t counter);
/* error_sampling.rs:20-24: uniform in [min, max] inclusive; returns -1 if max > P */
int gswo_fill_uniform(const gswo_ring *r, uint64_t seed, uint64_t stream, uint64_t min, uint64_t max, uint32_t *out,
                      size_t count);
/* same generator without the max <= P assertion (needed for q = 2^32 keys) */
void gswo_fill_uniform_unchecked(const gswo_ring *r, uint64_t seed, uint64_t stream, uint64_t min, uint64_t max,
                                 uint32_t *out, size_t count);
void gswo_fill_bits(uint64_t seed, uint64_t stream, uint8_t *out, size_t count);
/* error_sampling.rs:73-84: uniform in [0, P >> 15] */
void gswo_sample_naive(const gswo_ring *r, uint64_t seed, uint64_t stream, uint32_t *out, size_t count);
/* error_sampling.rs:32-71: discrete Gaussian, sigma = sig_num/sig_den, signed value mapped to +-R */
void gswo_sample_gaussian(const gswo_ring *r, uint64_t seed, uint64_t stream, double sigma, uint32_t *out,
                          size_t count);
/* signed integer samples of the same stream (for statistical tests) */
void gswo_sample_gaussian_i32(uint64_t seed, uint64_t stream, double sigma, int32_t *out, size_t count);

/* 64-bit digest of a u32 buffer (matches the device digest kernel) */
uint64_t gswo_digest(const uint32_t *x, size_t count);

#ifdef __cplusplus
}
#endif
#endif
