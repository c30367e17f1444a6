/*
 * gsw_oracle.c -- CPU restatement of the GSW arithmetic of n0900/tfhe.
 * TEST INFRASTRUCTURE ONLY; see gsw_oracle.h for scope and parity status
 * ("parity unpinned" for mult/add/mult_const/nand: the reference holds no
 * vectors for them).  Each function cites the reference lines it follows.
 */
#include "gsw_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ ring */

int gswo_ring_init(gswo_ring *r, int kind, int log2q) {
    if (kind == GSWO_RING_FP_M31) {
        /* field.rs:8: modulus 2147483647; field.rs:39: Num_Bits = Fp::NUM_BITS = 31 */
        r->kind = kind;
        r->ell = 31;
        r->mask = GSWO_FP_P;
        return 0;
    }
    if (kind == GSWO_RING_ZPOW2 && log2q >= 1 && log2q <= 32) {
        /* pow2_ring.rs:19 Num_Bits = M; pow2_ring.rs:39-48 mask = 2^M - 1 */
        r->kind = kind;
        r->ell = log2q;
        r->mask = log2q == 32 ? 0xFFFFFFFFu : ((1u << log2q) - 1u);
        return 0;
    }
    return -1;
}

/* value mod (2^31 - 1) for any 64-bit value */
static inline uint32_t fp_reduce64(uint64_t x) {
    x = (x & GSWO_FP_P) + (x >> 31); /* < 2^31 + 2^33 */
    x = (x & GSWO_FP_P) + (x >> 31); /* < 2^31 + 8    */
    x = (x & GSWO_FP_P) + (x >> 31); /* <= 2^31 - 1 + 1 */
    if (x >= GSWO_FP_P) x -= GSWO_FP_P;
    return (uint32_t)x;
}

/* From<u64>: Fp reduces mod p (relied on at helper.rs:125,151); Zpow2::new masks (pow2_ring.rs:57-59) */
uint32_t gswo_from_u64(const gswo_ring *r, uint64_t x) {
    if (r->kind == GSWO_RING_FP_M31) return fp_reduce64(x);
    return (uint32_t)(x & r->mask);
}

uint32_t gswo_add(const gswo_ring *r, uint32_t a, uint32_t b) {
    if (r->kind == GSWO_RING_FP_M31) {
        uint64_t s = (uint64_t)a + b;
        if (s >= GSWO_FP_P) s -= GSWO_FP_P;
        return (uint32_t)s;
    }
    return (a + b) & r->mask; /* pow2_ring.rs:75-81 */
}

uint32_t gswo_sub(const gswo_ring *r, uint32_t a, uint32_t b) {
    if (r->kind == GSWO_RING_FP_M31) return a >= b ? a - b : a + GSWO_FP_P - b;
    return (a - b) & r->mask; /* pow2_ring.rs:86-92 */
}

uint32_t gswo_mul(const gswo_ring *r, uint32_t a, uint32_t b) {
    if (r->kind == GSWO_RING_FP_M31) return fp_reduce64((uint64_t)a * b);
    return (uint32_t)(((uint64_t)a * b) & r->mask); /* pow2_ring.rs:97-103 */
}

uint32_t gswo_neg(const gswo_ring *r, uint32_t a) {
    if (r->kind == GSWO_RING_FP_M31) return a == 0 ? 0 : GSWO_FP_P - a;
    return (~a + 1u) & r->mask; /* pow2_ring.rs:105-112 */
}

/* field.rs:36-38 (max_value = P-1, field.rs:47-49); pow2_ring.rs:31-33,181-187 */
uint64_t gswo_max_u64(const gswo_ring *r) {
    if (r->kind == GSWO_RING_FP_M31) return (uint64_t)GSWO_FP_P - 1;
    return (uint64_t)r->mask;
}

uint32_t gswo_fp_inv(uint32_t a) {
    /* a^(p-2) mod p */
    gswo_ring r;
    gswo_ring_init(&r, GSWO_RING_FP_M31, 31);
    uint32_t result = 1, base = a;
    uint32_t e = GSWO_FP_P - 2;
    while (e) {
        if (e & 1) result = gswo_mul(&r, result, base);
        base = gswo_mul(&r, base, base);
        e >>= 1;
    }
    return result;
}

/* gsw_impl.rs:127-131: 1 iff from(max/4) <= x <= from(3*max/4), Ord on canonical values */
uint32_t gswo_is_zero_one(const gswo_ring *r, uint32_t x) {
    uint64_t mx = gswo_max_u64(r);
    uint32_t lo = gswo_from_u64(r, mx / 4);
    uint32_t hi = gswo_from_u64(r, 3 * mx / 4);
    return (x >= lo && x <= hi) ? 1u : 0u;
}

/* ------------------------------------------------------- linear algebra */

/* gsw_impl.rs:31,107: nalgebra `&DMatrix * &DMatrix` over R, here row-major */
void gswo_matmul(const gswo_ring *r, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t rows, size_t inner,
                 size_t cols) {
    uint64_t *acc = (uint64_t *)malloc(sizeof(uint64_t) * (cols ? cols : 1));
    for (size_t i = 0; i < rows; ++i) {
        memset(acc, 0, sizeof(uint64_t) * cols);
        for (size_t k = 0; k < inner; ++k) {
            uint64_t x = a[i * inner + k];
            if (x == 0) continue;
            const uint32_t *brow = b + k * cols;
            if (r->kind == GSWO_RING_FP_M31) {
                for (size_t c = 0; c < cols; ++c) {
                    uint64_t p = x * brow[c];                /* < 2^62 */
                    acc[c] += (p & GSWO_FP_P) + (p >> 31);   /* < 2^32 per term */
                }
            } else {
                for (size_t c = 0; c < cols; ++c) acc[c] += x * brow[c]; /* wraps mod 2^64, 2^M | 2^64 */
            }
        }
        for (size_t c = 0; c < cols; ++c) out[i * cols + c] = gswo_from_u64(r, acc[c]);
    }
    free(acc);
}

uint32_t gswo_dot(const gswo_ring *r, const uint32_t *a, const uint32_t *b, size_t len) {
    uint32_t acc = 0;
    for (size_t i = 0; i < len; ++i) acc = gswo_add(r, acc, gswo_mul(r, a[i], b[i]));
    return acc;
}

/* --------------------------------------------------------------- gadget */

/* gsw/mod.rs:58-62: g_l = R::from(1 << l) */
void gswo_gadget_vector(const gswo_ring *r, uint32_t *g) {
    for (int l = 0; l < r->ell; ++l) g[l] = gswo_from_u64(r, 1ull << l);
}

/* helper.rs:7-21 with to_le_bits_re (field.rs:32-34, pow2_ring.rs:21-30): little-endian bits */
void gswo_bit_decomp_matrix(const gswo_ring *r, const uint32_t *in, uint32_t *out, size_t rows, size_t cols) {
    const size_t ell = (size_t)r->ell;
    for (size_t i = 0; i < rows; ++i)
        for (size_t j = 0; j < cols; ++j) {
            uint32_t x = in[i * cols + j];
            for (size_t k = 0; k < ell; ++k) out[i * cols * ell + j * ell + k] = (x >> k) & 1u;
        }
}

/* helper.rs:24-49 */
void gswo_bit_decomp_inv_matrix(const gswo_ring *r, const uint32_t *in, uint32_t *out, size_t rows, size_t cols,
                                int quirk) {
    const size_t ell = (size_t)r->ell;
    for (size_t i = 0; i < rows; ++i)
        for (size_t j = 0; j < cols; ++j) {
            const uint32_t *chunk = in + i * cols * ell + j * ell;
            if (quirk) {
                uint64_t repr = 0;
                for (size_t k = 0; k < ell; ++k)
                    if (chunk[k] == 1u) repr |= 1ull << k; /* helper.rs:32-34 */
                out[i * cols + j] = gswo_from_u64(r, repr); /* helper.rs:37 */
            } else {
                uint32_t acc = 0;
                for (size_t k = 0; k < ell; ++k)
                    acc = gswo_add(r, acc, gswo_mul(r, chunk[k], gswo_from_u64(r, 1ull << k)));
                out[i * cols + j] = acc;
            }
        }
}

/* helper.rs:51-62 */
void gswo_flatten_matrix(const gswo_ring *r, uint32_t *mat, size_t rows, size_t cols_times_ell) {
    size_t cols = cols_times_ell / (size_t)r->ell;
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * rows * (cols ? cols : 1));
    gswo_bit_decomp_inv_matrix(r, mat, tmp, rows, cols, 1);
    gswo_bit_decomp_matrix(r, tmp, mat, rows, cols);
    free(tmp);
}

/* helper.rs:65-77: out[j*ell + k] = b_j * g_k */
void gswo_powers_of_2(const gswo_ring *r, const uint32_t *b, size_t len, uint32_t *out) {
    uint32_t g[32];
    gswo_gadget_vector(r, g);
    for (size_t j = 0; j < len; ++j)
        for (int k = 0; k < r->ell; ++k) out[j * (size_t)r->ell + (size_t)k] = gswo_mul(r, b[j], g[k]);
}

/* ----------------------------------------------------------------- keys */

/* sk.rs:20-28: s = (1, -t), v = PowersOf2(s) */
void gswo_sk_new(const gswo_ring *r, const uint32_t *t, size_t n, uint32_t *s, uint32_t *v) {
    s[0] = gswo_from_u64(r, 1);
    for (size_t i = 0; i < n; ++i) s[i + 1] = gswo_neg(r, t[i]);
    gswo_powers_of_2(r, s, n + 1, v);
}

/* pk.rs:23-39: b = B t + e, pk_matrix = [b | B] */
void gswo_pk_new(const gswo_ring *r, const uint32_t *rand_b, const uint32_t *e, const uint32_t *t, size_t m, size_t n,
                 uint32_t *b, uint32_t *pk) {
    for (size_t i = 0; i < m; ++i) {
        uint32_t acc = gswo_dot(r, rand_b + i * n, t, n);
        b[i] = gswo_add(r, acc, e[i]);
        pk[i * (n + 1)] = b[i];
        memcpy(pk + i * (n + 1) + 1, rand_b + i * n, sizeof(uint32_t) * n);
    }
}

/* --------------------------------------------------------- scheme, dense */

static void rbits_times_pk(const gswo_ring *r, size_t n, size_t m, const uint32_t *pk, const uint8_t *rbits,
                           uint32_t *prod /* N x (n+1) */) {
    const size_t big_n = (size_t)r->ell * (n + 1), w = n + 1;
    uint64_t *acc = (uint64_t *)malloc(sizeof(uint64_t) * w);
    for (size_t i = 0; i < big_n; ++i) {
        memset(acc, 0, sizeof(uint64_t) * w);
        for (size_t k = 0; k < m; ++k) {
            if (!rbits[i * m + k]) continue;
            const uint32_t *row = pk + k * w;
            for (size_t c = 0; c < w; ++c) acc[c] += row[c];
        }
        for (size_t c = 0; c < w; ++c) prod[i * w + c] = gswo_from_u64(r, acc[c]);
    }
    free(acc);
}

/* gsw_impl.rs:27-42 (default features): BitDecomp(R * pk) + message * I */
void gswo_encrypt(const gswo_ring *r, size_t n, size_t m, const uint32_t *pk, const uint8_t *rbits, uint32_t msg,
                  uint32_t *out) {
    const size_t big_n = (size_t)r->ell * (n + 1), w = n + 1;
    uint32_t *prod = (uint32_t *)malloc(sizeof(uint32_t) * big_n * w);
    rbits_times_pk(r, n, m, pk, rbits, prod);
    gswo_bit_decomp_matrix(r, prod, out, big_n, w);
    for (size_t i = 0; i < big_n; ++i) out[i * big_n + i] = gswo_add(r, out[i * big_n + i], msg);
    free(prod);
}

/* gsw_impl.rs:48-52: row Num_Bits-1 dotted with v, then the quarter-window test */
uint32_t gswo_decrypt(const gswo_ring *r, size_t n, const uint32_t *v, const uint32_t *c) {
    const size_t big_n = (size_t)r->ell * (n + 1);
    size_t i = (size_t)r->ell - 1;
    return gswo_is_zero_one(r, gswo_dot(r, c + i * big_n, v, big_n));
}

static uint32_t mp_peel(const gswo_ring *r, const uint32_t *slice /* ell entries of C*v */) {
    /* gsw_impl.rs:72-82 */
    const int ell = r->ell;
    uint64_t recovered_bits = 0;
    for (int i = 0; i < ell; ++i) {
        uint32_t entry = slice[ell - 1 - i];
        uint32_t recovered_exp = gswo_from_u64(r, recovered_bits << (ell - i - 1));
        uint32_t current_exp = gswo_sub(r, entry, recovered_exp);
        recovered_bits ^= (uint64_t)gswo_is_zero_one(r, current_exp) << i;
    }
    return gswo_from_u64(r, recovered_bits);
}

/* gsw_impl.rs:67-84 */
int gswo_mp_decrypt(const gswo_ring *r, size_t n, const uint32_t *v, const uint32_t *c, uint32_t *out) {
    if (r->kind == GSWO_RING_FP_M31) return -1; /* gsw_impl.rs:68-70 panics */
    const size_t big_n = (size_t)r->ell * (n + 1);
    uint32_t slice[32];
    for (int k = 0; k < r->ell; ++k) slice[k] = gswo_dot(r, c + (size_t)k * big_n, v, big_n);
    *out = mp_peel(r, slice);
    return 0;
}

/* gsw_impl.rs:87-94 */
void gswo_ct_add(const gswo_ring *r, const uint32_t *c1, const uint32_t *c2, uint32_t *out, size_t count) {
    for (size_t i = 0; i < count; ++i) out[i] = gswo_add(r, c1[i], c2[i]);
}

/* gsw_impl.rs:97-101 */
void gswo_ct_mult_const(const gswo_ring *r, uint32_t *c, uint32_t k, size_t count) {
    for (size_t i = 0; i < count; ++i) c[i] = gswo_mul(r, c[i], k);
}

/* gsw_impl.rs:104-111 (default features: no flatten) */
void gswo_ct_mult(const gswo_ring *r, const uint32_t *c1, const uint32_t *c2, uint32_t *out, size_t big_n) {
    gswo_matmul(r, c1, c2, out, big_n, big_n, big_n);
}

/* gsw_impl.rs:114-123: I - C1*C2 */
void gswo_ct_nand(const gswo_ring *r, const uint32_t *c1, const uint32_t *c2, uint32_t *out, size_t big_n) {
    gswo_matmul(r, c1, c2, out, big_n, big_n, big_n);
    for (size_t i = 0; i < big_n * big_n; ++i) out[i] = gswo_neg(r, out[i]);
    for (size_t i = 0; i < big_n; ++i) out[i * big_n + i] = gswo_add(r, out[i * big_n + i], 1u);
}

/* ------------------------------------------------------- compact form */

static inline uint32_t gadget_entry(const gswo_ring *r, size_t i, size_t j) {
    /* transpose of build_gadget_matrix (gsw/mod.rs:47-56): G[i][j] = 2^(i mod ell) [j == i div ell] */
    size_t ell = (size_t)r->ell;
    return (i / ell == j) ? gswo_from_u64(r, 1ull << (i % ell)) : 0u;
}

void gswo_encrypt_compact(const gswo_ring *r, size_t n, size_t m, const uint32_t *pk, const uint8_t *rbits,
                          uint32_t msg, uint32_t *out) {
    const size_t big_n = (size_t)r->ell * (n + 1), w = n + 1;
    rbits_times_pk(r, n, m, pk, rbits, out);
    for (size_t i = 0; i < big_n; ++i) {
        size_t j = i / (size_t)r->ell;
        out[i * w + j] = gswo_add(r, out[i * w + j], gswo_mul(r, msg, gadget_entry(r, i, j)));
    }
}

void gswo_mul_gadget_rows(const gswo_ring *r, size_t n, const uint32_t *c1c, const uint32_t *c2c, uint32_t *out,
                          size_t row0, size_t nrows, unsigned flags) {
    const size_t w = n + 1, ell = (size_t)r->ell;
    uint64_t *acc = (uint64_t *)malloc(sizeof(uint64_t) * w);
    for (size_t i = row0; i < row0 + nrows; ++i) {
        memset(acc, 0, sizeof(uint64_t) * w);
        for (size_t j = 0; j < w; ++j) {
            uint32_t x = c1c[i * w + j];
            while (x) {
                int k = __builtin_ctz(x);
                x &= x - 1;
                const uint32_t *row = c2c + (j * ell + (size_t)k) * w;
                for (size_t c = 0; c < w; ++c) acc[c] += row[c];
            }
        }
        uint32_t *o = out + (i - row0) * w;
        for (size_t c = 0; c < w; ++c) {
            uint32_t p = gswo_from_u64(r, acc[c]);
            if (flags & GSWO_FLAG_NAND) p = gswo_sub(r, gadget_entry(r, i, c), p);
            o[c] = p;
        }
    }
    free(acc);
}

void gswo_mul_gadget(const gswo_ring *r, size_t n, const uint32_t *c1c, const uint32_t *c2c, uint32_t *out,
                     unsigned flags) {
    gswo_mul_gadget_rows(r, n, c1c, c2c, out, 0, (size_t)r->ell * (n + 1), flags);
}

void gswo_mul_gadget_batch(const gswo_ring *r, size_t n, const uint32_t *c1c, size_t stride1, const uint32_t *c2c,
                           size_t stride2, uint32_t *out, size_t stride_out, size_t batch, unsigned flags,
                           int threads) {
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : omp_get_max_threads())
#endif
    for (long b = 0; b < (long)batch; ++b)
        gswo_mul_gadget(r, n, c1c + (size_t)b * stride1, c2c + (size_t)b * stride2, out + (size_t)b * stride_out, flags);
}

uint32_t gswo_compact_row_dot_s(const gswo_ring *r, size_t n, const uint32_t *s, const uint32_t *cc, size_t row) {
    return gswo_dot(r, cc + row * (n + 1), s, n + 1);
}

/* <C[i,:], v> = <Cc[i,:], s> because v = PowersOf2(s) (helper.rs:146-164 invariant) */
uint32_t gswo_decrypt_compact(const gswo_ring *r, size_t n, const uint32_t *s, const uint32_t *cc) {
    return gswo_is_zero_one(r, gswo_compact_row_dot_s(r, n, s, cc, (size_t)r->ell - 1));
}

int gswo_mp_decrypt_compact(const gswo_ring *r, size_t n, const uint32_t *s, const uint32_t *cc, uint32_t *out) {
    if (r->kind == GSWO_RING_FP_M31) return -1;
    uint32_t slice[32];
    for (int k = 0; k < r->ell; ++k) slice[k] = gswo_compact_row_dot_s(r, n, s, cc, (size_t)k);
    *out = mp_peel(r, slice);
    return 0;
}

/* ------------------------------------------------------------- sampling */

static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

uint64_t gswo_rng_u64(uint64_t seed, uint64_t stream, uint64_t counter) {
    return mix64(mix64(seed ^ (stream * 0xD1342543DE82EF95ull)) + counter);
}

static inline uint32_t uniform_from(uint64_t rnd, uint64_t min, uint64_t max) {
    uint64_t range = max - min + 1; /* <= 2^32 */
    return (uint32_t)(min + (((rnd >> 32) * range) >> 32));
}

void gswo_fill_uniform_unchecked(const gswo_ring *r, uint64_t seed, uint64_t stream, uint64_t min, uint64_t max,
                                 uint32_t *out, size_t count) {
    for (size_t i = 0; i < count; ++i)
        out[i] = gswo_from_u64(r, uniform_from(gswo_rng_u64(seed, stream, i), min, max));
}

/* error_sampling.rs:20-24: assert!(max <= P); R::from(rng.random_range(min..=max)) */
int gswo_fill_uniform(const gswo_ring *r, uint64_t seed, uint64_t stream, uint64_t min, uint64_t max, uint32_t *out,
                      size_t count) {
    if (max > GSWO_FP_P) return -1;
    gswo_fill_uniform_unchecked(r, seed, stream, min, max, out, count);
    return 0;
}

void gswo_fill_bits(uint64_t seed, uint64_t stream, uint8_t *out, size_t count) {
    /* 64 bits per draw, little-endian within the draw */
    for (size_t i = 0; i < count; ++i) out[i] = (uint8_t)((gswo_rng_u64(seed, stream, i >> 6) >> (i & 63)) & 1u);
}

/* error_sampling.rs:75-79: rnd_ring_elm(0, P >> 15) * NOISE_CONST (= 1) */
void gswo_sample_naive(const gswo_ring *r, uint64_t seed, uint64_t stream, uint32_t *out, size_t count) {
    gswo_fill_uniform_unchecked(r, seed, stream, 0, GSWO_FP_P >> 15, out, count);
}

/* Inverse-CDF sampler for the discrete Gaussian P(k) ~ exp(-k^2 / (2 sigma^2)).
 * The reference draws from prio 0.17's exact rational sampler
 * (error_sampling.rs:37-49,53-65); the distribution is the same, the stream is not. */
#define GAUSS_MAX_K 4096
static int gauss_table(double sigma, uint64_t *cum, int *kmax_out) {
    int kmax = (int)ceil(13.0 * sigma) + 1;
    if (kmax > GAUSS_MAX_K) kmax = GAUSS_MAX_K;
    int len = 2 * kmax + 1;
    long double total = 0.0L;
    for (int k = -kmax; k <= kmax; ++k) total += expl(-(long double)k * k / (2.0L * sigma * sigma));
    long double run = 0.0L;
    for (int idx = 0; idx < len; ++idx) {
        int k = idx - kmax;
        run += expl(-(long double)k * k / (2.0L * sigma * sigma)) / total;
        long double scaled = run * 18446744073709551616.0L;
        cum[idx] = scaled >= 18446744073709551615.0L ? 0xFFFFFFFFFFFFFFFFull : (uint64_t)scaled;
    }
    cum[len - 1] = 0xFFFFFFFFFFFFFFFFull;
    *kmax_out = kmax;
    return len;
}

void gswo_sample_gaussian_i32(uint64_t seed, uint64_t stream, double sigma, int32_t *out, size_t count) {
    uint64_t *cum = (uint64_t *)malloc(sizeof(uint64_t) * (2 * GAUSS_MAX_K + 1));
    int kmax;
    int len = gauss_table(sigma, cum, &kmax);
    for (size_t i = 0; i < count; ++i) {
        uint64_t u = gswo_rng_u64(seed, stream, i);
        int lo = 0, hi = len - 1;
        while (lo < hi) {
            int mid = (lo + hi) / 2;
            if (u <= cum[mid]) hi = mid; else lo = mid + 1;
        }
        out[i] = lo - kmax;
    }
    free(cum);
}

/* error_sampling.rs:53-65: Minus -> -(R::from(d) * 1), NoSign -> 0, Plus -> R::from(d) * 1 */
void gswo_sample_gaussian(const gswo_ring *r, uint64_t seed, uint64_t stream, double sigma, uint32_t *out,
                          size_t count) {
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (count ? count : 1));
    gswo_sample_gaussian_i32(seed, stream, sigma, tmp, count);
    for (size_t i = 0; i < count; ++i) {
        if (tmp[i] < 0) out[i] = gswo_neg(r, gswo_from_u64(r, (uint64_t)(-(int64_t)tmp[i])));
        else out[i] = gswo_from_u64(r, (uint64_t)tmp[i]);
    }
    free(tmp);
}

uint64_t gswo_digest(const uint32_t *x, size_t count) {
    uint64_t d = 0;
    for (size_t i = 0; i < count; ++i) d += ((uint64_t)x[i] + 1ull) * ((2ull * i + 1ull) * 0x9E3779B97F4A7C15ull);
    return d;
}
