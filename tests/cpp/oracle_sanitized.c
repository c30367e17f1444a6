/* Driver for the CPU oracle (oracle/gsw_oracle.c) built with -fsanitize=address,undefined by
 * tests/test_host_sanitizers.py: runs every entry point once on small, exactly sized heap buffers and
 * cross-checks them against each other, so that an out-of-bounds access or undefined arithmetic in the
 * checker itself is found by the sanitizer. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsw_oracle.h"

static int failures = 0;
#define EXPECT(cond)                                         \
    do {                                                     \
        if (!(cond)) {                                       \
            printf("FAILED line %d: %s\n", __LINE__, #cond); \
            ++failures;                                      \
        }                                                    \
    } while (0)

static uint32_t *u32(size_t count) { return (uint32_t *)malloc(count ? count * 4 : 4); }

static void run_ring(int kind, int log2q, size_t n) {
    gswo_ring r;
    EXPECT(gswo_ring_init(&r, kind, log2q) == 0);
    const size_t ell = (size_t)r.ell, m = n * ell, width = n + 1, big_n = ell * width;
    const uint64_t maxv = gswo_max_u64(&r);

    /* keys: pk * s == e (gsw_impl.rs:150-163) */
    uint32_t *t = u32(n), *e = u32(m), *rand_b = u32(m * n), *s = u32(width), *v = u32(big_n), *b = u32(m), *pk = u32(m * width);
    gswo_fill_uniform_unchecked(&r, 7, 1, 0, maxv, t, n);
    gswo_sample_gaussian(&r, 7, 2, 1.0, e, m);
    gswo_fill_uniform_unchecked(&r, 7, 3, 0, maxv, rand_b, m * n);
    gswo_sk_new(&r, t, n, s, v);
    gswo_pk_new(&r, rand_b, e, t, m, n, b, pk);
    uint32_t *es = u32(m);
    gswo_matmul(&r, pk, s, es, m, width, 1);
    EXPECT(memcmp(es, e, m * 4) == 0);

    /* encrypt / decrypt, dense and compact; BitDecomp of the compact form is the dense form */
    uint8_t *rbits = (uint8_t *)malloc(big_n * m);
    uint32_t *dense[2], *compact[2];
    for (uint32_t bit = 0; bit < 2; ++bit) {
        gswo_fill_bits(7, 10 + bit, rbits, big_n * m);
        dense[bit] = u32(big_n * big_n);
        compact[bit] = u32(big_n * width);
        gswo_encrypt(&r, n, m, pk, rbits, bit, dense[bit]);
        gswo_encrypt_compact(&r, n, m, pk, rbits, bit, compact[bit]);
        EXPECT(gswo_decrypt(&r, n, v, dense[bit]) == bit);
        EXPECT(gswo_decrypt_compact(&r, n, s, compact[bit]) == bit);
        uint32_t *back = u32(big_n * width), *bits = u32(big_n * big_n);
        gswo_bit_decomp_inv_matrix(&r, dense[bit], back, big_n, width, 0);
        gswo_bit_decomp_matrix(&r, compact[bit], bits, big_n, width);
        /* the dense ciphertext carries the message on its diagonal, so it is BitDecomp(compact) only up to that */
        gswo_bit_decomp_inv_matrix(&r, bits, back, big_n, width, 1);
        EXPECT(memcmp(back, compact[bit], big_n * width * 4) == 0);
        free(back);
        free(bits);
    }
    /* products: dense product of flattened operands, linearly inverted, equals the compact product */
    uint32_t *f1 = u32(big_n * big_n), *f2 = u32(big_n * big_n), *prod = u32(big_n * big_n), *pc = u32(big_n * width),
             *pc2 = u32(big_n * width), *rows = u32(3 * width);
    gswo_bit_decomp_matrix(&r, compact[1], f1, big_n, width);
    gswo_bit_decomp_matrix(&r, compact[1], f2, big_n, width);
    gswo_ct_mult(&r, f1, f2, prod, big_n);
    gswo_bit_decomp_inv_matrix(&r, prod, pc, big_n, width, 0);
    gswo_mul_gadget(&r, n, compact[1], compact[1], pc2, 0);
    EXPECT(memcmp(pc, pc2, big_n * width * 4) == 0);
    EXPECT(gswo_decrypt_compact(&r, n, s, pc2) == 1);
    gswo_mul_gadget_rows(&r, n, compact[1], compact[1], rows, big_n - 3, 3, 0);
    EXPECT(memcmp(rows, pc2 + (big_n - 3) * width, 3 * width * 4) == 0);
    gswo_mul_gadget_batch(&r, n, compact[1], 0, compact[0], 0, pc, 0, 1, GSWO_FLAG_NAND, 2);
    EXPECT(gswo_decrypt_compact(&r, n, s, pc) == 1);
    gswo_ct_nand(&r, f1, f2, prod, big_n);
    gswo_flatten_matrix(&r, prod, big_n, big_n);
    gswo_ct_add(&r, f1, f2, prod, big_n * big_n);
    gswo_ct_mult_const(&r, prod, gswo_from_u64(&r, 3), big_n * big_n);
    uint32_t msg = 0;
    const int rc = gswo_mp_decrypt_compact(&r, n, s, compact[1], &msg);
    EXPECT(kind == GSWO_RING_FP_M31 ? rc != 0 : (rc == 0 && msg == 1));
    EXPECT((gswo_mp_decrypt(&r, n, v, dense[1], &msg) != 0) == (kind == GSWO_RING_FP_M31));
    (void)gswo_compact_row_dot_s(&r, n, s, compact[0], big_n - 1);
    uint32_t *pw = u32(big_n), *g = u32(ell);
    gswo_powers_of_2(&r, s, width, pw);
    gswo_gadget_vector(&r, g);
    EXPECT(memcmp(pw, v, big_n * 4) == 0 && g[0] == gswo_from_u64(&r, 1));
    (void)gswo_digest(pc, big_n * width);
    (void)gswo_dot(&r, s, s, width);
    gswo_sample_naive(&r, 1, 1, e, m);
    int32_t *gi = (int32_t *)malloc(16 * 4);
    gswo_sample_gaussian_i32(1, 1, 3.2, gi, 16);
    free(gi);
    EXPECT(gswo_fill_uniform(&r, 1, 1, 0, 0x100000000ull, e, 1) != 0); /* max > P is refused (error_sampling.rs:21) */

    free(t); free(e); free(rand_b); free(s); free(v); free(b); free(pk); free(es); free(rbits);
    free(dense[0]); free(dense[1]); free(compact[0]); free(compact[1]);
    free(f1); free(f2); free(prod); free(pc); free(pc2); free(rows); free(pw); free(g);
}

int main(void) {
    run_ring(GSWO_RING_FP_M31, 31, 3);
    run_ring(GSWO_RING_ZPOW2, 32, 2);
    run_ring(GSWO_RING_ZPOW2, 30, 3);
    gswo_ring fp;
    gswo_ring_init(&fp, GSWO_RING_FP_M31, 31);
    EXPECT(gswo_mul(&fp, gswo_fp_inv(12345), 12345) == 1);
    EXPECT(gswo_from_u64(&fp, 0xFFFFFFFFFFFFFFFFull) < GSWO_FP_P && gswo_neg(&fp, 0) == 0);
    EXPECT(gswo_add(&fp, GSWO_FP_P - 1, GSWO_FP_P - 1) == GSWO_FP_P - 2 && gswo_sub(&fp, 0, 1) == GSWO_FP_P - 1);
    printf(failures ? "oracle_sanitized: %d FAILURES\n" : "oracle_sanitized: ok\n", failures);
    return failures ? 1 : 0;
}
