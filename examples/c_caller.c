/* A plain C caller of the C ABI (include/gsw_hip.h): what a cgo / FFI binding does.
 * Builds two uniform compact ciphertexts over Fp (n given on the command line), multiplies them with
 * gsw_mul_gadget_batch and its NAND form, and prints the digests of the operands' product.
 *   gcc -std=c99 -Iinclude examples/c_caller.c -Ltfhe_amd -lgsw_hip -Wl,-rpath,$PWD/tfhe_amd -o examples/c_caller
 */
#include <stdio.h>
#include <stdlib.h>

#include "gsw_hip.h"

#define TRY(call)                                                     \
    do {                                                              \
        if ((call) != GSW_OK) {                                       \
            fprintf(stderr, "%s: %s\n", #call, gsw_last_error());     \
            return 1;                                                 \
        }                                                             \
    } while (0)

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4;
    gsw_ctx *ctx = NULL;
    TRY(gsw_ctx_create(&ctx, GSW_RING_FP_M31, 31, n, 31 * n, 0));
    const size_t elems = gsw_ctx_compact_elems(ctx);
    const int ld = gsw_ctx_ld(ctx), big_n = gsw_ctx_big_n(ctx);
    void *buf = NULL, *dig = NULL;
    TRY(gsw_malloc(ctx, 4 * elems * 4, &buf)); /* c1, c2, product, nand */
    TRY(gsw_malloc(ctx, 2 * sizeof(uint64_t), &dig));
    uint32_t *c1 = (uint32_t *)buf, *c2 = c1 + elems, *out = c2 + elems;
    /* uniform values in [0, p); then the padding columns are cleared by a round trip through the host */
    TRY(gsw_fill_uniform(ctx, 7, 1, 0, 2147483646u, 0, c1, 2 * elems));
    uint32_t *host = (uint32_t *)malloc(2 * elems * sizeof(uint32_t));
    if (!host) return 1;
    TRY(gsw_memcpy_d2h(ctx, host, c1, 2 * elems * 4));
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < big_n; ++i)
            for (int j = n + 1; j < ld; ++j) host[(size_t)k * elems + (size_t)i * ld + j] = 0;
    TRY(gsw_memcpy_h2d(ctx, c1, host, 2 * elems * 4));
    TRY(gsw_mul_gadget_batch(ctx, c1, elems, c2, elems, out, elems, 1, 0));
    TRY(gsw_mul_gadget_batch(ctx, c1, elems, c2, elems, out + elems, elems, 1, GSW_FLAG_NAND));
    TRY(gsw_digest_batch(ctx, out, elems, elems, 2, (uint64_t *)dig));
    uint64_t h[2];
    TRY(gsw_memcpy_d2h(ctx, h, dig, sizeof h));
    printf("%s n=%d N=%d ld=%d mult=%016llx nand=%016llx\n", gsw_version(), n, big_n, ld, (unsigned long long)h[0],
           (unsigned long long)h[1]);
    free(host);
    TRY(gsw_free(ctx, dig));
    TRY(gsw_free(ctx, buf));
    gsw_ctx_destroy(ctx);
    return 0;
}
