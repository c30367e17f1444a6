// The reference's own unit tests (src/gsw/gsw_impl.rs:150-219, src/gsw/sk.rs:37-44) and the untested
// operations (mult / nand / add / mult_const, checked by decryption) run against the C++ mirror of
// `FheScheme` (include/gsw_hip.hpp) on a real GPU.  Exit code 0 = all passed.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "gsw_hip.hpp"

using namespace gsw;

static int failures = 0;
#define EXPECT(cond)                                                         \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
            ++failures;                                                      \
        }                                                                    \
    } while (0)

template <class F>
static bool throws_with(F &&f, const char *needle) {
    try {
        f();
    } catch (const std::runtime_error &e) {
        return std::string(e.what()).find(needle) != std::string::npos;
    }
    return false;
}

// gsw_impl.rs:209-219 test_inputs
template <class R>
static void test_inputs(GSW<R> &fhe) {
    auto keys = fhe.keygen();
    DMatrix enc0 = fhe.encrypt(keys.second, 0);
    EXPECT(fhe.decrypt(keys.first, enc0) == 0);
    DMatrix enc1 = fhe.encrypt(keys.second, 1);
    EXPECT(fhe.decrypt(keys.first, enc1) == 1);
}

static void sk_pk_invariant() {  // gsw_impl.rs:150-163: pk_matrix * s == e
    GSW<Fp> fhe(10, 10 * Fp::Num_Bits, Sampler::Naive, 1);
    auto keys = fhe.keygen();
    const GswSk &sk = keys.first;
    const GswPk &pk = keys.second;
    const int m = pk.pk_matrix.rows;
    for (int i = 0; i < m; ++i) {
        uint64_t acc = 0;
        for (int j = 0; j < pk.pk_matrix.cols; ++j) acc = (acc + (uint64_t)Fp::mul(pk.pk_matrix(i, j), sk.s[j])) % Fp::P;
        EXPECT(acc <= (Fp::P >> 15));  // the NaiveSampler's range (error_sampling.rs:75-79)
    }
}

static void test_v_decomp() {  // sk.rs:37-44
    GSW<Fp> fhe(5, 5 * 31);
    GswSk sk = fhe.make_sk({3, 1, 4, 1, 5});
    for (int i = 0; i < Fp::Num_Bits; ++i) EXPECT(sk.v[i] == (1u << i));
}

static void encryption_decryption_naive() {  // gsw_impl.rs:165-174
    GSW<Fp> fhe(10, 10 * Fp::Num_Bits, Sampler::Naive, 2);
    test_inputs(fhe);
}

static void encryption_decryption_discr_gaussian() {  // gsw_impl.rs:177-186
    GSW<Zpow2<30>> fhe(10, 10 * 30, Sampler::DiscrGaussian, 3);
    test_inputs(fhe);
}

static void encryption_decryption_pow_of_two() {  // gsw_impl.rs:189-207
    GSW<Zpow2<31>> fhe(10, 10 * 31, Sampler::DiscrGaussian, 4);
    auto keys = fhe.keygen();
    const uint32_t msgs[] = {0x12345678u, 0x7FFFFFFEu, 1u, 0u, 0x2AAAAAAAu};
    for (uint32_t msg : msgs) {
        DMatrix enc = fhe.encrypt(keys.second, msg);
        EXPECT(fhe.mp_decrypt(keys.first, enc) == msg);
    }
}

static void mult_nand_add_on_reference_format() {
    // the reference's dense ciphertexts (BitDecomp(R*A) + mu*I) through the dense product of
    // gsw_impl.rs:107; sigma = 1 errors (the NaiveSampler's noise does not survive a product)
    GSW<Fp> fhe(3, 93, Sampler::DiscrGaussian, 5);
    auto keys = fhe.keygen();
    for (int mu1 = 0; mu1 < 2; ++mu1)
        for (int mu2 = 0; mu2 < 2; ++mu2) {
            DMatrix c1 = fhe.encrypt(keys.second, mu1), c2 = fhe.encrypt(keys.second, mu2);
            EXPECT(fhe.decrypt(keys.first, fhe.mult(c1, c2)) == (uint32_t)(mu1 * mu2));
            EXPECT(fhe.decrypt(keys.first, fhe.nand(c1, c2)) == (uint32_t)(1 - mu1 * mu2));
        }
    DMatrix c1 = fhe.encrypt(keys.second, 1);
    DMatrix twice = fhe.add(c1, c1);
    DMatrix scaled = c1;
    fhe.mult_const(scaled, 2);
    EXPECT(twice == scaled);  // C + C == 2 * C entrywise (gsw_impl.rs:87-101)
}

static void gadget_path_equals_dense_product_of_flattened_operands() {
    GSW<Fp> fhe(3, 93, Sampler::DiscrGaussian, 6);
    auto keys = fhe.keygen();
    CompactCiphertext a = fhe.encrypt_compact(keys.second, 1), b = fhe.encrypt_compact(keys.second, 1);
    CompactCiphertext prod = fhe.mult(a, b), nd = fhe.nand(a, b);
    EXPECT(fhe.decrypt(keys.first, prod) == 1);
    EXPECT(fhe.decrypt(keys.first, nd) == 0);
    // gsw_impl.rs:107 on the dense forms, then the linear BitDecomp^-1, is the same compact ciphertext
    DMatrix dense = fhe.mult(fhe.to_dense(a), fhe.to_dense(b));
    CompactCiphertext via_dense = fhe.flatten_to_compact(dense);
    EXPECT(fhe.to_dense(via_dense) == fhe.to_dense(prod));
    EXPECT(fhe.decrypt(keys.first, fhe.to_dense(prod)) == 1);
}

static void nand_gates_over_a_pool() {
    // XOR(a, b) from four NAND gates (gsw_impl.rs:114-123) on slots of one resident pool
    GSW<Fp> fhe(10, 310, Sampler::DiscrGaussian, 8);
    auto keys = fhe.keygen();
    gsw_ctx *ctx = fhe.ctx();
    const size_t elems = gsw_ctx_compact_elems(ctx);
    for (uint32_t a = 0; a < 2; ++a)
        for (uint32_t b = 0; b < 2; ++b) {
            CompactCiphertext pool{DeviceBuffer(ctx, elems * 4 * 6), 6};
            check(gsw_memset_zero(ctx, pool.buf.get(), pool.buf.bytes()));
            std::vector<uint32_t> host(elems);
            const uint32_t bits[2] = {a, b};
            for (int k = 0; k < 2; ++k) {
                CompactCiphertext c = fhe.encrypt_compact(keys.second, bits[k]);
                c.buf.download(host.data(), elems * 4);
                check(gsw_memcpy_h2d(ctx, pool.buf.u32() + k * elems, host.data(), elems * 4));
            }
            fhe.eval_circuit(pool, {{GSW_GATE_NAND, 0, 1, 2}, {GSW_GATE_NAND, 0, 2, 3}, {GSW_GATE_NAND, 1, 2, 4},
                                    {GSW_GATE_NAND, 3, 4, 5}});
            EXPECT(fhe.decrypt(keys.first, pool, 2) == 1 - a * b);
            EXPECT(fhe.decrypt(keys.first, pool, 5) == (a ^ b));
        }
    CompactCiphertext pool{DeviceBuffer(ctx, elems * 4 * 2), 2};
    EXPECT(throws_with([&] { fhe.eval_circuit(pool, {{GSW_GATE_MULT, 0, 1, 1}}); }, "own operand"));
}

static void wire_round_trip_and_threshold_hooks() {
    GSW<Fp> alice(10, 310, Sampler::DiscrGaussian, 9), bob(10, 310, Sampler::DiscrGaussian, 10);
    auto keys = alice.keygen();
    // the public key travels to a second context, which encrypts and multiplies; the product travels back
    GswPk pk_b = bob.pk_from_wire(alice.to_wire(keys.second));
    EXPECT(pk_b.pk_matrix == keys.second.pk_matrix);
    CompactCiphertext one = bob.encrypt_compact(pk_b, 1), zero = bob.encrypt_compact(pk_b, 0);
    CompactCiphertext back = alice.ciphertext_from_wire(bob.to_wire(bob.nand(one, zero)));
    GswSk sk = alice.sk_from_wire(alice.to_wire(keys.first));
    EXPECT(sk.s == keys.first.s && sk.v == keys.first.v);
    EXPECT(alice.decrypt(sk, back) == 1);
    // a flipped payload byte is caught by the digest
    std::vector<uint8_t> wire = bob.to_wire(one);
    wire[40] ^= 1;
    EXPECT(throws_with([&] { alice.ciphertext_from_wire(wire); }, "digest"));
    // three additive shares of s: the partial decryptions recombine to the plain decryption
    std::vector<std::vector<uint32_t>> shares(3, std::vector<uint32_t>(sk.s.size()));
    uint64_t x = 88172645463325252ull;
    for (size_t j = 0; j < sk.s.size(); ++j) {
        uint64_t sum = 0;
        for (int p = 0; p < 2; ++p) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            shares[p][j] = (uint32_t)(x % Fp::P);
            sum += shares[p][j];
        }
        shares[2][j] = (uint32_t)(((uint64_t)sk.s[j] + 2 * Fp::P - sum % Fp::P) % Fp::P);
    }
    std::vector<uint32_t> partials;
    for (auto &sh : shares) partials.push_back(alice.part_dec(sh, back));
    EXPECT(alice.combine_partials(partials) == 1);
}

// keygen on the device gives the key the host-assembled path gives for the same draws, and the trait-level
// mult / nand of reference-format ciphertexts take the MFMA path when asked to at this size
static void device_keygen_and_dense_mfma_path() {
    GSW<Fp> a(10, 310, Sampler::DiscrGaussian, 21), b(10, 310, Sampler::DiscrGaussian, 21);
    auto dev = a.keygen_device();
    auto host = b.keygen();
    EXPECT(dev.first.t == host.first.t && dev.first.v == host.first.v);
    EXPECT(dev.second.pk_matrix.data == host.second.pk_matrix.data && dev.second.b == host.second.b);
    DMatrix c1 = a.encrypt(dev.second, 1), c0 = a.encrypt(dev.second, 0);
    DMatrix plain = a.mult(c1, c1);  // N = 341: the plain kernel by default
    EXPECT(a.last_dense_path() == 0 && a.decrypt(dev.first, plain) == 1);
    a.set_dense_mode(2);
    DMatrix fast = a.mult(c1, c1);
    EXPECT(a.last_dense_path() == 1 && fast.data == plain.data);
    EXPECT(a.decrypt(dev.first, a.mult(c1, c0)) == 0 && a.decrypt(dev.first, a.nand(c1, c1)) == 0);
    EXPECT(a.decrypt(dev.first, a.nand(c1, c0)) == 1 && a.last_dense_path() == 1);
    // a product of products -- the reference's default build never flattens (Cargo.toml:16-17), so the left operand has
    // arbitrary entries: the digit x digit MFMA products, against the plain kernel
    DMatrix pp = a.mult(fast, c1);
    EXPECT(a.last_dense_path() == 2);
    a.set_dense_mode(0);
    DMatrix pp_plain = a.mult(fast, c1);
    EXPECT(a.last_dense_path() == 0 && pp.data == pp_plain.data);
    a.set_dense_mode(2);
    // gates over a pool with the prepared-operand cache: NAND(1, 1) = 0, then NAND(0, 1) = 1
    CompactCiphertext pool{DeviceBuffer(a.ctx(), gsw_ctx_compact_elems(a.ctx()) * 4 * 4), 4};
    for (int sl = 0; sl < 2; ++sl) {
        CompactCiphertext one = a.encrypt_compact(dev.second, 1);
        std::vector<uint32_t> tmp(gsw_ctx_compact_elems(a.ctx()));
        one.buf.download(tmp.data(), tmp.size() * 4);
        check(gsw_memcpy_h2d(a.ctx(), pool.buf.u32() + (size_t)sl * gsw_ctx_compact_elems(a.ctx()), tmp.data(), tmp.size() * 4));
    }
    auto cache = a.make_cache(4);
    a.eval_circuit(pool, {{GSW_GATE_NAND, 0, 1, 2}, {GSW_GATE_NAND, 2, 1, 3}}, cache);
    EXPECT(a.decrypt(dev.first, pool, 2) == 0 && a.decrypt(dev.first, pool, 3) == 1);
    EXPECT(cache.valid[1] == 1 && cache.valid[2] == 0 && cache.valid[3] == 0);
}

static void panics_of_the_reference() {
    GSW<Fp> fhe(2, 62);
    EXPECT(throws_with([&] { fhe.mult(DMatrix(3, 3), DMatrix(4, 4)); }, "different sizes"));       // gsw_impl.rs:105-106
    EXPECT(throws_with([&] { fhe.add(DMatrix(3, 3), DMatrix(3, 4)); }, "different sizes"));        // gsw_impl.rs:88-89
    EXPECT(throws_with([&] { fhe.make_pk({1, 2, 3, 4}, 2, 2, {1}, {1, 2}); }, "Dimension Mismatch"));  // pk.rs:24-25
    auto keys = fhe.keygen();
    DMatrix c = fhe.encrypt(keys.second, 1);
    EXPECT(throws_with([&] { fhe.mp_decrypt(keys.first, c); }, "Only supported for pow2 rings"));  // gsw_impl.rs:69
    GSW<Zpow2<32>> wide(4, 128);
    EXPECT(throws_with([&] { wide.keygen(); }, "max <= P"));  // error_sampling.rs:21 via gsw_impl.rs:18
    GSW<Zpow2<32>> wide_ok(4, 128, Sampler::DiscrGaussian, 7, 0, /*check_max=*/false);
    test_inputs(wide_ok);
}

int main() {
    try {
        sk_pk_invariant();
        test_v_decomp();
        encryption_decryption_naive();
        encryption_decryption_discr_gaussian();
        encryption_decryption_pow_of_two();
        mult_nand_add_on_reference_format();
        gadget_path_equals_dense_product_of_flattened_operands();
        nand_gates_over_a_pool();
        wire_round_trip_and_threshold_hooks();
        device_keygen_and_dense_mfma_path();
        panics_of_the_reference();
    } catch (const std::exception &e) {
        std::printf("FAIL: unexpected exception: %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "%d check(s) failed\n" : "all host-API checks passed\n", failures);
    return failures ? 1 : 0;
}
