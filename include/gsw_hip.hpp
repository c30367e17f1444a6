// gsw_hip.hpp -- C++ host-side mirror of the reference's `FheScheme<R>` for `GSW<R, T>`
// (src/gsw/mod.rs:12-38, src/gsw/gsw_impl.rs:12-125 of n0900/tfhe) on top of the C ABI in gsw_hip.h.
//
// Same method names, argument meaning and error behaviour as the Rust trait: where the reference
// panics (size mismatch gsw_impl.rs:88-89,105-106; mp_decrypt over Fp :69; max <= P
// error_sampling.rs:21; dimension mismatch pk.rs:24-25) these methods throw std::runtime_error
// with the same message.  All arithmetic runs in the HIP library; there is no CPU path.
//
// Ciphertext = DMatrix<R>, dense N x N, COLUMN-MAJOR like nalgebra (src/field.rs:107-112), holding
// canonical values (the reference's `Fp` is Montgomery form in memory: its shim converts with
// `Into<u64>` / `Fp::from`, field.rs:52-56).  `CompactCiphertext` is the device-resident
// N x (n+1) form the throughput path works on; `GSW::flatten_to_compact` / `to_dense` convert.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gsw_hip.h"

namespace gsw {

inline void check(int rc) {
    if (rc != GSW_OK) throw std::runtime_error(gsw_last_error());
}

// ---- ring elements (RingElement, src/lib.rs:15-29) -------------------------------------------
struct Fp {  // src/field.rs:7-56
    static constexpr int kind = GSW_RING_FP_M31;
    static constexpr int log2q = 31;
    static constexpr uint64_t P = 2147483647ull;
    static constexpr int Num_Bits = 31;
    static uint64_t max_u64() { return P - 1; }
    static uint32_t from_u64(uint64_t x) { return (uint32_t)(x % P); }
    static uint32_t neg(uint32_t a) { return a == 0 ? 0 : (uint32_t)(P - a); }
    static uint32_t mul(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % P); }
};

template <int M>
struct Zpow2 {  // src/pow2_ring.rs:14-188
    static_assert(M >= 1 && M <= 32, "the engine stores ring elements in 32 bits");
    static constexpr int kind = GSW_RING_ZPOW2;
    static constexpr int log2q = M;
    static constexpr int Num_Bits = M;
    static constexpr uint32_t mask = M == 32 ? 0xFFFFFFFFu : ((1u << (M % 32)) - 1u);
    static uint64_t max_u64() { return mask; }
    static uint32_t from_u64(uint64_t x) { return (uint32_t)x & mask; }
    static uint32_t neg(uint32_t a) { return (~a + 1u) & mask; }
    static uint32_t mul(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)a * b) & mask; }
};

// ---- a minimal column-major matrix standing for nalgebra's DMatrix<R> ------------------------
struct DMatrix {
    int rows = 0, cols = 0;
    std::vector<uint32_t> data;  // column-major: (r, c) at c * rows + r
    DMatrix() = default;
    DMatrix(int r, int c) : rows(r), cols(c), data((size_t)r * c, 0u) {}
    uint32_t &operator()(int r, int c) { return data[(size_t)c * rows + r]; }
    uint32_t operator()(int r, int c) const { return data[(size_t)c * rows + r]; }
    bool operator==(const DMatrix &o) const { return rows == o.rows && cols == o.cols && data == o.data; }
    std::vector<uint32_t> row_major() const {
        std::vector<uint32_t> out((size_t)rows * cols);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) out[(size_t)r * cols + c] = (*this)(r, c);
        return out;
    }
    static DMatrix from_row_major(int r, int c, const std::vector<uint32_t> &src) {
        DMatrix m(r, c);
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < c; ++j) m(i, j) = src[(size_t)i * c + j];
        return m;
    }
};

struct GswSk {  // src/gsw/sk.rs:13-17
    std::vector<uint32_t> t, s, v;
};
struct GswPk {  // src/gsw/pk.rs:13-16
    std::vector<uint32_t> b;
    DMatrix pk_matrix;  // m x (n + 1)
};

enum class Sampler { Naive, DiscrGaussian };  // error_sampling.rs:73-84 / :32-71 (sigma = 1)

// RAII device block
class DeviceBuffer {
  public:
    DeviceBuffer(gsw_ctx *ctx, size_t bytes) : ctx_(ctx), bytes_(bytes) { check(gsw_malloc(ctx, bytes, &ptr_)); }
    ~DeviceBuffer() { if (ptr_) gsw_free(ctx_, ptr_); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    DeviceBuffer(DeviceBuffer &&o) noexcept : ctx_(o.ctx_), ptr_(o.ptr_), bytes_(o.bytes_) { o.ptr_ = nullptr; }
    void *get() const { return ptr_; }
    uint32_t *u32() const { return static_cast<uint32_t *>(ptr_); }
    size_t bytes() const { return bytes_; }
    void upload(const void *src, size_t n) { check(gsw_memcpy_h2d(ctx_, ptr_, src, n)); }
    void download(void *dst, size_t n) const { check(gsw_memcpy_d2h(ctx_, dst, ptr_, n)); }

  private:
    gsw_ctx *ctx_;
    void *ptr_ = nullptr;
    size_t bytes_;
};

// device-resident compact ciphertext(s): N x (n + 1), leading dimension gsw_ctx_ld()
struct CompactCiphertext {
    DeviceBuffer buf;
    int batch;
};

template <class R>
class GSW {
  public:
    // GSW::<R, T> { n, m, err_sampling, .. } (src/gsw/mod.rs:29-38)
    GSW(int n, int m, Sampler sampler = Sampler::DiscrGaussian, uint64_t seed = 0, int device = 0,
        bool check_max = true)
        : n_(n), m_(m), sampler_(sampler), seed_(seed), check_max_(check_max) {
        check(gsw_ctx_create(&ctx_, R::kind, R::log2q, n, m, device));
        big_n_ = gsw_ctx_big_n(ctx_);
    }
    ~GSW() { gsw_ctx_destroy(ctx_); }
    GSW(const GSW &) = delete;
    GSW &operator=(const GSW &) = delete;

    gsw_ctx *ctx() const { return ctx_; }
    int big_n() const { return big_n_; }

    // ---- FheScheme ----
    std::pair<GswSk, GswPk> keygen() {  // gsw_impl.rs:17-25
        const uint64_t mx = R::max_u64();
        std::vector<uint32_t> t = uniform(0, mx, (size_t)n_);
        GswSk sk = make_sk(t);
        std::vector<uint32_t> err = errors((size_t)m_);
        std::vector<uint32_t> rand_b = uniform(0, mx, (size_t)m_ * n_);  // row-major m x n
        return {sk, make_pk(rand_b, m_, n_, err, t)};
    }

    // keygen with the samples, b = B t + e, pk_matrix and its prepared form made on the device (gsw_keygen_device):
    // only t crosses to the host; the public key comes back for the host-side GswPk
    std::pair<GswSk, GswPk> keygen_device() {
        std::vector<uint32_t> t((size_t)n_);
        const int ld = gsw_ctx_ld(ctx_);
        DeviceBuffer d_pk(ctx_, (size_t)m_ * ld * 4), prepared(ctx_, gsw_ctx_pk_prepared_bytes(ctx_));
        const uint64_t first = draw_ + 1;
        draw_ += 3;  // t, e, B
        check(gsw_keygen_device(ctx_, seed_, first, sampler_ == Sampler::Naive ? GSW_SAMPLER_NAIVE : GSW_SAMPLER_GAUSSIAN,
                                1.0, check_max_ ? 0 : 1, t.data(), d_pk.u32(), prepared.get()));
        std::vector<uint32_t> padded((size_t)m_ * ld);
        d_pk.download(padded.data(), padded.size() * 4);
        GswPk pk;
        pk.pk_matrix = DMatrix(m_, n_ + 1);
        pk.b.resize(m_);
        for (int i = 0; i < m_; ++i) {
            pk.b[i] = padded[(size_t)i * ld];
            for (int j = 0; j <= n_; ++j) pk.pk_matrix(i, j) = padded[(size_t)i * ld + j];
        }
        return {make_sk(t), pk};
    }

    GswSk make_sk(const std::vector<uint32_t> &t) const {  // sk.rs:20-28
        GswSk sk;
        sk.t = t;
        sk.s.push_back(R::from_u64(1));
        for (uint32_t x : t) sk.s.push_back(R::neg(x));
        for (uint32_t x : sk.s)  // powers_of_2 (helper.rs:65-77) with build_gadget_vector (mod.rs:58-62)
            for (int k = 0; k < R::Num_Bits; ++k) sk.v.push_back(R::mul(x, R::from_u64(1ull << k)));
        return sk;
    }

    GswPk make_pk(const std::vector<uint32_t> &rand_b, int rows, int cols, const std::vector<uint32_t> &e,
                  const std::vector<uint32_t> &t) {  // pk.rs:23-39: b = B t + e, pk = [b | B]
        if (cols != (int)t.size())
            throw std::runtime_error("Dimension Mismatch ncols " + std::to_string(cols) + ", " + std::to_string(t.size()));
        if (rows != (int)e.size())
            throw std::runtime_error("Dimension Mismatch nrows " + std::to_string(rows) + ", " + std::to_string(e.size()));
        DeviceBuffer d_b(ctx_, (size_t)rows * cols * 4), d_e(ctx_, (size_t)rows * 4), d_bt(ctx_, (size_t)rows * 4);
        d_b.upload(rand_b.data(), rand_b.size() * 4);
        d_e.upload(e.data(), e.size() * 4);
        check(gsw_matvec(ctx_, d_b.u32(), rows, cols, cols, t.data(), d_bt.u32()));
        check(gsw_add(ctx_, d_bt.u32(), d_e.u32(), d_bt.u32(), (size_t)rows));
        GswPk pk;
        pk.b.resize(rows);
        d_bt.download(pk.b.data(), (size_t)rows * 4);
        pk.pk_matrix = DMatrix(rows, cols + 1);
        for (int i = 0; i < rows; ++i) {
            pk.pk_matrix(i, 0) = pk.b[i];
            for (int j = 0; j < cols; ++j) pk.pk_matrix(i, j + 1) = rand_b[(size_t)i * cols + j];
        }
        return pk;
    }

    // gsw_impl.rs:27-42: returns the reference's dense matrix BitDecomp(R * pk) + message * I
    DMatrix encrypt(const GswPk &pk, uint32_t message) {
        CompactCiphertext x = encrypt_compact(pk, 0);  // R * pk in compact form (message 0)
        DMatrix c = to_dense(x, 0);
        for (int i = 0; i < c.cols; ++i) c(i, i) = ring_add(c(i, i), message);  // :34-36
#ifdef GSW_USE_FLATTEN  // the reference's cargo feature use_flatten (Cargo.toml:16-17): gsw_impl.rs:38-39
        flatten_matrix(c);
#endif
        return c;
    }

    // the flattened form of the same ciphertext, kept on the device: R * pk + message * G
    CompactCiphertext encrypt_compact(const GswPk &pk, uint32_t message) {
        DeviceBuffer prepared(ctx_, gsw_ctx_pk_prepared_bytes(ctx_));
        std::vector<uint32_t> pk_rm = pk.pk_matrix.row_major();
        check(gsw_prepare_pk(ctx_, pk_rm.data(), prepared.get()));
        DeviceBuffer msg(ctx_, 4);
        msg.upload(&message, 4);
        CompactCiphertext out{DeviceBuffer(ctx_, gsw_ctx_compact_elems(ctx_) * 4), 1};
        // rnd_dmatrix(N, m, 0, 1) of gsw_impl.rs:30 is drawn inside the call by the seeded device generator
        check(gsw_encrypt_batch_seeded(ctx_, prepared.get(), seed_, ++draw_, msg.u32(), out.buf.u32(),
                                       gsw_ctx_compact_elems(ctx_), nullptr, 1));
        check(gsw_sync(ctx_));
        return out;
    }

    uint32_t decrypt(const GswSk &sk, const DMatrix &c) {  // gsw_impl.rs:48-52
        require_square(c);
        DeviceBuffer d(ctx_, c.data.size() * 4);
        std::vector<uint32_t> rm = c.row_major();
        d.upload(rm.data(), rm.size() * 4);
        uint32_t bit = 0;
        check(gsw_decrypt_dense(ctx_, sk.v.data(), d.u32(), &bit));
        return bit;
    }
    uint32_t decrypt(const GswSk &sk, const CompactCiphertext &c, int b = 0) {
        uint32_t bit = 0;
        check(gsw_decrypt_batch(ctx_, sk.s.data(), c.buf.u32() + (size_t)b * gsw_ctx_compact_elems(ctx_), 0, 1, &bit,
                                nullptr));
        return bit;
    }

    uint32_t mp_decrypt(const GswSk &sk, const DMatrix &c) {  // gsw_impl.rs:67-84
        if (R::kind == GSW_RING_FP_M31) throw std::runtime_error("Only supported for pow2 rings!");
        CompactCiphertext cc = flatten_to_compact(c);  // <C[k,:], v> = <BitDecomp^-1_lin(C)[k,:], s>
        uint32_t msg = 0;
        check(gsw_mp_decrypt_batch(ctx_, sk.s.data(), cc.buf.u32(), 0, 1, &msg));
        return msg;
    }

    DMatrix add(const DMatrix &c1, const DMatrix &c2) {  // gsw_impl.rs:87-94
        same_size(c1, c2);
        DeviceBuffer a(ctx_, c1.data.size() * 4), b(ctx_, c2.data.size() * 4);
        a.upload(c1.data.data(), c1.data.size() * 4);  // elementwise: the storage order does not matter
        b.upload(c2.data.data(), c2.data.size() * 4);
        check(gsw_add(ctx_, a.u32(), b.u32(), a.u32(), c1.data.size()));
        DMatrix out(c1.rows, c1.cols);
        a.download(out.data.data(), out.data.size() * 4);
#ifdef GSW_USE_FLATTEN  // gsw_impl.rs:91-92
        flatten_matrix(out);
#endif
        return out;
    }

    void mult_const(DMatrix &c, uint32_t constant) {  // gsw_impl.rs:97-101 (in place)
        DeviceBuffer a(ctx_, c.data.size() * 4);
        a.upload(c.data.data(), c.data.size() * 4);
        check(gsw_mult_const(ctx_, a.u32(), constant, c.data.size()));
        a.download(c.data.data(), c.data.size() * 4);
#ifdef GSW_USE_FLATTEN  // gsw_impl.rs:99-100
        flatten_matrix(c);
#endif
    }

    // flatten_matrix (helper.rs:58-62), computed on the GPU, in place on a host matrix
    void flatten_matrix(DMatrix &c) {
        if (c.cols % R::Num_Bits) throw std::runtime_error("flatten needs a column count that is a multiple of Num_Bits");
        DeviceBuffer d(ctx_, c.data.size() * 4);
        std::vector<uint32_t> rm = c.row_major();
        d.upload(rm.data(), rm.size() * 4);
        check(gsw_flatten(ctx_, d.u32(), c.rows, c.cols / R::Num_Bits));
        d.download(rm.data(), rm.size() * 4);
        c = DMatrix::from_row_major(c.rows, c.cols, rm);
    }

    // A left operand in the reference's own ciphertext format (0 / 1 off the diagonal) takes the int8-MFMA
    // contraction, anything else the plain N^3 kernel; set_dense_mode / last_dense_path expose the choice.
    DMatrix mult(const DMatrix &c1, const DMatrix &c2) { return dense_product(c1, c2, false); }  // gsw_impl.rs:104-111
    DMatrix nand(const DMatrix &c1, const DMatrix &c2) { return dense_product(c1, c2, true); }   // gsw_impl.rs:114-123
    void set_dense_mode(int mode) { check(gsw_ctx_set_dense_fast(ctx_, mode)); }  // 0 never, 1 from 1024 rows, 2 always
    int last_dense_path() const { return gsw_ctx_last_dense_path(ctx_); }          // 0 plain kernel, 1 MFMA

    // the throughput path on device-resident flattened ciphertexts: Pc = G^-1(C1c) * C2c
    CompactCiphertext mult(const CompactCiphertext &c1, const CompactCiphertext &c2) { return gadget(c1, c2, 0); }
    CompactCiphertext nand(const CompactCiphertext &c1, const CompactCiphertext &c2) {
        return gadget(c1, c2, GSW_FLAG_NAND);
    }

    // a list of mult / nand / add / mult_const gates over the entries of `pool` (a batch used as slots),
    // with the meaning of applying them in order; independent products share one batched launch
    void eval_circuit(CompactCiphertext &pool, const std::vector<gsw_gate> &gates) {
        check(gsw_eval_circuit(ctx_, pool.buf.u32(), gsw_ctx_compact_elems(ctx_), (uint32_t)pool.batch, gates.data(),
                               gates.size()));
    }

    // the same with a cache of prepared right operands beside the pool (one entry and one flag per slot)
    struct PreparedCache {
        DeviceBuffer buf;
        std::vector<uint8_t> valid;
    };
    PreparedCache make_cache(int slots) {
        return PreparedCache{DeviceBuffer(ctx_, gsw_ctx_prepared_bytes(ctx_) * (size_t)slots), std::vector<uint8_t>((size_t)slots, 0)};
    }
    void eval_circuit(CompactCiphertext &pool, const std::vector<gsw_gate> &gates, PreparedCache &cache) {
        if ((int)cache.valid.size() != pool.batch) throw std::runtime_error("one cache entry per pool slot");
        check(gsw_eval_circuit_cached(ctx_, pool.buf.u32(), gsw_ctx_compact_elems(ctx_), (uint32_t)pool.batch,
                                      cache.buf.get(), cache.valid.data(), gates.data(), gates.size()));
    }

    // ---- wire format (layout in gsw_hip.h; the reference has no serialisation) ----
    std::vector<uint8_t> to_wire(const CompactCiphertext &c) {
        std::vector<uint8_t> out(gsw_wire_bytes(ctx_, GSW_WIRE_CIPHERTEXT, c.batch));
        check(gsw_wire_write_compact(ctx_, c.buf.u32(), gsw_ctx_compact_elems(ctx_), c.batch, out.data(), out.size()));
        return out;
    }
    CompactCiphertext ciphertext_from_wire(const std::vector<uint8_t> &wire) {
        int count = 0;
        check(gsw_wire_read_compact(ctx_, wire.data(), wire.size(), nullptr, 0, 0, &count));
        const size_t elems = gsw_ctx_compact_elems(ctx_);
        CompactCiphertext out{DeviceBuffer(ctx_, elems * 4 * (size_t)(count > 0 ? count : 1)), count};
        check(gsw_wire_read_compact(ctx_, wire.data(), wire.size(), out.buf.u32(), elems, count, &count));
        return out;
    }
    std::vector<uint8_t> to_wire(const GswPk &pk) {
        std::vector<uint8_t> out(gsw_wire_bytes(ctx_, GSW_WIRE_PUBLIC_KEY, 1));
        std::vector<uint32_t> rm = pk.pk_matrix.row_major();
        check(gsw_wire_write_host(ctx_, GSW_WIRE_PUBLIC_KEY, rm.data(), out.data(), out.size()));
        return out;
    }
    GswPk pk_from_wire(const std::vector<uint8_t> &wire) {
        std::vector<uint32_t> rm((size_t)m_ * (n_ + 1));
        check(gsw_wire_read_host(ctx_, GSW_WIRE_PUBLIC_KEY, wire.data(), wire.size(), rm.data()));
        GswPk pk;
        pk.pk_matrix = DMatrix::from_row_major(m_, n_ + 1, rm);
        pk.b.resize(m_);
        for (int i = 0; i < m_; ++i) pk.b[i] = rm[(size_t)i * (n_ + 1)];
        return pk;
    }
    std::vector<uint8_t> to_wire(const GswSk &sk) {
        std::vector<uint8_t> out(gsw_wire_bytes(ctx_, GSW_WIRE_SECRET_KEY, 1));
        check(gsw_wire_write_host(ctx_, GSW_WIRE_SECRET_KEY, sk.t.data(), out.data(), out.size()));
        return out;
    }
    GswSk sk_from_wire(const std::vector<uint8_t> &wire) {
        std::vector<uint32_t> t((size_t)n_);
        check(gsw_wire_read_host(ctx_, GSW_WIRE_SECRET_KEY, wire.data(), wire.size(), t.data()));
        return make_sk(t);
    }

    // ---- threshold hooks: additive shares of sk.s (tfhe_gsw_fp.rs:14-18), part_dec (a todo!() at :27-29) ----
    uint32_t part_dec(const std::vector<uint32_t> &share, const CompactCiphertext &c, int b = 0) {
        if ((int)share.size() != n_ + 1) throw std::runtime_error("a share of sk.s has n + 1 entries");
        uint32_t partial = 0;
        check(gsw_partial_decrypt_batch(ctx_, share.data(), c.buf.u32() + (size_t)b * gsw_ctx_compact_elems(ctx_), 0, 1,
                                        &partial));
        return partial;
    }
    uint32_t combine_partials(const std::vector<uint32_t> &partials) {
        uint32_t bit = 0;
        check(gsw_combine_partials(ctx_, partials.data(), (int)partials.size(), 1, &bit));
        return bit;
    }

    // ---- conversions between the reference's dense form and the compact form ----
    // BitDecomp^-1 (linear) of a dense matrix, uploaded as a compact ciphertext
    CompactCiphertext flatten_to_compact(const DMatrix &c) {
        require_square(c);
        DeviceBuffer d(ctx_, c.data.size() * 4);
        std::vector<uint32_t> rm = c.row_major();
        d.upload(rm.data(), rm.size() * 4);
        CompactCiphertext out{DeviceBuffer(ctx_, gsw_ctx_compact_elems(ctx_) * 4), 1};
        check(gsw_memset_zero(ctx_, out.buf.get(), out.buf.bytes()));
        check(gsw_bit_decomp_inv(ctx_, d.u32(), big_n_, n_ + 1, out.buf.u32(), gsw_ctx_ld(ctx_), 0));
        check(gsw_sync(ctx_));
        return out;
    }
    // BitDecomp of a compact ciphertext: the reference's N x N matrix of the flattened ciphertext
    DMatrix to_dense(const CompactCiphertext &c, int b = 0) {
        DeviceBuffer d(ctx_, (size_t)big_n_ * big_n_ * 4);
        check(gsw_bit_decomp(ctx_, c.buf.u32() + (size_t)b * gsw_ctx_compact_elems(ctx_), big_n_, n_ + 1,
                             gsw_ctx_ld(ctx_), d.u32()));
        std::vector<uint32_t> rm((size_t)big_n_ * big_n_);
        d.download(rm.data(), rm.size() * 4);
        return DMatrix::from_row_major(big_n_, big_n_, rm);
    }

  private:
    static uint32_t ring_add(uint32_t a, uint32_t b) { return R::from_u64((uint64_t)a + b); }
    void require_square(const DMatrix &c) const {
        if (c.rows != big_n_ || c.cols != big_n_) throw std::runtime_error("ciphertext must be N x N");
    }
    static void same_size(const DMatrix &a, const DMatrix &b) {
        if (a.rows != b.rows || a.cols != b.cols)
            throw std::runtime_error("Cannot add Ciphertexts because they are different sizes");  // gsw_impl.rs:88-89,105-106
    }
    DMatrix dense_product(const DMatrix &c1, const DMatrix &c2, bool is_nand) {
        same_size(c1, c2);
        if (c1.rows != c1.cols) throw std::runtime_error("Cannot add Ciphertexts because they are different sizes");
        const size_t count = c1.data.size();
        DeviceBuffer a(ctx_, count * 4), b(ctx_, count * 4), o(ctx_, count * 4);
        std::vector<uint32_t> ra = c1.row_major(), rb = c2.row_major();
        a.upload(ra.data(), count * 4);
        b.upload(rb.data(), count * 4);
        check(is_nand ? gsw_nand_dense(ctx_, a.u32(), b.u32(), o.u32(), c1.rows)
                      : gsw_mul_dense(ctx_, a.u32(), b.u32(), o.u32(), c1.rows));
#ifdef GSW_USE_FLATTEN  // gsw_impl.rs:108-109,120-121
        if (c1.cols % R::Num_Bits == 0) check(gsw_flatten(ctx_, o.u32(), c1.rows, c1.cols / R::Num_Bits));
#endif
        std::vector<uint32_t> ro(count);
        o.download(ro.data(), count * 4);
        return DMatrix::from_row_major(c1.rows, c1.cols, ro);
    }
    CompactCiphertext gadget(const CompactCiphertext &c1, const CompactCiphertext &c2, unsigned flags) {
        if (c1.batch != c2.batch) throw std::runtime_error("Cannot add Ciphertexts because they are different sizes");
        const size_t stride = gsw_ctx_compact_elems(ctx_);
        CompactCiphertext out{DeviceBuffer(ctx_, stride * 4 * c1.batch), c1.batch};
        check(gsw_mul_gadget_batch(ctx_, c1.buf.u32(), stride, c2.buf.u32(), stride, out.buf.u32(), stride, c1.batch,
                                   flags));
        check(gsw_sync(ctx_));
        return out;
    }
    std::vector<uint32_t> uniform(uint64_t lo, uint64_t hi, size_t count) {
        DeviceBuffer d(ctx_, count * 4);
        check(gsw_fill_uniform(ctx_, seed_, ++draw_, lo, hi, check_max_ ? 0 : 1, d.u32(), count));
        std::vector<uint32_t> out(count);
        d.download(out.data(), count * 4);
        return out;
    }
    std::vector<uint32_t> errors(size_t count) {
        DeviceBuffer d(ctx_, count * 4);
        if (sampler_ == Sampler::Naive) check(gsw_sample_naive(ctx_, seed_, ++draw_, d.u32(), count));
        else check(gsw_sample_gaussian(ctx_, seed_, ++draw_, 1.0, d.u32(), count));
        std::vector<uint32_t> out(count);
        d.download(out.data(), count * 4);
        return out;
    }

    gsw_ctx *ctx_ = nullptr;
    int n_, m_, big_n_ = 0;
    Sampler sampler_;
    uint64_t seed_, draw_ = 0;
    bool check_max_;
};

}  // namespace gsw
