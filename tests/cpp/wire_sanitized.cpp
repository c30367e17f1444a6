// Host-only driver for the wire-format reader/writer (tfhe_amd/csrc/gsw_wire.cpp), built with
// -fsanitize=address,undefined by tests/test_host_sanitizers.py.  Every buffer is an exact-size heap block,
// so a read past the bytes the caller handed over is caught by the sanitizer, not by luck.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "gsw_wire.h"

using namespace gswwire;

static int failures = 0;
#define EXPECT(cond)                                                  \
    do {                                                              \
        if (!(cond)) {                                                \
            printf("FAILED line %d: %s\n", __LINE__, #cond);          \
            ++failures;                                               \
        }                                                             \
    } while (0)

static uint64_t rng_state = 0x1234567ull;
static uint32_t rnd() {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}

// an exact-size copy of the first `len` bytes: over-reads land in ASan's red zone
static std::unique_ptr<uint8_t[]> exact(const std::vector<uint8_t> &src, size_t len) {
    std::unique_ptr<uint8_t[]> p(new uint8_t[len ? len : 1]);
    if (len) memcpy(p.get(), src.data(), len);
    return p;
}

static std::vector<uint8_t> make_object(const Params &p, int kind, uint32_t count) {
    Shape sh{};
    shape_of(p, kind, sh);
    const size_t elems = (size_t)count * sh.rows * sh.cols;
    std::vector<uint8_t> buf(object_bytes(p, kind, count));
    write_header(p, kind, sh, count, buf.data());
    for (size_t i = 0; i < elems; ++i) put_u32(buf.data() + kHeaderBytes + 4 * i, rnd() % (p.ring_kind == 0 ? kFpP : p.mask));
    write_trailer(buf.data(), elems);
    return buf;
}

static bool accepts(const Params &p, int kind, const uint8_t *buf, size_t len, uint32_t *count = nullptr) {
    uint32_t c = 0;
    std::string err;
    const bool ok = check(p, kind, buf, len, &c, err);
    if (!ok) EXPECT(!err.empty());
    if (count) *count = c;
    return ok;
}

int main() {
    const Params fp{0, 31, kFpP, 2, 62, 93, 3};
    const Params z8{1, 8, 255u, 3, 24, 32, 4};
    for (const Params &p : {fp, z8}) {
        for (int kind = 1; kind <= 3; ++kind) {
            const uint32_t count = kind == 1 ? 3 : 1;
            const std::vector<uint8_t> good = make_object(p, kind, count);
            uint32_t got = 0;
            EXPECT(accepts(p, kind, exact(good, good.size()).get(), good.size(), &got) && got == count);
            // truncated at every length, and one byte too long
            for (size_t len = 0; len < good.size(); ++len) EXPECT(!accepts(p, kind, exact(good, len).get(), len));
            std::vector<uint8_t> longer = good;
            longer.push_back(0);
            EXPECT(!accepts(p, kind, longer.data(), longer.size()));
            // the other ring / other kinds refuse it
            EXPECT(!accepts(p.ring_kind == 0 ? z8 : fp, kind, good.data(), good.size()));
            EXPECT(!accepts(p, kind == 1 ? 2 : 1, good.data(), good.size()));
            // every single-bit flip is caught (header fields, payload via the digest, the digest itself)
            for (size_t bit = 0; bit < good.size() * 8; bit += 1 + rnd() % 5) {
                std::vector<uint8_t> bad = good;
                bad[bit / 8] ^= (uint8_t)(1u << (bit % 8));
                EXPECT(!accepts(p, kind, exact(bad, bad.size()).get(), bad.size()));
            }
            // entry counts that would overflow 4 * count * rows * cols or reach far past the buffer
            for (uint32_t cnt : {0xFFFFFFFFu, 0x80000000u, 0x40000001u, 0x10000u, count + 1}) {
                std::vector<uint8_t> bad = good;
                put_u32(bad.data() + 24, cnt);
                EXPECT(!accepts(p, kind, exact(bad, bad.size()).get(), bad.size()));
            }
            // a non-canonical element with a recomputed (valid) digest
            std::vector<uint8_t> bad = good;
            put_u32(bad.data() + kHeaderBytes, p.ring_kind == 0 ? kFpP : p.mask + 1);
            write_trailer(bad.data(), (bad.size() - kHeaderBytes - kTrailerBytes) / 4);
            EXPECT(!accepts(p, kind, bad.data(), bad.size()));
        }
        // an empty ciphertext object is a valid object of zero entries
        const std::vector<uint8_t> empty = make_object(p, 1, 0);
        uint32_t got = 7;
        EXPECT(empty.size() == kHeaderBytes + kTrailerBytes && accepts(p, 1, exact(empty, empty.size()).get(), empty.size(), &got) && got == 0);
        EXPECT(object_bytes(p, 9, 1) == 0 && !accepts(p, 9, empty.data(), empty.size()));
    }
    // random garbage of random lengths
    for (int it = 0; it < 3000; ++it) {
        const size_t len = rnd() % 200;
        std::vector<uint8_t> junk(len);
        for (auto &b : junk) b = (uint8_t)rnd();
        if (len >= 4 && (it & 1)) memcpy(junk.data(), "GSW1", 4);
        EXPECT(!accepts(fp, 1 + it % 3, exact(junk, len).get(), len));
    }
    // sizes that do not fit size_t are reported as 0, not wrapped
    const Params huge{0, 31, kFpP, 1 << 20, 1 << 24, 31u * ((1u << 20) + 1), (1u << 20) + 1};
    EXPECT(object_bytes(huge, 1, (size_t)-1) == 0);
    // the sampler's table: monotone, ends at 2^64 - 1, symmetric mass
    for (double sigma : {0.3, 1.0, 3.2, 300.0}) {
        std::vector<uint64_t> cum;
        int kmax = 0;
        gaussian_table(sigma, cum, kmax);
        EXPECT((int)cum.size() == 2 * kmax + 1 && cum.back() == 0xFFFFFFFFFFFFFFFFull);
        for (size_t i = 1; i < cum.size(); ++i) EXPECT(cum[i] >= cum[i - 1]);
        const double below = (double)cum[kmax - 1] / 18446744073709551616.0;
        const double upto = (double)cum[kmax] / 18446744073709551616.0;
        EXPECT(below > 0.0 && below < 0.5 && upto > 0.5 && (below + upto > 0.999999 && below + upto < 1.000001));
    }
    printf(failures ? "wire_sanitized: %d FAILURES\n" : "wire_sanitized: ok\n", failures);
    return failures ? 1 : 0;
}
