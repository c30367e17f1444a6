// Host-only parts of the engine: the wire format's writer and validating reader (they parse untrusted
// bytes) and the sampler's probability table.  No HIP here, so that this file also builds with a plain
// host compiler under AddressSanitizer / UBSan (tests/test_host_sanitizers.py).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace gswwire {

constexpr size_t kHeaderBytes = 32, kTrailerBytes = 8;
constexpr uint32_t kFpP = 2147483647u;

// what a wire object is checked against: the context's ring and dimensions
struct Params {
    int ring_kind;  // GSW_RING_FP_M31 (0) / GSW_RING_ZPOW2 (1) of include/gsw_hip.h
    int bits;       // bits per element (R::Num_Bits)
    uint32_t mask;  // p for Fp, 2^M - 1 for Zpow2
    int n, m;
    uint32_t big_n;  // bits * (n + 1)
    uint32_t width;  // n + 1
};

struct Shape {
    uint32_t rows, cols;
};

// kind: GSW_WIRE_CIPHERTEXT (1) N x (n+1), GSW_WIRE_PUBLIC_KEY (2) m x (n+1), GSW_WIRE_SECRET_KEY (3) 1 x n
bool shape_of(const Params &p, int kind, Shape &s);
bool canonical(const Params &p, uint32_t v);
// bytes of an object of `count` entries; 0 for an unknown kind or a size that does not fit size_t
size_t object_bytes(const Params &p, int kind, size_t count);
void put_u32(uint8_t *dst, uint32_t v);
uint32_t get_u32(const uint8_t *src);
// position-weighted digest over `count` little-endian u32 values (the device's gsw_digest_batch of one entry)
uint64_t digest(const uint8_t *payload, size_t count);
void write_header(const Params &p, int kind, const Shape &sh, uint32_t count, uint8_t *buf);
void write_trailer(uint8_t *buf, size_t elems);
// Validates everything but the payload's placement: magic, kind, ring, reserved bytes, (n, m), shape, entry
// count against the byte count (before any multiplication can overflow), canonical elements, digest.
// Returns false with a message in `err`; on success *count is the entry count.
bool check(const Params &p, int kind, const uint8_t *buf, size_t bytes, uint32_t *count, std::string &err);

// cumulative table of the discrete Gaussian P(k) ~ exp(-k^2 / (2 sigma^2)), k in [-kmax, kmax], scaled to 2^64
void gaussian_table(double sigma, std::vector<uint64_t> &cum, int &kmax);

}  // namespace gswwire
