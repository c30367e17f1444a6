// Wire format (layout documented in include/gsw_hip.h) and the sampler table: host code only.
#include "gsw_wire.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace gswwire {

namespace {

std::string format(const char *fmt, unsigned long long a = 0, unsigned long long b = 0, unsigned long long c = 0,
                   unsigned long long d = 0) {
    char buf[256];
    snprintf(buf, sizeof buf, fmt, a, b, c, d);
    return buf;
}

}  // namespace

bool shape_of(const Params &p, int kind, Shape &s) {
    switch (kind) {
        case 1: s = {p.big_n, p.width}; return true;
        case 2: s = {(uint32_t)p.m, p.width}; return true;
        case 3: s = {1u, (uint32_t)p.n}; return true;
        default: return false;
    }
}

bool canonical(const Params &p, uint32_t v) { return p.ring_kind == 0 ? v < kFpP : (v & ~p.mask) == 0; }

size_t object_bytes(const Params &p, int kind, size_t count) {
    Shape sh{};
    if (!shape_of(p, kind, sh)) return 0;
    const size_t per = (size_t)sh.rows * sh.cols;
    if (per != 0 && count > (SIZE_MAX - kHeaderBytes - kTrailerBytes) / 4 / per) return 0;
    return kHeaderBytes + 4 * count * per + kTrailerBytes;
}

void put_u32(uint8_t *dst, uint32_t v) {
    for (int i = 0; i < 4; ++i) dst[i] = (uint8_t)(v >> (8 * i));
}

uint32_t get_u32(const uint8_t *src) {
    return src[0] | (uint32_t)src[1] << 8 | (uint32_t)src[2] << 16 | (uint32_t)src[3] << 24;
}

uint64_t digest(const uint8_t *payload, size_t count) {
    uint64_t acc = 0;
    for (size_t i = 0; i < count; ++i) acc += ((uint64_t)get_u32(payload + 4 * i) + 1) * (2 * (uint64_t)i + 1);
    return acc * 0x9E3779B97F4A7C15ull;
}

void write_header(const Params &p, int kind, const Shape &sh, uint32_t count, uint8_t *h) {
    memset(h, 0, kHeaderBytes);
    memcpy(h, "GSW1", 4);
    h[4] = (uint8_t)kind;
    h[5] = (uint8_t)p.ring_kind;
    h[6] = (uint8_t)p.bits;
    put_u32(h + 8, (uint32_t)p.n);
    put_u32(h + 12, (uint32_t)p.m);
    put_u32(h + 16, sh.rows);
    put_u32(h + 20, sh.cols);
    put_u32(h + 24, count);
}

void write_trailer(uint8_t *buf, size_t elems) {
    const uint64_t d = digest(buf + kHeaderBytes, elems);
    for (int i = 0; i < 8; ++i) buf[kHeaderBytes + 4 * elems + i] = (uint8_t)(d >> (8 * i));
}

bool check(const Params &p, int kind, const uint8_t *buf, size_t bytes, uint32_t *count, std::string &err) {
    Shape sh{};
    if (!shape_of(p, kind, sh)) return err = format("unknown wire object kind %llu", (unsigned long long)kind), false;
    if (bytes < kHeaderBytes + kTrailerBytes) return err = format("wire object truncated (%llu bytes)", bytes), false;
    if (memcmp(buf, "GSW1", 4) != 0) return err = "bad magic: not a GSW1 wire object", false;
    if (buf[4] != kind) return err = format("wire object holds kind %llu, expected %llu", buf[4], (unsigned long long)kind), false;
    if (buf[5] != p.ring_kind || buf[6] != p.bits)
        return err = format("wire object is over ring (%llu, %llu bits), context over (%llu, %llu bits)", buf[5], buf[6],
                            (unsigned long long)p.ring_kind, (unsigned long long)p.bits), false;
    if (buf[7] != 0 || get_u32(buf + 28) != 0) return err = "reserved header bytes are not zero", false;
    if (get_u32(buf + 8) != (uint32_t)p.n || get_u32(buf + 12) != (uint32_t)p.m)
        return err = format("wire object has (n, m) = (%llu, %llu), context (%llu, %llu)", get_u32(buf + 8),
                            get_u32(buf + 12), (unsigned long long)p.n, (unsigned long long)p.m), false;
    if (get_u32(buf + 16) != sh.rows || get_u32(buf + 20) != sh.cols)
        return err = format("wire object is %llu x %llu, expected %llu x %llu", get_u32(buf + 16), get_u32(buf + 20),
                            sh.rows, sh.cols), false;
    const uint32_t cnt = get_u32(buf + 24);
    if (kind != 1 && cnt != 1) return err = "a key object holds exactly one entry", false;
    // the count is untrusted: bound it by the bytes at hand before multiplying
    const size_t per_entry = (size_t)sh.rows * sh.cols;
    if (per_entry == 0 || cnt > (bytes - kHeaderBytes - kTrailerBytes) / 4 / per_entry)
        return err = format("wire object is %llu bytes, too short for the %llu entries its header names", bytes, cnt), false;
    const size_t elems = (size_t)cnt * per_entry;
    if (bytes != kHeaderBytes + 4 * elems + kTrailerBytes)
        return err = format("wire object is %llu bytes, its header implies %llu", bytes,
                            kHeaderBytes + 4 * elems + kTrailerBytes), false;
    const uint8_t *payload = buf + kHeaderBytes;
    for (size_t i = 0; i < elems; ++i) {
        const uint32_t v = get_u32(payload + 4 * i);
        if (!canonical(p, v)) return err = format("element %llu (%llu) is not a canonical ring value", i, v), false;
    }
    uint64_t d = 0;
    for (int i = 0; i < 8; ++i) d |= (uint64_t)payload[4 * elems + i] << (8 * i);
    if (d != digest(payload, elems)) return err = "digest mismatch: wire object corrupted", false;
    *count = cnt;
    return true;
}

void gaussian_table(double sigma, std::vector<uint64_t> &cum, int &kmax) {
    kmax = (int)std::ceil(13.0 * sigma) + 1;
    if (kmax > 4096) kmax = 4096;
    const int len = 2 * kmax + 1;
    cum.resize(len);
    long double total = 0.0L;
    for (int k = -kmax; k <= kmax; ++k) total += expl(-(long double)k * k / (2.0L * sigma * sigma));
    long double run = 0.0L;
    for (int idx = 0; idx < len; ++idx) {
        const int k = idx - kmax;
        run += expl(-(long double)k * k / (2.0L * sigma * sigma)) / total;
        const long double scaled = run * 18446744073709551616.0L;
        cum[idx] = scaled >= 18446744073709551615.0L ? 0xFFFFFFFFFFFFFFFFull : (uint64_t)scaled;
    }
    cum[len - 1] = 0xFFFFFFFFFFFFFFFFull;
}

}  // namespace gswwire
