// gfx950 (MI355X, CDNA4) kernels for the GSW hot path.
//
// The product of the reference, `ciphertext1 * cipertext2` (src/gsw/gsw_impl.rs:107), with a
// flattened left operand C1 = BitDecomp(C1c) collapses to Pc = BitDecomp(C1c) * C2c mod q:
// an (N x N binary) x (N x (n+1)) contraction.  The binary operand is never materialised: its
// int8 MFMA fragments are expanded in registers from the packed words of C1c.  The right operand
// is split into four signed base-256 digits (each in [-128, 127], the int8 MFMA is signed) stored in
// MFMA-fragment order, so each 32x32x32 int8 MFMA consumes exactly one word per row (32 K-slots) of
// the left operand and one 1 KiB fragment of one limb; accumulation is exact in int32 and the four
// limbs of one output element live in the same lane, so recombination needs no cross-lane traffic.
//
// K-slot order.  Slot (h, i, t) of a word -- lane half h = lane >> 5, fragment dword i, byte t --
// stands for bit 4h + i + 8t.  The order is free as long as both operands agree; this one makes
// the left fragment two VALU ops per dword: ((x >> 4h) >> i) & 0x01010101.
#include "gsw_kernels.h"

#include <cstdlib>
#include <type_traits>

namespace gswk {

// Batch entry b lives in slot slots[b] of its pool (or slot b when no slot list is given).
__device__ __forceinline__ size_t slot_of(const uint32_t *slots, int b) { return slots ? (size_t)slots[b] : (size_t)b; }

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ ring helpers (device)

__device__ __forceinline__ uint32_t fp_reduce64(uint64_t x) {
    // x mod (2^31 - 1) by Mersenne folding (modulus: field.rs:8)
    x = (x & kFpP) + (x >> 31);
    x = (x & kFpP) + (x >> 31);
    x = (x & kFpP) + (x >> 31);
    return (uint32_t)(x >= kFpP ? x - kFpP : x);
}

template <int RING>
__device__ __forceinline__ uint32_t ring_from_u64(uint64_t x, uint32_t mask) {
    if (RING == kRingFp) return fp_reduce64(x);
    return (uint32_t)x & mask;  // Zpow2::new (pow2_ring.rs:57-59)
}

template <int RING>
__device__ __forceinline__ uint32_t ring_add(uint32_t a, uint32_t b, uint32_t mask) {
    if (RING == kRingFp) {
        uint32_t s = a + b;  // < 2^32
        return s >= kFpP ? s - kFpP : s;
    }
    return (a + b) & mask;  // pow2_ring.rs:75-81
}

template <int RING>
__device__ __forceinline__ uint32_t ring_sub(uint32_t a, uint32_t b, uint32_t mask) {
    if (RING == kRingFp) return a >= b ? a - b : a + kFpP - b;
    return (a - b) & mask;  // pow2_ring.rs:86-92
}

template <int RING>
__device__ __forceinline__ uint32_t ring_mul(uint32_t a, uint32_t b, uint32_t mask) {
    return ring_from_u64<RING>((uint64_t)a * b, mask);  // pow2_ring.rs:97-103
}

// G[i][j] = 2^(i mod ell) [j == i div ell]: transpose of build_gadget_matrix (gsw/mod.rs:47-56)
template <int RING>
__device__ __forceinline__ uint32_t gadget_entry(int row, int col, int ell, uint32_t mask) {
    int j = row / ell;
    if (j != col) return 0u;
    return ring_from_u64<RING>(1ull << (row - j * ell), mask);
}

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint64_t rng_u64(uint64_t seed, uint64_t stream, uint64_t counter) {
    return mix64(mix64(seed ^ (stream * 0xD1342543DE82EF95ull)) + counter);
}

// The four signed base-256 digits d_l in [-128, 127] of a canonical ring element, as int8 bytes of one word:
// sum_l d_l 256^l = x' with x' = x mod q.  y = x' + 0x80808080 propagates the carries, byte l of y is d_l + 128.
// Four digits reach from -2155905152 to 2139062143; a word above that wraps to x - 2^32, which is the same
// element of Z/2^M; an element of Fp above it (fewer than 2^23 of them, just below p) is taken as x - p instead.
template <int RING>
__device__ __forceinline__ uint32_t signed_digits(uint32_t x) {
    if (RING == kRingFp && x > 0x7F7F7F7Fu) x -= kFpP;
    return (x + 0x80808080u) ^ 0x80808080u;
}

// Scaled bits (Fp only).  The left fragments are cheaper to make when bit i of a byte stays where it is:
// dword i of a fragment becomes y & (0x01010101 << i) -- bytes in {0, 2^i}, one instruction -- instead of
// (y >> i) & 0x01010101 -- bytes in {0, 1}, two.  The right operand pays for it once, in its preparation: the row
// that meets the K slot of bit k is multiplied by 2^-c, c = k & 3, before it is split into digits, so that
// (2^c bit) (2^-c x) = bit x mod p.  In Fp = Z/(2^31 - 1) that product is a rotation of the 31-bit value by c
// places to the right.  The limb sums grow with the weights (1 + 2 + 4 + 8) / 4, so mode 2 (c = k & 3) serves
// contractions short enough for the 32-bit recombination with those sums, mode 1 (c = k & 1: dwords 0, 1 masked in
// place, dwords 2, 3 after one shared shift) the longer ones, mode 0 (no scaling) Z/2^M, where 2 has no inverse.
__device__ __forceinline__ uint32_t fp_scale_down(uint32_t x, int c) {
    return ((x >> c) | (x << (31 - c))) & kFpP;  // x * 2^-c mod p for canonical x, 0 <= c < 31
}

template <int SCALE>
__device__ __forceinline__ int scale_shift_of_bit(int bit) {
    return SCALE == 2 ? (bit & 3) : SCALE == 1 ? (bit & 1) : 0;
}

// The int8 fragment (16 K slots of one row) from its word, already shifted to the lane's half: y = x >> 4 h.
template <int SCALE>
__device__ __forceinline__ v4i expand_bits(uint32_t y) {
    v4i f;
    if (SCALE == 2) {
        f[0] = (int)(y & 0x01010101u);
        f[1] = (int)(y & 0x02020202u);
        f[2] = (int)(y & 0x04040404u);
        f[3] = (int)(y & 0x08080808u);
    } else if (SCALE == 1) {
        const uint32_t y2 = y >> 2;
        f[0] = (int)(y & 0x01010101u);
        f[1] = (int)(y & 0x02020202u);
        f[2] = (int)(y2 & 0x01010101u);
        f[3] = (int)(y2 & 0x02020202u);
    } else {
        f[0] = (int)(y & 0x01010101u);
        f[1] = (int)((y >> 1) & 0x01010101u);
        f[2] = (int)((y >> 2) & 0x01010101u);
        f[3] = (int)((y >> 3) & 0x01010101u);
    }
    return f;
}

// ------------------------------------------------------------------ right-operand preparation
//
// One workgroup per (K element e, column tile nt, batch entry): reads the 32 source rows
// e*ell .. e*ell+31 x 32 columns (128 B contiguous per row), splits every word into its four signed
// digits (signed_digits) and writes the 1 KiB fragments through LDS so that the global stores are
// 16 B per lane and fully coalesced.  K slots past the operand keep the digit 0.
// Full tile: thread (cc = tid & 31, i = (tid >> 5) & 3, h = tid >> 7) owns dword i of lane
// 32h + cc in each of the four limb fragments.  Packed tail tile (at most 8 valid columns): one
// fragment whose MFMA column l*8 + cc carries limb l of column cc.

template <int RING>
__global__ __launch_bounds__(256) void prepare_rhs_kernel(PrepArgs a, int ell) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[4 * 256];
    const int tid = threadIdx.x;
    const int e = blockIdx.x, nt = blockIdx.y, b = blockIdx.z;
    const int cidx = tid & 31, i = (tid >> 5) & 3, h = tid >> 7;
    const bool packed = a.packed_tail && nt == (int)gridDim.y - 1;
    const uint32_t *src = a.src + (size_t)slot_of(a.src_slot, b) * a.src_stride;
    const int col = nt * kTileCols + (packed ? (cidx & 7) : cidx);
    uint32_t limb[4] = {0u, 0u, 0u, 0u};
    if (e < a.elems && col < a.cols_valid) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int bit = 4 * h + i + 8 * t;
            const int row = e * ell + bit;
            if (bit < ell && row < a.rows_valid) {
                const uint32_t x = signed_digits<RING>(src[(size_t)row * a.ld_src + col]);
#pragma unroll
                for (int l = 0; l < 4; ++l) limb[l] |= ((x >> (8 * l)) & 0xFFu) << (8 * t);
            }
        }
    }
    uint8_t *dst = a.dst + slot_of(a.dst_slot, b) * a.dst_stride + (size_t)nt * a.e4 * (4 * kFragBytes);
    if (packed) {
        const int l = cidx >> 3;
        tile[(32 * h + cidx) * 4 + i] = l == 0 ? limb[0] : l == 1 ? limb[1] : l == 2 ? limb[2] : limb[3];
        __syncthreads();
        if (tid < 64) reinterpret_cast<uint4 *>(dst + (size_t)e * kFragBytes)[tid] = reinterpret_cast<const uint4 *>(tile)[tid];
    } else {
#pragma unroll
        for (int l = 0; l < 4; ++l) tile[l * 256 + (32 * h + cidx) * 4 + i] = limb[l];
        __syncthreads();
        reinterpret_cast<uint4 *>(dst + (size_t)e * (4 * kFragBytes))[tid] = reinterpret_cast<const uint4 *>(tile)[tid];
    }
}

// The same for the 16x16x64 kernels: one workgroup per (stage t, MFMA step p, column tile, batch
// entry) builds the fragments [limb][column group] of that step from elements 4t + p (K groups 0, 1)
// and 4t + 2 + p (K groups 2, 3).  Lane 16 g + c of a fragment: column c of the group, K group g.
template <int RING, int SCALE>
__global__ __launch_bounds__(256) void prepare_rhs16_kernel(PrepArgs a, int ell) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[8 * 256];
    const int tid = threadIdx.x;
    const int t = blockIdx.x >> 1, step = blockIdx.x & 1, nt = blockIdx.y, b = blockIdx.z;
    const int cidx = tid & 31, i = (tid >> 5) & 3, h = tid >> 7;
    const bool packed = a.packed_tail && nt == (int)gridDim.y - 1;
    const bool narrow = packed && a.packed_tail == kTailNarrow;  // 16 MFMA columns: 4 l + c
    const uint32_t *src = a.src + (size_t)slot_of(a.src_slot, b) * a.src_stride;
    const int col = nt * kTileCols + (narrow ? (cidx & 3) : packed ? (cidx & 7) : cidx);
#pragma unroll
    for (int gh = 0; gh < 2; ++gh) {
        const int e = 4 * t + 2 * gh + step;
        uint32_t limb[4] = {0u, 0u, 0u, 0u};
        if (e < a.elems && col < a.cols_valid) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int bit = 4 * h + i + 8 * tt;
                const int row = e * ell + bit;
                if (bit < ell && row < a.rows_valid) {
                    uint32_t v = src[(size_t)row * a.ld_src + col];
                    if (SCALE) v = fp_scale_down(v, scale_shift_of_bit<SCALE>(bit));  // see "Scaled bits"
                    const uint32_t x = signed_digits<RING>(v);
#pragma unroll
                    for (int l = 0; l < 4; ++l) limb[l] |= ((x >> (8 * l)) & 0xFFu) << (8 * tt);
                }
            }
        }
        const int g = 2 * gh + h;
        const int cg = cidx >> 4, c16 = cidx & 15;
        if (packed) {
            const int l = narrow ? (cidx >> 2) & 3 : cidx >> 3;
            tile[(cg * 64 + g * 16 + c16) * 4 + i] = l == 0 ? limb[0] : l == 1 ? limb[1] : l == 2 ? limb[2] : limb[3];
        } else {
#pragma unroll
            for (int l = 0; l < 4; ++l) tile[((l * 2 + cg) * 64 + g * 16 + c16) * 4 + i] = limb[l];
        }
    }
    __syncthreads();
    uint8_t *dst = a.dst + slot_of(a.dst_slot, b) * a.dst_stride + (size_t)nt * a.e4 * (4 * kFragBytes);
    if (narrow) {
        // one fragment (column group 0) per MFMA step; the other half of the tile's space is kept zero so that
        // equal operands have equal prepared images
        if (tid < 64)
            reinterpret_cast<uint4 *>(dst + (size_t)blockIdx.x * kFragBytes)[tid] = reinterpret_cast<const uint4 *>(tile)[tid];
        else if (tid < 128)
            reinterpret_cast<uint4 *>(dst + (size_t)(gridDim.x + blockIdx.x) * kFragBytes)[tid - 64] = make_uint4(0u, 0u, 0u, 0u);
    } else if (packed) {
        if (tid < 128)
            reinterpret_cast<uint4 *>(dst + (size_t)blockIdx.x * (2 * kFragBytes))[tid] =
                reinterpret_cast<const uint4 *>(tile)[tid];
    } else {
        uint4 *d = reinterpret_cast<uint4 *>(dst + (size_t)blockIdx.x * (8 * kFragBytes));
        d[tid] = reinterpret_cast<const uint4 *>(tile)[tid];
        d[tid + 256] = reinterpret_cast<const uint4 *>(tile)[tid + 256];
    }
}

hipError_t launch_prepare(const RingDesc &ring, const PrepArgs &a, hipStream_t stream) {
    if (a.batch <= 0) return hipSuccess;
    if (a.layout16) {
        dim3 grid((unsigned)(a.e4 / 2), (unsigned)col_tiles(a.cols_valid), (unsigned)a.batch);
        if (a.scale_mode < 0 || a.scale_mode > 2 || (a.scale_mode && ring.kind != kRingFp)) return hipErrorInvalidValue;
        if (ring.kind != kRingFp) hipLaunchKernelGGL((prepare_rhs16_kernel<kRingZpow2, 0>), grid, dim3(256), 0, stream, a, ring.ell);
        else if (a.scale_mode == 2) hipLaunchKernelGGL((prepare_rhs16_kernel<kRingFp, 2>), grid, dim3(256), 0, stream, a, ring.ell);
        else if (a.scale_mode == 1) hipLaunchKernelGGL((prepare_rhs16_kernel<kRingFp, 1>), grid, dim3(256), 0, stream, a, ring.ell);
        else hipLaunchKernelGGL((prepare_rhs16_kernel<kRingFp, 0>), grid, dim3(256), 0, stream, a, ring.ell);
    } else {
        if (a.scale_mode) return hipErrorInvalidValue;  // the 32x32x32 kernels read unscaled digits
        dim3 grid((unsigned)a.e4, (unsigned)col_tiles(a.cols_valid), (unsigned)a.batch);
        if (ring.kind == kRingFp) hipLaunchKernelGGL(prepare_rhs_kernel<kRingFp>, grid, dim3(256), 0, stream, a, ring.ell);
        else hipLaunchKernelGGL(prepare_rhs_kernel<kRingZpow2>, grid, dim3(256), 0, stream, a, ring.ell);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ the contraction
//
// Workgroup: WAVES waves, output tile 64*WAVES rows x 32 columns x 4 limbs.  Wave w owns rows
// 64w .. 64w+63 as two 32-row groups; per K element it issues 2 (row groups) x NFRAG MFMA
// 32x32x32 i8 that share 2 left and NFRAG right fragments (NFRAG = 4 limbs; 1 in a packed tail
// tile, where one fragment carries all four limbs of up to 8 columns).
// LDS image of a stage (4 K elements): 16 B per tile row (its 4 left words) followed by 4*NFRAG
// fragments (element s, limb l at (NFRAG s + l) KiB), each read lane-linearly (conflict free).

// Everything the epilogue needs besides the accumulators.
struct TileCtx {
    int b, nt, mt, wave, lane;
    int ell;
    uint32_t mask;
};

// Epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5).  The accumulators started at GemmArgs::acc_init, so for Fp every limb sum is
// non-negative; for Z/2^M a negative sum read as uint32 is off by a multiple of 2^32.
template <int RING, int NFRAG>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &p, const TileCtx &c, int tile_rows,
                                              const v16i (&acc)[2][NFRAG]) {
    const int lane = c.lane, half = lane >> 5, c32 = lane & 31;
    const int col = c.nt * kTileCols + (NFRAG == 4 ? c32 : (c32 & 7));
    const bool col_owner = NFRAG == 4 ? true : c32 < 8;
    const uint32_t addend = (p.mode == kModeEncrypt) ? p.addend[c.b] : 0u;
    uint32_t *out = p.out + (size_t)slot_of(p.out_slot, c.b) * p.out_stride;
    unsigned long long dig = 0;
    const bool want_digest = p.digests != nullptr;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int row = c.mt * tile_rows + c.wave * 64 + g * 32 + rr;
            uint64_t s0, s1, s2, s3;
            if (NFRAG == 4) {
                s0 = (uint32_t)acc[g][0][r];
                s1 = (uint32_t)acc[g][1 % NFRAG][r];
                s2 = (uint32_t)acc[g][2 % NFRAG][r];
                s3 = (uint32_t)acc[g][3 % NFRAG][r];
            } else {
                // packed tile: limb l of column cc sits in MFMA column 8 l + cc of the same half
                const int v = acc[g][0][r];
                const int base = (lane & 32) | (lane & 7);
                s0 = (uint32_t)__shfl(v, base, 64);
                s1 = (uint32_t)__shfl(v, base + 8, 64);
                s2 = (uint32_t)__shfl(v, base + 16, 64);
                s3 = (uint32_t)__shfl(v, base + 24, 64);
            }
            uint32_t val = ring_from_u64<RING>(s0 + (s1 << 8) + (s2 << 16) + (s3 << 24), c.mask);
            if (col_owner && row < p.rows && col < p.ld_out) {
                if (col >= p.out_cols) {
                    val = 0u;  // keep the padding columns of the compact form zero
                } else if (p.mode == kModeNand) {
                    val = ring_sub<RING>(gadget_entry<RING>(row, col, c.ell, c.mask), val, c.mask);
                } else if (p.mode == kModeEncrypt) {
                    val = ring_add<RING>(val, ring_mul<RING>(addend, gadget_entry<RING>(row, col, c.ell, c.mask), c.mask),
                                         c.mask);
                }
                const size_t idx = (size_t)row * p.ld_out + col;
                out[idx] = val;
                if (want_digest) {
                        // (x + 1) * (2 idx + 1) as one 32 x 32 -> 64 multiply-add; the constant factor of the
                        // digest is applied once per wave (multiplication mod 2^64 distributes over the sum)
                        const unsigned w = 2u * (unsigned)idx + 1u;
                        dig += (unsigned long long)val * w + w;
                    }
            } else if (NFRAG == 1 && !col_owner && row < p.rows && c.nt * kTileCols + c32 < p.ld_out) {
                // a packed tile computes 8 columns; the other lanes zero the rest of the padding columns
                const size_t idx = (size_t)row * p.ld_out + c.nt * kTileCols + c32;
                out[idx] = 0u;
                if (want_digest) dig += 2u * (unsigned)idx + 1u;
            }
        }
    }
    if (p.digests) {
        // one atomic per wave; the sum mod 2^64 does not depend on the order of arrival
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)dig, off, 64);
            const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(dig >> 32), off, 64);
            dig += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 0) atomicAdd(&p.digests[c.b], dig * 0x9E3779B97F4A7C15ull);
    }
}

// One K element: expand the two left fragments from their words and run the MFMAs.
template <int NFRAG>
__device__ __forceinline__ void gemm_kstep(uint32_t x0, uint32_t x1, int sh, const unsigned char *frags, int lane,
                                           v16i (&acc)[2][NFRAG]) {
    const uint32_t y0 = x0 >> sh, y1 = x1 >> sh;
    v4i f0, f1;
    f0[0] = (int)(y0 & 0x01010101u);
    f0[1] = (int)((y0 >> 1) & 0x01010101u);
    f0[2] = (int)((y0 >> 2) & 0x01010101u);
    f0[3] = (int)((y0 >> 3) & 0x01010101u);
    f1[0] = (int)(y1 & 0x01010101u);
    f1[1] = (int)((y1 >> 1) & 0x01010101u);
    f1[2] = (int)((y1 >> 2) & 0x01010101u);
    f1[3] = (int)((y1 >> 3) & 0x01010101u);
#pragma unroll
    for (int l = 0; l < NFRAG; ++l) {
        const v4i fb = *reinterpret_cast<const v4i *>(frags + l * kFragBytes + lane * 16);
        acc[0][l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0, fb, acc[0][l], 0, 0, 0);
        acc[1][l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f1, fb, acc[1][l], 0, 0, 0);
    }
}

// Workgroups id and id+8 share an XCD (round-robin dispatch): give each XCD whole (batch entry,
// column tile) units so that the mt_count row tiles that stream the same right-operand panel run
// on one L2.  Placement only affects speed, never results.  Returns false for surplus workgroups.
__device__ __forceinline__ bool map_workgroup(int batch, int nt_count, int mt_count, int &b, int &nt, int &mt) {
    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    mt = local % mt_count;
    const int unit = (local / mt_count) * 8 + xcd;
    if (unit >= batch * nt_count) return false;
    b = unit / nt_count;
    nt = unit - b * nt_count;
    return true;
}

// Blocked placement for the 16x16x64 kernels: one batch entry per XCD at a time, its tiles ordered in
// passes of pass_rows row tiles x pass_cols column tiles (8 x 8 by default).  The 64 workgroups an XCD
// runs at a time (32 CUs x 2) then share every right-operand panel among 8 row tiles and every
// left-operand panel among 8 column tiles, so each byte crosses the fabric into that L2 about once
// per pass instead of once per workgroup.  Speed only: results do not depend on placement.
__device__ __forceinline__ bool map_workgroup_blocked(int batch, int nt_count, int mt_count, int kPassRows,
                                                      int kPassCols, int &b, int &nt, int &mt, int id = -1) {
    if (id < 0) id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int per = mt_count * nt_count;
    b = (local / per) * 8 + xcd;
    if (b >= batch) return false;
    int w = local % per;
    for (int m0 = 0; m0 < mt_count; m0 += kPassRows) {
        const int mc = min(kPassRows, mt_count - m0);
        for (int n0 = 0; n0 < nt_count; n0 += kPassCols) {
            const int cnt = mc * min(kPassCols, nt_count - n0);
            if (w < cnt) {
                nt = n0 + w / mc;
                mt = m0 + w % mc;
                return true;
            }
            w -= cnt;
        }
    }
    return false;
}

// ---- pipelined kernel: LDS-DMA issued from inline asm so that the loads of the next two stages
// stay in flight across the per-stage barrier (counted vmcnt, raw s_barrier, three LDS buffers).

__device__ __forceinline__ void lds_dma16(const void *gsrc, uint32_t lds_wave_base) {
    // global_load_lds_dwordx4: lane i's 16 bytes land at M0 + 16 i.  M0 is saved and restored inside
    // the statement; the compiler does not count this load, the kernel's own s_waitcnt vmcnt does.
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_wave_base)
        : "memory");
}

// The same load with the address as a uniform base (scalar registers) + a per-lane byte offset + a constant:
// no vector instruction is spent on the address.  The constant moves the LDS side as well: lane i's 16 bytes
// land at M0 + IMM + 16 i.
template <int IMM>
__device__ __forceinline__ void lds_dma16_base(const void *uniform_base, uint32_t lane_offset, uint32_t lds_wave_base) {
    static_assert(IMM >= 0 && IMM < 4096, "13-bit signed instruction offset");
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:%4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_offset), "s"(uniform_base), "s"(lds_wave_base), "n"(IMM)
        : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// The same for a body that carries LDS reads across the barrier (the skewed pipeline reads the fragments of a stage's
// last MFMA group at the end of the stage and uses them after the next barrier): the wave's own LDS reads have to
// have returned before it arrives, because right behind the barrier the other waves re-load that buffer and nothing
// orders an LDS-DMA write behind a ds_read that is still queued.  The waves of a thin row tile that own no rows are
// the ones that do it: they issue all their loads of a stage in one burst right behind the barrier, the pieces of
// the last group among them.  Without the lgkmcnt(0) an L2-warm load (a few hundred cycles from issue to landing)
// overtook a queued read about once in 10^5 stages at n = 10 and 10^9 at n = 256: one wrong product in ten to fifty
// thousand, seen only as digests that differ between identical runs.
template <int N>
__device__ __forceinline__ void wait_loads_and_own_reads_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

constexpr int kPipeBuffers = 3;
template <int WAVES>
constexpr int pipe_lds_bytes() { return kPipeBuffers * (WAVES * 1024 + kStageElems * 4 * kFragBytes); }

// ---- the same pipeline on the 16x16x64 MFMA shape.
//
// A wave still owns 64 rows x 32 columns x 4 limbs, now as 4 row groups x (2 column groups x 4 limbs)
// 16x16 accumulators; operand bytes per useful op are the same as with the 32x32 shape, but the chip
// holds a higher clock on this shape.  One MFMA covers two K elements: lane l of an operand fragment
// has row/column l & 15 and K group g = l >> 4; g >> 1 picks the element, g & 1 the half h of its
// word, fragment dword i byte t stands for bit 4h + i + 8t as before.  Within a stage of 4 elements
// MFMA step p pairs elements p and p + 2, so a lane reads its two words (one per step) as one 8-byte
// LDS load.  Right-operand fragments of a stage: [step p][limb l][column group cg] (packed tail:
// [p][cg], MFMA column 8 l + c of the 32 carrying limb l of column c).

// One wave's share of a digest: wave reduction, then one atomic (the sum mod 2^64 does not depend on the order
// of arrival).  The digest's odd constant factor is applied here, once per wave.
__device__ __forceinline__ void digest_commit(const GemmArgs &p, int b, int lane, unsigned long long dig) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)dig, off, 64);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(dig >> 32), off, 64);
        dig += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 0) atomicAdd(&p.digests[b], dig * 0x9E3779B97F4A7C15ull);
}

// Start values of the accumulators (GemmArgs::acc_init): limb l of a full tile sits in accumulators 2 l and
// 2 l + 1 of a row group; in a packed tail tile accumulator f carries limb 2 f + (column >> 3) in MFMA column
// 8 (limb & 1) + c; in a narrow one the single accumulator carries limb l in MFMA column 4 l + c.
template <int RG, int NB>
__device__ __forceinline__ void init_acc16(v4i (&acc)[RG][NB], const GemmArgs &p, int lane) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
        for (int f = 0; f < NB; ++f) {
            const int v = (int)(NB == 8 ? p.acc_init[f >> 1] : NB == 2 ? p.acc_init[2 * f + ((lane >> 3) & 1)] : p.acc_init[(lane >> 2) & 3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rg][f][r] = v;
        }
}

// Largest limb sum the 32-bit recombination below takes: 257 a (1 + 2^-15) + 2^31 - 2^16 must stay below 2^32.
constexpr uint32_t kLeanLimbMax = 8355000u;

// sum_l 2^(8 l) a_l as a canonical ring element, for limb sums 0 <= a_l <= kLeanLimbMax (Z/2^M: any int32 read as
// uint32).  Fp: with X = a0 + 2^8 a1 and Y = a2 + 2^8 a3 (both below 2^31), 2^16 Y = 2^31 (Y >> 15) +
// 2^16 (Y & 0x7FFF), and 2^31 = 1 mod p, so the sum is X + (Y >> 15) + ((Y & 0x7FFF) << 16) < 2^32: one fold.
template <int RING>
__device__ __forceinline__ uint32_t combine_limbs_lean(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t mask) {
    if (RING == kRingZpow2) return (a0 + (a1 << 8) + (a2 << 16) + (a3 << 24)) & mask;  // pow2_ring.rs:97-103
    const uint32_t x = (a1 << 8) + a0;
    const uint32_t y = (a3 << 8) + a2;
    const uint32_t s = x + ((y & 0x7FFFu) << 16) + (y >> 15);
    const uint32_t f = (s & kFpP) + (s >> 31);  // <= p
    return min(f, f - kFpP);                    // p -> 0; below p the difference wraps above f
}

// Epilogue of a wave whose 16 RG rows x 32 columns all lie inside the product and hold no gadget entry (the one
// non-zero of a row of G sits in column row / ell): every element is stored, nothing is tested per element,
// nand is a plain negation and the encryption addend meets only zeros.  Same results as the general epilogue
// below, which handles the edge tiles, the packed tail tile and the tiles that cross the gadget's diagonal band.
// Offsets into the output are 32-bit (the caller checked rows * ld_out < 2^30).
template <int RING, int RG, bool NEGATE, bool DIGEST>
__device__ __forceinline__ void gemm16_epilogue_interior(const GemmArgs &p, const TileCtx &c, int tile_rows,
                                                         const v4i (&acc)[RG][8]) {
    const int lane = c.lane, quad = lane >> 4, c16 = lane & 15;
    unsigned char *out = reinterpret_cast<unsigned char *>(p.out + (size_t)slot_of(p.out_slot, c.b) * p.out_stride);
    // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 (lane >> 4) + reg
    const uint32_t idx0 =
        (uint32_t)(c.mt * tile_rows + c.wave * (16 * RG) + 4 * quad) * (uint32_t)p.ld_out + (uint32_t)(c.nt * kTileCols + c16);
    unsigned long long dig = 0;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t idx_row = idx0 + (uint32_t)((rg * 16 + r) * p.ld_out);
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                uint32_t val = combine_limbs_lean<RING>((uint32_t)acc[rg][0 + cg][r], (uint32_t)acc[rg][2 + cg][r],
                                                        (uint32_t)acc[rg][4 + cg][r], (uint32_t)acc[rg][6 + cg][r], c.mask);
                if (NEGATE) {
                    if (RING == kRingFp) {
                        const uint32_t t = kFpP - val;  // 0 -> p -> 0
                        val = min(t, t - kFpP);
                    } else {
                        val = (0u - val) & c.mask;
                    }
                }
                const uint32_t idx = idx_row + 16u * cg;
                *reinterpret_cast<uint32_t *>(out + 4u * idx) = val;
                if (DIGEST) {
                    // (x + 1) * (2 idx + 1): one 32 x 32 -> 64 multiply-add where x + 1 cannot wrap
                    const unsigned w = 2u * idx + 1u;
                    if (RING == kRingFp) dig += (unsigned long long)(val + 1u) * w;
                    else dig += (unsigned long long)val * w + w;
                }
            }
        }
    }
    if (DIGEST) digest_commit(p, c.b, lane, dig);
}

// PACKED: kTailNone (a full tile: 8 accumulators per row group), kTailPacked (2), kTailNarrow (1)
template <int RING, int PACKED, int RG = 4>
__device__ __forceinline__ void gemm16_epilogue(const GemmArgs &p, const TileCtx &c, int tile_rows,
                                                const v4i (&acc)[RG][PACKED == 0 ? 8 : PACKED == 1 ? 2 : 1]) {
    const int lane = c.lane, quad = lane >> 4, c16 = lane & 15;
    const uint32_t addend = (p.mode == kModeEncrypt) ? p.addend[c.b] : 0u;
    uint32_t *out = p.out + (size_t)slot_of(p.out_slot, c.b) * p.out_stride;
#ifdef GSW_EXP_NOEPI
    {   // timing experiment (wrong results): what the kernel costs without its epilogue
        int x = 0;
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int f = 0; f < (PACKED == 0 ? 8 : PACKED == 1 ? 2 : 1); ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) x ^= acc[rg][f][r];
        const int row = c.mt * tile_rows + c.wave * (16 * RG) + quad;
        if (row < p.rows) out[(size_t)row * p.ld_out + c.nt * kTileCols + c16] = (uint32_t)x;
        return;
    }
#endif
    if constexpr (!PACKED) {
        // wave-uniform: are this wave's rows and the tile's columns all inside, away from the gadget's non-zeros?
        const int first_row = c.mt * tile_rows + c.wave * (16 * RG), first_col = c.nt * kTileCols;
        bool lean = first_row + 16 * RG <= p.rows && first_col + kTileCols <= p.out_cols &&
                    p.lean_combine && (unsigned long long)p.rows * (unsigned)p.ld_out < (1ull << 30);
        if (lean && p.mode != kModeMul) {
            const int j_lo = first_row / c.ell, j_hi = (first_row + 16 * RG - 1) / c.ell;
            lean = j_hi < first_col || j_lo >= first_col + kTileCols;
        }
        if (lean) {
            const bool negate = p.mode == kModeNand, digest = p.digests != nullptr;
            if (negate && digest) gemm16_epilogue_interior<RING, RG, true, true>(p, c, tile_rows, acc);
            else if (negate) gemm16_epilogue_interior<RING, RG, true, false>(p, c, tile_rows, acc);
            else if (digest) gemm16_epilogue_interior<RING, RG, false, true>(p, c, tile_rows, acc);
            else gemm16_epilogue_interior<RING, RG, false, false>(p, c, tile_rows, acc);
            return;
        }
    }
    unsigned long long dig = 0;
    const bool want_digest = p.digests != nullptr;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 (lane >> 4) + reg
            const int rr = 4 * quad + r;
            const int row = c.mt * tile_rows + c.wave * (16 * RG) + rg * 16 + rr;
#pragma unroll
            for (int cg = 0; cg < (PACKED ? 1 : 2); ++cg) {
                uint64_t s0, s1, s2, s3;
                int col;
                bool owner;
                if (!PACKED) {
                    constexpr int kAll = PACKED == 0 ? 8 : 1;  // (index folded for the other instantiations)
                    s0 = (uint32_t)acc[rg][(0 * 2 + cg) % kAll][r];
                    s1 = (uint32_t)acc[rg][(1 * 2 + cg) % kAll][r];
                    s2 = (uint32_t)acc[rg][(2 * 2 + cg) % kAll][r];
                    s3 = (uint32_t)acc[rg][(3 * 2 + cg) % kAll][r];
                    col = c.nt * kTileCols + cg * 16 + c16;
                    owner = true;
                } else if (PACKED == kTailPacked) {
                    const int v0 = acc[rg][0][r], v1 = acc[rg][PACKED == kTailPacked ? 1 : 0][r];
                    const int base = (lane & 48) | (lane & 7);
                    s0 = (uint32_t)__shfl(v0, base, 64);
                    s1 = (uint32_t)__shfl(v0, base + 8, 64);
                    s2 = (uint32_t)__shfl(v1, base, 64);
                    s3 = (uint32_t)__shfl(v1, base + 8, 64);
                    col = c.nt * kTileCols + (c16 & 7);
                    owner = c16 < 8;
                } else {
                    // narrow tile: limb l of column cc sits in MFMA column 4 l + cc of the one accumulator
                    const int v0 = acc[rg][0][r];
                    const int base = (lane & 48) | (lane & 3);
                    s0 = (uint32_t)__shfl(v0, base, 64);
                    s1 = (uint32_t)__shfl(v0, base + 4, 64);
                    s2 = (uint32_t)__shfl(v0, base + 8, 64);
                    s3 = (uint32_t)__shfl(v0, base + 12, 64);
                    col = c.nt * kTileCols + (c16 & 3);
                    owner = c16 < 4;
                }
                uint32_t val = ring_from_u64<RING>(s0 + (s1 << 8) + (s2 << 16) + (s3 << 24), c.mask);
                if (owner && row < p.rows && col < p.ld_out) {
                    if (col >= p.out_cols) {
                        val = 0u;  // keep the padding columns of the compact form zero
                    } else if (p.mode == kModeNand) {
                        val = ring_sub<RING>(gadget_entry<RING>(row, col, c.ell, c.mask), val, c.mask);
                    } else if (p.mode == kModeEncrypt) {
                        val = ring_add<RING>(
                            val, ring_mul<RING>(addend, gadget_entry<RING>(row, col, c.ell, c.mask), c.mask), c.mask);
                    }
                    const size_t idx = (size_t)row * p.ld_out + col;
                    out[idx] = val;
                    if (want_digest) {
                        // (x + 1) * (2 idx + 1) as one 32 x 32 -> 64 multiply-add; the constant factor of the
                        // digest is applied once per wave (multiplication mod 2^64 distributes over the sum)
                        const unsigned w = 2u * (unsigned)idx + 1u;
                        dig += (unsigned long long)val * w + w;
                    }
                }
            }
            if (PACKED && row < p.rows) {
                // a packed tile computes 8 columns; zero the padding columns behind them (the row pitch may
                // reach up to 32 columns past the tile's first)
#pragma unroll
                for (int extra = 0; extra < 2; ++extra) {
                    const int zc = c.nt * kTileCols + extra * 16 + c16;
                    if ((extra == 1 || c16 >= (PACKED == kTailNarrow ? 4 : 8)) && zc < p.ld_out) {
                        const size_t idx = (size_t)row * p.ld_out + zc;
                        out[idx] = 0u;
                        if (want_digest) dig += 2u * (unsigned)idx + 1u;
                    }
                }
            }
        }
    }
    if (p.digests) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)dig, off, 64);
            const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(dig >> 32), off, 64);
            dig += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 0) atomicAdd(&p.digests[c.b], dig * 0x9E3779B97F4A7C15ull);
    }
}

// ---- 16x16x64 pipeline with 64-byte left-operand loads.
//
// A 16-byte-per-row load of the left operand touches 64 different cache lines per wave instruction,
// and those requests, not their bytes, dominate the cost of staging.  Here a wave fetches its own
// 64 rows in pieces of 16 rows x 64 bytes (the words of four stages at once: 16 requests per
// instruction), every fourth stage, into a private two-deep LDS ring.  Lane 4r + j of a piece loads
// the 16 bytes of stage j ^ ((r >> 2) & 3) of row r, so that in the lane-linear LDS image row r keeps
// stage s at 64 r + 16 (s ^ ((r >> 2) & 3)): the 8-byte reads of one stage by 16 rows then fall on 16
// different 16-byte slots of the 256-byte bank row (conflict free).  The right operand keeps the
// three-buffer ring; loads past the last stage are clamped re-loads so that the counted waits stay
// static (they land in buffers nobody reads again and are drained before the workgroup ends).

// An 8-byte LDS read that the compiler cannot pair with a neighbour.  Two reads 1 KiB apart would be
// fused into ds_read2st64_b64, which is serviced in groups of 16 lanes with 32 banks (8 LDS cycles,
// and 2-way conflicts on the left-operand image) where ds_read_b64 takes 2 cycles conflict free.
__device__ __forceinline__ uint2 lds_read_b64_alone(uint32_t lds_byte_address) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    // volatile: never merged with a neighbour, and a constant part of the address still goes into the offset field
    const u32x2 v = *reinterpret_cast<__attribute__((address_space(3))) const volatile u32x2 *>((uintptr_t)lds_byte_address);
    return make_uint2(v.x, v.y);
}

#ifndef GSW_SPREAD_LOADS
#define GSW_SPREAD_LOADS 1
#endif

template <int WAVES>
constexpr int rowblock_lds_bytes() { return 2 * WAVES * 4096 + kPipeBuffers * kStageElems * 4 * kFragBytes; }

// PACKED: kTailNone (full tile), kTailPacked (tail tile of 32 MFMA columns), kTailNarrow (of 16: its stages are
// two fragments, 2 KiB apart in the prepared operand).
template <int RING, int WAVES, int PACKED, int SCALE = 0>
__device__ __forceinline__ void gemm16_rowblock_body(const GemmArgs &p, int ell, uint32_t mask, int b, int nt,
                                                     int mt, unsigned char *smem, int tid = (int)threadIdx.x) {
    constexpr int kRows = 64 * WAVES;
    constexpr int kLhsRing = 2 * WAVES * 4096;          // two super-stages of 64 B per row
    constexpr int kRhsStage = kStageElems * 4 * kFragBytes;
    constexpr int kNB = PACKED == kTailNone ? 8 : PACKED == kTailPacked ? 2 : 1;
    constexpr int kPieces = 2 * kNB;
    constexpr int kRounds = (kPieces + WAVES - 1) / WAVES;

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)smem);

    const uint32_t *lhs = p.lhs + (size_t)slot_of(p.lhs_slot, b) * p.lhs_stride;
    const uint8_t *rhs = p.rhs + slot_of(p.rhs_slot, b) * p.rhs_stride + (size_t)nt * p.e4 * (4 * kFragBytes);
    const int stages = p.e4 / kStageElems;
    const bool wave_active = mt * kRows + wave * 64 < p.rows;

    // left-operand piece rg: lane 4r + j -> row 16 rg + r of this wave, stage slot j
    const int ld_r = lane >> 2, ld_j = (lane & 3) ^ ((ld_r >> 2) & 3);
    const uint32_t *a_row[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        int row = mt * kRows + wave * 64 + rg * 16 + ld_r;
        if (row > p.rows - 1) row = p.rows - 1;  // clamped rows are computed but never stored
        a_row[rg] = lhs + (size_t)row * p.ld_lhs;
    }

    v4i acc[4][kNB];
    init_acc16(acc, p, lane);
    const int r16 = lane & 15;
    const int sh = ((lane >> 4) & 1) * 4;
    const int swz = (r16 >> 2) & 3;
    const int word_pair = (lane >> 5) * 8;

    auto issue_lhs_piece = [&](int super_stage, int rg) {
        int st = super_stage * 4 + ld_j;
        if (st > stages - 1) st = stages - 1;  // past the end: re-load the last stage (never read)
        const uint32_t base = lds0 + (uint32_t)((super_stage & 1) * (WAVES * 4096) + wave * 4096);
        lds_dma16(a_row[rg] + st * kStageElems, __builtin_amdgcn_readfirstlane(base + (uint32_t)(rg * 1024)));
    };
    auto issue_lhs = [&](int super_stage) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) issue_lhs_piece(super_stage, rg);
    };
    const uint32_t rhs_lane_offset = (uint32_t)((wave % kPieces) * kFragBytes + lane * 16);
    auto issue_rhs_round = [&](int t, int buf, int j) {
        if (t > stages - 1) t = stages - 1;
        const uint32_t base = lds0 + (uint32_t)(kLhsRing + buf * kRhsStage);
        const uint8_t *stage_src = rhs + (size_t)t * (kPieces * kFragBytes);
        const int piece = (j * WAVES + wave) % kPieces;
        if constexpr (kRounds == 1)  // the packed tiles: a uniform base and one per-lane offset, no address arithmetic
            lds_dma16_base<0>(stage_src, rhs_lane_offset, __builtin_amdgcn_readfirstlane(base + (uint32_t)(piece * kFragBytes)));
        else
            lds_dma16(stage_src + piece * kFragBytes + lane * 16,
                      __builtin_amdgcn_readfirstlane(base + (uint32_t)(piece * kFragBytes)));
    };
    auto issue_rhs = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < kRounds; ++j) issue_rhs_round(t, buf, j);
    };

    issue_lhs(0);
    issue_rhs(0, 0);
    issue_rhs(1, 1);
    int buf = 0;
    auto stage_body = [&](auto slot_tag, auto active_tag, int t) {
        constexpr int kSlot = decltype(slot_tag)::value;      // t mod 4
        constexpr bool kActive = decltype(active_tag)::value;
        // The previous iteration issued kRounds right-operand loads, plus 4 left-operand loads when it
        // opened a super-stage; everything older has landed once only those are pending.
        wait_vmcnt_then_barrier<kRounds + (kSlot == 1 ? 4 : 0)>();
        // An active wave spreads this stage's loads between its MFMA groups (same number, same order
        // class for the counted wait): issued in one burst right after the barrier, the four waves of
        // the workgroup queue up at the texture addresser while their MFMA pipes sit idle.
        constexpr bool kSpread = kActive && GSW_SPREAD_LOADS;
        if (!kSpread) {
            if (kSlot == 0) issue_lhs((t >> 2) + 1);
            issue_rhs(t + 2, buf == 0 ? 2 : buf - 1);
        }
        if constexpr (kActive) {
            const uint32_t sa = lds0 + (uint32_t)(((t >> 2) & 1) * (WAVES * 4096) + wave * 4096 + r16 * 64 +
                                                  ((kSlot ^ swz) * 16) + word_pair);
            const unsigned char *sb = smem + kLhsRing + buf * kRhsStage + lane * 16;
            uint2 w[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                w[rg] = lds_read_b64_alone(sa + (uint32_t)(rg * 1024));
            }
            constexpr int kGroup = PACKED == kTailNone ? 4 : kNB;  // a packed tile: one group per MFMA step
            constexpr int kGroups = 2 * kNB / kGroup;
            v4i fb[2][kGroup];
            v4i fa[4];
#pragma unroll
            for (int j = 0; j < kGroup; ++j) fb[0][j] = *reinterpret_cast<const v4i *>(sb + j * kFragBytes);
#pragma unroll
            for (int grp = 0; grp < kGroups; ++grp) {
                const int cur = grp & 1;
                if (grp + 1 < kGroups) {
#pragma unroll
                    for (int j = 0; j < kGroup; ++j)
                        fb[cur ^ 1][j] = *reinterpret_cast<const v4i *>(sb + ((grp + 1) * kGroup + j) * kFragBytes);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above this group's MFMAs
                if constexpr (kSpread) {
                    // kGroups groups carry kRounds right-operand rounds and 4 left-operand pieces
                    constexpr int kLhsPer = 4 / kGroups, kRhsPer = (kRounds + kGroups - 1) / kGroups;
                    if (kSlot == 0) {
#pragma unroll
                        for (int k = 0; k < kLhsPer; ++k) issue_lhs_piece((t >> 2) + 1, grp * kLhsPer + k);
                    }
#pragma unroll
                    for (int k = 0; k < kRhsPer; ++k)
                        if (grp * kRhsPer + k < kRounds) issue_rhs_round(t + 2, buf == 0 ? 2 : buf - 1, grp * kRhsPer + k);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int step = grp * kGroup / kNB;
                const int f0 = grp * kGroup - step * kNB;
                if (f0 == 0) {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) fa[rg] = expand_bits<SCALE>((step == 0 ? w[rg].x : w[rg].y) >> sh);
                }
#pragma unroll
                for (int j = 0; j < kGroup; ++j) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int rg = (j & 1) ? 3 - k : k;  // back and forth: one operand changes between neighbours
                        acc[rg][f0 + j] =
                            __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[rg], fb[cur][j], acc[rg][f0 + j], 0, 0, 0);
                    }
                }
            }
        }
        buf = buf == 2 ? 0 : buf + 1;
    };
    auto main_loop = [&](auto active_tag) {
        for (int t0 = 0; t0 < stages; t0 += 4) {
            stage_body(std::integral_constant<int, 0>{}, active_tag, t0);
            if (t0 + 1 < stages) stage_body(std::integral_constant<int, 1>{}, active_tag, t0 + 1);
            if (t0 + 2 < stages) stage_body(std::integral_constant<int, 2>{}, active_tag, t0 + 2);
            if (t0 + 3 < stages) stage_body(std::integral_constant<int, 3>{}, active_tag, t0 + 3);
        }
    };
    if (wave_active) main_loop(std::true_type{});
    else main_loop(std::false_type{});
    // the clamped re-loads issued by the last iterations must land before this workgroup's LDS is reused
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave_active) {
        const TileCtx c{b, nt, mt, wave, lane, ell, mask};
        gemm16_epilogue<RING, PACKED>(p, c, kRows, acc);
    }
}

// The tall-tile experiment (variants 10 / 11): only in libraries built with -DGSW_EXP_TALL.
#ifdef GSW_EXP_TALL
#include "gsw_kernels_tall.inc"
#endif


// ---- Skewed 256-row pipeline (full column tiles).
//
// Same tile, rings and loads as the row-block kernel above, but a wave's instruction stream is software
// pipelined so that it never stops issuing MFMAs for its own bookkeeping:
//  * the last MFMA group of stage t runs after the barrier that opens stage t + 1, covering the latency
//    of the first right-operand fragment reads of that stage;
//  * the left-operand words of stage t + 1 (a private ring: no barrier needed) are read, counted and
//    expanded into int8 fragments between the MFMAs of stage t, and the step-1 fragments of a stage
//    between the MFMAs of its step 0.
// Two fragment sets (fa0 / fa1) alternate; the right-operand fragments keep two register groups.

#ifndef GSW_SKEW_INTERLEAVE
#define GSW_SKEW_INTERLEAVE 1
#endif

// Loads a wave may leave in flight when it enters a stage in slot `slot` of a super-stage (`first`: the very first
// stage).  A stage issues, in this order, portion q = 0..3: piece q of the next super-stage's left operand when the
// stage opens a super-stage (slot 0), then round q of the right operand DEPTH - 1 stages ahead when q < rounds.
// Entering stage t needs the right operand of stage t (issued DEPTH - 1 stages ago) and, in slot 0, the left
// operand issued four stages ago; everything issued after those may still be under way.
constexpr int skew_loads_in_flight(int rounds, int depth, int slot, bool first) {
    if (first) return (depth - 2) * rounds;  // the prologue: left operand, then the stages in order
    auto opens = [](int s) { return ((s % 4) + 4) % 4 == 0; };
    // after the last right-operand round of stage t - (depth - 1): the left pieces that stage still issued
    int n = opens(slot - (depth - 1)) && rounds < 4 ? 4 - rounds : 0;
    for (int d = depth - 2; d >= 1; --d) n += rounds + (opens(slot - d) ? 4 : 0);
    if (slot == 0) {
        // after the last left piece of stage t - 4: that stage's later right-operand rounds, then stages t - 3 .. t - 1
        const int after_left = (rounds == 4 ? 1 : 0) + 3 * rounds;
        if (after_left < n) n = after_left;
    }
    return n;
}
static_assert(skew_loads_in_flight(4, 3, 0, true) == 4 && skew_loads_in_flight(4, 3, 0, false) == 4 &&
              skew_loads_in_flight(4, 3, 1, false) == 8 && skew_loads_in_flight(4, 3, 2, false) == 4 &&
              skew_loads_in_flight(4, 3, 3, false) == 4, "the four-wave, three-buffer pipeline's counts");

// Where a tile body gets its next tile from.  NoTileStream: nowhere, the body computes the one tile it was given.
struct NoTileStream {
    static constexpr bool kOn = false;
};

template <int RING, int SCALE, int WAVES = 4, int DEPTH = 3, class STREAM = NoTileStream>
__device__ __forceinline__ void gemm16_skew_body(const GemmArgs &p, int ell, uint32_t mask, int b, int nt, int mt,
                                                 unsigned char *smem, int tid = (int)threadIdx.x,
                                                 STREAM *stream = nullptr) {
    static_assert((WAVES == 4 || WAVES == 8) && DEPTH >= 3 && DEPTH <= 5, "left operand: two super-stages of four stages");
    constexpr int kRows = 64 * WAVES;
    constexpr int kLhsRing = 2 * WAVES * 4096;
    constexpr int kRhsStage = kStageElems * 4 * kFragBytes;
    constexpr int kNB = 8;
    constexpr int kPieces = 2 * kNB;
    constexpr int kRounds = kPieces / WAVES;  // four waves: 4, one per MFMA group

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)smem);

    const int stages = p.e4 / kStageElems;
    const int ld_r = lane >> 2, ld_j = (lane & 3) ^ ((ld_r >> 2) & 3);
    // the tile at hand (a streaming body moves on to further tiles)
    const uint8_t *rhs;
    bool wave_active;
    const uint32_t *a_row[4];
    auto set_tile = [&]() {
        const uint32_t *lhs = p.lhs + (size_t)slot_of(p.lhs_slot, b) * p.lhs_stride;
        rhs = p.rhs + slot_of(p.rhs_slot, b) * p.rhs_stride + (size_t)nt * p.e4 * (4 * kFragBytes);
        wave_active = mt * kRows + wave * 64 < p.rows;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            int row = mt * kRows + wave * 64 + rg * 16 + ld_r;
            if (row > p.rows - 1) row = p.rows - 1;
            a_row[rg] = lhs + (size_t)row * p.ld_lhs;
        }
    };
    set_tile();

    v4i acc[4][kNB];
    const int r16 = lane & 15;
    const int sh = ((lane >> 4) & 1) * 4;
    const int swz = (r16 >> 2) & 3;
    const int word_pair = (lane >> 5) * 8;

    auto issue_lhs_piece = [&](int super_stage, int rg) {
        int st = super_stage * 4 + ld_j;
        if (st > stages - 1) st = stages - 1;
        const uint32_t base = lds0 + (uint32_t)((super_stage & 1) * (WAVES * 4096) + wave * 4096);
        lds_dma16(a_row[rg] + st * kStageElems, __builtin_amdgcn_readfirstlane(base + (uint32_t)(rg * 1024)));
    };
    // Round j of a stage: the waves load pieces j * WAVES .. + WAVES - 1, wave w the piece j * WAVES + w, so that the
    // rounds bring the fragments in the order the MFMA groups read them: the round that lands last -- its count may
    // have left vmcnt only just before the stage's barrier -- holds the fragments read last, a thousand cycles later.
    // (Handing every wave four consecutive pieces instead, with the round in the instruction's constant, reads a
    // just-landed piece right after the barrier, and one product in 10^4 came out wrong.)
    // The address is a uniform base (scalar arithmetic) plus one per-lane offset: no vector instruction spent on it.
    const uint32_t rhs_lane_offset = (uint32_t)(wave * kFragBytes + lane * 16);
    auto issue_rhs_round = [&](int t, int buf, auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        if (t > stages - 1) t = stages - 1;
        const uint32_t base = lds0 + (uint32_t)(kLhsRing + buf * kRhsStage);
        const uint8_t *round_src = rhs + (size_t)t * (kPieces * kFragBytes) + j * WAVES * kFragBytes;
        lds_dma16_base<0>(round_src, rhs_lane_offset,
                          __builtin_amdgcn_readfirstlane(base + (uint32_t)((j * WAVES + wave) * kFragBytes)));
    };
    // portion q (0..3) of the loads of the stage-t epoch: piece q of the next super-stage's left operand when t
    // opens a super-stage, and right-operand round q of stage t + DEPTH - 1 (into the buffer stage t - 1 has left)
    auto issue_portion = [&](int t, int buf, bool opens, auto q_tag) {
        constexpr int q = decltype(q_tag)::value;
        if (opens) issue_lhs_piece((t >> 2) + 1, q);
        if constexpr (q < kRounds) issue_rhs_round(t + DEPTH - 1, buf == 0 ? DEPTH - 1 : buf - 1, q_tag);
    };
    auto read_words = [&](int t, uint32_t (&wx)[4], uint32_t (&wy)[4]) {
        const uint32_t sa = lds0 + (uint32_t)(((t >> 2) & 1) * (WAVES * 4096) + wave * 4096 + r16 * 64 +
                                              (((t & 3) ^ swz) * 16) + word_pair);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const uint2 w = lds_read_b64_alone(sa + (uint32_t)(rg * 1024));
            wx[rg] = w.x;
            wy[rg] = w.y;
        }
    };
    // keeps the compiler from sinking a fragment set's expansion below this point
    auto pin = [&](v4i (&fa)[4]) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) asm volatile("" : "+v"(fa[rg]));
    };
    auto expand = [&](v4i (&fa)[4], const uint32_t (&word)[4]) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) fa[rg] = expand_bits<SCALE>(word[rg] >> sh);
    };
    auto read_frags = [&](v4i (&fb)[4], int buf, int grp) {
        const unsigned char *sb = smem + kLhsRing + buf * kRhsStage + lane * 16 + grp * 4 * kFragBytes;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const v4i *>(sb + j * kFragBytes);
    };
    auto mfma_group = [&](const v4i (&fa)[4], const v4i (&fb)[4], int f0) {
        // Four MFMAs share a right fragment (the operand with the random bytes), and the row groups go back and forth
        // so that only one operand changes from one MFMA to the next: the chip is power-limited, and this order is
        // 0.6-0.9 % faster than the plain nested loops, 1.5-1.8 % faster than four MFMAs sharing the left fragment.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rg = (j & 1) ? 3 - k : k;
                acc[rg][f0 + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[rg], fb[j], acc[rg][f0 + j], 0, 0, 0);
            }
        }
    };
    // 16 MFMAs with V vector ALU instructions after each of them (scheduler directive, no code)
    auto interleave = [&](auto valu_tag) {
#if GSW_SKEW_INTERLEAVE
        constexpr int V = decltype(valu_tag)::value;
#define GSW_SGB1                                       \
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);   \
    __builtin_amdgcn_sched_group_barrier(0x2, V, 0);
#define GSW_SGB4 GSW_SGB1 GSW_SGB1 GSW_SGB1 GSW_SGB1
        GSW_SGB4 GSW_SGB4 GSW_SGB4 GSW_SGB4
#undef GSW_SGB4
#undef GSW_SGB1
#endif
    };

    // prologue: left operand of super-stage 0, right operand of stages 0 .. DEPTH - 2
    auto issue_prologue = [&]() {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) issue_lhs_piece(0, rg);
#pragma unroll
        for (int st = 0; st < DEPTH - 1; ++st) {
            issue_rhs_round(st, st, std::integral_constant<int, 0>{});
            issue_rhs_round(st, st, std::integral_constant<int, 1>{});
            if constexpr (kRounds == 4) {
                issue_rhs_round(st, st, std::integral_constant<int, 2>{});
                issue_rhs_round(st, st, std::integral_constant<int, 3>{});
            }
        }
    };
    issue_prologue();

    v4i fa0[4], fa1[4], fbA[4], fbB[4];
    uint32_t wy[4];  // the second words (MFMA step 1) of the current stage
    int buf;

    // kHalf: the last stage of a K dimension whose last real word is the stage's first (n + 1 = 1 mod 4, as for
    // every n that is a multiple of 4): MFMA step 1 pairs the stage's words 1 and 3, which meet only zero digits,
    // so its two groups, their fragment reads and the expansion of its left fragments are left out.
    auto stage = [&](auto slot_tag, auto first_tag, auto active_tag, int t, auto half_tag) {
        constexpr int kSlot = decltype(slot_tag)::value;
        constexpr bool kFirst = decltype(first_tag)::value;
        constexpr bool kActive = decltype(active_tag)::value;
        constexpr bool kHalf = decltype(half_tag)::value;
        wait_loads_and_own_reads_then_barrier<skew_loads_in_flight(kRounds, DEPTH, kSlot, kFirst)>();
        const bool opens = kSlot == 0;
        if constexpr (!kActive) {
            issue_portion(t, buf, opens, std::integral_constant<int, 0>{});
            issue_portion(t, buf, opens, std::integral_constant<int, 1>{});
            issue_portion(t, buf, opens, std::integral_constant<int, 2>{});
            issue_portion(t, buf, opens, std::integral_constant<int, 3>{});
        } else {
            if constexpr (kFirst) {
                uint32_t wx[4];
                read_words(t, wx, wy);
                expand(fa0, wx);
            }
            // phase 0: first fragments of this stage; the previous stage's last group covers their latency
            read_frags(fbA, buf, 0);
            issue_portion(t, buf, opens, std::integral_constant<int, 0>{});
            if constexpr (!kFirst) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(fa1, fbB, 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            // phase 1: step 0, fragments 0..3; the step-1 fragments of this stage are expanded meanwhile
            read_frags(fbB, buf, 1);
            issue_portion(t, buf, opens, std::integral_constant<int, 1>{});
            mfma_group(fa0, fbA, 0);
            if constexpr (!kHalf) {
                expand(fa1, wy);
                pin(fa1);
                interleave(std::integral_constant<int, 2>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            // phase 2: step 0, fragments 4..7; the next stage's left-operand words are fetched
            if constexpr (!kHalf) read_frags(fbA, buf, 2);
            issue_portion(t, buf, opens, std::integral_constant<int, 2>{});
            mfma_group(fa0, fbB, 4);
            if constexpr (kHalf) {
                issue_portion(t, buf, opens, std::integral_constant<int, 3>{});  // the counted waits of the loads stay as they are
            } else {
                uint32_t nx[4], ny[4];
                read_words(t + 1, nx, ny);
                __builtin_amdgcn_sched_barrier(0);
                // phase 3: step 1, fragments 0..3; the next stage's step-0 fragments are expanded
                read_frags(fbB, buf, 3);
                issue_portion(t, buf, opens, std::integral_constant<int, 3>{});
                mfma_group(fa1, fbA, 0);
                expand(fa0, nx);
                pin(fa0);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) wy[rg] = ny[rg];
                interleave(std::integral_constant<int, 2>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            // step 1, fragments 4..7 (fa1 x fbB) is left for the next stage's phase 0
        }
        buf = buf == DEPTH - 1 ? 0 : buf + 1;
    };
    // the last stage may be a half one; it is the only stage of a loop turn when stages = 1 mod 4 (slot 0)
    const bool half_last = p.k_words > 0 && p.k_words + 3 == p.e4 && stages > 4 && (stages & 3) == 1;
    const int full_stages = half_last ? stages - 1 : stages;
    auto main_loop = [&](auto active_tag) {
        stage(std::integral_constant<int, 0>{}, std::true_type{}, active_tag, 0, std::false_type{});
        if (1 < full_stages) stage(std::integral_constant<int, 1>{}, std::false_type{}, active_tag, 1, std::false_type{});
        if (2 < full_stages) stage(std::integral_constant<int, 2>{}, std::false_type{}, active_tag, 2, std::false_type{});
        if (3 < full_stages) stage(std::integral_constant<int, 3>{}, std::false_type{}, active_tag, 3, std::false_type{});
        for (int t0 = 4; t0 < full_stages; t0 += 4) {
            stage(std::integral_constant<int, 0>{}, std::false_type{}, active_tag, t0, std::false_type{});
            if (t0 + 1 < full_stages)
                stage(std::integral_constant<int, 1>{}, std::false_type{}, active_tag, t0 + 1, std::false_type{});
            if (t0 + 2 < full_stages)
                stage(std::integral_constant<int, 2>{}, std::false_type{}, active_tag, t0 + 2, std::false_type{});
            if (t0 + 3 < full_stages)
                stage(std::integral_constant<int, 3>{}, std::false_type{}, active_tag, t0 + 3, std::false_type{});
        }
        if (half_last) stage(std::integral_constant<int, 0>{}, std::false_type{}, active_tag, stages - 1, std::true_type{});
    };
    for (;;) {
        init_acc16(acc, p, lane);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) wy[rg] = 0u;
        buf = 0;
        if (wave_active) main_loop(std::true_type{});
        else main_loop(std::false_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const TileCtx c{b, nt, mt, wave, lane, ell, mask};
        const bool was_active = wave_active;
        bool primed = false;
        if constexpr (STREAM::kOn) {
            // Every wave is past its last read of the tile's stages after the barrier in here; the next full tile's first
            // loads go out before this tile's last MFMA group and epilogue, which cover their latency.
            primed = stream->next_full_tile(smem, tid, b, nt, mt);
            if (primed) {
                set_tile();
                issue_prologue();
            }
        }
        if (was_active) {
            if (!half_last) mfma_group(fa1, fbB, 4);  // the last stage's deferred group
            gemm16_epilogue<RING, false>(p, c, kRows, acc);
        }
        if (!primed) return;
    }
}

#ifdef GSW_EXP_TILE_CLOCK
// diagnostic build: shader-clock cycles and count of the workgroups of each kind
// (0 full tile, 1 last column tile, 2 last row tile, 3 both)
// [8..10]: earliest start, latest end and the sum of the workgroups' durations on the constant-rate counter
// (s_memrealtime): sum / (resident workgroups x (end - start)) is the share of the slots' time that was occupied
__device__ unsigned long long g_tile_clock[16];
hipError_t read_tile_clock(unsigned long long *out16, bool reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tile_clock), 128);
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {};
        z[8] = ~0ull;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_tile_clock), z, 128);
    }
    return e;
}
#endif

template <int RING, int SCALE>
__global__ __launch_bounds__(256, 2) void binlimb_gemm16_skew(GemmArgs p, int ell, uint32_t mask, int nt_count,
                                                              int mt_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int b, nt, mt;
    if (!map_workgroup_blocked(p.batch, nt_count, mt_count, p.pass_rows, p.pass_cols, b, nt, mt)) return;
#ifdef GSW_EXP_TILE_CLOCK
    struct TileClock {
        unsigned long long t0, r0;
        int kind;
        __device__ TileClock(int k) : t0(__builtin_readcyclecounter()), r0(__builtin_amdgcn_s_memrealtime()), kind(k) {}
        __device__ ~TileClock() {
            if (threadIdx.x == 0) {
                const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
                atomicAdd(&g_tile_clock[2 * kind], __builtin_readcyclecounter() - t0);
                atomicAdd(&g_tile_clock[2 * kind + 1], 1ull);
                atomicMin(&g_tile_clock[8], r0);
                atomicMax(&g_tile_clock[9], r1);
                atomicAdd(&g_tile_clock[10], r1 - r0);
            }
        }
    } tile_clock((nt == nt_count - 1 ? 1 : 0) + (mt == mt_count - 1 ? 2 : 0));
#endif
#if defined(GSW_EXP_DEPHASE) || defined(GSW_EXP_PRIO)
    // experiments on the two workgroups that share a CU (ids 256 apart: 8 XCDs x 32 CUs are filled first)
    const bool second = (blockIdx.x >> 8) & 1;
#endif
#ifdef GSW_EXP_DEPHASE
    if (second)
        for (int i = 0; i < GSW_EXP_DEPHASE; ++i) __builtin_amdgcn_s_sleep(16);  // 16 x 64 cycles each
#endif
#ifdef GSW_EXP_PRIO
    if (second) __builtin_amdgcn_s_setprio(GSW_EXP_PRIO);
#endif
    if (p.packed_tail && nt == nt_count - 1) {
        if (p.packed_tail == kTailNarrow) gemm16_rowblock_body<RING, 4, kTailNarrow, SCALE>(p, ell, mask, b, nt, mt, smem);
        else gemm16_rowblock_body<RING, 4, kTailPacked, SCALE>(p, ell, mask, b, nt, mt, smem);
    } else {
        gemm16_skew_body<RING, SCALE>(p, ell, mask, b, nt, mt, smem);
    }
}

#if defined(GSW_EXP_WIDE8) || defined(GSW_EXP_PERSIST)
#include "gsw_kernels_exp.inc"
#endif

// Superseded contraction kernels (variants 0, 1, 4..8) and timing-only ablation builds: compiled only into
// libraries built with -DGSW_DIAGNOSTIC_BUILDS, for A/B timing.  The shipped library holds the default
// kernel (variant 9) and the simple kernel below (variants 2, 3) as an independent cross-check.
#ifdef GSW_DIAGNOSTIC_BUILDS
#include "gsw_kernels_diag.inc"
#endif

// Counter calibration: streams `bytes` (a multiple of 64 KiB) once through the same LDS-DMA instruction
// the contraction uses, 1 KiB per wave instruction (mode 0) or 16 rows x 64 B at `pitch` bytes (mode 1),
// so that FETCH_SIZE can be compared with a known byte count in the kernel's own access patterns.
__global__ __launch_bounds__(256) void stream_read_kernel(const uint8_t *src, size_t bytes, int mode, size_t pitch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)smem);
    const size_t pieces = bytes / 1024;
    const size_t stride = (size_t)gridDim.x * 4;
    for (size_t piece = (size_t)blockIdx.x * 4 + wave; piece < pieces; piece += stride) {
        const uint8_t *g = mode == 0 ? src + piece * 1024 + lane * 16
                                     : src + (piece * 16 + (lane >> 2)) * pitch + (lane & 3) * 16;
        lds_dma16(g, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 1024u));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

hipError_t launch_stream_read(const void *src, size_t bytes, int mode, size_t pitch, hipStream_t stream) {
    hipLaunchKernelGGL(stream_read_kernel, dim3(2048), dim3(256), 4096, stream, (const uint8_t *)src, bytes, mode,
                       pitch);
    return hipGetLastError();
}

// ---- simple kernel: two LDS buffers, compiler-managed waits (LDS-DMA builtin or register staging).
// Kept as the straightforward form of the same contraction, for A/B timing and as a cross-check.

constexpr int kStageLhsBytes = kTileRows * 16;
constexpr int kStageRhsBytes = kStageElems * 4 * kFragBytes;
constexpr int kStageBytes = kStageLhsBytes + kStageRhsBytes;
constexpr int kGemmLds = 2 * kStageBytes;

template <bool DMA>
__device__ __forceinline__ void stage_copy16(const void *gsrc, unsigned char *lds_wave_base, int lane) {
    if (DMA) {
        // LDS-DMA: the destination is the wave-uniform base + 16 * lane
        __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
    } else {
        const uint4 v = *reinterpret_cast<const uint4 *>(gsrc);
        *reinterpret_cast<uint4 *>(lds_wave_base + 16 * lane) = v;
    }
}

template <int RING, bool DMA>
__global__ __launch_bounds__(256, 2) void binlimb_gemm_kernel(GemmArgs p, int ell, uint32_t mask, int nt_count,
                                                               int mt_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, nt, mt;
    if (!map_workgroup(p.batch, nt_count, mt_count, b, nt, mt)) return;

    const uint32_t *lhs = p.lhs + (size_t)slot_of(p.lhs_slot, b) * p.lhs_stride;
    const uint8_t *rhs = p.rhs + slot_of(p.rhs_slot, b) * p.rhs_stride + (size_t)nt * p.e4 * (4 * kFragBytes);
    int row_load = mt * kTileRows + tid;
    if (row_load > p.rows - 1) row_load = p.rows - 1;  // clamped rows are computed but never stored
    const uint32_t *a_src = lhs + (size_t)row_load * p.ld_lhs;
    const uint8_t *b_src = rhs + 16 * tid;
    const int stages = p.e4 / kStageElems;

    v16i acc[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int l = 0; l < 4; ++l)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][l][r] = (int)p.acc_init[l];
    const int r32 = lane & 31;
    const int sh = (lane >> 5) * 4;

    auto stage = [&](int t, int buf) {
        unsigned char *base = smem + buf * kStageBytes;
        stage_copy16<DMA>(a_src + t * kStageElems, base + wave * 1024, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            stage_copy16<DMA>(b_src + (size_t)t * kStageRhsBytes + j * 4096,
                              base + kStageLhsBytes + j * 4096 + wave * 1024, lane);
    };

    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < stages; ++t) {
        const int buf = t & 1;
        if (t + 1 < stages) stage(t + 1, buf ^ 1);
        const unsigned char *sa = smem + buf * kStageBytes;
        const unsigned char *sb = sa + kStageLhsBytes;
        const uint4 a0 = *reinterpret_cast<const uint4 *>(sa + (wave * 64 + r32) * 16);
        const uint4 a1 = *reinterpret_cast<const uint4 *>(sa + (wave * 64 + 32 + r32) * 16);
        gemm_kstep<4>(a0.x, a1.x, sh, sb + 0 * 4 * kFragBytes, lane, acc);
        gemm_kstep<4>(a0.y, a1.y, sh, sb + 1 * 4 * kFragBytes, lane, acc);
        gemm_kstep<4>(a0.z, a1.z, sh, sb + 2 * 4 * kFragBytes, lane, acc);
        gemm_kstep<4>(a0.w, a1.w, sh, sb + 3 * 4 * kFragBytes, lane, acc);
        __syncthreads();
    }
    const TileCtx c{b, nt, mt, wave, lane, ell, mask};
    gemm_epilogue<RING, 4>(p, c, kTileRows, acc);
}

// Start values C_l of the four limb accumulators for Fp (for Z/2^M see launch_gemm: a negative sum read as uint32
// is off by a multiple of 2^32, so nothing is needed for correctness there).  A limb sum lies in [-128 k, 127 k] for k K-slots, so C_l >= 128 k keeps it
// non-negative, and sum_l 2^(8 l) C_l = 0 mod p leaves the recombined value unchanged.  With all C_l + 127 k
// below `top` (kLeanLimbMax for the 32-bit recombination, else 2^28 for the 64-bit one).  C_3 and C_2 are
// stepped until the target -(2^16 C_2 + 2^24 C_3) mod p has a representative C_0 + 2^8 C_1 in range, which
// covers every integer of [257 lo, 257 hi] once hi - lo >= 255.
static bool fp_acc_init(long units, long top, uint32_t (&c)[4]) {
    const long lo = 128 * units, hi = top - 127 * units;
    if (hi - lo < 255) return false;
    const long P = (long)kFpP;
    const long span = hi - lo < 2048 ? hi - lo : 2048;
    for (long c3 = lo; c3 <= lo + span; ++c3) {
        for (long c2 = lo; c2 <= lo + span; ++c2) {
            const long t = (P - (long)((((unsigned __int128)c2 << 16) + ((unsigned __int128)c3 << 24)) % (unsigned long)P)) % P;
            long v = t;
            if (v < 257 * lo) v += (257 * lo - v + P - 1) / P * P;
            if (v > 257 * hi) continue;
            long c1 = (v - lo) / 256;
            if (c1 > hi) c1 = hi;
            const long c0 = v - 256 * c1;
            if (c0 < lo || c0 > hi || c1 < lo) continue;
            c[0] = (uint32_t)c0, c[1] = (uint32_t)c1, c[2] = (uint32_t)c2, c[3] = (uint32_t)c3;
            return true;
        }
    }
    return false;
}

long scale_units(int e4, int mode) { return (long)e4 * (mode == 2 ? 120 : mode == 1 ? 48 : 32); }

int bit_scale_mode(const RingDesc &ring, int e4) {
    if (ring.kind != kRingFp) return 0;
    // the widest scaling whose limb sums still fit the 32-bit recombination; failing that (very long
    // contractions: the 64-bit one either way) the widest that fits its bound
    for (int mode = 2; mode >= 0; --mode)
        if (255 * scale_units(e4, mode) + 255 <= (long)kLeanLimbMax) return mode;
    for (int mode = 2; mode >= 1; --mode)
        if (255 * scale_units(e4, mode) + 255 <= (1L << 28)) return mode;
    return 0;
}

bool limb_start_values(const RingDesc &ring, long k_slots, bool wide, uint32_t (&c)[4], int *lean) {
    *lean = 1;
    if (ring.kind != kRingFp) {
        // Z/2^M needs no start values (a sum read as uint32 is right mod 2^32 whatever its sign), but sums that
        // hover around 0 change sign all the time and every change flips the upper bits of the accumulator: start
        // values that vanish mod 2^32 (2^24 + 2^8 (2^24 - 2^16) = 2^32) keep them positive, 1.5 % faster at n = 512
        c[0] = 1u << 24, c[1] = (1u << 24) - (1u << 16), c[2] = 1u << 24, c[3] = 1u << 24;
        if (wide) *lean = 0;  // every tile through the general epilogue (64-bit sum, then the mask)
        return true;
    }
    if (!wide && fp_acc_init(k_slots, (long)kLeanLimbMax, c)) return true;
    *lean = 0;
    return fp_acc_init(k_slots, 1L << 28, c);
}

hipError_t launch_gemm(const RingDesc &ring, const GemmArgs &args, int variant, hipStream_t stream, int *lean_used) {
    if (args.batch <= 0) return hipSuccess;
    GemmArgs a = args;
    {
        // every left-operand word carries at most 32 ones; K slots past the right operand meet the digit 0.
        // GSW_FORCE_WIDE_COMBINE=1 (tests): the 64-bit recombination and its start values at any size.
        thread_local long cached_slots = -1;
        thread_local int cached_kind = -1, cached_wide = -1, cached_lean = 0;
        thread_local uint32_t cached_init[4];
        if (a.scale_mode < 0 || a.scale_mode > 2 || (a.scale_mode && (ring.kind != kRingFp || (variant != 9 && variant != 12)))) return hipErrorInvalidValue;
        // the limb sums' bound in units of one {0, 1} K slot (scaled bits weigh 2^c)
        const long k_slots = scale_units(a.e4, a.scale_mode);
        const char *env = getenv("GSW_FORCE_WIDE_COMBINE");
        const int wide = env && env[0] == '1';
        if (k_slots != cached_slots || ring.kind != cached_kind || wide != cached_wide) {
            if (!limb_start_values(ring, k_slots, wide, cached_init, &cached_lean)) return hipErrorInvalidValue;
            cached_slots = k_slots, cached_kind = ring.kind, cached_wide = wide;
        }
        for (int l = 0; l < 4; ++l) a.acc_init[l] = cached_init[l];
        a.lean_combine = cached_lean;
    }
#ifdef GSW_DIAGNOSTIC_BUILDS
    // timing-only ablations of the shape's padding (wrong or partial results): what the tail column tile, the
    // thin last row tile and the last, mostly empty K stage cost
    if (const char *e = getenv("GSW_ABL_SHAPE")) {
        const int m = atoi(e);
        if ((m & 1) && a.out_cols % kTileCols) a.out_cols -= a.out_cols % kTileCols, a.packed_tail = 0;
        if ((m & 2) && a.rows % kTileRows) a.rows -= a.rows % kTileRows;
        if ((m & 4) && a.e4 > kStageElems) a.e4 -= kStageElems;  // the prepared operand keeps its stride: wrong data, same work per stage
    }
#endif
#ifdef GSW_DIAGNOSTIC_BUILDS
    if (variant >= 10 && variant < 138 && ring.kind == kRingFp) {
        // diagnostic builds: variant = 10 + ABL
        const int nt = col_tiles(a.out_cols), mt = (a.rows + 255) / 256;
        const long blocks = (((long)a.batch * nt + 7) / 8) * 8 * mt;
        dim3 grid((unsigned)blocks);
#define GSW_ABL(N)                                                                                             \
    case 10 + N:                                                                                               \
        hipLaunchKernelGGL(binlimb_gemm16_ablate<N>, grid, dim3(256), pipe_lds_bytes<4>(), stream, a, ring.ell, \
                           ring.mask, nt, mt);                                                                 \
        break;
        switch (variant) {
            GSW_ABL(1) GSW_ABL(2) GSW_ABL(3) GSW_ABL(4) GSW_ABL(8) GSW_ABL(12) GSW_ABL(15) GSW_ABL(16) GSW_ABL(33) GSW_ABL(65)
            default: return hipErrorInvalidValue;
        }
#undef GSW_ABL
        return hipGetLastError();
    }
#endif
#ifdef GSW_EXP_TALL
    if (variant == 10 || variant == 11) {
        const int nt = col_tiles(a.out_cols), mt = (a.rows + 511) / 512;
        const long blocks = (((long)a.batch + 7) / 8) * 8 * nt * mt;
        if (blocks > 0x7FFFFFFFL) return hipErrorInvalidValue;
        dim3 grid((unsigned)blocks);
        if (variant == 11) {
            if (ring.kind == kRingFp)
                hipLaunchKernelGGL((binlimb_gemm16_tall_skew<kRingFp, 8>), grid, dim3(256), tall_lds_bytes<8>(), stream, a,
                                   ring.ell, ring.mask, nt, mt);
            else
                hipLaunchKernelGGL((binlimb_gemm16_tall_skew<kRingZpow2, 8>), grid, dim3(256), tall_lds_bytes<8>(), stream,
                                   a, ring.ell, ring.mask, nt, mt);
            return hipGetLastError();
        }
        if (ring.kind == kRingFp)
            hipLaunchKernelGGL((binlimb_gemm16_tall<kRingFp, 8>), grid, dim3(256), tall_lds_bytes<8>(), stream, a, ring.ell,
                               ring.mask, nt, mt);
        else
            hipLaunchKernelGGL((binlimb_gemm16_tall<kRingZpow2, 8>), grid, dim3(256), tall_lds_bytes<8>(), stream, a,
                               ring.ell, ring.mask, nt, mt);
        return hipGetLastError();
    }
#endif
#ifdef GSW_EXP_WIDE8
    if (variant == 12) {
        if (lean_used) *lean_used = a.lean_combine;
        const int nt = col_tiles(a.out_cols), mt = (a.rows + 511) / 512;
        const long blocks = (((long)a.batch + 7) / 8) * 8 * nt * mt;
        if (blocks > 0x7FFFFFFFL) return hipErrorInvalidValue;
        dim3 grid((unsigned)blocks);
        constexpr int kLds = 2 * 8 * 4096 + (GSW_EXP_WIDE8 > 3 ? GSW_EXP_WIDE8 : 3) * kStageElems * 4 * kFragBytes;
        const bool fp = ring.kind == kRingFp;
#define GSW_LAUNCH8(R, S) hipLaunchKernelGGL((binlimb_gemm16_skew8<R, S>), grid, dim3(512), kLds, stream, a, ring.ell, ring.mask, nt, mt)
        if (!fp) GSW_LAUNCH8(kRingZpow2, 0);
        else if (a.scale_mode == 2) GSW_LAUNCH8(kRingFp, 2);
        else if (a.scale_mode == 1) GSW_LAUNCH8(kRingFp, 1);
        else GSW_LAUNCH8(kRingFp, 0);
#undef GSW_LAUNCH8
        return hipGetLastError();
    }
#endif
    if (variant < 0 || variant > 9) return hipErrorInvalidValue;
    if (lean_used) *lean_used = a.lean_combine;  // what the kernel launched below is told
#ifndef GSW_DIAGNOSTIC_BUILDS
    if (variant != 2 && variant != 3 && variant != 9) return hipErrorInvalidValue;  // not in the shipped library
#endif
    if (a.packed_tail && (variant == 2 || variant == 3)) return hipErrorInvalidValue;  // no packed tile there
    const int tile_rows = (variant == 1 || variant == 5 || variant == 7 || variant == 8) ? 512 : 256;
    const int nt = col_tiles(a.out_cols);
    const int mt = (a.rows + tile_rows - 1) / tile_rows;
    const long units = (long)a.batch * nt;
    const long groups = (units + 7) / 8;
    // variants 4..9 place whole batch entries per XCD (map_workgroup_blocked), the others column tiles
    const long blocks = variant >= 4 ? (((long)a.batch + 7) / 8) * 8 * nt * mt : groups * 8 * mt;
    if (blocks > 0x7FFFFFFFL) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks);
    const bool fp = ring.kind == kRingFp;
#define GSW_LAUNCH(KERNEL, THREADS, LDS) \
    hipLaunchKernelGGL(KERNEL, grid, dim3(THREADS), LDS, stream, a, ring.ell, ring.mask, nt, mt)
    switch (variant) {
        case 2:
            if (fp) GSW_LAUNCH((binlimb_gemm_kernel<kRingFp, true>), 256, kGemmLds);
            else GSW_LAUNCH((binlimb_gemm_kernel<kRingZpow2, true>), 256, kGemmLds);
            break;
        case 3:
            if (fp) GSW_LAUNCH((binlimb_gemm_kernel<kRingFp, false>), 256, kGemmLds);
            else GSW_LAUNCH((binlimb_gemm_kernel<kRingZpow2, false>), 256, kGemmLds);
            break;
#ifdef GSW_DIAGNOSTIC_BUILDS
        case 0:
            if (fp) GSW_LAUNCH((binlimb_gemm_pipelined<kRingFp, 4>), 256, pipe_lds_bytes<4>());
            else GSW_LAUNCH((binlimb_gemm_pipelined<kRingZpow2, 4>), 256, pipe_lds_bytes<4>());
            break;
        case 1:
            if (fp) GSW_LAUNCH((binlimb_gemm_pipelined<kRingFp, 8>), 512, pipe_lds_bytes<8>());
            else GSW_LAUNCH((binlimb_gemm_pipelined<kRingZpow2, 8>), 512, pipe_lds_bytes<8>());
            break;
        case 4:
            if (fp) GSW_LAUNCH((binlimb_gemm16_pipelined<kRingFp, 4>), 256, pipe_lds_bytes<4>());
            else GSW_LAUNCH((binlimb_gemm16_pipelined<kRingZpow2, 4>), 256, pipe_lds_bytes<4>());
            break;
        case 5:
            if (fp) GSW_LAUNCH((binlimb_gemm16_pipelined<kRingFp, 8>), 512, pipe_lds_bytes<8>());
            else GSW_LAUNCH((binlimb_gemm16_pipelined<kRingZpow2, 8>), 512, pipe_lds_bytes<8>());
            break;
        case 6:
            if (fp) GSW_LAUNCH((binlimb_gemm16_rowblock<kRingFp, 4>), 256, rowblock_lds_bytes<4>());
            else GSW_LAUNCH((binlimb_gemm16_rowblock<kRingZpow2, 4>), 256, rowblock_lds_bytes<4>());
            break;
        case 7:
            if (fp) GSW_LAUNCH((binlimb_gemm16_rowblock<kRingFp, 8>), 512, rowblock_lds_bytes<8>());
            else GSW_LAUNCH((binlimb_gemm16_rowblock<kRingZpow2, 8>), 512, rowblock_lds_bytes<8>());
            break;
        case 8:
            if (fp) GSW_LAUNCH((binlimb_gemm16_stagger<kRingFp>), 512, stagger_lds_bytes());
            else GSW_LAUNCH((binlimb_gemm16_stagger<kRingZpow2>), 512, stagger_lds_bytes());
            break;
#endif
        default: {
#ifdef GSW_EXP_PERSIST
            if (const char *e = getenv("GSW_PERSIST"); e && (e[0] == '1' || e[0] == '2' || e[0] == '3')) {
                void *cnt = nullptr;
                hipError_t pe = hipGetSymbolAddress(&cnt, HIP_SYMBOL(g_persist_count));
                if (pe == hipSuccess) pe = hipMemsetD32Async((hipDeviceptr_t)cnt, e[0] == '3' ? 64 : 0, 8, stream);
                if (pe != hipSuccess) return pe;
                grid = dim3(512);
                constexpr int kLds = rowblock_lds_bytes<4>();
                if (e[0] == '3') {
                    if (!fp) GSW_LAUNCH((binlimb_gemm16_skew_streaming<kRingZpow2, 0>), 256, kLds);
                    else if (a.scale_mode == 2) GSW_LAUNCH((binlimb_gemm16_skew_streaming<kRingFp, 2>), 256, kLds);
                    else if (a.scale_mode == 1) GSW_LAUNCH((binlimb_gemm16_skew_streaming<kRingFp, 1>), 256, kLds);
                    else return hipErrorInvalidValue;
                    break;
                }
                if (e[0] == '2') {
                    if (!fp) GSW_LAUNCH((binlimb_gemm16_skew_persistent<kRingZpow2, 0, true>), 256, kLds);
                    else if (a.scale_mode == 2) GSW_LAUNCH((binlimb_gemm16_skew_persistent<kRingFp, 2, true>), 256, kLds);
                    else if (a.scale_mode == 1) GSW_LAUNCH((binlimb_gemm16_skew_persistent<kRingFp, 1, true>), 256, kLds);
                    else return hipErrorInvalidValue;
                    break;
                }
                if (!fp) GSW_LAUNCH((binlimb_gemm16_skew_persistent<kRingZpow2, 0>), 256, kLds);
                else if (a.scale_mode == 2) GSW_LAUNCH((binlimb_gemm16_skew_persistent<kRingFp, 2>), 256, kLds);
                else if (a.scale_mode == 1) GSW_LAUNCH((binlimb_gemm16_skew_persistent<kRingFp, 1>), 256, kLds);
                else return hipErrorInvalidValue;
                break;
            }
#endif
#ifdef GSW_EXP_ONEWG
            constexpr int kSkewLds = rowblock_lds_bytes<4>() + 24 * 1024;  // experiment: one workgroup per CU
#else
            constexpr int kSkewLds = rowblock_lds_bytes<4>();
#endif
            if (!fp) GSW_LAUNCH((binlimb_gemm16_skew<kRingZpow2, 0>), 256, kSkewLds);
            else if (a.scale_mode == 2) GSW_LAUNCH((binlimb_gemm16_skew<kRingFp, 2>), 256, kSkewLds);
            else if (a.scale_mode == 1) GSW_LAUNCH((binlimb_gemm16_skew<kRingFp, 1>), 256, kSkewLds);
            else GSW_LAUNCH((binlimb_gemm16_skew<kRingFp, 0>), 256, kSkewLds);
            break;
        }
    }
#undef GSW_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------ gadget helpers

// bit_decomp_matrix (helper.rs:7-21): bit k of column j goes to column j*ell + k, little endian.
//
// G^-1 as a streaming kernel, in the format its consumer reads:
//   kDecompU32  one uint32 (ring 0 / 1) per bit -- the reference's matrix (N x N: 254 MB at n = 256)
//   kDecompU8   one byte per bit -- rows of an int8 MFMA operand
//   kDecompBits the row's bits as one little-endian bit string, 32 per word -- the densest form
// U32 / U8 (bit_slice_kernel): a workgroup takes 256 consecutive elements (row-major order) with one coalesced load,
// every lane slices its element into ell outputs in LDS (lane t writes slots t*ell + k: an odd ell makes the
// 64 lanes of a store hit distinct banks, an even one is skewed by the lane number), and the workgroup then
// streams the staged segment to memory as 16-byte words, 1 KiB contiguous per wave instruction, with the few
// outputs before the first and after the last 16-byte boundary stored singly.
// Bits (bit_decomp_kernel): the input words of a run are staged instead and every lane gathers 128 output bits.
// All index arithmetic is 32-bit; rows and segments are spread over the grid.
constexpr int kSliceLanes = 256;

template <typename OUT>
__global__ __launch_bounds__(kSliceLanes) void bit_slice_kernel(const uint32_t *in, unsigned elems, unsigned cols,
                                                                unsigned ld_in, OUT *out, int ell_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char slice_lds[];
    OUT *stage = reinterpret_cast<OUT *>(slice_lds);
    constexpr unsigned kVec = 16 / sizeof(OUT);  // outputs per 16-byte store
    const unsigned ell = (unsigned)ell_;
    const unsigned tid = threadIdx.x;
    // Output rows are tight, so element e = row * cols + col of the matrix owns outputs [e * ell, (e + 1) * ell) of the
    // flat output whatever its row: segments are runs of 256 consecutive elements and may cross rows.
    const unsigned segs = (elems + kSliceLanes - 1) / kSliceLanes;
    // an even ell would put every lane's slot k in the same bank: lane t starts its bits at k = t mod ell
    const unsigned k_first = (ell & 1u) == 0 ? tid % ell : 0u;
    for (unsigned seg = blockIdx.x; seg < segs; seg += gridDim.x) {
        const unsigned e0 = seg * kSliceLanes;
        const unsigned count = min((unsigned)kSliceLanes, elems - e0);  // elements of this segment
        OUT *dst = out + (size_t)e0 * ell;
        // The stage mirrors the destination from its last 16-byte boundary on: output i of the segment sits in
        // slot mis + i, so 16-byte words of the stage are 16-byte words of memory.
        const unsigned mis = (unsigned)((reinterpret_cast<uintptr_t>(dst) / sizeof(OUT)) % kVec);
        __syncthreads();  // the previous segment has left the stage
        if (tid < count) {
            const unsigned e = e0 + tid, row = e / cols, col = e - row * cols;
            const uint32_t x = in[(size_t)row * ld_in + col];
            OUT *mine = stage + mis + tid * ell;
            unsigned k = k_first;
            for (unsigned i = 0; i < ell; ++i) {
                mine[k] = (OUT)((x >> k) & 1u);
                if (++k == ell) k = 0;
            }
        }
        __syncthreads();
        const unsigned end = mis + count * ell;                      // slots [mis, end) hold outputs
        const unsigned v_first = mis ? 1u : 0u, v_end = end / kVec;  // whole 16-byte words [v_first, v_end)
        OUT *base = dst - mis;                                        // 16-byte aligned
        for (unsigned v = v_first + tid; v < v_end; v += kSliceLanes)
            *reinterpret_cast<uint4 *>(base + v * kVec) = *reinterpret_cast<const uint4 *>(stage + v * kVec);
        // the outputs before the first and after the last boundary, one by one
        if (mis && mis + tid < min(kVec, end)) base[mis + tid] = stage[mis + tid];
        const unsigned tail0 = max(v_end, v_first) * kVec;
        if (tail0 + tid < end) base[tail0 + tid] = stage[tail0 + tid];
    }
}

constexpr int kDecompLanes = 256;
constexpr int kDecompStage = 4096;  // input words staged per run (+ 2 for the elements a run straddles)

template <int FMT>
__global__ __launch_bounds__(kDecompLanes) void bit_decomp_kernel(const uint32_t *in, int rows, int cols, int ld_in,
                                                                  void *out_v, int ell_, unsigned out_pitch,
                                                                  unsigned run_bits, unsigned runs_per_row) {
    constexpr unsigned PER = FMT == kDecompU32 ? 4u : FMT == kDecompU8 ? 16u : 128u;  // output bits per lane
    __shared__ uint32_t stage[kDecompStage + 2];
    const unsigned ell = (unsigned)ell_;
    const unsigned wide = (unsigned)cols * ell;  // bits per row
    const unsigned total_runs = (unsigned)rows * runs_per_row;
    for (unsigned run = blockIdx.x; run < total_runs; run += gridDim.x) {
        const unsigned row = run / runs_per_row, o_base = (run - row * runs_per_row) * run_bits;
        const unsigned j_base = o_base / ell;
        const unsigned j_end = (min(o_base + run_bits, wide) + ell - 1u) / ell;  // one past the last element touched
        const uint32_t *src = in + (size_t)row * ld_in;
        __syncthreads();  // the previous run's readers are done with the stage
        for (unsigned t = threadIdx.x; j_base + t < j_end; t += kDecompLanes) stage[t] = src[j_base + t];
        __syncthreads();
        const unsigned o0 = o_base + threadIdx.x * PER;
        if (threadIdx.x * PER >= run_bits || o0 >= wide) continue;  // both barriers of this run are behind us
        unsigned j = o0 / ell, k = o0 - j * ell;
        if constexpr (FMT == kDecompBits) {
            // 128 consecutive bits of the row's bit string: four words, each gathered from the elements it spans
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t acc = 0;
                unsigned filled = 0;
                while (filled < 32u && j < (unsigned)cols) {
                    const uint32_t piece = stage[j - j_base] >> k;  // the ell - k bits of element j not yet used
                    acc |= piece << filled;
                    const unsigned take = min(ell - k, 32u - filled);
                    filled += take;
                    k += take;
                    if (k == ell) { k = 0; ++j; }
                }
                // operands are canonical (< 2^ell), so `piece` never carries bits beyond the element
                w[q] = filled < 32u ? acc & ((1u << filled) - 1u) : acc;
            }
            uint32_t *dst = (uint32_t *)out_v + (size_t)row * out_pitch + (o0 >> 5);
            const unsigned words_left = out_pitch - (o0 >> 5);
            if (words_left >= 4u && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0)
                *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
            else
                for (unsigned q = 0; q < 4u && q < words_left; ++q) dst[q] = w[q];
        } else {
            uint32_t cur = stage[j - j_base] >> k;
            uint32_t bits[PER];
#pragma unroll
            for (unsigned t = 0; t < PER; ++t) {
                bits[t] = cur & 1u;
                cur >>= 1;
                if (++k == ell) {
                    k = 0;
                    ++j;
                    cur = j < j_end ? stage[j - j_base] : 0u;
                }
            }
            if constexpr (FMT == kDecompU32) {
                uint32_t *dst = (uint32_t *)out_v + (size_t)row * out_pitch + o0;
                if (o0 + 4u <= wide && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0)
                    *reinterpret_cast<uint4 *>(dst) = make_uint4(bits[0], bits[1], bits[2], bits[3]);
                else
                    for (unsigned t = 0; t < 4u && o0 + t < wide; ++t) dst[t] = bits[t];
            } else {
                uint8_t *dst = (uint8_t *)out_v + (size_t)row * out_pitch + o0;
                if (o0 + 16u <= wide && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
                    uint32_t p4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        p4[q] = bits[4 * q] | bits[4 * q + 1] << 8 | bits[4 * q + 2] << 16 | bits[4 * q + 3] << 24;
                    *reinterpret_cast<uint4 *>(dst) = make_uint4(p4[0], p4[1], p4[2], p4[3]);
                } else {
                    for (unsigned t = 0; t < 16u && o0 + t < wide; ++t) dst[t] = (uint8_t)bits[t];
                }
            }
        }
    }
}

// bit_decomp_inv_matrix (helper.rs:24-49)
template <int RING>
__global__ void bit_decomp_inv_kernel(const uint32_t *in, int rows, int cols, uint32_t *out, int ld_out, int quirk,
                                      int ell, uint32_t mask) {
    // every column up to the leading dimension is written: the padding columns of a compact ciphertext are zero
    const size_t total = (size_t)rows * ld_out;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t i = idx / ld_out, j = idx - i * ld_out;
        if (j >= (size_t)cols) {
            out[idx] = 0u;
            continue;
        }
        const uint32_t *chunk = in + (i * cols + j) * ell;
        uint32_t res;
        if (quirk) {
            uint64_t repr = 0;
            for (int k = 0; k < ell; ++k)
                if (chunk[k] == 1u) repr |= 1ull << k;  // helper.rs:32-34
            res = ring_from_u64<RING>(repr, mask);      // helper.rs:37
        } else {
            res = 0;
            for (int k = 0; k < ell; ++k)
                res = ring_add<RING>(res, ring_mul<RING>(chunk[k], ring_from_u64<RING>(1ull << k, mask), mask), mask);
        }
        out[i * ld_out + j] = res;
    }
}

template <int RING>
__global__ void add_kernel(const uint32_t *a, const uint32_t *b, uint32_t *out, size_t count, uint32_t mask) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < count;
         idx += (size_t)gridDim.x * blockDim.x)
        out[idx] = ring_add<RING>(a[idx], b[idx], mask);
}

template <int RING>
__global__ void mult_const_kernel(uint32_t *x, uint32_t k, size_t count, uint32_t mask) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < count;
         idx += (size_t)gridDim.x * blockDim.x)
        x[idx] = ring_mul<RING>(x[idx], k, mask);
}

static inline unsigned grid_for(size_t total, int block) {
    size_t g = (total + block - 1) / block;
    if (g > 256u * 16u) g = 256u * 16u;
    if (g == 0) g = 1;
    return (unsigned)g;
}

size_t bit_decomp_pitch(int cols, int ell, int format) {
    const size_t wide = (size_t)cols * ell;
    return format == kDecompBits ? (wide + 31) / 32 : wide;  // in output elements (words / bytes / words)
}

hipError_t launch_bit_decomp(const RingDesc &ring, const uint32_t *in, int rows, int cols, int ld_in, void *out,
                             int format, hipStream_t stream) {
    const size_t wide = (size_t)cols * ring.ell;
    if (rows <= 0 || wide == 0) return hipSuccess;
    if (wide > 0x7FFFFFFFu || format < kDecompU32 || format > kDecompBits) return hipErrorInvalidValue;
    const unsigned per = format == kDecompU32 ? 4u : format == kDecompU8 ? 16u : 128u;
    // a run is what one workgroup iteration covers: every lane's share, but no more input than the stage holds
    unsigned run = kDecompLanes * per;
    if ((size_t)kDecompStage * ring.ell < run) run = (unsigned)(kDecompStage * ring.ell) / per * per;
    const unsigned runs_per_row = (unsigned)((wide + run - 1) / run);
    const size_t total_runs = (size_t)rows * runs_per_row;
    if (total_runs > 0xFFFFFFFFu) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(total_runs < 256u * 32u ? total_runs : 256u * 32u);
    const unsigned pitch = (unsigned)bit_decomp_pitch(cols, ring.ell, format);
    if (format != kDecompBits) {
        const size_t elems = (size_t)rows * cols;
        if (elems > 0xFFFFFFFFu - kSliceLanes) return hipErrorInvalidValue;
        const size_t segs = (elems + kSliceLanes - 1) / kSliceLanes;
        const unsigned sgrid = (unsigned)(segs < 256u * 16u ? segs : 256u * 16u);
        const size_t elem = format == kDecompU32 ? 4 : 1;
        const size_t lds = (((size_t)kSliceLanes * ring.ell + 16) * elem + 15) & ~(size_t)15;  // + one word of skew
        if (format == kDecompU32)
            hipLaunchKernelGGL(bit_slice_kernel<uint32_t>, dim3(sgrid), dim3(kSliceLanes), lds, stream, in,
                               (unsigned)elems, (unsigned)cols, (unsigned)ld_in, (uint32_t *)out, ring.ell);
        else
            hipLaunchKernelGGL(bit_slice_kernel<uint8_t>, dim3(sgrid), dim3(kSliceLanes), lds, stream, in,
                               (unsigned)elems, (unsigned)cols, (unsigned)ld_in, (uint8_t *)out, ring.ell);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(bit_decomp_kernel<kDecompBits>, dim3(grid), dim3(kDecompLanes), 0, stream, in, rows, cols, ld_in,
                       out, ring.ell, pitch, run, runs_per_row);
    return hipGetLastError();
}

hipError_t launch_bit_decomp_inv(const RingDesc &ring, const uint32_t *in, int rows, int cols, uint32_t *out,
                                 int ld_out, int quirk, hipStream_t stream) {
    const size_t total = (size_t)rows * cols;
    if (total == 0) return hipSuccess;
    dim3 grid(grid_for(total, 256)), block(256);
    if (ring.kind == kRingFp)
        hipLaunchKernelGGL(bit_decomp_inv_kernel<kRingFp>, grid, block, 0, stream, in, rows, cols, out, ld_out, quirk,
                           ring.ell, ring.mask);
    else
        hipLaunchKernelGGL(bit_decomp_inv_kernel<kRingZpow2>, grid, block, 0, stream, in, rows, cols, out, ld_out,
                           quirk, ring.ell, ring.mask);
    return hipGetLastError();
}

hipError_t launch_add(const RingDesc &ring, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t count,
                      hipStream_t stream) {
    if (count == 0) return hipSuccess;
    dim3 grid(grid_for(count, 256)), block(256);
    if (ring.kind == kRingFp)
        hipLaunchKernelGGL(add_kernel<kRingFp>, grid, block, 0, stream, a, b, out, count, ring.mask);
    else
        hipLaunchKernelGGL(add_kernel<kRingZpow2>, grid, block, 0, stream, a, b, out, count, ring.mask);
    return hipGetLastError();
}

hipError_t launch_mult_const(const RingDesc &ring, uint32_t *x, uint32_t k, size_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    dim3 grid(grid_for(count, 256)), block(256);
    if (ring.kind == kRingFp)
        hipLaunchKernelGGL(mult_const_kernel<kRingFp>, grid, block, 0, stream, x, k, count, ring.mask);
    else
        hipLaunchKernelGGL(mult_const_kernel<kRingZpow2>, grid, block, 0, stream, x, k, count, ring.mask);
    return hipGetLastError();
}

// ------------------------------------------------------------------ dense product (parity path)
//
// mult (gsw_impl.rs:104-111) / nand (gsw_impl.rs:114-123) for arbitrary entries: the plain
// dim^3 product over the ring, 16x16 LDS tiles, 64-bit accumulation.

template <int RING>
__global__ __launch_bounds__(256) void mul_dense_kernel(const uint32_t *a, const uint32_t *b, uint32_t *out, int dim,
                                                         int nand, uint32_t mask) {
    __shared__ uint32_t ta[16][17];
    __shared__ uint32_t tb[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int row = blockIdx.y * 16 + ty, col = blockIdx.x * 16 + tx;
    uint64_t acc = 0;
    for (int k0 = 0; k0 < dim; k0 += 16) {
        ta[ty][tx] = (row < dim && k0 + tx < dim) ? a[(size_t)row * dim + k0 + tx] : 0u;
        tb[ty][tx] = (k0 + ty < dim && col < dim) ? b[(size_t)(k0 + ty) * dim + col] : 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint64_t prod = (uint64_t)ta[ty][k] * tb[k][tx];
            if (RING == kRingFp) acc += (prod & kFpP) + (prod >> 31);  // < 2^32 per term
            else acc += prod;                                          // wraps mod 2^64, 2^M | 2^64
        }
        __syncthreads();
    }
    if (row < dim && col < dim) {
        uint32_t val = ring_from_u64<RING>(acc, mask);
        if (nand) {
            val = ring_sub<RING>(0u, val, mask);
            if (row == col) val = ring_add<RING>(val, ring_from_u64<RING>(1ull, mask), mask);
        }
        out[(size_t)row * dim + col] = val;
    }
}

// ---- dense product with a flattened left operand: the reference's own ciphertext format on the MFMA path.
//
// A default-feature ciphertext is BitDecomp(R * A) + mu * I (gsw_impl.rs:27-42): entries in {0, 1} except for
// the diagonal, which carries the message on top.  For such a left operand C1 = B + diag(d), B binary,
//     C1 * C2 = B * C2 + diag(d) * C2,
// and B * C2 is the binary x byte-limb contraction with an N-column right operand.  pack_flat_lhs_kernel
// packs B into ell-bit words (the layout of a compact ciphertext), extracts d, and counts the entries that do
// not fit the form (anything but 0 / 1 off the diagonal); dense_fixup_kernel adds the diagonal term and applies
// nand's I - P (gsw_impl.rs:114-123).

// One workgroup per row: the row's entries are read with coalesced loads (lane = entry), their low bits gathered
// into a bit string in LDS by wave ballots, and the ell-bit words cut out of that string.
__global__ __launch_bounds__(256) void pack_flat_lhs_kernel(const uint32_t *c1, int dim, int ell, int words, int ld,
                                                             uint32_t *packed, uint32_t *diag,
                                                             unsigned long long *misfits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pack_lds[];
    unsigned long long *bits = reinterpret_cast<unsigned long long *>(pack_lds);  // ceil(dim / 64) + 1 words
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned bad = 0, scaled = 0;  // misfits; rows whose diagonal entry is neither 0 nor 1
    for (int i = blockIdx.x; i < dim; i += gridDim.x) {
        const uint32_t *src = c1 + (size_t)i * dim;
        const int chunks = (dim + 63) >> 6;
        __syncthreads();  // the previous row's words have been cut
        for (int e0 = (tid >> 6) << 6; e0 < dim; e0 += 256) {  // each wave takes 64 consecutive entries at a time
            const int e = e0 + lane;
            uint32_t bit = 0;
            if (e < dim) {
                const uint32_t v = src[e];
                if (e == i) {
                    // the diagonal entry: 0 / 1 stay in B, anything else goes to d
                    if (v <= 1u) { bit = v; diag[i] = 0u; }
                    else { diag[i] = v; ++scaled; }
                } else {
                    bad += v > 1u;
                    bit = v & 1u;
                }
            }
            const unsigned long long mask = __ballot(bit != 0);
            if (lane == 0) bits[e0 >> 6] = mask;
        }
        if (tid == 0) bits[chunks] = 0ull;  // the word after the end: reads of the last field may touch it
        __syncthreads();
        for (int j = tid; j < ld; j += 256) {
            uint32_t w = 0;
            if (j < words) {
                const unsigned pos = (unsigned)j * (unsigned)ell;  // ell bits from bit `pos` of the string
                const unsigned q = pos >> 6, r = pos & 63u;
                unsigned long long field = bits[q] >> r;
                if (r + (unsigned)ell > 64u) field |= bits[q + 1] << (64u - r);
                w = (uint32_t)field & (ell >= 32 ? 0xFFFFFFFFu : ((1u << ell) - 1u));
            }
            packed[(size_t)i * ld + j] = w;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bad += (unsigned)__shfl_xor((int)bad, off, 64);
        scaled += (unsigned)__shfl_xor((int)scaled, off, 64);
    }
    if (lane == 0) {
        if (bad) atomicAdd(&misfits[0], (unsigned long long)bad);
        if (scaled) atomicAdd(&misfits[1], (unsigned long long)scaled);
    }
}

template <int RING>
__global__ __launch_bounds__(256) void dense_fixup_kernel(uint32_t *out, const uint32_t *c2, const uint32_t *diag,
                                                           int dim, int nand, uint32_t mask) {
    const size_t total = (size_t)dim * dim;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int i = (int)(idx / (size_t)dim), j = (int)(idx - (size_t)i * dim);
        uint32_t v = out[idx];
        const uint32_t d = diag[i];
        if (d) v = ring_add<RING>(v, ring_mul<RING>(d, c2[idx], mask), mask);
        if (nand) {
            v = ring_sub<RING>(0u, v, mask);
            if (i == j) v = ring_add<RING>(v, ring_from_u64<RING>(1ull, mask), mask);
        }
        out[idx] = v;
    }
}

hipError_t launch_pack_flat_lhs(const RingDesc &ring, const uint32_t *c1, int dim, int words, int ld, uint32_t *packed,
                                uint32_t *diag, unsigned long long *misfits, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(misfits, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess || dim <= 0) return e;
    const size_t lds = (((size_t)dim + 63) / 64 + 1) * 8;
    hipLaunchKernelGGL(pack_flat_lhs_kernel, dim3((unsigned)(dim < 256 * 8 ? dim : 256 * 8)), dim3(256), lds, stream, c1,
                       dim, ring.ell, words, ld, packed, diag, misfits);
    return hipGetLastError();
}

hipError_t launch_dense_fixup(const RingDesc &ring, uint32_t *out, const uint32_t *c2, const uint32_t *diag, int dim,
                              int nand, hipStream_t stream) {
    if (dim <= 0) return hipSuccess;
    dim3 grid(grid_for((size_t)dim * dim, 256)), block(256);
    if (ring.kind == kRingFp)
        hipLaunchKernelGGL(dense_fixup_kernel<kRingFp>, grid, block, 0, stream, out, c2, diag, dim, nand, ring.mask);
    else
        hipLaunchKernelGGL(dense_fixup_kernel<kRingZpow2>, grid, block, 0, stream, out, c2, diag, dim, nand, ring.mask);
    return hipGetLastError();
}

hipError_t launch_mul_dense(const RingDesc &ring, const uint32_t *a, const uint32_t *b, uint32_t *out, int dim,
                            int nand, hipStream_t stream) {
    if (dim <= 0) return hipSuccess;
    dim3 grid((unsigned)((dim + 15) / 16), (unsigned)((dim + 15) / 16)), block(256);
    if (ring.kind == kRingFp)
        hipLaunchKernelGGL(mul_dense_kernel<kRingFp>, grid, block, 0, stream, a, b, out, dim, nand, ring.mask);
    else
        hipLaunchKernelGGL(mul_dense_kernel<kRingZpow2>, grid, block, 0, stream, a, b, out, dim, nand, ring.mask);
    return hipGetLastError();
}

// ---- dense product with arbitrary entries on MFMA: 4 x 4 signed-digit products.
//
// mult (gsw_impl.rs:104-111) on matrices that are not in the flattened form -- the reference's default build never
// flattens (Cargo.toml:16-17), so mult(mult(a, b), c) multiplies by a left operand with arbitrary entries.  Both
// operands are split into four signed base-256 digits (signed_digits): C1 = sum_a 256^a L_a, C2 = sum_b 256^b R_b, and
//     C1 C2 = sum_d 256^d T_d,   T_d = sum_{a + b = d} L_a R_b,
// sixteen int8 products with exact int32 accumulation (|T_d| <= 4 * 2^14 * dim < 2^31 for dim < 32768), seven
// accumulators per output element (four for Z/2^M: 256^d vanishes mod 2^32 from d = 4 on, ten products).
// dense_split_kernel writes the digit planes -- [digit][row][k] bytes for the left operand, [digit][column][k] for
// the right one (transposed through LDS), k padded to 64 and rows / columns to 64 with zeros -- so that every
// 16 x 64 MFMA fragment is 16 rows of 64 contiguous bytes.  dense_limb_gemm_kernel: workgroup tile 64 x 64, four
// waves of 32 x 32, two LDS buffers of 32 fragments (LDS-DMA, fragment order, conflict-free 16-byte reads), one
// barrier per 64 K slots; the XCD-aware order gives each XCD 8 row tiles x 8 column tiles at a time.

constexpr int kDenseTile = 64;

template <int RING, bool TRANSPOSE>
__global__ __launch_bounds__(256) void dense_split_kernel(const uint32_t *src, int dim, int pad, uint8_t *planes) {
    // planes: [4][pad][pad] bytes; element (i, k) of plane d = digit d of src[i][k] (TRANSPOSE: of src[k][i])
    __shared__ uint32_t tile[kDenseTile][kDenseTile + 1];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.y * kDenseTile, k0 = blockIdx.x * kDenseTile;
    // load 64 x 64 words with coalesced reads along the source's rows, digits already split
    for (int e = tid; e < kDenseTile * kDenseTile; e += 256) {
        const int y = e >> 6, x = e & 63;
        // source row / column of this word
        const int r = TRANSPOSE ? k0 + y : i0 + y, c = TRANSPOSE ? i0 + x : k0 + x;
        const uint32_t v = (r < dim && c < dim) ? signed_digits<RING>(src[(size_t)r * dim + c]) : 0u;
        if (TRANSPOSE) tile[x][y] = v;  // tile[i][k]
        else tile[y][x] = v;
    }
    __syncthreads();
    // every thread packs four consecutive k of one i into one dword per digit plane
    const size_t plane = (size_t)pad * pad;
    for (int e = tid; e < kDenseTile * (kDenseTile / 4); e += 256) {
        const int i = e >> 4, kq = (e & 15) * 4;
        const uint32_t w0 = tile[i][kq], w1 = tile[i][kq + 1], w2 = tile[i][kq + 2], w3 = tile[i][kq + 3];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t packed = ((w0 >> (8 * d)) & 0xFFu) | (((w1 >> (8 * d)) & 0xFFu) << 8) |
                                    (((w2 >> (8 * d)) & 0xFFu) << 16) | (((w3 >> (8 * d)) & 0xFFu) << 24);
            *reinterpret_cast<uint32_t *>(planes + d * plane + (size_t)(i0 + i) * pad + k0 + kq) = packed;
        }
    }
}

template <int RING>
__global__ __launch_bounds__(256, 2) void dense_limb_gemm_kernel(const uint8_t *lhs_planes, const uint8_t *rhs_planes, uint32_t *out,
                                                                int dim, int pad, int row0, int rows, int nand, uint32_t mask,
                                                                int col_tiles_n, int row_tiles_n) {
    // lhs_planes holds rows row0 .. row0 + rows_pad - 1 of the left operand (a slab); out rows are absolute
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kND = RING == kRingFp ? 7 : 4;  // accumulators (powers of 256) per output element
    constexpr int kStage = 32 * kFragBytes;      // 16 left + 16 right fragments
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // placement: an XCD (workgroups id mod 8) works on 8 row tiles x 8 column tiles at a time, row tile fastest
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int ct_groups = (col_tiles_n + 7) / 8;
    const int per_group = 8 * 8;
    const int grp = local / per_group, in = local % per_group;
    const int rgrp = grp / ct_groups, cgrp = grp % ct_groups;
    const int rt = (rgrp * 8 + (in & 7)) * 8 + xcd, ctile = cgrp * 8 + (in >> 3);
    if (rt >= row_tiles_n || ctile >= col_tiles_n) return;
    const int r_base = rt * kDenseTile, c_base = ctile * kDenseTile;  // r_base relative to the slab
    const size_t lhs_plane = (size_t)(row_tiles_n * kDenseTile) * pad, rhs_plane = (size_t)pad * pad;
    const int steps = pad / 64;
    const int fr = lane & 15, fg = lane >> 4;  // fragment lane: row / column fr, K group fg

    // this wave stages fragments 8 wave .. 8 wave + 7 of a stage: 0..15 left [row block][digit], 16..31 right [column block][digit]
    const uint8_t *src[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int f = wave * 8 + j, blk = (f & 15) >> 2, d = f & 3;
        src[j] = f < 16 ? lhs_planes + d * lhs_plane + (size_t)(r_base + 16 * blk + fr) * pad + 16 * fg
                        : rhs_planes + d * rhs_plane + (size_t)(c_base + 16 * blk + fr) * pad + 16 * fg;
    }
    auto stage = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds(src[j] + (size_t)t * 64,
                                             (__attribute__((address_space(3))) void *)(smem + buf * kStage + (wave * 8 + j) * kFragBytes),
                                             16, 0, 0);
    };
    const int wr = wave >> 1, wc = wave & 1;  // this wave's 32 x 32 quadrant
    v4i acc[2][2][kND];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int d = 0; d < kND; ++d) acc[i][j][d] = v4i{0, 0, 0, 0};

    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < steps; ++t) {
        const int buf = t & 1;
        if (t + 1 < steps) stage(t + 1, buf ^ 1);
        const unsigned char *sa = smem + buf * kStage + lane * 16;
        const unsigned char *sb = sa + 16 * kFragBytes;
        v4i fa[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int a = 0; a < 4; ++a) fa[i][a] = *reinterpret_cast<const v4i *>(sa + ((2 * wr + i) * 4 + a) * kFragBytes);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const v4i fb = *reinterpret_cast<const v4i *>(sb + ((2 * wc + j) * 4 + b) * kFragBytes);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a + b >= kND) continue;  // Z/2^M: 256^(a+b) vanishes
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j][a + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i][a], fb, acc[i][j][a + b], 0, 0, 0);
                }
            }
        __syncthreads();
    }
    // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r_base + wr * 32 + i * 16 + 4 * fg + r, col = c_base + wc * 32 + j * 16 + fr;
                if (row >= rows || col >= dim) continue;
                uint32_t val;
                if (RING == kRingFp) {
                    // 256^d mod p for d = 4, 5, 6: 2^32 = 2, 2^40 = 2^9, 2^48 = 2^17; |sum| < 2^58, made positive by p 2^28
                    const long long v = (long long)acc[i][j][0][r] + (long long)acc[i][j][1][r] * 256 + (long long)acc[i][j][2][r] * 65536 +
                                        (long long)acc[i][j][3][r] * 16777216 + (long long)acc[i][j][4 % kND][r] * 2 +
                                        (long long)acc[i][j][5 % kND][r] * 512 + (long long)acc[i][j][6 % kND][r] * 131072;
                    val = fp_reduce64((unsigned long long)(v + ((long long)kFpP << 28)));
                } else {
                    val = ((uint32_t)acc[i][j][0][r] + ((uint32_t)acc[i][j][1][r] << 8) + ((uint32_t)acc[i][j][2][r] << 16) +
                           ((uint32_t)acc[i][j][3][r] << 24)) & mask;
                }
                const int arow = row0 + row;
                if (nand) {  // I - P (gsw_impl.rs:114-123)
                    val = ring_sub<RING>(0u, val, mask);
                    if (arow == col) val = ring_add<RING>(val, ring_from_u64<RING>(1ull, mask), mask);
                }
                out[(size_t)arow * dim + col] = val;
            }
}

int dense_limb_pad(int dim) { return (dim + kDenseTile - 1) / kDenseTile * kDenseTile; }

hipError_t launch_dense_split_transposed(const RingDesc &ring, const uint32_t *src, int dim, uint8_t *planes, hipStream_t stream) {
    const int pad = dense_limb_pad(dim);
    dim3 grid((unsigned)(pad / kDenseTile), (unsigned)(pad / kDenseTile));
    if (ring.kind == kRingFp) hipLaunchKernelGGL((dense_split_kernel<kRingFp, true>), grid, dim3(256), 0, stream, src, dim, pad, planes);
    else hipLaunchKernelGGL((dense_split_kernel<kRingZpow2, true>), grid, dim3(256), 0, stream, src, dim, pad, planes);
    return hipGetLastError();
}

// The left operand's slab: rows row0 .. row0 + rows - 1 of a dim x dim matrix into planes [4][rows_pad][pad]
template <int RING>
__global__ __launch_bounds__(256) void dense_split_rows_kernel(const uint32_t *src, int dim, int pad, int row0, int rows, int rows_pad,
                                                                uint8_t *planes) {
    const size_t plane = (size_t)rows_pad * pad;
    const size_t quads = (size_t)rows_pad * (pad / 4);
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (size_t)gridDim.x * 256) {
        const int i = (int)(q / (size_t)(pad / 4)), k = (int)(q - (size_t)i * (pad / 4)) * 4;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[j] = (i < rows && k + j < dim) ? signed_digits<RING>(src[(size_t)(row0 + i) * dim + k + j]) : 0u;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t packed = ((w[0] >> (8 * d)) & 0xFFu) | (((w[1] >> (8 * d)) & 0xFFu) << 8) |
                                    (((w[2] >> (8 * d)) & 0xFFu) << 16) | (((w[3] >> (8 * d)) & 0xFFu) << 24);
            *reinterpret_cast<uint32_t *>(planes + d * plane + (size_t)i * pad + k) = packed;
        }
    }
}

hipError_t launch_dense_split_rows(const RingDesc &ring, const uint32_t *src, int dim, int row0, int rows, uint8_t *planes,
                                   hipStream_t stream) {
    const int pad = dense_limb_pad(dim), rows_pad = dense_limb_pad(rows);
    dim3 grid(grid_for((size_t)rows_pad * (pad / 4), 256)), block(256);
    if (ring.kind == kRingFp)
        hipLaunchKernelGGL(dense_split_rows_kernel<kRingFp>, grid, block, 0, stream, src, dim, pad, row0, rows, rows_pad, planes);
    else
        hipLaunchKernelGGL(dense_split_rows_kernel<kRingZpow2>, grid, block, 0, stream, src, dim, pad, row0, rows, rows_pad, planes);
    return hipGetLastError();
}

hipError_t launch_dense_limb_gemm(const RingDesc &ring, const uint8_t *lhs_planes, const uint8_t *rhs_planes, uint32_t *out, int dim,
                                  int row0, int rows, int nand, hipStream_t stream) {
    if (dim <= 0 || rows <= 0) return hipSuccess;
    if (dim >= 32768) return hipErrorInvalidValue;  // int32 accumulators: 4 * 2^14 * dim < 2^31
    const int pad = dense_limb_pad(dim);
    const int row_tiles = dense_limb_pad(rows) / kDenseTile, col_tiles_n = pad / kDenseTile;
    const long groups = (long)((row_tiles + 63) / 64) * ((col_tiles_n + 7) / 8);
    const long blocks = groups * 64 * 8;
    if (blocks > 0x7FFFFFFFL) return hipErrorInvalidValue;
    const size_t lds = 2 * 32 * kFragBytes;
    if (ring.kind == kRingFp)
        hipLaunchKernelGGL(dense_limb_gemm_kernel<kRingFp>, dim3((unsigned)blocks), dim3(256), lds, stream, lhs_planes, rhs_planes, out,
                           dim, pad, row0, rows, nand, ring.mask, col_tiles_n, row_tiles);
    else
        hipLaunchKernelGGL(dense_limb_gemm_kernel<kRingZpow2>, dim3((unsigned)blocks), dim3(256), lds, stream, lhs_planes, rhs_planes, out,
                           dim, pad, row0, rows, nand, ring.mask, col_tiles_n, row_tiles);
    return hipGetLastError();
}

// ------------------------------------------------------------------ decrypt support
//
// <row, vec> mod q for a few rows of each batch entry: decrypt reads row ell-1
// (gsw_impl.rs:48-52), mp_decrypt rows 0..ell-1 (gsw_impl.rs:71-72).

template <int RING>
__global__ __launch_bounds__(256) void row_dot_kernel(const uint32_t *x, size_t stride, int ld, int cols, int row0,
                                                       const uint32_t *vec, uint32_t *out, uint32_t mask) {
    __shared__ uint64_t part[256];
    const int k = blockIdx.x, b = blockIdx.y;
    const uint32_t *row = x + (size_t)b * stride + (size_t)(row0 + k) * ld;
    uint64_t acc = 0;
    for (int j = threadIdx.x; j < cols; j += 256) {
        const uint64_t prod = (uint64_t)row[j] * vec[j];
        if (RING == kRingFp) acc += (prod & kFpP) + (prod >> 31);
        else acc += prod;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)b * gridDim.x + k] = ring_from_u64<RING>(part[0], mask);
}

hipError_t launch_row_dot(const RingDesc &ring, const uint32_t *x, size_t stride, int ld, int cols, int row0,
                          int nrows, const uint32_t *vec, int batch, uint32_t *out, hipStream_t stream) {
    if (batch <= 0 || nrows <= 0) return hipSuccess;
    // the batch rides in gridDim.y, which holds at most 65535: longer batches go in slabs
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int cur = batch - b0 < 65535 ? batch - b0 : 65535;
        dim3 grid((unsigned)nrows, (unsigned)cur), block(256);
        const uint32_t *xs = x + (size_t)b0 * stride;
        uint32_t *os = out + (size_t)b0 * nrows;
        if (ring.kind == kRingFp)
            hipLaunchKernelGGL(row_dot_kernel<kRingFp>, grid, block, 0, stream, xs, stride, ld, cols, row0, vec, os,
                               ring.mask);
        else
            hipLaunchKernelGGL(row_dot_kernel<kRingZpow2>, grid, block, 0, stream, xs, stride, ld, cols, row0, vec, os,
                               ring.mask);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// pk_matrix = [B t + e | B] (pk.rs:28-38) assembled on the device: row i gets bt[i] + e[i] in column 0, row i of
// B (m x n, tight) in columns 1..n, zeros in the padding columns up to the leading dimension.
template <int RING>
__global__ __launch_bounds__(256) void assemble_pk_kernel(const uint32_t *bt, const uint32_t *e, const uint32_t *rand_b,
                                                           int m, int n, int ld, uint32_t *pk, uint32_t mask) {
    const size_t total = (size_t)m * ld;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int i = (int)(idx / (size_t)ld), j = (int)(idx - (size_t)i * ld);
        uint32_t v = 0u;
        if (j == 0) v = ring_add<RING>(bt[i], e[i], mask);
        else if (j <= n) v = rand_b[(size_t)i * n + (j - 1)];
        pk[idx] = v;
    }
}

hipError_t launch_assemble_pk(const RingDesc &ring, const uint32_t *bt, const uint32_t *e, const uint32_t *rand_b, int m,
                              int n, int ld, uint32_t *pk, hipStream_t stream) {
    if (m <= 0) return hipSuccess;
    dim3 grid(grid_for((size_t)m * ld, 256)), block(256);
    if (ring.kind == kRingFp)
        hipLaunchKernelGGL(assemble_pk_kernel<kRingFp>, grid, block, 0, stream, bt, e, rand_b, m, n, ld, pk, ring.mask);
    else
        hipLaunchKernelGGL(assemble_pk_kernel<kRingZpow2>, grid, block, 0, stream, bt, e, rand_b, m, n, ld, pk, ring.mask);
    return hipGetLastError();
}

// ------------------------------------------------------------------ sampling and digests

// rnd_ring_elm (error_sampling.rs:20-24): R::from(uniform in [min, max])
template <int RING>
__global__ void fill_uniform_kernel(uint64_t seed, uint64_t stream_id, uint64_t min, uint64_t range, uint32_t *out,
                                    size_t count, uint32_t mask) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < count;
         idx += (size_t)gridDim.x * blockDim.x) {
        const uint64_t r = rng_u64(seed, stream_id, idx);
        out[idx] = ring_from_u64<RING>(min + (((r >> 32) * range) >> 32), mask);
    }
}

hipError_t launch_fill_uniform(const RingDesc &ring, uint64_t seed, uint64_t stream_id, uint64_t min, uint64_t max,
                               uint32_t *out, size_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    dim3 grid(grid_for(count, 256)), block(256);
    const uint64_t range = max - min + 1;
    if (ring.kind == kRingFp)
        hipLaunchKernelGGL(fill_uniform_kernel<kRingFp>, grid, block, 0, stream, seed, stream_id, min, range, out,
                           count, ring.mask);
    else
        hipLaunchKernelGGL(fill_uniform_kernel<kRingZpow2>, grid, block, 0, stream, seed, stream_id, min, range, out,
                           count, ring.mask);
    return hipGetLastError();
}

// rnd_dmatrix(N, m, 0, 1) (gsw_impl.rs:30) as packed words: word (b, row, j) = ell uniform bits from
// draw number ((first_entry + b)*rows + row)*words + j (the last word keeps only last_bits of them when ell
// does not divide m); padding words j >= words are zero.  A thread makes four consecutive words of a row and
// stores them as 16 bytes (VEC) or one by one (a destination that is not 16-byte aligned); the batch entry is
// blockIdx.y, so the index arithmetic stays in 32 bits.  `key` is the generator's stream key, mixed on the host.
template <bool VEC>
__global__ __launch_bounds__(256) void random_words_kernel(uint64_t key, uint32_t *out, int rows, int words, int ld,
                                                           uint32_t keep, uint32_t keep_last, size_t stride,
                                                           size_t first_entry) {
    const uint32_t ld4 = (uint32_t)ld >> 2;
    const uint32_t quads = (uint32_t)rows * ld4;
    uint32_t *dst = out + (size_t)blockIdx.y * stride;
    const uint64_t row0 = (first_entry + blockIdx.y) * (uint64_t)rows;
    for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < quads; q += gridDim.x * 256u) {
        const uint32_t row = q / ld4;
        const int j0 = (int)(q - row * ld4) * 4;
        const uint64_t draw = (row0 + row) * (uint64_t)words + (uint64_t)j0;
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = j0 + i;
            w[i] = j < words ? (uint32_t)mix64(key + draw + (uint64_t)i) & (j == words - 1 ? keep_last : keep) : 0u;
        }
        uint32_t *p = dst + (size_t)row * ld + j0;
        if (VEC) {
            *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] = w[i];
        }
    }
}

hipError_t launch_random_words(uint64_t seed, uint64_t stream_id, uint32_t *out, int rows, int words, int ld, int ell,
                               int last_bits, int batch, size_t stride, hipStream_t stream, size_t first_entry) {
    if (rows <= 0 || batch <= 0) return hipSuccess;
    if (ld <= 0 || (ld & 3) || words > ld || (size_t)rows * (size_t)ld >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    const uint64_t key = mix64(seed ^ (stream_id * 0xD1342543DE82EF95ull));
    const uint32_t keep = ell >= 32 ? 0xFFFFFFFFu : ((1u << ell) - 1u);
    const uint32_t keep_last = last_bits >= 32 ? 0xFFFFFFFFu : ((1u << last_bits) - 1u);
    const bool vec = (reinterpret_cast<uintptr_t>(out) & 15u) == 0 && (stride & 3u) == 0;
    const unsigned blocks_x = grid_for((size_t)rows * (size_t)(ld / 4), 256);
    for (int b0 = 0; b0 < batch; b0 += 65535) {  // gridDim.y is a 16-bit quantity
        const int cur = batch - b0 < 65535 ? batch - b0 : 65535;
        dim3 grid(blocks_x, (unsigned)cur), block(256);
        if (vec)
            hipLaunchKernelGGL(random_words_kernel<true>, grid, block, 0, stream, key, out + (size_t)b0 * stride, rows,
                               words, ld, keep, keep_last, stride, first_entry + (size_t)b0);
        else
            hipLaunchKernelGGL(random_words_kernel<false>, grid, block, 0, stream, key, out + (size_t)b0 * stride, rows,
                               words, ld, keep, keep_last, stride, first_entry + (size_t)b0);
    }
    return hipGetLastError();
}

// Inverse-CDF sampling from a signed integer table (cum[idx] = P(X <= idx - kmax) * 2^64), mapped to
// the ring as error_sampling.rs:60-64 does: negative -> -(R::from(|k|)), positive -> R::from(k).
template <int RING>
__global__ void sample_table_kernel(uint64_t seed, uint64_t stream_id, const uint64_t *cum, int table_len, int kmax,
                                    uint32_t *out, size_t count, uint32_t mask) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < count;
         idx += (size_t)gridDim.x * blockDim.x) {
        const uint64_t u = rng_u64(seed, stream_id, idx);
        int lo = 0, hi = table_len - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (u <= cum[mid]) hi = mid; else lo = mid + 1;
        }
        const int k = lo - kmax;
        const uint32_t mag = ring_from_u64<RING>((uint64_t)(k < 0 ? -k : k), mask);
        out[idx] = k < 0 ? ring_sub<RING>(0u, mag, mask) : mag;
    }
}

hipError_t launch_sample_table(const RingDesc &ring, uint64_t seed, uint64_t stream_id, const uint64_t *cum,
                               int table_len, int kmax, uint32_t *out, size_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    dim3 grid(grid_for(count, 256)), block(256);
    if (ring.kind == kRingFp)
        hipLaunchKernelGGL(sample_table_kernel<kRingFp>, grid, block, 0, stream, seed, stream_id, cum, table_len,
                           kmax, out, count, ring.mask);
    else
        hipLaunchKernelGGL(sample_table_kernel<kRingZpow2>, grid, block, 0, stream, seed, stream_id, cum, table_len,
                           kmax, out, count, ring.mask);
    return hipGetLastError();
}

// digest_b = sum_i (x_i + 1) * (2 i + 1) * 0x9E3779B97F4A7C15 mod 2^64 (order independent)
__global__ __launch_bounds__(256) void digest_kernel(const uint32_t *x, size_t stride, size_t count,
                                                      unsigned long long *digests) {
    __shared__ unsigned long long part[256];
    const int b = blockIdx.y;
    const uint32_t *p = x + (size_t)b * stride;
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
        acc += ((unsigned long long)p[i] + 1ull) * ((2ull * i + 1ull) * 0x9E3779B97F4A7C15ull);
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(&digests[b], part[0]);
}

// Counts the words of a batch of rows x ld matrices that break the operand invariants of the contraction:
// a value that is not canonical (>= p, or a bit at or above M), or a non-zero word in a padding column.
template <int RING>
__global__ __launch_bounds__(256) void validate_kernel(const uint32_t *x, size_t stride, int rows, int cols, int ld,
                                                        uint32_t mask, unsigned long long *bad) {
    const int b = blockIdx.y;
    const uint32_t *p = x + (size_t)b * stride;
    const size_t total = (size_t)rows * ld;
    unsigned count = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t v = p[i];
        const int col = (int)(i % (size_t)ld);
        if (col >= cols) count += v != 0u;
        else count += RING == kRingFp ? v >= kFpP : (v & ~mask) != 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) count += (unsigned)__shfl_xor((int)count, off, 64);
    if ((threadIdx.x & 63) == 0 && count) atomicAdd(bad, (unsigned long long)count);
}

hipError_t launch_validate(const RingDesc &ring, const uint32_t *x, size_t stride, int rows, int cols, int ld,
                           int batch, unsigned long long *bad, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(bad, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess || batch <= 0 || rows <= 0) return e;
    unsigned gx = (unsigned)(((size_t)rows * ld + 256 * 16 - 1) / (256 * 16));
    if (gx > 64) gx = 64;
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int cur = batch - b0 < 65535 ? batch - b0 : 65535;
        dim3 grid(gx, (unsigned)cur), block(256);
        const uint32_t *xs = x + (size_t)b0 * stride;
        if (ring.kind == kRingFp)
            hipLaunchKernelGGL(validate_kernel<kRingFp>, grid, block, 0, stream, xs, stride, rows, cols, ld, ring.mask, bad);
        else
            hipLaunchKernelGGL(validate_kernel<kRingZpow2>, grid, block, 0, stream, xs, stride, rows, cols, ld, ring.mask, bad);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_digest(const uint32_t *x, size_t stride, size_t count, int batch, unsigned long long *digests,
                         hipStream_t stream) {
    if (batch <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(digests, 0, sizeof(unsigned long long) * (size_t)batch, stream);
    if (e != hipSuccess) return e;
    if (count == 0) return hipSuccess;
    unsigned gx = (unsigned)((count + 256 * 16 - 1) / (256 * 16));
    if (gx > 64) gx = 64;
    if (gx == 0) gx = 1;
    // the batch rides in gridDim.y, which holds at most 65535: longer batches go in slabs
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int cur = batch - b0 < 65535 ? batch - b0 : 65535;
        hipLaunchKernelGGL(digest_kernel, dim3(gx, (unsigned)cur), dim3(256), 0, stream, x + (size_t)b0 * stride, stride,
                           count, digests + b0);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace gswk
