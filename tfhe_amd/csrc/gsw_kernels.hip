// gfx950 (MI355X, CDNA4) kernels for the GSW hot path.
//
// The product of the reference, `ciphertext1 * cipertext2` (src/gsw/gsw_impl.rs:107), with a
// flattened left operand C1 = BitDecomp(C1c) collapses to Pc = BitDecomp(C1c) * C2c mod q:
// an (N x N binary) x (N x (n+1)) contraction.  The binary operand is never materialised: its
// int8 MFMA fragments are expanded in registers from the packed words of C1c.  The right operand
// is split into four signed byte limbs (value - 128) stored in MFMA-fragment order, so each
// 32x32x32 int8 MFMA consumes exactly one word per row (32 K-slots) of the left operand and one
// 1 KiB fragment of one limb; accumulation is exact in int32 and the four limbs of one output
// element live in the same lane, so recombination needs no cross-lane traffic.
//
// K-slot order.  Slot (h, i, t) of a word -- lane half h = lane >> 5, fragment dword i, byte t --
// stands for bit 4h + i + 8t.  The order is free as long as both operands agree; this one makes
// the left fragment two VALU ops per dword: ((x >> 4h) >> i) & 0x01010101.
#include "gsw_kernels.h"

namespace gswk {

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

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint64_t rng_u64(uint64_t seed, uint64_t stream, uint64_t counter) {
    return mix64(mix64(seed ^ (stream * 0xD1342543DE82EF95ull)) + counter);
}

// ------------------------------------------------------------------ right-operand preparation
//
// One workgroup per (K element e, column tile nt, batch entry): reads the 32 source rows
// e*ell .. e*ell+31 x 32 columns (128 B contiguous per row), splits every word into four byte
// limbs, biases them by -128 (the int8 MFMA is signed) and writes the four 1 KiB fragments
// through LDS so that the global stores are 16 B per lane and fully coalesced.
// Thread (cc = tid & 31, h = tid >> 7, i = (tid >> 5) & 3) owns fragment dword i of lane 32h + cc.

__global__ __launch_bounds__(256) void prepare_rhs_kernel(PrepArgs a, int ell) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[4 * 256];
    const int tid = threadIdx.x;
    const int e = blockIdx.x, nt = blockIdx.y, b = blockIdx.z;
    const int nt_count = gridDim.y;
    const int cc = tid & 31, i = (tid >> 5) & 3, h = tid >> 7;
    const uint32_t *src = a.src + (size_t)b * a.src_stride;
    const int col = nt * kTileCols + cc;
    uint32_t limb[4] = {0u, 0u, 0u, 0u};
    if (e < a.elems && col < a.cols_valid) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int bit = 4 * h + i + 8 * t;
            const int row = e * ell + bit;
            if (bit < ell && row < a.rows_valid) {
                const uint32_t x = src[(size_t)row * a.ld_src + col] ^ 0x80808080u;  // byte - 128 as int8
#pragma unroll
                for (int l = 0; l < 4; ++l) limb[l] |= ((x >> (8 * l)) & 0xFFu) << (8 * t);
            }
        }
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) tile[l * 256 + (32 * h + cc) * 4 + i] = limb[l];
    __syncthreads();
    uint8_t *dst = a.dst + (size_t)b * a.dst_stride + ((size_t)nt * a.e4 + e) * (4 * kFragBytes);
    (void)nt_count;
    reinterpret_cast<uint4 *>(dst)[tid] = reinterpret_cast<const uint4 *>(tile)[tid];
}

hipError_t launch_prepare(const RingDesc &ring, const PrepArgs &a, hipStream_t stream) {
    if (a.batch <= 0) return hipSuccess;
    dim3 grid((unsigned)a.e4, (unsigned)col_tiles(a.cols_valid), (unsigned)a.batch);
    hipLaunchKernelGGL(prepare_rhs_kernel, grid, dim3(256), 0, stream, a, ring.ell);
    return hipGetLastError();
}

// ------------------------------------------------------------------ the contraction
//
// Workgroup: 256 threads = 4 waves, output tile 256 rows x 32 columns x 4 limbs.
// Wave w owns rows 64w .. 64w+63 as two 32-row groups; per K element it issues
// 2 (row groups) x 4 (limbs) MFMA 32x32x32 i8 that share 2 left and 4 right fragments.
// Pipeline: two LDS buffers; stage t+1 is fetched by LDS-DMA while stage t is computed.
// LDS image of a stage: 4 KiB of left words (row r of the tile at 16 r: its 4 words) followed by
// 16 fragments (element s, limb l at (4 s + l) KiB), each read lane-linearly (conflict free).

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

    // Workgroups id and id+8 share an XCD (round-robin dispatch): give each XCD whole
    // (batch entry, column tile) units so that the mt_count row tiles that stream the same
    // right-operand panel run on one L2.  Placement only affects speed, never results.
    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int mt = local % mt_count;
    const int unit = (local / mt_count) * 8 + xcd;
    if (unit >= p.batch * nt_count) return;
    const int b = unit / nt_count, nt = unit - b * nt_count;

    const uint32_t *lhs = p.lhs + (size_t)b * p.lhs_stride;
    const uint8_t *rhs = p.rhs + (size_t)b * p.rhs_stride + (size_t)nt * p.e4 * (4 * kFragBytes);
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
            for (int r = 0; r < 16; ++r) acc[g][l][r] = 0;
    uint32_t pop0 = 0, pop1 = 0;
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
        const uint32_t w0[4] = {a0.x, a0.y, a0.z, a0.w};
        const uint32_t w1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int s = 0; s < kStageElems; ++s) {
            pop0 += __popc(w0[s]);
            pop1 += __popc(w1[s]);
            const uint32_t y0 = w0[s] >> sh, y1 = w1[s] >> sh;
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
            for (int l = 0; l < 4; ++l) {
                const v4i fb = *reinterpret_cast<const v4i *>(sb + (s * 4 + l) * kFragBytes + lane * 16);
                acc[0][l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0, fb, acc[0][l], 0, 0, 0);
                acc[1][l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f1, fb, acc[1][l], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // Epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2)
    // + 4 (lane >> 5).  The limbs were biased by -128, so every limb sum gets 128 * popcount(row)
    // back; lane r32 of either half holds the popcount of row r32 of each group.
    const int col = nt * kTileCols + r32;
    const uint32_t addend = (p.mode == kModeEncrypt) ? p.addend[b] : 0u;
    uint32_t *out = p.out + (size_t)b * p.out_stride;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t pop = g == 0 ? pop0 : pop1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const uint32_t bias = 128u * (uint32_t)__shfl((int)pop, rr, 64);
            const int row = mt * kTileRows + wave * 64 + g * 32 + rr;
            const uint64_t s0 = (uint32_t)acc[g][0][r] + bias, s1 = (uint32_t)acc[g][1][r] + bias;
            const uint64_t s2 = (uint32_t)acc[g][2][r] + bias, s3 = (uint32_t)acc[g][3][r] + bias;
            uint32_t val = ring_from_u64<RING>(s0 + (s1 << 8) + (s2 << 16) + (s3 << 24), mask);
            if (row < p.rows && col < p.ld_out) {
                if (col >= p.out_cols) {
                    val = 0u;  // keep the padding columns of the compact form zero
                } else if (p.mode == kModeNand) {
                    val = ring_sub<RING>(gadget_entry<RING>(row, col, ell, mask), val, mask);
                } else if (p.mode == kModeEncrypt) {
                    val = ring_add<RING>(val, ring_mul<RING>(addend, gadget_entry<RING>(row, col, ell, mask), mask),
                                         mask);
                }
                out[(size_t)row * p.ld_out + col] = val;
            }
        }
    }
}

hipError_t launch_gemm(const RingDesc &ring, const GemmArgs &a, int variant, hipStream_t stream) {
    if (a.batch <= 0) return hipSuccess;
    const int nt = col_tiles(a.out_cols);
    const int mt = (a.rows + kTileRows - 1) / kTileRows;
    const long units = (long)a.batch * nt;
    const long groups = (units + 7) / 8;
    const long blocks = groups * 8 * mt;
    if (blocks > 0x7FFFFFFFL) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks), block(256);
#define GSW_LAUNCH(RING, DMA)                                                                                   \
    hipLaunchKernelGGL((binlimb_gemm_kernel<RING, DMA>), grid, block, kGemmLds, stream, a, ring.ell, ring.mask, \
                       nt, mt)
    if (ring.kind == kRingFp) {
        if (variant == 0) GSW_LAUNCH(kRingFp, true); else GSW_LAUNCH(kRingFp, false);
    } else {
        if (variant == 0) GSW_LAUNCH(kRingZpow2, true); else GSW_LAUNCH(kRingZpow2, false);
    }
#undef GSW_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------ gadget helpers

// bit_decomp_matrix (helper.rs:7-21): bit k of column j goes to column j*ell + k, little endian
__global__ void bit_decomp_kernel(const uint32_t *in, int rows, int cols, int ld_in, uint32_t *out, int ell) {
    const size_t wide = (size_t)cols * ell;
    const size_t total = (size_t)rows * wide;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t i = idx / wide, jk = idx - i * wide;
        const int j = (int)(jk / ell), k = (int)(jk - (size_t)j * ell);
        out[idx] = (in[i * ld_in + j] >> k) & 1u;
    }
}

// bit_decomp_inv_matrix (helper.rs:24-49)
template <int RING>
__global__ void bit_decomp_inv_kernel(const uint32_t *in, int rows, int cols, uint32_t *out, int ld_out, int quirk,
                                      int ell, uint32_t mask) {
    const size_t total = (size_t)rows * cols;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t i = idx / cols, j = idx - i * cols;
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

hipError_t launch_bit_decomp(const RingDesc &ring, const uint32_t *in, int rows, int cols, int ld_in, uint32_t *out,
                             hipStream_t stream) {
    const size_t total = (size_t)rows * cols * ring.ell;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(bit_decomp_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, in, rows, cols, ld_in,
                       out, ring.ell);
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
    dim3 grid((unsigned)nrows, (unsigned)batch), block(256);
    if (ring.kind == kRingFp)
        hipLaunchKernelGGL(row_dot_kernel<kRingFp>, grid, block, 0, stream, x, stride, ld, cols, row0, vec, out,
                           ring.mask);
    else
        hipLaunchKernelGGL(row_dot_kernel<kRingZpow2>, grid, block, 0, stream, x, stride, ld, cols, row0, vec, out,
                           ring.mask);
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
// draw number (b*rows + row)*words + j (the last word keeps only last_bits of them when ell does not
// divide m); padding words j >= words are zero.
__global__ void random_words_kernel(uint64_t seed, uint64_t stream_id, uint32_t *out, int rows, int words, int ld,
                                    int ell, int last_bits, int batch, size_t stride) {
    const size_t per = (size_t)rows * ld;
    const size_t total = per * batch;
    const uint32_t keep = ell >= 32 ? 0xFFFFFFFFu : ((1u << ell) - 1u);
    const uint32_t keep_last = last_bits >= 32 ? 0xFFFFFFFFu : ((1u << last_bits) - 1u);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t b = idx / per, rem = idx - b * per;
        const size_t row = rem / ld;
        const int j = (int)(rem - row * ld);
        uint32_t w = 0;
        if (j < words)
            w = (uint32_t)rng_u64(seed, stream_id, (b * rows + row) * words + j) & (j == words - 1 ? keep_last : keep);
        out[b * stride + rem] = w;
    }
}

hipError_t launch_random_words(uint64_t seed, uint64_t stream_id, uint32_t *out, int rows, int words, int ld, int ell,
                               int last_bits, int batch, size_t stride, hipStream_t stream) {
    const size_t total = (size_t)rows * ld * batch;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(random_words_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, seed, stream_id, out,
                       rows, words, ld, ell, last_bits, batch, stride);
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

hipError_t launch_digest(const uint32_t *x, size_t stride, size_t count, int batch, unsigned long long *digests,
                         hipStream_t stream) {
    if (batch <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(digests, 0, sizeof(unsigned long long) * (size_t)batch, stream);
    if (e != hipSuccess) return e;
    if (count == 0) return hipSuccess;
    unsigned gx = (unsigned)((count + 256 * 16 - 1) / (256 * 16));
    if (gx > 64) gx = 64;
    if (gx == 0) gx = 1;
    hipLaunchKernelGGL(digest_kernel, dim3(gx, (unsigned)batch), dim3(256), 0, stream, x, stride, count, digests);
    return hipGetLastError();
}

}  // namespace gswk
