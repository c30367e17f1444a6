// Launch interface between the C-ABI layer (gsw_capi.hip) and the gfx950 kernels
// (gsw_kernels.hip).  Everything here is device-pointer based and asynchronous on `stream`.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace gswk {

constexpr int kRingFp = 0;
constexpr int kRingZpow2 = 1;
constexpr uint32_t kFpP = 2147483647u;

// Tile geometry of the contraction kernel.
constexpr int kTileRows = 256;    // output rows per workgroup
constexpr int kTileCols = 32;     // output columns (each 4 byte limbs wide) per workgroup
constexpr int kStageElems = 4;    // left-operand elements (32 K-slots each) per pipeline stage
constexpr int kFragBytes = 1024;  // one 32x32 int8 MFMA operand fragment

struct RingDesc {
    int kind;       // kRingFp / kRingZpow2
    int ell;        // bits per element (R::Num_Bits)
    uint32_t mask;  // p for Fp, 2^M - 1 for Zpow2
};

// epilogue modes of the contraction
constexpr int kModeMul = 0;      // out = P
constexpr int kModeNand = 1;     // out = G - P
constexpr int kModeEncrypt = 2;  // out = P + addend[b] * G

struct GemmArgs {
    const uint32_t *lhs;  // binary operand as packed words: M rows x ld_lhs, ell bits used per word
    size_t lhs_stride;    // uint32 elements between batch entries
    const uint8_t *rhs;   // prepared limb planes: [NT][E4][4][1024]
    size_t rhs_stride;    // bytes between batch entries
    uint32_t *out;        // M rows x ld_out
    size_t out_stride;
    const uint32_t *addend;  // kModeEncrypt: one message per batch entry
    int rows;                // M
    int e4;                  // left-operand words per row, multiple of kStageElems
    int k_words;             // of which the first k_words meet right-operand digits that may be non-zero (0: not stated,
                             // all of them); lets the kernel drop the MFMA step of a last stage that holds padding only
    int ld_lhs;
    int out_cols;  // valid output columns W
    int ld_out;
    int batch;
    int mode;
    int packed_tail;              // the last column tile is a packed one: kTailPacked / kTailNarrow (see tail_mode)
    int pass_rows, pass_cols;     // placement passes of the blocked workgroup order (row tiles x column tiles)
    unsigned long long *digests;  // optional: digests[b] += digest of the stored tile (zeroed by the caller)
    // optional slot lists (device, `batch` entries): entry b reads lhs + lhs_slot[b] * lhs_stride and
    // rhs + rhs_slot[b] * rhs_stride (several entries may share one prepared right operand) and writes
    // out + out_slot[b] * out_stride; null means slot b
    const uint32_t *lhs_slot;
    const uint32_t *out_slot;
    const uint32_t *rhs_slot;
    // Filled in by launch_gemm.  The accumulators of limb l start at acc_init[l] instead of 0: the right operand's
    // limbs are signed digits, so a limb sum may be negative, and for Fp the start values (chosen so that
    // sum_l 2^(8 l) acc_init[l] = 0 mod p) keep every sum non-negative without changing the result.  lean_combine:
    // the sums stay below the bound of the 32-bit recombination (combine_limbs_lean) and interior tiles may take the
    // branch-free epilogue; 0 sends every tile through the general epilogue (64-bit recombination).
    uint32_t acc_init[4];
    int lean_combine;
    // How the left fragments are made (see "Scaled bits" in gsw_kernels.hip): 0 = bytes in {0, 1}; Fp and the default
    // kernel only: 1 / 2 = bit i of a byte kept in place (bytes in {0, 2^c}, c = i & 1 / i), the prepared right operand
    // scaled by 2^-c to match.  Must be the mode the right operand was prepared with (bit_scale_mode of the ring
    // and e4 when the default kernel runs, else 0).
    int scale_mode;
};

inline int col_tiles(int out_cols) { return (out_cols + kTileCols - 1) / kTileCols; }
// A last column tile with at most 8 valid columns can be stored packed: one fragment per K element
// whose MFMA column 8 l + c carries limb l of column c (a quarter of the MFMA work of a full tile).
inline bool can_pack_tail(int out_cols) { return out_cols % kTileCols >= 1 && out_cols % kTileCols <= 8; }
// How the last column tile of a prepared operand is stored.  kTailPacked: at most 8 valid columns, 32 MFMA columns
// (two accumulators per row group), MFMA column 8 l + c carries limb l of column c.  kTailNarrow: at most 4 valid
// columns, 16 MFMA columns (one accumulator, half the MFMAs), MFMA column 4 l + c; one fragment per MFMA step, the
// stages 2 KiB apart.  Only the default kernel reads the narrow form (narrow_ok).
constexpr int kTailNone = 0, kTailPacked = 1, kTailNarrow = 2;
inline int tail_mode(int out_cols, bool narrow_ok) {
    const int r = out_cols % kTileCols;
    if (r >= 1 && r <= 4 && narrow_ok) return kTailNarrow;
    return can_pack_tail(out_cols) ? kTailPacked : kTailNone;
}
// (the narrow form leaves part of the packed tile's space unused)
inline size_t prepared_bytes(int out_cols, int e4, bool packed_tail) {
    const size_t frags = (size_t)col_tiles(out_cols) * 4 - (packed_tail ? 3 : 0);
    return frags * e4 * kFragBytes;
}

// Right-operand preparation: src is rows_valid x cols_valid (leading dim ld_src); K-slot (e, bit) is
// source row e*ell + bit.  dst gets [NT][e4][4 limbs][1024 B] planes of signed base-256 digits (each in
// [-128, 127], see signed_digits in gsw_kernels.hip).
struct PrepArgs {
    const uint32_t *src;
    size_t src_stride;  // uint32 elements between batch entries
    uint8_t *dst;
    size_t dst_stride;  // bytes
    int rows_valid;
    int cols_valid;
    int ld_src;
    int elems;  // valid K elements (rows_valid = elems * ell)
    int e4;
    int batch;
    int packed_tail;  // kTailNone / kTailPacked / kTailNarrow
    int layout16;  // fragment order of the 16x16x64 kernels (variants 4, 5) instead of the 32x32x32 ones
    int scale_mode;  // rows scaled by 2^-c for the contraction's scaled bits (GemmArgs::scale_mode); layout16 only
    const uint32_t *src_slot;  // optional slot list (device, `batch` entries) into src; null means slot b
    const uint32_t *dst_slot;  // optional slot list into dst (a cache of prepared operands); null means slot b
};

hipError_t launch_prepare(const RingDesc &ring, const PrepArgs &a, hipStream_t stream);
// variant: 0 = pipelined 32x32x32 MFMA, 256-row tile; 1 = same, 512-row tile;
//          2 = simple two-buffer LDS-DMA kernel; 3 = simple kernel with register staging;
//          4 = pipelined 16x16x64 MFMA, 256-row tile; 5 = same, 512-row tile;
//          6 = 16x16x64 with 64-byte left-operand loads, 256-row tile; 7 = same, 512-row tile.
// The prepared right operand must have been built for the variant's fragment order.
// *lean_used (optional) receives GemmArgs::lean_combine as the launched kernel got it: 1 = 32-bit recombination,
// 0 = the general epilogue with the 64-bit one on every tile.
hipError_t launch_gemm(const RingDesc &ring, const GemmArgs &a, int variant, hipStream_t stream, int *lean_used = nullptr);
// The accumulator start values launch_gemm uses for a contraction over k_slots K slots (GemmArgs::acc_init) and
// whether the 32-bit recombination applies (host only; `wide` asks for the 64-bit one).  False: no values exist.
// (k_slots counts K slots whose left-operand byte is 0 or 1; with scaled bits pass scale_units(e4, mode))
bool limb_start_values(const RingDesc &ring, long k_slots, bool wide, uint32_t (&c)[4], int *lean);
// The scaling of the left fragments' bits the default kernel uses for a contraction over e4 left-operand words
// (0 for Z/2^M), and the bound of its limb sums in units of {0, 1} K slots.
int bit_scale_mode(const RingDesc &ring, int e4);
long scale_units(int e4, int mode);

#ifdef GSW_EXP_TILE_CLOCK
hipError_t read_tile_clock(unsigned long long *out16, bool reset);
#endif

// output formats of the gadget decomposition (GSW_DECOMP_* of the C header)
constexpr int kDecompU32 = 0;   // one uint32 per bit: the reference's matrix of ring 0 / 1 values
constexpr int kDecompU8 = 1;    // one byte per bit
constexpr int kDecompBits = 2;  // the row's bits as a little-endian bit string, 32 per word
// elements (uint32 / bytes / uint32 words) per output row, rows tight
size_t bit_decomp_pitch(int cols, int ell, int format);
hipError_t launch_bit_decomp(const RingDesc &ring, const uint32_t *in, int rows, int cols, int ld_in, void *out,
                             int format, hipStream_t stream);
hipError_t launch_bit_decomp_inv(const RingDesc &ring, const uint32_t *in, int rows, int cols, uint32_t *out,
                                 int ld_out, int quirk, hipStream_t stream);
hipError_t launch_add(const RingDesc &ring, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t count,
                      hipStream_t stream);
hipError_t launch_mult_const(const RingDesc &ring, uint32_t *x, uint32_t k, size_t count, hipStream_t stream);
hipError_t launch_mul_dense(const RingDesc &ring, const uint32_t *a, const uint32_t *b, uint32_t *out, int dim,
                            int nand, hipStream_t stream);
// Dense product with a flattened left operand (see pack_flat_lhs_kernel): c1 dim x dim -> packed dim x ld words
// of ell bits, diag[dim] (diagonal entries other than 0 / 1), misfits[0] = entries that do not fit the form,
// misfits[1] = rows with a non-zero diag entry.
hipError_t launch_pack_flat_lhs(const RingDesc &ring, const uint32_t *c1, int dim, int words, int ld, uint32_t *packed,
                                uint32_t *diag, unsigned long long *misfits, hipStream_t stream);
// out += diag(d) * c2, then out = I - out when nand
hipError_t launch_dense_fixup(const RingDesc &ring, uint32_t *out, const uint32_t *c2, const uint32_t *diag, int dim,
                              int nand, hipStream_t stream);
// Dense product with arbitrary entries as 4 x 4 signed-digit int8 products (see dense_limb_gemm_kernel).
// dense_limb_pad: dim rounded up to the 64 x 64 tile.  Right operand: launch_dense_split_transposed writes its digit
// planes [4][pad][pad] ([digit][column][k]); left operand: launch_dense_split_rows writes those of rows
// row0 .. row0 + rows - 1 as [4][dense_limb_pad(rows)][pad]; launch_dense_limb_gemm computes those rows of the
// product (nand: of I - product).  dim < 32768.
int dense_limb_pad(int dim);
hipError_t launch_dense_split_transposed(const RingDesc &ring, const uint32_t *src, int dim, uint8_t *planes, hipStream_t stream);
hipError_t launch_dense_split_rows(const RingDesc &ring, const uint32_t *src, int dim, int row0, int rows, uint8_t *planes,
                                   hipStream_t stream);
hipError_t launch_dense_limb_gemm(const RingDesc &ring, const uint8_t *lhs_planes, const uint8_t *rhs_planes, uint32_t *out, int dim,
                                  int row0, int rows, int nand, hipStream_t stream);
// out[b * nrows + k] = <x_b[row0 + k, 0..cols), vec> mod q
hipError_t launch_row_dot(const RingDesc &ring, const uint32_t *x, size_t stride, int ld, int cols, int row0,
                          int nrows, const uint32_t *vec, int batch, uint32_t *out, hipStream_t stream);
// pk_matrix = [bt + e | rand_b] with a padded leading dimension (pk.rs:28-38)
hipError_t launch_assemble_pk(const RingDesc &ring, const uint32_t *bt, const uint32_t *e, const uint32_t *rand_b, int m,
                              int n, int ld, uint32_t *pk, hipStream_t stream);
hipError_t launch_fill_uniform(const RingDesc &ring, uint64_t seed, uint64_t stream_id, uint64_t min, uint64_t max,
                               uint32_t *out, size_t count, hipStream_t stream);
// entry b of the call draws as entry first_entry + b of one long batch would
hipError_t launch_random_words(uint64_t seed, uint64_t stream_id, uint32_t *out, int rows, int words, int ld,
                               int ell, int last_bits, int batch, size_t stride, hipStream_t stream,
                               size_t first_entry = 0);
hipError_t launch_sample_table(const RingDesc &ring, uint64_t seed, uint64_t stream_id, const uint64_t *cum,
                               int table_len, int kmax, uint32_t *out, size_t count, hipStream_t stream);
// counter calibration (see stream_read_kernel)
hipError_t launch_stream_read(const void *src, size_t bytes, int mode, size_t pitch, hipStream_t stream);
// *bad = number of non-canonical or non-zero-padding words over the batch (see validate_kernel)
hipError_t launch_validate(const RingDesc &ring, const uint32_t *x, size_t stride, int rows, int cols, int ld,
                           int batch, unsigned long long *bad, hipStream_t stream);
hipError_t launch_digest(const uint32_t *x, size_t stride, size_t count, int batch, unsigned long long *digests,
                         hipStream_t stream);

}  // namespace gswk
