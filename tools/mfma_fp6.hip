// Microbenchmark and exactness check: the block-scaled MFMA v_mfma_scale_f32_16x16x128_f8f6f4 with a binary
// left operand in FP4 (E2M1: 1.0 = code 2) and 5-bit signed digits in FP6 (E2M3: code k stands for k/8, k = 0..16,
// bit 5 = sign), all scales 2^0.  Every product and every partial sum is a multiple of 1/8 far below 2^24 / 8, so
// an f32 accumulation is exact if the instruction adds what it is given.
//  1. exactness against an integer reference that assumes only that element j of K group g of a row's lane meets
//     element j of K group g of a column's lane;
//  2. the rate a register-only loop sustains (random digits, bits of density 1/2), next to the int8 loop's.
// Usage: mfma_fp6 [iterations]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int kFmtFp6 = 2;  // E2M3
constexpr int kFmtFp4 = 4;  // E2M1
constexpr int kScaleOne = 0x7F7F7F7F;  // E8M0 127 = 2^0 in every byte

// one MFMA per wave on the operands of in[] (lane: 8 dwords of A, then 8 dwords of B), result as integers x 8
__global__ __launch_bounds__(64) void one_mfma(const int *in, float *out, int reps) {
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = in[threadIdx.x * 16 + i];
        b[i] = in[threadIdx.x * 16 + 8 + i];
    }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r)
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, kFmtFp4, kFmtFp6, 0, kScaleOne, 0, kScaleOne);
    for (int r = 0; r < 4; ++r) out[threadIdx.x * 4 + r] = acc[r];
}

template <int MODE>  // 0: FP4 x FP6   1: int8 16x16x64 (binary A)   2: FP4 x FP4   3: FP8 x FP8
__global__ __launch_bounds__(256, 2) void mfma_loop(const int *in, float *out, int iters) {
    v8i a0, a1, b[4];
    for (int i = 0; i < 8; ++i) {
        a0[i] = in[threadIdx.x * 8 + i];
        a1[i] = in[2048 + threadIdx.x * 8 + i];
        for (int l = 0; l < 4; ++l) b[l][i] = in[4096 + l * 2048 + threadIdx.x * 8 + i];
    }
    if (MODE == 0 || MODE == 2) {
        for (int i = 0; i < 8; ++i) {  // nibbles 0 / 2 = 0.0 / 1.0
            a0[i] = (a0[i] & 0x11111111) << 1;
            a1[i] = (a1[i] & 0x11111111) << 1;
        }
    }
    if (MODE == 0) {
        // 6-bit codes: keep magnitude <= 16 -> clear bit 4 of every code unless the low four bits are 0 (any 6-bit
        // pattern with magnitude code <= 16 is a valid digit); simply mask each code to sign + 4 bits
        for (int l = 0; l < 4; ++l)
            for (int i = 0; i < 6; ++i) {
                // codes are packed 6 bits each over 192 bits; masking bit 4 of each code: pattern repeats every 3 dwords
                const unsigned m0 = ~((1u << 4) | (1u << 10) | (1u << 16) | (1u << 22) | (1u << 28));
                const unsigned m1 = ~((1u << 2) | (1u << 8) | (1u << 14) | (1u << 20) | (1u << 26));
                const unsigned m2 = ~((1u << 0) | (1u << 6) | (1u << 12) | (1u << 18) | (1u << 24) | (1u << 30));
                b[l][i] &= (int)(i % 3 == 0 ? m0 : i % 3 == 1 ? m1 : m2);
            }
    }
    if (MODE == 1) {
        for (int i = 0; i < 4; ++i) {
            a0[i] &= 0x01010101;
            a1[i] &= 0x01010101;
        }
    }
    if (MODE == 1) {
        v4i acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0;
        v4i x0 = {a0[0], a0[1], a0[2], a0[3]}, x1 = {a1[0], a1[1], a1[2], a1[3]};
        v4i y[4];
        for (int l = 0; l < 4; ++l) y[l] = v4i{b[l][0], b[l][1], b[l][2], b[l][3]};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x0, y[l], acc[l], 0, 0, 0);
                acc[4 + l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x1, y[l], acc[4 + l], 0, 0, 0);
            }
        }
        int s = 0;
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[blockIdx.x * 256 + threadIdx.x] = (float)s;
    } else {
        constexpr int FA = MODE == 3 ? 0 : kFmtFp4;
        constexpr int FB = MODE == 0 ? kFmtFp6 : MODE == 2 ? kFmtFp4 : 0;
        v4f acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                acc[l] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a0, b[l], acc[l], FA, FB, 0, kScaleOne, 0, kScaleOne);
                acc[4 + l] =
                    __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a1, b[l], acc[4 + l], FA, FB, 0, kScaleOne, 0, kScaleOne);
            }
        }
        float s = 0;
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

template <int MODE>
static void run(const char *name, const int *d_in, float *d_out, int iters, int blocks) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((mfma_loop<MODE>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double k = MODE == 1 ? 64.0 : 128.0;
        const double ops = 2.0 * 16 * 16 * k * 8.0 * iters * 4.0 * blocks;
        // digit bits carried per second: slots x bits per right-operand digit
        const double bits = MODE == 1 ? 8.0 : MODE == 0 ? 5.0 : MODE == 2 ? 3.0 : 5.0;
        printf("%-40s rep %d: %8.3f ms  %8.1f TOP/s  %8.1f T(slot x digit bit)/s\n", name, rep, ms, ops / ms / 1e9,
               ops / 2 * bits / ms / 1e9);
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    // ---- exactness
    std::vector<int> host(64 * 16);
    std::vector<int> bits(64 * 32), digits(64 * 32);  // per lane: 32 K elements of A (row lane) and B (column lane)
    srand(7);
    int bad_total = 0;
    for (int trial = 0; trial < 64; ++trial) {
        memset(host.data(), 0, host.size() * 4);
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 32; ++j) {
                const int bit = trial == 0 ? 1 : rand() & 1;
                int d = trial == 1 ? -16 : trial == 2 ? 16 : (rand() % 33) - 16;
                bits[l * 32 + j] = bit;
                digits[l * 32 + j] = d;
                // A: FP4 nibble j of the lane's 128 bits
                host[l * 16 + j / 8] |= (bit ? 2 : 0) << (4 * (j % 8));
                // B: FP6 code j of the lane's 192 bits
                const unsigned code = (unsigned)(d < 0 ? -d : d) | (d < 0 ? 0x20u : 0u);
                const int pos = 6 * j;
                host[l * 16 + 8 + pos / 32] |= (int)(code << (pos % 32));
                if (pos % 32 > 26) host[l * 16 + 8 + pos / 32 + 1] |= (int)(code >> (32 - pos % 32));
            }
        int *d_in;
        float *d_out;
        CHECK(hipMalloc(&d_in, host.size() * 4));
        CHECK(hipMalloc(&d_out, 64 * 4 * 4));
        CHECK(hipMemcpy(d_in, host.data(), host.size() * 4, hipMemcpyHostToDevice));
        const int reps = trial < 3 ? 4000 : 1 + trial * 37;  // long accumulations: up to 4000 x 128 x 16 / 8 ~ 2^20
        hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, d_in, d_out, reps);
        std::vector<float> got(256);
        CHECK(hipMemcpy(got.data(), d_out, 256 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int col = l & 15, row = 4 * (l >> 4) + r;
                long want = 0;
                for (int g = 0; g < 4; ++g)
                    for (int j = 0; j < 32; ++j) want += (long)bits[(row + 16 * g) * 32 + j] * digits[(col + 16 * g) * 32 + j];
                want *= reps;
                if ((double)got[l * 4 + r] * 8.0 != (double)want) {
                    if (bad < 4 && bad_total < 12)
                        printf("trial %d row %d col %d: got %.3f x 8, want %ld\n", trial, row, col, got[l * 4 + r], want);
                    ++bad;
                }
            }
        bad_total += bad;
        CHECK(hipFree(d_in));
        CHECK(hipFree(d_out));
    }
    printf("exactness: %d wrong results over 64 trials of 256\n", bad_total);

    // ---- rate
    const int blocks = 256 * 2;
    std::vector<int> rnd(4096 + 4 * 2048);
    for (auto &x : rnd) x = rand() ^ (rand() << 16);
    int *d_in;
    float *d_out;
    CHECK(hipMalloc(&d_in, rnd.size() * 4));
    CHECK(hipMalloc(&d_out, blocks * 256 * 4));
    CHECK(hipMemcpy(d_in, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice));
    run<1>("int8 16x16x64, A bytes in {0,1}", d_in, d_out, iters, blocks);
    run<0>("FP4 {0,1} x FP6 digits 16x16x128", d_in, d_out, iters, blocks);
    run<2>("FP4 {0,1} x FP4 random 16x16x128", d_in, d_out, iters, blocks);
    run<3>("FP8 x FP8 random 16x16x128", d_in, d_out, iters, blocks);
    run<1>("int8 16x16x64, A bytes in {0,1}", d_in, d_out, iters, blocks);
    run<0>("FP4 {0,1} x FP6 digits 16x16x128", d_in, d_out, iters, blocks);
    return 0;
}
