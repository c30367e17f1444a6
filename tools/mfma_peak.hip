// Microbenchmark: what int8 MFMA rate does this MI355X sustain from registers alone?
// Bare v_mfma_i32_32x32x32_i8 / 16x16x64_i8 loops on random operands, no memory traffic in the
// loop.  Gives the practical ceiling (clock under load included) that the contraction kernel's
// roofline fraction can be read against.  Usage: mfma_peak [iterations]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

template <int SHAPE, int BINARY_A>
__global__ __launch_bounds__(256, 2) void mfma_loop(const v4i *in, int *out, int iters) {
    v4i a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256], b[4];
    for (int i = 0; i < 4; ++i) b[i] = in[threadIdx.x + 512 + 256 * i];
    if (BINARY_A == 1) {
        for (int i = 0; i < 4; ++i) {
            a0[i] &= 0x01010101;
            a1[i] &= 0x01010101;
        }
    }
    if (BINARY_A == 3) {  // bit i of a byte kept in place in fragment dword i: bytes in {0, 2^i}, i = 0..3
        for (int i = 0; i < 4; ++i) {
            a0[i] &= 0x01010101 << i;
            a1[i] &= 0x01010101 << i;
        }
    }
    if (BINARY_A == 4) {  // the same with i & 1: bytes in {0, 1} and {0, 2}
        for (int i = 0; i < 4; ++i) {
            a0[i] &= 0x01010101 << (i & 1);
            a1[i] &= 0x01010101 << (i & 1);
        }
    }
    if (BINARY_A == 2) {  // the binary operand in the B position instead
        for (int l = 0; l < 4; ++l)
            for (int i = 0; i < 4; ++i) b[l][i] &= 0x01010101;
    }
    if (SHAPE == 32) {
        v16i acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                acc[l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b[l], acc[l], 0, 0, 0);
                acc[4 + l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b[l], acc[4 + l], 0, 0, 0);
            }
        }
        int s = 0;
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        v4i acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b[l], acc[l], 0, 0, 0);
                acc[4 + l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b[l], acc[4 + l], 0, 0, 0);
            }
        }
        int s = 0;
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

template <int SHAPE, int BINARY_A>
static void run(const char *name, const v4i *d_in, int *d_out, int iters, int blocks) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((mfma_loop<SHAPE, BINARY_A>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double ops_per_mfma = SHAPE == 32 ? 2.0 * 32 * 32 * 32 : 2.0 * 16 * 16 * 64;
        const double ops = ops_per_mfma * 8.0 * iters * 4.0 * blocks;  // 8 MFMA per iteration, 4 waves per block
        printf("%-34s rep %d: %8.3f ms  %8.1f TOP/s\n", name, rep, ms, ops / ms / 1e9);
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int blocks = 256 * 2;
    std::vector<int> host(4 * 256 * 6);
    srand(1);
    for (auto &x : host) x = rand() ^ (rand() << 16);
    v4i *d_in;
    int *d_out;
    CHECK(hipMalloc(&d_in, host.size() * 4));
    CHECK(hipMalloc(&d_out, blocks * 256 * 4));
    CHECK(hipMemcpy(d_in, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    run<32, 0>("32x32x32 i8, random A and B", d_in, d_out, iters, blocks);
    run<32, 1>("32x32x32 i8, A bytes in {0,1}", d_in, d_out, iters, blocks);
    run<16, 0>("16x16x64 i8, random A and B", d_in, d_out, iters, blocks);
    run<16, 1>("16x16x64 i8, A bytes in {0,1}", d_in, d_out, iters, blocks);
    run<16, 3>("16x16x64 i8, A bytes in {0,2^i}, i<4", d_in, d_out, iters, blocks);
    run<16, 4>("16x16x64 i8, A bytes in {0,1},{0,2}", d_in, d_out, iters, blocks);
    run<16, 1>("16x16x64 i8, A bytes in {0,1} again", d_in, d_out, iters, blocks);
    run<32, 2>("32x32x32 i8, B bytes in {0,1}", d_in, d_out, iters, blocks);
    run<16, 2>("16x16x64 i8, B bytes in {0,1}", d_in, d_out, iters, blocks);
    return 0;
}
