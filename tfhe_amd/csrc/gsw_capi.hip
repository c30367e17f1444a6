// C ABI of the engine (include/gsw_hip.h): context, device memory helpers, argument checks and
// the batch schedulers in front of the gfx950 kernels.  No CPU fallback exists: every operation
// either runs its HIP kernel or returns an error.
#include "../../include/gsw_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "gsw_kernels.h"
#include "gsw_wire.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// No exception may cross the C ABI: the entry points that allocate host memory are function-try-blocks.
#define GSW_CATCH                                                                                     \
    catch (const std::bad_alloc &) { return fail(GSW_E_NOMEM, "host allocation failed"); }            \
    catch (const std::exception &e) { return fail(GSW_E_INVALID, "unexpected exception: %s", e.what()); }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? GSW_E_NOMEM : GSW_E_HIP, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                    \
    } while (0)

struct TimedLaunch {
    hipEvent_t start, stop;
};

// ROCTx ranges around the batch entry points, so that a rocprofv3 --marker-trace timeline shows which call a
// kernel belongs to.  Off unless GSW_ROCTX=1; the marker library is looked up at run time, so the engine does
// not depend on it.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *v = getenv("GSW_ROCTX");
        if (!v || atoi(v) == 0) return;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr;
                pop = nullptr;
            }
        }
    }
};
const Roctx &roctx() {
    static const Roctx r;
    return r;
}
struct ScopedRange {
    bool on;
    explicit ScopedRange(const char *name) : on(roctx().push != nullptr) {
        if (on) roctx().push(name);
    }
    ~ScopedRange() {
        if (on) roctx().pop();
    }
};

// Scratch of gsw_mul_gadget_batch_static: the prepared right operands of one chunk.  A file-scope device
// array: every GPU the library runs on holds its own copy, at its own address, until the process ends.
// In host code the name is only a handle for hipGetSymbolAddress -- never a device address: host code
// uses the per-device pointers resolved below.
__device__ __align__(256) unsigned char g_static_scratch[GSW_STATIC_SCRATCH_BYTES];

constexpr int kMaxDevices = 64;
std::mutex g_static_mutex;
void *g_static_ptr[kMaxDevices];  // device address of g_static_scratch on each GPU, resolved on first use
// Every context (and stream) of a GPU shares that GPU's array.  g_static_last[device] is recorded behind the last
// contraction of every gsw_mul_gadget_batch_static call and waited for by the next call's stream before its first
// preparation, so calls from different contexts queue up instead of overwriting each other's staged operands.
// Created with the pointer above (at gsw_ctx_create): the entry point itself creates nothing.
hipEvent_t g_static_last[kMaxDevices];

// Device address of the scratch array on `device`, which must be the current device.  Called by
// gsw_ctx_create, the library's set-up call on a device; resolved once per device.
int resolve_static_scratch(int device, void **out) {
    if (device < 0 || device >= kMaxDevices) return fail(GSW_E_INVALID, "device %d beyond the %d supported", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (!g_static_ptr[device]) {
        void *ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_static_scratch)));
        if (!ptr) return fail(GSW_E_HIP, "hipGetSymbolAddress returned a null address for the static scratch array");
        HIP_TRY(hipEventCreateWithFlags(&g_static_last[device], hipEventDisableTiming));
        g_static_ptr[device] = ptr;
    }
    *out = g_static_ptr[device];
    return GSW_OK;
}

}  // namespace

struct gsw_ctx {
    gswk::RingDesc ring;
    int n, m;
    int big_n;   // ell * (n + 1)
    int width;   // n + 1
    int ld;      // row pitch of compact ciphertexts: width rounded up to ld_align words
    int e4;      // width rounded up to 4: K elements of the contraction
    int words;   // ceil(m / ell): packed words per row of the encryption bit matrix
    int ld_r;    // words rounded up to 4
    int device;
    hipStream_t own_stream;
    hipStream_t stream;
    hipStream_t aux_stream;  // right-operand preparation of chunk k+1 runs here beside chunk k's contraction
    hipEvent_t ev_entry, ev_prep[2], ev_gemm[2];
    int overlap_chunk;       // products per pipelined chunk; 0 = one stream, workspace-sized chunks
    int pass_rows, pass_cols;  // workgroup placement passes (row tiles x column tiles)
    void *ws;
    size_t ws_bytes;
    size_t ws_limit;
    uint64_t n_prepared, n_products;  // right operands prepared / products contracted since creation
    int dense_fast;        // MFMA path of gsw_mul_dense / gsw_nand_dense: 0 never, 1 from 1024 rows up, 2 whenever the form fits
    int last_dense_path;   // what the last dense product used: 0 = plain dim^3 kernel, 1 = binary x digit contraction
                           // (flattened left operand), 2 = digit x digit products (arbitrary entries)
    int last_combine;      // recombination of the last contraction launch: 1 = 32-bit, 0 = 64-bit on every tile, -1 = none yet
    void *static_scratch;  // this GPU's copy of the library's static scratch array (gsw_mul_gadget_batch_static)
    bool in_static_call;   // inside gsw_mul_gadget_batch_static: nothing may allocate, not even timing events
    void *small;         // scratch of the helpers that return values to the host (row dots, tables, staging)
    size_t small_bytes;
    int gemm_variant;
    int packed_tail;  // how the last column tile of prepared operands is stored (gswk::tail_mode)
    bool timing;
    std::vector<TimedLaunch> timed[2];
    // slot lists of gsw_eval_circuit: pinned host staging, device copy, event marking the upload's end
    uint32_t *h_slots;
    uint32_t *d_slots;
    size_t slots_cap;
    hipEvent_t ev_slots;
};

namespace {

// scaled bits go with the default kernel only (the cross-check kernels read unscaled digits)
// (GSW_BIT_SCALE=0 in the environment turns them off for A/B timing and cross-checks)
int scale_mode_for(const gsw_ctx *ctx, int e4) {
    static const bool on = [] { const char *e = getenv("GSW_BIT_SCALE"); return !(e && e[0] == '0'); }();
    return on && (ctx->gemm_variant == 9 || ctx->gemm_variant == 12) ? gswk::bit_scale_mode(ctx->ring, e4) : 0;
}

int use_device(const gsw_ctx *ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    return GSW_OK;
}

int ensure_workspace(gsw_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return GSW_OK;
    if (ctx->ws) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
        HIP_TRY(hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    HIP_TRY(hipMalloc(&ctx->ws, bytes));
    ctx->ws_bytes = bytes;
    return GSW_OK;
}

// Scratch for the helpers around the hot path (decrypt's row dots, key vectors, sampler tables): kept by the
// context and grown on demand, so that a decrypt inside a chain loop allocates nothing after its first call.
// Every user enqueues on ctx->stream, which orders successive uses.
int ensure_small(gsw_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->small_bytes) return GSW_OK;
    if (ctx->small) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipFree(ctx->small));
        ctx->small = nullptr;
        ctx->small_bytes = 0;
    }
    size_t want = bytes < ((size_t)1 << 20) ? ((size_t)1 << 20) : bytes + bytes / 2;
    want = (want + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc(&ctx->small, want));
    ctx->small_bytes = want;
    return GSW_OK;
}

// Makes ctx->stream wait for everything enqueued on the auxiliary stream so far.  Error returns from
// the two-stream pipeline go through here: no preparation is left running unordered with later work.
void join_aux(gsw_ctx *ctx) {
    if (hipEventRecord(ctx->ev_entry, ctx->aux_stream) == hipSuccess)
        (void)hipStreamWaitEvent(ctx->stream, ctx->ev_entry, 0);
    else
        (void)hipStreamSynchronize(ctx->aux_stream);
}

struct ScopedTimer {
    gsw_ctx *ctx;
    int which;
    hipStream_t stream;
    TimedLaunch t{};
    bool on;
    ScopedTimer(gsw_ctx *c, int w, hipStream_t s) : ctx(c), which(w), stream(s), on(c->timing && !c->in_static_call) {
        if (on) {
            if (hipEventCreate(&t.start) != hipSuccess || hipEventCreate(&t.stop) != hipSuccess) on = false;
            else hipEventRecord(t.start, stream);
        }
    }
    ~ScopedTimer() {
        if (on) {
            hipEventRecord(t.stop, stream);
            try {
                ctx->timed[which].push_back(t);
            } catch (...) {  // out of host memory: this launch goes untimed
                (void)hipEventDestroy(t.start);
                (void)hipEventDestroy(t.stop);
            }
        }
    }
};

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }


// checks shared by the two contraction entry points
int check_gemm_common(const gsw_ctx *ctx, const void *lhs, size_t stride1, const void *out, size_t stride_out,
                      int batch, unsigned flags) {
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (batch < 0) return fail(GSW_E_INVALID, "negative batch");
    if (flags & ~GSW_FLAG_NAND) return fail(GSW_E_INVALID, "unknown flags 0x%x", flags);
    if (batch == 0) return GSW_OK;
    if (!lhs || !out) return fail(GSW_E_INVALID, "null operand");
    if (!aligned16(lhs) || (stride1 & 3u)) return fail(GSW_E_INVALID, "left operand must be 16-byte aligned with a stride multiple of 4");
    if (!aligned16(out)) return fail(GSW_E_INVALID, "output must be 16-byte aligned");
    if (stride_out < gsw_ctx_compact_elems(ctx) && batch > 1)
        return fail(GSW_E_INVALID, "output stride %zu smaller than one ciphertext (%zu)", stride_out,
                    gsw_ctx_compact_elems(ctx));
    return GSW_OK;
}

// Products are written tile by tile while other tiles still read the operands: the output may not share
// memory with an operand (the reference returns a fresh matrix, gsw_impl.rs:107).
int check_no_alias(const gsw_ctx *ctx, const uint32_t *operand, size_t stride, size_t operand_elems, const uint32_t *out,
                   size_t stride_out, int batch, const char *which) {
    const size_t elems = gsw_ctx_compact_elems(ctx);
    if (batch <= 0 || !operand || !out) return GSW_OK;
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(operand), b0 = reinterpret_cast<uintptr_t>(out);
    const uintptr_t a1 = a0 + 4 * ((size_t)(batch - 1) * stride + operand_elems);
    const uintptr_t b1 = b0 + 4 * ((size_t)(batch - 1) * stride_out + elems);
    if (a0 < b1 && b0 < a1) return fail(GSW_E_INVALID, "the output overlaps the %s operand", which);
    return GSW_OK;
}

int run_gemm(gsw_ctx *ctx, const gswk::GemmArgs &args) {
    gswk::GemmArgs a = args;
    ctx->n_products += (uint64_t)(a.batch > 0 ? a.batch : 0);
    a.pass_rows = ctx->pass_rows;
    a.pass_cols = ctx->pass_cols;
    ScopedTimer timer(ctx, 0, ctx->stream);
    HIP_TRY(gswk::launch_gemm(ctx->ring, a, ctx->gemm_variant, ctx->stream, &ctx->last_combine));
    return GSW_OK;
}

int run_prepare(gsw_ctx *ctx, const gswk::PrepArgs &a, hipStream_t stream) {
    ctx->n_prepared += (uint64_t)(a.batch > 0 ? a.batch : 0);
    ScopedTimer timer(ctx, 1, stream);
    HIP_TRY(gswk::launch_prepare(ctx->ring, a, stream));
    return GSW_OK;
}

// streams and events of a context; null handles (never created) are skipped
void destroy_handles(gsw_ctx *ctx) {
    if (ctx->ev_entry) (void)hipEventDestroy(ctx->ev_entry);
    if (ctx->ev_slots) (void)hipEventDestroy(ctx->ev_slots);
    for (int i = 0; i < 2; ++i) {
        if (ctx->ev_prep[i]) (void)hipEventDestroy(ctx->ev_prep[i]);
        if (ctx->ev_gemm[i]) (void)hipEventDestroy(ctx->ev_gemm[i]);
    }
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
}

}  // namespace

extern "C" {

const char *gsw_last_error(void) { return g_last_error.c_str(); }
const char *gsw_version(void) { return "gsw-mi355x 0.1 (gfx950)"; }

int gsw_ctx_create(gsw_ctx **out, int ring_kind, int log2q, int n, int m, int device) try {
    if (!out) return fail(GSW_E_INVALID, "null output pointer");
    *out = nullptr;
    gswk::RingDesc ring{};
    if (ring_kind == GSW_RING_FP_M31) {
        ring = {gswk::kRingFp, 31, gswk::kFpP};  // field.rs:8,39
    } else if (ring_kind == GSW_RING_ZPOW2) {
        if (log2q < 1 || log2q > 32)
            return fail(GSW_E_UNSUPPORTED, "Zpow2<%d>: only 1 <= M <= 32 is supported", log2q);
        ring = {gswk::kRingZpow2, log2q, log2q == 32 ? 0xFFFFFFFFu : ((1u << log2q) - 1u)};  // pow2_ring.rs:39-48
    } else {
        return fail(GSW_E_UNSUPPORTED, "unknown ring kind %d", ring_kind);
    }
    if (n < 1 || m < 1) return fail(GSW_E_INVALID, "n and m must be positive (n=%d, m=%d)", n, m);
    if ((long)ring.ell * (n + 1) > (1L << 20)) return fail(GSW_E_INVALID, "n=%d too large", n);
    if (m > (1 << 24)) return fail(GSW_E_INVALID, "m=%d too large", m);
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(GSW_E_INVALID, "device %d out of range (%d visible)", device, count);
    HIP_TRY(hipSetDevice(device));
    gsw_ctx *ctx = new gsw_ctx();
    ctx->ring = ring;
    ctx->n = n;
    ctx->m = m;
    ctx->big_n = ring.ell * (n + 1);
    ctx->width = n + 1;
    ctx->e4 = (n + 1 + 3) & ~3;
    {
        // 64-byte row pitch: the contraction's 64-byte left-operand pieces then never straddle a
        // 64-byte boundary (1 % faster at n = 256 than a 16-byte pitch, for 4.6 % more memory)
        int align = 16;
        if (const char *v = getenv("GSW_LD_ALIGN")) align = atoi(v);
        if (align != 4 && align != 8 && align != 16 && align != 32) align = 16;
        ctx->ld = (n + 1 + align - 1) / align * align;
    }
    // element indices inside one compact ciphertext stay below 2^31: the contraction's fused digest weighs
    // element i with the 32-bit value 2 i + 1, which equals the 64-bit weight of gsw_digest_batch only there
    if ((uint64_t)ctx->big_n * (uint64_t)ctx->ld >= (1ull << 31)) {
        delete ctx;
        return fail(GSW_E_INVALID, "n=%d too large: a compact ciphertext must hold fewer than 2^31 words", n);
    }
    ctx->words = (m + ring.ell - 1) / ring.ell;
    ctx->ld_r = (ctx->words + 3) & ~3;
    ctx->device = device;
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    ctx->ws_limit = (size_t)2 << 30;
    // 16x16x64 MFMA, 256-row tiles, 64-byte left-operand loads, software-pipelined instruction stream
    ctx->gemm_variant = 9;
    ctx->timing = false;
    if (const char *v = getenv("GSW_GEMM_VARIANT")) ctx->gemm_variant = atoi(v);
#ifdef GSW_DIAGNOSTIC_BUILDS
    // superseded kernels 0..8 and timing-only builds 10.. (wrong results by design) exist in this library
    bool known = ctx->gemm_variant >= 0 && ctx->gemm_variant < 138;
#else
    bool known = ctx->gemm_variant == 9 || ctx->gemm_variant == 2 || ctx->gemm_variant == 3;
#endif
#ifdef GSW_EXP_TALL
    known = known || ctx->gemm_variant == 10 || ctx->gemm_variant == 11;
#endif
#ifdef GSW_EXP_WIDE8
    known = known || ctx->gemm_variant == 12;
#endif
    if (!known) {
        delete ctx;
        return fail(GSW_E_INVALID, "GSW_GEMM_VARIANT must be 9 (default), 2 or 3 (simple cross-check kernels); the "
                                   "superseded kernels exist only in a -DGSW_DIAGNOSTIC_BUILDS library");
    }
    ctx->packed_tail = (ctx->gemm_variant == 2 || ctx->gemm_variant == 3) ? gswk::kTailNone
                                                                          : gswk::tail_mode(ctx->width, ctx->gemm_variant == 9 || ctx->gemm_variant == 12);
    if (int rc = resolve_static_scratch(device, &ctx->static_scratch)) {  // `device` is current (hipSetDevice above)
        delete ctx;
        return rc;
    }
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_entry, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_slots, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&ctx->ev_prep[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_gemm[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        destroy_handles(ctx);  // whatever was created before the failure
        delete ctx;
        return fail(GSW_E_HIP, "stream/event creation failed: %s", hipGetErrorString(e));
    }
    ctx->stream = ctx->own_stream;
    ctx->dense_fast = 1;
    ctx->last_combine = -1;
    ctx->overlap_chunk = 128;
    // placement passes (row tiles x column tiles an XCD works on together): 8 x 8, and 16 x 4 once a product has
    // a dozen column tiles (n = 512: a right-operand panel is 2.1 MB there, and taller passes re-read it half
    // as often; +1..2 % on the kernel, level at n = 256)
    const bool wide = gswk::col_tiles(ctx->width) >= 12;
    ctx->pass_rows = wide ? 16 : 8;
    ctx->pass_cols = wide ? 4 : 8;
    if (const char *v = getenv("GSW_PASS_ROWS")) ctx->pass_rows = atoi(v) > 0 ? atoi(v) : ctx->pass_rows;
    if (const char *v = getenv("GSW_PASS_COLS")) ctx->pass_cols = atoi(v) > 0 ? atoi(v) : ctx->pass_cols;
    if (const char *v = getenv("GSW_OVERLAP_CHUNK")) ctx->overlap_chunk = atoi(v) > 0 ? atoi(v) : 0;
    *out = ctx;
    return GSW_OK;
}
GSW_CATCH

void gsw_ctx_destroy(gsw_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->aux_stream);
    for (auto &v : ctx->timed)
        for (auto &t : v) {
            (void)hipEventDestroy(t.start);
            (void)hipEventDestroy(t.stop);
        }
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->small) (void)hipFree(ctx->small);
    if (ctx->d_slots) (void)hipFree(ctx->d_slots);
    if (ctx->h_slots) (void)hipHostFree(ctx->h_slots);
    destroy_handles(ctx);
    delete ctx;
}

int gsw_ctx_set_stream(gsw_ctx *ctx, void *hip_stream) {
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return GSW_OK;
}

int gsw_sync(gsw_ctx *ctx) {
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GSW_OK;
}

int gsw_ctx_ell(const gsw_ctx *ctx) { return ctx ? ctx->ring.ell : 0; }
int gsw_ctx_big_n(const gsw_ctx *ctx) { return ctx ? ctx->big_n : 0; }
int gsw_ctx_ld(const gsw_ctx *ctx) { return ctx ? ctx->ld : 0; }
int gsw_ctx_gemm_variant(const gsw_ctx *ctx) { return ctx ? ctx->gemm_variant : -1; }
size_t gsw_ctx_compact_elems(const gsw_ctx *ctx) { return ctx ? (size_t)ctx->big_n * ctx->ld : 0; }
size_t gsw_ctx_prepared_bytes(const gsw_ctx *ctx) { return ctx ? gswk::prepared_bytes(ctx->width, ctx->e4, ctx->packed_tail) : 0; }
size_t gsw_ctx_rbits_elems(const gsw_ctx *ctx) { return ctx ? (size_t)ctx->big_n * ctx->ld_r : 0; }
size_t gsw_ctx_pk_prepared_bytes(const gsw_ctx *ctx) { return ctx ? gswk::prepared_bytes(ctx->width, ctx->ld_r, ctx->packed_tail) : 0; }

int gsw_ctx_set_workspace_limit(gsw_ctx *ctx, size_t bytes) {
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (bytes < gsw_ctx_prepared_bytes(ctx))
        return fail(GSW_E_INVALID, "workspace limit below one prepared operand (%zu bytes)", gsw_ctx_prepared_bytes(ctx));
    ctx->ws_limit = bytes;
    return GSW_OK;
}

int gsw_ctx_enable_timing(gsw_ctx *ctx, int enable) {
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    ctx->timing = enable != 0;
    return GSW_OK;
}

int gsw_ctx_counters(const gsw_ctx *ctx, uint64_t *prepared_operands, uint64_t *contracted_products) {
    if (!ctx || !prepared_operands || !contracted_products) return fail(GSW_E_INVALID, "bad argument");
    *prepared_operands = ctx->n_prepared;
    *contracted_products = ctx->n_products;
    return GSW_OK;
}

int gsw_ctx_reset_timing(gsw_ctx *ctx) {
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto &v : ctx->timed) {
        for (auto &t : v) {
            (void)hipEventDestroy(t.start);
            (void)hipEventDestroy(t.stop);
        }
        v.clear();
    }
    return GSW_OK;
}

int gsw_ctx_kernel_time(gsw_ctx *ctx, int which, double *total_ms, long *launches) try {
    if (!ctx || which < 0 || which > 1 || !total_ms || !launches) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
    double total = 0.0;
    for (auto &t : ctx->timed[which]) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, t.start, t.stop));
        total += ms;
    }
    *total_ms = total;
    *launches = (long)ctx->timed[which].size();
    return GSW_OK;
}
GSW_CATCH

// ---- device memory helpers ----

int gsw_malloc(gsw_ctx *ctx, size_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    *d_ptr = nullptr;
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 16));
    return GSW_OK;
}

int gsw_free(gsw_ctx *ctx, void *d_ptr) {
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(d_ptr));
    return GSW_OK;
}

int gsw_memcpy_h2d(gsw_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GSW_OK;
}

int gsw_memcpy_d2h(gsw_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GSW_OK;
}

int gsw_memset_zero(gsw_ctx *ctx, void *d_dst, size_t bytes) {
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipMemsetAsync(d_dst, 0, bytes, ctx->stream));
    return GSW_OK;
}

int gsw_upload_compact(gsw_ctx *ctx, uint32_t *d_dst, const uint32_t *h_src) {
    if (!ctx || !d_dst || !h_src) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipMemsetAsync(d_dst, 0, gsw_ctx_compact_elems(ctx) * 4, ctx->stream));
    HIP_TRY(hipMemcpy2DAsync(d_dst, (size_t)ctx->ld * 4, h_src, (size_t)ctx->width * 4, (size_t)ctx->width * 4,
                             (size_t)ctx->big_n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GSW_OK;
}

int gsw_download_compact(gsw_ctx *ctx, uint32_t *h_dst, const uint32_t *d_src) {
    if (!ctx || !h_dst || !d_src) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipMemcpy2DAsync(h_dst, (size_t)ctx->width * 4, d_src, (size_t)ctx->ld * 4, (size_t)ctx->width * 4,
                             (size_t)ctx->big_n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GSW_OK;
}

// ---- wire format (the parser and writer are host code of their own: gsw_wire.cpp) ----

namespace {

using gswwire::kHeaderBytes;

gswwire::Params wire_params(const gsw_ctx *ctx) {
    gswwire::Params p{};
    p.ring_kind = ctx->ring.kind == gswk::kRingFp ? GSW_RING_FP_M31 : GSW_RING_ZPOW2;
    p.bits = ctx->ring.ell;
    p.mask = ctx->ring.mask;
    p.n = ctx->n;
    p.m = ctx->m;
    p.big_n = (uint32_t)ctx->big_n;
    p.width = (uint32_t)ctx->width;
    return p;
}

int wire_check(const gsw_ctx *ctx, int kind, const uint8_t *buf, size_t bytes, uint32_t *count) {
    std::string err;
    if (!gswwire::check(wire_params(ctx), kind, buf, bytes, count, err)) return fail(GSW_E_INVALID, "%s", err.c_str());
    return GSW_OK;
}

}  // namespace

size_t gsw_wire_bytes(const gsw_ctx *ctx, int kind, int count) {
    if (!ctx || count < 0) return 0;
    return gswwire::object_bytes(wire_params(ctx), kind, (size_t)count);
}

int gsw_wire_write_host(gsw_ctx *ctx, int kind, const uint32_t *h_values, void *h_out, size_t out_bytes) {
    gswwire::Shape sh{};
    if (!ctx || !h_values || !h_out) return fail(GSW_E_INVALID, "bad argument");
    const gswwire::Params wp = wire_params(ctx);
    if ((kind != GSW_WIRE_PUBLIC_KEY && kind != GSW_WIRE_SECRET_KEY) || !gswwire::shape_of(wp, kind, sh))
        return fail(GSW_E_INVALID, "kind %d is not a host-resident object", kind);
    if (out_bytes != gsw_wire_bytes(ctx, kind, 1)) return fail(GSW_E_INVALID, "output buffer must be %zu bytes", gsw_wire_bytes(ctx, kind, 1));
    uint8_t *buf = (uint8_t *)h_out;
    const size_t elems = (size_t)sh.rows * sh.cols;
    gswwire::write_header(wp, kind, sh, 1, buf);
    for (size_t i = 0; i < elems; ++i) {
        if (!gswwire::canonical(wp, h_values[i]))
            return fail(GSW_E_INVALID, "element %zu (%u) is not a canonical ring value", i, h_values[i]);
        gswwire::put_u32(buf + kHeaderBytes + 4 * i, h_values[i]);
    }
    gswwire::write_trailer(buf, elems);
    return GSW_OK;
}

int gsw_wire_read_host(gsw_ctx *ctx, int kind, const void *h_in, size_t in_bytes, uint32_t *h_values) {
    if (!ctx || !h_in || !h_values) return fail(GSW_E_INVALID, "bad argument");
    if (kind != GSW_WIRE_PUBLIC_KEY && kind != GSW_WIRE_SECRET_KEY)
        return fail(GSW_E_INVALID, "kind %d is not a host-resident object", kind);
    uint32_t count = 0;
    if (int rc = wire_check(ctx, kind, (const uint8_t *)h_in, in_bytes, &count)) return rc;
    gswwire::Shape sh{};
    gswwire::shape_of(wire_params(ctx), kind, sh);
    const uint8_t *payload = (const uint8_t *)h_in + kHeaderBytes;
    for (size_t i = 0; i < (size_t)sh.rows * sh.cols; ++i) h_values[i] = gswwire::get_u32(payload + 4 * i);
    return GSW_OK;
}

int gsw_wire_write_compact(gsw_ctx *ctx, const uint32_t *d_cc, size_t stride, int batch, void *h_out,
                           size_t out_bytes) {
    if (!ctx || !d_cc || !h_out || batch < 0) return fail(GSW_E_INVALID, "bad argument");
    if (stride < gsw_ctx_compact_elems(ctx) && batch > 1) return fail(GSW_E_INVALID, "stride smaller than one ciphertext");
    if (out_bytes != gsw_wire_bytes(ctx, GSW_WIRE_CIPHERTEXT, batch))
        return fail(GSW_E_INVALID, "output buffer must be %zu bytes", gsw_wire_bytes(ctx, GSW_WIRE_CIPHERTEXT, batch));
    if (int rc = use_device(ctx)) return rc;
    gswwire::Shape sh{};
    const gswwire::Params wp = wire_params(ctx);
    gswwire::shape_of(wp, GSW_WIRE_CIPHERTEXT, sh);
    uint8_t *buf = (uint8_t *)h_out;
    gswwire::write_header(wp, GSW_WIRE_CIPHERTEXT, sh, (uint32_t)batch, buf);
    const size_t per = (size_t)sh.rows * sh.cols;
    // the payload is little-endian u32, which is the host's own representation on every ROCm platform
    for (int b = 0; b < batch; ++b)
        HIP_TRY(hipMemcpy2DAsync(buf + kHeaderBytes + 4 * per * b, (size_t)ctx->width * 4, d_cc + (size_t)b * stride,
                                 (size_t)ctx->ld * 4, (size_t)ctx->width * 4, (size_t)ctx->big_n,
                                 hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    gswwire::write_trailer(buf, per * batch);
    return GSW_OK;
}

int gsw_wire_read_compact(gsw_ctx *ctx, const void *h_in, size_t in_bytes, uint32_t *d_cc, size_t stride,
                          int max_batch, int *batch_out) {
    if (!ctx || !h_in || !batch_out || max_batch < 0) return fail(GSW_E_INVALID, "bad argument");
    uint32_t count = 0;
    if (int rc = wire_check(ctx, GSW_WIRE_CIPHERTEXT, (const uint8_t *)h_in, in_bytes, &count)) return rc;
    *batch_out = (int)count;
    if (!d_cc) return GSW_OK;  // header query: how many entries to make room for
    if (count > (uint32_t)max_batch) return fail(GSW_E_INVALID, "wire object holds %u ciphertexts, room for %d", count, max_batch);
    if (stride < gsw_ctx_compact_elems(ctx) && count > 1) return fail(GSW_E_INVALID, "stride smaller than one ciphertext");
    if (int rc = use_device(ctx)) return rc;
    const size_t per = (size_t)ctx->big_n * ctx->width;
    const uint8_t *payload = (const uint8_t *)h_in + kHeaderBytes;
    for (uint32_t b = 0; b < count; ++b) {
        uint32_t *dst = d_cc + (size_t)b * stride;
        HIP_TRY(hipMemsetAsync(dst, 0, gsw_ctx_compact_elems(ctx) * 4, ctx->stream));
        HIP_TRY(hipMemcpy2DAsync(dst, (size_t)ctx->ld * 4, payload + 4 * per * b, (size_t)ctx->width * 4,
                                 (size_t)ctx->width * 4, (size_t)ctx->big_n, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GSW_OK;
}

// ---- gadget helpers ----

int gsw_bit_decomp(gsw_ctx *ctx, const uint32_t *d_in, int rows, int cols, int ld_in, uint32_t *d_out) {
    if (!ctx || !d_in || !d_out || rows < 0 || cols < 0 || ld_in < cols) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(gswk::launch_bit_decomp(ctx->ring, d_in, rows, cols, ld_in, d_out, gswk::kDecompU32, ctx->stream));
    return GSW_OK;
}

size_t gsw_bit_decomp_row_elems(const gsw_ctx *ctx, int cols, int format) {
    if (!ctx || cols < 0 || format < GSW_DECOMP_U32 || format > GSW_DECOMP_BITS) return 0;
    return gswk::bit_decomp_pitch(cols, ctx->ring.ell, format);
}

int gsw_bit_decomp_as(gsw_ctx *ctx, const uint32_t *d_in, int rows, int cols, int ld_in, void *d_out, int format) {
    if (!ctx || !d_in || !d_out || rows < 0 || cols < 0 || ld_in < cols) return fail(GSW_E_INVALID, "bad argument");
    if (format < GSW_DECOMP_U32 || format > GSW_DECOMP_BITS) return fail(GSW_E_INVALID, "unknown decomposition format %d", format);
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(gswk::launch_bit_decomp(ctx->ring, d_in, rows, cols, ld_in, d_out, format, ctx->stream));
    return GSW_OK;
}

int gsw_bit_decomp_inv(gsw_ctx *ctx, const uint32_t *d_in, int rows, int cols, uint32_t *d_out, int ld_out,
                       int quirk) {
    if (!ctx || !d_in || !d_out || rows < 0 || cols < 0 || ld_out < cols) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(gswk::launch_bit_decomp_inv(ctx->ring, d_in, rows, cols, d_out, ld_out, quirk, ctx->stream));
    return GSW_OK;
}

int gsw_flatten(gsw_ctx *ctx, uint32_t *d_mat, int rows, int cols) {
    if (!ctx || !d_mat || rows < 0 || cols < 0) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    // helper.rs:51-54: bit_decomp_inv then bit_decomp, through a rows x cols scratch matrix
    if (int rc = ensure_small(ctx, (size_t)rows * cols * 4 + 16)) return rc;
    uint32_t *tmp = (uint32_t *)ctx->small;
    HIP_TRY(gswk::launch_bit_decomp_inv(ctx->ring, d_mat, rows, cols, tmp, cols, 1, ctx->stream));
    HIP_TRY(gswk::launch_bit_decomp(ctx->ring, tmp, rows, cols, cols, d_mat, gswk::kDecompU32, ctx->stream));
    return GSW_OK;
}

// ---- elementwise and dense operations ----

int gsw_add(gsw_ctx *ctx, const uint32_t *d_a, const uint32_t *d_b, uint32_t *d_out, size_t count) {
    if (!ctx || !d_a || !d_b || !d_out) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(gswk::launch_add(ctx->ring, d_a, d_b, d_out, count, ctx->stream));
    return GSW_OK;
}

int gsw_mult_const(gsw_ctx *ctx, uint32_t *d_inout, uint32_t constant, size_t count) {
    if (!ctx || !d_inout) return fail(GSW_E_INVALID, "bad argument");
    if (ctx->ring.kind == gswk::kRingFp ? constant >= gswk::kFpP : (constant & ~ctx->ring.mask) != 0)
        return fail(GSW_E_INVALID, "constant %u is not a canonical ring element", constant);
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(gswk::launch_mult_const(ctx->ring, d_inout, constant, count, ctx->stream));
    return GSW_OK;
}

// mult / nand on the reference's own N x N matrices.  A left operand in the reference's ciphertext format
// (entries 0 / 1 off the diagonal: BitDecomp(R * A) + mu * I, gsw_impl.rs:27-42) goes through the int8-MFMA
// contraction with an N-column right operand plus a diagonal term; anything else -- arbitrary entries, a
// dimension other than the context's N -- through sixteen signed-digit int8 products on MFMA; small matrices, or
// operands whose digit planes exceed the workspace limit, through the plain dim^3 kernel.  Same results every way
// (all are the exact product over the ring).
static int dense_product(gsw_ctx *ctx, const uint32_t *d_c1, const uint32_t *d_c2, uint32_t *d_out, int dim, int nand) {
    ScopedRange range("gsw_mul_dense");
    if (!ctx || !d_c1 || !d_c2 || !d_out || dim < 0) return fail(GSW_E_INVALID, "bad argument");
    if (d_out == d_c1 || d_out == d_c2) return fail(GSW_E_INVALID, "output must not alias an input");
    if (int rc = use_device(ctx)) return rc;
    ctx->last_dense_path = 0;
    const int packed_tail = ctx->gemm_variant == 9 ? gswk::tail_mode(dim, true) : gswk::kTailNone;
    const size_t prep = gswk::prepared_bytes(dim, ctx->e4, packed_tail);
    // below about a thousand rows the detection pass and its host synchronisation cost more than the plain
    // kernel takes (N = 341: 0.06 ms against 0.02 ms); dense_fast == 2 takes the MFMA path at any size
    const bool worth_it = ctx->dense_fast == 2 || (ctx->dense_fast == 1 && dim >= 1024);
    const bool shape_fits = worth_it && ctx->gemm_variant == 9 && dim == ctx->big_n && dim > 0 && prep <= ctx->ws_limit;
    if (shape_fits) {
        // packed left operand (N x ld words), diagonal vector, misfit counter: from the context's scratch
        const size_t packed_bytes = ((size_t)dim * ctx->ld * 4 + 255) & ~(size_t)255;
        const size_t diag_bytes = ((size_t)dim * 4 + 255) & ~(size_t)255;
        const size_t rows_bytes = (((size_t)dim / 256 + 2) * 4 + 255) & ~(size_t)255;
        if (int rc = ensure_small(ctx, packed_bytes + diag_bytes + 256 + rows_bytes)) return rc;
        uint32_t *packed = (uint32_t *)ctx->small;
        uint32_t *diag = (uint32_t *)((uint8_t *)ctx->small + packed_bytes);
        unsigned long long *d_misfits = (unsigned long long *)((uint8_t *)ctx->small + packed_bytes + diag_bytes);
        uint32_t *d_rows = (uint32_t *)((uint8_t *)ctx->small + packed_bytes + diag_bytes + 256);
        HIP_TRY(gswk::launch_pack_flat_lhs(ctx->ring, d_c1, dim, ctx->width, ctx->ld, packed, diag, d_misfits, ctx->stream));
        // first rows of the slabs the product is cut into (see below)
        // the tallest slab that still gives every XCD an entry: the row tiles of a slab share each right-operand panel
        // (N = 15 903: 11.7 ms with slabs of 2048 rows against 12.2 ms with 512)
        int slab = 256;
        for (int cand = 512; cand <= 2048 && (dim + cand - 1) / cand >= 8; cand *= 2) slab = cand;
        const int entries = (dim + slab - 1) / slab;
        std::vector<uint32_t> h_rows((size_t)entries, 0u);
        for (int e = 0; e < entries; ++e) h_rows[e] = (uint32_t)(e * slab + slab <= dim ? e * slab : (dim > slab ? dim - slab : 0));
        HIP_TRY(hipMemcpyAsync(d_rows, h_rows.data(), h_rows.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        unsigned long long counts[2] = {0, 0};  // misfits, rows with a diagonal term
        HIP_TRY(hipMemcpyAsync(counts, d_misfits, sizeof counts, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (counts[0] == 0) {
            if (int rc = ensure_workspace(ctx, prep)) return rc;
            gswk::PrepArgs pa{};
            pa.src = d_c2;
            pa.src_stride = 0;
            pa.dst = (uint8_t *)ctx->ws;
            pa.dst_stride = prep;
            pa.rows_valid = dim;
            pa.cols_valid = dim;
            pa.ld_src = dim;
            pa.elems = ctx->width;
            pa.e4 = ctx->e4;
            pa.batch = 1;
            pa.packed_tail = packed_tail;
            pa.layout16 = 1;
            pa.scale_mode = scale_mode_for(ctx, pa.e4);
            // The kernel places one batch entry per XCD at a time, so the single product is cut into slabs of
            // `slab` rows that play the part of batch entries sharing one right operand.  Slot lists with a
            // stride of one row let every entry start at any row: the last slab is moved up to end at row N
            // and recomputes the rows it shares with its neighbour (the same values, stored twice).
            if (int rc = run_prepare(ctx, pa, ctx->stream)) return rc;
            gswk::GemmArgs a{};
            a.lhs = packed;
            a.lhs_stride = (size_t)ctx->ld;
            a.rhs = (const uint8_t *)ctx->ws;
            a.rhs_stride = 0;
            a.out = d_out;
            a.out_stride = (size_t)dim;
            a.rows = dim < slab ? dim : slab;
            a.e4 = ctx->e4;
    a.k_words = ctx->width;
            a.ld_lhs = ctx->ld;
            a.out_cols = dim;
            a.ld_out = dim;
            a.batch = entries;
            a.mode = gswk::kModeMul;
            a.packed_tail = packed_tail;
            a.scale_mode = scale_mode_for(ctx, a.e4);
            a.lhs_slot = d_rows;
            a.out_slot = d_rows;
            if (int rc = run_gemm(ctx, a)) return rc;
            // the diagonal term diag(d) * C2 and nand's I - P (gsw_impl.rs:114-123), when there is either
            if (counts[1] != 0 || nand)
                HIP_TRY(gswk::launch_dense_fixup(ctx->ring, d_out, d_c2, diag, dim, nand, ctx->stream));
            ctx->last_dense_path = 1;
            return GSW_OK;
        }
    }
    // Any other left operand: sixteen signed-digit int8 products on MFMA (dense_limb_gemm_kernel), when the digit
    // planes of the right operand and of at least 64 rows of the left one fit the workspace; the left operand
    // goes through in row slabs.
    if (worth_it && dim > 0 && dim < 32768) {
        const size_t pad = (size_t)gswk::dense_limb_pad(dim);
        const size_t rhs_bytes = 4 * pad * pad;
        if (rhs_bytes + 4 * pad * 64 <= ctx->ws_limit) {
            size_t slab = (ctx->ws_limit - rhs_bytes) / (4 * pad) / 64 * 64;  // rows of the left operand per pass
            if (slab >= pad) slab = pad;
            else if (slab >= 4096) slab = slab / 4096 * 4096;  // whole placement groups of 64 row tiles
            if (int rc = ensure_workspace(ctx, rhs_bytes + 4 * pad * slab)) return rc;
            uint8_t *rhs_planes = (uint8_t *)ctx->ws, *lhs_planes = rhs_planes + rhs_bytes;
            ctx->n_prepared += 1;
            {
                ScopedTimer timer(ctx, 1, ctx->stream);
                HIP_TRY(gswk::launch_dense_split_transposed(ctx->ring, d_c2, dim, rhs_planes, ctx->stream));
            }
            for (size_t r0 = 0; r0 < (size_t)dim; r0 += slab) {
                const int rows = (int)((size_t)dim - r0 < slab ? (size_t)dim - r0 : slab);
                HIP_TRY(gswk::launch_dense_split_rows(ctx->ring, d_c1, dim, (int)r0, rows, lhs_planes, ctx->stream));
                ScopedTimer timer(ctx, 0, ctx->stream);
                HIP_TRY(gswk::launch_dense_limb_gemm(ctx->ring, lhs_planes, rhs_planes, d_out, dim, (int)r0, rows, nand, ctx->stream));
            }
            ctx->last_dense_path = 2;
            return GSW_OK;
        }
    }
    HIP_TRY(gswk::launch_mul_dense(ctx->ring, d_c1, d_c2, d_out, dim, nand, ctx->stream));
    return GSW_OK;
}

int gsw_mul_dense(gsw_ctx *ctx, const uint32_t *d_c1, const uint32_t *d_c2, uint32_t *d_out, int dim) {
    return dense_product(ctx, d_c1, d_c2, d_out, dim, 0);
}

int gsw_nand_dense(gsw_ctx *ctx, const uint32_t *d_c1, const uint32_t *d_c2, uint32_t *d_out, int dim) {
    return dense_product(ctx, d_c1, d_c2, d_out, dim, 1);
}

int gsw_ctx_set_dense_fast(gsw_ctx *ctx, int enable) {
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (enable < 0 || enable > 2) return fail(GSW_E_INVALID, "dense path mode must be 0, 1 or 2");
    ctx->dense_fast = enable;
    return GSW_OK;
}

int gsw_ctx_last_dense_path(const gsw_ctx *ctx) { return ctx ? ctx->last_dense_path : -1; }
int gsw_ctx_last_combine(const gsw_ctx *ctx) { return ctx ? ctx->last_combine : -1; }

// ---- the hot path ----

static int prepare_rhs_on(gsw_ctx *ctx, const uint32_t *d_c2c, size_t stride_c2c, void *d_prepared, int batch,
                          hipStream_t stream, const uint32_t *d_slots = nullptr, const uint32_t *d_dst_slots = nullptr) {
    if (!ctx || batch < 0) return fail(GSW_E_INVALID, "bad argument");
    if (batch == 0) return GSW_OK;
    if (!d_c2c || !d_prepared || !aligned16(d_prepared)) return fail(GSW_E_INVALID, "null or unaligned operand");
    if (int rc = use_device(ctx)) return rc;
    gswk::PrepArgs a{};
    a.src = d_c2c;
    a.src_stride = stride_c2c;
    a.dst = (uint8_t *)d_prepared;
    a.dst_stride = gsw_ctx_prepared_bytes(ctx);
    a.rows_valid = ctx->big_n;
    a.cols_valid = ctx->width;
    a.ld_src = ctx->ld;
    a.elems = ctx->width;
    a.e4 = ctx->e4;
    a.batch = batch;
    a.packed_tail = ctx->packed_tail;
    a.layout16 = ctx->gemm_variant >= 4;
    a.scale_mode = scale_mode_for(ctx, a.e4);
    a.src_slot = d_slots;
    a.dst_slot = d_dst_slots;
    return run_prepare(ctx, a, stream);
}

int gsw_validate_compact(gsw_ctx *ctx, const uint32_t *d_cc, size_t stride, int batch, uint64_t *h_bad_out) {
    if (!ctx || !h_bad_out || batch < 0 || (batch > 0 && !d_cc)) return fail(GSW_E_INVALID, "bad argument");
    *h_bad_out = 0;
    if (batch == 0) return GSW_OK;
    if (stride < gsw_ctx_compact_elems(ctx) && batch > 1) return fail(GSW_E_INVALID, "stride smaller than one ciphertext");
    if (int rc = use_device(ctx)) return rc;
    if (int rc = ensure_small(ctx, 256)) return rc;
    unsigned long long *d_bad = (unsigned long long *)ctx->small;
    HIP_TRY(gswk::launch_validate(ctx->ring, d_cc, stride, ctx->big_n, ctx->width, ctx->ld, batch, d_bad, ctx->stream));
    unsigned long long bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *h_bad_out = bad;
    return GSW_OK;
}

int gsw_prepare_rhs(gsw_ctx *ctx, const uint32_t *d_c2c, size_t stride_c2c, void *d_prepared, int batch) {
    if (!ctx) return fail(GSW_E_INVALID, "bad argument");
    return prepare_rhs_on(ctx, d_c2c, stride_c2c, d_prepared, batch, ctx->stream);
}

static int mul_prepared(gsw_ctx *ctx, const uint32_t *d_c1c, size_t stride1, const void *d_prepared,
                        size_t stride_prepared_bytes, uint32_t *d_out, size_t stride_out, int batch, unsigned flags,
                        uint64_t *d_digests, const uint32_t *d_lhs_slots = nullptr,
                        const uint32_t *d_out_slots = nullptr, const uint32_t *d_rhs_slots = nullptr) {
    if (int rc = check_gemm_common(ctx, d_c1c, stride1, d_out, stride_out, batch, flags)) return rc;
    if (batch == 0) return GSW_OK;
    if (!d_prepared || !aligned16(d_prepared) || (stride_prepared_bytes & 15u))
        return fail(GSW_E_INVALID, "prepared operand must be 16-byte aligned");
    if (int rc = use_device(ctx)) return rc;
    gswk::GemmArgs a{};
    a.lhs = d_c1c;
    a.lhs_stride = stride1;
    a.rhs = (const uint8_t *)d_prepared;
    a.rhs_stride = stride_prepared_bytes;
    a.out = d_out;
    a.out_stride = stride_out;
    a.addend = nullptr;
    a.rows = ctx->big_n;
    a.e4 = ctx->e4;
    a.k_words = ctx->width;
    a.ld_lhs = ctx->ld;
    a.out_cols = ctx->width;
    a.ld_out = ctx->ld;
    a.batch = batch;
    a.mode = (flags & GSW_FLAG_NAND) ? gswk::kModeNand : gswk::kModeMul;
    a.packed_tail = ctx->packed_tail;
    a.scale_mode = scale_mode_for(ctx, a.e4);
    a.digests = (unsigned long long *)d_digests;
    a.lhs_slot = d_lhs_slots;
    a.out_slot = d_out_slots;
    a.rhs_slot = d_rhs_slots;
    return run_gemm(ctx, a);
}

int gsw_mul_gadget_prepared(gsw_ctx *ctx, const uint32_t *d_c1c, size_t stride1, const void *d_prepared,
                            size_t stride_prepared_bytes, uint32_t *d_out, size_t stride_out, int batch,
                            unsigned flags) {
    if (ctx && batch > 0)
        if (int rc = check_no_alias(ctx, d_c1c, stride1, gsw_ctx_compact_elems(ctx), d_out, stride_out, batch, "left"))
            return rc;
    return mul_prepared(ctx, d_c1c, stride1, d_prepared, stride_prepared_bytes, d_out, stride_out, batch, flags,
                        nullptr);
}

// least amount of staged data (prepared right operands, random bit matrices) that is worth a pipeline stage of its own
constexpr size_t kMinChunkBytes = (size_t)64 << 20;

// How a batch is cut into chunks of prepared right operands: pipelined over the two workspace halves (the
// preparation of chunk k+1 beside the contraction of chunk k) or one after another on one stream.
struct ChunkPlan {
    bool pipelined = false;
    size_t capacity = 0;                         // right operands one workspace half (or the workspace) holds
    std::vector<std::pair<size_t, int>> chunks;  // (first entry, entries)
};

static ChunkPlan plan_chunks(const gsw_ctx *ctx, int batch) {
    ChunkPlan plan;
    const size_t prep = gsw_ctx_prepared_bytes(ctx);
    size_t pipe_chunk = (size_t)ctx->overlap_chunk;
    // shorter batches (one level of a circuit) still get four chunks to overlap, down to 32 products
    if (pipe_chunk && (size_t)batch < 4 * pipe_chunk) pipe_chunk = (size_t)batch / 4 > 32 ? (size_t)batch / 4 : 32;
    // small shapes: a chunk whose preparation is shorter than the launches and event hand-offs that pipelining adds
    // is not worth cutting off -- a chunk holds at least kMinChunkBytes of prepared operands (n = 10: 1 366 products)
    if (pipe_chunk && pipe_chunk * prep < kMinChunkBytes) pipe_chunk = (kMinChunkBytes + prep - 1) / prep;
    if (pipe_chunk > ctx->ws_limit / (2 * prep)) pipe_chunk = ctx->ws_limit / (2 * prep);
    if (pipe_chunk >= 1 && (size_t)batch > pipe_chunk) {
        plan.pipelined = true;
        plan.capacity = pipe_chunk;
        // the first chunk's preparation is the only one nothing hides: keep it to a quarter chunk
        size_t step = pipe_chunk >= 64 ? pipe_chunk / 4 : pipe_chunk;
        for (size_t b0 = 0; b0 < (size_t)batch; b0 += step, step = pipe_chunk)
            plan.chunks.emplace_back(b0, (int)((size_t)batch - b0 < step ? (size_t)batch - b0 : step));
        return plan;
    }
    size_t chunk = ctx->ws_limit / prep;
    if (chunk < 1) chunk = 1;
    if (chunk > (size_t)batch) chunk = (size_t)batch;
    plan.capacity = chunk;
    for (size_t b0 = 0; b0 < (size_t)batch; b0 += chunk)
        plan.chunks.emplace_back(b0, (int)((size_t)batch - b0 < chunk ? (size_t)batch - b0 : chunk));
    return plan;
}

// Slot lists of a gate-list run (device arrays): entry b uses d_c1c + lhs[b] * stride1 and d_out + out[b] * stride_out.
// Right operands: per chunk of the plan, `rhs_distinct` lists each distinct slot once -- that many operands are
// prepared -- and rhs_index[b] says which of them entry b multiplies by.
struct SlotLists {
    const uint32_t *lhs = nullptr, *out = nullptr, *rhs_index = nullptr, *rhs_distinct = nullptr;
    const ChunkPlan *plan = nullptr;
    const std::vector<int> *distinct_counts = nullptr;  // per chunk of the plan
    explicit operator bool() const { return lhs != nullptr; }
};

static int mul_batch(gsw_ctx *ctx, const uint32_t *d_c1c, size_t stride1, const uint32_t *d_c2c, size_t stride2,
                     uint32_t *d_out, size_t stride_out, int batch, unsigned flags, uint64_t *d_digests,
                     SlotLists slots = SlotLists()) {
    ScopedRange range("gsw_mul_gadget_batch");
    if (int rc = check_gemm_common(ctx, d_c1c, stride1, d_out, stride_out, batch, flags)) return rc;
    if (batch == 0) return GSW_OK;
    if (!d_c2c) return fail(GSW_E_INVALID, "null right operand");
    if (!slots) {  // gate lists check slot numbers instead
        const size_t elems = gsw_ctx_compact_elems(ctx);
        if (int rc = check_no_alias(ctx, d_c1c, stride1, elems, d_out, stride_out, batch, "left")) return rc;
        if (int rc = check_no_alias(ctx, d_c2c, stride2, elems, d_out, stride_out, batch, "right")) return rc;
    }
    if (int rc = use_device(ctx)) return rc;
    if (d_digests) HIP_TRY(hipMemsetAsync(d_digests, 0, sizeof(uint64_t) * (size_t)batch, ctx->stream));
    const size_t prep = gsw_ctx_prepared_bytes(ctx);
    if (stride2 == 0 && !slots) {
        // a shared right operand is prepared once
        if (int rc = ensure_workspace(ctx, prep)) return rc;
        if (int rc = gsw_prepare_rhs(ctx, d_c2c, 0, ctx->ws, 1)) return rc;
        return mul_prepared(ctx, d_c1c, stride1, ctx->ws, 0, d_out, stride_out, batch, flags, d_digests);
    }
    ChunkPlan own_plan;
    if (!slots) own_plan = plan_chunks(ctx, batch);
    const ChunkPlan &plan = slots ? *slots.plan : own_plan;
    if (int rc = ensure_workspace(ctx, (plan.pipelined ? 2 : 1) * plan.capacity * prep)) return rc;
    // The body reports failures through its return value; whatever it had enqueued on the auxiliary stream by
    // then is joined into ctx->stream below, so that no preparation keeps writing a workspace half unordered
    // with the caller's next call.
    auto body = [&]() -> int {
        hipStream_t prep_stream = plan.pipelined ? ctx->aux_stream : ctx->stream;
        if (plan.pipelined) {
            // Two-stage pipeline over chunks: the (HBM-bound) preparation of chunk k+1 runs on the auxiliary
            // stream beside the (MFMA-bound) contraction of chunk k; two workspace halves, events for the
            // hand-offs in both directions, no host synchronisation.
            HIP_TRY(hipEventRecord(ctx->ev_entry, ctx->stream));
            HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_entry, 0));
        }
        size_t distinct_seen = 0;
        for (size_t k = 0; k < plan.chunks.size(); ++k) {
            const size_t b0 = plan.chunks[k].first;
            const int cur = plan.chunks[k].second;
            const int half = plan.pipelined ? (int)(k & 1) : 0;
            uint8_t *ws = (uint8_t *)ctx->ws + (size_t)half * plan.capacity * prep;
            if (plan.pipelined && k >= 2) HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_gemm[half], 0));
            if (slots) {
                // with slot lists the bases stay put and the lists advance; each distinct right operand once
                const int distinct = (*slots.distinct_counts)[k];
                if (int rc = prepare_rhs_on(ctx, d_c2c, stride2, ws, distinct, prep_stream, slots.rhs_distinct + distinct_seen))
                    return rc;
                distinct_seen += (size_t)distinct;
            } else if (int rc = prepare_rhs_on(ctx, d_c2c + b0 * stride2, stride2, ws, cur, prep_stream)) {
                return rc;
            }
            if (plan.pipelined) {
                HIP_TRY(hipEventRecord(ctx->ev_prep[half], ctx->aux_stream));
                HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_prep[half], 0));
            }
            const size_t o = slots ? 0 : b0;
            if (int rc = mul_prepared(ctx, d_c1c + o * stride1, stride1, ws, prep, d_out + o * stride_out, stride_out,
                                      cur, flags, d_digests ? d_digests + b0 : nullptr,
                                      slots ? slots.lhs + b0 : nullptr, slots ? slots.out + b0 : nullptr,
                                      slots ? slots.rhs_index + b0 : nullptr))
                return rc;
            if (plan.pipelined) HIP_TRY(hipEventRecord(ctx->ev_gemm[half], ctx->stream));
        }
        return GSW_OK;
    };
    const int rc = body();
    if (rc != GSW_OK && plan.pipelined) join_aux(ctx);
    return rc;
}

int gsw_mul_gadget_batch(gsw_ctx *ctx, const uint32_t *d_c1c, size_t stride1, const uint32_t *d_c2c, size_t stride2,
                         uint32_t *d_out, size_t stride_out, int batch, unsigned flags) {
    return mul_batch(ctx, d_c1c, stride1, d_c2c, stride2, d_out, stride_out, batch, flags, nullptr);
}

int gsw_mul_gadget_batch_digest(gsw_ctx *ctx, const uint32_t *d_c1c, size_t stride1, const uint32_t *d_c2c,
                                size_t stride2, uint32_t *d_out, size_t stride_out, int batch, unsigned flags,
                                uint64_t *d_digests) {
    if (!d_digests) return fail(GSW_E_INVALID, "null digest buffer");
    return mul_batch(ctx, d_c1c, stride1, d_c2c, stride2, d_out, stride_out, batch, flags, d_digests);
}

// gsw_mul_gadget_batch without device allocation: the prepared operands of a chunk live in this GPU's copy of
// g_static_scratch (ctx->static_scratch, resolved at gsw_ctx_create), chunks run one after another on
// ctx->stream.  No hipMalloc / hipFree / event creation on this path; the stream's order protects the
// scratch between chunks and between successive calls on the same stream.
int gsw_mul_gadget_batch_static(gsw_ctx *ctx, const uint32_t *d_c1c, size_t stride1, const uint32_t *d_c2c,
                                size_t stride2, uint32_t *d_out, size_t stride_out, int batch, unsigned flags) {
    if (int rc = check_gemm_common(ctx, d_c1c, stride1, d_out, stride_out, batch, flags)) return rc;
    if (batch == 0) return GSW_OK;
    if (!d_c2c) return fail(GSW_E_INVALID, "null right operand");
    const size_t elems = gsw_ctx_compact_elems(ctx);
    if (int rc = check_no_alias(ctx, d_c1c, stride1, elems, d_out, stride_out, batch, "left")) return rc;
    if (int rc = check_no_alias(ctx, d_c2c, stride2, elems, d_out, stride_out, batch, "right")) return rc;
    const size_t prep = gsw_ctx_prepared_bytes(ctx);
    if (prep > GSW_STATIC_SCRATCH_BYTES)
        return fail(GSW_E_UNSUPPORTED, "one prepared operand of this context takes %zu bytes, the static scratch holds %zu: "
                                       "n=%d is beyond what gsw_mul_gadget_batch_static supports", prep,
                    (size_t)GSW_STATIC_SCRATCH_BYTES, ctx->n);
    if (!ctx->static_scratch) return fail(GSW_E_HIP, "the static scratch array was not resolved on device %d", ctx->device);
    if (int rc = use_device(ctx)) return rc;
    struct Guard {
        gsw_ctx *c;
        explicit Guard(gsw_ctx *c_) : c(c_) { c->in_static_call = true; }
        ~Guard() { c->in_static_call = false; }
    } guard(ctx);
    void *scratch = ctx->static_scratch;  // the device address, never the array's name
    // One user of the array at a time, in the order of the calls: the lock covers this call's enqueues, the event
    // makes this stream wait for the previous call's last contraction (a no-op on the stream that recorded it).
    std::lock_guard<std::mutex> lock(g_static_mutex);
    hipEvent_t last = g_static_last[ctx->device];
    HIP_TRY(hipStreamWaitEvent(ctx->stream, last, 0));
    auto body = [&]() -> int {
        if (stride2 == 0) {
            // one right operand for the whole batch: prepared once
            if (int rc = prepare_rhs_on(ctx, d_c2c, 0, scratch, 1, ctx->stream)) return rc;
            return mul_prepared(ctx, d_c1c, stride1, scratch, 0, d_out, stride_out, batch, flags, nullptr);
        }
        size_t chunk = GSW_STATIC_SCRATCH_BYTES / prep;
        if (chunk > (size_t)batch) chunk = (size_t)batch;
        for (size_t b0 = 0; b0 < (size_t)batch; b0 += chunk) {
            const int cur = (int)((size_t)batch - b0 < chunk ? (size_t)batch - b0 : chunk);
            if (int rc = prepare_rhs_on(ctx, d_c2c + b0 * stride2, stride2, scratch, cur, ctx->stream)) return rc;
            if (int rc = mul_prepared(ctx, d_c1c + b0 * stride1, stride1, scratch, prep, d_out + b0 * stride_out,
                                      stride_out, cur, flags, nullptr))
                return rc;
        }
        return GSW_OK;
    };
    const int rc = body();
    // also after a failure: whatever was enqueued still reads or writes the array
    const hipError_t e = hipEventRecord(last, ctx->stream);
    if (rc == GSW_OK && e != hipSuccess) return fail(GSW_E_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
    return rc;
}

// ---- gate lists over a pool of resident ciphertexts ----

// d_cache / h_valid: optional cache of prepared right operands beside the pool (gsw_eval_circuit_cached)
static int eval_circuit_impl(gsw_ctx *ctx, uint32_t *d_pool, size_t slot_stride, uint32_t n_slots, const gsw_gate *h_gates,
                             size_t n_gates, void *d_cache, uint8_t *h_valid) {
    ScopedRange range("gsw_eval_circuit");
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (n_gates == 0) return GSW_OK;
    if (!d_pool || !h_gates) return fail(GSW_E_INVALID, "null operand");
    if ((d_cache != nullptr) != (h_valid != nullptr) || (d_cache && !aligned16(d_cache)))
        return fail(GSW_E_INVALID, "the prepared-operand cache needs a 16-byte aligned device array and its host flags");
    const size_t elems = gsw_ctx_compact_elems(ctx);
    if (!aligned16(d_pool) || (slot_stride & 3u) || slot_stride < elems)
        return fail(GSW_E_INVALID, "pool must be 16-byte aligned with a slot stride >= %zu and a multiple of 4", elems);
    if (n_gates > (size_t)INT32_MAX) return fail(GSW_E_INVALID, "too many gates");
    for (size_t i = 0; i < n_gates; ++i) {
        const gsw_gate &g = h_gates[i];
        if (g.op > GSW_GATE_MULT_CONST) return fail(GSW_E_INVALID, "gate %zu: unknown op %u", i, g.op);
        if (g.a >= n_slots || g.out >= n_slots || (g.op != GSW_GATE_MULT_CONST && g.b >= n_slots))
            return fail(GSW_E_INVALID, "gate %zu: slot out of range (pool has %u)", i, n_slots);
        // every gate is checked here, before the first launch: a rejected list leaves the pool untouched
        if (g.op == GSW_GATE_MULT_CONST &&
            (ctx->ring.kind == gswk::kRingFp ? g.b >= gswk::kFpP : (g.b & ~ctx->ring.mask) != 0))
            return fail(GSW_E_INVALID, "gate %zu: constant %u is not a canonical ring element", i, g.b);
        if ((g.op == GSW_GATE_MULT || g.op == GSW_GATE_NAND) && (g.out == g.a || g.out == g.b))
            return fail(GSW_E_INVALID, "gate %zu: a product cannot overwrite its own operand (slot %u)", i, g.out);
    }
    if (int rc = use_device(ctx)) return rc;

    // Schedule: maximal runs of consecutive product gates with one epilogue (mult or nand) in which no
    // gate reads or overwrites a slot written by an earlier gate of the run, nor writes a slot an
    // earlier gate of the run reads -- such a run is one batched launch with the list's sequential meaning.
    struct Run {
        size_t first, count;  // product gates [first, first+count) of the packed slot lists; count 0 = elementwise gate
        size_t gate;          // index into h_gates of an elementwise gate
        unsigned flags;
        ChunkPlan plan;               // how the run's products are cut into chunks of prepared right operands
        std::vector<int> distinct;    // per chunk: distinct right-operand slots (each is prepared once)
        size_t distinct_first = 0;    // where the run's distinct lists start in the staged array
    };
    std::vector<Run> runs;
    std::vector<uint32_t> lhs, rhs, out;
    std::vector<uint32_t> read_in(n_slots, 0), written_in(n_slots, 0);  // run number + 1 that touched the slot
    uint32_t run_id = 0;
    bool open = false;
    for (size_t i = 0; i < n_gates; ++i) {
        const gsw_gate &g = h_gates[i];
        if (g.op == GSW_GATE_ADD || g.op == GSW_GATE_MULT_CONST) {
            runs.push_back({0, 0, i, 0, {}, {}, 0});
            open = false;
            continue;
        }
        const unsigned flags = g.op == GSW_GATE_NAND ? GSW_FLAG_NAND : 0u;
        const bool hazard = open && (written_in[g.a] == run_id || written_in[g.b] == run_id ||
                                     written_in[g.out] == run_id || read_in[g.out] == run_id);
        if (!open || hazard || runs.back().flags != flags) {
            runs.push_back({lhs.size(), 0, 0, flags, {}, {}, 0});
            ++run_id;
            open = true;
        }
        lhs.push_back(g.a);
        rhs.push_back(g.b);
        out.push_back(g.out);
        read_in[g.a] = read_in[g.b] = run_id;
        written_in[g.out] = run_id;
        ++runs.back().count;
    }

    const size_t products = lhs.size();
    std::vector<uint32_t> rhs_index(products), rhs_distinct;
    rhs_distinct.reserve(products);
    if (!d_cache) {
        // Right operands: within a chunk every distinct slot is prepared once, whatever the number of gates
        // that multiply by it; rhs_index[b] is entry b's position in its chunk's list of distinct slots.
        std::vector<uint32_t> position(n_slots, 0), stamp(n_slots, 0);
        uint32_t chunk_id = 0;
        for (Run &r : runs) {
            if (r.count == 0) continue;
            r.plan = plan_chunks(ctx, (int)r.count);
            r.distinct_first = rhs_distinct.size();
            for (const auto &c : r.plan.chunks) {
                ++chunk_id;
                uint32_t distinct = 0;
                for (size_t b = r.first + c.first; b < r.first + c.first + (size_t)c.second; ++b) {
                    const uint32_t slot = rhs[b];
                    if (stamp[slot] != chunk_id) {
                        stamp[slot] = chunk_id;
                        position[slot] = distinct++;
                        rhs_distinct.push_back(slot);
                    }
                    rhs_index[b] = position[slot];
                }
                r.distinct.push_back((int)distinct);
            }
        }
    } else {
        // With a cache every slot's prepared form lives in its own cache entry: a run prepares the right
        // operands whose entry is stale (each once), multiplies straight out of the cache, and the slots it
        // writes go stale.  The flags are walked in list order here, on the host, before anything is launched.
        for (Run &r : runs) {
            if (r.count == 0) {
                h_valid[h_gates[r.gate].out] = 0;  // add / mult_const overwrite their output slot
                continue;
            }
            r.distinct_first = rhs_distinct.size();
            uint32_t missing = 0;
            for (size_t b = r.first; b < r.first + r.count; ++b) {
                rhs_index[b] = rhs[b];  // the cache entry is the slot number
                if (!h_valid[rhs[b]]) {
                    h_valid[rhs[b]] = 1;
                    rhs_distinct.push_back(rhs[b]);
                    ++missing;
                }
            }
            r.distinct.push_back((int)missing);
            for (size_t b = r.first; b < r.first + r.count; ++b) h_valid[out[b]] = 0;
        }
    }

    if (products) {
        const size_t staged = 3 * products + rhs_distinct.size();
        if (staged > ctx->slots_cap) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (ctx->d_slots) HIP_TRY(hipFree(ctx->d_slots));
            if (ctx->h_slots) HIP_TRY(hipHostFree(ctx->h_slots));
            ctx->d_slots = nullptr;
            ctx->h_slots = nullptr;
            ctx->slots_cap = 0;
            const size_t cap = 4 * products < 4096 ? 4096 : 4 * products;
            HIP_TRY(hipMalloc((void **)&ctx->d_slots, cap * sizeof(uint32_t)));
            HIP_TRY(hipHostMalloc((void **)&ctx->h_slots, cap * sizeof(uint32_t), hipHostMallocDefault));
            ctx->slots_cap = cap;
        } else {
            HIP_TRY(hipEventSynchronize(ctx->ev_slots));  // the previous upload has left the staging buffer
        }
        memcpy(ctx->h_slots, lhs.data(), products * sizeof(uint32_t));
        memcpy(ctx->h_slots + products, rhs_index.data(), products * sizeof(uint32_t));
        memcpy(ctx->h_slots + 2 * products, out.data(), products * sizeof(uint32_t));
        memcpy(ctx->h_slots + 3 * products, rhs_distinct.data(), rhs_distinct.size() * sizeof(uint32_t));
        HIP_TRY(hipMemcpyAsync(ctx->d_slots, ctx->h_slots, staged * sizeof(uint32_t), hipMemcpyHostToDevice,
                               ctx->stream));
        HIP_TRY(hipEventRecord(ctx->ev_slots, ctx->stream));
    }
    for (const Run &r : runs) {
        if (r.count == 0) {
            const gsw_gate &g = h_gates[r.gate];
            uint32_t *dst = d_pool + (size_t)g.out * slot_stride;
            const uint32_t *a = d_pool + (size_t)g.a * slot_stride;
            if (g.op == GSW_GATE_ADD) {
                HIP_TRY(gswk::launch_add(ctx->ring, a, d_pool + (size_t)g.b * slot_stride, dst, elems, ctx->stream));
            } else {
                if (dst != a)
                    HIP_TRY(hipMemcpyAsync(dst, a, elems * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
                HIP_TRY(gswk::launch_mult_const(ctx->ring, dst, g.b, elems, ctx->stream));
            }
            continue;
        }
        if (d_cache) {
            // stale right operands into their cache entries, then the whole run as one contraction out of the cache
            const uint32_t *missing = ctx->d_slots + 3 * products + r.distinct_first;
            const size_t prep = gsw_ctx_prepared_bytes(ctx);
            if (r.distinct[0] > 0)
                if (int rc = prepare_rhs_on(ctx, d_pool, slot_stride, d_cache, r.distinct[0], ctx->stream, missing, missing))
                    return rc;
            if (int rc = mul_prepared(ctx, d_pool, slot_stride, d_cache, prep, d_pool, slot_stride, (int)r.count, r.flags,
                                      nullptr, ctx->d_slots + r.first, ctx->d_slots + 2 * products + r.first,
                                      ctx->d_slots + products + r.first))
                return rc;
            continue;
        }
        SlotLists sl;
        sl.lhs = ctx->d_slots + r.first;
        sl.rhs_index = ctx->d_slots + products + r.first;
        sl.out = ctx->d_slots + 2 * products + r.first;
        sl.rhs_distinct = ctx->d_slots + 3 * products + r.distinct_first;
        sl.plan = &r.plan;
        sl.distinct_counts = &r.distinct;
        if (int rc = mul_batch(ctx, d_pool, slot_stride, d_pool, slot_stride, d_pool, slot_stride, (int)r.count,
                               r.flags, nullptr, sl))
            return rc;
    }
    return GSW_OK;
}

int gsw_eval_circuit(gsw_ctx *ctx, uint32_t *d_pool, size_t slot_stride, uint32_t n_slots, const gsw_gate *h_gates,
                     size_t n_gates) try {
    return eval_circuit_impl(ctx, d_pool, slot_stride, n_slots, h_gates, n_gates, nullptr, nullptr);
}
GSW_CATCH

int gsw_eval_circuit_cached(gsw_ctx *ctx, uint32_t *d_pool, size_t slot_stride, uint32_t n_slots, void *d_prepared_pool,
                            uint8_t *h_valid, const gsw_gate *h_gates, size_t n_gates) try {
    if (n_gates && (!d_prepared_pool || !h_valid)) return fail(GSW_E_INVALID, "null prepared-operand cache");
    if (n_gates == 0 || !ctx) return eval_circuit_impl(ctx, d_pool, slot_stride, n_slots, h_gates, n_gates, nullptr, nullptr);
    // The flags are walked ahead of the launches, on a copy: a list rejected by the checks leaves the caller's
    // flags (and the pool) as they were; a failure once launches have begun leaves them all cleared, because
    // the copy no longer describes the cache; success writes the copy back.
    std::vector<uint8_t> valid(h_valid, h_valid + n_slots);
    const uint64_t launched_before = ctx->n_prepared + ctx->n_products;
    const int rc = eval_circuit_impl(ctx, d_pool, slot_stride, n_slots, h_gates, n_gates, d_prepared_pool, valid.data());
    if (rc == GSW_OK) memcpy(h_valid, valid.data(), n_slots);
    else if (ctx->n_prepared + ctx->n_products != launched_before) memset(h_valid, 0, n_slots);
    return rc;
}
GSW_CATCH

// ---- encrypt ----

int gsw_prepare_pk(gsw_ctx *ctx, const uint32_t *h_pk_matrix, void *d_pk_prepared) try {
    if (!ctx || !h_pk_matrix || !d_pk_prepared || !aligned16(d_pk_prepared)) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    // stage pk_matrix (m x (n+1), pk.rs:28-38) on the device with the padded leading dimension
    const size_t bytes = (size_t)ctx->m * ctx->ld * 4;
    if (int rc = ensure_small(ctx, bytes + 16)) return rc;
    uint32_t *tmp = (uint32_t *)ctx->small;
    HIP_TRY(hipMemsetAsync(tmp, 0, bytes, ctx->stream));
    HIP_TRY(hipMemcpy2DAsync(tmp, (size_t)ctx->ld * 4, h_pk_matrix, (size_t)ctx->width * 4, (size_t)ctx->width * 4,
                             (size_t)ctx->m, hipMemcpyHostToDevice, ctx->stream));
    gswk::PrepArgs a{};
    a.src = tmp;
    a.src_stride = 0;
    a.dst = (uint8_t *)d_pk_prepared;
    a.dst_stride = gsw_ctx_pk_prepared_bytes(ctx);
    a.rows_valid = ctx->m;
    a.cols_valid = ctx->width;
    a.ld_src = ctx->ld;
    a.elems = ctx->words;
    a.e4 = ctx->ld_r;
    a.batch = 1;
    a.packed_tail = ctx->packed_tail;
    a.layout16 = ctx->gemm_variant >= 4;
    a.scale_mode = scale_mode_for(ctx, a.e4);
    if (int rc = run_prepare(ctx, a, ctx->stream)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the caller's host matrix is free again
    return GSW_OK;
}
GSW_CATCH

int gsw_pack_rbits(gsw_ctx *ctx, const uint8_t *h_bits, uint32_t *d_rbits) try {
    if (!ctx || !h_bits || !d_rbits) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    const int ell = ctx->ring.ell, words = ctx->words;
    std::vector<uint32_t> packed((size_t)ctx->big_n * ctx->ld_r, 0u);
    for (int i = 0; i < ctx->big_n; ++i)
        for (int j = 0; j < words; ++j) {
            uint32_t w = 0;
            for (int k = 0; k < ell && j * ell + k < ctx->m; ++k) {
                const uint8_t bit = h_bits[(size_t)i * ctx->m + (size_t)j * ell + k];
                if (bit > 1) return fail(GSW_E_INVALID, "encryption bits must be 0 or 1");
                w |= (uint32_t)bit << k;
            }
            packed[(size_t)i * ctx->ld_r + j] = w;
        }
    return gsw_memcpy_h2d(ctx, d_rbits, packed.data(), packed.size() * 4);
}
GSW_CATCH

int gsw_random_rbits(gsw_ctx *ctx, uint64_t seed, uint64_t stream, uint32_t *d_rbits, int batch) {
    if (!ctx || !d_rbits || batch < 0) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(gswk::launch_random_words(seed, stream, d_rbits, ctx->big_n, ctx->words, ctx->ld_r, ctx->ring.ell,
                                      ctx->m - (ctx->words - 1) * ctx->ring.ell, batch, gsw_ctx_rbits_elems(ctx),
                                      ctx->stream));
    return GSW_OK;
}

int gsw_encrypt_batch(gsw_ctx *ctx, const void *d_pk_prepared, const uint32_t *d_rbits, size_t stride_rbits,
                      const uint32_t *d_messages, uint32_t *d_out, size_t stride_out, int batch) {
    ScopedRange range("gsw_encrypt_batch");
    if (int rc = check_gemm_common(ctx, d_rbits, stride_rbits, d_out, stride_out, batch, 0)) return rc;
    if (batch == 0) return GSW_OK;
    if (!d_pk_prepared || !aligned16(d_pk_prepared) || !d_messages) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    gswk::GemmArgs a{};
    a.lhs = d_rbits;
    a.lhs_stride = stride_rbits;
    a.rhs = (const uint8_t *)d_pk_prepared;
    a.rhs_stride = 0;  // one public key for the whole batch
    a.out = d_out;
    a.out_stride = stride_out;
    a.addend = d_messages;
    a.rows = ctx->big_n;
    a.e4 = ctx->ld_r;
    a.k_words = ctx->words;
    a.ld_lhs = ctx->ld_r;
    a.out_cols = ctx->width;
    a.ld_out = ctx->ld;
    a.batch = batch;
    a.mode = gswk::kModeEncrypt;
    a.packed_tail = ctx->packed_tail;
    a.scale_mode = scale_mode_for(ctx, a.e4);
    return run_gemm(ctx, a);
}

int gsw_encrypt_batch_prepared(gsw_ctx *ctx, const void *d_pk_prepared, const uint32_t *d_rbits, size_t stride_rbits,
                               const uint32_t *d_messages, uint32_t *d_out, size_t stride_out, void *d_prepared_out,
                               int batch) {
    if (batch > 0 && (!d_prepared_out || !aligned16(d_prepared_out))) return fail(GSW_E_INVALID, "null or unaligned prepared output");
    if (int rc = gsw_encrypt_batch(ctx, d_pk_prepared, d_rbits, stride_rbits, d_messages, d_out, stride_out, batch)) return rc;
    // the fresh ciphertexts straight into the form a product reads its right operand in
    return prepare_rhs_on(ctx, d_out, stride_out, d_prepared_out, batch, ctx->stream);
}

// Encryption from a seed: the random bit matrices never exist as a whole.  They are drawn chunk by chunk into the two
// halves of the workspace on the auxiliary stream, beside the contraction of the chunk before (the generator is
// VALU- and HBM-bound, the contraction MFMA-bound); with d_prepared_out the fresh ciphertexts are also put into
// prepared form there, behind the next chunk's bits.  Same events and hand-offs as the product pipeline.
int gsw_encrypt_batch_seeded(gsw_ctx *ctx, const void *d_pk_prepared, uint64_t seed, uint64_t stream_id,
                             const uint32_t *d_messages, uint32_t *d_out, size_t stride_out, void *d_prepared_out,
                             int batch) try {
    ScopedRange range("gsw_encrypt_batch_seeded");
    if (!ctx) return fail(GSW_E_INVALID, "null context");
    if (batch < 0) return fail(GSW_E_INVALID, "negative batch");
    if (batch == 0) return GSW_OK;
    if (!d_pk_prepared || !aligned16(d_pk_prepared) || !d_messages || !d_out || !aligned16(d_out))
        return fail(GSW_E_INVALID, "null or unaligned argument");
    if (d_prepared_out && !aligned16(d_prepared_out)) return fail(GSW_E_INVALID, "unaligned prepared output");
    if (stride_out < gsw_ctx_compact_elems(ctx) && batch > 1)
        return fail(GSW_E_INVALID, "output stride %zu smaller than one ciphertext (%zu)", stride_out,
                    gsw_ctx_compact_elems(ctx));
    if (int rc = use_device(ctx)) return rc;
    const size_t elems = gsw_ctx_rbits_elems(ctx), bytes = elems * 4, prep = gsw_ctx_prepared_bytes(ctx);
    size_t chunk = 32;
    if (chunk * bytes < kMinChunkBytes) chunk = (kMinChunkBytes + bytes - 1) / bytes;  // small shapes: fewer, longer launches
    if (chunk > ctx->ws_limit / (2 * bytes)) chunk = ctx->ws_limit / (2 * bytes);
    if (chunk < 1) chunk = 1;  // one bit matrix per half is the least the pipeline can work with
    if (chunk > (size_t)batch) chunk = (size_t)batch;
    std::vector<std::pair<size_t, int>> chunks;  // a short first chunk: its bits are the only ones nothing hides
    const bool short_first = chunk >= 16 && (chunk / 4) * bytes >= kMinChunkBytes / 4;
    for (size_t b0 = 0, step = short_first ? chunk / 4 : chunk; b0 < (size_t)batch; b0 += step, step = chunk)
        chunks.emplace_back(b0, (int)((size_t)batch - b0 < step ? (size_t)batch - b0 : step));
    if (int rc = ensure_workspace(ctx, 2 * chunk * bytes)) return rc;
    const int last_bits = ctx->m - (ctx->words - 1) * ctx->ring.ell;
    auto draw = [&](size_t k) -> int {  // the bits of chunk k, on the auxiliary stream, into half k & 1
        const int half = (int)(k & 1);
        if (k >= 2) HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_gemm[half], 0));  // chunk k - 2 has read them
        HIP_TRY(gswk::launch_random_words(seed, stream_id, (uint32_t *)ctx->ws + (size_t)half * chunk * elems,
                                          ctx->big_n, ctx->words, ctx->ld_r, ctx->ring.ell, last_bits, chunks[k].second,
                                          elems, ctx->aux_stream, chunks[k].first));
        HIP_TRY(hipEventRecord(ctx->ev_prep[half], ctx->aux_stream));
        return GSW_OK;
    };
    auto body = [&]() -> int {
        HIP_TRY(hipEventRecord(ctx->ev_entry, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_entry, 0));
        if (int rc = draw(0)) return rc;
        for (size_t k = 0; k < chunks.size(); ++k) {
            const size_t b0 = chunks[k].first;
            const int cur = chunks[k].second, half = (int)(k & 1);
            HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_prep[half], 0));
            if (int rc = gsw_encrypt_batch(ctx, d_pk_prepared, (const uint32_t *)ctx->ws + (size_t)half * chunk * elems, elems,
                                           d_messages + b0, d_out + b0 * stride_out, stride_out, cur))
                return rc;
            HIP_TRY(hipEventRecord(ctx->ev_gemm[half], ctx->stream));
            if (k + 1 < chunks.size())
                if (int rc = draw(k + 1)) return rc;
            if (d_prepared_out) {
                HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_gemm[half], 0));
                if (int rc = prepare_rhs_on(ctx, d_out + b0 * stride_out, stride_out, (uint8_t *)d_prepared_out + b0 * prep,
                                            cur, ctx->aux_stream))
                    return rc;
            }
        }
        return GSW_OK;
    };
    const int rc = body();
    join_aux(ctx);  // the last preparations (and, after a failure, whatever was enqueued) before the caller's next call
    return rc;
}
GSW_CATCH

// ---- decrypt ----

static int row_dots(gsw_ctx *ctx, const uint32_t *h_vec, int cols, const uint32_t *d_x, size_t stride, int ld,
                    int row0, int nrows, int batch, std::vector<uint32_t> &host_out) {
    ScopedRange range("gsw_decrypt");
    const size_t vec_bytes = ((size_t)cols * 4 + 255) & ~(size_t)255;
    const size_t out_bytes = (size_t)batch * nrows * 4;
    if (int rc = ensure_small(ctx, vec_bytes + out_bytes + 16)) return rc;
    uint8_t *tmp = (uint8_t *)ctx->small;
    host_out.resize((size_t)batch * nrows);
    HIP_TRY(hipMemcpyAsync(tmp, h_vec, (size_t)cols * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(gswk::launch_row_dot(ctx->ring, d_x, stride, ld, cols, row0, nrows, (const uint32_t *)tmp, batch,
                                 (uint32_t *)(tmp + vec_bytes), ctx->stream));
    HIP_TRY(hipMemcpyAsync(host_out.data(), tmp + vec_bytes, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GSW_OK;
}

// gsw_impl.rs:127-131
static uint32_t is_zero_one(const gswk::RingDesc &ring, uint32_t x) {
    const uint64_t mx = ring.kind == gswk::kRingFp ? (uint64_t)gswk::kFpP - 1 : (uint64_t)ring.mask;
    const uint32_t lo = (uint32_t)(mx / 4), hi = (uint32_t)(3 * mx / 4);  // both < q: R::from is the identity
    return (x >= lo && x <= hi) ? 1u : 0u;
}

int gsw_decrypt_batch(gsw_ctx *ctx, const uint32_t *h_s, const uint32_t *d_cc, size_t stride, int batch,
                      uint32_t *h_bits_out, uint32_t *h_raw_out) try {
    if (!ctx || !h_s || !d_cc || !h_bits_out || batch < 0) return fail(GSW_E_INVALID, "bad argument");
    if (batch == 0) return GSW_OK;
    if (int rc = use_device(ctx)) return rc;
    std::vector<uint32_t> raw;
    if (int rc = row_dots(ctx, h_s, ctx->width, d_cc, stride, ctx->ld, ctx->ring.ell - 1, 1, batch, raw)) return rc;
    for (int b = 0; b < batch; ++b) {
        h_bits_out[b] = is_zero_one(ctx->ring, raw[b]);
        if (h_raw_out) h_raw_out[b] = raw[b];
    }
    return GSW_OK;
}
GSW_CATCH

int gsw_mp_decrypt_batch(gsw_ctx *ctx, const uint32_t *h_s, const uint32_t *d_cc, size_t stride, int batch,
                         uint32_t *h_messages_out) try {
    if (!ctx || !h_s || !d_cc || !h_messages_out || batch < 0) return fail(GSW_E_INVALID, "bad argument");
    if (ctx->ring.kind == gswk::kRingFp) return fail(GSW_E_UNSUPPORTED, "Only supported for pow2 rings!");  // gsw_impl.rs:69
    if (batch == 0) return GSW_OK;
    if (int rc = use_device(ctx)) return rc;
    const int ell = ctx->ring.ell;
    std::vector<uint32_t> dots;
    if (int rc = row_dots(ctx, h_s, ctx->width, d_cc, stride, ctx->ld, 0, ell, batch, dots)) return rc;
    for (int b = 0; b < batch; ++b) {
        // gsw_impl.rs:72-82: peel the message bits from the most significant gadget row down
        const uint32_t *slice = dots.data() + (size_t)b * ell;
        uint64_t recovered = 0;
        for (int i = 0; i < ell; ++i) {
            const uint32_t entry = slice[ell - 1 - i];
            const uint32_t recovered_exp = (uint32_t)(recovered << (ell - i - 1)) & ctx->ring.mask;
            const uint32_t current = (entry - recovered_exp) & ctx->ring.mask;
            recovered ^= (uint64_t)is_zero_one(ctx->ring, current) << i;
        }
        h_messages_out[b] = (uint32_t)recovered & ctx->ring.mask;
    }
    return GSW_OK;
}
GSW_CATCH

int gsw_partial_decrypt_batch(gsw_ctx *ctx, const uint32_t *h_share, const uint32_t *d_cc, size_t stride, int batch,
                              uint32_t *h_partial_out) try {
    if (!ctx || !h_share || !d_cc || !h_partial_out || batch < 0) return fail(GSW_E_INVALID, "bad argument");
    if (batch == 0) return GSW_OK;
    if (int rc = use_device(ctx)) return rc;
    std::vector<uint32_t> raw;
    if (int rc = row_dots(ctx, h_share, ctx->width, d_cc, stride, ctx->ld, ctx->ring.ell - 1, 1, batch, raw)) return rc;
    memcpy(h_partial_out, raw.data(), sizeof(uint32_t) * (size_t)batch);
    return GSW_OK;
}
GSW_CATCH

int gsw_combine_partials(gsw_ctx *ctx, const uint32_t *h_partials, int parties, int batch, uint32_t *h_bits_out) {
    if (!ctx || !h_partials || !h_bits_out || parties < 1 || batch < 0) return fail(GSW_E_INVALID, "bad argument");
    const uint64_t q = ctx->ring.kind == gswk::kRingFp ? (uint64_t)gswk::kFpP : (uint64_t)ctx->ring.mask + 1;
    for (int b = 0; b < batch; ++b) {
        uint64_t sum = 0;
        for (int p = 0; p < parties; ++p) {
            const uint32_t v = h_partials[(size_t)p * batch + b];
            if (v >= q) return fail(GSW_E_INVALID, "partial %d of entry %d is not a canonical ring value", p, b);
            sum = (sum + v) % q;
        }
        h_bits_out[b] = is_zero_one(ctx->ring, (uint32_t)sum);  // gsw_impl.rs:127-131 on the recombined value
    }
    return GSW_OK;
}

int gsw_matvec(gsw_ctx *ctx, const uint32_t *d_mat, int rows, int cols, int ld, const uint32_t *h_vec,
               uint32_t *d_out) {
    if (!ctx || !d_mat || !h_vec || !d_out || rows < 0 || cols < 1 || ld < cols) return fail(GSW_E_INVALID, "bad argument");
    if (rows == 0) return GSW_OK;
    if (int rc = use_device(ctx)) return rc;
    if (int rc = ensure_small(ctx, (size_t)cols * 4 + 16)) return rc;
    uint32_t *d_vec = (uint32_t *)ctx->small;
    HIP_TRY(hipMemcpyAsync(d_vec, h_vec, (size_t)cols * 4, hipMemcpyHostToDevice, ctx->stream));
    // the grid's x dimension carries the rows in slabs
    for (int r0 = 0; r0 < rows; r0 += 32768) {
        const int cur = rows - r0 < 32768 ? rows - r0 : 32768;
        HIP_TRY(gswk::launch_row_dot(ctx->ring, d_mat, 0, ld, cols, r0, cur, d_vec, 1, d_out + r0, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the caller's host vector is free again
    return GSW_OK;
}

int gsw_decrypt_dense(gsw_ctx *ctx, const uint32_t *h_v, const uint32_t *d_c, uint32_t *h_bit_out) try {
    if (!ctx || !h_v || !d_c || !h_bit_out) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    std::vector<uint32_t> raw;
    if (int rc = row_dots(ctx, h_v, ctx->big_n, d_c, 0, ctx->big_n, ctx->ring.ell - 1, 1, 1, raw)) return rc;
    *h_bit_out = is_zero_one(ctx->ring, raw[0]);
    return GSW_OK;
}
GSW_CATCH

// ---- sampling and digests ----

int gsw_fill_uniform(gsw_ctx *ctx, uint64_t seed, uint64_t stream, uint64_t min, uint64_t max, int unchecked,
                     uint32_t *d_out, size_t count) {
    if (!ctx || !d_out) return fail(GSW_E_INVALID, "bad argument");
    if (!unchecked && max > gswk::kFpP) return fail(GSW_E_INVALID, "assertion failed: max <= P");  // error_sampling.rs:21
    if (min > max || max - min >= (1ull << 32)) return fail(GSW_E_INVALID, "range must fit 32 bits");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(gswk::launch_fill_uniform(ctx->ring, seed, stream, min, max, d_out, count, ctx->stream));
    return GSW_OK;
}

int gsw_sample_naive(gsw_ctx *ctx, uint64_t seed, uint64_t stream, uint32_t *d_out, size_t count) {
    // error_sampling.rs:75-79: rnd_ring_elm(0, P >> 15) * NOISE_CONST, NOISE_CONST = 1
    return gsw_fill_uniform(ctx, seed, stream, 0, gswk::kFpP >> 15, 1, d_out, count);
}

int gsw_sample_gaussian(gsw_ctx *ctx, uint64_t seed, uint64_t stream, double sigma, uint32_t *d_out, size_t count) try {
    if (!ctx || !d_out || !(sigma > 0.0) || sigma > 300.0) return fail(GSW_E_INVALID, "bad argument (0 < sigma <= 300)");
    if (int rc = use_device(ctx)) return rc;
    std::vector<uint64_t> cum;
    int kmax = 0;
    gswwire::gaussian_table(sigma, cum, kmax);
    if (int rc = ensure_small(ctx, cum.size() * 8)) return rc;
    uint64_t *d_cum = (uint64_t *)ctx->small;
    HIP_TRY(hipMemcpyAsync(d_cum, cum.data(), cum.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(gswk::launch_sample_table(ctx->ring, seed, stream, d_cum, (int)cum.size(), kmax, d_out, count, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the host table goes out of scope
    return GSW_OK;
}
GSW_CATCH

// keygen (gsw_impl.rs:17-25) without the host in the data path: the samples, b = B t + e, pk_matrix = [b | B] and its
// prepared form are all made on the device; only the secret t (n values) comes back, because GswSk lives on the host.
int gsw_keygen_device(gsw_ctx *ctx, uint64_t seed, uint64_t first_stream, int sampler, double sigma, int unchecked,
                      uint32_t *h_t_out, uint32_t *d_pk_matrix_out, void *d_pk_prepared) try {
    ScopedRange range("gsw_keygen_device");
    if (!ctx || !h_t_out || !d_pk_prepared || !aligned16(d_pk_prepared)) return fail(GSW_E_INVALID, "bad argument");
    if (sampler != GSW_SAMPLER_NAIVE && sampler != GSW_SAMPLER_GAUSSIAN) return fail(GSW_E_INVALID, "unknown sampler %d", sampler);
    if (sampler == GSW_SAMPLER_GAUSSIAN && (!(sigma > 0.0) || sigma > 300.0)) return fail(GSW_E_INVALID, "bad argument (0 < sigma <= 300)");
    const uint64_t max = ctx->ring.kind == gswk::kRingFp ? (uint64_t)gswk::kFpP - 1 : (uint64_t)ctx->ring.mask;
    if (!unchecked && max > gswk::kFpP) return fail(GSW_E_INVALID, "assertion failed: max <= P");  // error_sampling.rs:21
    if (int rc = use_device(ctx)) return rc;
    const size_t n = (size_t)ctx->n, m = (size_t)ctx->m;
    auto round = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    std::vector<uint64_t> cum;
    int kmax = 0;
    if (sampler == GSW_SAMPLER_GAUSSIAN) gswwire::gaussian_table(sigma, cum, kmax);
    const size_t o_t = 0, o_e = o_t + round(n * 4), o_bt = o_e + round(m * 4), o_cum = o_bt + round(m * 4),
                 o_b = o_cum + round(cum.size() * 8 + 8), o_pk = o_b + round(m * n * 4), total = o_pk + round(m * ctx->ld * 4);
    if (int rc = ensure_small(ctx, total)) return rc;
    uint8_t *base = (uint8_t *)ctx->small;
    uint32_t *d_t = (uint32_t *)(base + o_t), *d_e = (uint32_t *)(base + o_e), *d_bt = (uint32_t *)(base + o_bt),
             *d_b = (uint32_t *)(base + o_b), *d_pk = (uint32_t *)(base + o_pk);
    uint64_t *d_cum = (uint64_t *)(base + o_cum);
    hipStream_t st = ctx->stream;
    // the draws, in the order and on the streams of gsw_impl.rs:18-21: t, e, B
    HIP_TRY(gswk::launch_fill_uniform(ctx->ring, seed, first_stream, 0, max, d_t, n, st));
    if (sampler == GSW_SAMPLER_NAIVE) {
        HIP_TRY(gswk::launch_fill_uniform(ctx->ring, seed, first_stream + 1, 0, gswk::kFpP >> 15, d_e, m, st));  // error_sampling.rs:75-79
    } else {
        HIP_TRY(hipMemcpyAsync(d_cum, cum.data(), cum.size() * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(gswk::launch_sample_table(ctx->ring, seed, first_stream + 1, d_cum, (int)cum.size(), kmax, d_e, m, st));
    }
    HIP_TRY(gswk::launch_fill_uniform(ctx->ring, seed, first_stream + 2, 0, max, d_b, m * n, st));
    // b = B t + e (pk.rs:26), pk_matrix = [b | B] (pk.rs:28-36)
    for (int r0 = 0; r0 < ctx->m; r0 += 32768) {
        const int cur = ctx->m - r0 < 32768 ? ctx->m - r0 : 32768;
        HIP_TRY(gswk::launch_row_dot(ctx->ring, d_b, 0, ctx->n, ctx->n, r0, cur, d_t, 1, d_bt + r0, st));
    }
    HIP_TRY(gswk::launch_assemble_pk(ctx->ring, d_bt, d_e, d_b, ctx->m, ctx->n, ctx->ld, d_pk, st));
    gswk::PrepArgs a{};
    a.src = d_pk;
    a.dst = (uint8_t *)d_pk_prepared;
    a.dst_stride = gsw_ctx_pk_prepared_bytes(ctx);
    a.rows_valid = ctx->m;
    a.cols_valid = ctx->width;
    a.ld_src = ctx->ld;
    a.elems = ctx->words;
    a.e4 = ctx->ld_r;
    a.batch = 1;
    a.packed_tail = ctx->packed_tail;
    a.layout16 = ctx->gemm_variant >= 4;
    a.scale_mode = scale_mode_for(ctx, a.e4);
    if (int rc = run_prepare(ctx, a, st)) return rc;
    if (d_pk_matrix_out)
        HIP_TRY(hipMemcpyAsync(d_pk_matrix_out, d_pk, m * ctx->ld * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(h_t_out, d_t, n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return GSW_OK;
}
GSW_CATCH

int gsw_diag_stream_read(gsw_ctx *ctx, const void *d_src, size_t src_bytes, size_t bytes, int mode, size_t pitch) {
    if (!ctx || !d_src || (bytes & 0xFFFF) || mode < 0 || mode > 1) return fail(GSW_E_INVALID, "bad argument");
    if (!aligned16(d_src)) return fail(GSW_E_INVALID, "source must be 16-byte aligned");
    if (mode == 1 && (pitch < 64 || (pitch & 15))) return fail(GSW_E_INVALID, "bad pitch");
    // the kernel touches [0, bytes) in mode 0 and rows 0 .. bytes/64 - 1 of `pitch` bytes in mode 1
    const size_t span = mode == 0 ? bytes : (bytes / 64) * pitch;
    if (span > src_bytes) return fail(GSW_E_INVALID, "source buffer too small: needs %zu bytes, has %zu", span, src_bytes);
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(gswk::launch_stream_read(d_src, bytes, mode, pitch, ctx->stream));
    return GSW_OK;
}

int gsw_diag_limb_start_values(int ring_kind, int bits, int k_slots, int wide, uint32_t *h_start4, int *lean_out) {
    if (!h_start4 || !lean_out || k_slots <= 0 || (ring_kind != GSW_RING_FP_M31 && ring_kind != GSW_RING_ZPOW2))
        return fail(GSW_E_INVALID, "bad argument");
    gswk::RingDesc ring;
    ring.kind = ring_kind == GSW_RING_FP_M31 ? gswk::kRingFp : gswk::kRingZpow2;
    ring.ell = ring_kind == GSW_RING_FP_M31 ? 31 : bits;
    ring.mask = ring_kind == GSW_RING_FP_M31 ? gswk::kFpP : (bits >= 32 ? 0xFFFFFFFFu : (1u << bits) - 1u);
    uint32_t c[4];
    // what launch_gemm does for the default kernel: the scaling mode of the contraction's length, its sums' bound
    const int e4 = (k_slots + 31) / 32;
    const long units = gswk::scale_units(e4, gswk::bit_scale_mode(ring, e4));
    if (!gswk::limb_start_values(ring, units, wide != 0, c, lean_out))
        return fail(GSW_E_INVALID, "no start values for %d K slots", k_slots);
    for (int l = 0; l < 4; ++l) h_start4[l] = c[l];
    return GSW_OK;
}

#ifdef GSW_EXP_TILE_CLOCK
int gsw_diag_tile_clock(gsw_ctx *ctx, uint64_t *h_out16, int reset) {
    if (!ctx || !h_out16) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(gswk::read_tile_clock((unsigned long long *)h_out16, reset != 0));
    return GSW_OK;
}
#endif

int gsw_diag_bit_scale_mode(int ring_kind, int bits, int k_slots) {
    if (k_slots <= 0 || (ring_kind != GSW_RING_FP_M31 && ring_kind != GSW_RING_ZPOW2)) return -1;
    gswk::RingDesc ring;
    ring.kind = ring_kind == GSW_RING_FP_M31 ? gswk::kRingFp : gswk::kRingZpow2;
    ring.ell = ring_kind == GSW_RING_FP_M31 ? 31 : bits;
    ring.mask = ring_kind == GSW_RING_FP_M31 ? gswk::kFpP : (bits >= 32 ? 0xFFFFFFFFu : (1u << bits) - 1u);
    return gswk::bit_scale_mode(ring, (k_slots + 31) / 32);
}

int gsw_digest_batch(gsw_ctx *ctx, const uint32_t *d_x, size_t stride, size_t count, int batch, uint64_t *d_digests) {
    if (!ctx || !d_x || !d_digests || batch < 0) return fail(GSW_E_INVALID, "bad argument");
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(gswk::launch_digest(d_x, stride, count, batch, (unsigned long long *)d_digests, ctx->stream));
    return GSW_OK;
}

}  // extern "C"
