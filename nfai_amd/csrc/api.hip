// api.hip — C ABI of libnfai_hip.so: context, buffers (the HipBufferManager side of
// VulkanBufferManager) and the 1:1 / fused operator entry points.  See include/nfai_hip.h for the
// reference member each function replaces.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <unordered_set>
#include <vector>

#include "common.h"

namespace nfai {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// Handles are pointers, but they are only dereferenced after a lookup in the table of live
// objects: a stale or made-up handle is an error status, never a wild read.
static std::mutex g_handles_mu;
static std::unordered_set<uint64_t> g_handles;

void handle_register(const void *p)
{
    std::lock_guard<std::mutex> lk(g_handles_mu);
    g_handles.insert(reinterpret_cast<uint64_t>(p));
}
void handle_unregister(const void *p)
{
    std::lock_guard<std::mutex> lk(g_handles_mu);
    g_handles.erase(reinterpret_cast<uint64_t>(p));
}
bool handle_live(uint64_t h)
{
    std::lock_guard<std::mutex> lk(g_handles_mu);
    return g_handles.count(h) != 0;
}

#ifdef NFAI_STAMPS
// Diagnostic build: launch slots of the stamp buffer in launch order (the tool installs the buffer, runs, reads it back).
static unsigned long long *g_stamp_buf = nullptr;
static uint32_t g_stamp_slots = 0, g_stamp_next = 0;
static StampSlot g_stamp_info[4096];
unsigned long long *stamp_next_slot(const char *name, uint32_t grid, uint32_t block)
{
    if (!g_stamp_buf || g_stamp_next >= g_stamp_slots || g_stamp_next >= 4096) return nullptr;
    StampSlot &si = g_stamp_info[g_stamp_next];
    snprintf(si.name, sizeof(si.name), "%s", name);
    si.grid = grid; si.block = block;
    return g_stamp_buf + (size_t)(g_stamp_next++) * STAMP_WAVES * STAMP_WORDS;
}
#endif

Ctx *ctx_of(nfai_ctx_t h)
{
    if (!handle_live(h)) return nullptr;
    Ctx *c = reinterpret_cast<Ctx *>(h);
    return c->magic == 0x4E464358 ? c : nullptr;
}

Buf *buf_of(nfai_buf_t h)
{
    if (!handle_live(h)) return nullptr;
    Buf *b = reinterpret_cast<Buf *>(h);
    return b->magic == 0x4E464246 ? b : nullptr;
}

uint64_t weight_row_bytes(int t, uint64_t n_cols)
{
    switch (t) {
        case NFAI_F32: return n_cols * 4;
        case NFAI_F16: return n_cols % 8 == 0 ? n_cols * 2 : 0;
        case NFAI_Q4_K:
        case NFAI_Q4_K_T16: return n_cols % 256 == 0 ? n_cols / 256 * 144 : 0;
        case NFAI_Q6_K:
        case NFAI_Q6_K_T16: return n_cols % 256 == 0 ? n_cols / 256 * 210 : 0;
    }
    return 0;
}

}  // namespace nfai

using namespace nfai;

#define CTX_OR_FAIL(c, h)                                                    \
    Ctx *c = ctx_of(h);                                                      \
    if (!c) return fail(NFAI_ERR_INVALID, "%s: invalid context handle", __func__); \
    HIP_TRY(hipSetDevice(c->device))

#define BUF_OR_FAIL(b, h)                                                   \
    Buf *b = buf_of(h);                                                     \
    if (!b) return fail(NFAI_ERR_INVALID, "%s: invalid buffer handle (%s)", __func__, #h)

#define LAUNCH_TRY(expr)                                                                        \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(_e == hipErrorInvalidValue ? NFAI_ERR_INVALID : NFAI_ERR_HIP,           \
                        "%s: launch failed: %s", __func__, hipGetErrorString(_e));              \
    } while (0)

NFAI_API const char *nfai_hip_last_error(void) { return g_last_error.c_str(); }
NFAI_API int32_t nfai_hip_abi_version(void) { return NFAI_HIP_ABI_VERSION; }

// ---- context ---------------------------------------------------------------------------------
// The context scratch backs the op-level entry points (the model object allocates its own workspaces, llama.hip).  Every user owns a
// DISJOINT range: several of them keep ticket words that must read zero between launches, so a range shared by two users corrupts
// the other's hand-off silently (the fused lm_head + ArgMax partials used to cover the attention tickets).
//   [0, 4 KB)           k_argmax partials + ticket                 (nfai_hip_argmax)
//   [4 KB, 16 KB)       fused lm_head + ArgMax partials + ticket   (nfai_hip_lmhead_argmax)
//   [16 KB, 1 MB)       attention tickets + slice partials         (nfai_hip_attn_decode, _gemv_qkv_rope, _engine_block)
//   [1 MB, 4 MB - 4 KB) top-k workspace                            (nfai_hip_topk)
//   [4 MB - 4 KB, 4 MB) cos/sin table of the position | position word (last 256 bytes)
constexpr size_t SCRATCH_BYTES = 4u << 20;
constexpr size_t SCR_ARGMAX = 0, SCR_ARGMAX_END = 4096;
constexpr size_t SCR_LMHEAD = 4096, SCR_LMHEAD_END = 16384;
constexpr size_t SCR_ATTN = 16384, SCR_ATTN_END = 1u << 20;
constexpr size_t SCR_TOPK = 1u << 20, SCR_TOPK_END = SCRATCH_BYTES - 4096;
constexpr size_t SCR_ROPECS = SCRATCH_BYTES - 4096, SCR_POS = SCRATCH_BYTES - 256;
static_assert(2 * ARGMAX_BLOCKS * 4 + 256 <= SCR_ARGMAX_END - SCR_ARGMAX, "k_argmax partials");
static_assert(argmax_fused_bytes() <= SCR_LMHEAD_END - SCR_LMHEAD, "fused lm_head + ArgMax workspace");
static_assert(SCR_ARGMAX_END <= SCR_LMHEAD && SCR_LMHEAD_END <= SCR_ATTN && SCR_ATTN_END <= SCR_TOPK && SCR_TOPK_END <= SCR_ROPECS, "scratch ranges overlap");
static_assert(SCR_ROPECS + 3072 <= SCR_POS, "cos/sin table (<= 3072 bytes, checked by its users) runs into the position word");

static int ctx_create_impl(int device, void *stream, bool own, nfai_ctx_t *out)
{
    if (!out) return fail(NFAI_ERR_INVALID, "ctx_create: out is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(NFAI_ERR_HIP, "ctx_create: no HIP device visible (%s) — this backend has no CPU fallback",
                    hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(NFAI_ERR_INVALID, "ctx_create: device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    Ctx *c = new Ctx();
    c->device = device;
    HIP_TRY(hipGetDeviceProperties(&c->prop, device));
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        const std::string arch = c->prop.gcnArchName;
        delete c;
        return fail(NFAI_ERR_UNSUPPORTED, "ctx_create: device arch %s is not gfx950 (kernels are built for MI355X only)",
                    arch.c_str());
    }
    if (own) {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    } else {
        c->stream = reinterpret_cast<hipStream_t>(stream);
    }
    c->owns_stream = own;
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    c->scratch_bytes = SCRATCH_BYTES;
    HIP_TRY(hipMalloc(&c->scratch, c->scratch_bytes));
    HIP_TRY(hipMemset(c->scratch, 0, c->scratch_bytes));
    handle_register(c);
    *out = reinterpret_cast<nfai_ctx_t>(c);
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_ctx_create(int32_t device, nfai_ctx_t *out) { return ctx_create_impl(device, nullptr, true, out); }
NFAI_API int32_t nfai_hip_ctx_create_on_stream(int32_t device, void *stream, nfai_ctx_t *out)
{
    return ctx_create_impl(device, stream, false, out);
}

static void canary_forget_ctx(Ctx *c);  // NFAI_HIP_DEBUG_CANARY (below, with nfai_hip_buf_alloc)

NFAI_API int32_t nfai_hip_ctx_destroy(nfai_ctx_t h)
{
    CTX_OR_FAIL(c, h);
    hipStreamSynchronize(c->stream);
    canary_forget_ctx(c);  // buffers that outlive their context must not be checked against a later context at the same address
    hipEventDestroy(c->ev0);
    hipEventDestroy(c->ev1);
    hipFree(c->scratch);
    if (c->owns_stream) hipStreamDestroy(c->stream);
    c->magic = 0;
    handle_unregister(c);
    delete c;
    return NFAI_OK;
}

// How the rows of the long weight-streaming launches are dealt to the XCDs on this context: shares[x] = unit groups per
// dealing round for the workgroups with blockIdx % 8 == x, probe_us[x] = mean time of such a workgroup in the calibration launch
// (equal shares).  All zero before the first model is finalised on the context, or when the dealing is off (NFAI_XCD_DEAL=0).
NFAI_API int32_t nfai_hip_ctx_xcd_shares(nfai_ctx_t h, uint16_t *shares8, float *probe_us8)
{
    CTX_OR_FAIL(c, h);
    for (int x = 0; x < 8; x++) {
        if (shares8) shares8[x] = c->xcd_state == 1 ? c->xcd_shares[x] : 0;
        if (probe_us8) probe_us8[x] = c->xcd_state == 1 ? c->xcd_probe_us[x] : 0.f;
    }
    return NFAI_OK;
}

static int canary_check_all(Ctx *c);  // NFAI_HIP_DEBUG_CANARY (below, with nfai_hip_buf_alloc)

NFAI_API int32_t nfai_hip_ctx_synchronize(nfai_ctx_t h)
{
    CTX_OR_FAIL(c, h);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return canary_check_all(c);
}

NFAI_API int32_t nfai_hip_ctx_device_info(nfai_ctx_t h, nfai_device_info *info)
{
    CTX_OR_FAIL(c, h);
    if (!info) return fail(NFAI_ERR_INVALID, "device_info: info is null");
    memset(info, 0, sizeof(*info));
    strncpy(info->name, c->prop.name, sizeof(info->name) - 1);
    strncpy(info->arch, c->prop.gcnArchName, sizeof(info->arch) - 1);
    info->total_mem_bytes = c->prop.totalGlobalMem;
    info->compute_units = (uint32_t)c->prop.multiProcessorCount;
    info->wavefront_size = (uint32_t)c->prop.warpSize;
    info->lds_bytes_per_cu = (uint32_t)c->prop.maxSharedMemoryPerMultiProcessor;
    info->clock_khz = (uint32_t)c->prop.clockRate;
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_timer_begin(nfai_ctx_t h)
{
    CTX_OR_FAIL(c, h);
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_timer_end(nfai_ctx_t h, float *ms)
{
    CTX_OR_FAIL(c, h);
    if (!ms) return fail(NFAI_ERR_INVALID, "timer_end: ms is null");
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return NFAI_OK;
}

// ---- buffers ---------------------------------------------------------------------------------
// NFAI_HIP_DEBUG_CANARY=1 (read once): every buffer of nfai_hip_buf_alloc sits between two 256-byte guards filled with a
// pattern; nfai_hip_buf_free and nfai_hip_ctx_synchronize verify them and fail with NFAI_ERR_HIP naming the buffer when a kernel
// (or a copy through a wrapped alias) wrote outside its buffer.  The build's stand-in for the validation layer the reference
// switches on unconditionally (NFAI.Vulkan/VulkanHelper.cs:14-17,126-131; robustBufferAccess is off there too, :206-219).
namespace {
constexpr uint64_t CANARY_BYTES = 256;
constexpr int CANARY_BYTE = 0xC5;
bool canary_on()
{
    static const bool on = getenv("NFAI_HIP_DEBUG_CANARY") && atoi(getenv("NFAI_HIP_DEBUG_CANARY")) != 0;
    return on;
}
std::mutex g_canary_mu;
std::vector<Buf *> g_canary_bufs;

// guards of one buffer; stream must be idle.  0 = intact.
int canary_check(Buf *b, const char *when)
{
    if (!b->guard_base) return NFAI_OK;
    const uint64_t padded = (b->bytes + 255) & ~255ull;
    unsigned char host[2 * CANARY_BYTES];
    HIP_TRY(hipMemcpy(host, b->guard_base, CANARY_BYTES, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(host + CANARY_BYTES, static_cast<char *>(b->ptr) + padded, CANARY_BYTES, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < 2 * CANARY_BYTES; i++)
        if (host[i] != CANARY_BYTE)
            return fail(NFAI_ERR_HIP, "%s: a write outside buffer %p (%llu bytes) was detected: guard byte %llu %s the buffer is 0x%02x (NFAI_HIP_DEBUG_CANARY)",
                        when, b->ptr, (unsigned long long)b->bytes, (unsigned long long)(i % CANARY_BYTES), i < CANARY_BYTES ? "before" : "after", host[i]);
    return NFAI_OK;
}
}  // namespace

static void canary_forget_ctx(Ctx *c)
{
    std::lock_guard<std::mutex> lk(g_canary_mu);
    for (size_t i = 0; i < g_canary_bufs.size();)
        if (g_canary_bufs[i]->ctx == c) {
            g_canary_bufs[i] = g_canary_bufs.back();
            g_canary_bufs.pop_back();
        } else {
            i++;
        }
}

static int canary_check_all(Ctx *c)
{
    if (!canary_on()) return NFAI_OK;
    std::lock_guard<std::mutex> lk(g_canary_mu);
    for (Buf *b : g_canary_bufs)
        if (b->ctx == c) {
            int rc = canary_check(b, "ctx_synchronize");
            if (rc) return rc;
        }
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_buf_alloc(nfai_ctx_t h, uint64_t bytes, nfai_buf_t *out)
{
    CTX_OR_FAIL(c, h);
    if (!out || bytes == 0) return fail(NFAI_ERR_INVALID, "buf_alloc: out null or zero bytes");
    Buf *b = new Buf();
    const uint64_t padded = (bytes + 255) & ~255ull;  // 16-byte vector accesses may touch the tail
    const uint64_t guard = canary_on() ? CANARY_BYTES : 0;
    void *base = nullptr;
    hipError_t e = hipMalloc(&base, padded + 2 * guard);
    if (e != hipSuccess) {
        delete b;
        return fail(NFAI_ERR_OOM, "buf_alloc: hipMalloc(%llu) failed: %s", (unsigned long long)padded, hipGetErrorString(e));
    }
    b->ptr = static_cast<char *>(base) + guard;
    // any failure from here on frees the allocation and the handle object; a guarded buffer is registered only once its guards
    // (and its zero fill) are enqueued
    hipError_t me = hipSuccess;
    if (guard) {
        b->guard_base = base;
        me = hipMemsetAsync(base, CANARY_BYTE, guard, c->stream);
        if (me == hipSuccess) me = hipMemsetAsync(static_cast<char *>(b->ptr) + padded, CANARY_BYTE, guard, c->stream);
    }
    if (me == hipSuccess) me = hipMemsetAsync(b->ptr, 0, padded, c->stream);
    if (me != hipSuccess) {
        (void)hipFree(base);
        delete b;
        return fail(NFAI_ERR_HIP, "buf_alloc: hipMemsetAsync failed: %s", hipGetErrorString(me));
    }
    b->bytes = bytes;
    b->owned = true;
    b->ctx = c;
    if (guard) {
        std::lock_guard<std::mutex> lk(g_canary_mu);
        g_canary_bufs.push_back(b);
    }
    handle_register(b);
    *out = reinterpret_cast<nfai_buf_t>(b);
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_buf_wrap(nfai_ctx_t h, void *device_ptr, uint64_t bytes, nfai_buf_t *out)
{
    CTX_OR_FAIL(c, h);
    if (!out || !device_ptr || bytes == 0) return fail(NFAI_ERR_INVALID, "buf_wrap: null argument");
    if ((reinterpret_cast<uintptr_t>(device_ptr) & 15) != 0) return fail(NFAI_ERR_INVALID, "buf_wrap: pointer not 16-byte aligned");
    Buf *b = new Buf();
    b->ptr = device_ptr;
    b->bytes = bytes;
    b->owned = false;
    b->ctx = c;
    handle_register(b);
    *out = reinterpret_cast<nfai_buf_t>(b);
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_buf_free(nfai_ctx_t h, nfai_buf_t bh)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(b, bh);
    int canary_rc = NFAI_OK;
    if (b->owned) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (b->guard_base) {
            canary_rc = canary_check(b, "buf_free");
            std::lock_guard<std::mutex> lk(g_canary_mu);
            g_canary_bufs.erase(std::remove(g_canary_bufs.begin(), g_canary_bufs.end(), b), g_canary_bufs.end());
        }
        HIP_TRY(hipFree(b->guard_base ? b->guard_base : b->ptr));
    }
    b->magic = 0;
    handle_unregister(b);
    delete b;
    return canary_rc;
}

NFAI_API int32_t nfai_hip_buf_upload(nfai_ctx_t h, nfai_buf_t bh, uint64_t off, const void *host, uint64_t bytes)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(b, bh);
    if (!host || off + bytes > b->bytes) return fail(NFAI_ERR_INVALID, "buf_upload: range [%llu,+%llu) exceeds %llu",
                                                     (unsigned long long)off, (unsigned long long)bytes, (unsigned long long)b->bytes);
    HIP_TRY(hipMemcpyAsync(static_cast<char *>(b->ptr) + off, host, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_buf_download(nfai_ctx_t h, nfai_buf_t bh, uint64_t off, void *host, uint64_t bytes)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(b, bh);
    if (!host || off + bytes > b->bytes) return fail(NFAI_ERR_INVALID, "buf_download: range [%llu,+%llu) exceeds %llu",
                                                     (unsigned long long)off, (unsigned long long)bytes, (unsigned long long)b->bytes);
    HIP_TRY(hipMemcpyAsync(host, static_cast<char *>(b->ptr) + off, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_buf_copy(nfai_ctx_t h, nfai_buf_t dh, uint64_t doff, nfai_buf_t sh, uint64_t soff, uint64_t bytes)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(d, dh);
    BUF_OR_FAIL(s, sh);
    if (doff + bytes > d->bytes || soff + bytes > s->bytes) return fail(NFAI_ERR_INVALID, "buf_copy: range out of bounds");
    HIP_TRY(hipMemcpyAsync(static_cast<char *>(d->ptr) + doff, static_cast<char *>(s->ptr) + soff, bytes,
                           hipMemcpyDeviceToDevice, c->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_buf_zero(nfai_ctx_t h, nfai_buf_t bh)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(b, bh);
    HIP_TRY(hipMemsetAsync(b->ptr, 0, b->bytes, c->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_buf_info(nfai_ctx_t h, nfai_buf_t bh, void **ptr, uint64_t *bytes)
{
    CTX_OR_FAIL(c, h);
    (void)c;
    BUF_OR_FAIL(b, bh);
    if (ptr) *ptr = b->ptr;
    if (bytes) *bytes = b->bytes;
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_weight_bytes(int32_t type, uint64_t n_rows, uint64_t n_cols, uint64_t *bytes)
{
    const uint64_t rb = weight_row_bytes(type, n_cols);
    if (rb == 0) return fail(NFAI_ERR_UNSUPPORTED, "weight_bytes: ggml type %d with %llu columns is not supported", type,
                             (unsigned long long)n_cols);
    if (bytes) *bytes = rb * n_rows;
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_weight_upload(nfai_ctx_t h, int32_t type, uint64_t n_rows, uint64_t n_cols, const void *host,
                                        nfai_buf_t *out)
{
    uint64_t bytes = 0;
    int rc = nfai_hip_weight_bytes(type, n_rows, n_cols, &bytes);
    if (rc) return rc;
    rc = nfai_hip_buf_alloc(h, bytes, out);
    if (rc) return rc;
    const bool t16 = n_rows > 0 && n_rows % 16 == 0;
    const bool q4_t16 = type == NFAI_Q4_K && t16, q6_t16 = type == NFAI_Q6_K && t16;
    if (type != NFAI_Q6_K && !q4_t16) return nfai_hip_buf_upload(h, *out, 0, host, bytes);
    // Q6_K: 210-byte native blocks are repacked into the aligned plane layout the kernels read;
    // Q4_K with a multiple of 16 rows: into the T16 tile layout of kernels_gemv_kqm.hip
    nfai_buf_t tmp = 0;
    if ((rc = nfai_hip_buf_alloc(h, bytes, &tmp))) return rc;
    if ((rc = nfai_hip_buf_upload(h, tmp, 0, host, bytes))) return rc;
    Ctx *c = ctx_of(h);
    hipError_t e;
    if (q4_t16) {
        e = launch_repack_q4k_t16(buf_of(tmp)->ptr, buf_of(*out)->ptr, n_rows, n_cols, c->stream);
        buf_of(*out)->w_layout = NFAI_Q4_K_T16;
        buf_of(*out)->w_rows = n_rows;
    } else if (q6_t16) {
        e = launch_repack_q6k_t16(buf_of(tmp)->ptr, buf_of(*out)->ptr, n_rows, n_cols, c->stream);
        buf_of(*out)->w_layout = NFAI_Q6_K_T16;
        buf_of(*out)->w_rows = n_rows;
    } else {
        e = launch_repack_q6k(buf_of(tmp)->ptr, buf_of(*out)->ptr, n_rows * n_cols / 256, c->stream);
    }
    if (e != hipSuccess) return fail(NFAI_ERR_HIP, "weight_upload: K-quant repack failed: %s", hipGetErrorString(e));
    return nfai_hip_buf_free(h, tmp);
}

// A weight buffer that nfai_hip_weight_upload repacked keeps its internal layout code; the ops must then be
// given the whole tensor (the T16 planes are addressed by its row count).
static int resolve_layout(const char *fn, const Buf *w, int type, uint64_t rows, int *out)
{
    *out = type;
    if (w->w_layout == 0 || ggml_type_of(w->w_layout) != type) return NFAI_OK;
    if (w->w_rows != rows)
        return fail(NFAI_ERR_INVALID, "%s: weight buffer was uploaded with %llu rows (tiled layout), the call uses %llu", fn,
                    (unsigned long long)w->w_rows, (unsigned long long)rows);
    *out = w->w_layout;
    return NFAI_OK;
}

// ---- 1:1 operators ---------------------------------------------------------------------------
#define NEED(b, n_elems, esz)                                                                       \
    if ((uint64_t)(n_elems) * (esz) > (b)->bytes)                                                   \
        return fail(NFAI_ERR_INVALID, "%s: buffer %s holds %llu bytes, needs %llu", __func__, #b,   \
                    (unsigned long long)(b)->bytes, (unsigned long long)((uint64_t)(n_elems) * (esz)))

NFAI_API int32_t nfai_hip_embed(nfai_ctx_t h, nfai_buf_t table, int32_t type, nfai_buf_t tok, nfai_buf_t y, uint32_t E)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bt, table);
    BUF_OR_FAIL(bk, tok);
    BUF_OR_FAIL(by, y);
    NEED(bk, 1, 4);
    NEED(by, E, 4);
    if (type == NFAI_Q4_K || type == NFAI_Q6_K) {
        const uint64_t rb = weight_row_bytes(type, E);
        if (rb == 0) return fail(NFAI_ERR_UNSUPPORTED, "embed: K-quant table needs E %% 256 == 0 (E=%u)", E);
        int lt;
        { int rc = resolve_layout(__func__, bt, type, bt->bytes / rb, &lt); if (rc) return rc; }
        LAUNCH_TRY(launch_embed_kq(bt->ptr, lt, bt->bytes / rb, static_cast<const uint32_t *>(bk->ptr), static_cast<float *>(by->ptr), E, c->stream));
        return NFAI_OK;
    }
    if (type != NFAI_F16 && type != NFAI_F32) return fail(NFAI_ERR_UNSUPPORTED, "embed: table type %d", type);
    LAUNCH_TRY(launch_embed(bt->ptr, type, static_cast<const uint32_t *>(bk->ptr), static_cast<float *>(by->ptr), E, c->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_rmsnorm(nfai_ctx_t h, nfai_buf_t x, nfai_buf_t g, nfai_buf_t y, uint32_t E, float eps)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bx, x);
    BUF_OR_FAIL(bg, g);
    BUF_OR_FAIL(by, y);
    NEED(bx, E, 4);
    NEED(bg, E, 4);
    NEED(by, E, 4);
    LAUNCH_TRY(launch_rmsnorm(static_cast<const float *>(bx->ptr), static_cast<const float *>(bg->ptr),
                              static_cast<float *>(by->ptr), E, eps, c->stream));
    return NFAI_OK;
}

static int gemv_common(Ctx *c, GemvArgs &a, const char *fn)
{
    a.n_cu = (uint32_t)c->prop.multiProcessorCount;
    hipError_t e = launch_gemv(a, c->stream);
    if (e == hipErrorInvalidValue)
        return fail(NFAI_ERR_INVALID, "%s: unsupported GEMV shape/type (type %d, K %u; fp16 needs K %% 8 == 0)", fn, a.w_type, a.K);
    if (e != hipSuccess) return fail(NFAI_ERR_HIP, "%s: launch failed: %s", fn, hipGetErrorString(e));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_gemv(nfai_ctx_t h, nfai_buf_t W, int32_t type, nfai_buf_t x, nfai_buf_t y, uint64_t y_off,
                               uint32_t N, uint32_t K)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bw, W);
    BUF_OR_FAIL(bx, x);
    BUF_OR_FAIL(by, y);
    const uint64_t rb = weight_row_bytes(type, K);
    if (rb == 0) return fail(NFAI_ERR_UNSUPPORTED, "gemv: ggml type %d with K=%u is not supported", type, K);
    NEED(bw, (uint64_t)N * rb, 1);
    NEED(bx, K, 4);
    NEED(by, y_off + N, 4);
    GemvArgs a;
    a.W[0] = bw->ptr;
    a.seg_rows[0] = N;
    { int rc = resolve_layout(__func__, bw, type, N, &a.w_type); if (rc) return rc; }
    a.x = static_cast<const float *>(bx->ptr);
    a.K = K;
    a.mode = GEMV_PLAIN;
    a.y = static_cast<float *>(by->ptr) + y_off;
    return gemv_common(c, a, __func__);
}

NFAI_API int32_t nfai_hip_rope(nfai_ctx_t h, nfai_buf_t in, uint64_t in_off, nfai_buf_t out, uint64_t out_off,
                               nfai_buf_t freqs, uint32_t rope_dims, uint32_t n_heads, uint32_t head_dim, uint32_t pos)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bi, in);
    BUF_OR_FAIL(bo, out);
    BUF_OR_FAIL(bf, freqs);
    if (head_dim % 2) return fail(NFAI_ERR_INVALID, "rope: odd head_dim");
    const uint64_t n = (uint64_t)n_heads * head_dim;
    NEED(bi, in_off + n, 4);
    NEED(bo, out_off + n, 4);
    NEED(bf, (rope_dims < head_dim ? rope_dims : head_dim) / 2, 4);
    LAUNCH_TRY(launch_rope(static_cast<const float *>(bi->ptr) + in_off, static_cast<float *>(bo->ptr) + out_off,
                           static_cast<const float *>(bf->ptr), rope_dims, n_heads, head_dim, pos, c->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_attn_scores(nfai_ctx_t h, nfai_buf_t q, nfai_buf_t kc, nfai_buf_t s, uint32_t H, uint32_t Hkv,
                                      uint32_t D, uint32_t S)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bq, q);
    BUF_OR_FAIL(bk, kc);
    BUF_OR_FAIL(bs, s);
    if (Hkv == 0 || H % Hkv || S == 0) return fail(NFAI_ERR_INVALID, "attn_scores: H=%u Hkv=%u S=%u", H, Hkv, S);
    NEED(bq, (uint64_t)H * D, 4);
    NEED(bk, (uint64_t)S * Hkv * D, 4);
    NEED(bs, (uint64_t)H * S, 4);
    LAUNCH_TRY(launch_attn_scores(static_cast<const float *>(bq->ptr), static_cast<const float *>(bk->ptr),
                                  static_cast<float *>(bs->ptr), H, Hkv, D, S, c->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_attn_softmax(nfai_ctx_t h, nfai_buf_t s, nfai_buf_t w, uint32_t H, uint32_t S, float eps)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bs, s);
    BUF_OR_FAIL(bw, w);
    if (S == 0) return fail(NFAI_ERR_INVALID, "attn_softmax: S=0");
    NEED(bs, (uint64_t)H * S, 4);
    NEED(bw, (uint64_t)H * S, 4);
    LAUNCH_TRY(launch_attn_softmax(static_cast<const float *>(bs->ptr), static_cast<float *>(bw->ptr), H, S, eps, c->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_attn_wsum(nfai_ctx_t h, nfai_buf_t w, nfai_buf_t vc, nfai_buf_t o, uint32_t H, uint32_t Hkv,
                                    uint32_t D, uint32_t S)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bw, w);
    BUF_OR_FAIL(bv, vc);
    BUF_OR_FAIL(bo, o);
    if (Hkv == 0 || H % Hkv || S == 0) return fail(NFAI_ERR_INVALID, "attn_wsum: H=%u Hkv=%u S=%u", H, Hkv, S);
    NEED(bw, (uint64_t)H * S, 4);
    NEED(bv, (uint64_t)S * Hkv * D, 4);
    NEED(bo, (uint64_t)H * D, 4);
    LAUNCH_TRY(launch_attn_wsum(static_cast<const float *>(bw->ptr), static_cast<const float *>(bv->ptr),
                                static_cast<float *>(bo->ptr), H, Hkv, D, S, c->stream));
    return NFAI_OK;
}

#define ELTWISE2(name, launch)                                                                              \
    NFAI_API int32_t name(nfai_ctx_t h, nfai_buf_t a, nfai_buf_t b, nfai_buf_t y, uint32_t n)                \
    {                                                                                                       \
        CTX_OR_FAIL(c, h);                                                                                  \
        BUF_OR_FAIL(ba, a);                                                                                 \
        BUF_OR_FAIL(bb, b);                                                                                 \
        BUF_OR_FAIL(by, y);                                                                                 \
        NEED(ba, n, 4);                                                                                     \
        NEED(bb, n, 4);                                                                                     \
        NEED(by, n, 4);                                                                                     \
        LAUNCH_TRY(launch(static_cast<const float *>(ba->ptr), static_cast<const float *>(bb->ptr),        \
                          static_cast<float *>(by->ptr), n, c->stream));                                    \
        return NFAI_OK;                                                                                     \
    }
ELTWISE2(nfai_hip_mul, launch_mul)
ELTWISE2(nfai_hip_add, launch_add)

NFAI_API int32_t nfai_hip_silu(nfai_ctx_t h, nfai_buf_t x, nfai_buf_t y, uint32_t n)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bx, x);
    BUF_OR_FAIL(by, y);
    NEED(bx, n, 4);
    NEED(by, n, 4);
    LAUNCH_TRY(launch_silu(static_cast<const float *>(bx->ptr), static_cast<float *>(by->ptr), n, c->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_argmax(nfai_ctx_t h, nfai_buf_t x, uint32_t n, nfai_buf_t out_idx)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bx, x);
    BUF_OR_FAIL(bo, out_idx);
    if (n == 0) return fail(NFAI_ERR_INVALID, "argmax: n=0");
    NEED(bx, n, 4);
    NEED(bo, 1, 4);
    LAUNCH_TRY(launch_argmax(static_cast<const float *>(bx->ptr), n, static_cast<uint32_t *>(bo->ptr), static_cast<char *>(c->scratch) + SCR_ARGMAX, nullptr,
                             nullptr, 0, c->stream));
    return NFAI_OK;
}

// ---- candidates of SamplingUtils.TopP (SamplingUtils.cs:5-33) -------------------------------------------------------------------
namespace nfai {
// Host half: what the device left (k largest logits, descending, ties by index; M = max(l / T); S = sum exp(l / T - M)) -> the
// (index, probability) pairs OrderByDescending(Prob).Take(k) yields (:9-13): p = exp(l / T - M) / S (:7, :38-40), ordered by
// probability descending, equal probabilities by index (the reference's sort is stable over the index order).
void topk_finish(const float *vals, const uint32_t *ids, float M, float S, float temperature, uint32_t k, uint32_t *ids_out, float *probs_out)
{
    for (uint32_t j = 0; j < k; j++) {
        const float p = expf(vals[j] / temperature - M) / S;
        uint32_t at = j;  // insertion: exp is monotonic, so only runs of EQUAL probabilities (different logits) can need reordering
        while (at > 0 && probs_out[at - 1] == p && ids_out[at - 1] > ids[j]) {
            probs_out[at] = probs_out[at - 1];
            ids_out[at] = ids_out[at - 1];
            at--;
        }
        probs_out[at] = p;
        ids_out[at] = ids[j];
    }
}

int topk_run(Ctx *c, const float *logits_dev, uint32_t n, float temperature, uint32_t k, void *work, uint32_t *ids_out, float *probs_out)
{
    if (!ids_out || !probs_out) return fail(NFAI_ERR_INVALID, "topk: null output");
    if (k == 0 || k > TOPK_MAX || k > n) return fail(NFAI_ERR_INVALID, "topk: k=%u outside [1, min(%u, n=%u)]", k, TOPK_MAX, n);
    if (!(temperature > 0.f)) return fail(NFAI_ERR_INVALID, "topk: temperature %g (the reference divides by it, SamplingUtils.cs:7)", temperature);
    hipError_t e = launch_topk(logits_dev, n, temperature, k, work, c->stream);
    if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? NFAI_ERR_INVALID : NFAI_ERR_HIP, "topk launch failed: %s", hipGetErrorString(e));
    struct { float v[TOPK_MAX]; uint32_t i[TOPK_MAX]; float M, S; } out;  // the head of the workspace behind its ticket words (320 + 8 bytes used at k = 40)
    HIP_TRY(hipMemcpyAsync(&out, static_cast<const char *>(work) + topk_out_offset(), sizeof(out), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    topk_finish(out.v, out.i, out.M, out.S, temperature, k, ids_out, probs_out);
    return NFAI_OK;
}
}  // namespace nfai

NFAI_API int32_t nfai_hip_topk(nfai_ctx_t h, nfai_buf_t x, uint32_t n, float temperature, uint32_t k, uint32_t *ids_out, float *probs_out)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bx, x);
    if (n == 0) return fail(NFAI_ERR_INVALID, "topk: n=0");
    NEED(bx, n, 4);
    if (topk_work_bytes(n) > SCR_TOPK_END - SCR_TOPK) return fail(NFAI_ERR_INVALID, "topk: n=%u too large for the context scratch", n);
    return topk_run(c, static_cast<const float *>(bx->ptr), n, temperature, k, static_cast<char *>(c->scratch) + SCR_TOPK, ids_out, probs_out);
}

// ---- fused operators -------------------------------------------------------------------------
// Scalars the fused kernels read from device memory (so graphs can be replayed) live at the END
// of the context scratch area when the op-level entry points are used.
static uint32_t *scratch_pos(Ctx *c) { return reinterpret_cast<uint32_t *>(static_cast<char *>(c->scratch) + SCR_POS); }
static float *scratch_ropecs(Ctx *c) { return reinterpret_cast<float *>(static_cast<char *>(c->scratch) + SCR_ROPECS); }
static float *scratch_attn(Ctx *c) { return reinterpret_cast<float *>(static_cast<char *>(c->scratch) + SCR_ATTN); }
static bool scratch_attn_fits(uint32_t H, uint32_t Hkv, uint32_t D) { return attn_partials_bytes(H, Hkv, D) <= SCR_ATTN_END - SCR_ATTN; }

NFAI_API int32_t nfai_hip_attn_decode(nfai_ctx_t h, nfai_buf_t q, nfai_buf_t kc, nfai_buf_t vc, nfai_buf_t o, uint32_t H,
                                      uint32_t Hkv, uint32_t D, uint32_t S, uint32_t C, int32_t kv_type)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bq, q);
    BUF_OR_FAIL(bk, kc);
    BUF_OR_FAIL(bv, vc);
    BUF_OR_FAIL(bo, o);
    if (S == 0 || S > C) return fail(NFAI_ERR_INVALID, "attn_decode: S=%u outside (0, C=%u]", S, C);
    if (kv_type != NFAI_F32 && kv_type != NFAI_F16) return fail(NFAI_ERR_UNSUPPORTED, "attn_decode: kv type %d", kv_type);
    const uint32_t esz = kv_type == NFAI_F16 ? 2 : 4;
    NEED(bq, (uint64_t)H * D, 4);
    NEED(bk, (uint64_t)S * Hkv * D, esz);
    NEED(bv, (uint64_t)S * Hkv * D, esz);
    NEED(bo, (uint64_t)H * D, 4);
    if (!scratch_attn_fits(H, Hkv, D)) return fail(NFAI_ERR_INVALID, "attn_decode: H*D too large for scratch");
    const uint32_t pos = S - 1;
    HIP_TRY(hipMemcpyAsync(scratch_pos(c), &pos, 4, hipMemcpyHostToDevice, c->stream));
    AttnArgs a;
    a.q = static_cast<const float *>(bq->ptr);
    a.kcache = bk->ptr;
    a.vcache = bv->ptr;
    a.kv_type = kv_type;
    a.kv_pos_stride = (uint64_t)Hkv * D;
    a.kv_head_stride = D;
    a.o = static_cast<float *>(bo->ptr);
    a.H = H; a.Hkv = Hkv; a.D = D; a.C = C;
    a.pos_dev = scratch_pos(c);
    a.partials = scratch_attn(c);
    hipError_t e = launch_attn_decode(a, c->stream);
    if (e == hipErrorInvalidValue) return fail(NFAI_ERR_INVALID, "attn_decode: unsupported shape (D must be 64 or 128, H/Hkv <= 8, C <= 32768)");
    if (e != hipSuccess) return fail(NFAI_ERR_HIP, "attn_decode: launch failed: %s", hipGetErrorString(e));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_gemm_f16(nfai_ctx_t h, nfai_buf_t A, nfai_buf_t W, nfai_buf_t R, nfai_buf_t C, uint32_t M, uint32_t N, uint32_t K,
                                   int32_t variant)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(ba, A);
    BUF_OR_FAIL(bw, W);
    BUF_OR_FAIL(bc, C);
    Buf *br = R ? buf_of(R) : nullptr;
    if (R && !br) return fail(NFAI_ERR_INVALID, "gemm_f16: invalid residual handle");
    if (M == 0 || N % 64 || K % 64 || K == 0) return fail(NFAI_ERR_INVALID, "gemm_f16: needs M > 0, N %% 64 == 0, K %% 64 == 0 (M=%u N=%u K=%u)", M, N, K);
    NEED(ba, (uint64_t)M * K, 2);
    NEED(bw, (uint64_t)N * K, 2);
    NEED(bc, (uint64_t)M * N, 4);
    if (br) NEED(br, (uint64_t)M * N, 4);
    GemmArgs g;
    g.A = ba->ptr; g.lda = K; g.B = bw->ptr; g.ldb = K; g.C = bc->ptr; g.ldc = N;
    g.R = br ? static_cast<const float *>(br->ptr) : nullptr;
    g.M = M; g.N = N; g.K = K; g.variant = variant;
    g.n_cu = (uint32_t)c->prop.multiProcessorCount;
    hipError_t e = launch_gemm_f16(g, c->stream);
    if (e == hipErrorInvalidValue) return fail(NFAI_ERR_INVALID, "gemm_f16: unsupported shape / variant %d (M=%u N=%u K=%u)", variant, M, N, K);
    if (e != hipSuccess) return fail(NFAI_ERR_HIP, "gemm_f16: launch failed: %s", hipGetErrorString(e));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_attn_prefill(nfai_ctx_t h, nfai_buf_t Q, nfai_buf_t K, nfai_buf_t Vt, nfai_buf_t O, uint32_t T, uint32_t H, uint32_t Hkv,
                                       uint32_t D, uint32_t Spad, uint32_t pos0)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bq, Q);
    BUF_OR_FAIL(bk, K);
    BUF_OR_FAIL(bv, Vt);
    BUF_OR_FAIL(bo, O);
    if (T == 0 || Hkv == 0 || H % Hkv || (D != 64 && D != 128) || Spad % 32 || (uint64_t)pos0 + T > Spad)
        return fail(NFAI_ERR_INVALID, "attn_prefill: bad shape (T=%u H=%u Hkv=%u D=%u Spad=%u pos0=%u)", T, H, Hkv, D, Spad, pos0);
    NEED(bq, (uint64_t)T * H * D, 2);
    NEED(bk, (uint64_t)Hkv * Spad * D, 2);
    NEED(bv, (uint64_t)Hkv * Spad * D, 2);
    NEED(bo, (uint64_t)T * H * D, 2);
    hipError_t e = launch_attn_prefill(bq->ptr, bk->ptr, bv->ptr, bo->ptr, T, H, Hkv, D, Spad, pos0, c->stream);
    if (e != hipSuccess) return fail(NFAI_ERR_HIP, "attn_prefill: launch failed: %s", hipGetErrorString(e));
    return NFAI_OK;
}

// Extended form of nfai_hip_gemm_f16 for tests and tools: every epilogue (fp32 + residual, fp16, SiLU*up fp16), head-batched
// operands (batch, b_div) and the causal tile skipping of the attention GEMMs — the configurations the MFMA prefill launches.
//   A [batch][M][K] fp16;  W [batch / b_div][N][K] fp16 (epi 2: W = gate [N/2][K], W1 = up [N/2][K], batch 1);
//   C [batch][M][N] fp32 (epi 0) / fp16 (epi 1), [M][N/2] fp16 (epi 2);  R as C (epi 0 only).
NFAI_API int32_t nfai_hip_gemm_f16_ex(nfai_ctx_t h, nfai_buf_t A, nfai_buf_t W, nfai_buf_t W1, nfai_buf_t R, nfai_buf_t C, uint32_t M,
                                      uint32_t N, uint32_t K, int32_t variant, int32_t epi, uint32_t batch, uint32_t b_div,
                                      uint32_t causal, uint32_t causal_pos0)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(ba, A);
    BUF_OR_FAIL(bw, W);
    BUF_OR_FAIL(bc, C);
    Buf *br = R ? buf_of(R) : nullptr, *bw1 = W1 ? buf_of(W1) : nullptr;
    if ((R && !br) || (W1 && !bw1)) return fail(NFAI_ERR_INVALID, "gemm_f16_ex: invalid residual / second weight handle");
    if (M == 0 || N % 64 || K % 64 || K == 0 || batch == 0 || b_div == 0 || batch % b_div || epi < 0 || epi > 2 || causal > 2)
        return fail(NFAI_ERR_INVALID, "gemm_f16_ex: bad shape (M=%u N=%u K=%u batch=%u b_div=%u epi=%d causal=%u)", M, N, K, batch, b_div, epi, causal);
    if (epi == 2 && (!bw1 || batch != 1 || br)) return fail(NFAI_ERR_INVALID, "gemm_f16_ex: the SiLU*up epilogue takes gate and up matrices, batch 1, no residual");
    if (epi == 1 && br) return fail(NFAI_ERR_INVALID, "gemm_f16_ex: the fp16 epilogue takes no residual");
    const uint64_t nw = epi == 2 ? N / 2 : N;
    NEED(ba, (uint64_t)batch * M * K, 2);
    NEED(bw, (uint64_t)(batch / b_div) * nw * K, 2);
    if (bw1) NEED(bw1, nw * K, 2);
    NEED(bc, (uint64_t)batch * M * nw, epi == 0 ? 4 : 2);
    if (br) NEED(br, (uint64_t)batch * M * N, 4);
    GemmArgs g;
    g.A = ba->ptr; g.lda = K; g.a_bs = (uint64_t)M * K;
    g.B = bw->ptr; g.ldb = K; g.b_bs = (uint64_t)N * K; g.b_div = b_div;
    if (epi == 2) { g.B1 = bw1->ptr; g.n0 = N / 2; }
    g.C = bc->ptr; g.ldc = (uint32_t)nw; g.c_bs = (uint64_t)M * nw; g.epi = epi;
    g.R = br ? static_cast<const float *>(br->ptr) : nullptr;
    g.M = M; g.N = N; g.K = K; g.batch = batch; g.variant = variant;
    g.causal = causal; g.causal_pos0 = causal_pos0;
    g.n_cu = (uint32_t)c->prop.multiProcessorCount;
    hipError_t e = launch_gemm_f16(g, c->stream);
    if (e == hipErrorInvalidValue) return fail(NFAI_ERR_INVALID, "gemm_f16_ex: unsupported shape / variant %d (M=%u N=%u K=%u epi=%d)", variant, M, N, K, epi);
    if (e != hipSuccess) return fail(NFAI_ERR_HIP, "gemm_f16_ex: launch failed: %s", hipGetErrorString(e));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_gemm_kq(nfai_ctx_t h, nfai_buf_t A, nfai_buf_t W, int32_t type, nfai_buf_t R, nfai_buf_t C, uint32_t M, uint32_t N,
                                  uint32_t K)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(ba, A);
    BUF_OR_FAIL(bw, W);
    BUF_OR_FAIL(bc, C);
    Buf *br = R ? buf_of(R) : nullptr;
    if (R && !br) return fail(NFAI_ERR_INVALID, "gemm_kq: invalid residual handle");
    if (M == 0 || N % 64 || K % 256 || K == 0) return fail(NFAI_ERR_INVALID, "gemm_kq: needs M > 0, N %% 64 == 0, K %% 256 == 0 (M=%u N=%u K=%u)", M, N, K);
    int lt;
    { int rc = resolve_layout(__func__, bw, type, N, &lt); if (rc) return rc; }
    if (lt != NFAI_Q4_K_T16 && lt != NFAI_Q6_K_T16)
        return fail(NFAI_ERR_UNSUPPORTED, "gemm_kq: W must be a Q4_K / Q6_K buffer uploaded whole with a multiple of 16 rows");
    NEED(ba, (uint64_t)M * K, 2);
    NEED(bc, (uint64_t)M * N, 4);
    if (br) NEED(br, (uint64_t)M * N, 4);
    GemmArgs g;
    g.A = ba->ptr; g.lda = K; g.B = bw->ptr; g.ldb = K; g.C = bc->ptr; g.ldc = N; g.b_type = lt;
    g.R = br ? static_cast<const float *>(br->ptr) : nullptr;
    g.M = M; g.N = N; g.K = K;
    g.n_cu = (uint32_t)c->prop.multiProcessorCount;
    hipError_t e = launch_gemm_kq(g, c->stream);
    if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? NFAI_ERR_INVALID : NFAI_ERR_HIP, "gemm_kq: launch failed: %s", hipGetErrorString(e));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_gemv_fused(nfai_ctx_t h, nfai_buf_t W, int32_t type, nfai_buf_t x, nfai_buf_t gamma, float eps,
                                     nfai_buf_t res, nfai_buf_t y, uint32_t N, uint32_t K)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bw, W);
    BUF_OR_FAIL(bx, x);
    BUF_OR_FAIL(by, y);
    Buf *bg = gamma ? buf_of(gamma) : nullptr;
    Buf *br = res ? buf_of(res) : nullptr;
    if ((gamma && !bg) || (res && !br)) return fail(NFAI_ERR_INVALID, "gemv_fused: invalid gamma/res handle");
    const uint64_t rb = weight_row_bytes(type, K);
    if (rb == 0) return fail(NFAI_ERR_UNSUPPORTED, "gemv_fused: ggml type %d with K=%u is not supported", type, K);
    NEED(bw, (uint64_t)N * rb, 1);
    NEED(bx, K, 4);
    NEED(by, N, 4);
    if (bg) NEED(bg, K, 4);
    if (br) NEED(br, N, 4);
    GemvArgs a;
    a.W[0] = bw->ptr;
    a.seg_rows[0] = N;
    { int rc = resolve_layout(__func__, bw, type, N, &a.w_type); if (rc) return rc; }
    a.x = static_cast<const float *>(bx->ptr);
    a.gamma = bg ? static_cast<const float *>(bg->ptr) : nullptr;
    a.eps = eps;
    a.K = K;
    a.mode = br ? GEMV_RESIDUAL : GEMV_PLAIN;
    a.res = br ? static_cast<const float *>(br->ptr) : nullptr;
    a.y = static_cast<float *>(by->ptr);
    return gemv_common(c, a, __func__);
}

// output RMSNorm -> lm_head -> SamplingUtils.ArgMax in ONE launch (LlamaModel.cs:123-125 + SamplingUtils.cs:43-57):
// logits (V floats) and the first index of their maximum (one uint32).  The model's token graph ends with this launch.
NFAI_API int32_t nfai_hip_lmhead_argmax(nfai_ctx_t h, nfai_buf_t W, int32_t type, nfai_buf_t x, nfai_buf_t gamma, float eps, nfai_buf_t logits,
                                        nfai_buf_t out_idx, uint32_t V, uint32_t E)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bw, W);
    BUF_OR_FAIL(bx, x);
    BUF_OR_FAIL(by, logits);
    BUF_OR_FAIL(bo, out_idx);
    Buf *bg = gamma ? buf_of(gamma) : nullptr;
    if (gamma && !bg) return fail(NFAI_ERR_INVALID, "lmhead_argmax: invalid gamma handle");
    const uint64_t rb = weight_row_bytes(type, E);
    if (rb == 0) return fail(NFAI_ERR_UNSUPPORTED, "lmhead_argmax: ggml type %d with E=%u is not supported", type, E);
    NEED(bw, (uint64_t)V * rb, 1);
    NEED(bx, E, 4);
    NEED(by, V, 4);
    NEED(bo, 1, 4);
    if (bg) NEED(bg, E, 4);
    GemvArgs a;
    a.W[0] = bw->ptr;
    a.seg_rows[0] = V;
    { int rc = resolve_layout(__func__, bw, type, V, &a.w_type); if (rc) return rc; }
    if (a.w_type != NFAI_F16 && a.w_type != NFAI_F32 && a.w_type != NFAI_Q4_K_T16 && a.w_type != NFAI_Q6_K_T16)
        return fail(NFAI_ERR_UNSUPPORTED, "lmhead_argmax: K-quant tables need a row count that is a multiple of 16 (V=%u)", V);
    a.x = static_cast<const float *>(bx->ptr);
    a.gamma = bg ? static_cast<const float *>(bg->ptr) : nullptr;
    a.eps = eps;
    a.K = E;
    a.mode = GEMV_PLAIN;
    a.y = static_cast<float *>(by->ptr);
    // workspace: its own range of the context scratch (ticket word zero between launches)
    a.argmax_part = static_cast<char *>(c->scratch) + SCR_LMHEAD;
    a.argmax_out = static_cast<uint32_t *>(bo->ptr);
    return gemv_common(c, a, __func__);
}

NFAI_API int32_t nfai_hip_gemv_gateup_silu(nfai_ctx_t h, nfai_buf_t Wg, nfai_buf_t Wu, int32_t type, nfai_buf_t x,
                                           nfai_buf_t gamma, float eps, nfai_buf_t y, uint32_t F, uint32_t K)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bg_, Wg);
    BUF_OR_FAIL(bu, Wu);
    BUF_OR_FAIL(bx, x);
    BUF_OR_FAIL(by, y);
    Buf *bg = gamma ? buf_of(gamma) : nullptr;
    if (gamma && !bg) return fail(NFAI_ERR_INVALID, "gemv_gateup_silu: invalid gamma handle");
    const uint64_t rb = weight_row_bytes(type, K);
    if (rb == 0) return fail(NFAI_ERR_UNSUPPORTED, "gemv_gateup_silu: ggml type %d with K=%u is not supported", type, K);
    NEED(bg_, (uint64_t)F * rb, 1);
    NEED(bu, (uint64_t)F * rb, 1);
    NEED(bx, K, 4);
    NEED(by, F, 4);
    if (bg) NEED(bg, K, 4);
    GemvArgs a;
    a.W[0] = bg_->ptr;
    a.W[1] = bu->ptr;
    a.seg_rows[0] = a.seg_rows[1] = F;
    {
        int t0, t1, rc;
        if ((rc = resolve_layout(__func__, bg_, type, F, &t0)) || (rc = resolve_layout(__func__, bu, type, F, &t1))) return rc;
        if (t0 != t1) return fail(NFAI_ERR_INVALID, "%s: gate and up weights were uploaded in different layouts", __func__);
        a.w_type = t0;
    }
    a.x = static_cast<const float *>(bx->ptr);
    a.gamma = bg ? static_cast<const float *>(bg->ptr) : nullptr;
    a.eps = eps;
    a.K = K;
    a.mode = GEMV_GATEUP;
    a.y = static_cast<float *>(by->ptr);
    return gemv_common(c, a, __func__);
}

NFAI_API int32_t nfai_hip_gemv_qkv_rope(nfai_ctx_t h, nfai_buf_t Wq, nfai_buf_t Wk, nfai_buf_t Wv, int32_t type,
                                        nfai_buf_t x, nfai_buf_t gamma, float eps, nfai_buf_t freqs, uint32_t rope_dims,
                                        nfai_buf_t q, nfai_buf_t kc, nfai_buf_t vc, uint32_t H, uint32_t Hkv, uint32_t D,
                                        uint32_t pos, int32_t kv_type, uint32_t E)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bq_, Wq);
    BUF_OR_FAIL(bk_, Wk);
    BUF_OR_FAIL(bv_, Wv);
    BUF_OR_FAIL(bx, x);
    BUF_OR_FAIL(bf, freqs);
    BUF_OR_FAIL(bq, q);
    BUF_OR_FAIL(bk, kc);
    BUF_OR_FAIL(bv, vc);
    Buf *bg = gamma ? buf_of(gamma) : nullptr;
    if (gamma && !bg) return fail(NFAI_ERR_INVALID, "gemv_qkv_rope: invalid gamma handle");
    if (kv_type != NFAI_F32 && kv_type != NFAI_F16) return fail(NFAI_ERR_UNSUPPORTED, "gemv_qkv_rope: kv type %d", kv_type);
    const uint64_t rb = weight_row_bytes(type, E);
    if (rb == 0) return fail(NFAI_ERR_UNSUPPORTED, "gemv_qkv_rope: ggml type %d with K=%u is not supported", type, E);
    const uint32_t esz = kv_type == NFAI_F16 ? 2 : 4;
    const uint32_t nfreq = (rope_dims < D ? rope_dims : D) / 2;
    NEED(bq_, (uint64_t)H * D * rb, 1);
    NEED(bk_, (uint64_t)Hkv * D * rb, 1);
    NEED(bv_, (uint64_t)Hkv * D * rb, 1);
    NEED(bx, E, 4);
    NEED(bf, nfreq, 4);
    NEED(bq, (uint64_t)H * D, 4);
    NEED(bk, ((uint64_t)pos + 1) * Hkv * D, esz);
    NEED(bv, ((uint64_t)pos + 1) * Hkv * D, esz);
    if (bg) NEED(bg, E, 4);
    if (nfreq * 8 > 3072) return fail(NFAI_ERR_INVALID, "gemv_qkv_rope: rope_dims too large");
    HIP_TRY(hipMemcpyAsync(scratch_pos(c), &pos, 4, hipMemcpyHostToDevice, c->stream));
    LAUNCH_TRY(launch_token_begin(nullptr, 0, nullptr, nullptr, 0, static_cast<const float *>(bf->ptr), scratch_ropecs(c), nfreq,
                                  scratch_pos(c), c->stream));
    GemvArgs a;
    a.W[0] = bq_->ptr; a.W[1] = bk_->ptr; a.W[2] = bv_->ptr;
    a.seg_rows[0] = H * D; a.seg_rows[1] = Hkv * D; a.seg_rows[2] = Hkv * D;
    {
        int t0, t1, t2, rc;
        if ((rc = resolve_layout(__func__, bq_, type, H * D, &t0)) || (rc = resolve_layout(__func__, bk_, type, Hkv * D, &t1)) ||
            (rc = resolve_layout(__func__, bv_, type, Hkv * D, &t2)))
            return rc;
        if (t0 != t1 || t0 != t2) return fail(NFAI_ERR_INVALID, "%s: q, k, v weights were uploaded in different layouts", __func__);
        a.w_type = t0;
    }
    a.x = static_cast<const float *>(bx->ptr);
    a.gamma = bg ? static_cast<const float *>(bg->ptr) : nullptr;
    a.eps = eps;
    a.K = E;
    a.mode = GEMV_QKV_ROPE;
    a.y = static_cast<float *>(bq->ptr);
    a.kcache = bk->ptr; a.vcache = bv->ptr;
    a.kv_type = kv_type;
    a.kv_pos_stride = (uint64_t)Hkv * D;
    a.kv_head_stride = D;
    a.rope_cs = scratch_ropecs(c);
    a.rope_dims = rope_dims; a.H = H; a.Hkv = Hkv; a.D = D;
    a.pos_dev = scratch_pos(c);
    return gemv_common(c, a, __func__);
}

// One launch of the weight-streaming engine on caller-held buffers (tests / tools): what nfai_hip_llama_* enqueues per block when
// NFAI_LLAMA_ENGINE is set.  `scratch` holds the hand-off granules h (E) | act (F) | x (E) (8 bytes each: value, tag) followed by
// 64 bytes of control words (epoch, error); the caller zeroes it once and may read the granules' low words afterwards.
NFAI_API int32_t nfai_hip_engine_block(nfai_ctx_t h, nfai_buf_t Wo, nfai_buf_t Wgate, nfai_buf_t Wup, nfai_buf_t Wdown, nfai_buf_t att,
                                       nfai_buf_t x_in, nfai_buf_t gamma_ffn, float eps, uint32_t E, uint32_t F, uint32_t HD,
                                       nfai_buf_t Wq, nfai_buf_t Wk, nfai_buf_t Wv, nfai_buf_t gamma_next, nfai_buf_t freqs,
                                       uint32_t rope_dims, nfai_buf_t q_out, nfai_buf_t kc, nfai_buf_t vc, uint32_t H, uint32_t Hkv,
                                       uint32_t D, uint32_t pos, int32_t kv_type, nfai_buf_t x_out, nfai_buf_t scratch)
{
    CTX_OR_FAIL(c, h);
    BUF_OR_FAIL(bwo, Wo);
    BUF_OR_FAIL(bwg, Wgate);
    BUF_OR_FAIL(bwu, Wup);
    BUF_OR_FAIL(bwd, Wdown);
    BUF_OR_FAIL(batt, att);
    BUF_OR_FAIL(bx, x_in);
    BUF_OR_FAIL(bg, gamma_ffn);
    BUF_OR_FAIL(bxo, x_out);
    BUF_OR_FAIL(bs, scratch);
    if (E % 512 || F % 512 || HD % 512) return fail(NFAI_ERR_INVALID, "engine_block: E, F, H*D must be multiples of 512");
    NEED(bwo, (uint64_t)E * HD, 2);
    NEED(bwg, (uint64_t)F * E, 2);
    NEED(bwu, (uint64_t)F * E, 2);
    NEED(bwd, (uint64_t)E * F, 2);
    NEED(batt, HD, 4);
    NEED(bx, E, 4);
    NEED(bg, E, 4);
    NEED(bxo, E, 4);
    const uint64_t ngran = 2ull * E + F;
    NEED(bs, ngran * 8 + 64, 1);
    uint64_t *gr = static_cast<uint64_t *>(bs->ptr);
    uint32_t *words = reinterpret_cast<uint32_t *>(gr + ngran);
    EngineArgs e;
    e.E = E; e.F = F; e.HD = HD;
    e.Wo = bwo->ptr; e.Wgate = bwg->ptr; e.Wup = bwu->ptr; e.Wdown = bwd->ptr;
    e.att = static_cast<const float *>(batt->ptr);
    e.x_in = static_cast<const float *>(bx->ptr);
    e.gamma_ffn = static_cast<const float *>(bg->ptr);
    e.eps = eps;
    e.g_h = gr; e.g_act = gr + E; e.g_x = gr + E + F;
    e.epoch = words; e.err = words + 1;
    e.x_out = static_cast<float *>(bxo->ptr);
    e.n_cu = (uint32_t)c->prop.multiProcessorCount;
    e.n_ops = 3;
    const float *fr = nullptr;
    uint32_t nfreq = 0;
    if (Wq) {
        BUF_OR_FAIL(bq_, Wq);
        BUF_OR_FAIL(bk_, Wk);
        BUF_OR_FAIL(bv_, Wv);
        BUF_OR_FAIL(bgn, gamma_next);
        BUF_OR_FAIL(bf, freqs);
        BUF_OR_FAIL(bq, q_out);
        BUF_OR_FAIL(bk, kc);
        BUF_OR_FAIL(bv, vc);
        if (kv_type != NFAI_F32 && kv_type != NFAI_F16) return fail(NFAI_ERR_UNSUPPORTED, "engine_block: kv type %d", kv_type);
        const uint32_t esz = kv_type == NFAI_F16 ? 2 : 4;
        nfreq = (rope_dims < D ? rope_dims : D) / 2;
        NEED(bq_, (uint64_t)H * D * E, 2);
        NEED(bk_, (uint64_t)Hkv * D * E, 2);
        NEED(bv_, (uint64_t)Hkv * D * E, 2);
        NEED(bgn, E, 4);
        NEED(bf, nfreq, 4);
        NEED(bq, (uint64_t)H * D, 4);
        NEED(bk, ((uint64_t)pos + 1) * Hkv * D, esz);
        NEED(bv, ((uint64_t)pos + 1) * Hkv * D, esz);
        if (nfreq * 8 > 3072) return fail(NFAI_ERR_INVALID, "engine_block: rope_dims too large");
        e.n_ops = 4;
        e.Wqkv[0] = bq_->ptr; e.Wqkv[1] = bk_->ptr; e.Wqkv[2] = bv_->ptr;
        e.qkv_rows[0] = H * D; e.qkv_rows[1] = Hkv * D; e.qkv_rows[2] = Hkv * D;
        e.gamma_next = static_cast<const float *>(bgn->ptr);
        e.q_out = static_cast<float *>(bq->ptr);
        e.kcache = bk->ptr; e.vcache = bv->ptr; e.kv_type = kv_type;
        e.kv_pos_stride = (uint64_t)Hkv * D; e.kv_head_stride = D;   // reference layout [C][Hkv*D]
        e.rope_cs = scratch_ropecs(c); e.rope_dims = rope_dims; e.D = D;
        e.pos_dev = scratch_pos(c);
        fr = static_cast<const float *>(bf->ptr);
        HIP_TRY(hipMemcpyAsync(scratch_pos(c), &pos, 4, hipMemcpyHostToDevice, c->stream));
    }
    // cos/sin table of the position (when q|k|v is on) and the epoch of this call's hand-offs
    LAUNCH_TRY(launch_token_begin(nullptr, 0, nullptr, nullptr, 0, fr, scratch_ropecs(c), nfreq, scratch_pos(c), c->stream, words));
    void *pdev = nullptr;
    HIP_TRY(hipMalloc(&pdev, engine_params_bytes()));
    EnginePlan plan;
    hipError_t le = engine_plan(e, pdev, plan);
    if (le == hipSuccess) le = launch_engine(plan, c->stream);
    uint32_t err = 0;
    if (le == hipSuccess) le = hipMemcpyAsync(&err, words + 1, 4, hipMemcpyDeviceToHost, c->stream);
    hipError_t se = hipStreamSynchronize(c->stream);
    hipFree(pdev);
    if (le == hipErrorInvalidValue) return fail(NFAI_ERR_INVALID, "engine_block: unsupported shape (E=%u F=%u HD=%u)", E, F, HD);
    if (le != hipSuccess) return fail(NFAI_ERR_HIP, "engine_block: launch failed: %s", hipGetErrorString(le));
    HIP_TRY(se);
    if (err) return fail(NFAI_ERR_HIP, "engine_block: a bounded wait inside the launch gave up (code 0x%x)", err);
    return NFAI_OK;
}

#ifdef NFAI_STAMPS
// Not part of include/nfai_hip.h: exists only in libnfai_hip_stamps.so (tools/stamps.py binds it by name).
NFAI_API int32_t nfai_hip_debug_stamps_install(void *buf, uint32_t n_slots)
{
    nfai::g_stamp_buf = static_cast<unsigned long long *>(buf);
    nfai::g_stamp_slots = n_slots;
    nfai::g_stamp_next = 0;
    return NFAI_OK;
}
NFAI_API int32_t nfai_hip_debug_stamps_info(uint32_t slot, char *name48, uint32_t *grid, uint32_t *block, uint32_t *n_used)
{
    if (n_used) *n_used = nfai::g_stamp_next;
    if (slot >= nfai::g_stamp_next) return NFAI_ERR_INVALID;
    memcpy(name48, nfai::g_stamp_info[slot].name, 48);
    *grid = nfai::g_stamp_info[slot].grid;
    *block = nfai::g_stamp_info[slot].block;
    return NFAI_OK;
}
#endif
