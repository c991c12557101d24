// pp.hip — the layer pipeline's exchange step behind the C ABI (SURVEY.md §8b "pp" group, §8e).
//
// The reference is single-device (it takes the last enumerated Vulkan device, VulkanHelper.cs:149-150) and has no
// exchange step of any kind; these entry points are what a multi-GPU NFAI host (one process per GPU, each owning a
// contiguous range of TransformerBlocks = a slice of the block loop LlamaModel.cs:118-121) calls between stages:
// the hidden state (n_embd fp32) moves to the next stage and the sampled token id returns from the last stage to
// the first, point to point over RCCL/xGMI.  There is no collective on the data path.
//
// Every operation is ENQUEUED on the context's stream (the stream the stage's hipGraph runs on), so a tick is
// [stage graph] -> [group of sends / receives] with no host synchronisation.  RCCL is bound at run time with dlopen
// (the library that is already mapped in the process is preferred, e.g. the one PyTorch ships): libnfai_hip.so has no
// link-time dependency on it and single-GPU hosts never load it.
#include <dlfcn.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <thread>

#include "common.h"

using namespace nfai;

namespace {

// the few RCCL declarations used (rccl.h: ncclUniqueId :43, ncclCommInitRank :220, ncclSend :700, ncclRecv :722,
// ncclBroadcast :591, ncclGroupStart :923); types restated so the build does not need the header
typedef struct { char internal[128]; } UniqueId;
typedef void *Comm;
typedef int Result;                      // ncclSuccess == 0
enum { DT_UINT32 = 3, DT_FLOAT32 = 7 };  // ncclUint32, ncclFloat32

struct Rccl {
    void *so = nullptr;
    Result (*GetUniqueId)(UniqueId *) = nullptr;
    Result (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    Result (*CommDestroy)(Comm) = nullptr;
    Result (*CommCount)(Comm, int *) = nullptr;
    Result (*CommUserRank)(Comm, int *) = nullptr;
    Result (*CommCuDevice)(Comm, int *) = nullptr;
    Result (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    Result (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    Result (*Broadcast)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    Result (*GroupStart)() = nullptr;
    Result (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(Result) = nullptr;
    // optional (bound when present): the asynchronous error of a communicator and the abort that unblocks its pending operations
    Result (*CommGetAsyncError)(Comm, Result *) = nullptr;   // rccl.h: ncclCommGetAsyncError
    Result (*CommAbort)(Comm) = nullptr;                     // rccl.h: ncclCommAbort
    std::string why;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names)  // a copy already mapped in this process first (PyTorch ships its own)
            if (!r.so) r.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (const char *env = getenv("NFAI_RCCL_PATH"))
            if (!r.so) r.so = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        for (const char *n : names)
            if (!r.so) r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.so) r.so = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!r.so) { r.why = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
#define SYM(field, name)                                                        \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.so, name));           \
    if (!r.field) { r.why = std::string("RCCL symbol missing: ") + name; return; }
        SYM(GetUniqueId, "ncclGetUniqueId")
        SYM(CommInitRank, "ncclCommInitRank")
        SYM(CommDestroy, "ncclCommDestroy")
        SYM(CommCount, "ncclCommCount")
        SYM(CommUserRank, "ncclCommUserRank")
        SYM(CommCuDevice, "ncclCommCuDevice")
        SYM(Send, "ncclSend")
        SYM(Recv, "ncclRecv")
        SYM(Broadcast, "ncclBroadcast")
        SYM(GroupStart, "ncclGroupStart")
        SYM(GroupEnd, "ncclGroupEnd")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(dlsym(r.so, "ncclCommGetAsyncError"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.so, "ncclCommAbort"));
    });
    return &r;
}

struct Pp {
    uint32_t magic = 0x4E465050;  // 'NFPP'
    Ctx *ctx = nullptr;
    Comm comm = nullptr;
    uint32_t rank = 0, world = 1;
    bool in_group = false;
};

Pp *pp_of(nfai_pp_t h)
{
    if (!handle_live(h)) return nullptr;
    Pp *p = reinterpret_cast<Pp *>(h);
    return p->magic == 0x4E465050 ? p : nullptr;
}

#define RCCL_OR_FAIL(r)                                                                   \
    Rccl *r = rccl();                                                                     \
    if (!r->why.empty()) return fail(NFAI_ERR_UNSUPPORTED, "%s: %s", __func__, r->why.c_str())

#define PP_OR_FAIL(p, h)                                                                  \
    Pp *p = pp_of(h);                                                                     \
    if (!p) return fail(NFAI_ERR_INVALID, "%s: invalid pipeline handle", __func__);       \
    HIP_TRY(hipSetDevice(p->ctx->device))

#define NCCL_TRY(r, expr)                                                                 \
    do {                                                                                  \
        Result _e = (expr);                                                               \
        if (_e != 0) return fail(NFAI_ERR_HIP, "%s: %s failed: %s", __func__, #expr, r->GetErrorString(_e)); \
    } while (0)

}  // namespace

NFAI_API int32_t nfai_hip_pp_unique_id(uint8_t *out128)
{
    if (!out128) return fail(NFAI_ERR_INVALID, "pp_unique_id: null output");
    RCCL_OR_FAIL(r);
    UniqueId id;
    NCCL_TRY(r, r->GetUniqueId(&id));
    memcpy(out128, id.internal, 128);
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_pp_init(nfai_ctx_t ch, uint32_t rank, uint32_t world, const uint8_t *id128, nfai_pp_t *out)
{
    Ctx *c = ctx_of(ch);
    if (!c) return fail(NFAI_ERR_INVALID, "pp_init: invalid context handle");
    if (!id128 || !out || world == 0 || rank >= world) return fail(NFAI_ERR_INVALID, "pp_init: bad arguments (rank %u of %u)", rank, world);
    HIP_TRY(hipSetDevice(c->device));
    RCCL_OR_FAIL(r);
    UniqueId id;
    memcpy(id.internal, id128, 128);
    Pp *p = new Pp();
    p->ctx = c; p->rank = rank; p->world = world;
    Result e = r->CommInitRank(&p->comm, (int)world, id, (int)rank);
    if (e != 0) { delete p; return fail(NFAI_ERR_HIP, "pp_init: ncclCommInitRank(rank %u of %u) failed: %s", rank, world, r->GetErrorString(e)); }
    handle_register(p);
    *out = reinterpret_cast<nfai_pp_t>(p);
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_pp_destroy(nfai_pp_t h)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    hipStreamSynchronize(p->ctx->stream);
    if (p->comm) r->CommDestroy(p->comm);
    p->magic = 0;
    handle_unregister(p);
    delete p;
    return NFAI_OK;
}

// One tick's operations are posted between begin and end (ncclGroupStart / ncclGroupEnd): both ends of every link post in the
// same tick, so a send that only completes against its matching receive cannot deadlock the schedule.
NFAI_API int32_t nfai_hip_pp_begin(nfai_pp_t h)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    if (p->in_group) return fail(NFAI_ERR_STATE, "pp_begin: a group is already open");
    NCCL_TRY(r, r->GroupStart());
    p->in_group = true;
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_pp_end(nfai_pp_t h)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    if (!p->in_group) return fail(NFAI_ERR_STATE, "pp_end: no open group");
    p->in_group = false;
    NCCL_TRY(r, r->GroupEnd());
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_pp_send_hidden(nfai_pp_t h, const void *hidden_dev, uint32_t n_floats, uint32_t peer)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    if (!hidden_dev || n_floats == 0 || peer >= p->world) return fail(NFAI_ERR_INVALID, "pp_send_hidden: bad arguments (peer %u of %u)", peer, p->world);
    NCCL_TRY(r, r->Send(hidden_dev, n_floats, DT_FLOAT32, (int)peer, p->comm, p->ctx->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_pp_recv_hidden(nfai_pp_t h, void *hidden_dev, uint32_t n_floats, uint32_t peer)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    if (!hidden_dev || n_floats == 0 || peer >= p->world) return fail(NFAI_ERR_INVALID, "pp_recv_hidden: bad arguments (peer %u of %u)", peer, p->world);
    NCCL_TRY(r, r->Recv(hidden_dev, n_floats, DT_FLOAT32, (int)peer, p->comm, p->ctx->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_pp_send_token(nfai_pp_t h, const void *token_dev, uint32_t peer)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    if (!token_dev || peer >= p->world) return fail(NFAI_ERR_INVALID, "pp_send_token: bad arguments");
    NCCL_TRY(r, r->Send(token_dev, 1, DT_UINT32, (int)peer, p->comm, p->ctx->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_pp_recv_token(nfai_pp_t h, void *token_dev, uint32_t peer)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    if (!token_dev || peer >= p->world) return fail(NFAI_ERR_INVALID, "pp_recv_token: bad arguments");
    NCCL_TRY(r, r->Recv(token_dev, 1, DT_UINT32, (int)peer, p->comm, p->ctx->stream));
    return NFAI_OK;
}

// The sampled token of the last stage made visible to every stage (a host that keeps one chat transcript per rank).
NFAI_API int32_t nfai_hip_pp_bcast_token(nfai_pp_t h, void *token_dev, uint32_t root)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    if (!token_dev || root >= p->world) return fail(NFAI_ERR_INVALID, "pp_bcast_token: bad arguments");
    NCCL_TRY(r, r->Broadcast(token_dev, token_dev, 1, DT_UINT32, (int)root, p->comm, p->ctx->stream));
    return NFAI_OK;
}

// What RCCL itself says about this communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) plus the PCI bus id of the
// device: bench.py puts it into its JSON line so that "did RCCL see N ranks on N different devices" can be read off the record.
NFAI_API int32_t nfai_hip_pp_info(nfai_pp_t h, uint32_t *nranks, uint32_t *rank, int32_t *device, char *pci_bus_id32)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    int n = 0, me = 0, dev = -1;
    NCCL_TRY(r, r->CommCount(p->comm, &n));
    NCCL_TRY(r, r->CommUserRank(p->comm, &me));
    NCCL_TRY(r, r->CommCuDevice(p->comm, &dev));
    if (nranks) *nranks = (uint32_t)n;
    if (rank) *rank = (uint32_t)me;
    if (device) *device = dev;
    if (pci_bus_id32) {
        pci_bus_id32[0] = 0;
        HIP_TRY(hipDeviceGetPCIBusId(pci_bus_id32, 32, dev));
    }
    return NFAI_OK;
}

// One tick's exchange in ONE call: ncclGroupStart, every send / receive of the tick, ncclGroupEnd, all on the stage stream.  The
// host side of a tick is then this call plus the stage's graph launch (per-operation calls from a managed host cost several
// microseconds each; at 8 stages a stage's device time per tick is ~0.2 ms).  kind: 0 send hidden, 1 receive hidden (count
// floats), 2 send token, 3 receive token (one uint32).
NFAI_API int32_t nfai_hip_pp_exchange(nfai_pp_t h, const nfai_pp_op *ops, uint32_t n_ops)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    if (n_ops == 0) return NFAI_OK;
    if (!ops) return fail(NFAI_ERR_INVALID, "pp_exchange: null operation list");
    if (p->in_group) return fail(NFAI_ERR_STATE, "pp_exchange: a group is already open");
    for (uint32_t i = 0; i < n_ops; i++)
        if (!ops[i].buf || ops[i].peer >= p->world || ops[i].kind > 3 || (ops[i].kind < 2 && ops[i].count == 0))
            return fail(NFAI_ERR_INVALID, "pp_exchange: bad operation %u (kind %u, peer %u of %u)", i, ops[i].kind, ops[i].peer, p->world);
    NCCL_TRY(r, r->GroupStart());
    Result first_err = 0;
    for (uint32_t i = 0; i < n_ops && first_err == 0; i++) {
        const nfai_pp_op &o = ops[i];
        switch (o.kind) {
            case 0: first_err = r->Send(o.buf, o.count, DT_FLOAT32, (int)o.peer, p->comm, p->ctx->stream); break;
            case 1: first_err = r->Recv(o.buf, o.count, DT_FLOAT32, (int)o.peer, p->comm, p->ctx->stream); break;
            case 2: first_err = r->Send(o.buf, 1, DT_UINT32, (int)o.peer, p->comm, p->ctx->stream); break;
            default: first_err = r->Recv(o.buf, 1, DT_UINT32, (int)o.peer, p->comm, p->ctx->stream); break;
        }
    }
    const Result end_err = r->GroupEnd();  // always closed, also after a failed post
    if (first_err != 0) return fail(NFAI_ERR_HIP, "pp_exchange: a send / receive failed: %s", r->GetErrorString(first_err));
    if (end_err != 0) return fail(NFAI_ERR_HIP, "pp_exchange: ncclGroupEnd failed: %s", r->GetErrorString(end_err));
    return NFAI_OK;
}

// ---- failure detection ("RCCL async error query; bounded spins") ---------------------------------------------------
// RCCL reports a dead peer, a broken link or a failed proxy thread ASYNCHRONOUSLY: the enqueue calls above have long returned and
// the stage stream simply never drains.  _check asks the communicator (ncclCommGetAsyncError); _wait is the bounded form of
// "synchronise the stage stream": it polls the stream and the communicator until the stream is idle, an asynchronous error shows
// up, or the deadline passes; _abort (ncclCommAbort) releases operations that can no longer complete, so that the process can
// leave.  All three name the rank in their message; the host decides what to do (bench.py: exit code 3).
static int pp_async_error(Rccl *r, Pp *p, const char *fn)
{
    if (!r->CommGetAsyncError || !p->comm) return NFAI_OK;
    Result async = 0;
    const Result e = r->CommGetAsyncError(p->comm, &async);
    if (e != 0) return fail(NFAI_ERR_HIP, "%s: rank %u of %u: ncclCommGetAsyncError failed: %s", fn, p->rank, p->world, r->GetErrorString(e));
    if (async != 0 && async != 7 /* ncclInProgress */)
        return fail(NFAI_ERR_HIP, "%s: rank %u of %u: RCCL reports an asynchronous error on the pipeline communicator: %s", fn, p->rank, p->world,
                    r->GetErrorString(async));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_pp_check(nfai_pp_t h)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    return pp_async_error(r, p, __func__);
}

NFAI_API int32_t nfai_hip_pp_wait(nfai_pp_t h, uint32_t timeout_ms)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; spin++) {
        const hipError_t q = hipStreamQuery(p->ctx->stream);
        if (q == hipSuccess) return pp_async_error(r, p, __func__);
        if (q != hipErrorNotReady) return fail(NFAI_ERR_HIP, "pp_wait: rank %u of %u: the stage stream failed: %s", p->rank, p->world, hipGetErrorString(q));
        (void)hipGetLastError();
        if ((spin & 63) == 63) {  // the communicator every 64 polls (a query takes a lock inside RCCL)
            const int rc = pp_async_error(r, p, __func__);
            if (rc) return rc;
        }
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if ((uint64_t)ms >= timeout_ms)
            return fail(NFAI_ERR_HIP, "pp_wait: rank %u of %u: the stage stream did not drain within %u ms (a peer that never posted its half of an "
                                      "exchange, or a stage kernel that never finished)", p->rank, p->world, timeout_ms);
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(200));  // short phases finish inside the busy polls
    }
}

NFAI_API int32_t nfai_hip_pp_abort(nfai_pp_t h)
{
    PP_OR_FAIL(p, h);
    RCCL_OR_FAIL(r);
    if (!r->CommAbort) return fail(NFAI_ERR_UNSUPPORTED, "pp_abort: this RCCL has no ncclCommAbort");
    if (p->comm) {
        const Result e = r->CommAbort(p->comm);
        p->comm = nullptr;  // aborted communicators are gone: _destroy only frees the handle
        if (e != 0) return fail(NFAI_ERR_HIP, "pp_abort: rank %u of %u: ncclCommAbort failed: %s", p->rank, p->world, r->GetErrorString(e));
    }
    return NFAI_OK;
}
