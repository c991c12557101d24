// llama.hip — model level of the HIP backend: what LlamaModel (graph: LlamaModel.cs:21-68,
// token loop :99-174) and TransformerBlock (wiring TransformerBlock.cs:31-125, op sequence
// :127-184) do in the reference, for a contiguous range of blocks (a pipeline stage).
//
// Per token the fused path issues, per block, five launches (+ the slice merge of attention):
//   [RMSNorm+Wq,Wk,Wv+RoPE+KV write] -> [attention] -> [Wo + residual]
//   -> [RMSNorm+Wgate,Wup+SiLU*up] -> [Wdown + residual]
// then [RMSNorm+lm_head] -> [argmax + token feedback + position advance].  Position and token
// live in device memory, so the whole token is ONE hipGraph that is replayed unchanged for every
// position, with no host round trip inside a greedy loop.  (The reference: 16 fence-waited
// dispatches and two host read-back/add/upload round trips per block)
#include <string.h>

#include <algorithm>
#include <functional>

#include <map>
#include <string>
#include <vector>

#include "common.h"

using namespace nfai;

namespace {

struct Tensor {
    int type = -1;
    uint64_t rows = 0, cols = 0;
    void *ptr = nullptr;
    bool owned = false;
    uint64_t bytes = 0;
};

struct Layer {
    Tensor attn_norm, wq, wk, wv, wo, ffn_norm, wgate, wup, wdown;
    void *kcache = nullptr, *vcache = nullptr;
};

enum KClass { KC_QKV = 0, KC_ATTN = 1, KC_WO = 2, KC_GATEUP = 3, KC_DOWN = 4, KC_LMHEAD = 5, KC_OTHER = 6, KC_ENGINE = 7, KC_N = 8 };

constexpr uint32_t RING_LEN = 8192;

struct Model {
    uint32_t magic = 0x4E464D44;  // 'NFMD'
    Ctx *ctx = nullptr;
    nfai_llama_desc d{};
    bool finalized = false;
    bool first_stage = false, last_stage = false;
    bool unfused = false, use_graph = true, kv_f16 = false;
    bool engine = false;             // requested: one engine launch per block where the tensors allow it
    bool attn_ticket = false;        // a bounded wait of the granule hand-off gave up once: this model stays on the ticket form, which never waits
    uint32_t dbg_withhold = 0;       // test hook (nfai_hip_debug_attn_withhold)
    uint64_t *d_gran = nullptr;      // engine hand-off granules: per block h (E) | act (F) | x (E)
    uint32_t *d_epoch = nullptr, *d_engerr = nullptr;
    void *d_engparams = nullptr;     // one parameter block per block's engine launch
    std::vector<EnginePlan> eng_plans;  // built by finalize when engine_ok
    Tensor token_embd, output_norm, output;
    std::vector<Layer> layers;  // index = block - layer_begin
    uint64_t kv_pos_stride = 0, kv_head_stride = 0;
    uint32_t kv_esz = 4;
    // device state
    uint32_t *d_pos = nullptr, *d_tok = nullptr, *d_ring = nullptr;
    float *d_freqs = nullptr, *d_ropecs = nullptr;
    void *d_argmax_part = nullptr;
    void *d_topk = nullptr;          // workspace of the top-k candidate launch (allocated by the first nfai_hip_llama_decode_topk)
    float *d_attn_part = nullptr;
    // activations
    float *x = nullptr, *h = nullptr, *q = nullptr, *att = nullptr, *act = nullptr, *logits = nullptr;
    // extra activations of the unfused 1:1 chain
    float *xn = nullptr, *qraw = nullptr, *scores = nullptr, *wts = nullptr, *proj = nullptr, *gate = nullptr, *up = nullptr;
    uint32_t *h_pin = nullptr;  // pinned staging for token / pos
    // prefill workspace (allocated when desc.max_batch > 0); T = max_batch rounded up to 128
    struct Prefill {
        uint32_t T = 0, Spad = 0;
        uint32_t *toks = nullptr;
        float *CS = nullptr;       // cos / sin of the chunk's positions [T][D/2][2] (the q | k | v epilogue)
        float *X = nullptr, *H1 = nullptr, *Q = nullptr, *K = nullptr, *V = nullptr, *ATT = nullptr, *G = nullptr, *U = nullptr, *SC = nullptr;
        void *XN = nullptr, *QH = nullptr, *KH = nullptr, *VT = nullptr, *P = nullptr, *ACT = nullptr;  // fp16
        void *WF16 = nullptr;      // K-quant models: the blocks' matrices widened to fp16 for the MFMA GEMMs (allocated on first use) —
        uint64_t wf16_bytes = 0;   // one slot per block, each widened ONCE and kept, when that fits the memory budget (288 GB of HBM: 6.4 GB
        bool wf16_all = false;     // at 3B, 16 GB at 8B); otherwise one slot, re-widened for every block of every chunk
        uint64_t wf16_slot = 0;    // bytes per block slot
        std::vector<uint8_t> wf16_done;  // per block: slot holds the current weights
    } pf;
    uint32_t pos_host = 0;
    const float *x_last = nullptr;   // where the last enqueued token left the hidden state (m->x, or m->h on the engine path)
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    // the BLOCKING calls (nfai_hip_llama_decode_step / _decode_topk) as one graph each: token word in (from pinned host memory),
    // the token, [the top-k candidate launch,] argmax + error word [+ candidates] out to pinned host memory — the host's share of a
    // sampled token is one hipGraphLaunch and one hipStreamSynchronize
    struct SyncGraph {
        hipGraph_t g = nullptr;
        hipGraphExec_t exec = nullptr;
        float temperature = 0.f;
        uint32_t k = 0;
    } g_step, g_topk;
    bool prefetch = false, s2_used = false;  // side-stream weight prefetch (NFAI_LLAMA_PREFETCH)
    hipStream_t s2 = nullptr;
    std::vector<hipEvent_t> pf_events;
    hipGraph_t stage_graph = nullptr;      // pipeline-stage graph, captured per (hidden_in, hidden_out)
    hipGraphExec_t stage_exec = nullptr;
    const void *stage_in = nullptr;
    void *stage_out = nullptr;
    // profiling
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_class;
    bool profiling = false;
    int prof_rep_cls = -1;           // profile_kernel: class whose launches are collected and replayed back to back
    std::vector<struct Op> prof_ops;
    hipEvent_t prof_rep_ev[2] = {nullptr, nullptr};
};

Model *model_of(nfai_model_t h)
{
    if (!handle_live(h)) return nullptr;
    Model *m = reinterpret_cast<Model *>(h);
    return m->magic == 0x4E464D44 ? m : nullptr;
}

#define MODEL_OR_FAIL(m, h)                                                      \
    Model *m = model_of(h);                                                      \
    if (!m) return fail(NFAI_ERR_INVALID, "%s: invalid model handle", __func__); \
    HIP_TRY(hipSetDevice(m->ctx->device))

int dalloc(void **p, size_t bytes, hipStream_t s)
{
    const size_t padded = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(p, padded);
    if (e != hipSuccess) return fail(NFAI_ERR_OOM, "hipMalloc(%zu) failed: %s", padded, hipGetErrorString(e));
    HIP_TRY(hipMemsetAsync(*p, 0, padded, s));
    return NFAI_OK;
}

#define DALLOC(ptr, bytes)                                                          \
    do {                                                                            \
        int _rc = dalloc(reinterpret_cast<void **>(&(ptr)), (bytes), m->ctx->stream); \
        if (_rc) return _rc;                                                        \
    } while (0)

Tensor *find_slot(Model *m, const char *name, bool *ignored)
{
    *ignored = false;
    const std::string n(name);
    if (n == "token_embd.weight") return &m->token_embd;
    if (n == "output_norm.weight") return &m->output_norm;
    if (n == "output.weight") return &m->output;
    if (n == "rope_freqs.weight") { *ignored = true; return nullptr; }  // llama3 scaling factors: the reference ignores them (TransformerBlock.cs:33-38)
    unsigned blk = 0;
    char rest[64] = {0};
    if (sscanf(name, "blk.%u.%63s", &blk, rest) == 2) {
        if (blk < m->d.layer_begin || blk >= m->d.layer_end) { *ignored = true; return nullptr; }
        Layer &L = m->layers[blk - m->d.layer_begin];
        const std::string r(rest);
        if (r == "attn_norm.weight") return &L.attn_norm;
        if (r == "attn_q.weight") return &L.wq;
        if (r == "attn_k.weight") return &L.wk;
        if (r == "attn_v.weight") return &L.wv;
        if (r == "attn_output.weight") return &L.wo;
        if (r == "ffn_norm.weight") return &L.ffn_norm;
        if (r == "ffn_gate.weight") return &L.wgate;
        if (r == "ffn_up.weight") return &L.wup;
        if (r == "ffn_down.weight") return &L.wdown;
    }
    return nullptr;
}

int check_shape(const char *what, const Tensor &t, uint64_t rows, uint64_t cols, bool matrix)
{
    if (!t.ptr) return fail(NFAI_ERR_STATE, "finalize: tensor %s was never set", what);
    if (t.rows != rows || t.cols != cols)
        return fail(NFAI_ERR_INVALID, "finalize: tensor %s is %llux%llu, expected %llux%llu", what, (unsigned long long)t.rows,
                    (unsigned long long)t.cols, (unsigned long long)rows, (unsigned long long)cols);
    if (!matrix && t.type != NFAI_F32) return fail(NFAI_ERR_UNSUPPORTED, "finalize: norm gain %s must be F32 (type %d)", what, t.type);
    if (matrix && t.type != NFAI_F16 && t.type != NFAI_F32 && !is_kquant(t.type))
        return fail(NFAI_ERR_UNSUPPORTED, "finalize: matrix %s has ggml type %d; kernels exist for F16/F32/Q4_K/Q6_K", what, t.type);
    return NFAI_OK;
}

// ---- launch recording (profiling) ----------------------------------------------------------------
struct Rec {
    Model *m;
    int begin(int cls)
    {
        if (!m->profiling) return NFAI_OK;
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        m->ev.push_back(a);
        m->ev.push_back(b);
        m->ev_class.push_back(cls);
        HIP_TRY(hipEventRecord(a, m->ctx->stream));
        return NFAI_OK;
    }
    int end()
    {
        if (!m->profiling) return NFAI_OK;
        HIP_TRY(hipEventRecord(m->ev.back(), m->ctx->stream));
        return NFAI_OK;
    }
};

#define K_TRY(cls, expr)                                                                                        \
    do {                                                                                                        \
        int _rc = rec.begin(cls);                                                                               \
        if (_rc) return _rc;                                                                                    \
        hipError_t _e = (expr);                                                                                 \
        if (_e != hipSuccess)                                                                                   \
            return fail(_e == hipErrorInvalidValue ? NFAI_ERR_INVALID : NFAI_ERR_HIP, "%s: %s failed: %s", __func__, #expr, \
                        hipGetErrorString(_e));                                                                 \
        _rc = rec.end();                                                                                        \
        if (_rc) return _rc;                                                                                    \
    } while (0)

// ---- launch scheduler with a one-op look-ahead ------------------------------------------------------
// Every short GEMV pays ~3 us of dispatch + first-byte latency + drain during which HBM idles.  With
// NFAI_LLAMA_PREFETCH, when op i+1 is an fp16 GEMV its "prefetch-only" twin (the same grid touching
// exactly the bytes each wave requests first, default cache policy) is launched on a side stream as
// soon as op i-1 has finished, i.e. concurrently with op i: the requests straddle the i -> i+1
// boundary and op i+1 finds its first two steps in L2 / Infinity Cache.  Pure performance hint: no
// result depends on it (the side stream writes nothing).
struct Op {
    int kind = 2;  // 0 gemv, 1 attention, 2 generic
    int cls = KC_OTHER;
    GemvArgs g;
    AttnArgs a;
    std::function<hipError_t(hipStream_t)> f;
};

struct Sched {
    Model *m;
    Rec rec;
    bool have = false;
    Op pending;
    size_t ev_i = 0;

    int launch_now(const Op &op)
    {
        hipStream_t s = m->ctx->stream;
        int rc = rec.begin(op.cls);
        if (rc) return rc;
        hipError_t e = op.kind == 0 ? launch_gemv(op.g, s) : (op.kind == 1 ? launch_attn_decode(op.a, s) : op.f(s));
        if (e != hipSuccess)
            return fail(e == hipErrorInvalidValue ? NFAI_ERR_INVALID : NFAI_ERR_HIP, "launch (class %d) failed: %s", op.cls,
                        hipGetErrorString(e));
        rc = rec.end();
        if (rc) return rc;
        if (m->profiling && m->prof_rep_cls == op.cls) m->prof_ops.push_back(op);  // replayed by profile_kernel
        return NFAI_OK;
    }
    int submit(const Op &op)
    {
        if (have) {
            const bool pf = m->prefetch && !m->profiling && op.kind == 0 && op.g.w_type == NFAI_F16;
            if (pf) {
                if (ev_i >= m->pf_events.size()) {
                    hipEvent_t e;
                    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    m->pf_events.push_back(e);
                }
                hipEvent_t ev = m->pf_events[ev_i++];
                HIP_TRY(hipEventRecord(ev, m->ctx->stream));       // everything before `pending`
                HIP_TRY(hipStreamWaitEvent(m->s2, ev, 0));
                GemvArgs g = op.g;
                g.prefetch_only = true;
                hipError_t e = launch_gemv(g, m->s2);
                if (e != hipSuccess) return fail(NFAI_ERR_HIP, "prefetch launch failed: %s", hipGetErrorString(e));
                m->s2_used = true;
            }
            int rc = launch_now(pending);
            if (rc) return rc;
        }
        pending = op;
        have = true;
        return NFAI_OK;
    }
    int flush()
    {
        if (have) {
            int rc = launch_now(pending);
            if (rc) return rc;
            have = false;
        }
        if (m->s2_used) {  // join the side stream (required to end a capture; harmless otherwise)
            if (ev_i >= m->pf_events.size()) {
                hipEvent_t e;
                HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                m->pf_events.push_back(e);
            }
            hipEvent_t ev = m->pf_events[ev_i++];
            HIP_TRY(hipEventRecord(ev, m->s2));
            HIP_TRY(hipStreamWaitEvent(m->ctx->stream, ev, 0));
            m->s2_used = false;
        }
        return NFAI_OK;
    }
};

#define S_TRY(expr)            \
    do {                       \
        int _rc = (expr);      \
        if (_rc) return _rc;   \
    } while (0)

static Op op_gemv(int cls, const GemvArgs &g) { Op o; o.kind = 0; o.cls = cls; o.g = g; return o; }
static Op op_attn(int cls, const AttnArgs &a) { Op o; o.kind = 1; o.cls = cls; o.a = a; return o; }
static Op op_fn(int cls, std::function<hipError_t(hipStream_t)> f) { Op o; o.kind = 2; o.cls = cls; o.f = std::move(f); return o; }

GemvArgs gemv_base(Model *m, const Tensor &w, const float *x, uint32_t K)
{
    GemvArgs a;
    a.W[0] = w.ptr;
    a.seg_rows[0] = (uint32_t)w.rows;
    a.w_type = w.type;
    a.x = x;
    a.K = K;
    a.eps = m->d.eps;
    a.n_cu = (uint32_t)m->ctx->prop.multiProcessorCount;
    a.pos_dev = m->d_pos;
    a.xcd_shares = m->ctx->xcd_state == 1 ? m->ctx->xcd_shares : nullptr;   // measured by nfai_hip_llama_finalize (gemv_xcd_calibrate)
    return a;
}

// [RMSNorm + Wq, Wk, Wv + RoPE + KV write] of block L on the activation vector x (TransformerBlock.cs:129-141).
// One launch when q, k, v share an encoding; Q4_K_M files keep attn_v in Q6_K on some blocks: then the segments that differ get
// their own launch (same kernel family, same epilogue).  qkv_launch: the launch that starts at segment `first` (-> `last`).
GemvArgs qkv_launch(Model *m, Layer &L, const float *x, int first, int &last)
{
    const nfai_llama_desc &d = m->d;
    const Tensor *seg[3] = {&L.wq, &L.wk, &L.wv};
    auto t16 = [](int ty) { return ty == NFAI_Q4_K_T16 || ty == NFAI_Q6_K_T16; };
    last = first;
    // segments of one encoding share a launch; so do T16 Q4_K and Q6_K segments (mixed kernel, kernels_gemv_kqm.hip)
    while (last + 1 < 3 && (seg[last + 1]->type == seg[first]->type || (t16(seg[last + 1]->type) && t16(seg[first]->type)))) last++;
    GemvArgs a = gemv_base(m, *seg[first], x, d.E);
    for (int i = first; i <= last; i++) {
        if (seg[i]->type != seg[first]->type) a.w_type = NFAI_KQ_MIXED;
        if (seg[i]->type == NFAI_Q6_K_T16) a.seg6_mask |= 1u << i;
    }
    for (int i = 0; i < 3; i++) {
        const bool in = i >= first && i <= last;
        a.W[i] = in ? seg[i]->ptr : seg[first]->ptr;
        a.seg_rows[i] = in ? (uint32_t)seg[i]->rows : 0;
    }
    a.gamma = static_cast<const float *>(L.attn_norm.ptr);
    a.mode = GEMV_QKV_ROPE;
    a.y = m->q;
    a.kcache = L.kcache; a.vcache = L.vcache;
    a.kv_type = m->kv_f16 ? NFAI_F16 : NFAI_F32;
    a.kv_pos_stride = m->kv_pos_stride; a.kv_head_stride = m->kv_head_stride;
    a.rope_cs = m->d_ropecs; a.rope_dims = d.rope_dims;
    a.H = d.H; a.Hkv = d.Hkv; a.D = d.D;
    return a;
}

int submit_qkv(Model *m, Layer &L, const float *x, Sched &sch, const GemvArgs::Begin *begin = nullptr)
{
    for (int first = 0; first < 3;) {
        int last;
        GemvArgs a = qkv_launch(m, L, x, first, last);
        if (begin && first == 0) a.begin = *begin;  // the token's first launch also does the per-token prologue
        S_TRY(sch.submit(op_gemv(KC_QKV, a)));
        first = last + 1;
    }
    return NFAI_OK;
}

// scores + softmax + weighted V of block L in one launch (TransformerBlock.cs:144-148)
AttnArgs attn_args(Model *m, Layer &L)
{
    const nfai_llama_desc &d = m->d;
    AttnArgs a;
    a.q = m->q; a.kcache = L.kcache; a.vcache = L.vcache;
    a.kv_type = m->kv_f16 ? NFAI_F16 : NFAI_F32;
    a.kv_pos_stride = m->kv_pos_stride; a.kv_head_stride = m->kv_head_stride;
    a.o = m->att; a.H = d.H; a.Hkv = d.Hkv; a.D = d.D; a.C = d.C;
    a.pos_dev = m->d_pos; a.partials = m->d_attn_part;
    a.n_cu = (uint32_t)m->ctx->prop.multiProcessorCount;
    // slice hand-off by {value, tag} granules: tag = token epoch x blocks + block (never the tag of an earlier launch on this workspace)
    a.epoch = m->attn_ticket ? nullptr : m->d_epoch;
    a.tag_mul = (uint32_t)m->layers.size() + 1; a.tag_add = (uint32_t)(&L - m->layers.data()) + 1; a.err = m->d_engerr;
    a.debug_withhold = m->dbg_withhold;
    return a;
}

int submit_attn(Model *m, Layer &L, Sched &sch) { return sch.submit(op_attn(KC_ATTN, attn_args(m, L))); }

// One block, fused path (TransformerBlock.cs:127-184 in five launches + the attention merge).
int block_fused(Model *m, Layer &L, Sched &sch, const GemvArgs::Begin *begin = nullptr)
{
    const nfai_llama_desc &d = m->d;
    S_TRY(submit_qkv(m, L, m->x, sch, begin));
    {
        GemvArgs a = gemv_base(m, L.wo, m->att, d.H * d.D);
        a.mode = GEMV_RESIDUAL; a.res = m->x; a.y = m->h;
        const AttnArgs at = attn_args(m, L);
        if (attn_wo_ok(at, a)) {  // attention + Wo + residual in one launch (fp16 Wo, the shapes of kernels_attn.hip's table)
            S_TRY(sch.submit(op_fn(KC_ATTN, [at, a](hipStream_t st) { return launch_attn_wo(at, a, st); })));
        } else {
            S_TRY(sch.submit(op_attn(KC_ATTN, at)));
            S_TRY(sch.submit(op_gemv(KC_WO, a)));
        }
    }
    {
        GemvArgs a = gemv_base(m, L.wgate, m->h, d.E);
        a.W[1] = L.wup.ptr; a.seg_rows[1] = (uint32_t)L.wup.rows;
        a.gamma = static_cast<const float *>(L.ffn_norm.ptr);
        a.mode = GEMV_GATEUP; a.y = m->act;
        S_TRY(sch.submit(op_gemv(KC_GATEUP, a)));
    }
    {
        GemvArgs a = gemv_base(m, L.wdown, m->act, d.F);
        a.mode = GEMV_RESIDUAL; a.res = m->h; a.y = m->x;
        S_TRY(sch.submit(op_gemv(KC_DOWN, a)));
    }
    return NFAI_OK;
}

// The engine path (kernels_engine.hip) needs every matrix of every block in fp16 and every K a multiple of 512 (one LDS-DMA
// piece = 512 weights of one row), every gathered vector a multiple of 128 granules, and one workgroup per CU.
bool engine_ok(const Model *m)
{
    if (!m->engine || m->unfused || !m->d_gran) return false;
    const nfai_llama_desc &d = m->d;
    const uint32_t HD = d.H * d.D;
    if (d.E % 512 || d.F % 512 || HD != d.E || (d.Hkv * d.D) % 2 || (d.E != 512 && d.E != 1024 && d.E != 2048 && d.E != 3072 && d.E != 4096)) return false;
    if ((2 * (size_t)d.E + std::max(HD, d.F)) * 4 + 2048 > 160 * 1024 || d.D > 128) return false;  // LDS: the three activation vectors
    for (const Layer &L : m->layers)
        for (const Tensor *t : {&L.wq, &L.wk, &L.wv, &L.wo, &L.wgate, &L.wup, &L.wdown})
            if (t->type != NFAI_F16) return false;
    return true;
}

// One block as the reference's 16-dispatch chain, op for op (parity mode; needs the host copy
// of the position because the 1:1 kernels take it by value).
int block_unfused(Model *m, Layer &L, Rec &rec)
{
    const nfai_llama_desc &d = m->d;
    hipStream_t s = m->ctx->stream;
    const uint32_t p = m->pos_host, S = p + 1, HD = d.H * d.D, KD = d.Hkv * d.D;
    float *krow = static_cast<float *>(L.kcache) + (uint64_t)p * KD;
    float *vrow = static_cast<float *>(L.vcache) + (uint64_t)p * KD;
    const float *an = static_cast<const float *>(L.attn_norm.ptr), *fn = static_cast<const float *>(L.ffn_norm.ptr);
    K_TRY(KC_OTHER, launch_rmsnorm(m->x, an, m->xn, d.E, d.eps, s));                         // :129
    { GemvArgs a = gemv_base(m, L.wq, m->xn, d.E); a.y = m->qraw; K_TRY(KC_QKV, launch_gemv(a, s)); }   // :131
    { GemvArgs a = gemv_base(m, L.wk, m->xn, d.E); a.y = krow; K_TRY(KC_QKV, launch_gemv(a, s)); }      // :133
    { GemvArgs a = gemv_base(m, L.wv, m->xn, d.E); a.y = vrow; K_TRY(KC_QKV, launch_gemv(a, s)); }      // :135
    K_TRY(KC_OTHER, launch_rope(m->qraw, m->q, m->d_freqs, d.rope_dims, d.H, d.D, p, s));    // :138
    K_TRY(KC_OTHER, launch_rope(krow, krow, m->d_freqs, d.rope_dims, d.Hkv, d.D, p, s));     // :141
    K_TRY(KC_ATTN, launch_attn_scores(m->q, static_cast<const float *>(L.kcache), m->scores, d.H, d.Hkv, d.D, S, s));  // :144
    K_TRY(KC_ATTN, launch_attn_softmax(m->scores, m->wts, d.H, S, d.eps, s));                // :146
    K_TRY(KC_ATTN, launch_attn_wsum(m->wts, static_cast<const float *>(L.vcache), m->att, d.H, d.Hkv, d.D, S, s));     // :148
    { GemvArgs a = gemv_base(m, L.wo, m->att, HD); a.y = m->proj; K_TRY(KC_WO, launch_gemv(a, s)); }    // :150
    K_TRY(KC_OTHER, launch_add(m->x, m->proj, m->h, d.E, s));                                // :153-158
    K_TRY(KC_OTHER, launch_rmsnorm(m->h, fn, m->xn, d.E, d.eps, s));                         // :163
    { GemvArgs a = gemv_base(m, L.wup, m->xn, d.E); a.y = m->up; K_TRY(KC_GATEUP, launch_gemv(a, s)); }     // :165
    { GemvArgs a = gemv_base(m, L.wgate, m->xn, d.E); a.y = m->gate; K_TRY(KC_GATEUP, launch_gemv(a, s)); } // :167
    K_TRY(KC_OTHER, launch_silu(m->gate, m->gate, d.F, s));                                  // :169
    K_TRY(KC_OTHER, launch_mul(m->up, m->gate, m->act, d.F, s));                             // :171
    { GemvArgs a = gemv_base(m, L.wdown, m->act, d.F); a.y = m->proj; K_TRY(KC_DOWN, launch_gemv(a, s)); }  // :173
    K_TRY(KC_OTHER, launch_add(m->h, m->proj, m->x, d.E, s));                                // :176-181
    return NFAI_OK;
}

// Everything one token needs on this stage, enqueued on the stream.  Reads token/pos from device.
int enqueue_token(Model *m, bool with_head)
{
    const nfai_llama_desc &d = m->d;
    hipStream_t s = m->ctx->stream;
    Rec rec{m};
    const uint32_t nfreq = (d.rope_dims < d.D ? d.rope_dims : d.D) / 2;
    const bool emb_kq = is_kquant(m->token_embd.type);
    // The per-token prologue (embedding row -> x, cos/sin table of the position, hand-off epoch: TokenEmbedShader + what
    // RoPEShader.cs:254-256 recomputes per element) rides on the first q|k|v launch of the token when that launch is one of
    // the streaming GEMV kernels and the table is in a layout they read; otherwise it is its own launch.
    GemvArgs::Begin begin;
    {
        auto streams = [](int ty) { return ty == NFAI_F16 || ty == NFAI_F32 || ty == NFAI_Q4_K_T16 || ty == NFAI_Q6_K_T16; };
        const int et = m->token_embd.type;
        static const bool env_off = getenv("NFAI_BEGIN_FUSED") && atoi(getenv("NFAI_BEGIN_FUSED")) == 0;
        const bool engine_path = engine_ok(m) && m->eng_plans.size() == m->layers.size();
        int last0;
        const GemvArgs first_qkv = qkv_launch(m, m->layers[0], m->x, 0, last0);
        begin.on = !env_off && !m->unfused && !engine_path && gemv_begin_ok(first_qkv) && nfreq * 2 <= 128 &&
                   (!m->first_stage || streams(et)) && (!m->first_stage || (et != NFAI_Q4_K_T16 && et != NFAI_Q6_K_T16) || d.E % 256 == 0);
        if (begin.on) {
            if (m->first_stage) {
                begin.emb = m->token_embd.ptr; begin.emb_type = et; begin.emb_rows = m->token_embd.rows;
                begin.tok = m->d_tok; begin.x_out = m->x;
            }
            begin.freqs = m->d_freqs; begin.cs_out = m->d_ropecs; begin.n_freq = nfreq; begin.epoch = m->d_epoch;
        }
    }
    if (!begin.on) {
        if (m->first_stage && emb_kq)
            K_TRY(KC_OTHER, launch_embed_kq(m->token_embd.ptr, m->token_embd.type, m->token_embd.rows, m->d_tok, m->x, d.E, s));
        K_TRY(KC_OTHER, launch_token_begin(m->first_stage && !emb_kq ? m->token_embd.ptr : nullptr, m->token_embd.type, m->d_tok, m->x, d.E,
                                           m->d_freqs, m->d_ropecs, nfreq, m->d_pos, s, m->d_epoch));
    }
    if (m->unfused) {
        for (Layer &L : m->layers) {
            int rc = block_unfused(m, L, rec);
            if (rc) return rc;
        }
        if (m->last_stage && with_head) {
            const Tensor &head = m->output.ptr ? m->output : m->token_embd;  // tied when output.weight is absent (LlamaModel.cs:64-67)
            K_TRY(KC_OTHER, launch_rmsnorm(m->x, static_cast<const float *>(m->output_norm.ptr), m->xn, d.E, d.eps, s));
            GemvArgs a = gemv_base(m, head, m->xn, d.E);
            a.y = m->logits;
            K_TRY(KC_LMHEAD, launch_gemv(a, s));
            K_TRY(KC_OTHER, launch_argmax(m->logits, d.V, m->d_tok, m->d_argmax_part, m->d_pos, m->d_ring, RING_LEN, s));
        } else {
            K_TRY(KC_OTHER, launch_pos_advance(m->d_pos, s));
        }
        return NFAI_OK;
    }
    Sched sch{m, rec};
    const float *x_final = m->x;
    if (engine_ok(m) && m->eng_plans.size() == m->layers.size()) {
        // [q|k|v of the first block] then per block [attention] [engine: Wo -> gate|up -> Wdown -> next block's q|k|v].
        // The block input / output alternate between m->x and m->h so that no CU overwrites a vector another CU still reads.
        S_TRY(submit_qkv(m, m->layers[0], m->x, sch));
        for (size_t i = 0; i < m->layers.size(); i++) {
            Layer &L = m->layers[i];
            float *xin = (i & 1) ? m->h : m->x, *xout = (i & 1) ? m->x : m->h;
            S_TRY(submit_attn(m, L, sch));
            const EnginePlan plan = m->eng_plans[i];
            S_TRY(sch.submit(op_fn(KC_ENGINE, [plan](hipStream_t st) { return launch_engine(plan, st); })));
            x_final = xout;
        }
    } else {
        for (Layer &L : m->layers) {
            int rc = block_fused(m, L, sch, (begin.on && &L == &m->layers[0]) ? &begin : nullptr);
            if (rc) return rc;
        }
    }
    m->x_last = x_final;
    if (m->last_stage && with_head) {
        const Tensor &head = m->output.ptr ? m->output : m->token_embd;
        GemvArgs a = gemv_base(m, head, x_final, d.E);
        a.gamma = static_cast<const float *>(m->output_norm.ptr);
        a.y = m->logits;
        // SamplingUtils.ArgMax + the end-of-token bookkeeping ride on the lm_head launch (LlamaModel.cs:125-130 in one launch): the
        // streaming GEMV kernels take it; the fallback kernel for K-quant tensors whose rows are not a multiple of 16 does not
        const bool am_fused = head.type == NFAI_F16 || head.type == NFAI_F32 || head.type == NFAI_Q4_K_T16 || head.type == NFAI_Q6_K_T16;
        if (am_fused) {
            a.argmax_part = static_cast<char *>(m->d_argmax_part) + 4096;
            a.argmax_out = m->d_tok; a.argmax_pos_inc = m->d_pos; a.argmax_ring = m->d_ring; a.argmax_ring_len = RING_LEN;
        }
        S_TRY(sch.submit(op_gemv(KC_LMHEAD, a)));
        if (!am_fused)
            S_TRY(sch.submit(op_fn(KC_OTHER, [m](hipStream_t st) {
                return launch_argmax(m->logits, m->d.V, m->d_tok, m->d_argmax_part, m->d_pos, m->d_ring, RING_LEN, st);
            })));
    } else {
        S_TRY(sch.submit(op_fn(KC_OTHER, [m](hipStream_t st) { return launch_pos_advance(m->d_pos, st); })));
    }
    return sch.flush();
}

int ensure_graph(Model *m)
{
    if (m->graph_exec) return NFAI_OK;
    hipStream_t s = m->ctx->stream;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_token(m, true);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc) {
        if (g) hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess) return fail(NFAI_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    m->graph = g;
    HIP_TRY(hipGraphInstantiate(&m->graph_exec, g, nullptr, nullptr, 0));
    return NFAI_OK;
}

// Enqueue one whole token (graph replay when allowed).
int run_token(Model *m)
{
    if (m->pos_host >= m->d.C)
        return fail(NFAI_ERR_KV_FULL, "KV cache full: position %u == capacity %u (the reference would write out of bounds here)",
                    m->pos_host, m->d.C);
    const bool graphable = m->use_graph && !m->unfused && !m->profiling;
    if (graphable) {
        int rc = ensure_graph(m);
        if (rc) return rc;
        HIP_TRY(hipGraphLaunch(m->graph_exec, m->ctx->stream));
    } else {
        int rc = enqueue_token(m, true);
        if (rc) return rc;
    }
    m->pos_host++;
    return NFAI_OK;
}

int set_token_async(Model *m, uint32_t tok)
{
    // Pageable source: hipMemcpyAsync stages it before returning, so a stack value is safe.
    HIP_TRY(hipMemcpyAsync(m->d_tok, &tok, 4, hipMemcpyHostToDevice, m->ctx->stream));
    return NFAI_OK;
}

// The engine launch of every block: parameter blocks written to device memory once (finalize), launched from the token graph.
int build_engine_plans(Model *m)
{
    m->eng_plans.clear();
    if (!engine_ok(m)) return NFAI_OK;
    const nfai_llama_desc &d = m->d;
    const size_t pb = engine_params_bytes();
    if (!m->d_engparams) DALLOC(m->d_engparams, pb * m->layers.size());
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    for (size_t i = 0; i < m->layers.size(); i++) {
        Layer &L = m->layers[i];
        float *xin = (i & 1) ? m->h : m->x, *xout = (i & 1) ? m->x : m->h;
        EngineArgs e;
        const bool more = i + 1 < m->layers.size();
        e.n_ops = more ? 4 : 3;
        e.E = d.E; e.F = d.F; e.HD = d.H * d.D;
        e.Wo = L.wo.ptr; e.Wgate = L.wgate.ptr; e.Wup = L.wup.ptr; e.Wdown = L.wdown.ptr;
        e.att = m->att; e.x_in = xin; e.x_out = xout;
        e.gamma_ffn = static_cast<const float *>(L.ffn_norm.ptr);
        e.eps = d.eps;
        uint64_t *gl = m->d_gran + i * (2 * (size_t)d.E + d.F);
        e.g_h = gl; e.g_act = gl + d.E; e.g_x = gl + d.E + d.F;
        e.epoch = m->d_epoch; e.err = m->d_engerr;
        e.n_cu = (uint32_t)m->ctx->prop.multiProcessorCount;
        if (more) {
            Layer &N = m->layers[i + 1];
            e.gamma_next = static_cast<const float *>(N.attn_norm.ptr);
            e.Wqkv[0] = N.wq.ptr; e.Wqkv[1] = N.wk.ptr; e.Wqkv[2] = N.wv.ptr;
            e.qkv_rows[0] = (uint32_t)N.wq.rows; e.qkv_rows[1] = (uint32_t)N.wk.rows; e.qkv_rows[2] = (uint32_t)N.wv.rows;
            e.q_out = m->q; e.kcache = N.kcache; e.vcache = N.vcache;
            e.kv_type = m->kv_f16 ? NFAI_F16 : NFAI_F32;
            e.kv_pos_stride = m->kv_pos_stride; e.kv_head_stride = m->kv_head_stride;
            e.rope_cs = m->d_ropecs; e.rope_dims = d.rope_dims; e.D = d.D; e.pos_dev = m->d_pos;
        }
        EnginePlan plan;
        hipError_t he = engine_plan(e, static_cast<char *>(m->d_engparams) + i * pb, plan);
        if (he == hipErrorInvalidValue) { m->eng_plans.clear(); return NFAI_OK; }  // a shape the engine does not take: five launches
        if (he != hipSuccess) return fail(NFAI_ERR_HIP, "engine_plan failed: %s", hipGetErrorString(he));
        m->eng_plans.push_back(plan);
    }
    return NFAI_OK;
}

uint64_t tensor_bytes(const Tensor &t) { return t.ptr ? weight_row_bytes(t.type, t.cols) * t.rows : 0; }

// A bounded wait inside a launch gave up (a workgroup was not resident, or a producer never published): the results of that
// token are not valid.  The word is sticky until the model is reset.
int engine_failed(Model *m, uint32_t code)
{
    if (code == 0x1000u)
        return fail(NFAI_ERR_HIP, "attention launch gave up waiting for the partial results of a KV slice (code 0x1000): are all its "
                                  "workgroups resident?  NFAI_ATTN_POLL=0 selects the ticket hand-off");
    return fail(NFAI_ERR_HIP, "engine launch gave up waiting (code 0x%x: 0x10 ring slot, 0x20 activation, 0x40 weights, 0x80 gather, "
                              "0x100-0x400 consumers, 0x1000 attention slices): are all %d workgroups resident?  NFAI_ENGINE=0 selects "
                              "the five-launch path", code, m->ctx->prop.multiProcessorCount);
}

}  // namespace

// ---- C ABI -----------------------------------------------------------------------------------
NFAI_API int32_t nfai_hip_llama_create(nfai_ctx_t ch, const nfai_llama_desc *desc, nfai_model_t *out)
{
    Ctx *c = ctx_of(ch);
    if (!c) return fail(NFAI_ERR_INVALID, "llama_create: invalid context handle");
    if (!desc || !out) return fail(NFAI_ERR_INVALID, "llama_create: null argument");
    HIP_TRY(hipSetDevice(c->device));
    const nfai_llama_desc &d = *desc;
    NFAI_REQUIRE(d.E && d.L && d.H && d.Hkv && d.D && d.F && d.V && d.C, "llama_create: zero dimension");
    NFAI_REQUIRE(d.H % d.Hkv == 0 && d.H / d.Hkv <= 8, "llama_create: H=%u must be a multiple (<= 8x) of Hkv=%u", d.H, d.Hkv);
    NFAI_REQUIRE(d.D == 64 || d.D == 128, "llama_create: head_dim %u (kernels exist for 64 and 128)", d.D);
    NFAI_REQUIRE(d.E % 8 == 0 && d.F % 8 == 0 && (d.H * d.D) % 8 == 0, "llama_create: E, F, H*D must be multiples of 8");
    NFAI_REQUIRE(d.layer_begin < d.layer_end && d.layer_end <= d.L, "llama_create: layer range [%u,%u) outside [0,%u)", d.layer_begin,
                 d.layer_end, d.L);
    NFAI_REQUIRE(d.rope_dims % 2 == 0 && d.rope_dims <= d.D, "llama_create: rope_dims %u", d.rope_dims);
    NFAI_REQUIRE(d.rope_n_freqs <= d.rope_dims / 2, "llama_create: rope_n_freqs %u > rope_dims/2", d.rope_n_freqs);
    NFAI_REQUIRE(d.C <= 32768, "llama_create: KV capacity %u > 32768", d.C);
    Model *m = new Model();
    m->ctx = c;
    m->d = d;
    m->first_stage = d.layer_begin == 0;
    m->last_stage = d.layer_end == d.L;
    m->unfused = (d.flags & NFAI_LLAMA_UNFUSED) != 0;
    m->use_graph = (d.flags & NFAI_LLAMA_NO_GRAPH) == 0;
    m->kv_f16 = (d.flags & NFAI_LLAMA_KV_F16) != 0;
    {
        const char *env = getenv("NFAI_PREFETCH");
        m->prefetch = env ? (env[0] == '1') : ((d.flags & NFAI_LLAMA_PREFETCH) != 0);
        if (m->prefetch || d.max_batch > 0) HIP_TRY(hipStreamCreateWithFlags(&m->s2, hipStreamNonBlocking));  // (the prefill reads weights ahead on it)
    }
    {
        const char *env = getenv("NFAI_ENGINE");
        m->engine = !m->unfused && (env ? (env[0] == '1') : ((d.flags & NFAI_LLAMA_ENGINE) != 0));
    }
    if (m->unfused && m->kv_f16) { delete m; return fail(NFAI_ERR_INVALID, "llama_create: the 1:1 chain keeps the reference's fp32 KV cache"); }
    m->layers.resize(d.layer_end - d.layer_begin);
    m->kv_esz = m->kv_f16 ? 2 : 4;
    if (m->unfused) {  // reference layout [C][Hkv*D] (MatrixMultiplyShader.cs:59-65, :286-287)
        m->kv_pos_stride = (uint64_t)d.Hkv * d.D;
        m->kv_head_stride = d.D;
    } else {           // head-major [Hkv][C][D]: each attention block streams one contiguous range
        m->kv_pos_stride = d.D;
        m->kv_head_stride = (uint64_t)d.C * d.D;
    }
    const size_t kvb = (size_t)d.C * d.Hkv * d.D * m->kv_esz;
    for (Layer &L : m->layers) {
        DALLOC(L.kcache, kvb);
        DALLOC(L.vcache, kvb);
    }
    DALLOC(m->d_pos, 256);
    DALLOC(m->d_tok, 256);
    m->d_engerr = m->d_tok + 1;   // the sticky error word sits next to the token word: ONE 8-byte copy takes both to the host after a blocking step
    DALLOC(m->d_ring, RING_LEN * 4);
    DALLOC(m->d_freqs, (d.D / 2 + 8) * 4);
    DALLOC(m->d_ropecs, (d.D + 16) * 4);
    DALLOC(m->d_argmax_part, 4096 + argmax_fused_bytes());  // k_argmax's partials | the fused lm_head + ArgMax launch's
    DALLOC(m->d_epoch, 256);
    if (m->engine) DALLOC(m->d_gran, (size_t)m->layers.size() * (2 * (size_t)d.E + d.F) * 8);
    DALLOC(m->d_attn_part, attn_partials_bytes(d.H, d.Hkv, d.D, true) + attn_wo_extra_bytes(d.H, d.D));
    DALLOC(m->x, d.E * 4);
    DALLOC(m->h, d.E * 4);
    DALLOC(m->q, d.H * d.D * 4);
    DALLOC(m->att, d.H * d.D * 4);
    DALLOC(m->act, d.F * 4);
    if (m->last_stage) DALLOC(m->logits, (size_t)d.V * 4);
    if (m->unfused) {
        DALLOC(m->xn, d.E * 4);
        DALLOC(m->qraw, d.H * d.D * 4);
        DALLOC(m->scores, (size_t)d.H * d.C * 4);
        DALLOC(m->wts, (size_t)d.H * d.C * 4);
        DALLOC(m->proj, d.E * 4);
        DALLOC(m->gate, d.F * 4);
        DALLOC(m->up, d.F * 4);
    }
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&m->h_pin), 4096, hipHostMallocDefault));
    memset(m->h_pin, 0, 4096);
    if (d.max_batch > 0 && !m->unfused) {
        Model::Prefill &w = m->pf;
        w.T = (d.max_batch + 127) / 128 * 128;
        w.Spad = (d.C + 63) / 64 * 64;
        const size_t T = w.T, HD = (size_t)d.H * d.D, KD = (size_t)d.Hkv * d.D;
        DALLOC(w.toks, T * 4);
        DALLOC(w.X, T * d.E * 4);
        DALLOC(w.H1, T * d.E * 4);
        DALLOC(w.Q, T * (HD + 2 * KD) * 4);  // q | k | v columns of one GEMM output
        DALLOC(w.CS, T * d.D * 4);
        DALLOC(w.ATT, T * HD * 4);
        DALLOC(w.G, T * d.F * 2 * 4);        // gate | up columns of one GEMM output
        DALLOC(w.SC, (size_t)d.H * T * w.Spad * 4);
        DALLOC(w.XN, T * std::max<size_t>(d.E, HD) * 2);
        DALLOC(w.QH, T * HD * 2);
        DALLOC(w.KH, KD * w.Spad * 2);
        DALLOC(w.VT, KD * w.Spad * 2);
        DALLOC(w.P, (size_t)d.H * T * w.Spad * 2);
        DALLOC(w.ACT, T * d.F * 2);
    }
    // RoPE frequency table as TransformerBlock.cs:33-38 builds it; entries >= rope_n_freqs are zero
    // (the reference uploads 32 entries only, TransformerBlock.cs:66).
    std::vector<float> fr(d.D / 2 + 8, 0.f);
    for (uint32_t i = 0; i < d.rope_dims / 2; i++) {
        const float f = 1.0f / powf(d.rope_base, (float)i / ((float)d.rope_dims / 2.0f));
        fr[i] = i < d.rope_n_freqs ? f : 0.0f;
    }
    HIP_TRY(hipMemcpyAsync(m->d_freqs, fr.data(), fr.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    handle_register(m);
    *out = reinterpret_cast<nfai_model_t>(m);
    return NFAI_OK;
}

static void drop_sync_graph(Model::SyncGraph &sg)
{
    if (sg.exec) { hipGraphExecDestroy(sg.exec); sg.exec = nullptr; }
    if (sg.g) { hipGraphDestroy(sg.g); sg.g = nullptr; }
}

static void drop_graphs(Model *m)
{
    drop_sync_graph(m->g_step);
    drop_sync_graph(m->g_topk);
    if (m->graph_exec) { hipGraphExecDestroy(m->graph_exec); m->graph_exec = nullptr; }
    if (m->graph) { hipGraphDestroy(m->graph); m->graph = nullptr; }
    if (m->stage_exec) { hipGraphExecDestroy(m->stage_exec); m->stage_exec = nullptr; }
    if (m->stage_graph) { hipGraphDestroy(m->stage_graph); m->stage_graph = nullptr; }
}

// One token, blocking; the argmax and the sticky error word of the in-kernel waits land in m->h_pin[0..1].  The slices' workgroups
// of the attention launch wait for each other (granule hand-off): when one of those bounded waits gives up (codes 0x1000-0x4000:
// a workgroup was not resident, e.g. another process shares the device), the token's results are not valid — the model switches
// to the ticket form, which never waits, for good, says so once on stderr, and the SAME token is run again from the same position.
NFAI_API int32_t nfai_hip_llama_destroy(nfai_model_t h)
{
    MODEL_OR_FAIL(m, h);
    hipStreamSynchronize(m->ctx->stream);
    drop_graphs(m);
    for (hipEvent_t e : m->pf_events) hipEventDestroy(e);
    if (m->s2) hipStreamDestroy(m->s2);
    for (hipEvent_t e : m->ev) hipEventDestroy(e);
    for (hipEvent_t e : m->prof_rep_ev) if (e) hipEventDestroy(e);
    auto free_t = [](Tensor &t) { if (t.owned && t.ptr) hipFree(t.ptr); };
    free_t(m->token_embd); free_t(m->output_norm); free_t(m->output);
    for (Layer &L : m->layers) {
        free_t(L.attn_norm); free_t(L.wq); free_t(L.wk); free_t(L.wv); free_t(L.wo);
        free_t(L.ffn_norm); free_t(L.wgate); free_t(L.wup); free_t(L.wdown);
        hipFree(L.kcache); hipFree(L.vcache);
    }
    void *ptrs[] = {m->d_topk, m->d_engparams, m->d_gran, m->d_epoch, m->d_pos, m->d_tok, m->d_ring, m->d_freqs, m->d_ropecs, m->d_argmax_part, m->d_attn_part, m->x, m->h,
                    m->q, m->att, m->act, m->logits, m->xn, m->qraw, m->scores, m->wts, m->proj, m->gate, m->up};
    for (void *p : ptrs) if (p) hipFree(p);
    void *pfp[] = {m->pf.CS, m->pf.toks, m->pf.X, m->pf.H1, m->pf.Q, m->pf.K, m->pf.V, m->pf.ATT, m->pf.G, m->pf.U, m->pf.SC,
                   m->pf.XN, m->pf.QH, m->pf.KH, m->pf.VT, m->pf.P, m->pf.ACT, m->pf.WF16};
    for (void *p : pfp) if (p) hipFree(p);
    if (m->h_pin) hipHostFree(m->h_pin);
    m->magic = 0;
    handle_unregister(m);
    delete m;
    return NFAI_OK;
}

static int set_tensor_impl(Model *m, const char *name, int type, uint64_t rows, uint64_t cols, const void *host, void *dev)
{
    if (!name) return fail(NFAI_ERR_INVALID, "set_tensor: null name");
    bool ignored = false;
    Tensor *t = find_slot(m, name, &ignored);
    if (ignored) return NFAI_OK;
    if (!t) return fail(NFAI_ERR_INVALID, "set_tensor: unknown tensor name '%s'", name);
    const uint64_t rb = weight_row_bytes(type, cols);
    if (rb == 0) return fail(NFAI_ERR_UNSUPPORTED, "set_tensor(%s): ggml type %d with %llu columns is not supported (the reference "
                             "throws \"Unsupported data type\" for everything but F32/F16, Parser.cs:111-114)", name, type,
                             (unsigned long long)cols);
    if (!dev && !host) return fail(NFAI_ERR_INVALID, "set_tensor(%s): null data", name);
    if (dev && (reinterpret_cast<uintptr_t>(dev) & 15)) return fail(NFAI_ERR_INVALID, "set_tensor_device(%s): pointer not 16-byte aligned", name);
    // The new storage is built completely before the slot is touched: a failing re-set (allocation, copy, repack) leaves the
    // previous tensor in place and valid.  Only then is the old owned storage freed and the slot swapped.
    Tensor nt;
    nt.type = type; nt.rows = rows; nt.cols = cols; nt.bytes = rb * rows;
    hipStream_t s = m->ctx->stream;
    const bool q4_t16 = type == NFAI_Q4_K && rows > 0 && rows % 16 == 0, q6_t16 = type == NFAI_Q6_K && rows > 0 && rows % 16 == 0;
    auto fail_free = [&](void *a, void *b, int rc) { if (a) hipFree(a); if (b) hipFree(b); return rc; };
    if (type == NFAI_Q6_K || q4_t16) {
        // native blocks (host or device) -> owned repacked copy: Q6_K planes (common.h), Q4_K / Q6_K T16 tiles (kernels_gemv_kqm.hip)
        void *native = dev, *staged = nullptr;
        if (!dev) {
            int rc = dalloc(&staged, nt.bytes, s);
            if (rc) return rc;
            hipError_t e = hipMemcpyAsync(staged, host, nt.bytes, hipMemcpyHostToDevice, s);
            if (e != hipSuccess) return fail_free(staged, nullptr, fail(NFAI_ERR_HIP, "set_tensor(%s): upload failed: %s", name, hipGetErrorString(e)));
            native = staged;
        }
        int rc = dalloc(&nt.ptr, nt.bytes, s);
        if (rc) return fail_free(staged, nullptr, rc);
        nt.owned = true;
        hipError_t e = q4_t16 ? launch_repack_q4k_t16(native, nt.ptr, rows, cols, s)
                     : q6_t16 ? launch_repack_q6k_t16(native, nt.ptr, rows, cols, s)
                              : launch_repack_q6k(native, nt.ptr, rows * cols / 256, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail_free(staged, nt.ptr, fail(NFAI_ERR_HIP, "set_tensor(%s): K-quant repack failed: %s", name, hipGetErrorString(e)));
        if (q4_t16) nt.type = NFAI_Q4_K_T16;
        if (q6_t16) nt.type = NFAI_Q6_K_T16;
        if (staged) hipFree(staged);
    } else if (dev) {
        nt.ptr = dev;
        nt.owned = false;
    } else {
        int rc = dalloc(&nt.ptr, nt.bytes, s);
        if (rc) return rc;
        nt.owned = true;
        hipError_t e = hipMemcpyAsync(nt.ptr, host, nt.bytes, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail_free(nt.ptr, nullptr, fail(NFAI_ERR_HIP, "set_tensor(%s): upload failed: %s", name, hipGetErrorString(e)));
    }
    std::fill(m->pf.wf16_done.begin(), m->pf.wf16_done.end(), 0);  // the kept fp16 copies of the K-quant prefill are of the old weights
    m->finalized = false;  // graphs captured over the old pointer are dropped by the next finalize
    if (t->owned && t->ptr) {
        hipStreamSynchronize(s);  // nothing enqueued may still read the old storage
        hipFree(t->ptr);
    }
    *t = nt;
    return NFAI_OK;
}

// The slots of a pipeline stage: separate models (KV cache, position, graph) over the donor's weights.
NFAI_API int32_t nfai_hip_llama_share_tensors(nfai_model_t h, nfai_model_t donor_h)
{
    MODEL_OR_FAIL(m, h);
    Model *src = model_of(donor_h);
    if (!src || src == m) return fail(NFAI_ERR_INVALID, "share_tensors: invalid donor handle");
    if (src->ctx->device != m->ctx->device) return fail(NFAI_ERR_INVALID, "share_tensors: donor lives on another device");
    const nfai_llama_desc &a = m->d, &b = src->d;
    if (a.E != b.E || a.L != b.L || a.H != b.H || a.Hkv != b.Hkv || a.D != b.D || a.F != b.F || a.V != b.V ||
        a.layer_begin != b.layer_begin || a.layer_end != b.layer_end)
        return fail(NFAI_ERR_INVALID, "share_tensors: donor has different dimensions or layer range");
    auto share = [](Tensor &dst, const Tensor &s2) {
        if (dst.owned && dst.ptr) hipFree(dst.ptr);
        dst = s2;
        dst.owned = false;
    };
    hipStreamSynchronize(m->ctx->stream);
    share(m->token_embd, src->token_embd);
    share(m->output_norm, src->output_norm);
    share(m->output, src->output);
    for (size_t i = 0; i < m->layers.size(); i++) {
        Layer &L = m->layers[i];
        const Layer &S = src->layers[i];
        share(L.attn_norm, S.attn_norm); share(L.wq, S.wq); share(L.wk, S.wk); share(L.wv, S.wv); share(L.wo, S.wo);
        share(L.ffn_norm, S.ffn_norm); share(L.wgate, S.wgate); share(L.wup, S.wup); share(L.wdown, S.wdown);
    }
    std::fill(m->pf.wf16_done.begin(), m->pf.wf16_done.end(), 0);
    m->finalized = false;
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_set_tensor(nfai_model_t h, const char *name, int32_t type, uint64_t rows, uint64_t cols, const void *host)
{
    MODEL_OR_FAIL(m, h);
    return set_tensor_impl(m, name, type, rows, cols, host, nullptr);
}

NFAI_API int32_t nfai_hip_llama_set_tensor_device(nfai_model_t h, const char *name, int32_t type, uint64_t rows, uint64_t cols, void *dev)
{
    MODEL_OR_FAIL(m, h);
    if (!dev) return fail(NFAI_ERR_INVALID, "set_tensor_device: null pointer");
    return set_tensor_impl(m, name, type, rows, cols, nullptr, dev);
}

NFAI_API int32_t nfai_hip_llama_finalize(nfai_model_t h)
{
    MODEL_OR_FAIL(m, h);
    const nfai_llama_desc &d = m->d;
    int rc;
    if (m->first_stage || (m->last_stage && !m->output.ptr))
        if ((rc = check_shape("token_embd.weight", m->token_embd, d.V, d.E, true))) return rc;
    if (m->last_stage) {
        if ((rc = check_shape("output_norm.weight", m->output_norm, 1, d.E, false))) return rc;
        if (m->output.ptr && (rc = check_shape("output.weight", m->output, d.V, d.E, true))) return rc;
    }
    for (size_t i = 0; i < m->layers.size(); i++) {
        Layer &L = m->layers[i];
        char nm[64];
#define CHK(field, suffix, r, c_, mat)                                         \
    snprintf(nm, sizeof(nm), "blk.%zu." suffix, i + d.layer_begin);            \
    if ((rc = check_shape(nm, L.field, r, c_, mat))) return rc;
        CHK(attn_norm, "attn_norm.weight", 1, d.E, false)
        CHK(wq, "attn_q.weight", d.H * d.D, d.E, true)
        CHK(wk, "attn_k.weight", d.Hkv * d.D, d.E, true)
        CHK(wv, "attn_v.weight", d.Hkv * d.D, d.E, true)
        CHK(wo, "attn_output.weight", d.E, d.H * d.D, true)
        CHK(ffn_norm, "ffn_norm.weight", 1, d.E, false)
        CHK(wgate, "ffn_gate.weight", d.F, d.E, true)
        CHK(wup, "ffn_up.weight", d.F, d.E, true)
        CHK(wdown, "ffn_down.weight", d.E, d.F, true)
#undef CHK
        if (!m->unfused && L.wgate.type != L.wup.type)
            return fail(NFAI_ERR_UNSUPPORTED, "finalize: blk.%zu ffn_gate and ffn_up have different tensor types", i + d.layer_begin);
    }
    drop_graphs(m);
    (void)gemv_xcd_calibrate(m->ctx);   // once per context: how fast each XCD streams (rows of the lm_head / gate | up launches are dealt by it)
    if ((rc = build_engine_plans(m))) return rc;
    m->finalized = true;
    return NFAI_OK;
}

struct TopkOut { float v[TOPK_MAX]; uint32_t i[TOPK_MAX]; float M, S; };  // the head of the top-k workspace (topk_out_offset())
static_assert(sizeof(TopkOut) + 16 <= 4096, "pinned staging: words 0..3 (argmax, error word, token in, spare), then the candidates");

// Capture [token word H2D] -> the token -> [top-k launch] -> [argmax, error word(, candidates) D2H].  Errors inside the capture end
// it before they are reported (a stream left in capture mode would poison every later call).
static int ensure_sync_graph(Model *m, Model::SyncGraph &sg, bool topk, float temperature, uint32_t k)
{
    if (sg.exec && (!topk || (sg.temperature == temperature && sg.k == k))) return NFAI_OK;
    drop_sync_graph(sg);
    hipStream_t s = m->ctx->stream;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = NFAI_OK;
    hipError_t e = hipMemcpyAsync(m->d_tok, m->h_pin + 2, 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) rc = enqueue_token(m, true);
    if (e == hipSuccess && !rc && topk) e = launch_topk(m->logits, m->d.V, temperature, k, m->d_topk, s);
    if (e == hipSuccess && !rc) e = hipMemcpyAsync(m->h_pin, m->d_tok, 8, hipMemcpyDeviceToHost, s);   // token word + error word (adjacent)
    if (e == hipSuccess && !rc && topk)
        e = hipMemcpyAsync(m->h_pin + 4, static_cast<const char *>(m->d_topk) + topk_out_offset(), sizeof(TopkOut), hipMemcpyDeviceToHost, s);
    hipGraph_t g = nullptr;
    const hipError_t e2 = hipStreamEndCapture(s, &g);
    if (rc || e != hipSuccess || e2 != hipSuccess) {
        if (g) hipGraphDestroy(g);
        if (rc) return rc;
        return fail(NFAI_ERR_HIP, "capturing the blocking-step graph failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    }
    sg.g = g;
    HIP_TRY(hipGraphInstantiate(&sg.exec, g, nullptr, nullptr, 0));
    sg.temperature = temperature;
    sg.k = k;
    return NFAI_OK;
}

// One token, blocking; the argmax and the sticky error word of the in-kernel waits land in m->h_pin[0..1] (with topk: the candidates
// in m->h_pin[4..]).  The slices' workgroups of the attention launch wait for each other (granule hand-off): when one of those
// bounded waits gives up (codes 0x1000-0x4000: a workgroup was not resident, e.g. another process shares the device), the token's
// results are not valid — the model switches to the ticket form, which never waits, for good, says so once on stderr, and the SAME
// token is run again from the same position.
static int step_blocking(Model *m, uint32_t token, bool topk = false, float temperature = 0.f, uint32_t k = 0)
{
    hipStream_t s = m->ctx->stream;
    for (int attempt = 0;; attempt++) {
        const uint32_t pos = m->pos_host;
        int rc;
        const bool graphable = m->use_graph && !m->unfused && !m->profiling;
        if (graphable) {
            if (m->pos_host >= m->d.C)
                return fail(NFAI_ERR_KV_FULL, "KV cache full: position %u == capacity %u (the reference would write out of bounds here)",
                            m->pos_host, m->d.C);
            Model::SyncGraph &sg = topk ? m->g_topk : m->g_step;
            if ((rc = ensure_sync_graph(m, sg, topk, temperature, k))) return rc;
            m->h_pin[2] = token;
            HIP_TRY(hipGraphLaunch(sg.exec, s));
            m->pos_host++;
        } else {
            if ((rc = set_token_async(m, token))) return rc;
            if ((rc = run_token(m))) return rc;
            if (topk) {
                const hipError_t e = launch_topk(m->logits, m->d.V, temperature, k, m->d_topk, s);
                if (e != hipSuccess) return fail(NFAI_ERR_HIP, "decode_topk: launch failed: %s", hipGetErrorString(e));
                HIP_TRY(hipMemcpyAsync(m->h_pin + 4, static_cast<const char *>(m->d_topk) + topk_out_offset(), sizeof(TopkOut), hipMemcpyDeviceToHost, s));
            }
            HIP_TRY(hipMemcpyAsync(m->h_pin, m->d_tok, 8, hipMemcpyDeviceToHost, s));   // token word + error word (adjacent)
        }
        HIP_TRY(hipStreamSynchronize(s));
        const uint32_t code = m->h_pin[1];
        if (code == 0) return NFAI_OK;
        if (attempt > 0 || m->attn_ticket || (code & ~0x7000u) != 0) return engine_failed(m, code);
        fprintf(stderr, "nfai_hip: an attention launch gave up waiting for a KV slice's partial results (code 0x%x) at position %u — are all its "
                        "workgroups resident?  Re-running the token on the ticket hand-off; this model keeps it from now on.\n", code, pos);
        m->attn_ticket = true;
        drop_graphs(m);  // captured with the granule form
        HIP_TRY(hipMemsetAsync(m->d_engerr, 0, 4, s));
        HIP_TRY(hipMemcpyAsync(m->d_pos, &pos, 4, hipMemcpyHostToDevice, s));  // the failed token may have advanced it
        HIP_TRY(hipStreamSynchronize(s));
        m->h_pin[1] = 0;
        m->pos_host = pos;
    }
}

#define NEED_FINAL(m) \
    if (!(m)->finalized) return fail(NFAI_ERR_STATE, "%s: call nfai_hip_llama_finalize first", __func__)

NFAI_API int32_t nfai_hip_llama_decode_step(nfai_model_t h, uint32_t token, float *logits_host, uint32_t *argmax)
{
    MODEL_OR_FAIL(m, h);
    NEED_FINAL(m);
    if (!(m->first_stage && m->last_stage)) return fail(NFAI_ERR_STATE, "decode_step: model is a pipeline stage; use nfai_hip_llama_stage_step");
    if (token >= m->d.V) return fail(NFAI_ERR_INVALID, "decode_step: token %u >= vocab %u", token, m->d.V);
    int rc = step_blocking(m, token);
    if (rc) return rc;
    if (argmax) *argmax = m->h_pin[0];
    if (logits_host) {
        hipStream_t s = m->ctx->stream;
        HIP_TRY(hipMemcpyAsync(logits_host, m->logits, (size_t)m->d.V * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return NFAI_OK;
}

// One token, then the candidates of the reference's DEFAULT sampler on the device (LlamaModel.cs:128-130,165: TopP over V logits
// read back to the host; here 8k + 8 bytes come back instead of 513 KB).
NFAI_API int32_t nfai_hip_llama_decode_topk(nfai_model_t h, uint32_t token, float temperature, uint32_t k, uint32_t *ids_out, float *probs_out)
{
    MODEL_OR_FAIL(m, h);
    NEED_FINAL(m);
    if (!(m->first_stage && m->last_stage)) return fail(NFAI_ERR_STATE, "decode_topk: model is a pipeline stage");
    if (token >= m->d.V) return fail(NFAI_ERR_INVALID, "decode_topk: token %u >= vocab %u", token, m->d.V);
    // every argument is checked BEFORE the token runs (topk_run's own predicates): a rejected call leaves the position where it was
    if (!ids_out || !probs_out) return fail(NFAI_ERR_INVALID, "decode_topk: null output");
    if (k == 0 || k > TOPK_MAX || k > m->d.V) return fail(NFAI_ERR_INVALID, "decode_topk: k=%u outside [1, min(%u, V=%u)]", k, TOPK_MAX, m->d.V);
    if (!(temperature > 0.f)) return fail(NFAI_ERR_INVALID, "decode_topk: temperature %g (the reference divides by it, SamplingUtils.cs:7)", temperature);
    if (!m->d_topk) DALLOC(m->d_topk, topk_work_bytes(m->d.V));
    // the token's kernels, the candidate launch and the read-back (error word + 8 * TOPK_MAX + 8 bytes) are ONE graph: one launch
    // and one host synchronisation per sampled token
    int rc = step_blocking(m, token, true, temperature, k);
    if (rc) return rc;
    const TopkOut *out = reinterpret_cast<const TopkOut *>(m->h_pin + 4);
    topk_finish(out->v, out->i, out->M, out->S, temperature, k, ids_out, probs_out);
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_set_token(nfai_model_t h, uint32_t token)
{
    MODEL_OR_FAIL(m, h);
    if (token >= m->d.V) return fail(NFAI_ERR_INVALID, "set_token: token %u >= vocab %u", token, m->d.V);
    return set_token_async(m, token);
}

NFAI_API int32_t nfai_hip_llama_decode_enqueue(nfai_model_t h, uint32_t n_steps)
{
    MODEL_OR_FAIL(m, h);
    NEED_FINAL(m);
    if (!(m->first_stage && m->last_stage)) return fail(NFAI_ERR_STATE, "decode_enqueue: model is a pipeline stage");
    if (m->pos_host + n_steps > m->d.C)
        return fail(NFAI_ERR_KV_FULL, "decode_enqueue: %u steps from position %u exceed KV capacity %u", n_steps, m->pos_host, m->d.C);
    for (uint32_t i = 0; i < n_steps; i++) {
        int rc = run_token(m);
        if (rc) return rc;
    }
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_fetch_tokens(nfai_model_t h, uint32_t n, uint32_t *tokens_out)
{
    MODEL_OR_FAIL(m, h);
    if (!tokens_out || n == 0 || n > RING_LEN || n > m->pos_host) return fail(NFAI_ERR_INVALID, "fetch_tokens: n=%u (pos %u, ring %u)", n, m->pos_host, RING_LEN);
    hipStream_t s = m->ctx->stream;
    std::vector<uint32_t> ring(RING_LEN);
    HIP_TRY(hipMemcpyAsync(ring.data(), m->d_ring, RING_LEN * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(m->h_pin + 1, m->d_engerr, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (m->h_pin[1]) return engine_failed(m, m->h_pin[1]);
    for (uint32_t i = 0; i < n; i++) tokens_out[i] = ring[(m->pos_host - n + i) % RING_LEN];
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_decode_greedy(nfai_model_t h, uint32_t first_token, uint32_t n_steps, uint32_t *tokens_out)
{
    int rc = nfai_hip_llama_set_token(h, first_token);
    if (rc) return rc;
    if ((rc = nfai_hip_llama_decode_enqueue(h, n_steps))) return rc;
    return nfai_hip_llama_fetch_tokens(h, n_steps, tokens_out);
}

// One chunk of T prompt tokens through every block on the MFMA path (kernels_prefill.hip).
static int prefill_chunk(Model *m, const uint32_t *tokens, uint32_t T)
{
    const nfai_llama_desc &d = m->d;
    Model::Prefill &w = m->pf;
    hipStream_t s = m->ctx->stream;
    const uint32_t pos0 = m->pos_host, S = pos0 + T, Spad = (S + 63) / 64 * 64, HD = d.H * d.D, KD = d.Hkv * d.D;
    const uint32_t G = d.H / d.Hkv;
    const int kvf16 = m->kv_f16 ? 1 : 0;
#define P_TRY(expr)                                                                                               \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess)                                                                                     \
            return fail(_e == hipErrorInvalidValue ? NFAI_ERR_INVALID : NFAI_ERR_HIP, "prefill: %s failed: %s", #expr, \
                        hipGetErrorString(_e));                                                                   \
    } while (0)
    // One projection: up to three weight tensors side by side in the output columns.  Runs of tensors with the same
    // encoding share a launch (fp16: k_gemm_f16*, T16 K-quants: the dequant-in-LDS k_gemm_kq); a Q4_K_M q|k|v with a
    // Q6_K attn_v is two launches writing two column blocks of the same [T][ldc] buffer.
    // A projection whose K range was split (long chunks, below) may leave its slabs in w.SC for the NEXT block's attention norm to add
    // up (combine + RMSNorm in one pass): pend_ks > 0 until that launch, or the plain combine after the last block, has consumed them.
    uint32_t pend_ks = 0;
    const float *pend_R = nullptr;
    const bool fuse_combine = !(getenv("NFAI_PREFILL_COMBINE_FUSED") && atoi(getenv("NFAI_PREFILL_COMBINE_FUSED")) == 0);   // read per call (a test flips it)
    auto gemm = [&](const void *A, uint32_t lda, const Tensor &W, const Tensor *W1, const Tensor *W2, float *C, const float *R, uint32_t N,
                    uint32_t K, bool may_defer = false) -> hipError_t {
        const Tensor *seg[3] = {&W, W1, W2};
        const int nseg = W2 ? 3 : (W1 ? 2 : 1);
        uint32_t col = 0;
        for (int first = 0; first < nseg;) {
            int last = first;
            while (last + 1 < nseg && seg[last + 1]->type == seg[first]->type) last++;
            GemmArgs g;
            g.A = A; g.lda = lda; g.ldb = K; g.ldc = N; g.M = T; g.K = K;
            g.B = seg[first]->ptr;
            g.N = (uint32_t)seg[first]->rows;
            if (last > first) { g.B1 = seg[first + 1]->ptr; g.n0 = (uint32_t)seg[first]->rows; g.N += (uint32_t)seg[first + 1]->rows; }
            if (last > first + 1) { g.B2 = seg[first + 2]->ptr; g.n1 = (uint32_t)seg[first + 1]->rows; g.N += (uint32_t)seg[first + 2]->rows; }
            g.C = C + col;
            g.R = R ? R + col : nullptr;
            g.n_cu = (uint32_t)m->ctx->prop.multiProcessorCount;
            hipError_t e;
            // Short prompts (17 .. 128 rows; the provider path's templated chat prompts): one row of 128 x BN tiles is N / BN = 48-64
            // workgroups walking all of K — a quarter of the chip, 45 us for Wdown at 3B.  The K range is split over `ks` launches' worth of
            // workgroups instead (the GEMM's batch dimension: batch z multiplies columns [z K / ks, (z + 1) K / ks) of A and W into slab
            // z) and k_sum_slabs adds residual + slabs in order (deterministic).  One tensor, fp16, fp32 output only (Wo, Wdown).
            static const bool split_short = !(getenv("NFAI_PREFILL_SPLITK_SHORT") && atoi(getenv("NFAI_PREFILL_SPLITK_SHORT")) == 0);
            if (split_short && seg[first]->type == NFAI_F16 && nseg == 1 && T <= 128 && g.N % 64 == 0 && C != nullptr) {
                const uint64_t tiles = g.N / 64, n_cu = g.n_cu, sc_floats = (uint64_t)d.H * w.T * w.Spad;
                uint32_t best = 1;
                uint64_t best_cost = ((tiles + n_cu - 1) / n_cu) * K;
                for (uint32_t ks : {2u, 3u, 4u, 6u, 8u}) {
                    if (K % (ks * 128) || K / ks < 512 || (uint64_t)ks * T * g.N > sc_floats) continue;
                    const uint64_t cost = ((tiles * ks + n_cu - 1) / n_cu) * (K / ks);
                    if (cost < best_cost) { best = ks; best_cost = cost; }
                }
                if (best > 1) {
                    GemmArgs gs = g;
                    gs.batch = best; gs.K = K / best; gs.a_bs = K / best; gs.b_bs = K / best; gs.c_bs = (uint64_t)T * g.N;
                    gs.C = w.SC; gs.R = nullptr;
                    if ((e = launch_gemm_f16(gs, s)) != hipSuccess) return e;
                    if ((e = launch_sum_slabs(w.SC, best, (uint64_t)T * g.N, g.R, static_cast<float *>(g.C), s)) != hipSuccess) return e;
                    col += g.N;
                    first = last + 1;
                    continue;
                }
            }
            // Long chunks (>= 256 rows), K >= 8192 (Wdown): four K quarters on 256 x 128 tiles + the ordered combine (tools/gemm_bench.py
            // splitk4-proxy: 37.1 against 47.4 us at 3B before the combine).  NFAI_PREFILL_SPLITK_LONG=0 switches it off.
            static const bool split_long = !(getenv("NFAI_PREFILL_SPLITK_LONG") && atoi(getenv("NFAI_PREFILL_SPLITK_LONG")) == 0);
            if (split_long && seg[first]->type == NFAI_F16 && nseg == 1 && T >= 256 && ((T + 127) / 128) % 2 == 0 && K >= 8192 && K % 256 == 0 && g.N % 128 == 0 &&
                C != nullptr && (uint64_t)4 * T * g.N <= (uint64_t)d.H * w.T * w.Spad) {
                GemmArgs gs = g;
                gs.batch = 4; gs.K = K / 4; gs.a_bs = K / 4; gs.b_bs = K / 4; gs.c_bs = (uint64_t)T * g.N;
                gs.C = w.SC; gs.R = nullptr;
                if ((e = launch_gemm_f16(gs, s)) != hipSuccess) return e;
                if (may_defer && fuse_combine && g.R && d.E % 4 == 0 && d.E <= 4096 && g.N == d.E) {
                    pend_ks = 4;          // the next attention norm (or the tail of the chunk) adds residual + slabs into C = w.X
                    pend_R = g.R;
                } else if ((e = launch_sum_slabs(w.SC, 4, (uint64_t)T * g.N, g.R, static_cast<float *>(g.C), s)) != hipSuccess) {
                    return e;
                }
                col += g.N;
                first = last + 1;
                continue;
            }
            if (seg[first]->type == NFAI_F16) {
                e = launch_gemm_f16(g, s);
            } else {
                g.b_type = seg[first]->type;
                e = launch_gemm_kq(g, s);
            }
            if (e != hipSuccess) return e;
            col += g.N;
            first = last + 1;
        }
        return hipSuccess;
    };
    // K-quant blocks, two implementations.  Default: widen the block's matrices into an fp16 scratch (13 us per matrix) and use
    // the direct-to-LDS fp16 GEMMs — 8.7 ms per 512 tokens at 3B Q4_K_M.  NFAI_PREFILL_FUSED=1: the dequant-in-LDS GEMM
    // (k_gemm_kq: quant bytes -> VGPR -> fp16 tile in LDS, no scratch, no extra HBM traffic) — 9.4 ms: its register-staged A
    // operand and ~80 VALU operations of dequantisation per 16 weights cost more than the widening pass saves (measured).
    static const bool widen = !(getenv("NFAI_PREFILL_FUSED") && atoi(getenv("NFAI_PREFILL_FUSED")));
    const uint32_t QKV = HD + 2 * KD;
    // NFAI_PREFILL_READAHEAD=1 (off by default): read-ahead of the next GEMM's fp16 weights on the side stream (kernels_prefill.hip:
    // k_read_ahead), issued when the GEMM in front of it starts, so at most two matrices' worth of bytes (<= 150 MB at 3B) compete for
    // the 256 MB Infinity Cache.  Built because the projections run 15-40 % faster on cache-resident weights (tools/gemm_bench.py);
    // measured in the prefill it LOSES: 6.39-6.44 ms against 5.93 ms per 512 tokens at 3B — beside a GEMM that lives on L2 hits the
    // read-ahead's own HBM stream costs more than the first-use latency it removes (as the side-stream widening did in round 2).
    static const bool read_ahead = getenv("NFAI_PREFILL_READAHEAD") && atoi(getenv("NFAI_PREFILL_READAHEAD")) == 1;
    size_t ra_ev = 0;
    bool ra_used = false;
    auto ahead = [&](std::initializer_list<const Tensor *> ts) -> int {
        if (!read_ahead || !m->s2) return NFAI_OK;
        bool any = false;
        for (const Tensor *t : ts) any = any || (t->ptr && t->type == NFAI_F16);
        if (!any) return NFAI_OK;
        if (ra_ev >= m->pf_events.size()) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            m->pf_events.push_back(e);
        }
        hipEvent_t ev = m->pf_events[ra_ev++];
        HIP_TRY(hipEventRecord(ev, s));              // everything enqueued so far: the read-ahead starts with the GEMM in front of it
        HIP_TRY(hipStreamWaitEvent(m->s2, ev, 0));
        for (const Tensor *t : ts)
            if (t->ptr && t->type == NFAI_F16) {
                hipError_t e = launch_read_ahead(t->ptr, t->rows * t->cols * 2, (uint32_t)m->ctx->prop.multiProcessorCount, m->s2);
                if (e != hipSuccess) return fail(NFAI_ERR_HIP, "prefill: read-ahead launch failed: %s", hipGetErrorString(e));
            }
        ra_used = true;
        return NFAI_OK;
    };
    HIP_TRY(hipMemcpyAsync(w.toks, tokens, (size_t)T * 4, hipMemcpyHostToDevice, s));
    if (is_kquant(m->token_embd.type))
        P_TRY(launch_embed_rows_kqt(m->token_embd.ptr, m->token_embd.type, m->token_embd.rows, w.toks, w.X, T, d.E, s));
    else
        P_TRY(launch_embed_rows(m->token_embd.ptr, m->token_embd.type, w.toks, w.X, T, d.E, s));
    // RoPE and the q / KV-cache stores in the q | k | v GEMM's epilogue (fp16 weights, also widened ones); NFAI_PREFILL_ROPE_FUSED=0:
    // GEMM -> fp32 q | k | v -> k_rope_store_tiles (bit-identical results, one launch and a 10 MB round trip more per block)
    const char *env_rf = getenv("NFAI_PREFILL_ROPE_FUSED");
    const bool rope_fused_ok = !(env_rf && atoi(env_rf) == 0) && d.D % 16 == 0 && d.rope_dims % 2 == 0;
    if (rope_fused_ok) P_TRY(launch_rope_table(m->d_freqs, pos0, T, d.D, d.rope_dims, w.CS, s));
    for (Layer &Lq : m->layers) {
        Layer L = Lq;
        if (widen && w.WF16) {
            const size_t li = (size_t)(&Lq - m->layers.data());
            const bool kept = w.wf16_all && w.wf16_done[li];  // widened by an earlier chunk / prefill and still current
            uint64_t off = w.wf16_all ? li * w.wf16_slot : 0;
            const uint64_t end = off + w.wf16_slot;
            for (Tensor *tq : {&L.wq, &L.wk, &L.wv, &L.wo, &L.wgate, &L.wup, &L.wdown}) {
                if (tq->type == NFAI_F16) continue;
                const uint64_t bytes = tq->rows * tq->cols * 2;
                if (off + bytes > end) return fail(NFAI_ERR_STATE, "prefill: fp16 weight scratch too small");
                void *dst = static_cast<uint8_t *>(w.WF16) + off;
                if (!kept) P_TRY(launch_dequant_t16_f16(tq->ptr, tq->type, tq->rows, tq->cols, dst, s));
                tq->ptr = dst; tq->type = NFAI_F16; tq->owned = false;
                off += (bytes + 255) / 256 * 256;
            }
            if (w.wf16_all) w.wf16_done[li] = 1;
        }
        if (pend_ks) {   // the previous block's Wdown left residual + K-split slabs: combine -> w.X and normalise in one pass
            P_TRY(launch_rmsnorm_rows_combine(w.SC, pend_ks, pend_R, w.X, static_cast<const float *>(L.attn_norm.ptr), w.XN, T, d.E, d.eps, s));
            pend_ks = 0;
        } else {
            P_TRY(launch_rmsnorm_rows(w.X, static_cast<const float *>(L.attn_norm.ptr), w.XN, T, d.E, d.eps, s));
        }
        S_TRY(ahead({&L.wo}));                                                           // while q | k | v computes
        if (rope_fused_ok && L.wq.type == NFAI_F16 && L.wk.type == NFAI_F16 && L.wv.type == NFAI_F16) {
            GemmArgs g;                                                                  // q | k | v + RoPE + q / cache stores in one launch
            g.A = w.XN; g.lda = d.E; g.ldb = d.E; g.M = T; g.N = QKV; g.K = d.E;
            g.B = L.wq.ptr; g.B1 = L.wk.ptr; g.B2 = L.wv.ptr; g.n0 = HD; g.n1 = KD;
            g.epi = 3;
            g.n_cu = (uint32_t)m->ctx->prop.multiProcessorCount;
            g.rope.cs = w.CS; g.rope.qh = w.QH; g.rope.kh = w.KH; g.rope.vt = w.VT; g.rope.kc = L.kcache; g.rope.vc = L.vcache;
            g.rope.pos_stride = m->kv_pos_stride; g.rope.head_stride = m->kv_head_stride;
            g.rope.H = d.H; g.rope.Hkv = d.Hkv; g.rope.D = d.D; g.rope.rope_dims = d.rope_dims; g.rope.pos0 = pos0; g.rope.Spad = Spad;
            g.rope.kv_f16 = (uint32_t)kvf16;
            P_TRY(launch_gemm_f16(g, s));
        } else {
            P_TRY(gemm(w.XN, d.E, L.wq, &L.wk, &L.wv, w.Q, nullptr, QKV, d.E));          // q | k | v in one launch
            P_TRY(launch_rope_store_rows(w.Q, w.Q + HD, w.Q + HD + KD, w.QH, L.kcache, L.vcache, kvf16, m->kv_pos_stride, m->kv_head_stride,
                                         m->d_freqs, d.rope_dims, d.H, d.Hkv, d.D, pos0, T, QKV, w.KH, w.VT, Spad, s));
        }
        // earlier positions (chunked prompts) and the zero padding; the chunk's own rows were written above
        P_TRY(launch_kv_to_f16(L.kcache, L.vcache, kvf16, m->kv_pos_stride, m->kv_head_stride, w.KH, w.VT, d.Hkv, d.D, S, Spad, pos0, S, s));
        // attention of the chunk.  Default: one launch (k_attn_prefill: scores, causal softmax and weighted V with the probabilities
        // kept in registers); NFAI_PREFILL_FLASH=0: Q.K^T GEMM -> row softmax -> P.V GEMM with materialised scores.
        static const bool flash = !(getenv("NFAI_PREFILL_FLASH") && atoi(getenv("NFAI_PREFILL_FLASH")) == 0);
        if (flash) {
            P_TRY(launch_attn_prefill(w.QH, w.KH, w.VT, w.XN, T, d.H, d.Hkv, d.D, Spad, pos0, s));
        } else {
            {   // scores[h][t][s] = q_h[t] . k_kvh[s]   (scaling and the causal limit are applied by the softmax)
                GemmArgs g;
                g.A = w.QH; g.lda = HD; g.a_bs = d.D;
                g.B = w.KH; g.ldb = d.D; g.b_bs = (uint64_t)Spad * d.D; g.b_div = G;
                g.C = w.SC; g.ldc = Spad; g.c_bs = (uint64_t)T * Spad;
                g.M = T; g.N = Spad; g.K = d.D; g.batch = d.H;
                g.causal = 1; g.causal_pos0 = pos0;
                P_TRY(launch_gemm_f16(g, s));
            }
            P_TRY(launch_softmax_causal_rows(w.SC, w.P, d.H, T, Spad, pos0, 1.0f / sqrtf((float)d.D), s));
            {   // att[t][h*D + d] = sum_s P[h][t][s] * V_kvh[s][d]
                GemmArgs g;
                g.A = w.P; g.lda = Spad; g.a_bs = (uint64_t)T * Spad;
                g.B = w.VT; g.ldb = Spad; g.b_bs = (uint64_t)d.D * Spad; g.b_div = G;
                g.C = w.XN; g.epi = 1; g.ldc = HD; g.c_bs = d.D;   // fp16 straight into the Wo GEMM's A operand
                g.M = T; g.N = d.D; g.K = Spad; g.batch = d.H;
                g.causal = 2; g.causal_pos0 = pos0;
                P_TRY(launch_gemm_f16(g, s));
            }
        }
        S_TRY(ahead({&L.wgate, &L.wup}));                                                // while Wo computes
        P_TRY(gemm(w.XN, HD, L.wo, nullptr, nullptr, w.H1, w.X, d.E, HD));                 // + residual (TransformerBlock.cs:153-158)
        P_TRY(launch_rmsnorm_rows(w.H1, static_cast<const float *>(L.ffn_norm.ptr), w.XN, T, d.E, d.eps, s));
        S_TRY(ahead({&L.wdown}));                                                        // while gate | up computes
        {   // gate | up in one launch, act = up * silu(gate) formed in the GEMM epilogue (fp16 [T][F])
            GemmArgs g;
            g.A = w.XN; g.lda = d.E; g.B = L.wgate.ptr; g.B1 = L.wup.ptr; g.n0 = d.F; g.ldb = d.E;
            g.C = w.ACT; g.epi = 2; g.ldc = d.F;
            g.M = T; g.N = 2 * d.F; g.K = d.E;
            g.n_cu = (uint32_t)m->ctx->prop.multiProcessorCount;
            if (L.wgate.type == NFAI_F16) {
                P_TRY(launch_gemm_f16(g, s));
            } else {
                g.b_type = L.wgate.type;  // finalize() guarantees gate and up share an encoding
                P_TRY(launch_gemm_kq(g, s));
            }
        }
        if (&Lq != &m->layers.back()) {                                                  // while Wdown computes: the next block's q, k, v
            const Layer &N = *(&Lq + 1);
            S_TRY(ahead({&N.wq, &N.wk, &N.wv}));
        }
        P_TRY(gemm(w.ACT, d.F, L.wdown, nullptr, nullptr, w.X, w.H1, d.E, d.F, true));     // + residual (:176-181)
    }
    if (pend_ks) {   // the last block of the stage: nobody normalises behind it
        P_TRY(launch_sum_slabs(w.SC, pend_ks, (uint64_t)T * d.E, pend_R, w.X, s));
        pend_ks = 0;
    }
    if (ra_used) {  // the side stream only reads weights; the join keeps destroy / set_tensor from racing with it
        if (ra_ev >= m->pf_events.size()) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            m->pf_events.push_back(e);
        }
        hipEvent_t ev = m->pf_events[ra_ev++];
        HIP_TRY(hipEventRecord(ev, m->s2));
        HIP_TRY(hipStreamWaitEvent(s, ev, 0));
    }
#undef P_TRY
    // the last token's hidden state continues on the M = 1 path (output norm + lm_head + argmax)
    HIP_TRY(hipMemcpyAsync(m->x, w.X + (size_t)(T - 1) * d.E, (size_t)d.E * 4, hipMemcpyDeviceToDevice, s));
    const uint32_t newpos = pos0 + T;
    HIP_TRY(hipMemcpyAsync(m->d_pos, &newpos, 4, hipMemcpyHostToDevice, s));
    m->pos_host = newpos;
    m->x_last = m->x;
    return NFAI_OK;
}

static bool prefill_mfma_ok(const Model *m)
{
    if (m->pf.T == 0 || m->unfused || !(m->first_stage && m->last_stage)) return false;
    const nfai_llama_desc &d = m->d;
    if (d.E % 64 || d.F % 64 || (d.H * d.D) % 64 || (d.Hkv * d.D) % 64) return false;
    const int et = m->token_embd.type;
    if (et != NFAI_F16 && et != NFAI_F32 && et != NFAI_Q4_K_T16 && et != NFAI_Q6_K_T16) return false;
    for (const Layer &L : m->layers)
        for (const Tensor *t : {&L.wq, &L.wk, &L.wv, &L.wo, &L.wgate, &L.wup, &L.wdown})
            if (t->type != NFAI_F16 && t->type != NFAI_Q4_K_T16 && t->type != NFAI_Q6_K_T16) return false;
    return true;
}

// head = false: only the KV cache is filled (nfai_hip_llama_ingest: prompt tokens whose output the reference's loop discards).
static int prefill_impl(nfai_model_t h, const uint32_t *tokens, uint32_t n, float *logits_last_host, bool head)
{
    MODEL_OR_FAIL(m, h);
    NEED_FINAL(m);
    if (!tokens || n == 0) return fail(NFAI_ERR_INVALID, "prefill: empty prompt");
    if (m->pos_host + n > m->d.C) return fail(NFAI_ERR_KV_FULL, "prefill: %u tokens from position %u exceed KV capacity %u", n, m->pos_host, m->d.C);
    for (uint32_t i = 0; i < n; i++)
        if (tokens[i] >= m->d.V) return fail(NFAI_ERR_INVALID, "prefill: token %u >= vocab %u", tokens[i], m->d.V);
    if (!prefill_mfma_ok(m)) {
        // no MFMA workspace / non-fp16 weights: the prompt goes through the M = 1 path token by token,
        // exactly as the reference feeds it (LlamaModel.cs:103-126)
        for (uint32_t i = 0; i < n; i++) {
            int rc = nfai_hip_llama_decode_step(h, tokens[i], i + 1 == n ? logits_last_host : nullptr, nullptr);
            if (rc) return rc;
        }
        return NFAI_OK;
    }
    hipStream_t s = m->ctx->stream;
    if (!m->pf.WF16 && !(getenv("NFAI_PREFILL_FUSED") && atoi(getenv("NFAI_PREFILL_FUSED")))) {  // fp16 scratch for one block's matrices (K-quant models)
        uint64_t need = 0;
        for (const Layer &L : m->layers) {
            uint64_t b = 0;
            for (const Tensor *t : {&L.wq, &L.wk, &L.wv, &L.wo, &L.wgate, &L.wup, &L.wdown})
                if (t->type != NFAI_F16) b += (t->rows * t->cols * 2 + 255) / 256 * 256;
            need = std::max(need, b);
        }
        if (need) {
            // Keep every block's fp16 copy (widened once, at the first prefill) when all of them fit a quarter of the device's memory
            // and leave 4 GB free: the per-block widening is a quarter of a K-quant prefill (64 us of 230 per block at 3B).  The decode
            // path never reads these copies.  NFAI_PREFILL_WIDE_ALL=0 / 1 forces one slot / all slots.
            const uint64_t all = need * m->layers.size();
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            const char *env = getenv("NFAI_PREFILL_WIDE_ALL");
            const bool fits = all + (4ull << 30) <= free_b;
            const bool want = env ? atoi(env) != 0 : all <= total_b / 4;
            m->pf.wf16_all = want && fits;
            m->pf.wf16_slot = need;
            m->pf.wf16_bytes = m->pf.wf16_all ? all : need;
            DALLOC(m->pf.WF16, m->pf.wf16_bytes);
            m->pf.wf16_done.assign(m->layers.size(), 0);
        }
    }
    for (uint32_t done = 0; done < n;) {
        const uint32_t T = std::min(n - done, m->d.max_batch);
        int rc = prefill_chunk(m, tokens + done, T);
        if (rc) return rc;
        done += T;
    }
    // logits of the LAST prompt token: output norm + lm_head + argmax on its hidden state.  The
    // position was already advanced past the prompt, so the head runs without the token bookkeeping.
    if (head) {
        Rec rec{m};
        const Tensor &head = m->output.ptr ? m->output : m->token_embd;
        GemvArgs a = gemv_base(m, head, m->x, m->d.E);
        a.gamma = static_cast<const float *>(m->output_norm.ptr);
        a.y = m->logits;
        const bool am_fused = head.type == NFAI_F16 || head.type == NFAI_F32 || head.type == NFAI_Q4_K_T16 || head.type == NFAI_Q6_K_T16;
        if (am_fused) {  // ArgMax in the lm_head launch, as in a decode step; no bookkeeping: the position was set above
            a.argmax_part = static_cast<char *>(m->d_argmax_part) + 4096;
            a.argmax_out = m->d_tok;
        }
        K_TRY(KC_LMHEAD, launch_gemv(a, s));
        if (!am_fused) K_TRY(KC_OTHER, launch_argmax(m->logits, m->d.V, m->d_tok, m->d_argmax_part, nullptr, nullptr, 0, s));
    }
    if (head && logits_last_host) HIP_TRY(hipMemcpyAsync(logits_last_host, m->logits, (size_t)m->d.V * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));  // `tokens` is the caller's (pageable) memory: the copy into the workspace has left it by now
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_prefill(nfai_model_t h, const uint32_t *tokens, uint32_t n, float *logits_last_host)
{
    return prefill_impl(h, tokens, n, logits_last_host, true);
}

// The prompt phase of LlamaModel.RunAsync (LlamaModel.cs:103-126): every prompt token but the last only has to leave its K and V
// rows behind — the loop overwrites the logits of token i with those of token i + 1 and samples once, after the last one (:128-130).
NFAI_API int32_t nfai_hip_llama_ingest(nfai_model_t h, const uint32_t *tokens, uint32_t n)
{
    if (n == 0) {   // a one-token prompt has nothing in front of the sampled step
        MODEL_OR_FAIL(m, h);
        NEED_FINAL(m);
        return NFAI_OK;
    }
    return prefill_impl(h, tokens, n, nullptr, false);
}

// The stage's work for one token: hidden state in -> this stage's blocks -> hidden state out
// (or lm_head + argmax on the last stage).  Captured once per (hidden_in, hidden_out) pair.
static int stage_enqueue(Model *m, const void *hidden_in, void *hidden_out)
{
    hipStream_t s = m->ctx->stream;
    if (!m->first_stage) HIP_TRY(hipMemcpyAsync(m->x, hidden_in, (size_t)m->d.E * 4, hipMemcpyDeviceToDevice, s));
    int rc = enqueue_token(m, true);
    if (rc) return rc;
    if (!m->last_stage) HIP_TRY(hipMemcpyAsync(hidden_out, m->x_last ? m->x_last : m->x, (size_t)m->d.E * 4, hipMemcpyDeviceToDevice, s));
    // the sticky error word of the launches that wait inside the kernel (attention slices, engine) travels to pinned host memory
    // with every step; nfai_hip_llama_stage_step looks at it on entry, so a stage that never synchronises still reports a
    // hand-off that gave up — one step late
    HIP_TRY(hipMemcpyAsync(m->h_pin + 1, m->d_engerr, 4, hipMemcpyDeviceToHost, s));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_stage_step(nfai_model_t h, uint32_t token, const void *hidden_in, void *hidden_out,
                                           float *logits_host, uint32_t *argmax)
{
    MODEL_OR_FAIL(m, h);
    NEED_FINAL(m);
    hipStream_t s = m->ctx->stream;
    if (m->pos_host >= m->d.C) return fail(NFAI_ERR_KV_FULL, "stage_step: KV cache full at position %u", m->pos_host);
    if (m->h_pin[1]) return engine_failed(m, m->h_pin[1]);  // written by an earlier step of this stage (see stage_enqueue)
    if (m->first_stage) {
        if (token != NFAI_TOKEN_ON_DEVICE) {
            if (token >= m->d.V) return fail(NFAI_ERR_INVALID, "stage_step: token %u >= vocab %u", token, m->d.V);
            int rc = set_token_async(m, token);
            if (rc) return rc;
        }
    } else if (!hidden_in) {
        return fail(NFAI_ERR_INVALID, "stage_step: hidden_in is required on a non-first stage");
    }
    if (!m->last_stage && !hidden_out) return fail(NFAI_ERR_INVALID, "stage_step: hidden_out is required on a non-last stage");
    if (m->use_graph && !m->unfused) {
        if (!m->stage_exec || m->stage_in != hidden_in || m->stage_out != hidden_out) {
            if (m->stage_exec) { hipGraphExecDestroy(m->stage_exec); m->stage_exec = nullptr; }
            if (m->stage_graph) { hipGraphDestroy(m->stage_graph); m->stage_graph = nullptr; }
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            int rc = stage_enqueue(m, hidden_in, hidden_out);
            hipGraph_t g = nullptr;
            hipError_t e = hipStreamEndCapture(s, &g);
            if (rc) { if (g) hipGraphDestroy(g); return rc; }
            if (e != hipSuccess) return fail(NFAI_ERR_HIP, "stage_step: hipStreamEndCapture failed: %s", hipGetErrorString(e));
            m->stage_graph = g;
            HIP_TRY(hipGraphInstantiate(&m->stage_exec, g, nullptr, nullptr, 0));
            m->stage_in = hidden_in;
            m->stage_out = hidden_out;
        }
        HIP_TRY(hipGraphLaunch(m->stage_exec, s));
    } else {
        int rc = stage_enqueue(m, hidden_in, hidden_out);
        if (rc) return rc;
    }
    m->pos_host++;
    if (m->last_stage && (logits_host || argmax)) {
        if (logits_host) HIP_TRY(hipMemcpyAsync(logits_host, m->logits, (size_t)m->d.V * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(m->h_pin, m->d_tok, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (argmax) *argmax = m->h_pin[0];
    }
    return NFAI_OK;
}

// Token hand-over between pipeline ends without a host round trip: 4-byte device-to-device copies
// on the context stream (the last stage's argmax -> a buffer RCCL sends; the received buffer ->
// the first stage's token word).
NFAI_API int32_t nfai_hip_llama_token_to_device(nfai_model_t h, void *dst_dev)
{
    MODEL_OR_FAIL(m, h);
    if (!dst_dev) return fail(NFAI_ERR_INVALID, "token_to_device: null pointer");
    HIP_TRY(hipMemcpyAsync(dst_dev, m->d_tok, 4, hipMemcpyDeviceToDevice, m->ctx->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_token_from_device(nfai_model_t h, const void *src_dev)
{
    MODEL_OR_FAIL(m, h);
    if (!src_dev) return fail(NFAI_ERR_INVALID, "token_from_device: null pointer");
    HIP_TRY(hipMemcpyAsync(m->d_tok, src_dev, 4, hipMemcpyDeviceToDevice, m->ctx->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_set_pos(nfai_model_t h, uint32_t pos)
{
    MODEL_OR_FAIL(m, h);
    if (pos > m->d.C) return fail(NFAI_ERR_INVALID, "set_pos: %u > capacity %u", pos, m->d.C);
    HIP_TRY(hipMemcpyAsync(m->d_pos, &pos, 4, hipMemcpyHostToDevice, m->ctx->stream));
    HIP_TRY(hipMemsetAsync(m->d_engerr, 0, 4, m->ctx->stream));
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    m->h_pin[1] = 0;  // the host mirror of the error word (after the stream is idle: no copy into it is pending)
    m->pos_host = pos;
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_reset(nfai_model_t h) { return nfai_hip_llama_set_pos(h, 0); }

NFAI_API int32_t nfai_hip_llama_pos(nfai_model_t h, uint32_t *pos)
{
    MODEL_OR_FAIL(m, h);
    if (pos) *pos = m->pos_host;
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_read(nfai_model_t h, int32_t which, float *host, uint64_t n)
{
    MODEL_OR_FAIL(m, h);
    const nfai_llama_desc &d = m->d;
    const float *src = nullptr;
    uint64_t cap = 0;
    switch (which) {
        case 0: src = m->x_last ? m->x_last : m->x; cap = d.E; break;
        case 1: src = m->q; cap = (uint64_t)d.H * d.D; break;
        case 2: src = m->att; cap = (uint64_t)d.H * d.D; break;
        case 3: src = m->act; cap = d.F; break;
        case 4: src = m->logits; cap = d.V; break;
    }
    if (!src || !host || n > cap) return fail(NFAI_ERR_INVALID, "llama_read: which=%d n=%llu", which, (unsigned long long)n);
    HIP_TRY(hipMemcpyAsync(host, src, n * 4, hipMemcpyDeviceToHost, m->ctx->stream));
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_read_kv(nfai_model_t h, uint32_t layer, int32_t is_v, uint32_t pos, float *host)
{
    MODEL_OR_FAIL(m, h);
    const nfai_llama_desc &d = m->d;
    if (layer < d.layer_begin || layer >= d.layer_end || pos >= d.C || !host) return fail(NFAI_ERR_INVALID, "read_kv: layer %u pos %u", layer, pos);
    Layer &L = m->layers[layer - d.layer_begin];
    const char *base = static_cast<const char *>(is_v ? L.vcache : L.kcache);
    hipStream_t s = m->ctx->stream;
    std::vector<uint16_t> tmp16;
    if (m->kv_f16) tmp16.resize((size_t)d.Hkv * d.D);
    for (uint32_t kh = 0; kh < d.Hkv; kh++) {
        const uint64_t idx = (uint64_t)pos * m->kv_pos_stride + (uint64_t)kh * m->kv_head_stride;
        void *dst = m->kv_f16 ? static_cast<void *>(tmp16.data() + (size_t)kh * d.D) : static_cast<void *>(host + (size_t)kh * d.D);
        HIP_TRY(hipMemcpyAsync(dst, base + idx * m->kv_esz, (size_t)d.D * m->kv_esz, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (m->kv_f16)
        for (size_t i = 0; i < tmp16.size(); i++) {
            _Float16 hv;
            memcpy(&hv, &tmp16[i], 2);
            host[i] = (float)hv;
        }
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_bytes_per_token(nfai_model_t h, uint32_t pos, uint64_t *total, uint64_t *dominant)
{
    MODEL_OR_FAIL(m, h);
    const nfai_llama_desc &d = m->d;
    uint64_t t = 0, dom = 0;
    for (const Layer &L : m->layers) {
        const uint64_t qkv = tensor_bytes(L.wq) + tensor_bytes(L.wk) + tensor_bytes(L.wv);
        const uint64_t gu = tensor_bytes(L.wgate) + tensor_bytes(L.wup);
        t += qkv + gu + tensor_bytes(L.wo) + tensor_bytes(L.wdown);
        // KV: read p+1 positions, write 1
        t += 2ull * d.Hkv * d.D * m->kv_esz * ((uint64_t)pos + 1) + 2ull * d.Hkv * d.D * m->kv_esz;
        if (gu > dom) dom = gu;
        if (qkv > dom) dom = qkv;
    }
    if (m->first_stage) t += weight_row_bytes(m->token_embd.type, d.E);
    if (m->last_stage) {
        const Tensor &head = m->output.ptr ? m->output : m->token_embd;
        t += tensor_bytes(head);
    }
    if (total) *total = t;
    if (dominant) *dominant = dom;
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_profile_kernel(nfai_model_t h, uint32_t token, int32_t cls, uint32_t reps, float *us_avg)
{
    MODEL_OR_FAIL(m, h);
    NEED_FINAL(m);
    if (!us_avg || reps == 0 || cls < 0 || cls >= KC_N) return fail(NFAI_ERR_INVALID, "profile_kernel: bad arguments");
    if (!(m->first_stage && m->last_stage)) return fail(NFAI_ERR_STATE, "profile_kernel: whole-model contexts only");
    if (m->unfused) return fail(NFAI_ERR_STATE, "profile_kernel: fused path only");
    int rc = set_token_async(m, token);
    if (rc) return rc;
    for (hipEvent_t e : m->ev) hipEventDestroy(e);
    m->ev.clear();
    m->ev_class.clear();
    for (hipEvent_t &e : m->prof_rep_ev)
        if (!e) HIP_TRY(hipEventCreate(&e));
    m->prof_rep_cls = cls;
    m->prof_ops.clear();
    m->profiling = true;
    rc = run_token(m);
    m->profiling = false;
    m->prof_rep_cls = -1;
    if (rc) return rc;
    if (m->prof_ops.empty()) return fail(NFAI_ERR_STATE, "profile_kernel: no launch of class %d in a step", cls);
    // Replay: every launch of the class in a step (one per block: 28 different weight sets at 3B, far beyond the 256 MB
    // Infinity Cache, so nothing is re-read from cache), `reps` rounds, back to back between ONE pair of events.  Decode
    // launches are idempotent (outputs never alias inputs), so the replay leaves the model state as the step left it.
    hipStream_t s = m->ctx->stream;
    // Engine launches hand vectors over under a per-token tag: a replay with the tag unchanged would find every hand-off already
    // complete and never wait.  So each replayed engine launch is preceded by the one-block kernel that advances the tag (as in
    // a real token), and the same number of those kernels alone is timed and subtracted.
    const bool bump = cls == KC_ENGINE;
    auto bump_launch = [&]() { return launch_token_begin(nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, m->d_pos, s, m->d_epoch); };
    HIP_TRY(hipEventRecord(m->prof_rep_ev[0], s));
    for (uint32_t r = 0; r < reps; r++)
        for (const Op &op : m->prof_ops) {
            if (bump && bump_launch() != hipSuccess) return fail(NFAI_ERR_HIP, "profile_kernel: tag launch failed");
            GemvArgs g = op.g;
            g.argmax_part = nullptr;  // a replayed lm_head must not advance the position again (the GEMV itself is idempotent)
            hipError_t e = op.kind == 0 ? launch_gemv(g, s) : (op.kind == 1 ? launch_attn_decode(op.a, s) : op.f(s));
            if (e != hipSuccess) return fail(NFAI_ERR_HIP, "profile_kernel: replay launch failed: %s", hipGetErrorString(e));
        }
    HIP_TRY(hipEventRecord(m->prof_rep_ev[1], s));
    HIP_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, m->prof_rep_ev[0], m->prof_rep_ev[1]));
    if (bump) {
        HIP_TRY(hipEventRecord(m->prof_rep_ev[0], s));
        for (size_t i = 0; i < (size_t)reps * m->prof_ops.size(); i++)
            if (bump_launch() != hipSuccess) return fail(NFAI_ERR_HIP, "profile_kernel: tag launch failed");
        HIP_TRY(hipEventRecord(m->prof_rep_ev[1], s));
        HIP_TRY(hipStreamSynchronize(s));
        float ms0 = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms0, m->prof_rep_ev[0], m->prof_rep_ev[1]));
        ms -= ms0;
    }
    *us_avg = ms * 1e3f / (float)((size_t)reps * m->prof_ops.size());
    m->prof_ops.clear();
    return NFAI_OK;
}

NFAI_API int32_t nfai_hip_llama_profile_step(nfai_model_t h, uint32_t token, float *ms_by_class, uint32_t *launches_by_class)
{
    MODEL_OR_FAIL(m, h);
    NEED_FINAL(m);
    if (!ms_by_class || !launches_by_class) return fail(NFAI_ERR_INVALID, "profile_step: null output");
    if (!(m->first_stage && m->last_stage)) return fail(NFAI_ERR_STATE, "profile_step: whole-model contexts only");
    int rc = set_token_async(m, token);
    if (rc) return rc;
    for (hipEvent_t e : m->ev) hipEventDestroy(e);
    m->ev.clear();
    m->ev_class.clear();
    m->profiling = true;
    rc = run_token(m);
    m->profiling = false;
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    for (int i = 0; i < KC_N; i++) { ms_by_class[i] = 0.f; launches_by_class[i] = 0; }
    for (size_t i = 0; i < m->ev_class.size(); i++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, m->ev[2 * i], m->ev[2 * i + 1]));
        ms_by_class[m->ev_class[i]] += ms;
        launches_by_class[m->ev_class[i]]++;
    }
    return NFAI_OK;
}

// Test hook (not in nfai_hip.h): slice `slice_plus_1 - 1` of kv head 0 of every attention launch publishes nothing, so the bounded
// waits of the granule hand-off give up and the fall-back to the ticket form can be exercised (tests/test_gpu_model.py).  0: off.
NFAI_API int32_t nfai_hip_debug_attn_withhold(nfai_model_t h, uint32_t slice_plus_1)
{
    MODEL_OR_FAIL(m, h);
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    m->dbg_withhold = slice_plus_1;
    drop_graphs(m);
    return NFAI_OK;
}
