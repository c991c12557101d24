// common.h — shared host/device helpers of libnfai_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/nfai_hip.h"

#define NFAI_API extern "C" __attribute__((visibility("default")))

namespace nfai {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return nfai::fail(_e == hipErrorOutOfMemory ? NFAI_ERR_OOM : NFAI_ERR_HIP,         \
                              "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                       \
    } while (0)

#define NFAI_REQUIRE(cond, ...)                                      \
    do {                                                             \
        if (!(cond)) return nfai::fail(NFAI_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// ---- in-kernel time stamps (diagnostic build only: python -m nfai_amd.build --stamps -> libnfai_hip_stamps.so) -------------
// Every wave of a stamped kernel records up to 8 readings of s_memrealtime (100 MHz, one clock for the whole chip) into its
// launch's slot of a buffer the tool hands in; no stamp exists in the product build (MI355X guide, "In-kernel stamps").
#ifdef NFAI_STAMPS
constexpr uint32_t STAMP_WAVES = 4096, STAMP_WORDS = 8;
struct StampSlot { char name[48]; uint32_t grid, block; };
unsigned long long *stamp_next_slot(const char *name, uint32_t grid, uint32_t block);  // null when no buffer is installed
#define NFAI_STAMP_PARAM unsigned long long *stamps;
#define NFAI_STAMP_SET(p, name, grid, block) (p).stamps = nfai::stamp_next_slot(name, grid, block)
#else
#define NFAI_STAMP_PARAM
#define NFAI_STAMP_SET(p, name, grid, block) ((void)0)
#endif

// ---- handles --------------------------------------------------------------------------------
struct Ctx {
    uint32_t magic = 0x4E464358;  // 'NFCX'
    int device = 0;
    // per-XCD streaming shares (gemv_xcd_calibrate, first model created on the context): 0 = not yet measured, 1 = measured, 2 = off
    int xcd_state = 0;
    uint16_t xcd_shares[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float xcd_probe_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // mean end of a workgroup of each label in the probe launch (equal shares)
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipDeviceProp_t prop;
    void *scratch = nullptr;  // small workspace for ops that need one (argmax partials, attention partials)
    size_t scratch_bytes = 0;
};

struct Buf {
    uint32_t magic = 0x4E464246;  // 'NFBF'
    int w_layout = 0;  // internal weight-type code when nfai_hip_weight_upload repacked the tensor (else 0)
    uint64_t w_rows = 0;  // rows of that tensor (the T16 planes depend on it)
    void *ptr = nullptr;
    uint64_t bytes = 0;
    bool owned = true;
    Ctx *ctx = nullptr;
    void *guard_base = nullptr;  // NFAI_HIP_DEBUG_CANARY: start of the allocation (256-byte guard | buffer | 256-byte guard)
};

Ctx *ctx_of(nfai_ctx_t h);
Buf *buf_of(nfai_buf_t h);
void handle_register(const void *p);
void handle_unregister(const void *p);
bool handle_live(uint64_t h);

uint64_t weight_row_bytes(int ggml_type, uint64_t n_cols);  // 0 if unsupported / misaligned

// ---- launch entry points implemented in the kernel translation units --------------------------
// Plain pointers; every function enqueues on `s` and returns hipGetLastError().
struct GemvArgs {
    // up to three weight segments (QKV) or two (gate/up); rows are [seg_rows[i]] each, all K wide
    const void *W[3] = {nullptr, nullptr, nullptr};
    uint32_t seg_rows[3] = {0, 0, 0};
    int w_type = NFAI_F16;
    uint32_t seg6_mask = 0;        // NFAI_KQ_MIXED: bit i = segment i is Q6_K_T16
    const float *x = nullptr;      // K floats
    const float *gamma = nullptr;  // non-null => RMSNorm(x, gamma, eps) prologue
    float eps = 0.f;
    uint32_t K = 0;
    // epilogues
    int mode = 0;                  // GemvMode
    float *y = nullptr;            // PLAIN/RESIDUAL/GATEUP output, QKV: q output (H*D)
    const float *res = nullptr;    // RESIDUAL
    // QKV
    void *kcache = nullptr, *vcache = nullptr;  // layer base pointers
    int kv_type = NFAI_F32;
    uint64_t kv_pos_stride = 0;    // elements between consecutive positions
    uint64_t kv_head_stride = 0;   // elements between consecutive kv heads
    const float *rope_cs = nullptr;  // [rope_dims/2][2] cos,sin for the current position (device)
    uint32_t rope_dims = 0, H = 0, Hkv = 0, D = 0;
    const uint32_t *pos_dev = nullptr;  // current position (device scalar)
    uint32_t n_cu = 256;
    // Rows dealt to the XCDs in proportion to how fast each one streams (calibrated once per context: gemv_xcd_calibrate): 8 shares
    // (rows per dealing round for the workgroups with blockIdx % 8 == x), or nullptr = equal shares.  Speed only: every row is
    // computed exactly once by the same arithmetic whatever the shares are.  Honoured by the fp16 / fp32 k_gemv for GEMV_PLAIN and
    // GEMV_GATEUP launches whose grid is a multiple of 8 (the lm_head and the gate | up launch).
    const uint16_t *xcd_shares = nullptr;
    bool prefetch_only = false;  // side-stream launch that only touches the first two steps of every wave's weights
    // GEMV_QKV_ROPE with RMSNorm only — the per-token prologue folded into the FIRST q|k|v launch of a token (LlamaModel.cs:116 +
    // the RoPE angles RoPEShader.cs:254-256 recomputes per element): begin.on: the activation vector is row tok[0] of the
    // embedding table `emb` (emb == nullptr: x as usual, e.g. a pipeline stage that received its hidden state), workgroup 0 also
    // stores it to x_out (later launches read it as the residual), writes the cos/sin table of the current position for the
    // other blocks' launches and advances the hand-off epoch; this launch's own RoPE forms cos/sin itself
    struct Begin {
        bool on = false;
        const void *emb = nullptr;   // [emb_rows][K]: F16 / F32 row-major, or the T16 layout of a Q4_K / Q6_K table
        int emb_type = NFAI_F16;
        uint64_t emb_rows = 0;
        const uint32_t *tok = nullptr;
        float *x_out = nullptr;
        const float *freqs = nullptr;  // [n_freq] RoPE frequencies (TransformerBlock.cs:33-38)
        float *cs_out = nullptr;       // [n_freq][2]
        uint32_t n_freq = 0;
        uint32_t *epoch = nullptr;
    } begin;
    // GEMV_PLAIN only: SamplingUtils.ArgMax over the N outputs in the same launch (the lm_head): argmax_part = workspace of
    // argmax_fused_bytes() bytes, zeroed once; the first index of the maximum -> argmax_out; then, if argmax_pos_inc != nullptr, the
    // end-of-token bookkeeping of launch_argmax (ring[pos % ring_len] = token, ++pos)
    void *argmax_part = nullptr;
    uint32_t *argmax_out = nullptr, *argmax_pos_inc = nullptr, *argmax_ring = nullptr;
    uint32_t argmax_ring_len = 0;
};
constexpr size_t argmax_fused_bytes() { return 1024 * 8 + 256; }  // [1024] values | [1024] indices | ticket
enum GemvMode { GEMV_PLAIN = 0, GEMV_RESIDUAL = 1, GEMV_QKV_ROPE = 2, GEMV_GATEUP = 3 };

// Internal weight-type codes of the T16 layouts (kernels_gemv_kqm.hip): same bytes as the ggml type, rows
// grouped in tiles of 16.  Never seen across the C ABI: uploads with rows % 16 == 0 are repacked into them.
constexpr int NFAI_Q4_K_T16 = 112, NFAI_Q6_K_T16 = 114;
constexpr int NFAI_KQ_MIXED = 115;  // GemvArgs::w_type of a q|k|v launch whose segments are Q4_K_T16 / Q6_K_T16 per seg6_mask
inline bool is_kquant(int t) { return t == NFAI_Q4_K || t == NFAI_Q6_K || t == NFAI_Q4_K_T16 || t == NFAI_Q6_K_T16; }
inline int ggml_type_of(int t) { return t == NFAI_Q4_K_T16 ? NFAI_Q4_K : (t == NFAI_Q6_K_T16 ? NFAI_Q6_K : t); }

hipError_t launch_gemv(const GemvArgs &a, hipStream_t s);     // any weight type; K-quants go to launch_gemv_kq / _kqm
bool gemv_begin_ok(const GemvArgs &a);                         // can this q|k|v launch carry the per-token prologue (GemvArgs::Begin)?
// Measures how fast the workgroups of each blockIdx % 8 label (= one XCD each) stream from HBM with every CU streaming, and derives
// the dealing shares of GemvArgs::xcd_shares; nullptr when switched off (NFAI_XCD_DEAL=0) or when the probe could not run.
const uint16_t *gemv_xcd_calibrate(Ctx *c);
hipError_t launch_gemv_kq(const GemvArgs &a, hipStream_t s);  // Q4_K (native blocks) / Q6_K (plane layout)
hipError_t launch_gemv_kqm(const GemvArgs &a, hipStream_t s); // Q4_K_T16: MFMA dot products
hipError_t launch_repack_q4k_t16(const void *native, void *tiled, uint64_t rows, uint64_t cols, hipStream_t s);
hipError_t launch_repack_q6k_t16(const void *native, void *tiled, uint64_t rows, uint64_t cols, hipStream_t s);
hipError_t launch_embed_kqt(const void *table, int type, uint64_t n_rows, const uint32_t *tok, float *y, uint32_t E, hipStream_t s);
hipError_t launch_embed_rows_kqt(const void *table, int type, uint64_t n_rows, const uint32_t *toks, float *y, uint32_t T, uint32_t E, hipStream_t s);
hipError_t launch_dequant_t16_f16(const void *W, int type, uint64_t rows, uint64_t cols, void *out_f16, hipStream_t s);  // T16 K-quant -> fp16 [rows][cols]
// Q6_K super-blocks are 210 bytes (not 16-byte aligned): in HBM they live as four planes
// ql | qh | scales | d (same bytes, naturally aligned accesses).  nblk = rows * cols / 256.
hipError_t launch_repack_q6k(const void *native, void *planes, uint64_t nblk, hipStream_t s);
hipError_t launch_embed_kq(const void *table, int type, uint64_t n_rows, const uint32_t *tok, float *y, uint32_t E,
                           hipStream_t s);

// ---- MFMA prefill (kernels_prefill.hip) ---------------------------------------------------------
// epi 3 (the q | k | v projection of a prompt chunk): RoPE and every store of RoPEShader / the KV-cache writes in the GEMM's epilogue —
// row t of the product is position pos0 + t; columns are q heads | k heads | v heads.  cs = cos / sin of the chunk's positions
// (launch_rope_table).  No fp32 q | k | v round trip through HBM and no separate launch.
struct GemmRope {
    const float *cs = nullptr;            // [T][D/2][2]
    void *qh = nullptr, *kh = nullptr, *vt = nullptr;   // fp16: q [T][H*D], K rows [Hkv][Spad][D], V^T [Hkv][D][Spad]
    void *kc = nullptr, *vc = nullptr;    // the KV cache of the block (fp32 or fp16)
    uint64_t pos_stride = 0, head_stride = 0;
    uint32_t H = 0, Hkv = 0, D = 0, rope_dims = 0, pos0 = 0, Spad = 0, kv_f16 = 0;
};
hipError_t launch_rope_table(const float *freqs, uint32_t pos0, uint32_t T, uint32_t D, uint32_t rope_dims, float *cs, hipStream_t s);

struct GemmArgs {  // C[M][N] (+R) = alpha * A[M][K] * B[N][K]^T, fp16 operands, fp32 accumulate / output
    const void *A = nullptr;   // fp16, or fp32 when a_f32 (converted while staging)
    bool a_f32 = false;
    const void *B = nullptr;   // fp16 [N][ldb]
    const void *B1 = nullptr, *B2 = nullptr;  // optional 2nd / 3rd row segment (rows n0.., n0+n1..); all segments ldb wide
    uint32_t n0 = 0, n1 = 0;
    void *C = nullptr;         // fp32 [M][ldc]; fp16 when epi != 0
    int epi = 0;               // 0: fp32 (+R)   1: fp16 out   2: fp16 out = up * silu(gate), B = gate rows, B1 = up rows, N = 2F, C is [M][F]
                               // 3: RoPE + q / KV-cache stores (rope; C unused)
    GemmRope rope;
    uint32_t n_cu = 256;
    const float *R = nullptr;  // optional residual, same layout as C (epi 0)
    uint32_t M = 0, N = 0, K = 0, lda = 0, ldb = 0, ldc = 0;
    uint32_t batch = 1, b_div = 1;
    uint64_t a_bs = 0, b_bs = 0, c_bs = 0;
    float alpha = 1.0f;
    int b_type = NFAI_F16;     // launch_gemm_kq: NFAI_Q4_K_T16 / NFAI_Q6_K_T16 (all weight segments of one type)
    int variant = 0;           // 0: chosen from the shape.  1: 128x64 tiles (4x1 waves)  2: 128x128 (2x2 waves)  3 / 4: direct-to-LDS 128x128, 2 / 3 stages  5 / 6 / 7: direct-to-LDS 128x64, 2 / 3 / 4 stages
    uint32_t causal = 0, causal_pos0 = 0;  // 1: C is a causal score matrix, 2: A is a causal probability matrix (skip masked tiles)
};
hipError_t launch_gemm_f16(const GemmArgs &a, hipStream_t s);
hipError_t launch_gemm_kq(const GemmArgs &a, hipStream_t s);   // dequant-in-LDS GEMM on T16 K-quant weights
hipError_t launch_f32_to_f16(const float *x, void *y_f16, uint64_t n, hipStream_t s);
hipError_t launch_sum_slabs(const float *slabs, uint32_t ks, uint64_t n, const float *R, float *C, hipStream_t s);  // C = R + sum of ks slabs of n floats
hipError_t launch_read_ahead(const void *w, uint64_t bytes, uint32_t n_cu, hipStream_t s);  // side-stream hint: the next GEMM's weights -> Infinity Cache
hipError_t launch_rmsnorm_rows(const float *x, const float *g, void *y_f16, uint32_t T, uint32_t E, float eps, hipStream_t s);
// x_out = R + slab 0 + ... + slab ks-1 (slabs of T * E floats; the order of launch_sum_slabs), y = RMSNorm(x_out) * g as fp16: combine + norm in one pass
hipError_t launch_rmsnorm_rows_combine(const float *slabs, uint32_t ks, const float *R, float *x_out, const float *g, void *y_f16, uint32_t T, uint32_t E, float eps,
                                       hipStream_t s);
hipError_t launch_rope_store_rows(const float *q, const float *k, const float *v, void *qh_f16, void *kc, void *vc, int kv_f16,
                                  uint64_t pos_stride, uint64_t head_stride, const float *freqs, uint32_t rope_dims, uint32_t H,
                                  uint32_t Hkv, uint32_t D, uint32_t pos0, uint32_t T, uint32_t ld, void *kh_f16, void *vt_f16,
                                  uint32_t Spad, hipStream_t s);
hipError_t launch_kv_to_f16(const void *kc, const void *vc, int kv_f16, uint64_t pos_stride, uint64_t head_stride, void *kh, void *vt,
                            uint32_t Hkv, uint32_t D, uint32_t S, uint32_t Spad, uint32_t skip_lo, uint32_t skip_hi, hipStream_t s);
// causal attention of a prompt chunk in one launch: qh [T][H*D], kh [Hkv][Spad][D], vt [Hkv][D][Spad] (fp16) -> out [T][H*D] fp16
hipError_t launch_attn_prefill(const void *qh, const void *kh, const void *vt, void *out_f16, uint32_t T, uint32_t H, uint32_t Hkv,
                               uint32_t D, uint32_t Spad, uint32_t pos0, hipStream_t s);
hipError_t launch_softmax_causal_rows(const float *sc, void *p_f16, uint32_t H, uint32_t T, uint32_t Spad, uint32_t pos0, float scale,
                                      hipStream_t s);
hipError_t launch_silu_mul_rows(const float *gate, const float *up, void *act_f16, uint32_t T, uint32_t F, uint32_t ld, hipStream_t s);
hipError_t launch_embed_rows(const void *table, int type, const uint32_t *toks, float *x, uint32_t T, uint32_t E, hipStream_t s);

struct AttnArgs {
    const float *q = nullptr;  // [H][D] fp32 (after RoPE)
    const void *kcache = nullptr, *vcache = nullptr;
    int kv_type = NFAI_F32;
    uint64_t kv_pos_stride = 0, kv_head_stride = 0;
    float *o = nullptr;        // [H][D]
    uint32_t H = 0, Hkv = 0, D = 0, C = 0;
    const uint32_t *pos_dev = nullptr;  // S = *pos_dev + 1
    float *partials = nullptr;          // workspace: attn_partials_bytes()
    uint32_t n_cu = 256;
    // Slice hand-off.  epoch == nullptr: ticket form (partials written through, one ticket per block, the last block of a kv head
    // merges).  epoch != nullptr: granule form — every partial word travels as an 8-byte {value, tag} granule and the block of
    // the LAST slice polls the granules themselves; tag = epoch[0] * tag_mul + tag_add must differ from the tag of every earlier
    // launch that used this workspace (the model: per-token epoch x blocks + block index).  A poll that gives up sets *err.
    const uint32_t *epoch = nullptr;
    uint32_t tag_mul = 1, tag_add = 0;
    uint32_t *err = nullptr;
    uint32_t debug_withhold = 0;  // test hook: slice (value - 1) of kv head 0 publishes nothing, so the others' bounded waits give up
};
constexpr uint32_t ATTN_NSPLIT_MAX = 32;
size_t attn_partials_bytes(uint32_t H, uint32_t Hkv, uint32_t D, bool granules = false);  // granules: the {value, tag} form (AttnArgs::epoch set)
hipError_t launch_attn_decode(const AttnArgs &a, hipStream_t s);
size_t attn_wo_extra_bytes(uint32_t H, uint32_t D);  // workspace behind attn_partials_bytes(…, true) for the fused launch below
// attention + (Wo + residual) in ONE launch (kernels_attn.hip, k_attn_wo): `g` is the Wo launch that would follow `a` (x = a.o).
// attn_wo_ok: the shapes / modes it is built for; everything else takes the two launches.
bool attn_wo_ok(const AttnArgs &a, const GemvArgs &g);
hipError_t launch_attn_wo(const AttnArgs &a, const GemvArgs &g, hipStream_t s);

// the weight-streaming engine (kernels_engine.hip): Wo + residual -> gate|up -> Wdown + residual -> next block's q|k|v, one launch
struct EngineArgs {
    uint32_t n_ops = 3;              // 3: ends with Wdown (last block of the range); 4: + the next block's q|k|v
    uint32_t E = 0, F = 0, HD = 0;
    const void *Wo = nullptr, *Wgate = nullptr, *Wup = nullptr, *Wdown = nullptr;   // fp16 [N][K]
    const void *Wqkv[3] = {nullptr, nullptr, nullptr};
    uint32_t qkv_rows[3] = {0, 0, 0};
    const float *att = nullptr, *x_in = nullptr, *gamma_ffn = nullptr, *gamma_next = nullptr;
    float eps = 0.f;
    uint64_t *g_h = nullptr, *g_act = nullptr, *g_x = nullptr;   // granule vectors (E, F, E), zero-initialised once
    const uint32_t *epoch = nullptr;                             // device word, advanced once per token
    float *x_out = nullptr;          // plain [E]: the block's output
    float *q_out = nullptr;
    void *kcache = nullptr, *vcache = nullptr;                   // the NEXT block's caches
    int kv_type = NFAI_F32;
    uint64_t kv_pos_stride = 0, kv_head_stride = 0;
    const float *rope_cs = nullptr;
    uint32_t rope_dims = 0, D = 0;
    const uint32_t *pos_dev = nullptr;
    uint32_t *err = nullptr;         // device word: non-zero after a bounded wait gave up
    uint32_t n_cu = 256;
};
struct EnginePlan {  // a planned engine launch: the op table in device memory, the kernel argument, the launch geometry
    void *params_dev = nullptr;
    alignas(8) unsigned char params[256] = {};
    uint32_t lds_bytes = 0, n_cu = 0;
    int ahead = 0;  // U: chunks per step (4, 6 or 8 = embedding width / 512)
};
size_t engine_params_bytes();
hipError_t engine_plan(const EngineArgs &a, void *params_dev, EnginePlan &plan);  // synchronous copy: outside stream capture
hipError_t launch_engine(const EnginePlan &plan, hipStream_t s);

// basic 1:1 ops
hipError_t launch_embed(const void *table, int type, const uint32_t *tok, float *y, uint32_t E, hipStream_t s);
hipError_t launch_rmsnorm(const float *x, const float *g, float *y, uint32_t E, float eps, hipStream_t s);
hipError_t launch_rope(const float *in, float *out, const float *freqs, uint32_t rope_dims, uint32_t n_heads,
                       uint32_t head_dim, uint32_t pos, hipStream_t s);
hipError_t launch_attn_scores(const float *q, const float *K, float *sc, uint32_t H, uint32_t Hkv, uint32_t D,
                              uint32_t S, hipStream_t s);
hipError_t launch_attn_softmax(const float *sc, float *w, uint32_t H, uint32_t S, float eps, hipStream_t s);
hipError_t launch_attn_wsum(const float *w, const float *V, float *o, uint32_t H, uint32_t Hkv, uint32_t D,
                            uint32_t S, hipStream_t s);
hipError_t launch_silu(const float *x, float *y, uint32_t n, hipStream_t s);
hipError_t launch_mul(const float *a, const float *b, float *y, uint32_t n, hipStream_t s);
hipError_t launch_add(const float *a, const float *b, float *y, uint32_t n, hipStream_t s);
// argmax over n floats -> out_idx; `partials` needs 2*ARGMAX_BLOCKS*4 bytes. If pos_inc != null it is
// incremented by the final block (end-of-token bookkeeping for graph replay); ring != null records
// the token at ring[(*pos_inc_old) % ring_len].
constexpr uint32_t ARGMAX_BLOCKS = 128;
hipError_t launch_argmax(const float *x, uint32_t n, uint32_t *out_idx, void *partials, uint32_t *pos_inc,
                         uint32_t *ring, uint32_t ring_len, hipStream_t s);
// top-k candidates of SamplingUtils.TopP on the device (kernels_basic.hip): the k largest of n logits (descending, ties: lower
// index first) + max(l / T) + sum exp(l / T - max) -> the head of `work` (topk_out_offset(): k floats, at +TOPK_MAX*4 k indices,
// at +TOPK_MAX*8 M and S).  work: topk_work_bytes(n) bytes, zeroed once.
constexpr uint32_t TOPK_MAX = 64, TOPK_THREADS = 256, TOPK_NT = 8, TOPK_BLOCKS_MAX = 1024;
inline uint32_t topk_blocks(uint32_t n)
{
    const uint32_t per = TOPK_THREADS * TOPK_NT;  // logits a block can hold in registers
    const uint32_t need = (n + per - 1) / per, spread = (n + TOPK_THREADS - 1) / TOPK_THREADS;
    const uint32_t b = need > (spread < 128 ? spread : 128) ? need : (spread < 128 ? spread : 128);  // a vocabulary is spread over >= 128 blocks
    return b > TOPK_BLOCKS_MAX ? TOPK_BLOCKS_MAX : (b ? b : 1);
}
size_t topk_work_bytes(uint32_t n);
size_t topk_out_offset();
hipError_t launch_topk(const float *x, uint32_t n, float temperature, uint32_t k, void *work, hipStream_t s);
// launch + 8k + 8 bytes back + the host half (api.hip); blocking
void topk_finish(const float *vals, const uint32_t *ids, float M, float S, float temperature, uint32_t k, uint32_t *ids_out, float *probs_out);
int topk_run(Ctx *c, const float *logits_dev, uint32_t n, float temperature, uint32_t k, void *work, uint32_t *ids_out, float *probs_out);
// per-token prologue: embed row -> x, cos/sin table for the current position
hipError_t launch_token_begin(const void *table, int type, const uint32_t *tok, float *x, uint32_t E,
                              const float *freqs, float *rope_cs, uint32_t n_freq, const uint32_t *pos_dev,
                              hipStream_t s, uint32_t *epoch = nullptr);  // epoch: the engine's per-token tag, advanced here
hipError_t launch_pos_advance(uint32_t *pos_dev, hipStream_t s);

}  // namespace nfai

// ---- device helpers (HIP translation units only) ------------------------------------------------
#if defined(__HIPCC__)
namespace nfai {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

// Sum across the 64 lanes of a wave; every lane gets the total.
// DPP within rows of 16 (quad_perm xor1, xor2; row_half_mirror; row_mirror), then the four row
// totals are read with v_readlane (wave-uniform) and added.
__device__ __forceinline__ float wave_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    int iv = __builtin_bit_cast(int, v);
    float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// Max across the 64 lanes, same shape as wave_sum (four DPP steps inside rows of 16, then the four row maxima by
// v_readlane): no ds_bpermute round trips through the LDS crossbar (six of them in the shuffle form).
__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Sum over the four rows of 16 lanes (lane r, r + 16, r + 32, r + 48), every lane gets it: (v[r] + v[r+16]) + (v[r+32] + v[r+48]),
// the order of `v += shfl_xor(v, 16); v += shfl_xor(v, 32)`, but inside the vector ALU (gfx950: v_permlane16_swap / v_permlane32_swap)
// instead of two dependent ds_bpermute round trips through the LDS crossbar with an s_waitcnt each.
__device__ __forceinline__ float rows4_sum(float v)
{
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    const auto r16 = __builtin_amdgcn_permlane16_swap(b, b, false, false);  // [0]: rows {0,0,2,2}, [1]: rows {1,1,3,3}
    const float t = __builtin_bit_cast(float, (uint32_t)r16[0]) + __builtin_bit_cast(float, (uint32_t)r16[1]);
    const uint32_t tb = __builtin_bit_cast(uint32_t, t);
    const auto r32 = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);  // [0]: halves {lo, lo}, [1]: halves {hi, hi}
    return __builtin_bit_cast(float, (uint32_t)r32[0]) + __builtin_bit_cast(float, (uint32_t)r32[1]);
}

// Sum within aligned groups of `width` lanes (power of two <= 64); every lane of the group gets it.
template <int width>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int o = width / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// 16-byte streaming load of read-once data (weights, KV): non-temporal so it does not displace
// the activations in L2 (MI355X guide, row nt-weights).
// The pointer is cast to the GLOBAL address space explicitly: through struct-carried pointers and
// selects hipcc otherwise emits flat_load, which counts on vmcnt AND lgkmcnt, completes out of order
// and forces s_waitcnt vmcnt(0) — i.e. no load pipelining at all.
__device__ __forceinline__ u32x4 load_nt16(const void *p)
{
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) u32x4 *)p);
}

__device__ __forceinline__ float h2f_lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2, w)[0]; }
__device__ __forceinline__ float h2f_hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2, w)[1]; }

// dot of 8 fp16 weights (one 16-byte load) with 8 fp32 activations, fp32 accumulate.
// fmaf((float)h, x, acc) lowers to v_fma_mix_f32 (fp16 operand converted inside the FMA).
__device__ __forceinline__ float dot8_f16(u32x4 w, f32x4 x0, f32x4 x1, float acc)
{
    acc = fmaf(h2f_lo(w[0]), x0[0], acc);
    acc = fmaf(h2f_hi(w[0]), x0[1], acc);
    acc = fmaf(h2f_lo(w[1]), x0[2], acc);
    acc = fmaf(h2f_hi(w[1]), x0[3], acc);
    acc = fmaf(h2f_lo(w[2]), x1[0], acc);
    acc = fmaf(h2f_hi(w[2]), x1[1], acc);
    acc = fmaf(h2f_lo(w[3]), x1[2], acc);
    acc = fmaf(h2f_hi(w[3]), x1[3], acc);
    return acc;
}

// ---- the per-token prologue inside the first q|k|v launch (GemvArgs::Begin) -------------------------------------------------------
struct BeginParams {
    const uint8_t *emb;
    uint64_t emb_rows;
    const uint32_t *tok;
    float *x_out;
    const float *freqs;
    float *cs_out;
    uint32_t *epoch;
    uint32_t n_freq;
    int emb_type;
    uint32_t on;
};

// four consecutive elements k .. k+3 (k % 4 == 0) of row `row` of an embedding table with E columns, widened to fp32 exactly as
// TokenEmbedShader (TokenEmbedShader.cs:131-159) / k_embed_q4t / k_embed_q6t give them
__device__ __forceinline__ f32x4 embed_load4(const uint8_t *table, int type, uint64_t n_rows, uint64_t row, uint32_t k, uint32_t E)
{
    typedef __attribute__((address_space(1))) uint8_t g8;
    const g8 *t = (const g8 *)table;
    if (type == NFAI_F16) {
        const u32x2 w = *reinterpret_cast<const __attribute__((address_space(1))) u32x2 *>(t + (row * E + k) * 2);
        return f32x4{h2f_lo(w[0]), h2f_hi(w[0]), h2f_lo(w[1]), h2f_hi(w[1])};
    }
    if (type == NFAI_F32) return *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>(t + (row * E + k) * 4);
    const uint32_t NB = E / 256, blk = k >> 8, kk = k & 255;
    const uint64_t tile = row >> 4, r = row & 15, tb = tile * NB + blk, nblk = n_rows * NB;
    f32x4 out;
    if (type == NFAI_Q4_K_T16) {  // k_embed_q4t
        const uint32_t sb = kk >> 5, l = kk & 31;
        const g8 *hdr = t + nblk * 128 + tb * 256 + r * 16;
        const float d = (float)*reinterpret_cast<const __attribute__((address_space(1))) _Float16 *>(hdr);
        const float dmin = (float)*reinterpret_cast<const __attribute__((address_space(1))) _Float16 *>(hdr + 2);
        const g8 *scales = hdr + 4;
        uint32_t sc, m;
        if (sb < 4) { sc = scales[sb] & 63; m = scales[sb + 4] & 63; }
        else { sc = (scales[sb + 4] & 0xF) | ((scales[sb - 4] >> 6) << 4); m = (scales[sb + 4] >> 4) | ((scales[sb] >> 6) << 4); }
        const uint32_t q4 = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(t + tb * 2048 + (l >> 4) * 1024 + ((sb >> 1) * 16 + r) * 16 + (l & 15));
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t q = (q4 >> (8 * e)) & 0xFFu;
            out[e] = d * (float)sc * (float)((sb & 1) ? (q >> 4) : (q & 0xF)) - dmin * (float)m;
        }
    } else {  // NFAI_Q6_K_T16: k_embed_q6t
        const uint32_t n = kk >> 7, qd = (kk >> 5) & 3, l = kk & 31, lh = l >> 4, b = l & 15;
        const uint32_t ln = (n * 2 + lh) * 16 + (uint32_t)r;
        const uint32_t ql4 = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(t + tb * 3072 + (qd & 1) * 1024 + ln * 16 + b);
        const uint32_t qh4 = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(t + tb * 3072 + 2048 + ln * 16 + b);
        const int sc = (int)(int8_t)t[nblk * 192 + tb * 256 + r * 16 + 8 * n + lh + 2 * qd];
        const float d = (float)reinterpret_cast<const __attribute__((address_space(1))) _Float16 *>(t + nblk * 208 + tb * 32)[r];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t ql = (ql4 >> (8 * e)) & 0xFFu, qh = (qh4 >> (8 * e)) & 0xFFu;
            const int q = (int)(((qd >= 2) ? (ql >> 4) : (ql & 0xF)) | (((qh >> (2 * qd)) & 3) << 4)) - 32;
            out[e] = d * (float)sc * (float)q;
        }
    }
    return out;
}

// cos / sin of RoPE pair `pair` at position pos, as k_token_begin tabulates them (RoPEShader.cs:254-256)
__device__ __forceinline__ f32x2 rope_cs_of(const float *freqs, uint32_t pair, uint32_t pos)
{
    const float theta = ((const __attribute__((address_space(1))) float *)freqs)[pair] * (float)pos;
    return f32x2{cosf(theta), sinf(theta)};
}

// First q|k|v launch of a token: every workgroup tabulates cos/sin of the position once, in LDS (cs_lds: 2 * n_freq floats, read by
// its own RoPE epilogues behind a workgroup barrier; libm range reduction at theta ~ 1e2 is too slow to repeat per finished row);
// workgroup 0 also writes the table for the other blocks' launches and advances the hand-off epoch.
__device__ __forceinline__ void begin_bookkeeping(const BeginParams &b, uint32_t pos, float *cs_lds)
{
    for (uint32_t d = threadIdx.x; d < b.n_freq; d += blockDim.x) {
        const f32x2 cs = rope_cs_of(b.freqs, d, pos);
        cs_lds[2 * d] = cs[0];
        cs_lds[2 * d + 1] = cs[1];
        if (blockIdx.x == 0) {
            b.cs_out[2 * d] = cs[0];
            b.cs_out[2 * d + 1] = cs[1];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && b.epoch) b.epoch[0] = b.epoch[0] + 1;
}
constexpr uint32_t BEGIN_CS_WORDS = 128;  // LDS words of that table (head_dim <= 128)

// ---- "first index of the largest value" across lanes / waves / workgroups (SamplingUtils.ArgMax, SamplingUtils.cs:55-56: values.Max()
//      then IndexOf: the LOWEST index among equal maxima) ---------------------------------------------------------------------------
__device__ __forceinline__ bool topk_better(float av, uint32_t ai, float bv, uint32_t bi) { return av > bv || (av == bv && ai < bi); }

__device__ __forceinline__ void wave_best(float &v, uint32_t &i)
{
#define NFAI_BEST_STEP(CTRL)                                                                                                      \
    {                                                                                                                             \
        const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)); \
        const uint32_t oi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)i, CTRL, 0xF, 0xF, true);                                 \
        if (topk_better(ov, oi, v, i)) { v = ov; i = oi; }                                                                        \
    }
    NFAI_BEST_STEP(0xB1)   // quad_perm [1,0,3,2]
    NFAI_BEST_STEP(0x4E)   // quad_perm [2,3,0,1]
    NFAI_BEST_STEP(0x141)  // row_half_mirror
    NFAI_BEST_STEP(0x140)  // row_mirror
#undef NFAI_BEST_STEP
    // every lane of a row of 16 now holds the row's best; the four rows meet through scalar registers
    float bv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    uint32_t bi = (uint32_t)__builtin_amdgcn_readlane((int)i, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r));
        const uint32_t oi = (uint32_t)__builtin_amdgcn_readlane((int)i, r);
        if (topk_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    v = bv;
    i = bi;
}


// ArgMax folded into the launch that produces the values (the lm_head GEMV): every thread arrives with the best (value, index) it
// has finished itself ((-inf, 0xFFFFFFFF) if none); the workgroup's best goes to part[blockIdx.x] (write-through), ONE lane takes a
// ticket, and the workgroup whose ticket is last combines the partials — (value desc, index asc) is a total order, so the result does
// not depend on which workgroup that is — and does the end-of-token bookkeeping of k_argmax: token fed back, ring, position.
// lds: 34 words no other code of the kernel touches.  Called by ALL threads of the workgroup, once, at the end of the kernel.
struct ArgmaxFused {
    float *part_v;        // [gridDim.x]
    uint32_t *part_i;     // [gridDim.x]
    uint32_t *ticket;     // zero between launches
    uint32_t *out_idx, *pos_inc, *ring;
    uint32_t ring_len;
};
constexpr uint32_t ARGMAX_FUSED_MAX_BLOCKS = 1024;
__device__ __forceinline__ void argmax_fused_tail(float bv, uint32_t bi, const ArgmaxFused &t, uint32_t *lds)
{
    float *sv = reinterpret_cast<float *>(lds);
    uint32_t *si = lds + 16, *last = lds + 32;
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    wave_best(bv, bi);
    if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < nw; w++)
            if (topk_better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
        __hip_atomic_store(&t.part_v[blockIdx.x], bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&t.part_i[blockIdx.x], bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t tk = __hip_atomic_fetch_add(t.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last[0] = (tk == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (last[0] == 0u) return;
    bv = -INFINITY;
    bi = 0xFFFFFFFFu;
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += blockDim.x) {
        const float v = __hip_atomic_load(&t.part_v[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t i = __hip_atomic_load(&t.part_i[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (topk_better(v, i, bv, bi)) { bv = v; bi = i; }
    }
    wave_best(bv, bi);
    __syncthreads();
    if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < nw; w++)
            if (topk_better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
        t.out_idx[0] = bi;
        __hip_atomic_store(t.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm (stream-ordered with the next launch)
        if (t.pos_inc) {
            const uint32_t pp = t.pos_inc[0];
            if (t.ring) t.ring[pp % t.ring_len] = bi;
            t.pos_inc[0] = pp + 1;
        }
    }
}

#ifdef NFAI_STAMPS
// STAMP(i): reading i of this wave (wave-uniform scalar registers; written out by stamp_flush at the end of the kernel)
struct Stamps {
    unsigned long long t[STAMP_WORDS] = {};
    __device__ __forceinline__ void at(int i) { __builtin_amdgcn_sched_barrier(0); t[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ void flush(unsigned long long *base, uint32_t wave_global, int n)
    {
        if (!base || wave_global >= STAMP_WAVES) return;
        if ((threadIdx.x & 63) == 0)
            for (int i = 0; i < n; i++) base[(size_t)wave_global * STAMP_WORDS + i] = t[i];
    }
};
#define STAMP_DECL nfai::Stamps _st
#define STAMP(i) _st.at(i)
#define STAMP_FLUSH(base, wg, n) _st.flush(base, wg, n)
#else
#define STAMP_DECL
#define STAMP(i) ((void)0)
#define STAMP_FLUSH(base, wg, n) ((void)0)
#endif

__device__ __forceinline__ float silu_ref(float x)
{
    // SiLUShader.cs:121-123: x * (1.0 / (1.0 + exp(-x)))
    float sig = 1.0f / (1.0f + expf(-x));
    return x * sig;
}

// Block-wide sum (<= 16 waves); every thread gets the total.  `red` is >= 16 floats of LDS.
// FRESH: `red` has not been read or written in this launch yet, so nothing has to be kept apart from the write below (the GEMV
// prologues: one barrier fewer on the way to the first FMA).
template <bool FRESH = false>
__device__ __forceinline__ float block_sum(float v, float *red)
{
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if constexpr (!FRESH) __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    // fixed trip count (<= 16 waves): the LDS reads issue back to back; with the runtime bound hipcc emits a serial
    // read-wait-add chain (measured on the K-quant GEMV: +3 % tokens/s from this change alone).  Same summation order.
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float v = red[min(i, nw - 1)];
        t += i < nw ? v : 0.f;
    }
    return t;
}

}  // namespace nfai
#endif
