/*
 * nfai_hip.h — C ABI of libnfai_hip.so: the MI355X (gfx950) backend for NFAI's Llama-3
 * TransformerBlock decode path.
 *
 * The reference (NicuTheodorAlexandru/NFAI) has no FFI of its own: its backend boundary is the
 * C# class VulkanBufferManager plus the ShaderWrapper-derived op classes that call it.  A C#
 * `NFAI.HIP` assembly binds the functions below with [LibraryImport("nfai_hip")] (stubs in
 * INTEGRATION.md / csharp/NFAI.HIP/); each entry cites the reference member it replaces.
 * All paths are relative to the reference root.
 *
 * Conventions
 *  - every function returns an int32 status (NFAI_OK == 0); nfai_hip_last_error() returns a
 *    thread-local UTF-8 message for the last non-zero status (the C# wrapper throws on non-zero,
 *    matching the reference's exceptions on any non-Success VkResult,
 *    NFAI.Vulkan/VulkanBufferManager.cs:61-87);
 *  - handles are opaque 64-bit values; the caller owns host memory, the library owns device
 *    memory (except buffers adopted with nfai_hip_buf_wrap);
 *  - one context per device, one HIP stream per context; a context is NOT thread-safe (the
 *    reference is single-threaded and blocking, VulkanBufferManager.cs:474-494); different
 *    contexts may be used from different threads;
 *  - op calls ENQUEUE on the context stream and return; upload/download/synchronize block.
 *    (The reference fence-waits every dispatch, ShaderWrapper.cs:208-245; results are identical
 *    because everything is stream-ordered.)
 *  - element counts are in ELEMENTS, offsets (`*_off`) in elements of the buffer's type unless a
 *    parameter says bytes;
 *  - no torch / C++ types cross this boundary.
 */
#ifndef NFAI_HIP_H
#define NFAI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFAI_HIP_ABI_VERSION 1

typedef uint64_t nfai_ctx_t;
typedef uint64_t nfai_buf_t;
typedef uint64_t nfai_model_t;
typedef uint64_t nfai_pp_t;

enum nfai_status {
    NFAI_OK = 0,
    NFAI_ERR_INVALID = 1,     /* bad argument / shape the kernels do not support */
    NFAI_ERR_HIP = 2,         /* a HIP runtime call failed (message has hipGetErrorString) */
    NFAI_ERR_OOM = 3,
    NFAI_ERR_KV_FULL = 4,     /* position == KV capacity; the reference has no check
                                 (MatrixMultiplyShader.cs:248-252) and writes out of bounds */
    NFAI_ERR_UNSUPPORTED = 5, /* e.g. a ggml tensor type without a kernel */
    NFAI_ERR_STATE = 6        /* call order (missing tensor, model not finalised, ...) */
};

/* ggml tensor type ids as stored in GGUF (NFAI.GGUF/Parser.cs:262-293 names the same ids). */
enum nfai_dtype { NFAI_F32 = 0, NFAI_F16 = 1, NFAI_Q4_K = 12, NFAI_Q6_K = 14 };

const char *nfai_hip_last_error(void);
int32_t nfai_hip_abi_version(void);

/* ---- context: replaces VulkanHelper.CreateVulkanInstance/PickPhysicalDevice/CreateLogicalDevice
 *      (NFAI.Vulkan/VulkanHelper.cs:12-242) + VulkanBufferManager ctor/Dispose
 *      (VulkanBufferManager.cs:20-35, 499-509), as used by LlamaModelFactory.cs:15-32. ---- */
typedef struct nfai_device_info {
    char name[128];
    char arch[64];            /* gcnArchName, must start with "gfx950" */
    uint64_t total_mem_bytes;
    uint32_t compute_units;
    uint32_t wavefront_size;
    uint32_t lds_bytes_per_cu;
    uint32_t clock_khz;
} nfai_device_info;

int32_t nfai_hip_ctx_create(int32_t device_ordinal, nfai_ctx_t *out);
/* Same, but enqueue on a stream owned by the caller (e.g. the stream RCCL point-to-point ops of
 * a pipeline stage run on); `hip_stream` is a hipStream_t. */
int32_t nfai_hip_ctx_create_on_stream(int32_t device_ordinal, void *hip_stream, nfai_ctx_t *out);
int32_t nfai_hip_ctx_destroy(nfai_ctx_t ctx);
int32_t nfai_hip_ctx_synchronize(nfai_ctx_t ctx);   /* ≙ vkQueueWaitIdle, VulkanBufferManager.cs:334 */
int32_t nfai_hip_ctx_device_info(nfai_ctx_t ctx, nfai_device_info *info);
/* Run record: how the rows of the long weight-streaming launches (lm_head, gate | up) are dealt to the 8 XCDs on this device.  The
 * first nfai_hip_llama_finalize on a context times a streaming probe (every workgroup reads an equal share of a 768 MB buffer) and
 * gives the workgroups of each blockIdx % 8 label a share of the rows proportional to their measured rate - a launch ends with its
 * slowest XCD.  Speed only: every row is computed once, by the same arithmetic, whatever the shares.  shares8 / probe_us8: 8 values
 * each (may be NULL); zeros before the first finalize or with NFAI_XCD_DEAL=0.  (No reference counterpart: VulkanHelper.cs:149-150
 * picks a device and never looks at its topology.) */
int32_t nfai_hip_ctx_xcd_shares(nfai_ctx_t ctx, uint16_t *shares8, float *probe_us8);
/* hipEvent pair on the context stream (bench.py times the launches with these, not with
 * torch.cuda.Event, which only sees torch's stream). */
int32_t nfai_hip_timer_begin(nfai_ctx_t ctx);
int32_t nfai_hip_timer_end(nfai_ctx_t ctx, float *elapsed_ms);

/* ---- buffers: replaces VulkanBufferManager.CreateBuffer / DestoryBuffer / UploadDeviceConstants /
 *      UploadDataToDeviceLocal / ReadDeviceBufferData / CopyBuffer
 *      (VulkanBufferManager.cs:42-88, 90-103, 196-244, 105-125, 283-303, 305-318) and the
 *      ShaderProperty<T> storage they back (NFAI.Vulkan.Shaders/ShaderProperty.cs:32-92). ---- */
int32_t nfai_hip_buf_alloc(nfai_ctx_t ctx, uint64_t bytes, nfai_buf_t *out);      /* zero-filled */
int32_t nfai_hip_buf_wrap(nfai_ctx_t ctx, void *device_ptr, uint64_t bytes, nfai_buf_t *out); /* non-owning */
int32_t nfai_hip_buf_free(nfai_ctx_t ctx, nfai_buf_t buf);
int32_t nfai_hip_buf_upload(nfai_ctx_t ctx, nfai_buf_t buf, uint64_t byte_off, const void *host, uint64_t bytes);
int32_t nfai_hip_buf_download(nfai_ctx_t ctx, nfai_buf_t buf, uint64_t byte_off, void *host, uint64_t bytes);
int32_t nfai_hip_buf_copy(nfai_ctx_t ctx, nfai_buf_t dst, uint64_t dst_byte_off, nfai_buf_t src,
                          uint64_t src_byte_off, uint64_t bytes);
int32_t nfai_hip_buf_zero(nfai_ctx_t ctx, nfai_buf_t buf);
int32_t nfai_hip_buf_info(nfai_ctx_t ctx, nfai_buf_t buf, void **device_ptr, uint64_t *bytes);
/* Weights stay in their native GGUF encoding in HBM (fp16 is NOT widened to fp32 as
 * AbstractComputeCollection.cs:62-77 does; the kernels convert in-register — same operand
 * values).  Validates (type, rows, cols) and uploads rows*row_bytes(type, cols) bytes. */
int32_t nfai_hip_weight_upload(nfai_ctx_t ctx, int32_t ggml_type, uint64_t n_rows, uint64_t n_cols,
                               const void *host_bytes, nfai_buf_t *out);
int32_t nfai_hip_weight_bytes(int32_t ggml_type, uint64_t n_rows, uint64_t n_cols, uint64_t *bytes);

/* ---- 1:1 operators: one per ShaderWrapper subclass of the reference.  fp32 activations. ---- */
/* TokenEmbedShader.Compute (TokenEmbedShader.cs:108-119, GLSL :131-159): y[0:E] = table[tok][0:E];
 * `tok` is a device buffer holding one uint32. */
int32_t nfai_hip_embed(nfai_ctx_t ctx, nfai_buf_t table, int32_t table_type, nfai_buf_t tok, nfai_buf_t y, uint32_t E);
/* RMSNormShader.Compute (RMSNormShader.cs:111-122, GLSL :124-151). */
int32_t nfai_hip_rmsnorm(nfai_ctx_t ctx, nfai_buf_t x, nfai_buf_t gamma, nfai_buf_t y, uint32_t E, float eps);
/* MatrixMultiplyShader.Compute, M = 1 (MatrixMultiplyShader.cs:230-253, GLSL :255-289):
 * y[y_off + j] = sum_k x[k] * W[j][k]; y_off = currentContextSize * N for the KV-cached variant. */
int32_t nfai_hip_gemv(nfai_ctx_t ctx, nfai_buf_t W, int32_t w_type, nfai_buf_t x, nfai_buf_t y,
                      uint64_t y_off, uint32_t N, uint32_t K);
/* RoPEShader.Compute(position) (RoPEShader.cs:188-212, GLSL :231-272) on ONE vector of
 * n_heads*head_dim floats at in[in_off], written to out[out_off] (may alias: the K variant is in
 * place on cache row `pos`, TransformerBlock.cs:73-74).  freqs: rope_dims/2 floats. */
int32_t nfai_hip_rope(nfai_ctx_t ctx, nfai_buf_t in, uint64_t in_off, nfai_buf_t out, uint64_t out_off,
                      nfai_buf_t freqs, uint32_t rope_dims, uint32_t n_heads, uint32_t head_dim, uint32_t pos);
/* AttentionScoreCalculationShader.ComputeAttention(seqLen) (…ScoreCalculationShader.cs:141-162,
 * GLSL :164-206): s[h*S + t], K cache [C][Hkv*D]; only t < S is written. */
int32_t nfai_hip_attn_scores(nfai_ctx_t ctx, nfai_buf_t q, nfai_buf_t kcache, nfai_buf_t s,
                             uint32_t H, uint32_t Hkv, uint32_t D, uint32_t S);
/* AttentionSoftmaxShader.ComputeSoftmax(seqLen) (AttentionSoftmaxShader.cs:117-132, GLSL :139-178). */
int32_t nfai_hip_attn_softmax(nfai_ctx_t ctx, nfai_buf_t s, nfai_buf_t w, uint32_t H, uint32_t S, float eps);
/* AttentionWeightedValueSumShader.ComputeWeightedSum(seqLen) (…ValueSumShader.cs:151-173, GLSL :175-216). */
int32_t nfai_hip_attn_wsum(nfai_ctx_t ctx, nfai_buf_t w, nfai_buf_t vcache, nfai_buf_t o,
                           uint32_t H, uint32_t Hkv, uint32_t D, uint32_t S);
/* SiLUShader.Compute (SiLUShader.cs:92-104, GLSL :106-128). */
int32_t nfai_hip_silu(nfai_ctx_t ctx, nfai_buf_t x, nfai_buf_t y, uint32_t n);
/* ElementWiseMultiplicationShader.Compute (ElementWiseMultiplicationShader.cs:99-119, GLSL :121-139). */
int32_t nfai_hip_mul(nfai_ctx_t ctx, nfai_buf_t a, nfai_buf_t b, nfai_buf_t y, uint32_t n);
/* the two host-side residual adds of TransformerBlock.Compute (TransformerBlock.cs:151-161, 174-181),
 * done on the device instead of read-back/add/upload. */
int32_t nfai_hip_add(nfai_ctx_t ctx, nfai_buf_t a, nfai_buf_t b, nfai_buf_t y, uint32_t n);
/* SamplingUtils.ArgMax (NFAI.Models.Llama3/SamplingUtils.cs:43-57): index of the first maximum,
 * written as one uint32 to `out_idx`. */
int32_t nfai_hip_argmax(nfai_ctx_t ctx, nfai_buf_t x, uint32_t n, nfai_buf_t out_idx);
/* The device half of SamplingUtils.TopP (NFAI.Models.Llama3/SamplingUtils.cs:5-13): values / temperature (:7), Softmax over all n
 * (:8, :35-41), OrderByDescending(Prob) (stable: equal probabilities in index order, :9-12), Take(k) (:13) -> ids_out[k],
 * probs_out[k] (host arrays; k <= 64, the reference's topK is 40).  One launch over the logits and 8k + 8 bytes back instead of n
 * floats; the nucleus cut and the Random.Shared draw (:14-31) stay with the caller.  Blocking. */
int32_t nfai_hip_topk(nfai_ctx_t ctx, nfai_buf_t x, uint32_t n, float temperature, uint32_t k, uint32_t *ids_out, float *probs_out);

/* ---- fused operators (no reference counterpart: each replaces the chain named) ---- */
/* scores -> softmax -> weighted sum in one KV-cache pass, GQA heads sharing each K/V read.
 * K/V caches [C][Hkv*D] (reference layout), kv_type NFAI_F32 or NFAI_F16. */
int32_t nfai_hip_attn_decode(nfai_ctx_t ctx, nfai_buf_t q, nfai_buf_t kcache, nfai_buf_t vcache, nfai_buf_t o,
                             uint32_t H, uint32_t Hkv, uint32_t D, uint32_t S, uint32_t C, int32_t kv_type);
/* [RMSNorm ->] GEMV [-> + residual]: y = res + W * (norm ? rmsnorm(x, gamma) : x).  gamma / res
 * may be 0 (absent). */
/* Batched form of MatrixMultiplyShader (inputRowCount = M > 1, which the reference never uses: MatrixMultiplyShader.cs:31-47
 * takes the row count, TransformerBlock.cs:47-101 always passes 1): C[M][N] fp32 (+ residual R, may be 0) = A[M][K] * W[N][K]^T
 * with fp16 operands on the matrix cores — the GEMM of the MFMA prefill, exposed for tests and tools.  variant: 0 = chosen
 * from the shape, 1 = 128x64 tiles, 2 = 128x128 tiles, 3 / 4 = direct-to-LDS staging of 128x128 tiles with 2 / 3 stages (N %% 128 == 0),
 * 5 / 6 / 7 = direct-to-LDS staging of 128x64 tiles with 2 / 3 / 4 stages. */
int32_t nfai_hip_gemm_f16(nfai_ctx_t ctx, nfai_buf_t A_f16, nfai_buf_t W_f16, nfai_buf_t R, nfai_buf_t C, uint32_t M, uint32_t N,
                          uint32_t K, int32_t variant);
/* nfai_hip_gemm_f16 with every epilogue and operand form the MFMA prefill launches (tests / tools; same reference member,
 * MatrixMultiplyShader.cs:31-47 with inputRowCount = M): epi 0 fp32 (+R), 1 fp16, 2 fp16 = up * silu(gate) (SiLUShader.cs:121-123 +
 * ElementWiseMultiplicationShader.cs:137 fused; W = gate rows, W1 = up rows, N = 2F, C is [M][N/2]); head-batched operands
 * A [batch][M][K], W [batch / b_div][N][K], C [batch][M][N]; causal 1: C is a score matrix [t][s], tiles with only s > causal_pos0 + t
 * are skipped (unwritten); causal 2: A is a probability matrix [t][s] (K = keys), K tiles past the last unmasked key are skipped. */
int32_t nfai_hip_gemm_f16_ex(nfai_ctx_t ctx, nfai_buf_t A_f16, nfai_buf_t W_f16, nfai_buf_t W1_f16, nfai_buf_t R, nfai_buf_t C,
                             uint32_t M, uint32_t N, uint32_t K, int32_t variant, int32_t epi, uint32_t batch, uint32_t b_div,
                             uint32_t causal, uint32_t causal_pos0);
/* Causal attention of a prompt chunk in ONE launch (the MFMA prefill's attention; the reference runs AttentionScoreCalculationShader.cs:
 * 164-206, AttentionSoftmaxShader.cs:139-178 and AttentionWeightedValueSumShader.cs:175-216 once per token): Q [T][H*D] fp16 (after
 * RoPE), K [Hkv][Spad][D] fp16, V^T [Hkv][D][Spad] fp16 (rows / columns past pos0 + T zero), O [T][H*D] fp16; query t sees keys
 * 0 .. pos0 + t.  D in {64, 128}, Spad %% 32 == 0, pos0 + T <= Spad. */
int32_t nfai_hip_attn_prefill(nfai_ctx_t ctx, nfai_buf_t Q_f16, nfai_buf_t K_f16, nfai_buf_t Vt_f16, nfai_buf_t O_f16, uint32_t T, uint32_t H,
                              uint32_t Hkv, uint32_t D, uint32_t Spad, uint32_t pos0);
/* The same batched product with W in Q4_K / Q6_K blocks (a buffer from nfai_hip_weight_upload with N %% 16 == 0 rows): the
 * dequant-in-LDS GEMM — quantised bytes -> registers -> fp16 tile in LDS -> MFMA; the weights are never widened in HBM. */
int32_t nfai_hip_gemm_kq(nfai_ctx_t ctx, nfai_buf_t A_f16, nfai_buf_t W, int32_t w_type, nfai_buf_t R, nfai_buf_t C, uint32_t M,
                         uint32_t N, uint32_t K);
int32_t nfai_hip_gemv_fused(nfai_ctx_t ctx, nfai_buf_t W, int32_t w_type, nfai_buf_t x, nfai_buf_t gamma,
                            float eps, nfai_buf_t res, nfai_buf_t y, uint32_t N, uint32_t K);
/* output RMSNorm (gamma may be 0) -> lm_head GEMV -> SamplingUtils.ArgMax in ONE launch (LlamaModel.cs:123-125, SamplingUtils.cs:43-57):
 * V logits and the first index of their maximum (one uint32 in out_idx).  F16 / F32 tables, and Q4_K / Q6_K tables with V %% 16 == 0. */
int32_t nfai_hip_lmhead_argmax(nfai_ctx_t ctx, nfai_buf_t W, int32_t w_type, nfai_buf_t x, nfai_buf_t gamma, float eps,
                               nfai_buf_t logits, nfai_buf_t out_idx, uint32_t V, uint32_t E);
/* RMSNorm -> Wgate, Wup GEMVs -> SiLU(gate) * up (TransformerBlock.cs:163-171 in one launch). */
int32_t nfai_hip_gemv_gateup_silu(nfai_ctx_t ctx, nfai_buf_t Wgate, nfai_buf_t Wup, int32_t w_type,
                                  nfai_buf_t x, nfai_buf_t gamma, float eps, nfai_buf_t y,
                                  uint32_t F, uint32_t K);
/* RMSNorm -> Wq, Wk, Wv GEMVs -> RoPE(q), RoPE(k) -> q buffer, K/V cache rows `pos`
 * (TransformerBlock.cs:129-141 in one launch).  Caches [C][Hkv*D]. */
int32_t nfai_hip_gemv_qkv_rope(nfai_ctx_t ctx, nfai_buf_t Wq, nfai_buf_t Wk, nfai_buf_t Wv, int32_t w_type,
                               nfai_buf_t x, nfai_buf_t gamma, float eps, nfai_buf_t freqs,
                               uint32_t rope_dims, nfai_buf_t q, nfai_buf_t kcache, nfai_buf_t vcache,
                               uint32_t H, uint32_t Hkv, uint32_t D, uint32_t pos, int32_t kv_type, uint32_t E);

/* Wo + residual -> RMSNorm + Wgate|Wup + SiLU*up -> Wdown + residual [-> RMSNorm + next block's Wq|Wk|Wv + RoPE + KV write] of one
 * TransformerBlock (TransformerBlock.cs:150-181, then :129-141 of the next block) as ONE launch of the weight-streaming engine
 * (kernels_engine.hip), fp16 weights, E / F / HD multiples of 512, on caller-held buffers: the op-level form of what the model
 * enqueues per block under NFAI_LLAMA_ENGINE.  Wq == 0: the launch ends with Wdown.  K/V caches in the reference layout
 * [C][Hkv*D].  scratch: (2E + F) * 8 + 64 bytes, zeroed once by the caller (hand-off granules {value, tag} of h | act | x, then
 * control words). */
int32_t nfai_hip_engine_block(nfai_ctx_t ctx, nfai_buf_t Wo, nfai_buf_t Wgate, nfai_buf_t Wup, nfai_buf_t Wdown, nfai_buf_t att,
                              nfai_buf_t x_in, nfai_buf_t gamma_ffn, float eps, uint32_t E, uint32_t F, uint32_t HD,
                              nfai_buf_t Wq, nfai_buf_t Wk, nfai_buf_t Wv, nfai_buf_t gamma_next, nfai_buf_t freqs,
                              uint32_t rope_dims, nfai_buf_t q_out, nfai_buf_t kcache, nfai_buf_t vcache, uint32_t H, uint32_t Hkv,
                              uint32_t D, uint32_t pos, int32_t kv_type, nfai_buf_t x_out, nfai_buf_t scratch);

/* ---- model level: replaces LlamaModel (graph LlamaModel.cs:21-68, token loop :99-174) and
 *      TransformerBlock (wiring TransformerBlock.cs:31-125, sequence :127-184). ---- */
enum nfai_llama_flags {
    NFAI_LLAMA_UNFUSED = 1u << 0,   /* run the 16-op chain 1:1 with the reference (parity mode) */
    NFAI_LLAMA_NO_GRAPH = 1u << 1,  /* fused kernels, eager launches (no hipGraph) */
    NFAI_LLAMA_KV_F16 = 1u << 2,    /* fp16 KV cache (default fp32 = the reference's) */
    NFAI_LLAMA_PREFETCH = 1u << 3,  /* side-stream prefetch of the next GEMV's first weight bytes (perf hint only) */
    NFAI_LLAMA_ENGINE = 1u << 4     /* fp16 models: Wo -> gate|up -> Wdown -> next q|k|v of a block as ONE launch of the
                                       weight-streaming engine (kernels_engine.hip) instead of four; same results up to the
                                       summation order.  Ignored (five-launch path) when a tensor is not fp16 or a width is
                                       not a multiple of 512.  The environment variable NFAI_ENGINE=0/1 overrides the flag. */
};

typedef struct nfai_llama_desc {
    uint32_t E, L, H, Hkv, D, F, V;
    uint32_t C;              /* KV capacity = ModelOptions.KVCacheSize (NFAI.Models/ModelOptions.cs:7) */
    float eps;               /* first metadata key containing "epsilon" (LlamaModel.cs:28) */
    float rope_base;         /* reference hard-codes 500000 (TransformerBlock.cs:33) */
    uint32_t rope_dims;      /* llama.rope.dimension_count (LlamaModel.cs:27) */
    uint32_t rope_n_freqs;   /* valid entries of the frequency table: rope_dims/2 = spec-correct;
                                32 reproduces the reference's truncation (TransformerBlock.cs:66) */
    uint32_t layer_begin, layer_end; /* this context's pipeline stage owns blocks [begin, end) */
    uint32_t flags;          /* nfai_llama_flags */
    uint32_t max_batch;      /* prefill chunk capacity in tokens (0 = decode only) */
} nfai_llama_desc;

int32_t nfai_hip_llama_create(nfai_ctx_t ctx, const nfai_llama_desc *desc, nfai_model_t *out);
int32_t nfai_hip_llama_destroy(nfai_model_t model);
/* GGUF tensor by name ("token_embd.weight", "blk.3.attn_q.weight", ..., "output_norm.weight",
 * optional "output.weight"; absent => lm_head tied to token_embd as LlamaModel.cs:64-67).
 * n_rows x n_cols = ggml ne1 x ne0.  Tensors of blocks outside [layer_begin, layer_end) are
 * ignored.  _set_tensor uploads from host; _set_tensor_device adopts bytes already in HBM. */
int32_t nfai_hip_llama_set_tensor(nfai_model_t model, const char *name, int32_t ggml_type,
                                  uint64_t n_rows, uint64_t n_cols, const void *host_bytes);
int32_t nfai_hip_llama_set_tensor_device(nfai_model_t model, const char *name, int32_t ggml_type,
                                         uint64_t n_rows, uint64_t n_cols, void *device_ptr);
int32_t nfai_hip_llama_finalize(nfai_model_t model);
/* Make `model` use the device-resident tensors `donor` already holds (same layer range, no copy, no second K-quant
 * repack): the in-flight sequences (slots) of a pipeline stage are separate models — own KV cache and position — over ONE set
 * of weights.  The reference aliases buffers the same way with ShaderProperty.BindShaderProprty (ShaderProperty.cs:95-108,
 * no reference count): `donor` must outlive `model`.  Call before nfai_hip_llama_finalize(model). */
int32_t nfai_hip_llama_share_tensors(nfai_model_t model, nfai_model_t donor);
/* One token through embed -> blocks -> output_norm -> lm_head (LlamaModel.cs:116-125) at the
 * current position; logits_host (V floats) and argmax may be NULL.  Blocking. */
int32_t nfai_hip_llama_decode_step(nfai_model_t model, uint32_t token, float *logits_host, uint32_t *argmax);
/* One token as _decode_step, then the candidates of the reference's DEFAULT sampler (LlamaModel.cs:128-130,165 call
 * SamplingUtils.TopP on V logits read back to the host): the k most probable tokens under softmax(logits / temperature) and their
 * probabilities, as nfai_hip_topk.  The caller finishes TopP (SamplingUtils.cs:14-31: nucleus 0.95, renormalise, draw).  Blocking;
 * less than 1 KB crosses PCIe per token. */
int32_t nfai_hip_llama_decode_topk(nfai_model_t model, uint32_t token, float temperature, uint32_t k, uint32_t *ids_out, float *probs_out);
/* Greedy loop with the token fed back on the device (ArgMax in place of the stochastic TopP,
 * SamplingUtils.cs:5-33 vs :43-57): n_steps tokens starting from `first_token`; tokens_out[i] is
 * the argmax after step i.  One hipGraph replay per token, no host round trip. */
int32_t nfai_hip_llama_decode_greedy(nfai_model_t model, uint32_t first_token, uint32_t n_steps, uint32_t *tokens_out);
/* Enqueue n_steps greedy steps without waiting (bench timing region); tokens land in the
 * model's device ring and are fetched with _fetch_tokens after a synchronize. */
int32_t nfai_hip_llama_decode_enqueue(nfai_model_t model, uint32_t n_steps);
int32_t nfai_hip_llama_set_token(nfai_model_t model, uint32_t token);
int32_t nfai_hip_llama_fetch_tokens(nfai_model_t model, uint32_t n, uint32_t *tokens_out);
/* Batched prompt ingestion on the MFMA path (the reference feeds the prompt token by token,
 * LlamaModel.cs:103-126): n tokens at positions pos..pos+n-1; logits of the LAST token.
 * K-quant models: the first call widens every block's matrices to fp16 copies for the MFMA GEMMs (2 bytes per weight, kept
 * until a tensor is replaced) when all of them fit a quarter of the device's memory, else one block's scratch is re-widened
 * per block (NFAI_PREFILL_WIDE_ALL=0 / 1 forces either); the decode path always streams the quantised blocks. */
int32_t nfai_hip_llama_prefill(nfai_model_t model, const uint32_t *tokens, uint32_t n, float *logits_last_host);
/* The prompt phase of LlamaModel.RunAsync (LlamaModel.cs:103-126 feeds tokenIds one by one and keeps only the LAST token's logits,
 * :128-130): n tokens at positions pos..pos+n-1 leave their K / V rows in the cache, nothing is sampled and no logits are formed
 * (no output norm, no lm_head).  The caller then runs the last prompt token through _decode_step / _decode_topk, whose output IS
 * sampled.  Same engine as _prefill: the MFMA path in chunks of desc.max_batch tokens when the model was created with max_batch > 0
 * (fp16 operands: the cache rows agree with the token-by-token path to the fp16 tolerance, 2e-2 absolute on unit-scale rows), the
 * M = 1 path token by token otherwise (bit-identical to _decode_step).  n == 0 is a no-op.  NFAI_ERR_KV_FULL when pos + n exceeds
 * the KV capacity, before anything runs.  Blocking. */
int32_t nfai_hip_llama_ingest(nfai_model_t model, const uint32_t *tokens, uint32_t n);
/* Pipeline stage: run this stage's blocks on a hidden state resident in device memory.
 * First stage: hidden_in == NULL and `token` is embedded.  Last stage: lm_head + argmax run and
 * `logits_host`/`argmax` are filled (blocking) when non-NULL.  Otherwise enqueue only. */
#define NFAI_TOKEN_ON_DEVICE 0xFFFFFFFFu /* stage_step: use the token word already in device memory */
int32_t nfai_hip_llama_stage_step(nfai_model_t model, uint32_t token, const void *hidden_in_dev,
                                  void *hidden_out_dev, float *logits_host, uint32_t *argmax);
/* 4-byte device-to-device copies of the model's token word (the last stage's argmax / the first
 * stage's next input), stream-ordered: lets RCCL carry the token between pipeline ends with no
 * host round trip (the reference reads V logits back and samples on the host, LlamaModel.cs:128-130). */
int32_t nfai_hip_llama_token_to_device(nfai_model_t model, void *dst_dev);
int32_t nfai_hip_llama_token_from_device(nfai_model_t model, const void *src_dev);
int32_t nfai_hip_llama_reset(nfai_model_t model);   /* ≙ MatrixMultiplyShader.ResetCache (:153-159) + currentToken = 0 */
int32_t nfai_hip_llama_set_pos(nfai_model_t model, uint32_t pos);
int32_t nfai_hip_llama_pos(nfai_model_t model, uint32_t *pos);
/* Debug / parity: copy an internal activation to host.  which: 0 hidden (E), 1 q after RoPE (H*D),
 * 2 attention output (H*D), 3 ffn activation (F), 4 logits (V). */
int32_t nfai_hip_llama_read(nfai_model_t model, int32_t which, float *host, uint64_t n);
int32_t nfai_hip_llama_read_kv(nfai_model_t model, uint32_t layer, int32_t is_v, uint32_t pos, float *host /* Hkv*D */);
/* Algorithmic HBM bytes one decode step at position `pos` reads+writes (SURVEY.md §8d formula),
 * and the bytes of the dominant (largest) GEMV launch. */
int32_t nfai_hip_llama_bytes_per_token(nfai_model_t model, uint32_t pos, uint64_t *total, uint64_t *dominant_kernel);
/* Per-kernel-class device time (hipEvents around every launch of one eager step; slow path, for
 * bench.py's roofline object).  ids: 0 qkv, 1 attn, 2 wo, 3 gateup, 4 down, 5 lmhead, 6 other. */
int32_t nfai_hip_llama_profile_step(nfai_model_t model, uint32_t token, float *ms_by_class /* 8 */, uint32_t *launches_by_class /* 8 */);
/* Average duration (us) of ONE kernel class: all its launches of a decode step (one per block, each on its own weights)
 * are replayed back to back, `reps` rounds, between a single pair of hipEvents on the launch stream (decode launches are
 * idempotent).  This is the per-launch duration rocprofv3 --kernel-trace reports; bench.py's roofline uses it. */
int32_t nfai_hip_llama_profile_kernel(nfai_model_t model, uint32_t token, int32_t kernel_class, uint32_t reps, float *us_avg);

/* ---- layer pipeline across GPUs (no reference counterpart: the reference is single-device, it takes the last enumerated
 *      Vulkan device, VulkanHelper.cs:149-150).  One process per GPU owns a contiguous range of TransformerBlocks (a slice of
 *      the block loop LlamaModel.cs:118-121, nfai_llama_desc.layer_begin/end); between stages the hidden state (n_embd fp32)
 *      moves point to point over RCCL/xGMI and the sampled token returns from the last stage to the first.  All operations
 *      are enqueued on the context's stream, in order with nfai_hip_llama_stage_step; RCCL is bound with dlopen on first use
 *      (NFAI_ERR_UNSUPPORTED when it is absent).  Operations that must progress together (one tick of the schedule) are
 *      posted between _begin and _end (ncclGroupStart / ncclGroupEnd). ---- */
int32_t nfai_hip_pp_unique_id(uint8_t *out128);   /* rank 0 creates it; the host hands the 128 bytes to every rank */
int32_t nfai_hip_pp_init(nfai_ctx_t ctx, uint32_t rank, uint32_t world, const uint8_t *unique_id128, nfai_pp_t *out);
int32_t nfai_hip_pp_destroy(nfai_pp_t pp);
int32_t nfai_hip_pp_begin(nfai_pp_t pp);
int32_t nfai_hip_pp_end(nfai_pp_t pp);
int32_t nfai_hip_pp_send_hidden(nfai_pp_t pp, const void *hidden_dev, uint32_t n_floats, uint32_t peer);
int32_t nfai_hip_pp_recv_hidden(nfai_pp_t pp, void *hidden_dev, uint32_t n_floats, uint32_t peer);
int32_t nfai_hip_pp_send_token(nfai_pp_t pp, const void *token_dev, uint32_t peer);   /* one uint32 */
int32_t nfai_hip_pp_recv_token(nfai_pp_t pp, void *token_dev, uint32_t peer);
int32_t nfai_hip_pp_bcast_token(nfai_pp_t pp, void *token_dev, uint32_t root);        /* in place, every rank */
/* A tick's whole exchange in one call: group start, the listed sends / receives, group end (same stream, same semantics as the
 * per-operation entry points between _begin and _end). */
typedef struct nfai_pp_op {
    void *buf;        /* device pointer */
    uint32_t count;   /* floats (kinds 0, 1); ignored for tokens */
    uint32_t peer;
    uint32_t kind;    /* 0 send hidden, 1 receive hidden, 2 send token, 3 receive token */
    uint32_t reserved;
} nfai_pp_op;
int32_t nfai_hip_pp_exchange(nfai_pp_t pp, const nfai_pp_op *ops, uint32_t n_ops);
/* RCCL's own view of the communicator (ncclCommCount, ncclCommUserRank, ncclCommCuDevice) and the device's PCI bus id
 * (32-byte buffer): for run records.  Any output may be NULL. */
int32_t nfai_hip_pp_info(nfai_pp_t pp, uint32_t *nranks, uint32_t *rank, int32_t *device, char *pci_bus_id32);
/* Failure detection (the reference has none: a failed Vulkan call throws, VulkanBufferManager.cs:61-87, and nothing watches a queue).
 * RCCL reports a dead peer or a broken link asynchronously, after the enqueue calls returned.  _check: NFAI_OK, or NFAI_ERR_HIP with
 * RCCL's message and this rank when ncclCommGetAsyncError holds an error.  _wait: the bounded form of "synchronise the stage
 * stream" - polls the stream and the communicator until the stream is idle (NFAI_OK), an asynchronous error shows up, or
 * timeout_ms passes (NFAI_ERR_HIP naming the rank).  _abort: ncclCommAbort - releases operations that can no longer complete so that
 * the process can leave; the handle stays valid for _destroy only. */
int32_t nfai_hip_pp_check(nfai_pp_t pp);
int32_t nfai_hip_pp_wait(nfai_pp_t pp, uint32_t timeout_ms);
int32_t nfai_hip_pp_abort(nfai_pp_t pp);

#ifdef __cplusplus
}
#endif
#endif /* NFAI_HIP_H */
