// kernels_basic.hip — one HIP kernel per reference ShaderWrapper subclass (the 1:1 operator
// surface), plus the per-token bookkeeping kernels of the fused path.  gfx950, wave64.
//
// These are the small ops: they are launch-latency bound at batch 1, so each is a single small
// grid with coalesced accesses; the time that matters lives in kernels_gemv.hip / kernels_attn.hip.
#include "common.h"

namespace nfai {

// ---------------------------------------------------------------------------------------------
// TokenEmbedShader (TokenEmbedShader.cs:131-159): out[d] = table[tok*E + d].  fp16 tables are
// widened in-register (the reference widened them at upload, AbstractComputeCollection.cs:62-77).
// ---------------------------------------------------------------------------------------------
__global__ void k_embed(const void *table, int type, const uint32_t *tok, float *y, uint32_t E)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= E) return;
    const uint64_t row = (uint64_t)tok[0] * E;
    if (type == NFAI_F16) y[d] = (float)reinterpret_cast<const _Float16 *>(table)[row + d];
    else y[d] = reinterpret_cast<const float *>(table)[row + d];
}

hipError_t launch_embed(const void *table, int type, const uint32_t *tok, float *y, uint32_t E, hipStream_t s)
{
    k_embed<<<(E + 255) / 256, 256, 0, s>>>(table, type, tok, y, E);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// RMSNormShader (RMSNormShader.cs:124-151).  The reference has every thread re-sum the vector;
// here one 1024-thread block reduces once.  y = (x / sqrt(mean(x^2) + eps)) * g, same operation
// order per element (divide, then multiply).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_rmsnorm(const float *x, const float *g, float *y, uint32_t E, float eps)
{
    __shared__ float red[16];
    float ss = 0.f;
    for (uint32_t i = threadIdx.x; i < E; i += blockDim.x) {
        float v = x[i];
        ss = fmaf(v, v, ss);
    }
    ss = block_sum(ss, red);
    const float rms = sqrtf(ss / (float)E + eps);
    for (uint32_t i = threadIdx.x; i < E; i += blockDim.x) y[i] = (x[i] / rms) * g[i];
}

hipError_t launch_rmsnorm(const float *x, const float *g, float *y, uint32_t E, float eps, hipStream_t s)
{
    k_rmsnorm<<<1, 1024, 0, s>>>(x, g, y, E, eps);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// RoPEShader (RoPEShader.cs:231-272) on one n_heads x head_dim vector.  One thread per pair.
// ---------------------------------------------------------------------------------------------
__global__ void k_rope(const float *in, float *out, const float *freqs, uint32_t rope_dims, uint32_t n_heads,
                       uint32_t head_dim, uint32_t pos)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;  // pair index over all heads
    const uint32_t half = head_dim / 2;
    if (idx >= n_heads * half) return;
    const uint32_t h = idx / half, pair = (idx % half) * 2;
    const uint32_t i1 = h * head_dim + pair;
    const float a = in[i1], b = in[i1 + 1];
    if (pair < rope_dims) {
        const float theta = freqs[pair / 2] * (float)pos;
        const float c = cosf(theta), sn = sinf(theta);
        out[i1] = c * a - sn * b;
        out[i1 + 1] = sn * a + c * b;
    } else {
        out[i1] = a;
        out[i1 + 1] = b;
    }
}

hipError_t launch_rope(const float *in, float *out, const float *freqs, uint32_t rope_dims, uint32_t n_heads,
                       uint32_t head_dim, uint32_t pos, hipStream_t s)
{
    const uint32_t n = n_heads * head_dim / 2;
    k_rope<<<(n + 255) / 256, 256, 0, s>>>(in, out, freqs, rope_dims, n_heads, head_dim, pos);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// AttentionScoreCalculationShader (…ScoreCalculationShader.cs:164-206): one wave per (t, h);
// lanes stride the head dimension, wave reduction.  s[h*S + t] packed with stride S.
// ---------------------------------------------------------------------------------------------
__global__ void k_attn_scores(const float *q, const float *K, float *sc, uint32_t H, uint32_t Hkv, uint32_t D, uint32_t S)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (wave >= H * S) return;
    const uint32_t h = wave / S, t = wave % S;
    const uint32_t kvh = h / (H / Hkv);
    const float *k = K + (uint64_t)t * Hkv * D + (uint64_t)kvh * D;
    float dot = 0.f;
    for (uint32_t i = lane; i < D; i += 64) dot = fmaf(q[h * D + i], k[i], dot);
    dot = wave_sum(dot);
    if (lane == 0) sc[(uint64_t)h * S + t] = dot * (1.0f / sqrtf((float)D));
}

hipError_t launch_attn_scores(const float *q, const float *K, float *sc, uint32_t H, uint32_t Hkv, uint32_t D,
                              uint32_t S, hipStream_t s)
{
    const uint64_t waves = (uint64_t)H * S;
    k_attn_scores<<<(uint32_t)((waves + 3) / 4), 256, 0, s>>>(q, K, sc, H, Hkv, D, S);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// AttentionSoftmaxShader (AttentionSoftmaxShader.cs:139-178): one 256-thread block per head.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_attn_softmax(const float *sc, float *w, uint32_t S, float eps)
{
    __shared__ float red[16];
    const float *sh = sc + (uint64_t)blockIdx.x * S;
    float *wh = w + (uint64_t)blockIdx.x * S;
    float m = -1.0e38f;
    for (uint32_t t = threadIdx.x; t < S; t += blockDim.x) m = fmaxf(m, sh[t]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (uint32_t t = threadIdx.x; t < S; t += blockDim.x) {
        const float d = fminf(fmaxf(sh[t] - m, -80.0f), 80.0f);
        const float e = expf(d);
        wh[t] = e;
        sum += e;
    }
    sum = block_sum(sum, red);
    const float inv = sum > eps ? 1.0f / sum : 0.0f;
    for (uint32_t t = threadIdx.x; t < S; t += blockDim.x) wh[t] = wh[t] * inv;
}

hipError_t launch_attn_softmax(const float *sc, float *w, uint32_t H, uint32_t S, float eps, hipStream_t s)
{
    k_attn_softmax<<<H, 256, 0, s>>>(sc, w, S, eps);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// AttentionWeightedValueSumShader (…ValueSumShader.cs:175-216): block per head, thread per d,
// t ascending (coalesced across d).
// ---------------------------------------------------------------------------------------------
__global__ void k_attn_wsum(const float *w, const float *V, float *o, uint32_t H, uint32_t Hkv, uint32_t D, uint32_t S)
{
    const uint32_t h = blockIdx.x;
    const uint32_t kvh = h / (H / Hkv);
    for (uint32_t d = threadIdx.x; d < D; d += blockDim.x) {
        float acc = 0.f;
        for (uint32_t t = 0; t < S; t++)
            acc = fmaf(w[(uint64_t)h * S + t], V[(uint64_t)t * Hkv * D + (uint64_t)kvh * D + d], acc);
        o[h * D + d] = acc;
    }
}

hipError_t launch_attn_wsum(const float *w, const float *V, float *o, uint32_t H, uint32_t Hkv, uint32_t D,
                            uint32_t S, hipStream_t s)
{
    k_attn_wsum<<<H, 128, 0, s>>>(w, V, o, H, Hkv, D, S);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// SiLUShader / ElementWiseMultiplicationShader / host residual add
// ---------------------------------------------------------------------------------------------
__global__ void k_silu(const float *x, float *y, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = silu_ref(x[i]);
}
__global__ void k_mul(const float *a, const float *b, float *y, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] * b[i];
}
__global__ void k_add(const float *a, const float *b, float *y, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}
hipError_t launch_silu(const float *x, float *y, uint32_t n, hipStream_t s)
{
    k_silu<<<(n + 255) / 256, 256, 0, s>>>(x, y, n);
    return hipGetLastError();
}
hipError_t launch_mul(const float *a, const float *b, float *y, uint32_t n, hipStream_t s)
{
    k_mul<<<(n + 255) / 256, 256, 0, s>>>(a, b, y, n);
    return hipGetLastError();
}
hipError_t launch_add(const float *a, const float *b, float *y, uint32_t n, hipStream_t s)
{
    k_add<<<(n + 255) / 256, 256, 0, s>>>(a, b, y, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// SamplingUtils.ArgMax (SamplingUtils.cs:43-57): first index of the maximum.  Two-stage inside
// one launch: every block reduces a slice to (value, index); the last block to finish (device
// ticket) reduces the partials.  NaN-free inputs assumed, ties resolve to the LOWEST index.
// End-of-token bookkeeping for hipGraph replay rides on the final block: the chosen token is fed
// back (out_idx is the next step's token), logged into `ring`, and *pos_inc advances.
// ---------------------------------------------------------------------------------------------
struct ArgmaxPartials {
    float val[ARGMAX_BLOCKS];
    uint32_t idx[ARGMAX_BLOCKS];
    uint32_t ticket;
};

__device__ __forceinline__ void argmax_combine(float &v, uint32_t &i, float ov, uint32_t oi)
{
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__global__ __launch_bounds__(256) void k_argmax(const float *x, uint32_t n, uint32_t *out_idx, ArgmaxPartials *part,
                                                uint32_t *pos_inc, uint32_t *ring, uint32_t ring_len)
{
    __shared__ float sv[4];
    __shared__ uint32_t si[4];
    __shared__ uint32_t is_last;
    float v = -INFINITY;
    uint32_t idx = 0xFFFFFFFFu;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        argmax_combine(v, idx, x[i], i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const uint32_t oi = __shfl_xor(idx, o);
        argmax_combine(v, idx, ov, oi);
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = v; si[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) argmax_combine(v, idx, sv[w], si[w]);
        // publish this block's partial with write-through agent-scope stores, then take a ticket
        __hip_atomic_store(&part->val[blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&part->idx[blockIdx.x], idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(&part->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    // final reduction by the last-arriving block; partials are read with agent-scope loads (sc1)
    v = -INFINITY;
    idx = 0xFFFFFFFFu;
    if (threadIdx.x < gridDim.x) {
        v = __hip_atomic_load(&part->val[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idx = __hip_atomic_load(&part->idx[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const uint32_t oi = __shfl_xor(idx, o);
        argmax_combine(v, idx, ov, oi);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = v; si[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) argmax_combine(v, idx, sv[w], si[w]);
        out_idx[0] = idx;
        // re-arm for the next launch (stream-ordered; same access path as the fetch_add)
        __hip_atomic_store(&part->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pos_inc) {
            const uint32_t p = pos_inc[0];
            if (ring) ring[p % ring_len] = idx;
            pos_inc[0] = p + 1;
        }
    }
}

hipError_t launch_argmax(const float *x, uint32_t n, uint32_t *out_idx, void *partials, uint32_t *pos_inc,
                         uint32_t *ring, uint32_t ring_len, hipStream_t s)
{
    static_assert(ARGMAX_BLOCKS <= 256, "final reduction reads one partial per thread");
    k_argmax<<<ARGMAX_BLOCKS, 256, 0, s>>>(x, n, out_idx, reinterpret_cast<ArgmaxPartials *>(partials), pos_inc,
                                            ring, ring_len);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Candidates for SamplingUtils.TopP (SamplingUtils.cs:5-33), the sampler LlamaModel.RunAsync actually calls (LlamaModel.cs:130,
// 165).  The reference divides all V logits by the temperature, takes a softmax over V, sorts V (index, probability) pairs and
// keeps the first topK = 40 — on the host, after reading 513 KB of logits back.  Here ONE launch leaves on the device what the
// rest of TopP needs: the k largest logits with their indices (ties: lower index first, as the stable OrderByDescending :9-12
// orders equal probabilities), max_i(l_i / T) and sum_i exp(l_i / T - max): 8k + 8 bytes go to the host, which forms the k
// probabilities, cuts the nucleus and draws (llama.hip: nfai_hip_llama_decode_topk).
//   level 1: every wave owns a contiguous range of the logits (<= TOPK_NT per lane, in registers) and extracts its k best by k
//            rounds of [lane-local best -> wave-wide best (DPP inside rows of 16, v_readlane across rows) -> owner retires it];
//            list, (max, sum of exp) of the range -> workspace (write-through), one ticket per block;
//   level 2: the block whose ticket is last merges the sorted lists: a thread keeps the heads of its lists, k rounds of
//            [thread-local best head -> block-wide best -> owner advances]; (max, sum) pairs are combined in fixed order.
// NaN-free, finite logits assumed (as k_argmax).
// ---------------------------------------------------------------------------------------------
struct TopkWork {  // workspace header; the per-wave lists follow (topk_work_bytes)
    uint32_t ticket, pad[3];
    float out_v[TOPK_MAX];     // the k largest logits, descending (ties: lower index first)
    uint32_t out_i[TOPK_MAX];
    float M, S;                // max_i(l_i / T), sum_i exp(l_i / T - M)
    float pad2[2];
};

size_t topk_work_bytes(uint32_t n)
{
    const uint32_t nw = topk_blocks(n) * (TOPK_THREADS / 64);
    return sizeof(TopkWork) + (size_t)nw * (TOPK_MAX * 8 + 8);
}
size_t topk_out_offset() { return offsetof(TopkWork, out_v); }

__global__ __launch_bounds__(TOPK_THREADS) void k_topk(const float *x, uint32_t n, float temperature, uint32_t k, TopkWork *w)
{
    constexpr uint32_t WPB = TOPK_THREADS / 64;
    __shared__ float sv[WPB];
    __shared__ uint32_t si[WPB];
    __shared__ uint32_t is_last;
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t nw = gridDim.x * WPB, gw = blockIdx.x * WPB + wid;
    float *cand_v = reinterpret_cast<float *>(w + 1);                       // [nw][TOPK_MAX]
    uint32_t *cand_i = reinterpret_cast<uint32_t *>(cand_v + (size_t)nw * TOPK_MAX);
    float *wm = reinterpret_cast<float *>(cand_i + (size_t)nw * TOPK_MAX);  // [nw] range max of l / T
    float *ws = wm + nw;                                                     // [nw] range sum of exp(l / T - max)
    // ---- level 1 ----------------------------------------------------------------------------------------------------------
    const uint32_t cw = (n + nw - 1) / nw, base = gw * cw, end = min(n, base + cw);  // cw <= 64 * TOPK_NT (launch_topk)
    float v[TOPK_NT];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < (int)TOPK_NT; j++) {
        const uint32_t i = base + j * 64 + lane;
        v[j] = i < end ? x[i] : -INFINITY;
        m = fmaxf(m, v[j] / temperature);  // SamplingUtils.cs:7: l / temperature
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < (int)TOPK_NT; j++) sum += (base + j * 64 + lane < end) ? expf(v[j] / temperature - m) : 0.f;  // :38
    sum = wave_sum(sum);
    for (uint32_t t = 0; t < k; t++) {
        float bv = v[0];
        uint32_t bi = base + lane;
#pragma unroll
        for (int j = 1; j < (int)TOPK_NT; j++)
            if (v[j] > bv) { bv = v[j]; bi = base + j * 64 + lane; }  // strictly greater: the lower index of a tie stays
        wave_best(bv, bi);
#pragma unroll
        for (int j = 0; j < (int)TOPK_NT; j++)
            if (base + j * 64 + lane == bi) v[j] = -INFINITY;  // the owner retires it
        if (lane == 0) {
            __hip_atomic_store(&cand_v[(size_t)gw * TOPK_MAX + t], bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&cand_i[(size_t)gw * TOPK_MAX + t], bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lane == 0) {
        __hip_atomic_store(&wm[gw], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws[gw], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every storing wave drains, the block meets, ONE lane takes the ticket (the hand-off of k_argmax)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tk = __hip_atomic_fetch_add(&w->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (tk == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    // ---- level 2: k-way merge of the nw sorted lists (agent-scope loads: written by other workgroups) -----------------------
    constexpr int LPT = (TOPK_BLOCKS_MAX * WPB + TOPK_THREADS - 1) / TOPK_THREADS;  // lists per thread
    float hv[LPT];
    uint32_t hi[LPT], hp[LPT];
#pragma unroll
    for (int q = 0; q < LPT; q++) {
        const uint32_t list = threadIdx.x + q * TOPK_THREADS;
        hp[q] = 0;
        hv[q] = list < nw ? __hip_atomic_load(&cand_v[(size_t)list * TOPK_MAX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -INFINITY;
        hi[q] = list < nw ? __hip_atomic_load(&cand_i[(size_t)list * TOPK_MAX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
    }
    for (uint32_t t = 0; t < k; t++) {
        float bv = hv[0];
        uint32_t bi = hi[0];
#pragma unroll
        for (int q = 1; q < LPT; q++)
            if (topk_better(hv[q], hi[q], bv, bi)) { bv = hv[q]; bi = hi[q]; }
        wave_best(bv, bi);
        if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
        __syncthreads();
        bv = sv[0];
        bi = si[0];
#pragma unroll
        for (uint32_t q = 1; q < WPB; q++)
            if (topk_better(sv[q], si[q], bv, bi)) { bv = sv[q]; bi = si[q]; }
        if (threadIdx.x == 0) { w->out_v[t] = bv; w->out_i[t] = bi; }
#pragma unroll
        for (int q = 0; q < LPT; q++) {
            const uint32_t list = threadIdx.x + q * TOPK_THREADS;
            if (list < nw && hi[q] == bi && hv[q] == bv) {  // indices are unique: this thread's list holds the winner
                hp[q]++;
                const bool more = hp[q] < k;
                hv[q] = more ? __hip_atomic_load(&cand_v[(size_t)list * TOPK_MAX + hp[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -INFINITY;
                hi[q] = more ? __hip_atomic_load(&cand_i[(size_t)list * TOPK_MAX + hp[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
            }
        }
        __syncthreads();  // sv / si are rewritten by the next round
    }
    // (max, sum of exp): M = max over the ranges, S = sum_w s_w * exp(m_w - M), thread-strided then a fixed tree
    float M = -INFINITY;
    for (uint32_t q = threadIdx.x; q < nw; q += TOPK_THREADS) M = fmaxf(M, __hip_atomic_load(&wm[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    M = wave_max(M);
    if (lane == 0) sv[wid] = M;
    __syncthreads();
    M = sv[0];
#pragma unroll
    for (uint32_t q = 1; q < WPB; q++) M = fmaxf(M, sv[q]);
    __syncthreads();
    float S = 0.f;
    for (uint32_t q = threadIdx.x; q < nw; q += TOPK_THREADS) {
        const float mq = __hip_atomic_load(&wm[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float sq = __hip_atomic_load(&ws[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S += sq * expf(mq - M);
    }
    S = wave_sum(S);
    if (lane == 0) sv[wid] = S;
    __syncthreads();
    if (threadIdx.x == 0) {
        S = sv[0];
#pragma unroll
        for (uint32_t q = 1; q < WPB; q++) S += sv[q];
        w->M = M;
        w->S = S;
        __hip_atomic_store(&w->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm (stream-ordered with the next launch)
    }
}

hipError_t launch_topk(const float *x, uint32_t n, float temperature, uint32_t k, void *work, hipStream_t s)
{
    if (n == 0 || k == 0 || k > TOPK_MAX || k > n || !(temperature > 0.f)) return hipErrorInvalidValue;
    const uint32_t blocks = topk_blocks(n);
    if ((uint64_t)blocks * TOPK_THREADS * TOPK_NT < n) return hipErrorInvalidValue;  // more logits than TOPK_BLOCKS_MAX blocks hold
    k_topk<<<blocks, TOPK_THREADS, 0, s>>>(x, n, temperature, k, reinterpret_cast<TopkWork *>(work));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Per-token prologue of the fused path (one launch): embedding row -> x (TokenEmbedShader), and
// the cos/sin table of the current position, shared by every layer's RoPE epilogue
// (RoPEShader.cs:254-256 recomputes cos/sin per element per layer).
// ---------------------------------------------------------------------------------------------
__global__ void k_token_begin(const void *table, int type, const uint32_t *tok, float *x, uint32_t E,
                              const float *freqs, float *rope_cs, uint32_t n_freq, const uint32_t *pos_dev, uint32_t *epoch)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (epoch != nullptr && d == 0) epoch[0] = epoch[0] + 1;  // tag of this token's hand-offs inside the engine launches
    if (table != nullptr && d < E) {
        const uint64_t row = (uint64_t)tok[0] * E;
        if (type == NFAI_F16) x[d] = (float)reinterpret_cast<const _Float16 *>(table)[row + d];
        else x[d] = reinterpret_cast<const float *>(table)[row + d];
    }
    if (d < n_freq) {
        const float theta = freqs[d] * (float)pos_dev[0];
        rope_cs[2 * d] = cosf(theta);
        rope_cs[2 * d + 1] = sinf(theta);
    }
}

hipError_t launch_token_begin(const void *table, int type, const uint32_t *tok, float *x, uint32_t E,
                              const float *freqs, float *rope_cs, uint32_t n_freq, const uint32_t *pos_dev,
                              hipStream_t s, uint32_t *epoch)
{
    const uint32_t n = (E > n_freq ? E : n_freq) > 0 ? (E > n_freq ? E : n_freq) : 1;  // at least one block: the epoch word
    k_token_begin<<<(n + 255) / 256, 256, 0, s>>>(table, type, tok, x, E, freqs, rope_cs, n_freq, pos_dev, epoch);
    return hipGetLastError();
}

__global__ void k_pos_advance(uint32_t *pos_dev) { pos_dev[0] = pos_dev[0] + 1; }
hipError_t launch_pos_advance(uint32_t *pos_dev, hipStream_t s)
{
    k_pos_advance<<<1, 1, 0, s>>>(pos_dev);
    return hipGetLastError();
}

}  // namespace nfai
