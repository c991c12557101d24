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
// keeps the first topK = 40 — on the host, after reading 513 KB of logits back.  Here two short launches leave on the device what
// the rest of TopP needs: the k largest logits with their indices (ties: lower index first, as the stable OrderByDescending :9-12
// orders equal probabilities), M = max_i(l_i / T) and S = sum_i exp(l_i / T - M): 8k + 8 bytes go to the host, which forms the k
// probabilities, cuts the nucleus and draws (llama.hip: nfai_hip_llama_decode_topk).
//   k_topk_heads : every wave owns a contiguous range of the logits (<= TOPK_NT per lane) and stores the range's best (value,
//                  index) — its "head".
//   k_topk_select: (a) every workgroup ranks the nw heads (LDS, counting) and takes tau = the k-th best of them.  k heads are at
//                  least as good as tau, so each of the k best LOGITS is at least as good as tau — and sits in one of the k ranges
//                  whose head is; (b) every wave sweeps its own range once more: what is at least as good as tau goes to a
//                  candidate list (k plus a handful on real logits; never more than k ranges' worth), and exp(l / T - M) is summed
//                  with the global M (the best head / T); (c) the workgroup whose ticket is last ranks the candidates by counting
//                  — (value descending, index ascending) is a strict total order over distinct indices — writes those ranked
//                  below k to out[rank], and adds the waves' sums in fixed order.
// Round 3's single launch extracted k elements per wave by k rounds of wave-wide maximum and merged the lists by k more rounds
// in one workgroup: 1.5 us per k (66 us at k = 40, rocprofv3); this form has no loop over k.  NaN-free, finite logits assumed
// (as k_argmax).  Fewer than k ranges (n < 4096): tau = -inf, every logit is a candidate.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t TOPK_CAND_CAP = TOPK_MAX * 64 * TOPK_NT;  // k ranges of at most 64 * TOPK_NT logits
constexpr uint32_t TOPK_LDS_CAP = 4096;                      // heads / candidates staged in LDS (more: read from memory)

struct TopkWork {  // workspace header; heads, per-wave sums and the candidate list follow (topk_work_bytes)
    uint32_t ticket, n_cand, pad[2];
    float out_v[TOPK_MAX];     // the k largest logits, descending (ties: lower index first)
    uint32_t out_i[TOPK_MAX];
    float M, S;                // max_i(l_i / T), sum_i exp(l_i / T - M)
    float pad2[2];
};

size_t topk_work_bytes(uint32_t n)
{
    const uint32_t nw = topk_blocks(n) * (TOPK_THREADS / 64);
    return sizeof(TopkWork) + (size_t)nw * 12 + (size_t)TOPK_CAND_CAP * 8;
}
size_t topk_out_offset() { return offsetof(TopkWork, out_v); }

struct TopkArrays {
    float *head_v; uint32_t *head_i; float *ws; float *cand_v; uint32_t *cand_i;
};
__device__ __forceinline__ TopkArrays topk_arrays(TopkWork *w, uint32_t nw)
{
    TopkArrays a;
    a.head_v = reinterpret_cast<float *>(w + 1);
    a.head_i = reinterpret_cast<uint32_t *>(a.head_v + nw);
    a.ws = reinterpret_cast<float *>(a.head_i + nw);
    a.cand_v = a.ws + nw;
    a.cand_i = reinterpret_cast<uint32_t *>(a.cand_v + TOPK_CAND_CAP);
    return a;
}

__global__ __launch_bounds__(TOPK_THREADS) void k_topk_heads(const float *x, uint32_t n, TopkWork *w)
{
    constexpr uint32_t WPB = TOPK_THREADS / 64;
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t nw = gridDim.x * WPB, gw = blockIdx.x * WPB + wid;
    const TopkArrays a = topk_arrays(w, nw);
    const uint32_t cw = (n + nw - 1) / nw, base = gw * cw, end = min(n, base + cw);  // cw <= 64 * TOPK_NT (launch_topk)
    float bv = -INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < (int)TOPK_NT; j++) {
        const uint32_t i = base + j * 64 + lane;
        const float v = i < end ? x[i] : -INFINITY;
        if (i < end && topk_better(v, i, bv, bi)) { bv = v; bi = i; }
    }
    wave_best(bv, bi);
    if (lane == 0) { a.head_v[gw] = bv; a.head_i[gw] = bi; }   // an empty range: (-inf, 0xFFFFFFFF)
}

// (value descending, index ascending) as ONE unsigned 64-bit order: the float's bits mapped monotonically to an unsigned word in
// the high half, the complemented index in the low half; a > b as keys <=> topk_better(a, b).  Key 0 is below every real element.
__device__ __forceinline__ unsigned long long topk_key(float v, uint32_t i)
{
    const uint32_t u = __float_as_uint(v + 0.0f);   // -0.0 -> +0.0: equal as floats, so the index decides (topk_better)
    const uint32_t f = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)f << 32) | (uint32_t)~i;
}
__device__ __forceinline__ float topk_key_value(unsigned long long key)
{
    const uint32_t f = (uint32_t)(key >> 32);
    return __uint_as_float((f & 0x80000000u) ? (f & 0x7FFFFFFFu) : ~f);
}
__device__ __forceinline__ uint32_t topk_key_index(unsigned long long key) { return ~(uint32_t)key; }

// how many of keys[0, n4) (n4 a multiple of 4, LDS, 16-byte aligned) are greater than `key`: 16-byte reads, eight in flight
__device__ __forceinline__ uint32_t topk_rank(const unsigned long long *keys, uint32_t n4, unsigned long long key)
{
    uint32_t rank = 0;
#pragma unroll 4
    for (uint32_t j = 0; j < n4; j += 4) {
        const ulonglong2 p = *reinterpret_cast<const ulonglong2 *>(keys + j);
        const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(keys + j + 2);
        rank += (p.x > key ? 1u : 0u) + (p.y > key ? 1u : 0u) + (q.x > key ? 1u : 0u) + (q.y > key ? 1u : 0u);
    }
    return rank;
}

__global__ __launch_bounds__(TOPK_THREADS) void k_topk_select(const float *x, uint32_t n, float temperature, uint32_t k, TopkWork *w)
{
    constexpr uint32_t WPB = TOPK_THREADS / 64;
    __shared__ __attribute__((aligned(16))) unsigned long long lk[TOPK_LDS_CAP];  // keys of the heads, later (last workgroup) of the candidates
    __shared__ float sv[WPB];
    __shared__ unsigned long long s_tau, s_max;
    __shared__ uint32_t is_last;
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t nw = gridDim.x * WPB, gw = blockIdx.x * WPB + wid;
    const TopkArrays a = topk_arrays(w, nw);
    // ---- (a) tau = the k-th best head, M from the best head (heads were written by the launch in front: plain loads) -----------
    const uint32_t nw4 = (nw + 3) & ~3u;
    for (uint32_t q = threadIdx.x; q < nw4; q += TOPK_THREADS) lk[q] = q < nw ? topk_key(a.head_v[q], a.head_i[q]) : 0ull;
    if (threadIdx.x == 0) s_tau = 0ull;   // fewer ranges than outputs: everything is a candidate
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < nw; q += TOPK_THREADS) {
        const unsigned long long key = lk[q];
        const uint32_t rank = topk_rank(lk, nw4, key);   // distinct indices -> distinct keys (empty ranges share the lowest one)
        if (rank == 0) s_max = key;
        if (nw >= k && rank == k - 1) s_tau = key;
    }
    __syncthreads();
    const unsigned long long tau = s_tau;
    const float M = topk_key_value(s_max) / temperature;   // SamplingUtils.cs:7: l / temperature, then :37 Max()
    // ---- (b) this wave's range: candidates and its share of the softmax denominator ------------------------------------------------
    const uint32_t cw = (n + nw - 1) / nw, base = gw * cw, end = min(n, base + cw);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < (int)TOPK_NT; j++) {
        const uint32_t i = base + j * 64 + lane;
        if (i < end) {
            const float v = x[i];
            sum += expf(v / temperature - M);                                                  // :38
            if (topk_key(v, i) >= tau && v > -INFINITY) {                                        // at least as good as tau
                const uint32_t slot = __hip_atomic_fetch_add(&w->n_cand, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (slot < TOPK_CAND_CAP) {  // cannot overflow: at most k ranges hold candidates (see above)
                    __hip_atomic_store(&a.cand_v[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&a.cand_i[slot], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) __hip_atomic_store(&a.ws[gw], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every storing wave drains, the workgroup meets, ONE lane takes the ticket (the hand-off of k_argmax)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tk = __hip_atomic_fetch_add(&w->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (tk == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    // ---- (c) rank the candidates (agent-scope loads: written by other workgroups) -------------------------------------------------
    const uint32_t m = min(__hip_atomic_load(&w->n_cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), TOPK_CAND_CAP);
    if (m <= TOPK_LDS_CAP) {
        const uint32_t m4 = (m + 3) & ~3u;
        for (uint32_t c = threadIdx.x; c < m4; c += TOPK_THREADS)
            lk[c] = c < m ? topk_key(__hip_atomic_load(&a.cand_v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                     __hip_atomic_load(&a.cand_i[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0ull;
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < m; c += TOPK_THREADS) {
            const unsigned long long key = lk[c];
            const uint32_t rank = topk_rank(lk, m4, key);
            if (rank < k) { w->out_v[rank] = topk_key_value(key); w->out_i[rank] = topk_key_index(key); }
        }
    } else {  // thousands of equal logits around tau: correct, slow (every comparison reads memory)
        for (uint32_t c = threadIdx.x; c < m; c += TOPK_THREADS) {
            const float v = __hip_atomic_load(&a.cand_v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t i = __hip_atomic_load(&a.cand_i[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t rank = 0;
            for (uint32_t j = 0; j < m && rank < k; j++)
                rank += topk_better(__hip_atomic_load(&a.cand_v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                    __hip_atomic_load(&a.cand_i[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), v, i) ? 1u : 0u;
            if (rank < k) { w->out_v[rank] = v; w->out_i[rank] = i; }
        }
    }
    // S = the waves' sums, thread-strided then a fixed tree (deterministic)
    float S = 0.f;
    for (uint32_t q = threadIdx.x; q < nw; q += TOPK_THREADS) S += __hip_atomic_load(&a.ws[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    S = wave_sum(S);
    __syncthreads();
    if (lane == 0) sv[wid] = S;
    __syncthreads();
    if (threadIdx.x == 0) {
        S = sv[0];
#pragma unroll
        for (uint32_t q = 1; q < WPB; q++) S += sv[q];
        w->M = M;
        w->S = S;
        __hip_atomic_store(&w->n_cand, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm (stream-ordered with the next launch)
        __hip_atomic_store(&w->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

hipError_t launch_topk(const float *x, uint32_t n, float temperature, uint32_t k, void *work, hipStream_t s)
{
    if (n == 0 || k == 0 || k > TOPK_MAX || k > n || !(temperature > 0.f)) return hipErrorInvalidValue;
    const uint32_t blocks = topk_blocks(n);
    if ((uint64_t)blocks * TOPK_THREADS * TOPK_NT < n) return hipErrorInvalidValue;  // more logits than TOPK_BLOCKS_MAX blocks hold
    static_assert(TOPK_BLOCKS_MAX * (TOPK_THREADS / 64) <= TOPK_LDS_CAP, "the heads are ranked in LDS");
    // fewer ranges than outputs: every logit is a candidate, which the list must hold
    if (blocks * (TOPK_THREADS / 64) < k && n > TOPK_CAND_CAP) return hipErrorInvalidValue;
    k_topk_heads<<<blocks, TOPK_THREADS, 0, s>>>(x, n, reinterpret_cast<TopkWork *>(work));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    k_topk_select<<<blocks, TOPK_THREADS, 0, s>>>(x, n, temperature, k, reinterpret_cast<TopkWork *>(work));
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
