/*
 * nfai_oracle.c — CPU restatement of the NFAI Llama-3 TransformerBlock decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library, and only
 * as the checker / the reported CPU baseline.  The product path (nfai_amd/) never links,
 * imports or falls back to it.
 *
 * PARITY UNPINNED: the reference (NicuTheodorAlexandru/NFAI @ 2025-05-23) ships no tests, no
 * golden vectors and no fixtures for this path (NFAI.sln:6-23 lists no test project), and it
 * cannot be built or run here (C#/.NET 9 + Vulkan + glslangValidator, none present).  This file
 * restates the arithmetic of the GLSL string templates, function by function, citing the
 * reference file:line each one follows.  It is cross-checked against an independent fp64 NumPy
 * evaluation (tests/test_oracle.py) and pinned by fixtures generated from itself
 * (tests/golden/, generator committed), not by outputs of the reference.
 *
 * Semantics: every buffer is fp32, every sum is a sequential fp32 accumulation in the index
 * order the GLSL loops use, `a*b` and `+` are separately rounded (built with -ffp-contract=off;
 * whether a Vulkan driver contracts them is driver-defined).  sin/cos/exp/sqrt/pow are libm
 * fp32 (GLSL precision is driver-defined).  Reference defects that are undefined behaviour
 * (Q-RoPE out-of-bounds writes, score-mask race, KV overflow: SURVEY.md §8a) are NOT reproduced;
 * deterministic quirks (RoPE base hard-coded to 500000, frequency table truncated to 32 entries,
 * lm_head always tied) are selectable by parameters.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * fp16 <-> fp32.  Widening is exact (NFAI.Core/AbstractComputeCollection.cs:62-77 casts
 * System.Half -> float element by element at upload; HalfToSingle at :81-140 is the same map).
 * ------------------------------------------------------------------------------------------ */
ORC_API float orc_half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h >> 15) << 31;
    uint32_t exp = (h >> 10) & 0x1F;
    uint32_t man = h & 0x3FF;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int shift = 0;
            while ((man & 0x400) == 0) { man <<= 1; shift++; }
            man &= 0x3FF;
            bits = sign | ((uint32_t)(127 - 15 - shift + 1) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* round-to-nearest-even fp32 -> fp16 (used only by the build's own quantiser / fp16-KV option;
 * the reference never narrows). */
ORC_API uint16_t orc_float_to_half(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000;
    uint32_t man = x & 0x7FFFFF;
    int32_t exp = (int32_t)((x >> 23) & 0xFF);
    if (exp == 255) return (uint16_t)(sign | 0x7C00 | (man ? 0x200 | (man >> 13) : 0));
    exp = exp - 127 + 15;
    if (exp >= 31) return (uint16_t)(sign | 0x7C00);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000;
        int shift = 14 - exp;
        uint32_t half = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half & 1))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)exp << 10) | (man >> 13);
    uint32_t rem = man & 0x1FFF;
    if (rem > 0x1000 || (rem == 0x1000 && (half & 1))) half++;
    return (uint16_t)(sign | half);
}

ORC_API void orc_widen_f16(const uint16_t *src, float *dst, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) dst[i] = orc_half_to_float(src[i]);
}

ORC_API void orc_narrow_f16(const float *src, uint16_t *dst, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) dst[i] = orc_float_to_half(src[i]);
}

/* ------------------------------------------------------------------------------------------
 * TokenEmbedShader  (NFAI.Vulkan.Shaders/TokenEmbedShader.cs:135-158)
 *   out[d] = emb[dimPerToken * tokenId + d]
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_embed(const float *emb, uint32_t tok, uint32_t E, float *out)
{
    for (uint32_t d = 0; d < E; d++) out[d] = emb[(size_t)E * tok + d];
}

/* ------------------------------------------------------------------------------------------
 * RMSNormShader  (RMSNormShader.cs:126-150)
 *   sumSq sequential over i; rms = sqrt(sumSq/E + eps); y[i] = (x[i] / rms) * g[i]
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_rmsnorm(const float *x, const float *g, float *y, uint32_t E, float eps)
{
    float sumSq = 0.0f;
    for (uint32_t i = 0; i < E; i++) {
        float v = x[i];
        sumSq += v * v;
    }
    float meanSq = sumSq / (float)E;
    float rms = sqrtf(meanSq + eps);
    for (uint32_t i = 0; i < E; i++) y[i] = (x[i] / rms) * g[i];
}

/* ------------------------------------------------------------------------------------------
 * MatrixMultiplyShader, M = 1, transpose = true  (MatrixMultiplyShader.cs:262-288)
 *   y[j + cacheOffset] = sum_{k ascending} x[k] * W[j*K + k]     (fp32, one thread per j)
 * The cached variant's cacheOffset = currentContextSize * inputRows * outputCols (:286) is the
 * caller's `y` pointer offset here.  OpenMP over j does not change any value.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_gemv(const float *W, const float *x, float *y, uint32_t N, uint32_t K)
{
#pragma omp parallel for schedule(static)
    for (uint32_t j = 0; j < N; j++) {
        const float *w = W + (size_t)j * K;
        float sum = 0.0f;
        for (uint32_t k = 0; k < K; k++) {
            float a = x[k];
            float b = w[k];
            sum += a * b;
        }
        y[j] = sum;
    }
}

/* Same GEMV on fp16 storage: operand values are identical to the widened fp32 copy the
 * reference uploads (AbstractComputeCollection.cs:62-77), so results are bit-identical to
 * orc_gemv on the widened matrix.  Used so big-model baselines need 2 B/weight of host RAM. */
static float *g_h2f_table; /* 65536 exact widenings; built once (idempotent, so a race only repeats work) */

static const float *h2f_table(void)
{
    if (!g_h2f_table) {
        float *t = (float *)malloc(65536 * sizeof(float));
        for (uint32_t i = 0; i < 65536; i++) t[i] = orc_half_to_float((uint16_t)i);
        g_h2f_table = t;
    }
    return g_h2f_table;
}

ORC_API void orc_gemv_f16w(const uint16_t *W, const float *x, float *y, uint32_t N, uint32_t K)
{
    const float *tab = h2f_table();
#pragma omp parallel for schedule(static)
    for (uint32_t j = 0; j < N; j++) {
        const uint16_t *w = W + (size_t)j * K;
        float sum = 0.0f;
        for (uint32_t k = 0; k < K; k++) {
            float a = x[k];
            float b = tab[w[k]];
            sum += a * b;
        }
        y[j] = sum;
    }
}

/* ------------------------------------------------------------------------------------------
 * RoPE frequency table  (TransformerBlock.cs:33-38): base hard-coded to 500000 there;
 *   freq[i] = 1 / powf(base, (float)i / ((float)ropeDim / 2))
 * The reference uploads only the first 32 entries (TransformerBlock.cs:66) — `n_valid` = 32
 * with the rest zero reproduces that; n_valid = ropeDim/2 is the spec-correct table.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_rope_freqs(float *freqs, uint32_t rope_dims, float base, uint32_t n_valid)
{
    uint32_t half = rope_dims / 2;
    for (uint32_t i = 0; i < half; i++) {
        float f = 1.0f / powf(base, (float)i / ((float)rope_dims / 2.0f));
        freqs[i] = (i < n_valid) ? f : 0.0f;
    }
}

/* ------------------------------------------------------------------------------------------
 * RoPEShader  (RoPEShader.cs:238-271) applied to ONE vector of n_heads x head_dim at `pos`
 * (the Q call, and the row `pos` of the K cache; the out-of-bounds lanes of the Q dispatch are
 * UB in the reference and not reproduced — SURVEY.md §8a).
 *   pairIdx = 2*z; if pairIdx < ropeDimensions: theta = freq[pairIdx/2] * float(pos)
 *     out[i1] = cos*in[i1] - sin*in[i2];  out[i2] = sin*in[i1] + cos*in[i2]   else copy.
 * in == out is allowed (the K variant binds both to the cache, TransformerBlock.cs:73-74).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_rope(const float *in, float *out, const float *freqs, uint32_t rope_dims,
                      uint32_t n_heads, uint32_t head_dim, uint32_t pos)
{
    for (uint32_t h = 0; h < n_heads; h++) {
        for (uint32_t pair = 0; pair < head_dim; pair += 2) {
            uint32_t i1 = h * head_dim + pair;
            uint32_t i2 = i1 + 1;
            float a = in[i1], b = in[i2];
            if (pair < rope_dims) {
                float theta = freqs[pair / 2] * (float)pos;
                float c = cosf(theta);
                float s = sinf(theta);
                float v1 = c * a - s * b;
                float v2 = s * a + c * b;
                out[i1] = v1;
                out[i2] = v2;
            } else {
                out[i1] = a;
                out[i2] = b;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * AttentionScoreCalculationShader  (AttentionScoreCalculationShader.cs:166-205)
 *   kvh = h / (H / Hkv);  s[h*S + t] = (sum_{i ascending} q[h*D+i] * K[t*Hkv*D + kvh*D + i]) * scale
 *   scale = 1/sqrtf(D) (:93).  Only t < S is defined (mask writes for t >= S race, not kept).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_attn_scores(const float *q, const float *Kc, float *s, uint32_t H, uint32_t Hkv,
                             uint32_t D, uint32_t S)
{
    float scale = 1.0f / sqrtf((float)D);
    for (uint32_t h = 0; h < H; h++) {
        uint32_t kvh = h / (H / Hkv);
        for (uint32_t t = 0; t < S; t++) {
            const float *k = Kc + (size_t)t * Hkv * D + (size_t)kvh * D;
            float dot = 0.0f;
            for (uint32_t i = 0; i < D; i++) dot += q[h * D + i] * k[i];
            s[(size_t)h * S + t] = dot * scale;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * AttentionSoftmaxShader  (AttentionSoftmaxShader.cs:141-177)
 *   m = max_t s (start -1e38); e_t = exp(clamp(s_t - m, -80, 80)); sum sequential;
 *   inv = sum > eps ? 1/sum : 0;  w_t = e_t * inv.     eps = the model's rms epsilon
 *   (TransformerBlock.cs:115-116 passes it through).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_attn_softmax(const float *s, float *w, uint32_t H, uint32_t S, float eps)
{
    for (uint32_t h = 0; h < H; h++) {
        const float *sh = s + (size_t)h * S;
        float *wh = w + (size_t)h * S;
        float m = -1.0e38f;
        for (uint32_t t = 0; t < S; t++) m = fmaxf(m, sh[t]);
        float sum = 0.0f;
        for (uint32_t t = 0; t < S; t++) {
            float d = sh[t] - m;
            d = fminf(fmaxf(d, -80.0f), 80.0f);
            float e = expf(d);
            wh[t] = e;
            sum += e;
        }
        float inv = sum > eps ? 1.0f / sum : 0.0f;
        for (uint32_t t = 0; t < S; t++) wh[t] = wh[t] * inv;
    }
}

/* ------------------------------------------------------------------------------------------
 * AttentionWeightedValueSumShader  (AttentionWeightedValueSumShader.cs:177-215)
 *   o[h*D + d] = sum_{t ascending} w[h*S + t] * V[t*Hkv*D + kvh*D + d]
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_attn_wsum(const float *w, const float *Vc, float *o, uint32_t H, uint32_t Hkv,
                           uint32_t D, uint32_t S)
{
    for (uint32_t h = 0; h < H; h++) {
        uint32_t kvh = h / (H / Hkv);
        for (uint32_t d = 0; d < D; d++) {
            float acc = 0.0f;
            for (uint32_t t = 0; t < S; t++)
                acc += w[(size_t)h * S + t] * Vc[(size_t)t * Hkv * D + (size_t)kvh * D + d];
            o[h * D + d] = acc;
        }
    }
}

/* SiLUShader (SiLUShader.cs:108-127): x * (1.0 / (1.0 + exp(-x))) */
ORC_API void orc_silu(const float *x, float *y, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        float v = x[i];
        float sig = 1.0f / (1.0f + expf(-v));
        y[i] = v * sig;
    }
}

/* ElementWiseMultiplicationShader (ElementWiseMultiplicationShader.cs:123-138) */
ORC_API void orc_mul(const float *a, const float *b, float *y, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) y[i] = a[i] * b[i];
}

/* host residual add (TransformerBlock.cs:153-158, 176-180): C# float add */
ORC_API void orc_add(const float *a, const float *b, float *y, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) y[i] = a[i] + b[i];
}

/* SamplingUtils.ArgMax (SamplingUtils.cs:43-57): index of the FIRST maximum */
ORC_API uint32_t orc_argmax(const float *v, uint32_t n)
{
    uint32_t best = 0;
    float m = v[0];
    for (uint32_t i = 1; i < n; i++)
        if (v[i] > m) { m = v[i]; best = i; }
    return best;
}

/* SamplingUtils.TopP (SamplingUtils.cs:5-33), the sampler LlamaModel.RunAsync calls (LlamaModel.cs:130,165), with the draw of
 * Random.Shared.NextSingle() (:24) passed in as `rand` so that the function is deterministic:
 *   :7      scaled = l / temperature
 *   :8,35-41 Softmax: max, MathF.Exp(l - max), Sum, e / sum.  Enumerable.Sum over floats accumulates in DOUBLE and rounds once
 *           to float (System.Linq: Sum<float, double>); Max is exact
 *   :9-12   OrderByDescending(Prob) is a STABLE sort: equal probabilities keep index order
 *   :13     Take(topK)
 *   :14-21  cumulative (float) += prob; the element that makes cumulative >= topP is still included
 *   :22-23  total = Sum (double accumulation), normalised = prob / total
 *   :25-31  running (float) += prob; first element with rand < running; else the last one
 * ids_out / probs_out (topK entries, may be NULL): the sorted top-K (index, softmax probability) BEFORE the nucleus cut;
 * n_kept_out: elements that survive the cut.  Returns the sampled index. */
ORC_API uint32_t orc_topp(const float *values, uint32_t n, float temperature, float topP, uint32_t topK, float rand, uint32_t *ids_out,
                          float *probs_out, uint32_t *n_kept_out)
{
    float *probs = (float *)malloc((size_t)n * sizeof(float));
    uint8_t *taken = (uint8_t *)calloc(n, 1);
    float maxLogit = values[0] / temperature;
    for (uint32_t i = 0; i < n; i++) {
        probs[i] = values[i] / temperature;
        if (probs[i] > maxLogit) maxLogit = probs[i];
    }
    double sum = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        probs[i] = expf(probs[i] - maxLogit);
        sum += (double)probs[i];
    }
    const float sumExp = (float)sum;
    for (uint32_t i = 0; i < n; i++) probs[i] = probs[i] / sumExp;
    if (topK > n) topK = n;
    uint32_t *idx = (uint32_t *)malloc((size_t)topK * sizeof(uint32_t));
    for (uint32_t k = 0; k < topK; k++) { /* stable descending selection: strictly greater wins, so the lowest index of a tie comes first */
        uint32_t best = n;
        for (uint32_t i = 0; i < n; i++)
            if (!taken[i] && (best == n || probs[i] > probs[best])) best = i;
        taken[best] = 1;
        idx[k] = best;
        if (ids_out) ids_out[k] = best;
        if (probs_out) probs_out[k] = probs[best];
    }
    float cumulative = 0.f;
    uint32_t kept = 0;
    for (uint32_t k = 0; k < topK; k++) {
        cumulative += probs[idx[k]];
        kept++;
        if (cumulative >= topP) break;
    }
    double tot = 0.0;
    for (uint32_t k = 0; k < kept; k++) tot += (double)probs[idx[k]];
    const float total = (float)tot;
    float running = 0.f;
    uint32_t chosen = idx[kept - 1];
    for (uint32_t k = 0; k < kept; k++) {
        running += probs[idx[k]] / total;
        if (rand < running) { chosen = idx[k]; break; }
    }
    if (n_kept_out) *n_kept_out = kept;
    free(idx);
    free(taken);
    free(probs);
    return chosen;
}

/* ------------------------------------------------------------------------------------------
 * ggml K-quant block codecs.  NOT in the reference (NFAI.GGUF/Parser.cs:111-114 throws
 * "Unsupported data type" for Q4_K/Q6_K) and ggml itself is absent from /root/reference and
 * this image: restated from the published block layouts (ggml-common.h, block_q4_K / block_q6_K;
 * llama.cpp b3xxx-era, format unchanged since GGUF v2).  PARITY UNPINNED.
 *
 *  Q4_K super-block, 256 weights, 144 B: half d; half dmin; u8 scales[12]; u8 qs[128]
 *    8 sub-blocks of 32; 6-bit scale sc_j and min m_j unpacked as get_scale_min_k4:
 *      j<4 : sc = scales[j] & 63            m = scales[j+4] & 63
 *      j>=4: sc = (scales[j+4] & 0xF) | ((scales[j-4] >> 6) << 4)
 *            m  = (scales[j+4] >>  4) | ((scales[j]   >> 6) << 4)
 *    sub-blocks 2i / 2i+1 share qs[32i .. 32i+31]: low nibbles / high nibbles
 *    w = d*sc*q - dmin*m
 *  Q6_K super-block, 256 weights, 210 B: u8 ql[128]; u8 qh[64]; i8 scales[16]; half d
 *    two halves of 128: for l in 0..31, with ql/qh/sc advanced by 64/32/8 per half:
 *      q1 = (ql[l]    & 0xF) | ((qh[l] >> 0 & 3) << 4)   -> y[l]      scale sc[l/16 + 0]
 *      q2 = (ql[l+32] & 0xF) | ((qh[l] >> 2 & 3) << 4)   -> y[l+32]   scale sc[l/16 + 2]
 *      q3 = (ql[l]    >>  4) | ((qh[l] >> 4 & 3) << 4)   -> y[l+64]   scale sc[l/16 + 4]
 *      q4 = (ql[l+32] >>  4) | ((qh[l] >> 6 & 3) << 4)   -> y[l+96]   scale sc[l/16 + 6]
 *    w = d * sc * (q - 32)
 * ------------------------------------------------------------------------------------------ */
#define QK_K 256
#define Q4K_BYTES 144
#define Q6K_BYTES 210

static void q4k_scale_min(int j, const uint8_t *q, uint8_t *d, uint8_t *m)
{
    if (j < 4) {
        *d = q[j] & 63;
        *m = q[j + 4] & 63;
    } else {
        *d = (uint8_t)((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4));
        *m = (uint8_t)((q[j + 4] >> 4) | ((q[j] >> 6) << 4));
    }
}

ORC_API void orc_dequant_q4k(const uint8_t *blocks, float *out, size_t n_weights)
{
    size_t nb = n_weights / QK_K;
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < nb; b++) {
        const uint8_t *blk = blocks + b * Q4K_BYTES;
        uint16_t dh, mh;
        memcpy(&dh, blk, 2);
        memcpy(&mh, blk + 2, 2);
        float d = orc_half_to_float(dh);
        float dmin = orc_half_to_float(mh);
        const uint8_t *scales = blk + 4;
        const uint8_t *qs = blk + 16;
        float *y = out + b * QK_K;
        int is = 0;
        for (int j = 0; j < QK_K; j += 64) {
            uint8_t sc, m;
            q4k_scale_min(is + 0, scales, &sc, &m);
            float d1 = d * sc, m1 = dmin * m;
            q4k_scale_min(is + 1, scales, &sc, &m);
            float d2 = d * sc, m2 = dmin * m;
            for (int l = 0; l < 32; l++) y[j + l] = d1 * (float)(qs[l] & 0xF) - m1;
            for (int l = 0; l < 32; l++) y[j + 32 + l] = d2 * (float)(qs[l] >> 4) - m2;
            qs += 32;
            is += 2;
        }
    }
}

ORC_API void orc_dequant_q6k(const uint8_t *blocks, float *out, size_t n_weights)
{
    size_t nb = n_weights / QK_K;
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < nb; b++) {
        const uint8_t *blk = blocks + b * Q6K_BYTES;
        const uint8_t *ql = blk;
        const uint8_t *qh = blk + 128;
        const int8_t *sc = (const int8_t *)(blk + 192);
        uint16_t dh;
        memcpy(&dh, blk + 208, 2);
        float d = orc_half_to_float(dh);
        float *y = out + b * QK_K;
        for (int n = 0; n < QK_K; n += 128) {
            for (int l = 0; l < 32; l++) {
                int is = l / 16;
                int q1 = (int)((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                int q2 = (int)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                int q3 = (int)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                int q4 = (int)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l] = d * (float)sc[is + 0] * (float)q1;
                y[l + 32] = d * (float)sc[is + 2] * (float)q2;
                y[l + 64] = d * (float)sc[is + 4] * (float)q3;
                y[l + 96] = d * (float)sc[is + 6] * (float)q4;
            }
            y += 128;
            ql += 64;
            qh += 32;
            sc += 8;
        }
    }
}

/* The build's own (simple, deterministic) quantisers — NOT llama.cpp's search-based ones; they
 * only need to produce valid blocks so the decode path has realistic bytes to read.  Round trip
 * error is bounded in tests; parity of the PATH is defined on dequantise(blocks), whatever
 * produced the blocks. */
ORC_API void orc_quantize_q4k(const float *src, uint8_t *blocks, size_t n_weights)
{
    size_t nb = n_weights / QK_K;
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < nb; b++) {
        const float *x = src + b * QK_K;
        uint8_t *blk = blocks + b * Q4K_BYTES;
        float scales[8], mins[8];
        float max_scale = 0.0f, max_min = 0.0f;
        for (int j = 0; j < 8; j++) {
            float lo = x[32 * j], hi = x[32 * j];
            for (int l = 1; l < 32; l++) {
                float v = x[32 * j + l];
                if (v < lo) lo = v;
                if (v > hi) hi = v;
            }
            if (lo > 0.0f) lo = 0.0f; /* min is stored as a non-negative offset: w = d*q - m */
            scales[j] = (hi - lo) / 15.0f;
            mins[j] = -lo;
            if (scales[j] > max_scale) max_scale = scales[j];
            if (mins[j] > max_min) max_min = mins[j];
        }
        float d = max_scale / 63.0f, dmin = max_min / 63.0f;
        uint16_t dh = orc_float_to_half(d), mh = orc_float_to_half(dmin);
        float df = orc_half_to_float(dh), mf = orc_half_to_float(mh);
        uint8_t ls[8], lm[8];
        for (int j = 0; j < 8; j++) {
            int a = df > 0.0f ? (int)lrintf(scales[j] / df) : 0;
            int c = mf > 0.0f ? (int)lrintf(mins[j] / mf) : 0;
            ls[j] = (uint8_t)(a < 0 ? 0 : a > 63 ? 63 : a);
            lm[j] = (uint8_t)(c < 0 ? 0 : c > 63 ? 63 : c);
        }
        memcpy(blk, &dh, 2);
        memcpy(blk + 2, &mh, 2);
        uint8_t *sc = blk + 4;
        memset(sc, 0, 12);
        for (int j = 0; j < 8; j++) {
            if (j < 4) {
                sc[j] = ls[j];
                sc[j + 4] = lm[j];
            } else {
                sc[j + 4] = (uint8_t)((ls[j] & 0xF) | ((lm[j] & 0xF) << 4));
                sc[j - 4] |= (uint8_t)((ls[j] >> 4) << 6);
                sc[j] |= (uint8_t)((lm[j] >> 4) << 6);
            }
        }
        uint8_t *qs = blk + 16;
        for (int j = 0; j < 8; j += 2) {
            float d1 = df * ls[j], m1 = mf * lm[j];
            float d2 = df * ls[j + 1], m2 = mf * lm[j + 1];
            for (int l = 0; l < 32; l++) {
                int q1 = d1 > 0.0f ? (int)lrintf((x[32 * j + l] + m1) / d1) : 0;
                int q2 = d2 > 0.0f ? (int)lrintf((x[32 * (j + 1) + l] + m2) / d2) : 0;
                q1 = q1 < 0 ? 0 : q1 > 15 ? 15 : q1;
                q2 = q2 < 0 ? 0 : q2 > 15 ? 15 : q2;
                qs[l] = (uint8_t)(q1 | (q2 << 4));
            }
            qs += 32;
        }
    }
}

ORC_API void orc_quantize_q6k(const float *src, uint8_t *blocks, size_t n_weights)
{
    size_t nb = n_weights / QK_K;
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < nb; b++) {
        const float *x = src + b * QK_K;
        uint8_t *blk = blocks + b * Q6K_BYTES;
        float sub[16];
        float max_abs_scale = 0.0f;
        for (int j = 0; j < 16; j++) {
            float amax = 0.0f;
            for (int l = 0; l < 16; l++) {
                float v = fabsf(x[16 * j + l]);
                if (v > amax) amax = v;
            }
            sub[j] = amax / 31.0f;
            if (sub[j] > max_abs_scale) max_abs_scale = sub[j];
        }
        float d = max_abs_scale / 127.0f;
        uint16_t dh = orc_float_to_half(d);
        float df = orc_half_to_float(dh);
        int8_t sc[16];
        for (int j = 0; j < 16; j++) {
            int a = df > 0.0f ? (int)lrintf(sub[j] / df) : 0;
            sc[j] = (int8_t)(a > 127 ? 127 : a < 1 ? (sub[j] > 0.0f ? 1 : 0) : a);
        }
        uint8_t L[QK_K];
        for (int j = 0; j < 16; j++) {
            float dd = df * (float)sc[j];
            for (int l = 0; l < 16; l++) {
                int q = dd != 0.0f ? (int)lrintf(x[16 * j + l] / dd) : 0;
                q = q < -32 ? -32 : q > 31 ? 31 : q;
                L[16 * j + l] = (uint8_t)(q + 32);
            }
        }
        uint8_t *ql = blk, *qh = blk + 128;
        for (int n = 0; n < QK_K; n += 128) {
            for (int l = 0; l < 32; l++) {
                uint8_t q1 = L[n + l], q2 = L[n + l + 32], q3 = L[n + l + 64], q4 = L[n + l + 96];
                ql[l] = (uint8_t)((q1 & 0xF) | ((q3 & 0xF) << 4));
                ql[l + 32] = (uint8_t)((q2 & 0xF) | ((q4 & 0xF) << 4));
                qh[l] = (uint8_t)((q1 >> 4) | ((q2 >> 4) << 2) | ((q3 >> 4) << 4) | ((q4 >> 4) << 6));
            }
            ql += 64;
            qh += 32;
        }
        memcpy(blk + 192, sc, 16);
        memcpy(blk + 208, &dh, 2);
    }
}

/* ------------------------------------------------------------------------------------------
 * Whole-model restatement.
 *   graph   : LlamaModel ctor            NFAI.Models.Llama3/LlamaModel.cs:21-68
 *   per-token step : LlamaModel.RunAsync  LlamaModel.cs:103-126 / 134-165
 *   block   : TransformerBlock.Compute    NFAI.Vulkan.Shaders/TransformerBlock.cs:127-184
 * Weight matrices may be given as fp32 (the reference's in-VRAM form) or as fp16 (2 B/weight;
 * same operand values).  K/V caches are fp32 [C][Hkv*D] as in the reference
 * (MatrixMultiplyShader.cs:59-65 with contextSize).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t E, L, H, Hkv, D, F, V, C;
    float eps;
    float rope_base;      /* reference: 500000 hard-coded (TransformerBlock.cs:33) */
    uint32_t rope_dims;   /* llama.rope.dimension_count */
    uint32_t rope_n_freqs; /* entries of the freq table that are valid (reference: 32) */
    uint32_t weights_f16; /* 1: matrices are uint16 fp16; 0: fp32 */
} orc_llama_desc;

typedef struct {
    const float *attn_norm, *ffn_norm;
    const void *wq, *wk, *wv, *wo, *wgate, *wup, *wdown;
    float *kcache, *vcache; /* [C][Hkv*D] */
} orc_layer;

typedef struct orc_llama {
    orc_llama_desc d;
    const void *token_embd; /* [V][E] */
    const void *output;     /* lm_head [V][E]; NULL => tied to token_embd (LlamaModel.cs:64-67) */
    const float *output_norm;
    orc_layer *layers;
    float *freqs;
    uint32_t pos; /* TransformerBlock.currentToken (TransformerBlock.cs:25,183) */
    /* scratch */
    float *x, *xn, *q, *qr, *s, *w, *att, *proj, *h, *hn, *gate, *up, *act, *down;
} orc_llama;

static void gemv_any(const orc_llama *m, const void *W, const float *x, float *y, uint32_t N, uint32_t K)
{
    if (m->d.weights_f16) orc_gemv_f16w((const uint16_t *)W, x, y, N, K);
    else orc_gemv((const float *)W, x, y, N, K);
}

ORC_API orc_llama *orc_llama_create(const orc_llama_desc *d)
{
    orc_llama *m = (orc_llama *)calloc(1, sizeof(orc_llama));
    m->d = *d;
    m->layers = (orc_layer *)calloc(d->L, sizeof(orc_layer));
    size_t kv = (size_t)d->C * d->Hkv * d->D;
    for (uint32_t l = 0; l < d->L; l++) {
        m->layers[l].kcache = (float *)calloc(kv, sizeof(float));
        m->layers[l].vcache = (float *)calloc(kv, sizeof(float));
    }
    m->freqs = (float *)calloc(d->rope_dims / 2 + 1, sizeof(float));
    orc_rope_freqs(m->freqs, d->rope_dims, d->rope_base, d->rope_n_freqs);
    uint32_t HD = d->H * d->D;
    m->x = (float *)calloc(d->E, 4);
    m->xn = (float *)calloc(d->E, 4);
    m->q = (float *)calloc(HD, 4);
    m->qr = (float *)calloc(HD, 4);
    m->s = (float *)calloc((size_t)d->H * d->C, 4);
    m->w = (float *)calloc((size_t)d->H * d->C, 4);
    m->att = (float *)calloc(HD, 4);
    m->proj = (float *)calloc(d->E, 4);
    m->h = (float *)calloc(d->E, 4);
    m->hn = (float *)calloc(d->E, 4);
    m->gate = (float *)calloc(d->F, 4);
    m->up = (float *)calloc(d->F, 4);
    m->act = (float *)calloc(d->F, 4);
    m->down = (float *)calloc(d->E, 4);
    return m;
}

ORC_API void orc_llama_destroy(orc_llama *m)
{
    if (!m) return;
    for (uint32_t l = 0; l < m->d.L; l++) {
        free(m->layers[l].kcache);
        free(m->layers[l].vcache);
    }
    free(m->layers); free(m->freqs);
    free(m->x); free(m->xn); free(m->q); free(m->qr); free(m->s); free(m->w); free(m->att);
    free(m->proj); free(m->h); free(m->hn); free(m->gate); free(m->up); free(m->act); free(m->down);
    free(m);
}

ORC_API void orc_llama_set_globals(orc_llama *m, const void *token_embd, const void *output,
                                   const float *output_norm)
{
    m->token_embd = token_embd;
    m->output = output;
    m->output_norm = output_norm;
}

ORC_API void orc_llama_set_layer(orc_llama *m, uint32_t l, const float *attn_norm, const void *wq,
                                 const void *wk, const void *wv, const void *wo,
                                 const float *ffn_norm, const void *wgate, const void *wup,
                                 const void *wdown)
{
    orc_layer *y = &m->layers[l];
    y->attn_norm = attn_norm; y->wq = wq; y->wk = wk; y->wv = wv; y->wo = wo;
    y->ffn_norm = ffn_norm; y->wgate = wgate; y->wup = wup; y->wdown = wdown;
}

ORC_API void orc_llama_reset(orc_llama *m) { m->pos = 0; }
ORC_API uint32_t orc_llama_pos(const orc_llama *m) { return m->pos; }
ORC_API float *orc_llama_kcache(orc_llama *m, uint32_t l) { return m->layers[l].kcache; }
ORC_API float *orc_llama_vcache(orc_llama *m, uint32_t l) { return m->layers[l].vcache; }

/* One TransformerBlock.Compute (TransformerBlock.cs:127-184) on m->x in place. */
static void block_step(orc_llama *m, uint32_t l)
{
    const orc_llama_desc *d = &m->d;
    orc_layer *y = &m->layers[l];
    uint32_t HD = d->H * d->D, KD = d->Hkv * d->D, p = m->pos, S = p + 1;
    orc_rmsnorm(m->x, y->attn_norm, m->xn, d->E, d->eps);               /* :129 */
    gemv_any(m, y->wq, m->xn, m->q, HD, d->E);                           /* :131 */
    gemv_any(m, y->wk, m->xn, y->kcache + (size_t)p * KD, KD, d->E);     /* :133 cache row p */
    gemv_any(m, y->wv, m->xn, y->vcache + (size_t)p * KD, KD, d->E);     /* :135 */
    orc_rope(m->q, m->qr, m->freqs, d->rope_dims, d->H, d->D, p);        /* :138 */
    float *krow = y->kcache + (size_t)p * KD;
    orc_rope(krow, krow, m->freqs, d->rope_dims, d->Hkv, d->D, p);       /* :141 in place */
    orc_attn_scores(m->qr, y->kcache, m->s, d->H, d->Hkv, d->D, S);      /* :144 */
    orc_attn_softmax(m->s, m->w, d->H, S, d->eps);                       /* :146 */
    orc_attn_wsum(m->w, y->vcache, m->att, d->H, d->Hkv, d->D, S);       /* :148 */
    gemv_any(m, y->wo, m->att, m->proj, d->E, HD);                       /* :150 */
    orc_add(m->x, m->proj, m->h, d->E);                                  /* :153-158 */
    orc_rmsnorm(m->h, y->ffn_norm, m->hn, d->E, d->eps);                 /* :163 */
    gemv_any(m, y->wup, m->hn, m->up, d->F, d->E);                       /* :165 */
    gemv_any(m, y->wgate, m->hn, m->gate, d->F, d->E);                   /* :167 */
    orc_silu(m->gate, m->gate, d->F);                                    /* :169 */
    orc_mul(m->up, m->gate, m->act, d->F);                               /* :171 inputA=up, inputB=silu */
    gemv_any(m, y->wdown, m->act, m->down, d->E, d->F);                  /* :173 */
    orc_add(m->h, m->down, m->x, d->E);                                  /* :176-181 */
}

/* Run layers [l0, l1) on a hidden state (pipeline stage restatement; the reference has no
 * stages — a stage is a contiguous slice of the block loop at LlamaModel.cs:118-121). */
ORC_API void orc_llama_layers(orc_llama *m, float *hidden, uint32_t l0, uint32_t l1)
{
    memcpy(m->x, hidden, (size_t)m->d.E * 4);
    for (uint32_t l = l0; l < l1; l++) block_step(m, l);
    memcpy(hidden, m->x, (size_t)m->d.E * 4);
}

/* One token through the whole model: embed -> blocks -> output_norm -> lm_head
 * (LlamaModel.cs:116-125).  Writes V logits; advances pos.  Returns -1 on KV overflow (the
 * reference has no check, MatrixMultiplyShader.cs:248-252; the build's stance is a hard error). */
ORC_API int orc_llama_step(orc_llama *m, uint32_t tok, float *logits)
{
    const orc_llama_desc *d = &m->d;
    if (m->pos >= d->C) return -1;
    if (d->weights_f16) {
        const uint16_t *row = (const uint16_t *)m->token_embd + (size_t)tok * d->E;
        for (uint32_t i = 0; i < d->E; i++) m->x[i] = orc_half_to_float(row[i]);
    } else {
        orc_embed((const float *)m->token_embd, tok, d->E, m->x);
    }
    for (uint32_t l = 0; l < d->L; l++) block_step(m, l);
    orc_rmsnorm(m->x, m->output_norm, m->xn, d->E, d->eps);
    if (logits) gemv_any(m, m->output ? m->output : m->token_embd, m->xn, logits, d->V, d->E);
    m->pos++;
    return 0;
}

ORC_API const float *orc_llama_hidden(const orc_llama *m) { return m->x; }
ORC_API const float *orc_llama_normed(const orc_llama *m) { return m->xn; }
ORC_API void orc_llama_advance(orc_llama *m) { m->pos++; }

#ifdef _OPENMP
#include <omp.h>
ORC_API int orc_num_threads(void) { return omp_get_max_threads(); }
ORC_API void orc_set_num_threads(int n) { omp_set_num_threads(n); }
#else
ORC_API int orc_num_threads(void) { return 1; }
ORC_API void orc_set_num_threads(int n) { (void)n; }
#endif
