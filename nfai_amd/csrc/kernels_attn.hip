// kernels_attn.hip — single-token attention over the KV cache: the reference's three dispatches
// AttentionScoreCalculationShader (…ScoreCalculationShader.cs:164-206), AttentionSoftmaxShader
// (AttentionSoftmaxShader.cs:139-178) and AttentionWeightedValueSumShader
// (…ValueSumShader.cs:175-216) as ONE pass over K and V.
//
// Bound: HBM (each cached K/V element is read once per token; 2 flop per element per query head).
//  - grid = (kv head, KV split): the G = H/Hkv query heads of a kv head share every K/V load (GQA);
//    the cache positions are cut into up to ATTN_NSPLIT_MAX slices so that a long context fills the
//    chip; each block writes (max, sum, unnormalised output) for its slice and a second tiny launch
//    merges the slices (log-sum-exp merge == the reference softmax up to rounding; its
//    clamp(s - max, -80, 80) only changes terms below e^-80).
//  - K/V rows go HBM -> VGPR with 16-byte non-temporal loads: D/4 lanes cover one position, so a
//    wave covers 64/(D/4) positions per load instruction; several positions are in flight per lane.
//  - scores of the block's slice live in LDS between the two phases; nothing is written to HBM
//    except the G*(D+2) floats per slice.
// The number of ACTIVE slices depends on the current sequence length, which is read from device
// memory (so a captured hipGraph can be replayed for every position); inactive blocks exit.
#include "common.h"

namespace nfai {

constexpr int ATTN_BLOCK = 256;
constexpr int ATTN_MIN_CHUNK = 16;   // positions per slice before another slice is opened
constexpr int ATTN_MAX_CHUNK = 1024; // LDS score capacity per query head (positions)
constexpr int ATTN_GMAX = 8;         // max query heads per kv head

struct AttnParams {
    const float *q;
    const void *kc, *vc;
    uint64_t pos_stride, head_stride;
    float *o;
    float *partials;  // [Hkv][NSPLIT_MAX][G][D + 2]
    uint32_t H, Hkv, D;
    const uint32_t *pos;
    int kv_f16;
};

__device__ __forceinline__ void attn_split(uint32_t S, uint32_t &nsplit, uint32_t &chunk)
{
    nsplit = (S + ATTN_MIN_CHUNK - 1) / ATTN_MIN_CHUNK;
    if (nsplit > ATTN_NSPLIT_MAX) nsplit = ATTN_NSPLIT_MAX;
    chunk = (S + nsplit - 1) / nsplit;
    nsplit = (S + chunk - 1) / chunk;
}

// load 4 consecutive cache elements as fp32
template <bool F16>
__device__ __forceinline__ f32x4 kv_load4(const void *base, uint64_t idx)
{
    if constexpr (F16) {
        const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(reinterpret_cast<const _Float16 *>(base) + idx));
        return f32x4{h2f_lo(w[0]), h2f_hi(w[0]), h2f_lo(w[1]), h2f_hi(w[1])};
    } else {
        const u32x4 w = load_nt16(reinterpret_cast<const float *>(base) + idx);
        return f32x4{__builtin_bit_cast(float, w[0]), __builtin_bit_cast(float, w[1]), __builtin_bit_cast(float, w[2]),
                     __builtin_bit_cast(float, w[3])};
    }
}

// LPP = lanes per position = D/4 (16 for D=64, 32 for D=128); G = query heads per kv head.
template <int LPP, int G, bool F16>
__global__ __launch_bounds__(ATTN_BLOCK) void k_attn_decode(const AttnParams p)
{
    constexpr int D = LPP * 4;
    constexpr int PPW = 64 / LPP;                 // positions per wave-instruction
    constexpr int NGRP = ATTN_BLOCK / LPP;        // position groups per block
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t S = p.pos[0] + 1;
    uint32_t nsplit, chunk;
    attn_split(S, nsplit, chunk);
    const uint32_t kvh = blockIdx.x, split = blockIdx.y;
    if (split >= nsplit) return;
    const uint32_t t0 = split * chunk, t1 = min(t0 + chunk, S), n = t1 - t0;

    float *stat = smem;                     // [G][2] = (slice max, slice sum of exp)
    float *sc = smem + 16;                  // [G][chunk_pad]
    const uint32_t chunk_pad = (chunk + 3) & ~3u;
    float *red = sc + G * chunk_pad;        // [G][NGRP][D] reduction of the V phase
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t grp = tid / LPP, li = tid % LPP;   // position group, lane within the position

    // this lane's 4 elements of each of the G query vectors
    f32x4 qv[G];
#pragma unroll
    for (int g = 0; g < G; g++) qv[g] = *reinterpret_cast<const f32x4 *>(p.q + (uint64_t)(kvh * G + g) * D + li * 4);
    const float scale = 1.0f / sqrtf((float)D);  // …ScoreCalculationShader.cs:93

    // ---- phase 1: scores of the slice ------------------------------------------------------
    const uint64_t hbase = (uint64_t)kvh * p.head_stride + li * 4;
    for (uint32_t i = grp; i < n; i += NGRP * 4) {
        f32x4 kx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t ii = i + u * NGRP;
            kx[u] = kv_load4<F16>(p.kc, (uint64_t)(t0 + min(ii, n - 1)) * p.pos_stride + hbase);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t ii = i + u * NGRP;
#pragma unroll
            for (int g = 0; g < G; g++) {
                float d = qv[g][0] * kx[u][0];
                d = fmaf(qv[g][1], kx[u][1], d);
                d = fmaf(qv[g][2], kx[u][2], d);
                d = fmaf(qv[g][3], kx[u][3], d);
                d = group_sum<LPP>(d);
                if (li == 0 && ii < n) sc[g * chunk_pad + ii] = d * scale;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: slice max and exp (AttentionSoftmaxShader.cs:148-169 restricted to the slice) --
    // wave w handles query heads w, w+4, ...
    for (uint32_t g = tid >> 6; g < G; g += ATTN_BLOCK / 64) {
        float m = -1.0e38f;
        for (uint32_t t = lane; t < n; t += 64) m = fmaxf(m, sc[g * chunk_pad + t]);
        m = wave_max(m);
        float sum = 0.f;
        for (uint32_t t = lane; t < n; t += 64) {
            const float e = expf(sc[g * chunk_pad + t] - m);
            sc[g * chunk_pad + t] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        if (lane == 0) { stat[g * 2] = m; stat[g * 2 + 1] = sum; }
    }
    __syncthreads();

    // ---- phase 3: weighted V sum over the slice ---------------------------------------------
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (uint32_t i = grp; i < n; i += NGRP * 4) {
        f32x4 vx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t ii = i + u * NGRP;
            vx[u] = kv_load4<F16>(p.vc, (uint64_t)(t0 + min(ii, n - 1)) * p.pos_stride + hbase);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t ii = i + u * NGRP;
            if (ii < n) {
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const float w = sc[g * chunk_pad + ii];
                    acc[g][0] = fmaf(w, vx[u][0], acc[g][0]);
                    acc[g][1] = fmaf(w, vx[u][1], acc[g][1]);
                    acc[g][2] = fmaf(w, vx[u][2], acc[g][2]);
                    acc[g][3] = fmaf(w, vx[u][3], acc[g][3]);
                }
            }
        }
    }
    // reduce over position groups through LDS: red[g][grp][D]
#pragma unroll
    for (int g = 0; g < G; g++) *reinterpret_cast<f32x4 *>(red + ((uint32_t)g * NGRP + grp) * D + li * 4) = acc[g];
    __syncthreads();
    for (uint32_t e = tid; e < (uint32_t)G * D; e += ATTN_BLOCK) {
        const uint32_t g = e / D, d = e % D;
        float sum = 0.f;
#pragma unroll 4
        for (int r = 0; r < NGRP; r++) sum += red[(g * NGRP + r) * D + d];
        if (nsplit == 1) {
            // single slice: normalise here (AttentionSoftmaxShader.cs:172-176: e * (1/sum))
            p.o[(uint64_t)(kvh * G + g) * D + d] = sum * (1.0f / stat[g * 2 + 1]);
        } else {
            p.partials[(((uint64_t)kvh * ATTN_NSPLIT_MAX + split) * G + g) * (D + 2) + d] = sum;
        }
    }
    if (nsplit > 1 && tid < (uint32_t)G) {
        float *pp = p.partials + (((uint64_t)kvh * ATTN_NSPLIT_MAX + split) * G + tid) * (D + 2) + D;
        pp[0] = stat[tid * 2];
        pp[1] = stat[tid * 2 + 1];
    }
}

// merge the slices: one block per query head, thread per output element
template <int DD>
__global__ __launch_bounds__(DD) void k_attn_merge(const AttnParams p)
{
    const uint32_t S = p.pos[0] + 1;
    uint32_t nsplit, chunk;
    attn_split(S, nsplit, chunk);
    if (nsplit == 1) return;  // the decode kernel already wrote the normalised output
    const uint32_t G = p.H / p.Hkv, h = blockIdx.x, kvh = h / G, g = h % G, d = threadIdx.x;
    const float *base = p.partials + (((uint64_t)kvh * ATTN_NSPLIT_MAX) * G + g) * (DD + 2);
    const uint64_t sstride = (uint64_t)G * (DD + 2);
    float m = -1.0e38f;
    for (uint32_t s = 0; s < nsplit; s++) m = fmaxf(m, base[s * sstride + DD]);
    float l = 0.f, o = 0.f;
    for (uint32_t s = 0; s < nsplit; s++) {
        const float f = expf(base[s * sstride + DD] - m);
        l = fmaf(base[s * sstride + DD + 1], f, l);
        o = fmaf(base[s * sstride + d], f, o);
    }
    p.o[(uint64_t)h * DD + d] = o * (1.0f / l);
}

size_t attn_partials_bytes(uint32_t H, uint32_t Hkv, uint32_t D)
{
    (void)Hkv;
    return (size_t)H * ATTN_NSPLIT_MAX * (D + 2) * sizeof(float);
}

template <int LPP, bool F16>
static hipError_t launch_g(const AttnParams &p, uint32_t G, dim3 grid, size_t lds, hipStream_t s)
{
    switch (G) {
        case 1: hipLaunchKernelGGL((k_attn_decode<LPP, 1, F16>), grid, dim3(ATTN_BLOCK), lds, s, p); break;
        case 2: hipLaunchKernelGGL((k_attn_decode<LPP, 2, F16>), grid, dim3(ATTN_BLOCK), lds, s, p); break;
        case 3: hipLaunchKernelGGL((k_attn_decode<LPP, 3, F16>), grid, dim3(ATTN_BLOCK), lds, s, p); break;
        case 4: hipLaunchKernelGGL((k_attn_decode<LPP, 4, F16>), grid, dim3(ATTN_BLOCK), lds, s, p); break;
        case 8: hipLaunchKernelGGL((k_attn_decode<LPP, 8, F16>), grid, dim3(ATTN_BLOCK), lds, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_attn_decode(const AttnArgs &a, hipStream_t s)
{
    if (a.Hkv == 0 || a.H % a.Hkv != 0) return hipErrorInvalidValue;
    const uint32_t G = a.H / a.Hkv;
    if (G > ATTN_GMAX) return hipErrorInvalidValue;
    if (a.D != 64 && a.D != 128) return hipErrorInvalidValue;
    AttnParams p{};
    p.q = a.q; p.kc = a.kcache; p.vc = a.vcache;
    p.pos_stride = a.kv_pos_stride; p.head_stride = a.kv_head_stride;
    p.o = a.o; p.partials = a.partials;
    p.H = a.H; p.Hkv = a.Hkv; p.D = a.D; p.pos = a.pos_dev;
    p.kv_f16 = a.kv_type == NFAI_F16;
    // LDS: scores for the largest slice the capacity C can produce + the V-phase reduction
    uint32_t max_chunk = (a.C + ATTN_NSPLIT_MAX - 1) / ATTN_NSPLIT_MAX;
    if (max_chunk < ATTN_MIN_CHUNK) max_chunk = ATTN_MIN_CHUNK;
    if (max_chunk > ATTN_MAX_CHUNK) return hipErrorInvalidValue;  // C <= 32768 positions
    max_chunk = (max_chunk + 3) & ~3u;
    const uint32_t lpp = a.D / 4, ngrp = ATTN_BLOCK / lpp;
    const size_t lds = ((size_t)G * max_chunk + (size_t)G * ngrp * a.D + 64) * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    const dim3 grid(a.Hkv, ATTN_NSPLIT_MAX);
    hipError_t e;
    const bool f16 = p.kv_f16;
    if (a.D == 64) e = f16 ? launch_g<16, true>(p, G, grid, lds, s) : launch_g<16, false>(p, G, grid, lds, s);
    else e = f16 ? launch_g<32, true>(p, G, grid, lds, s) : launch_g<32, false>(p, G, grid, lds, s);
    if (e != hipSuccess) return e;
    if (a.D == 64) hipLaunchKernelGGL((k_attn_merge<64>), dim3(a.H), dim3(64), 0, s, p);
    else hipLaunchKernelGGL((k_attn_merge<128>), dim3(a.H), dim3(128), 0, s, p);
    return hipGetLastError();
}

}  // namespace nfai
