// kernels_attn.hip — single-token attention over the KV cache: the reference's three dispatches
// AttentionScoreCalculationShader (…ScoreCalculationShader.cs:164-206), AttentionSoftmaxShader
// (AttentionSoftmaxShader.cs:139-178) and AttentionWeightedValueSumShader
// (…ValueSumShader.cs:175-216) as ONE launch and one pass over K and V.
//
// Bound: HBM (each cached K/V element is read once per token; 2 flop per element per query head).
//  - grid = (kv head, KV slice): the G = H/Hkv query heads of a kv head share every K/V load (GQA);
//    the cached positions are cut into up to ATTN_NSPLIT_MAX slices so that a few hundred positions
//    already fill the chip.  A block computes (max, sum of exp, unnormalised output) of its slice.
//  - K/V rows go HBM -> VGPR with 16-byte non-temporal loads: D/4 lanes cover one position, a wave
//    covers 64/(D/4) positions per load instruction; the first four K rows AND the first four V rows
//    of every lane group are requested before any arithmetic (V does not depend on the scores).
//  - the scores of the slice stay in LDS between the phases.
//  - the slices are merged in the SAME launch (log-sum-exp merge == the reference softmax up to rounding; its
//    clamp(s - max, -80, 80) only alters terms below e^-80), in fixed slice order 0..nsplit-1, so runs are bit-reproducible
//    and no result depends on which block finishes last.  Two hand-off forms:
//      POLL (the model path): every block publishes its sums and (max, sum of exp) as 8-byte {value, tag} granules (sc1
//        stores; tag = per-token epoch x blocks + block) and merges ITS share of the kv head's outputs after polling the
//        other slices' granules with sc1 loads — no drain, no ticket, no re-arm (MI355X guide, valid hand-off forms);
//      ticket (the op-level entry, small devices, NFAI_ATTN_POLL=0): every partial word is stored write-through, each storing
//        wave drains its stores, the block's barrier, ONE lane takes a ticket with an agent-scope atomic; the block whose
//        ticket is last reloads the partials and merges (guide, visibility table row 1).
//  - k_attn_wo (below): the same program in four waves of a workgroup whose other four waves stream Wo into registers
//    meanwhile and apply it to the merged output: attention + Wo + residual in one launch.
// The number of ACTIVE slices depends on the sequence length, which is read from device memory so
// a captured hipGraph can be replayed for every position; inactive blocks exit at once.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include <mutex>

namespace nfai {

constexpr int ATTN_BLOCK = 256;
constexpr int ATTN_MIN_CHUNK = 32;    // positions per slice before another slice is opened (swept 16..96 at 3B, context 520-776: 32 is best by 1.5 %)
constexpr int ATTN_MAX_CHUNK = 1024;  // LDS score capacity per query head (positions)
constexpr int ATTN_GMAX = 8;          // max query heads per kv head

struct AttnParams {
    const float *q;
    const void *kc, *vc;
    uint64_t pos_stride, head_stride;
    float *o;
    float *partials;     // [Hkv][NSPLIT_MAX][G][D + 2]
    uint32_t *tickets;   // [Hkv], zero between launches
    uint32_t H, Hkv, D;
    const uint32_t *pos;
    uint32_t min_chunk, max_split;  // slicing policy (defaults ATTN_MIN_CHUNK / ATTN_NSPLIT_MAX; env-tunable for sweeps)
    // granule hand-off (POLL): the workspace holds 8-byte {value, tag} granules [Hkv][NSPLIT_MAX][G][D + 2]
    const uint32_t *epoch;
    uint32_t tag_mul, tag_add;
    uint32_t *err;
    // fused attention + Wo launch (k_attn_wo): the merged attention output also travels as granules [H*D], read by the Wo waves
    uint64_t *att_gran;
    const uint8_t *wo;       // fp16 [E][HD]
    const float *wo_res;     // residual [E]
    float *wo_y;             // [E]
    uint32_t E, HD, nbw;     // Wo rows, row length, workgroups that share the rows
    uint32_t wo_lds_off;     // byte offset of the Wo waves' LDS region (after the attention waves')
    uint32_t wo_delay;       // x 64 clocks between launch start and the Wo waves' weight requests (workgroups that run a slice)
    uint32_t withhold;       // test hook (AttnArgs::debug_withhold)
    NFAI_STAMP_PARAM
};

__device__ __forceinline__ void attn_split(uint32_t S, uint32_t min_chunk, uint32_t max_split, uint32_t &nsplit, uint32_t &chunk)
{
    nsplit = (S + min_chunk - 1) / min_chunk;
    if (nsplit > max_split) nsplit = max_split;
    chunk = (S + nsplit - 1) / nsplit;
    nsplit = (S + chunk - 1) / chunk;
}

// load 4 consecutive cache elements as fp32
template <bool F16>
__device__ __forceinline__ f32x4 kv_load4(const void *base, uint64_t idx)
{
    if constexpr (F16) {
        const u32x2 w = __builtin_nontemporal_load(
            (const __attribute__((address_space(1))) u32x2 *)(reinterpret_cast<const _Float16 *>(base) + idx));
        return f32x4{h2f_lo(w[0]), h2f_hi(w[0]), h2f_lo(w[1]), h2f_hi(w[1])};
    } else {
        const u32x4 w = load_nt16(reinterpret_cast<const float *>(base) + idx);
        return __builtin_bit_cast(f32x4, w);  // whole-vector cast (element-wise bit_cast on w[i] reads w[0], hipcc 7.2)
    }
}

// sum over aligned groups of 16 lanes with DPP (no LDS crossbar); every lane gets its group's sum
__device__ __forceinline__ float row16_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

template <int LPP> __device__ __forceinline__ float pos_sum(float v)
{
    v = row16_sum(v);
    if constexpr (LPP == 32) {
        // two rows of 16 per position: v_permlane16_swap (gfx950) hands each row its neighbour's total inside the vector ALU.
        // The shuffle form (ds_bpermute + s_waitcnt lgkmcnt(0) per dot product, twelve in a row with nothing to overlap: seen
        // in the ISA) cost the score phase ~0.2 us per launch.  Same two addends, same sum.
        const uint32_t b = __builtin_bit_cast(uint32_t, v);
        const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);  // r[0]: rows {0,0,2,2}, r[1]: rows {1,1,3,3}
        v = __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
    }
    return v;
}

__device__ __forceinline__ void st_agent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// LPP = lanes per position = D/4 (16 for D=64, 32 for D=128); G = query heads per kv head.
// ONLINE: one pass over the slice — every lane group keeps a running (max, sum of exp, weighted V) for the positions it
// visits (online softmax), K and V rows of an iteration are requested together, and the groups are combined once at the
// end; the two-pass form (scores -> LDS, softmax by one wave per head, then V) needs two more barriers and walks K and V
// one after the other.
constexpr uint32_t ATTN_SPIN_CAP = 1u << 15;  // polling passes (a pass is a memory round trip + a short sleep): ~0.1 s, then give up

__device__ __forceinline__ void publish(uint64_t *g, uint32_t tag, float v)
{
    // one global_store_dwordx2 sc1: written through to memory, value and tag together (MI355X guide: a {value, tag} granule needs
    // no fence and no flag — the reader polls the data itself)
    __hip_atomic_store((__attribute__((address_space(1))) uint64_t *)g, ((uint64_t)tag << 32) | (uint64_t)__builtin_bit_cast(uint32_t, v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The attention waves' program.  PUB: the merged output is also published as {value, tag} granules (p.att_gran) for the Wo waves of
// the fused launch; every path through the body executes three workgroup barriers and, when the context has more than one slice,
// a fourth (k_attn_wo's Wo waves execute the same number).
template <int LPP, int G, bool F16, bool ONLINE, bool POLL, bool PUB>
__device__ __forceinline__ void attn_body(const AttnParams &p, const uint32_t kvh, const uint32_t split)
{
    constexpr int D = LPP * 4;
    constexpr int NGRP = ATTN_BLOCK / LPP;  // position groups per block
    constexpr int PF = 4;                   // positions per group per iteration
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t S = p.pos[0] + 1;
    uint32_t nsplit, chunk;
    attn_split(S, p.min_chunk, p.max_split, nsplit, chunk);
    if (split >= nsplit) return;
    STAMP_DECL;
    STAMP(0);
#ifdef NFAI_STAMPS
    const uint32_t stamp_wave = (split * p.Hkv + kvh) * (ATTN_BLOCK / 64) + (threadIdx.x >> 6);
#endif
    const uint32_t t0 = split * chunk, t1 = min(t0 + chunk, S), n = t1 - t0;

    float *stat = smem;               // [G][2] = (slice max, slice sum of exp); [32] = last-block flag
    float *sc = smem + 64;            // [G][chunk_pad]
    const uint32_t chunk_pad = (chunk + 3) & ~3u;
    float *red = sc + G * chunk_pad;  // [G][NGRP][D] reduction of the V phase
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t grp = tid / LPP, li = tid % LPP;  // position group, lane within the position

    // ---- requests first: this group's first PF K rows and first PF V rows, then q --------------
    const uint64_t hbase = (uint64_t)kvh * p.head_stride + li * 4;
    f32x4 kx[PF], vx[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) kx[u] = kv_load4<F16>(p.kc, (uint64_t)(t0 + min(grp + u * NGRP, n - 1)) * p.pos_stride + hbase);
#pragma unroll
    for (int u = 0; u < PF; u++) vx[u] = kv_load4<F16>(p.vc, (uint64_t)(t0 + min(grp + u * NGRP, n - 1)) * p.pos_stride + hbase);
    f32x4 qv[G];
#pragma unroll
    for (int g = 0; g < G; g++)
        qv[g] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>(
            (const __attribute__((address_space(1))) float *)p.q + (uint64_t)(kvh * G + g) * D + li * 4);
    const float scale = 1.0f / sqrtf((float)D);  // …ScoreCalculationShader.cs:93
    STAMP(1);  // first K rows, V rows and q requested

    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (ONLINE) {
        constexpr uint32_t STEP = NGRP * PF;
        const uint32_t niter = (n + STEP - 1) / STEP;  // block-uniform
        float m[G], l[G];
#pragma unroll
        for (int g = 0; g < G; g++) { m[g] = -1.0e38f; l[g] = 0.f; }
        auto load_kv = [&](f32x4 (&kr)[PF], f32x4 (&vr)[PF], uint32_t it) {
#pragma unroll
            for (int u = 0; u < PF; u++) kr[u] = kv_load4<F16>(p.kc, (uint64_t)(t0 + min(it * STEP + grp + u * NGRP, n - 1)) * p.pos_stride + hbase);
#pragma unroll
            for (int u = 0; u < PF; u++) vr[u] = kv_load4<F16>(p.vc, (uint64_t)(t0 + min(it * STEP + grp + u * NGRP, n - 1)) * p.pos_stride + hbase);
        };
        auto step = [&](const f32x4 (&kr)[PF], const f32x4 (&vr)[PF], uint32_t it) {
            float sv[PF][G];
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const bool live = it * STEP + grp + u * NGRP < n;
#pragma unroll
                for (int g = 0; g < G; g++) {
                    float d = qv[g][0] * kr[u][0];
                    d = fmaf(qv[g][1], kr[u][1], d);
                    d = fmaf(qv[g][2], kr[u][2], d);
                    d = fmaf(qv[g][3], kr[u][3], d);
                    d = pos_sum<LPP>(d);
                    sv[u][g] = live ? d * scale : -1.0e38f;
                }
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                float mn = m[g];
#pragma unroll
                for (int u = 0; u < PF; u++) mn = fmaxf(mn, sv[u][g]);
                const float c = __expf(m[g] - mn);  // 1 while nothing has been seen (both -1e38); v_exp_f32 path: 15 exponentials per iteration
                l[g] *= c;
                acc[g] *= c;
#pragma unroll
                for (int u = 0; u < PF; u++) {
                    const bool live = it * STEP + grp + u * NGRP < n;
                    const float e = live ? __expf(sv[u][g] - mn) : 0.f;
                    l[g] += e;
                    acc[g][0] = fmaf(e, vr[u][0], acc[g][0]);
                    acc[g][1] = fmaf(e, vr[u][1], acc[g][1]);
                    acc[g][2] = fmaf(e, vr[u][2], acc[g][2]);
                    acc[g][3] = fmaf(e, vr[u][3], acc[g][3]);
                }
                m[g] = mn;
            }
        };
        if (niter == 1) {
            step(kx, vx, 0);
        } else {
            f32x4 kb[PF], vb[PF];
            load_kv(kb, vb, 1);
            for (uint32_t it = 0; it < niter; it += 2) {  // unconditional, clamped loads: exact vmcnt bookkeeping
                step(kx, vx, it);
                load_kv(kx, vx, min(it + 2, niter - 1));
                if (it + 1 < niter) step(kb, vb, it + 1);
                load_kv(kb, vb, min(it + 3, niter - 1));
            }
        }
        // combine the lane groups: block max per head, rescale, block sum of exp
        float *gstat = red + G * NGRP * D;  // [G][NGRP][2]
        if (li == 0) {
#pragma unroll
            for (int g = 0; g < G; g++) { gstat[(g * NGRP + grp) * 2] = m[g]; gstat[(g * NGRP + grp) * 2 + 1] = l[g]; }
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; g++) {
            float M = -1.0e38f;
#pragma unroll
            for (int r = 0; r < NGRP; r++) M = fmaxf(M, gstat[(g * NGRP + r) * 2]);
            const float w = expf(m[g] - M);
            acc[g] *= w;
            if (tid == (uint32_t)g) {
                float L = 0.f;
#pragma unroll
                for (int r = 0; r < NGRP; r++) L += gstat[(g * NGRP + r) * 2 + 1] * expf(gstat[(g * NGRP + r) * 2] - M);
                stat[g * 2] = M;
                stat[g * 2 + 1] = L;
            }
        }
    } else {
        // ---- phase 1: scores of the slice -> LDS ------------------------------------------------
        // A slice of the benchmark's context is ONE iteration (STEP positions, everything requested above).  Longer slices
        // ping-pong over two register sets so that two iterations of K rows are in flight (the second set is only ever
        // loaded when the slice has a second iteration: nothing changes for short contexts).
        constexpr uint32_t STEP = NGRP * PF;
        const uint32_t niter = (n + STEP - 1) / STEP;  // block-uniform
        auto load_rows = [&](const void *cache, f32x4 (&r)[PF], uint32_t it) {
    #pragma unroll
            for (int u = 0; u < PF; u++) r[u] = kv_load4<F16>(cache, (uint64_t)(t0 + min(it * STEP + grp + u * NGRP, n - 1)) * p.pos_stride + hbase);
        };
        auto scores = [&](const f32x4 (&r)[PF], uint32_t it) {
    #pragma unroll
            for (int u = 0; u < PF; u++) {
                const uint32_t ii = it * STEP + grp + u * NGRP;
    #pragma unroll
                for (int g = 0; g < G; g++) {
                    float d = qv[g][0] * r[u][0];
                    d = fmaf(qv[g][1], r[u][1], d);
                    d = fmaf(qv[g][2], r[u][2], d);
                    d = fmaf(qv[g][3], r[u][3], d);
                    d = pos_sum<LPP>(d);
                    if (li == 0 && ii < n) sc[g * chunk_pad + ii] = d * scale;
                }
            }
        };
        // long slices: NSET register sets of rows in flight; the loads are unconditional (clamped to the last iteration) so
        // that hipcc's vmcnt bookkeeping stays exact; only the arithmetic is skipped past the end
        auto pipe = [&](auto nset_tag, const void *cache, f32x4 (&first)[PF], auto &&use) {
            constexpr int NSET = decltype(nset_tag)::value;
            f32x4 extra[NSET - 1][PF];
    #pragma unroll
            for (int j = 1; j < NSET; j++) load_rows(cache, extra[j - 1], min((uint32_t)j, niter - 1));
            for (uint32_t it = 0; it < niter; it += NSET) {
                use(first, it);
                load_rows(cache, first, min(it + NSET, niter - 1));
    #pragma unroll
                for (int j = 1; j < NSET; j++) {
                    if (it + j < niter) use(extra[j - 1], it + j);
                    load_rows(cache, extra[j - 1], min(it + j + NSET, niter - 1));
                }
            }
        };
        // two sets (measured at 8192 positions: four sets change nothing — 64 KB in flight per CU is not the limit there, the
        // sequential K phase / V phase / hand-off structure is)
        if (niter == 1) scores(kx, 0);
        else pipe(std::integral_constant<int, 2>{}, p.kc, kx, scores);
        __syncthreads();
        STAMP(2);  // scores of the slice in LDS

        // ---- phase 2: slice max, exp, sum (AttentionSoftmaxShader.cs:148-169 on the slice) -----------
        for (uint32_t g = tid >> 6; g < (uint32_t)G; g += ATTN_BLOCK / 64) {  // wave per query head
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
        STAMP(3);  // slice softmax done

        // ---- phase 3: weighted V sum over the slice ---------------------------------------------
        auto weigh = [&](const f32x4 (&r)[PF], uint32_t it) {
    #pragma unroll
            for (int u = 0; u < PF; u++) {
                const uint32_t ii = it * STEP + grp + u * NGRP;
                if (ii < n) {
    #pragma unroll
                    for (int g = 0; g < G; g++) {
                        const float w = sc[g * chunk_pad + ii];
                        acc[g][0] = fmaf(w, r[u][0], acc[g][0]);
                        acc[g][1] = fmaf(w, r[u][1], acc[g][1]);
                        acc[g][2] = fmaf(w, r[u][2], acc[g][2]);
                        acc[g][3] = fmaf(w, r[u][3], acc[g][3]);
                    }
                }
            }
        };
        if (niter == 1) weigh(vx, 0);
        else pipe(std::integral_constant<int, 2>{}, p.vc, vx, weigh);
    }
    // reduce over position groups through LDS: red[g][grp][D]
#pragma unroll
    for (int g = 0; g < G; g++) *reinterpret_cast<f32x4 *>(red + ((uint32_t)g * NGRP + grp) * D + li * 4) = acc[g];
    if (POLL && tid == 0) stat[32] = 0.f;
    __syncthreads();
    if constexpr (POLL) {
        // ---- granule hand-off, merge spread over the slices' own blocks: every block publishes its sums and (max, sum of exp) as
        //      {value, tag} granules, then merges ITS share of the kv head's G*D outputs (elements [GD*split/nsplit, GD*(split+1)/
        //      nsplit)) over all slices in the fixed order 0..nsplit-1.  A block polls a few hundred granules instead of one block
        //      per kv head polling all of them: a pass is one short memory round trip, and a retry costs little ---------------------
        constexpr uint32_t ROW = D + 2;  // granules per (slice, query head): D sums, slice max, slice sum of exp
        constexpr uint32_t GD = (uint32_t)G * D;
        const uint32_t tag = p.epoch[0] * p.tag_mul + p.tag_add;
        uint64_t *gbase = reinterpret_cast<uint64_t *>(p.partials) + (uint64_t)kvh * ATTN_NSPLIT_MAX * G * ROW;
        constexpr int PIT = (GD / 2 + ATTN_BLOCK - 1) / ATTN_BLOCK;  // element PAIRS per thread
#pragma unroll
        for (int it = 0; it < PIT; it++) {
            const uint32_t pi = tid + it * ATTN_BLOCK, e = min(pi * 2, GD - 2);
            const uint32_t g = e / D, d = e % D;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll 4
            for (int r = 0; r < NGRP; r++) {
                const f32x2 v = *reinterpret_cast<const f32x2 *>(red + (g * NGRP + r) * D + d);
                s0 += v[0];
                s1 += v[1];
            }
            if (pi * 2 < GD) {
                if (nsplit == 1) {
                    const float inv = 1.0f / stat[g * 2 + 1];  // AttentionSoftmaxShader.cs:172-176: e * (1/sum)
                    *reinterpret_cast<f32x2 *>(p.o + (uint64_t)(kvh * G + g) * D + d) = f32x2{s0 * inv, s1 * inv};
                    if constexpr (PUB) {
                        publish(p.att_gran + (uint64_t)(kvh * G + g) * D + d, tag, s0 * inv);
                        publish(p.att_gran + (uint64_t)(kvh * G + g) * D + d + 1, tag, s1 * inv);
                    }
                } else if (!(p.withhold && split + 1 == p.withhold && kvh == 0)) {  // (test hook: a slice that never publishes)
                    uint64_t *row = gbase + ((uint64_t)split * G + g) * ROW;
                    publish(row + d, tag, s0);
                    publish(row + d + 1, tag, s1);
                }
            }
        }
        if (nsplit == 1) { STAMP_FLUSH(p.stamps, stamp_wave, 4); return; }
        if (tid < (uint32_t)G) {
            uint64_t *row = gbase + ((uint64_t)split * G + tid) * ROW;
            publish(row + D, tag, stat[tid * 2]);
            publish(row + D + 1, tag, stat[tid * 2 + 1]);
        }
        STAMP(4);  // V phase, LDS reduction, granules issued
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)gbase, 0, (int)(ATTN_NSPLIT_MAX * G * ROW * 8), 0x00020000);
        float *mw = sc;           // [G][NSPLIT_MAX] merge weights exp(m_s - M), zero for inactive slices — sc is free now
        float *flag = stat + 32;  // non-zero: a poll gave up (cleared before the barrier above)
        constexpr int NGW = (G + ATTN_BLOCK / 64 - 1) / (ATTN_BLOCK / 64);  // query heads whose merge weights this wave computes
        const uint32_t wv = tid >> 6;
        // ONE sweep per pass: the (max, sum of exp) pairs of every query head (one wave per head, lane = slice; this block's own
        // pair comes from LDS) and all slices of this thread's output element (8-byte sc1 loads), checked together
        const uint32_t e0 = GD * split / nsplit, e1 = GD * (split + 1) / nsplit;
        for (uint32_t eb = e0; eb < e1; eb += ATTN_BLOCK) {  // block-uniform trip count (one trip unless GD / nsplit > 256)
            const bool first = eb == e0;
            const uint32_t e = min(eb + tid, e1 - 1);
            const uint32_t g = e / D, d = e % D;
            u32x2 a[ATTN_NSPLIT_MAX];
            u32x4 sv[NGW];
            bool ok = false;
            for (uint32_t spins = 0; spins < ATTN_SPIN_CAP; spins++) {
#pragma unroll
                for (int s2 = 0; s2 < (int)ATTN_NSPLIT_MAX; s2++)
                    a[s2] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)((((min((uint32_t)s2, nsplit - 1) * G + g) * ROW) + d) * 8), 0, 16));
                if (first) {
#pragma unroll
                    for (int j = 0; j < NGW; j++) {
                        const uint32_t gh = min(wv + j * (ATTN_BLOCK / 64), (uint32_t)G - 1);
                        sv[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((((min(lane, nsplit - 1) * G + gh) * ROW) + D) * 8), 0, 16));
                    }
                }
                ok = true;
#pragma unroll
                for (int s2 = 0; s2 < (int)ATTN_NSPLIT_MAX; s2++) ok = ok && a[s2][1] == tag;
                if (first) {
#pragma unroll
                    for (int j = 0; j < NGW; j++) ok = ok && (lane >= nsplit || lane == split || (sv[j][1] == tag && sv[j][3] == tag));
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!__all(ok)) {
                flag[0] = 1.f;
                if (lane == 0 && p.err) __hip_atomic_fetch_or(p.err, 0x1000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (first) {
#pragma unroll
                for (int j = 0; j < NGW; j++) {
                    const uint32_t gh = wv + j * (ATTN_BLOCK / 64);
                    if (gh < (uint32_t)G) {  // wave-uniform
                        const f32x4 vf = __builtin_bit_cast(f32x4, sv[j]);
                        const float m_s = lane == split ? stat[gh * 2] : vf[0], l_s = lane == split ? stat[gh * 2 + 1] : vf[2];
                        const float M = wave_max(lane < nsplit ? m_s : -1.0e38f);
                        const float f = lane < nsplit ? expf(m_s - M) : 0.0f;
                        const float L = wave_sum(lane < nsplit ? l_s * f : 0.0f);
                        if (lane < ATTN_NSPLIT_MAX) mw[gh * ATTN_NSPLIT_MAX + lane] = f;
                        if (lane == 0) stat[16 + gh] = 1.0f / L;
                    }
                }
                __syncthreads();
                STAMP(5);  // every granule of the first sweep seen, merge weights in LDS
            }
            if (flag[0] != 0.f) return;  // a poll gave up (the error word is set): no output is better than a wrong one
            if (eb + tid < e1) {
                float o = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < (int)ATTN_NSPLIT_MAX; s2++) {
                    const f32x2 vf = __builtin_bit_cast(f32x2, a[s2]);
                    o = fmaf(vf[0], (uint32_t)s2 < nsplit ? mw[g * ATTN_NSPLIT_MAX + s2] : 0.f, o);
                }
                o *= stat[16 + g];
                p.o[(uint64_t)(kvh * G + g) * D + d] = o;
                if constexpr (PUB) publish(p.att_gran + (uint64_t)(kvh * G + g) * D + d, tag, o);
            }
        }
#ifdef NFAI_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(7);  // this block's share merged and stored
        STAMP_FLUSH(p.stamps, stamp_wave, 8);
#endif
        return;
    } else {
    float *my_part = p.partials + ((uint64_t)kvh * ATTN_NSPLIT_MAX + split) * G * (D + 2);
    for (uint32_t e = tid; e < (uint32_t)G * D; e += ATTN_BLOCK) {
        const uint32_t g = e / D, d = e % D;
        float sum = 0.f;
#pragma unroll 4
        for (int r = 0; r < NGRP; r++) sum += red[(g * NGRP + r) * D + d];
        if (nsplit == 1) {
            // single slice: normalise here (AttentionSoftmaxShader.cs:172-176: e * (1/sum))
            p.o[(uint64_t)(kvh * G + g) * D + d] = sum * (1.0f / stat[g * 2 + 1]);
        } else {
            st_agent(my_part + g * (D + 2) + d, sum);
        }
    }
    if (nsplit == 1) { STAMP_FLUSH(p.stamps, stamp_wave, 4); return; }
    if (tid < (uint32_t)G) {
        st_agent(my_part + tid * (D + 2) + D, stat[tid * 2]);
        st_agent(my_part + tid * (D + 2) + D + 1, stat[tid * 2 + 1]);
    }
    // ---- hand-off: drain every storing wave, barrier, one ticket per block --------------------
    STAMP(4);  // V phase, LDS reduction, partial stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(5);  // every wave's partial stores acknowledged
    if (tid == 0) {
        const uint32_t t = __hip_atomic_fetch_add(&p.tickets[kvh], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stat[32] = (t == nsplit - 1) ? 1.0f : 0.0f;
    }
    __syncthreads();
    STAMP(6);  // ticket drawn
    if (stat[32] == 0.0f) { STAMP_FLUSH(p.stamps, stamp_wave, 7); return; }

    // ---- merge by the last block of this kv head: slices in fixed order 0..nsplit-1 --------------
    const float *base = p.partials + (uint64_t)kvh * ATTN_NSPLIT_MAX * G * (D + 2);
    float *mw = sc;  // [G][NSPLIT_MAX] merge weights exp(m_s - M), zero for inactive slices — sc is free now
    // ONE memory round trip: every thread requests all slices of its output elements (clamped
    // address, zero weight for inactive slices) and the per-slice (max, sum) pairs before anything
    // is used; the merge weights are then computed by one wave per query head (lane = slice).
    constexpr int EIT = (G * D + ATTN_BLOCK - 1) / ATTN_BLOCK;
    float a[EIT][ATTN_NSPLIT_MAX];
#pragma unroll
    for (int it = 0; it < EIT; it++) {
        const uint32_t e = min(tid + it * ATTN_BLOCK, (uint32_t)G * D - 1);
        const float *pp = base + (uint64_t)(e / D) * (D + 2) + (e % D);
#pragma unroll
        for (int s2 = 0; s2 < (int)ATTN_NSPLIT_MAX; s2++) a[it][s2] = ld_agent(pp + (uint64_t)min((uint32_t)s2, nsplit - 1) * G * (D + 2));
    }
    for (uint32_t g = tid >> 6; g < (uint32_t)G; g += ATTN_BLOCK / 64) {
        const uint32_t sl = min(lane, nsplit - 1);
        const float *pp = base + ((uint64_t)sl * G + g) * (D + 2) + D;
        const float m_s = ld_agent(pp), l_s = ld_agent(pp + 1);
        const float M = wave_max(lane < nsplit ? m_s : -1.0e38f);
        const float f = lane < nsplit ? expf(m_s - M) : 0.0f;
        const float L = wave_sum(l_s * f);
        if (lane < ATTN_NSPLIT_MAX) mw[g * ATTN_NSPLIT_MAX + lane] = f;
        if (lane == 0) stat[g * 2] = 1.0f / L;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EIT; it++) {
        const uint32_t e = tid + it * ATTN_BLOCK;
        if (e < (uint32_t)G * D) {
            const uint32_t g = e / D, d = e % D;
            float o = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < (int)ATTN_NSPLIT_MAX; s2++) o = fmaf(a[it][s2], mw[g * ATTN_NSPLIT_MAX + s2], o);
            p.o[(uint64_t)(kvh * G + g) * D + d] = o * stat[g * 2];
        }
    }
    if (tid == 0) __hip_atomic_store(&p.tickets[kvh], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
#ifdef NFAI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(7);  // merge done and stored (last block of the kv head only)
    STAMP_FLUSH(p.stamps, stamp_wave, 8);
#endif
    }
}

template <int LPP, int G, bool F16, bool ONLINE, bool POLL>
__global__ __launch_bounds__(ATTN_BLOCK) void k_attn_decode(const AttnParams p)
{
    attn_body<LPP, G, F16, ONLINE, POLL, false>(p, blockIdx.x, blockIdx.y);
}

// ---- attention + Wo in one launch ------------------------------------------------------------------------------------------
// Workgroup = four attention waves (the program above, kv head = id % Hkv, slice = id / Hkv) + four Wo waves.  The Wo waves
// request their share of Wo (rows id*E/nbw ..., R rows x U KiB per wave, straight to registers, non-temporal) and the residual at
// the START of the launch, so the weight stream overlaps the whole attention; then they take part in the attention waves'
// workgroup barriers (raw s_barrier: no drain of the loads in flight), poll the merged attention output — {value, tag}
// granules published by the blocks of the last slices — into LDS, meet on an LDS counter and finish with R x U dot products
// per lane from registers.  Saves one kernel boundary and the Wo launch's own first-byte latency per block (TransformerBlock.cs:
// 144-161 in one launch).  Every poll is bounded; a poll that gives up sets the error word and the Wo waves store nothing.
constexpr uint32_t WO_WAVES = 4;

__device__ __forceinline__ uint32_t wo_xs_index(uint32_t k)  // the activation layout of kernels_gemv.hip (fp16 weights)
{
    const uint32_t chunk = k >> 9, within = k & 511;
    return (chunk << 9) + (((within >> 2) & 1) << 8) + ((within >> 3) << 2) + (within & 3);
}

template <int LPP, int G, bool F16, int R, int U>
__global__ __launch_bounds__(ATTN_BLOCK + WO_WAVES * 64) void k_attn_wo(const AttnParams p)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t kvh = blockIdx.x % p.Hkv, split = blockIdx.x / p.Hkv;
    if (wave < ATTN_BLOCK / 64) {
        if (split < p.max_split) attn_body<LPP, G, F16, false, true, true>(p, kvh, split);
        return;
    }
    // ---- Wo waves ----------------------------------------------------------------------------------------------------------
    if (blockIdx.x >= p.nbw) return;
    typedef __attribute__((address_space(1))) uint8_t g_u8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(smem) + p.wo_lds_off);  // HD floats, permuted
    uint32_t *cnt = reinterpret_cast<uint32_t *>(xs + p.HD);                                     // one arrival word per Wo wave
    const uint32_t ww = wave - ATTN_BLOCK / 64, lane = threadIdx.x & 63;
    const uint32_t rb = (uint32_t)(((uint64_t)p.E * blockIdx.x) / p.nbw), re = (uint32_t)(((uint64_t)p.E * (blockIdx.x + 1)) / p.nbw);
    STAMP_DECL;
    STAMP(0);
#ifdef NFAI_STAMPS
    const uint32_t stamp_wave = p.Hkv * p.max_split * (ATTN_BLOCK / 64) + blockIdx.x * WO_WAVES + ww;  // behind the attention waves' rows
#endif
    // (1) the weights and the residual, requested a little after the slice's own K and V rows (below); then the attention waves'
    //     barriers: three per active slice, a fourth when there is more than one slice (see attn_body).
    uint32_t nsplit, chunk;
    attn_split(p.pos[0] + 1, p.min_chunk, p.max_split, nsplit, chunk);
    const bool has_slice = split < nsplit;
    // this wave's arrival word is cleared BEFORE a workgroup barrier and read by the others only after it: LDS keeps what an
    // earlier launch left there, and another model's launch may have left this very tag
    if (lane == 0) __hip_atomic_store(&cnt[ww], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // A CU's memory requests are served in order.  Requested at once, 18 KB of weights per wave sit in front of the slice's K and V
    // rows; requested after the slice's first barrier (scores in LDS, 3 us in) the stream is still arriving when the slice
    // publishes and the polls of the hand-off queue behind it.  Measured optimum (3B, positions 520-647; tokens/s): no delay 626,
    // 0.8 us 624, 1.3 us 631, 1.6 us 631, 1.9 us 628, 2.3 us 624, 3.2 us 613, after barrier 1: 618 (two launches: 621).
    if (has_slice)
        for (uint32_t i = 0; i < p.wo_delay; i++) __builtin_amdgcn_s_sleep(1);  // 64 clocks each
    asm volatile("" ::: "memory");
    u32x4 w[R][U];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t row = min(rb + ww * R + r, p.E - 1);
        const g_u8 *src = (const g_u8 *)p.wo + (uint64_t)row * p.HD * 2 + lane * 16;
#pragma unroll
        for (int u = 0; u < U; u++) w[r][u] = load_nt16((const void *)(src + u * 1024));
    }
    const float e = ((const __attribute__((address_space(1))) float *)p.wo_res)[min(rb + ww * R + min(lane, (uint32_t)R - 1), p.E - 1)];
    asm volatile("" ::: "memory");
    STAMP(1);  // weights requested
    __builtin_amdgcn_s_barrier();  // slice: barrier 1 of the attention waves; no slice: those waves end without a barrier
    if (has_slice) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        if (nsplit > 1) __builtin_amdgcn_s_barrier();
    }
    STAMP(2);  // the attention waves of this workgroup are done with barriers
    // (3) the attention output: this wave's quarter of the HD granules, 16-byte sc1 loads (two granules), until every tag matches
    const uint32_t tag = p.epoch[0] * p.tag_mul + p.tag_add;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.att_gran, 0, (int)(p.HD * 8), 0x00020000);
    u32x4 v[U];
    bool ok = false;
    for (uint32_t spins = 0; spins < ATTN_SPIN_CAP; spins++) {
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((ww * U + u) * 64 + lane) * 16), 0, 16));
        ok = true;
#pragma unroll
        for (int u = 0; u < U; u++) ok = ok && v[u][1] == tag && v[u][3] == tag;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    const bool good = __all(ok);
    STAMP(3);  // this wave's quarter of the attention output seen
#pragma unroll
    for (int u = 0; u < U; u++) {
        const f32x4 vf = __builtin_bit_cast(f32x4, v[u]);  // whole-vector cast (on an ELEMENT lvalue hipcc 7.2 reads element 0)
        const uint32_t k = (((ww * U + u) * 64 + lane) * 2);
        *reinterpret_cast<f32x2 *>(xs + wo_xs_index(k)) = f32x2{vf[0], vf[2]};  // k is even: the pair stays adjacent under the permutation
    }
    // (4) the four Wo waves meet on LDS words (the attention waves are done with barriers, some have ended): word ww = this launch's
    //     tag once wave ww's quarter is in LDS (tag + 1: its poll gave up)
    if (lane == 0) {
        if (!good && p.err) __hip_atomic_fetch_or(p.err, 0x2000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&cnt[ww], good ? tag : tag + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    for (uint32_t spins = 0;; spins++) {
        uint32_t seen = 0, bad = 0;
#pragma unroll
        for (uint32_t i = 0; i < WO_WAVES; i++) {
            const uint32_t a = __hip_atomic_load(&cnt[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            seen += (a == tag || a == tag + 1) ? 1u : 0u;
            bad += (a == tag + 1) ? 1u : 0u;
        }
        if (bad) return;
        if (seen == WO_WAVES) break;
        if (spins > (1u << 22)) {
            if (lane == 0 && p.err) __hip_atomic_fetch_or(p.err, 0x4000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    STAMP(4);  // all four quarters in LDS
    // (5) R rows x U KiB from registers
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xs + (u << 9) + (lane << 2));
        const f32x4 x1 = *reinterpret_cast<const f32x4 *>(xs + (u << 9) + 256 + (lane << 2));
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = dot8_f16(w[r][u], x0, x1, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t row = rb + ww * R + r;
        if (lane == (uint32_t)r && row < re) p.wo_y[row] = e + acc[r];  // host residual add of TransformerBlock.cs:153-158
    }
#ifdef NFAI_STAMPS
    STAMP(5);  // weights landed, multiplied, reduced, stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(6);
    STAMP_FLUSH(p.stamps, stamp_wave, 7);
#endif
}

size_t attn_partials_bytes(uint32_t H, uint32_t Hkv, uint32_t D, bool granules)
{
    // one ticket word per kv head (256 B reserved), then partial outputs + (max, sum) per (head, slice): 4-byte words in the ticket
    // form, 8-byte {value, tag} granules in the polling form
    (void)Hkv;
    return (size_t)H * ATTN_NSPLIT_MAX * (D + 2) * (granules ? sizeof(uint64_t) : sizeof(float)) + 256;
}

size_t attn_wo_extra_bytes(uint32_t H, uint32_t D) { return (size_t)H * D * sizeof(uint64_t) + 256; }

// The slices' workgroups of the granule form wait for each other, so ALL of them must be resident at once: what the occupancy
// query says a CU holds of this very kernel (registers, LDS, waves) times the CUs must cover the grid; otherwise the ticket form,
// which never waits, is launched.  (The query knows nothing of other processes on the device: for that case every wait is bounded
// and llama.hip re-runs the token on the ticket form.)
template <typename K>
static bool grid_resident(K kernel, uint32_t threads, size_t lds, uint32_t blocks, uint32_t n_cu)
{
    // per kernel instantiation: the last answer, keyed by (device, threads, LDS size — which only changes with the KV capacity).
    // Contexts on different threads share the memo, hence the lock (a launch path, taken once per enqueue, not per kernel).
    static std::mutex mu;
    static size_t memo_lds = ~(size_t)0;
    static uint32_t memo_threads = 0;
    static int memo_per_cu = 0, memo_dev = -1;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lk(mu);
    if (memo_lds != lds || memo_threads != threads || memo_dev != dev) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)threads, lds) != hipSuccess) { (void)hipGetLastError(); return false; }
        memo_per_cu = per_cu; memo_lds = lds; memo_threads = threads; memo_dev = dev;
    }
    return memo_per_cu > 0 && (uint64_t)memo_per_cu * n_cu >= blocks;
}

template <int LPP, bool F16, bool ONLINE, bool POLL>
static bool resident_gp(uint32_t G, dim3 grid, size_t lds, uint32_t n_cu)
{
    const uint32_t blocks = grid.x * grid.y;
    switch (G) {
        case 1: return grid_resident(k_attn_decode<LPP, 1, F16, ONLINE, POLL>, ATTN_BLOCK, lds, blocks, n_cu);
        case 2: return grid_resident(k_attn_decode<LPP, 2, F16, ONLINE, POLL>, ATTN_BLOCK, lds, blocks, n_cu);
        case 3: return grid_resident(k_attn_decode<LPP, 3, F16, ONLINE, POLL>, ATTN_BLOCK, lds, blocks, n_cu);
        case 4: return grid_resident(k_attn_decode<LPP, 4, F16, ONLINE, POLL>, ATTN_BLOCK, lds, blocks, n_cu);
        case 8: return grid_resident(k_attn_decode<LPP, 8, F16, ONLINE, POLL>, ATTN_BLOCK, lds, blocks, n_cu);
    }
    return false;
}

template <int LPP, bool F16, bool ONLINE, bool POLL>
static hipError_t launch_gp(const AttnParams &p, uint32_t G, dim3 grid, size_t lds, hipStream_t s)
{
    switch (G) {
        case 1: hipLaunchKernelGGL((k_attn_decode<LPP, 1, F16, ONLINE, POLL>), grid, dim3(ATTN_BLOCK), lds, s, p); break;
        case 2: hipLaunchKernelGGL((k_attn_decode<LPP, 2, F16, ONLINE, POLL>), grid, dim3(ATTN_BLOCK), lds, s, p); break;
        case 3: hipLaunchKernelGGL((k_attn_decode<LPP, 3, F16, ONLINE, POLL>), grid, dim3(ATTN_BLOCK), lds, s, p); break;
        case 4: hipLaunchKernelGGL((k_attn_decode<LPP, 4, F16, ONLINE, POLL>), grid, dim3(ATTN_BLOCK), lds, s, p); break;
        case 8: hipLaunchKernelGGL((k_attn_decode<LPP, 8, F16, ONLINE, POLL>), grid, dim3(ATTN_BLOCK), lds, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int LPP, bool F16, bool ONLINE>
static hipError_t launch_g(const AttnParams &p, uint32_t G, dim3 grid, size_t lds, hipStream_t s, uint32_t n_cu)
{
    if (p.epoch && resident_gp<LPP, F16, ONLINE, true>(G, grid, lds, n_cu)) return launch_gp<LPP, F16, ONLINE, true>(p, G, grid, lds, s);
    AttnParams q = p;
    q.epoch = nullptr;
    return launch_gp<LPP, F16, ONLINE, false>(q, G, grid, lds, s);
}

static hipError_t fill_params(const AttnArgs &a, AttnParams &p, dim3 &grid, size_t &lds, bool &online)
{
    if (a.Hkv == 0 || a.H % a.Hkv != 0 || a.Hkv > 64) return hipErrorInvalidValue;
    const uint32_t G = a.H / a.Hkv;
    if (G > ATTN_GMAX) return hipErrorInvalidValue;
    if (a.D != 64 && a.D != 128) return hipErrorInvalidValue;
    p.q = a.q; p.kc = a.kcache; p.vc = a.vcache;
    p.pos_stride = a.kv_pos_stride; p.head_stride = a.kv_head_stride;
    // workspace layout: [64 ticket words (zero between launches)] [partials]; the tickets sit at a
    // FIXED offset so that a workspace shared by calls of different shapes never aliases them
    p.o = a.o;
    p.tickets = reinterpret_cast<uint32_t *>(a.partials);
    p.partials = a.partials + 64;
    p.H = a.H; p.Hkv = a.Hkv; p.D = a.D; p.pos = a.pos_dev;
    static const int env_poll = getenv("NFAI_ATTN_POLL") ? atoi(getenv("NFAI_ATTN_POLL")) : 1;  // 0: ticket hand-off everywhere (A/B runs)
    // granule hand-off: the slices' workgroups poll each other, so all Hkv x slices of them must be resident at once; where the
    // occupancy query says they are not (a partition with few CUs) the ticket form, which never waits, is used
    p.epoch = env_poll ? a.epoch : nullptr;  // (launch_g checks that the whole grid can be resident)
    p.tag_mul = a.tag_mul; p.tag_add = a.tag_add; p.err = a.err; p.withhold = a.debug_withhold;
    static const int env_mc = getenv("NFAI_ATTN_MIN_CHUNK") ? atoi(getenv("NFAI_ATTN_MIN_CHUNK")) : 0;
    static const int env_ms = getenv("NFAI_ATTN_MAX_SPLIT") ? atoi(getenv("NFAI_ATTN_MAX_SPLIT")) : 0;
    p.min_chunk = env_mc >= 4 ? (uint32_t)env_mc : ATTN_MIN_CHUNK;
    p.max_split = (env_ms >= 1 && env_ms <= (int)ATTN_NSPLIT_MAX) ? (uint32_t)env_ms : ATTN_NSPLIT_MAX;
    // LDS: scalars + scores of the largest slice the capacity C can produce (also holds the merge
    // weights: G*NSPLIT_MAX*2 floats) + the V-phase reduction
    uint32_t max_chunk = (a.C + p.max_split - 1) / p.max_split;
    if (max_chunk < p.min_chunk) max_chunk = p.min_chunk;
    if (max_chunk > ATTN_MAX_CHUNK) return hipErrorInvalidValue;  // C <= 32768 positions
    max_chunk = (max_chunk + 3) & ~3u;
    if (max_chunk < 2 * ATTN_NSPLIT_MAX) max_chunk = 2 * ATTN_NSPLIT_MAX;
    const uint32_t lpp = a.D / 4, ngrp = ATTN_BLOCK / lpp;
    lds = (64 + (size_t)G * max_chunk + (size_t)G * ngrp * a.D + 64 + (size_t)G * ngrp * 2) * sizeof(float);  // + group stats of the one-pass form
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    grid = dim3(a.Hkv, p.max_split);
    // One-pass (online softmax, v_exp_f32) or two-pass form.  The position is device-side, so the choice is made from the KV
    // capacity the model was created with: above 2048 positions the one-pass form (8192 positions: 520 vs 493 tokens/s at
    // 3B), otherwise the two-pass form, which mirrors the reference's three dispatches literally (max, exp(clamp), sum with
    // libm expf) and is as fast at a few hundred positions (611 vs 610).  NFAI_ATTN_ONLINE=0/1 overrides.
    static const int env_online = getenv("NFAI_ATTN_ONLINE") ? atoi(getenv("NFAI_ATTN_ONLINE")) : -1;
    online = env_online == 1 || (env_online < 0 && a.C > 2048);
    return hipSuccess;
}

hipError_t launch_attn_decode(const AttnArgs &a, hipStream_t s)
{
    AttnParams p{};
    dim3 grid;
    size_t lds = 0;
    bool online = false;
    hipError_t e = fill_params(a, p, grid, lds, online);
    if (e != hipSuccess) return e;
    const uint32_t G = a.H / a.Hkv;
    const bool f16 = a.kv_type == NFAI_F16;
    NFAI_STAMP_SET(p, "attn_decode", a.Hkv * p.max_split, ATTN_BLOCK);
    if (online) {
        if (a.D == 64) return f16 ? launch_g<16, true, true>(p, G, grid, lds, s, a.n_cu) : launch_g<16, false, true>(p, G, grid, lds, s, a.n_cu);
        return f16 ? launch_g<32, true, true>(p, G, grid, lds, s, a.n_cu) : launch_g<32, false, true>(p, G, grid, lds, s, a.n_cu);
    }
    if (a.D == 64) return f16 ? launch_g<16, true, false>(p, G, grid, lds, s, a.n_cu) : launch_g<16, false, false>(p, G, grid, lds, s, a.n_cu);
    return f16 ? launch_g<32, true, false>(p, G, grid, lds, s, a.n_cu) : launch_g<32, false, false>(p, G, grid, lds, s, a.n_cu);
}

// ---- attention + Wo: the shapes it is built for (anything else takes the two launches) -------------------------------------
struct AttnWoShape { uint32_t lpp, G, R, U; };
static const AttnWoShape kAttnWoShapes[] = {
    {32, 3, 3, 6},  // Llama-3.2-3B: 24/8 heads of 128, E = HD = 3072 over 256 workgroups
    {32, 4, 4, 8},  // Llama-3.1-8B: 32/8 heads of 128, E = HD = 4096
    {16, 4, 2, 4},  // Llama-3.2-1B: 32/8 heads of 64, E = HD = 2048
    {32, 2, 1, 1},  // test model: 4/2 heads of 128, E = HD = 512 over 128 workgroups
};

static bool attn_wo_plan(const AttnArgs &a, const GemvArgs &g, uint32_t &nbw, AttnWoShape &shape)
{
    static const int env = getenv("NFAI_ATTN_WO") ? atoi(getenv("NFAI_ATTN_WO")) : 1;
    static const int env_poll = getenv("NFAI_ATTN_POLL") ? atoi(getenv("NFAI_ATTN_POLL")) : 1;
    static const int env_online = getenv("NFAI_ATTN_ONLINE") ? atoi(getenv("NFAI_ATTN_ONLINE")) : -1;
    if (!env || !env_poll || env_online == 1 || !a.epoch || a.C > 2048) return false;
    if (g.mode != GEMV_RESIDUAL || g.gamma || !g.res || g.x != a.o) return false;
    if (a.Hkv == 0 || a.H % a.Hkv || (a.D != 64 && a.D != 128)) return false;
    const uint32_t HD = a.H * a.D, E = g.seg_rows[0];
    if (g.K != HD || E == 0) return false;
    // (A Q4_K Wo was built into this launch too — one 16-row tile per workgroup, the Wo waves staging their pieces of the attention
    //  output as fixed-point MFMA fragments after the hand-off — and measured at 3B Q4_K_M: 973 tokens/s against 1019 with the two
    //  launches.  The staging, integer dot products and cross-wave sum that follow the hand-off take as long as the whole separate
    //  launch, whose stream is only 7 KB per CU; removed.  Round 4 tried the form that stages nothing — every Wo wave polls only the
    //  256 elements of its own super-block, fp32 sums of q x and of x per sub-block scaled as ggml dequantises, no meeting before the
    //  products — parity-green, 12.6 us against 7.95 + 3.6 as two launches, 1083 -> 1045 tokens/s: the hand-off costs what the
    //  launch boundary costs.  profiles/round4_attn_wo_q4_ab.txt)
    if (g.w_type != NFAI_F16 || HD % 512) return false;
    nbw = E / 4 < a.n_cu ? E / 4 : a.n_cu;
    if (nbw == 0) return false;
    const uint32_t rows = (E + nbw - 1) / nbw;
    shape = AttnWoShape{a.D / 4, a.H / a.Hkv, (rows + WO_WAVES - 1) / WO_WAVES, HD / 512};
    for (const AttnWoShape &k : kAttnWoShapes)
        if (k.lpp == shape.lpp && k.G == shape.G && k.R == shape.R && k.U == shape.U) return true;
    return false;
}

template <int LPP, int G, int R, int U>
static bool resident_aw(bool f16, uint32_t nblocks, size_t lds, uint32_t n_cu)
{
    return f16 ? grid_resident(k_attn_wo<LPP, G, true, R, U>, ATTN_BLOCK + WO_WAVES * 64, lds, nblocks, n_cu)
               : grid_resident(k_attn_wo<LPP, G, false, R, U>, ATTN_BLOCK + WO_WAVES * 64, lds, nblocks, n_cu);
}

template <int LPP, int G, int R, int U>
static hipError_t launch_aw(const AttnParams &p, bool f16, uint32_t nblocks, size_t lds, hipStream_t s)
{
    if (f16) hipLaunchKernelGGL((k_attn_wo<LPP, G, true, R, U>), dim3(nblocks), dim3(ATTN_BLOCK + WO_WAVES * 64), lds, s, p);
    else hipLaunchKernelGGL((k_attn_wo<LPP, G, false, R, U>), dim3(nblocks), dim3(ATTN_BLOCK + WO_WAVES * 64), lds, s, p);
    return hipGetLastError();
}

// shapes, workspace pointers, grid and LDS size of the fused launch; false: this pair of ops takes the two launches
static bool attn_wo_prepare(const AttnArgs &a, const GemvArgs &g, AttnParams &p, AttnWoShape &sh, uint32_t &nblocks, size_t &lds)
{
    uint32_t nbw;
    if (!attn_wo_plan(a, g, nbw, sh)) return false;
    dim3 grid;
    bool online = false;
    lds = 0;
    if (fill_params(a, p, grid, lds, online) != hipSuccess || online || !p.epoch) return false;
    const uint32_t HD = a.H * a.D;
    // the merged attention output as granules sits behind the slices' granules in the workspace
    p.att_gran = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(a.partials) + attn_partials_bytes(a.H, a.Hkv, a.D, true));
    p.wo = static_cast<const uint8_t *>(g.W[0]); p.wo_res = g.res; p.wo_y = g.y;
    p.E = g.seg_rows[0]; p.HD = HD; p.nbw = nbw;
    static const int env_delay = getenv("NFAI_ATTN_WO_DELAY") ? atoi(getenv("NFAI_ATTN_WO_DELAY")) : 54;  // x 64 clocks = 1.4 us
    p.wo_delay = env_delay >= 0 && env_delay < 4096 ? (uint32_t)env_delay : 54;
    p.wo_lds_off = (uint32_t)((lds + 15) & ~(size_t)15);
    nblocks = a.Hkv * p.max_split > nbw ? a.Hkv * p.max_split : nbw;
    lds = p.wo_lds_off + ((size_t)HD + 16) * sizeof(float);
    if (lds > 64 * 1024) return false;
    // every workgroup of this launch waits for others: the whole grid must be resident at once (occupancy query of the kernel itself)
    const bool f16 = a.kv_type == NFAI_F16;
    if (sh.lpp == 32 && sh.G == 3) return resident_aw<32, 3, 3, 6>(f16, nblocks, lds, a.n_cu);
    if (sh.lpp == 32 && sh.G == 4) return resident_aw<32, 4, 4, 8>(f16, nblocks, lds, a.n_cu);
    if (sh.lpp == 16 && sh.G == 4) return resident_aw<16, 4, 2, 4>(f16, nblocks, lds, a.n_cu);
    return resident_aw<32, 2, 1, 1>(f16, nblocks, lds, a.n_cu);
}

bool attn_wo_ok(const AttnArgs &a, const GemvArgs &g)
{
    AttnParams p{};
    AttnWoShape sh;
    uint32_t nblocks;
    size_t lds;
    return attn_wo_prepare(a, g, p, sh, nblocks, lds);
}

hipError_t launch_attn_wo(const AttnArgs &a, const GemvArgs &g, hipStream_t s)
{
    AttnParams p{};
    AttnWoShape sh;
    uint32_t nblocks;
    size_t lds;
    if (!attn_wo_prepare(a, g, p, sh, nblocks, lds)) return hipErrorInvalidValue;
    const bool f16 = a.kv_type == NFAI_F16;
    NFAI_STAMP_SET(p, "attn_wo", nblocks, ATTN_BLOCK + WO_WAVES * 64);
    if (sh.lpp == 32 && sh.G == 3) return launch_aw<32, 3, 3, 6>(p, f16, nblocks, lds, s);
    if (sh.lpp == 32 && sh.G == 4) return launch_aw<32, 4, 4, 8>(p, f16, nblocks, lds, s);
    if (sh.lpp == 16 && sh.G == 4) return launch_aw<16, 4, 2, 4>(p, f16, nblocks, lds, s);
    return launch_aw<32, 2, 1, 1>(p, f16, nblocks, lds, s);
}

}  // namespace nfai
