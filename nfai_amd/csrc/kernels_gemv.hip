// kernels_gemv.hip — the weight-streaming GEMV family: MatrixMultiplyShader at M = 1
// (MatrixMultiplyShader.cs:255-289) rewritten for gfx950, with the neighbouring small ops of
// TransformerBlock.Compute (TransformerBlock.cs:127-184) fused in as prologue / epilogue.
//
// Bound: HBM.  Every weight byte is read exactly once per token, straight HBM -> VGPR with
// 16-byte non-temporal loads (one wave-instruction = one contiguous 1 KiB piece of a row); the
// weights never touch LDS (MI355X guide, "GEMV / M <= 16" row: an LDS round trip of once-read
// data is pure overhead).  LDS holds only what is re-used: the activation vector x (after the
// optional RMSNorm), laid out so each lane's ds_read_b128 is conflict-free.
//
//  work unit   = one output row, or a row PAIR that an epilogue needs together
//                (RoPE rotates rows 2i,2i+1; SiLU*up needs gate row j and up row j)
//  wave        = a contiguous, balanced range of units; walks them UPW units at a time as a flat
//                sequence of (unit group, K-chunk group) steps with the next step's loads issued
//                before the current step's FMAs (register double buffer) so R*U 1-KiB loads per
//                wave are always in flight, across row boundaries too
//  block       = 4..8 waves sharing x in LDS; grid = one or two blocks per CU, sized by the
//                host so that units divide evenly over all waves (no tail wave)
#include <stdlib.h>

#include "common.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace nfai {

struct GemvParams {
    const uint8_t *W[3];
    uint32_t seg_end[3];   // QKV: cumulative row ends of the q / k / v segments
    uint64_t row_bytes;
    const float *x;
    const float *gamma;
    float eps;
    uint32_t K, KC, NU;
    float *y;
    const float *res;
    void *kc, *vc;
    uint64_t kv_pos_stride, kv_head_stride;
    const float *rope_cs;
    uint32_t rope_dims, D;
    const uint32_t *pos;
    int kv_f16;
    uint32_t prefetch_only;  // 1: touch the first two steps of every wave's weights (default cache policy) and exit
    ArgmaxFused am;          // GEMV_PLAIN: first index of the largest output, taken in this launch (am.ticket == nullptr: off)
    BeginParams begin;       // GEMV_QKV_ROPE, split-K form only: the per-token prologue in this launch (begin.on == 0: off)
    // XD kernels: unit groups dealt to the XCDs by share.  Dealing round r hands out S consecutive groups: label x (blockIdx % 8) gets
    // the n[x] groups [r * S + off[x], + n[x]); the label's groups in round order are its slots, dealt round-robin to its waves.
    struct { uint16_t n[8], off[8]; uint32_t S; } xd;
    NFAI_STAMP_PARAM
};

#define GLOBAL_AS __attribute__((address_space(1)))

template <int WT> struct WTraits;
template <> struct WTraits<NFAI_F16> { static constexpr int EPL = 8; };  // elements per 16-byte lane load
template <> struct WTraits<NFAI_F32> { static constexpr int EPL = 4; };

// LDS index of activation element k (see header: two 1-KiB planes per 512-element chunk for fp16
// weights so that lane l's two float4 reads sit at l*16 bytes in each plane).
template <int EPL> __device__ __forceinline__ uint32_t xs_index(uint32_t k)
{
    if constexpr (EPL == 8) {
        const uint32_t chunk = k >> 9, within = k & 511;
        return (chunk << 9) + (((within >> 2) & 1) << 8) + ((within >> 3) << 2) + (within & 3);
    } else {
        return k;
    }
}

template <int WT, int U, bool GUARD>
__device__ __forceinline__ void issue_loads(u32x4 (&dst)[U], const uint8_t *row, uint32_t chunk0, uint32_t lane,
                                            uint32_t K)
{
    constexpr int EPL = WTraits<WT>::EPL;
    constexpr int EB = (WT == NFAI_F16) ? 2 : 4;
#pragma unroll
    for (int j = 0; j < U; j++) {
        const uint32_t k = (chunk0 + j) * (64 * EPL) + lane * EPL;
        if constexpr (GUARD) {
            if (k + EPL <= K) dst[j] = load_nt16(row + (uint64_t)k * EB);
            else dst[j] = u32x4{0, 0, 0, 0};
        } else {
            dst[j] = load_nt16(row + (uint64_t)k * EB);
        }
    }
}

template <int WT>
__device__ __forceinline__ float dot_chunk(u32x4 w, const float *xs, uint32_t chunk, uint32_t lane, float acc)
{
    if constexpr (WT == NFAI_F16) {
        const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xs + (chunk << 9) + (lane << 2));
        const f32x4 x1 = *reinterpret_cast<const f32x4 *>(xs + (chunk << 9) + 256 + (lane << 2));
        return dot8_f16(w, x0, x1, acc);
    } else {
        const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xs + (chunk << 8) + (lane << 2));
        // whole-vector bit cast: __builtin_bit_cast on a vector ELEMENT lvalue reads element 0 (hipcc 7.2)
        const f32x4 wf = __builtin_bit_cast(f32x4, w);
        acc = fmaf(wf[0], x0[0], acc);
        acc = fmaf(wf[1], x0[1], acc);
        acc = fmaf(wf[2], x0[2], acc);
        acc = fmaf(wf[3], x0[3], acc);
        return acc;
    }
}

template <int MODE>
__device__ __forceinline__ const uint8_t *row_ptr(const GemvParams &p, uint32_t unit, int sub)
{
    if constexpr (MODE == GEMV_GATEUP) {
        return p.W[sub] + (uint64_t)unit * p.row_bytes;  // sub 0 = gate row, 1 = up row
    } else if constexpr (MODE == GEMV_QKV_ROPE) {
        const uint32_t row = unit * 2 + sub;
        if (row < p.seg_end[0]) return p.W[0] + (uint64_t)row * p.row_bytes;
        if (row < p.seg_end[1]) return p.W[1] + (uint64_t)(row - p.seg_end[0]) * p.row_bytes;
        return p.W[2] + (uint64_t)(row - p.seg_end[1]) * p.row_bytes;
    } else {
        return p.W[0] + (uint64_t)unit * p.row_bytes;
    }
}

__device__ __forceinline__ void kv_store(void *base, int f16, uint64_t idx, float v)
{
    if (f16) reinterpret_cast<_Float16 *>(base)[idx] = (_Float16)v;
    else reinterpret_cast<float *>(base)[idx] = v;
}

// What the epilogue of `unit` will read from memory, requested when the unit's first K step is
// consumed so the latency is hidden behind the rest of the row (a load issued in the epilogue
// itself would add a full memory round trip to every short kernel).  Unconditional, clamped.
template <int MODE>
__device__ __forceinline__ void epilogue_prefetch(const GemvParams &p, uint32_t unit, float &e0, float &e1, const float *cs_lds = nullptr)
{
    if constexpr (MODE == GEMV_RESIDUAL) {
        e0 = ((const GLOBAL_AS float *)p.res)[unit];
    } else if constexpr (MODE == GEMV_QKV_ROPE) {
        const uint32_t row = unit * 2;
        const uint32_t r = row < p.seg_end[0] ? row : (row < p.seg_end[1] ? row - p.seg_end[0] : row - p.seg_end[1]);
        const uint32_t d = min(r % p.D, max(p.rope_dims, 2u) - 2);
        f32x2 cs;
        if (cs_lds) cs = *reinterpret_cast<const f32x2 *>(cs_lds + d);  // first launch of a token (split-K form): the workgroup's own table
        else cs = *reinterpret_cast<const GLOBAL_AS f32x2 *>((const GLOBAL_AS float *)p.rope_cs + d);  // [pair][2], pair = d/2
        e0 = cs[0];
        e1 = cs[1];
    }
}

// Epilogue of one unit; called by one lane with the fully reduced sums.
template <int MODE>
__device__ __forceinline__ void epilogue(const GemvParams &p, uint32_t unit, float a0, float a1, float e0, float e1, uint32_t pos)
{
    if constexpr (MODE == GEMV_PLAIN) {
        p.y[unit] = a0;
    } else if constexpr (MODE == GEMV_RESIDUAL) {
        // host residual add of TransformerBlock.cs:153-158 / 176-180: input + projection
        p.y[unit] = e0 + a0;
    } else if constexpr (MODE == GEMV_GATEUP) {
        // SiLUShader.cs:121-123 on the gate, ElementWiseMultiplicationShader.cs:137 with A = up
        p.y[unit] = a1 * silu_ref(a0);
    } else {
        // RoPEShader.cs:249-262 on the pair (row, row+1); V rows are stored unrotated
        const uint32_t row = unit * 2;
        const uint32_t seg = row < p.seg_end[0] ? 0u : (row < p.seg_end[1] ? 1u : 2u);
        const uint32_t r = seg == 0 ? row : (seg == 1 ? row - p.seg_end[0] : row - p.seg_end[1]);
        const uint32_t head = r / p.D, d = r % p.D;
        float o0 = a0, o1 = a1;
        if (seg < 2 && d < p.rope_dims) {
            o0 = e0 * a0 - e1 * a1;
            o1 = e1 * a0 + e0 * a1;
        }
        if (seg == 0) {
            p.y[row] = o0;
            p.y[row + 1] = o1;
        } else {
            const uint64_t idx = (uint64_t)pos * p.kv_pos_stride + (uint64_t)head * p.kv_head_stride + d;
            void *base = seg == 1 ? p.kc : p.vc;
            kv_store(base, p.kv_f16, idx, o0);
            kv_store(base, p.kv_f16, idx + 1, o1);
        }
    }
}

// Walks a wave's steps: (unit group g, K-chunk group cg).  One walker for the loads that are being
// issued (two steps ahead) and one for the step being consumed.
struct StepWalk {
    uint32_t g = 0, cg = 0;
    __device__ __forceinline__ void next(uint32_t cpg)
    {
        if (++cg == cpg) { cg = 0; ++g; }
    }
};

template <int WT, int MODE, int UPW, int U, bool GUARD, bool NORM, bool BEGIN = false, bool XD = false>
__global__ __launch_bounds__(512) void k_gemv(const GemvParams p)
{
    static_assert(!BEGIN || (MODE == GEMV_QKV_ROPE && NORM), "the per-token prologue rides on the RMSNorm'd q|k|v launch");
    constexpr int RPU = (MODE == GEMV_QKV_ROPE || MODE == GEMV_GATEUP) ? 2 : 1;
    constexpr int R = UPW * RPU;
    constexpr int EPL = WTraits<WT>::EPL;
    // x float4s held per thread across the prologue: 4 covers K <= 16*blockDim (every RMSNorm'd
    // input: K = n_embd), 16 covers K <= 64*blockDim (the FFN down projection, K = ffn length)
    constexpr int XN = NORM ? 4 : 16;
    extern __shared__ __attribute__((aligned(16))) float xs[];  // KC*64*EPL floats, then 16 for reductions, then 48 words of the fused ArgMax

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * nwaves + wid, tw = gridDim.x * nwaves;
#ifndef GEMV_INTERLEAVE
#define GEMV_INTERLEAVE 1
#endif
#if GEMV_INTERLEAVE
    // groups dealt round-robin over all waves of the launch: every wave (and with it every XCD) reads rows from all over the matrix
    const uint32_t total_groups = (p.NU + UPW - 1) / UPW;
    const uint32_t u_end = p.NU;
    // XD: the same, with the XCDs' shares taken from the calibration (an XCD that streams 10 % faster gets 10 % more rows): the
    // workgroups with blockIdx % 8 == x form one label (one XCD under round-robin placement — speed only, never correctness)
    const uint32_t xl = blockIdx.x & 7u;
    const uint32_t xn = XD ? __builtin_amdgcn_readfirstlane((uint32_t)p.xd.n[xl]) : 1u, xoff = XD ? __builtin_amdgcn_readfirstlane((uint32_t)p.xd.off[xl]) : 0u;
    const uint32_t xw = (gridDim.x >> 3) * nwaves, xlw = (blockIdx.x >> 3) * nwaves + wid;   // waves of the label, this wave among them
    uint32_t ngroups_;
    if constexpr (XD) {
        const uint32_t rounds = total_groups / p.xd.S, rem = total_groups % p.xd.S;
        const uint32_t slots = rounds * xn + (rem > xoff ? min(rem - xoff, xn) : 0u);
        ngroups_ = slots > xlw ? (slots - xlw + xw - 1) / xw : 0;
    } else {
        ngroups_ = total_groups > gw ? (total_groups - gw + tw - 1) / tw : 0;
    }
    const uint32_t ngroups = ngroups_;
    auto unit_at = [&](uint32_t g, uint32_t q) {
        if constexpr (XD) {
            const uint32_t slot = g * xw + xlw;
            return ((slot / xn) * p.xd.S + xoff + slot % xn) * UPW + q;
        } else {
            return (g * tw + gw) * UPW + q;
        }
    };
#else
    const uint32_t u_begin = (uint32_t)(((uint64_t)p.NU * gw) / tw);
    const uint32_t u_end = (uint32_t)(((uint64_t)p.NU * (gw + 1)) / tw);
    const uint32_t ngroups = (u_end - u_begin + UPW - 1) / UPW;
    auto unit_at = [&](uint32_t g, uint32_t q) { return u_begin + g * UPW + q; };
#endif
    const uint32_t cpg = p.KC / U;  // K-chunk groups per unit group
    const uint32_t nsteps = ngroups * cpg;
    const uint32_t kpad = p.KC * 64 * EPL;
    STAMP_DECL;
    STAMP(0);  // wave started

    // ---- prefetch-only launch (runs on a side stream while the PREVIOUS kernel of the token is still
    //      streaming): request exactly the bytes this kernel's waves will ask for first, with the
    //      default cache policy so they stay in L2 / Infinity Cache across the kernel boundary -------
    if (p.prefetch_only) {
        uint32_t acc = 0;
        for (uint32_t st = 0; st < min(nsteps, 2u); st++) {
            const uint32_t g = st / cpg, cg = st % cpg;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t u = min(min(unit_at(g, r / RPU), u_end - 1), p.NU - 1);
                const uint8_t *row = row_ptr<MODE>(p, u, r % RPU);
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const uint32_t k = min((cg * U + j) * (64 * EPL) + lane * EPL, p.K - EPL);
                    const u32x4 v = *reinterpret_cast<const GLOBAL_AS u32x4 *>((const GLOBAL_AS uint8_t *)row + (uint64_t)k * (WT == NFAI_F16 ? 2 : 4));
                    acc ^= v[0];
                }
            }
        }
        asm volatile("" ::"v"(acc));  // keep the loads; nothing is stored
        return;
    }

    // ---- (1) the activation loads go out FIRST: vmcnt retires in order, so the prologue below can
    //      wait for x while every weight load issued after it stays in flight --------------------
    f32x4 xv[XN], gv[NORM ? XN : 1];
#pragma unroll
    for (int i = 0; i < XN; i++) {
        // unconditional loads from a clamped address + select: a branch around each load would make
        // hipcc wait vmcnt(0) per element
        const uint32_t k = (threadIdx.x + i * blockDim.x) * 4;
        const uint32_t kk = min(k, p.K - 4);
        f32x4 v;
        if (BEGIN && p.begin.emb) {  // first launch of a token: x is the token's embedding row; workgroup 0 stores it for the residual
            v = embed_load4(p.begin.emb, p.begin.emb_type, p.begin.emb_rows, p.begin.tok[0], kk, p.K);
            if (blockIdx.x == 0 && k < p.K) *reinterpret_cast<f32x4 *>(p.begin.x_out + kk) = v;
        } else {
            v = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.x + kk);
        }
        xv[i] = k < p.K ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NORM) gv[i] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.gamma + kk);
    }

    __builtin_amdgcn_sched_barrier(0);  // the activation loads stay first in program order (vmcnt retires in order)
    // lane q < UPW finishes unit q of each group: what its FIRST epilogue reads (residual element /
    // cos,sin pair) and the position are requested now, with the activations
    float e0 = 0.f, e1 = 0.f;
    float *cs_lds = BEGIN ? xs + kpad + 16 + 48 : nullptr;  // first launch of a token: cos / sin of the position tabulated in LDS
    if constexpr (!BEGIN) epilogue_prefetch<MODE>(p, min(min(unit_at(0, min(lane, (uint32_t)UPW - 1)), u_end - 1), p.NU - 1), e0, e1);
    const uint32_t pos_v = (MODE == GEMV_QKV_ROPE) ? p.pos[0] : 0u;
    if constexpr (BEGIN) begin_bookkeeping(p.begin, pos_v, cs_lds);

    // ---- (2) weight loads of the first TWO steps (they do not depend on x) ----------------------
    u32x4 bufA[R][U], bufB[R][U];
    const uint8_t *rows[R];
    StepWalk iw;  // issue walker
    auto issue = [&](u32x4 (&buf)[R][U]) {
        if (iw.cg == 0) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t u = min(min(unit_at(iw.g, r / RPU), u_end - 1), p.NU - 1);
                rows[r] = row_ptr<MODE>(p, u, r % RPU);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) issue_loads<WT, U, GUARD>(buf[r], rows[r], iw.cg * U, lane, p.K);
        iw.next(cpg);
    };
    // unconditional (row indices are clamped to valid rows): a branch around the issue would make the
    // compiler's vmcnt bookkeeping take the worst path and wait for the weights before using x
    issue(bufA);
#ifdef GEMV_B_EARLY
    issue(bufB);
#endif
    STAMP(1);  // activation loads and the first weight step issued

    // ---- (3) prologue: x (optionally RMSNorm'd: RMSNormShader.cs:136-149) -> LDS ---------------
    {
        float *red = xs + kpad;
        float rms = 1.f;
        if constexpr (NORM) {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < XN; i++) {
                ss = fmaf(xv[i][0], xv[i][0], ss);
                ss = fmaf(xv[i][1], xv[i][1], ss);
                ss = fmaf(xv[i][2], xv[i][2], ss);
                ss = fmaf(xv[i][3], xv[i][3], ss);
            }
#ifdef NFAI_STAMPS
            asm volatile("" ::"v"(ss));
            STAMP(6);  // this wave's x has arrived (sum of squares of its own elements formed)
#endif
            ss = block_sum<true>(ss, red);
#ifdef NFAI_STAMPS
            asm volatile("" ::"v"(ss));
            STAMP(7);  // the workgroup's sum is known to this wave
#endif
            rms = sqrtf(ss / (float)p.K + p.eps);
        }
#pragma unroll
        for (int i = 0; i < XN; i++) {
            const uint32_t k = (threadIdx.x + i * blockDim.x) * 4;
            if (k < kpad) {
                f32x4 v = xv[i];
                if constexpr (NORM) {
                    if (k < p.K) {
                        v[0] = (v[0] / rms) * gv[i][0];
                        v[1] = (v[1] / rms) * gv[i][1];
                        v[2] = (v[2] / rms) * gv[i][2];
                        v[3] = (v[3] / rms) * gv[i][3];
                    }
                }
                *reinterpret_cast<f32x4 *>(xs + xs_index<EPL>(k)) = v;
            }
        }
        __syncthreads();
    }
    // The second step's requests go out AFTER the prologue: a wave issues in order, and pushing two steps into the CU's memory
    // queue takes 2.8-3.4 us (tools/stamps.py) during which x — long arrived — is not looked at.  With one step ahead of the
    // prologue the first FMA comes ~1 us earlier and the second step's requests overlap it (3B fp16: 637 -> 642 tokens/s, the
    // q|k|v launch 11.8 -> 11.3 us in eager timing; -DGEMV_B_EARLY restores the old order).
#ifndef GEMV_B_EARLY
    issue(bufB);
#endif
    if constexpr (BEGIN) epilogue_prefetch<MODE>(p, min(min(unit_at(0, min(lane, (uint32_t)UPW - 1)), u_end - 1), p.NU - 1), e0, e1, cs_lds);  // (table: barrier above)
    STAMP(2);  // x (normalised) is in LDS, second weight step issued

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.f;

    float best_v = -INFINITY;
    uint32_t best_i = 0xFFFFFFFFu;
    StepWalk cw;  // compute walker
    auto consume = [&](u32x4 (&buf)[R][U]) {
#pragma unroll
        for (int j = 0; j < U; j++) {
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = dot_chunk<WT>(buf[r][j], xs, cw.cg * U + j, lane, acc[r]);
        }
        if (cw.cg == cpg - 1) {
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]);
#pragma unroll
            for (int q = 0; q < UPW; q++) {
                const uint32_t u = unit_at(cw.g, q);
                // lane q finishes unit q (uniform values; spreads the stores over lanes)
                if (lane == (uint32_t)q && u < u_end) {
                    epilogue<MODE>(p, u, acc[q * RPU], RPU == 2 ? acc[q * RPU + RPU - 1] : 0.f, e0, e1, pos_v);
                    if constexpr (MODE == GEMV_PLAIN) {  // running best of the rows this lane has finished (lm_head + ArgMax)
                        if (topk_better(acc[q], u, best_v, best_i)) { best_v = acc[q]; best_i = u; }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = 0.f;
            // what the NEXT group's epilogue reads, a whole group ahead
            epilogue_prefetch<MODE>(p, min(min(unit_at(cw.g + 1, min(lane, (uint32_t)UPW - 1)), u_end - 1), p.NU - 1), e0, e1, cs_lds);
        }
        cw.next(cpg);
    };

    // ---- (4) ping-pong: consume A, refill A with step+2, consume B, refill B with step+3.  No
    //      register copies (a copy of an in-flight load's destination would wait for it) ----------
    //      Refills inside the loop are unconditional so every wait is an exact count; the last
    //      one to three steps are peeled.
    uint32_t st = 0;
#ifdef NFAI_STAMPS
    if (nsteps > 4) {  // first step on its own so that "first weights landed and multiplied" gets a stamp (same order of work)
        consume(bufA);
        STAMP(3);
        issue(bufA);
        consume(bufB);
        issue(bufB);
        st = 2;
    } else {
        STAMP(3);
    }
#endif
    for (; st + 3 < nsteps; st += 2) {
        consume(bufA);
        issue(bufA);
        consume(bufB);
        issue(bufB);
    }
    const uint32_t rem = nsteps - st;  // 0 (no work) .. 3
    if (rem == 3) {
        consume(bufA);
        issue(bufA);
        consume(bufB);
        consume(bufA);
    } else if (rem == 2) {
        consume(bufA);
        consume(bufB);
    } else if (rem == 1) {
        consume(bufA);
    }
    if constexpr (MODE == GEMV_PLAIN) {
        // SamplingUtils.ArgMax over the outputs in the same launch (block-uniform: a kernel argument)
        if (p.am.ticket != nullptr) argmax_fused_tail(best_v, best_i, p.am, reinterpret_cast<uint32_t *>(xs + kpad + 16));
    }
#ifdef NFAI_STAMPS
    STAMP(4);  // last FMA, reduction and epilogue stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(5);  // stores acknowledged
    STAMP_FLUSH(p.stamps, gw, 8);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same GEMV as MANY SHORT WORKGROUPS that split K over their waves (round 3).  k_gemv above keeps one or two long-running
// workgroups per CU whose waves walk whole rows behind a workgroup-wide activation prologue (x -> RMSNorm -> LDS -> barrier): the
// right shape for the 788 MB lm_head, but a per-block matrix of 19-100 MB pays that serial head and the slowest wave's tail once per
// launch with nothing to overlap them with.  Here:
//   workgroup = NW waves; wave w owns the K-slice [w * CH * 64 * EPL, +CH * 64 * EPL): ITS slice of x lives in registers — no LDS, no
//               barrier in front of the stream; the workgroup owns RW consecutive rows (RW / 2 row pairs for the modes that finish
//               rows in pairs); every wave requests its slice of all RW rows at once (RW * CH 16-byte loads per lane, straight to
//               registers, non-temporal), multiplies as they land, reduces each row over the wave (DPP) and the waves meet ONCE, in
//               LDS, at the end; thread t < RW / RPU then finishes unit t (same epilogues as k_gemv);
//   grid      = rows / RW workgroups (hundreds to thousands): several are resident per CU, so the head of one overlaps the tail of
//               another — which a persistent wave cannot do with itself;
//   RMSNorm   = (sum_k w_k * (x_k * g_k)) / rms: the gains are applied element by element as in RMSNormShader.cs:136-149, the
//               division by rms = sqrt(mean(x^2) + eps) — one scalar per vector — to the finished dot product, so no dot product
//               waits for the whole of x (each wave adds its slice's sum of squares in LDS; combined in wave order at the end).
//               The reordering moves two roundings of ~6e-8 relative per term; the tests' summation-order bound is 30x that.
// tools/gemv_sk_bench.hip (profiles/round3_gemv_sk_bench.txt), us per launch in a chain of cold launches, k_gemv -> this form:
// 3B q|k|v 7.9 -> 7.4, Wo 6.3 -> 4.9, gate|up 17.5 -> 17.3, Wdown 10.3 -> 9.9; 1B q|k|v 6.6 -> 4.5, gate|up 13.6 -> 12.1, Wdown 8.2 -> 7.3.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int SK_MAX_WAVES = 16;

template <int WT, int MODE, int CH, int RW, bool NORM>
__global__ __launch_bounds__(SK_MAX_WAVES * 64) void k_gemv_sk(const GemvParams p)
{
    constexpr int RPU = (MODE == GEMV_QKV_ROPE || MODE == GEMV_GATEUP) ? 2 : 1;
    constexpr int RWU = RW / RPU;                       // units per workgroup
    constexpr int EPL = WTraits<WT>::EPL;               // elements per 16-byte load: 8 (fp16) / 4 (fp32)
    constexpr int EB = (WT == NFAI_F16) ? 2 : 4;
    constexpr int XV = EPL / 4;                         // f32x4 of x per chunk and lane
    __shared__ float part[RW][SK_MAX_WAVES];
    __shared__ float ssq[SK_MAX_WAVES];
    __shared__ uint32_t am_lds[48];
    __shared__ float cs_tab[BEGIN_CS_WORDS];
    const uint32_t lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t unit0 = blockIdx.x * RWU;
    STAMP_DECL;
    STAMP(0);  // wave started
    const bool begin = MODE == GEMV_QKV_ROPE && NORM && p.begin.on;  // block-uniform
    // ---- this wave's slice of the activations (and gains) first: vmcnt retires in order
    f32x4 xa[CH][XV], ga[NORM ? CH : 1][XV];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t k = ((w * CH + c) * 64 + lane) * EPL;
#pragma unroll
        for (int h = 0; h < XV; h++) {
            if (begin && p.begin.emb) {  // the token's embedding row (first launch of a token); workgroup 0 stores it for the residual
                xa[c][h] = embed_load4(p.begin.emb, p.begin.emb_type, p.begin.emb_rows, p.begin.tok[0], k + 4 * h, p.K);
                if (blockIdx.x == 0) *reinterpret_cast<f32x4 *>(p.begin.x_out + k + 4 * h) = xa[c][h];
            } else {
                xa[c][h] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.x + k + 4 * h);
            }
            if constexpr (NORM) ga[c][h] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.gamma + k + 4 * h);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // what unit t's epilogue reads besides the sums (residual element / cos, sin / position): thread t, now
    float e0 = 0.f, e1 = 0.f;
    const uint32_t my_unit = min(unit0 + min(threadIdx.x, (uint32_t)RWU - 1), p.NU - 1);
    if (!begin) epilogue_prefetch<MODE>(p, my_unit, e0, e1);
    const uint32_t pos_v = (MODE == GEMV_QKV_ROPE) ? p.pos[0] : 0u;
    if (begin) begin_bookkeeping(p.begin, pos_v, cs_tab);  // cos / sin table of the position in LDS (+ global copy, epoch: workgroup 0)
    // ---- every weight request of this wave at once
    u32x4 wv[RW][CH];
#pragma unroll
    for (int r = 0; r < RW; r++) {
        const uint8_t *row = row_ptr<MODE>(p, min(unit0 + r / RPU, p.NU - 1), r % RPU);
#pragma unroll
        for (int c = 0; c < CH; c++) wv[r][c] = load_nt16(row + (uint64_t)(((w * CH + c) * 64 + lane) * EPL) * EB);
    }
    STAMP(1);  // activation and weight requests issued
    if constexpr (NORM) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int h = 0; h < XV; h++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    ss = fmaf(xa[c][h][e], xa[c][h][e], ss);
                    xa[c][h][e] = xa[c][h][e] * ga[c][h][e];  // RMSNormShader.cs:148: (x / rms) * g, the division deferred
                }
        ss = wave_sum(ss);
        if (lane == 0) ssq[w] = ss;
    }
    STAMP(2);  // this wave's x is in registers
#pragma unroll
    for (int r = 0; r < RW; r++) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if constexpr (WT == NFAI_F16) {
                acc = dot8_f16(wv[r][c], xa[c][0], xa[c][XV - 1], acc);
            } else {
                const f32x4 wf = __builtin_bit_cast(f32x4, wv[r][c]);
                acc = fmaf(wf[0], xa[c][0][0], acc);
                acc = fmaf(wf[1], xa[c][0][1], acc);
                acc = fmaf(wf[2], xa[c][0][2], acc);
                acc = fmaf(wf[3], xa[c][0][3], acc);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) part[r][w] = acc;
        if (r == 0) STAMP(3);  // first row's slice landed and multiplied
    }
    __syncthreads();
    // ---- thread t finishes unit t: the waves' partial sums in wave order, RMSNorm's division, the mode's epilogue
    float best_v = -INFINITY;
    uint32_t best_i = 0xFFFFFFFFu;
    if (threadIdx.x < (uint32_t)RWU && unit0 + threadIdx.x < p.NU) {
        const uint32_t t = threadIdx.x, unit = unit0 + t;
        float a[RPU];
#pragma unroll
        for (int q = 0; q < RPU; q++) {
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < SK_MAX_WAVES; i++) s4[i & 3] += (uint32_t)i < nw ? part[t * RPU + q][min((uint32_t)i, nw - 1)] : 0.f;
            a[q] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        if constexpr (NORM) {
            float t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < SK_MAX_WAVES; i++) t4[i & 3] += (uint32_t)i < nw ? ssq[min((uint32_t)i, nw - 1)] : 0.f;
            const float rms = sqrtf(((t4[0] + t4[1]) + (t4[2] + t4[3])) / (float)p.K + p.eps);  // RMSNormShader.cs:143-146
#pragma unroll
            for (int q = 0; q < RPU; q++) a[q] = a[q] / rms;
        }
        if (begin) epilogue_prefetch<MODE>(p, unit, e0, e1, cs_tab);
        epilogue<MODE>(p, unit, a[0], a[RPU - 1], e0, e1, pos_v);
        if constexpr (MODE == GEMV_PLAIN) { best_v = a[0]; best_i = unit; }
    }
    if constexpr (MODE == GEMV_PLAIN) {
        if (p.am.ticket != nullptr) argmax_fused_tail(best_v, best_i, p.am, am_lds);
    }
#ifdef NFAI_STAMPS
    STAMP(4);  // rows reduced, epilogue stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(5);
    STAMP_FLUSH(p.stamps, blockIdx.x * nw + w, 6);
#endif
}

struct SkPlan { bool ok; int ch, rw; uint32_t nw, grid; };

// shapes the split-K form takes: K = NW * CH chunks of one 16-byte load per lane, 2 <= NW <= 16, CH in {1, 2, 4}; everything else
// (ragged K, the lm_head-sized streams, the side-stream prefetch) stays with k_gemv
static SkPlan plan_gemv_sk(const GemvArgs &a, uint32_t NU, int epl, int rpu)
{
    SkPlan s{};
    static const int env_on = getenv("NFAI_GEMV_SK") ? atoi(getenv("NFAI_GEMV_SK")) : 1;
    static const int env_rw = getenv("NFAI_GEMV_SK_RW") ? atoi(getenv("NFAI_GEMV_SK_RW")) : 0;
    // bit per GemvMode.  Default: PLAIN and RESIDUAL only.  Measured inside the model (3B fp16, tokens/s; profiles/round3_gemv_sk_model.txt):
    // Wdown 672 against 670 with k_gemv (1B: 10.1 against 10.8 us per launch), but q|k|v 645 and gate|up 650 against 670 — with the
    // RMSNorm'd modes every one of the 1000-2000 workgroups fetches x and the gains again (as many bytes through a CU's memory path as
    // the weights of a q|k|v launch), which the stand-alone benchmark, with x hot in L2 and no epilogue, did not show.
    static const int env_modes = getenv("NFAI_GEMV_SK_MODES") ? atoi(getenv("NFAI_GEMV_SK_MODES")) : 3;
    if (!env_on || a.prefetch_only || !((env_modes >> (a.mode & 3)) & 1)) return s;
    const uint32_t ce = 64u * epl;
    if (a.K % ce) return s;
    const uint32_t kc = a.K / ce;
    for (int ch : {1, 2, 4}) {
        if (kc % ch == 0 && kc / ch >= 2 && kc / ch <= (uint32_t)SK_MAX_WAVES) { s.ch = ch; s.nw = kc / ch; break; }
    }
    if (!s.ch) return s;
    const uint64_t rows = (uint64_t)NU * rpu, bytes = rows * a.K * (epl == 8 ? 2 : 4);
    if (bytes > (256ull << 20)) return s;  // the lm_head: a long stream is what the persistent form is good at (6.7 TB/s)
    // rows per workgroup: 4 for the short matrices (more, smaller workgroups: 3B q|k|v 7.4 us against 7.7 / 9.1 with 8 / 16), 16
    // loads per lane for the long ones (3B gate|up 17.3 against 18.2 with 4)
    int rw = bytes >= (60ull << 20) ? 16 / s.ch : 4;
    if (env_rw == 4 || env_rw == 8 || env_rw == 16) rw = env_rw;
    while (rw * s.ch > 16) rw /= 2;
    if (rw < 4) rw = 4;
    if (rw * s.ch > 16) return s;
    s.rw = rw;
    s.grid = (NU + rw / rpu - 1) / (rw / rpu);
    if (a.argmax_part && s.grid > ARGMAX_FUSED_MAX_BLOCKS) return s;  // the fused ArgMax keeps one partial per workgroup
    s.ok = true;
    return s;
}

template <int WT, int MODE, int CH, int RW>
static hipError_t launch_sk(const GemvParams &p, const SkPlan &sp, hipStream_t s)
{
    if (p.gamma != nullptr) hipLaunchKernelGGL((k_gemv_sk<WT, MODE, CH, RW, true>), dim3(sp.grid), dim3(sp.nw * 64), 0, s, p);
    else hipLaunchKernelGGL((k_gemv_sk<WT, MODE, CH, RW, false>), dim3(sp.grid), dim3(sp.nw * 64), 0, s, p);
    return hipGetLastError();
}

template <int WT, int MODE>
static hipError_t dispatch_sk(const GemvParams &p, const SkPlan &sp, hipStream_t s)
{
#define NFAI_SK(CH_, RW_) if (sp.ch == CH_ && sp.rw == RW_) return launch_sk<WT, MODE, CH_, RW_>(p, sp, s);
    NFAI_SK(1, 4) NFAI_SK(1, 8) NFAI_SK(1, 16) NFAI_SK(2, 4) NFAI_SK(2, 8) NFAI_SK(4, 4)
#undef NFAI_SK
    return hipErrorInvalidValue;
}

template <int WT>
static hipError_t dispatch_sk_mode(const GemvParams &p, const SkPlan &sp, int mode, hipStream_t s)
{
    switch (mode) {
        case GEMV_PLAIN: return dispatch_sk<WT, GEMV_PLAIN>(p, sp, s);
        case GEMV_RESIDUAL: return dispatch_sk<WT, GEMV_RESIDUAL>(p, sp, s);
        case GEMV_QKV_ROPE: return dispatch_sk<WT, GEMV_QKV_ROPE>(p, sp, s);
        case GEMV_GATEUP: return dispatch_sk<WT, GEMV_GATEUP>(p, sp, s);
    }
    return hipErrorInvalidValue;
}

// ---- host side: shape checks and the (waves per block, units per wave step, K unroll) choice ----
struct GemvPlan {
    int upw, u;
    bool guard, ok;
    uint32_t grid, block, lds_bytes;
};

static GemvPlan plan_gemv(uint32_t NU, uint32_t K, int epl, int rpu, uint32_t n_cu, bool norm, int mode)
{
    GemvPlan pl{};
    pl.ok = true;
    const uint32_t ce = 64u * epl;
    pl.guard = (K % ce) != 0;
    const uint32_t kc = (K + ce - 1) / ce;
    // K unroll: 1-KiB loads per row per step
    pl.u = pl.guard ? 1 : (kc % 4 == 0 ? 4 : (kc % 3 == 0 ? 3 : (kc % 2 == 0 ? 2 : 1)));
    // waves per block: prefer an exact split of the units over n_cu * wpb waves
    uint32_t wpb = 4;
    const uint32_t cands[] = {4, 8, 6, 5, 7};
    bool exact = false;
    for (uint32_t c : cands) {
        if (NU % (n_cu * c) == 0) { wpb = c; exact = true; break; }
    }
    uint32_t grid = n_cu;
    if (!exact && NU < n_cu * 4) { grid = (NU + 3) / 4; wpb = 4; }  // tiny problems: fewer blocks
    if (grid == 0) grid = 1;
    // tuning overrides (bench sweeps only): blocks per CU and waves per block
    static const int env_bpc = getenv("NFAI_GEMV_BPC") ? atoi(getenv("NFAI_GEMV_BPC")) : 0;
    static const int env_wpb = getenv("NFAI_GEMV_WPB") ? atoi(getenv("NFAI_GEMV_WPB")) : 0;
    static const int env_maxld = getenv("NFAI_GEMV_MAXLD") ? atoi(getenv("NFAI_GEMV_MAXLD")) : 8;  // 3B fp16 tokens/s: 4: 664, 6: 667, 8: 666, 12: 654, 16: 658, 24: 644
    static const int env_wpb_mode[4] = {getenv("NFAI_GEMV_WPB_PLAIN") ? atoi(getenv("NFAI_GEMV_WPB_PLAIN")) : 0,
                                        getenv("NFAI_GEMV_WPB_RES") ? atoi(getenv("NFAI_GEMV_WPB_RES")) : 0,
                                        getenv("NFAI_GEMV_WPB_QKV") ? atoi(getenv("NFAI_GEMV_WPB_QKV")) : 0,
                                        getenv("NFAI_GEMV_WPB_GATEUP") ? atoi(getenv("NFAI_GEMV_WPB_GATEUP")) : 0};
    const int wpb_req = env_wpb ? env_wpb : env_wpb_mode[mode & 3];
    if (wpb_req >= 1 && wpb_req <= 8 && NU >= n_cu * 8) wpb = (uint32_t)wpb_req;
    // workgroups per CU by epilogue mode (NFAI_GEMV_BPC_<MODE> overrides one mode, NFAI_GEMV_BPC all)
    static const int env_bpc_mode[4] = {getenv("NFAI_GEMV_BPC_PLAIN") ? atoi(getenv("NFAI_GEMV_BPC_PLAIN")) : 0,
                                        getenv("NFAI_GEMV_BPC_RES") ? atoi(getenv("NFAI_GEMV_BPC_RES")) : 0,
                                        getenv("NFAI_GEMV_BPC_QKV") ? atoi(getenv("NFAI_GEMV_BPC_QKV")) : 0,
                                        getenv("NFAI_GEMV_BPC_GATEUP") ? atoi(getenv("NFAI_GEMV_BPC_GATEUP")) : 0};
    static const int def_bpc_mode[4] = {1, 1, 2, 2};  // measured at 3B fp16 (tokens/s): all 1: 642; q|k|v 2: 651; + gate|up 2: 653.5; q|k|v 3: 610; Wdown 2: slower
    int bpc = env_bpc ? env_bpc : (env_bpc_mode[mode & 3] ? env_bpc_mode[mode & 3] : def_bpc_mode[mode & 3]);
    if (bpc >= 1 && bpc <= 4 && NU >= n_cu * 8) grid = n_cu * (uint32_t)bpc;
    const uint32_t upw_total = (NU + grid * wpb - 1) / (grid * wpb);  // units per wave (max)
    // units per step: rows_in_flight * u <= 16 loads per lane per step (x2 for the register double
    // buffer = 128 VGPRs), and divide upw_total if we can
    int upw = 1;
    for (int c = 4; c >= 1; c--) {
        if (c * rpu * pl.u <= env_maxld && upw_total % c == 0) { upw = c; break; }
    }
    // the block must hold x in XN float4 per thread (XN = 4 with RMSNorm, 16 without)
    const uint32_t xn = norm ? 4 : 16;
    while (wpb < 8 && (uint64_t)xn * wpb * 64 * 4 < (uint64_t)kc * ce) wpb++;
    if ((uint64_t)xn * wpb * 64 * 4 < (uint64_t)kc * ce) pl.ok = false;
    pl.upw = upw;
    pl.grid = grid;
    pl.block = wpb * 64;
    pl.lds_bytes = (kc * ce + 16 + 48 + BEGIN_CS_WORDS) * 4;  // x | reduction | fused ArgMax | cos/sin table of a token's first launch
    return pl;
}

template <int WT, int MODE, int UPW, int U, bool GUARD>
static hipError_t launch_one(const GemvParams &p, const GemvPlan &pl, hipStream_t s)
{
    if constexpr (MODE == GEMV_QKV_ROPE) {
        if (p.gamma != nullptr && p.begin.on) {  // the token's first launch: its own instantiation, nothing of it in the other 27
            hipLaunchKernelGGL((k_gemv<WT, MODE, UPW, U, GUARD, true, true>), dim3(pl.grid), dim3(pl.block), pl.lds_bytes, s, p);
            return hipGetLastError();
        }
    }
    if constexpr (!GUARD && WT == NFAI_F16 && (MODE == GEMV_PLAIN || MODE == GEMV_GATEUP)) {
        if (p.xd.S != 0) {  // rows dealt to the XCDs by calibrated share (the lm_head, gate | up): own instantiations
            if (p.gamma != nullptr)
                hipLaunchKernelGGL((k_gemv<WT, MODE, UPW, U, GUARD, true, false, true>), dim3(pl.grid), dim3(pl.block), pl.lds_bytes, s, p);
            else
                hipLaunchKernelGGL((k_gemv<WT, MODE, UPW, U, GUARD, false, false, true>), dim3(pl.grid), dim3(pl.block), pl.lds_bytes, s, p);
            return hipGetLastError();
        }
    }
    if (p.gamma != nullptr)
        hipLaunchKernelGGL((k_gemv<WT, MODE, UPW, U, GUARD, true>), dim3(pl.grid), dim3(pl.block), pl.lds_bytes, s, p);
    else
        hipLaunchKernelGGL((k_gemv<WT, MODE, UPW, U, GUARD, false>), dim3(pl.grid), dim3(pl.block), pl.lds_bytes, s, p);
    return hipGetLastError();
}

template <int WT, int MODE>
static hipError_t dispatch_plan(const GemvParams &p, const GemvPlan &pl, hipStream_t s)
{
    if (pl.guard) {
        switch (pl.upw) {
            case 1: return launch_one<WT, MODE, 1, 1, true>(p, pl, s);
            case 2: return launch_one<WT, MODE, 2, 1, true>(p, pl, s);
            case 3: return launch_one<WT, MODE, 3, 1, true>(p, pl, s);
            default: return launch_one<WT, MODE, 4, 1, true>(p, pl, s);
        }
    }
#define NFAI_CASE(UPW_, U_) \
    if (pl.upw == UPW_ && pl.u == U_) return launch_one<WT, MODE, UPW_, U_, false>(p, pl, s);
    NFAI_CASE(1, 1) NFAI_CASE(1, 2) NFAI_CASE(1, 3) NFAI_CASE(1, 4)
    NFAI_CASE(2, 1) NFAI_CASE(2, 2) NFAI_CASE(2, 3) NFAI_CASE(2, 4)
    NFAI_CASE(3, 1) NFAI_CASE(3, 2) NFAI_CASE(3, 3) NFAI_CASE(3, 4)
    NFAI_CASE(4, 1) NFAI_CASE(4, 2) NFAI_CASE(4, 3) NFAI_CASE(4, 4)
#undef NFAI_CASE
    return hipErrorInvalidValue;
}

template <int WT>
static hipError_t dispatch_mode(const GemvParams &p, const GemvPlan &pl, int mode, hipStream_t s)
{
    switch (mode) {
        case GEMV_PLAIN: return dispatch_plan<WT, GEMV_PLAIN>(p, pl, s);
        case GEMV_RESIDUAL: return dispatch_plan<WT, GEMV_RESIDUAL>(p, pl, s);
        case GEMV_QKV_ROPE: return dispatch_plan<WT, GEMV_QKV_ROPE>(p, pl, s);
        case GEMV_GATEUP: return dispatch_plan<WT, GEMV_GATEUP>(p, pl, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_gemv(const GemvArgs &a, hipStream_t s)
{
    if (a.w_type == NFAI_Q4_K_T16 || a.w_type == NFAI_Q6_K_T16 || a.w_type == NFAI_KQ_MIXED) return launch_gemv_kqm(a, s);
    if (a.w_type == NFAI_Q4_K || a.w_type == NFAI_Q6_K) return launch_gemv_kq(a, s);
    GemvParams p{};
    const int rpu = (a.mode == GEMV_QKV_ROPE || a.mode == GEMV_GATEUP) ? 2 : 1;
    const int epl = a.w_type == NFAI_F16 ? 8 : 4;
    if (a.w_type != NFAI_F16 && a.w_type != NFAI_F32) return hipErrorInvalidValue;
    if (a.K == 0 || a.K % 8 != 0) return hipErrorInvalidValue;
    uint32_t total_rows = 0;
    for (int i = 0; i < 3; i++) {
        p.W[i] = reinterpret_cast<const uint8_t *>(a.W[i]);
        total_rows += a.seg_rows[i];
    }
    p.seg_end[0] = a.seg_rows[0];
    p.seg_end[1] = a.seg_rows[0] + a.seg_rows[1];
    p.seg_end[2] = total_rows;
    if (a.mode == GEMV_QKV_ROPE) {
        // rotation pairs must not straddle a segment or a head
        if ((a.seg_rows[0] | a.seg_rows[1] | a.seg_rows[2] | a.D) & 1u) return hipErrorInvalidValue;
        p.NU = total_rows / 2;
    } else if (a.mode == GEMV_GATEUP) {
        if (a.seg_rows[0] != a.seg_rows[1]) return hipErrorInvalidValue;
        p.NU = a.seg_rows[0];
    } else {
        p.NU = a.seg_rows[0];
    }
    if (p.NU == 0) return hipSuccess;
    p.row_bytes = (uint64_t)a.K * (a.w_type == NFAI_F16 ? 2 : 4);
    p.x = a.x;
    p.gamma = a.gamma;
    p.eps = a.eps;
    p.K = a.K;
    p.y = a.y;
    p.res = a.res;
    p.kc = a.kcache;
    p.vc = a.vcache;
    p.kv_pos_stride = a.kv_pos_stride;
    p.kv_head_stride = a.kv_head_stride;
    p.rope_cs = a.rope_cs;
    p.rope_dims = a.rope_dims;
    p.D = a.D;
    p.pos = a.pos_dev;
    p.kv_f16 = a.kv_type == NFAI_F16;
    p.prefetch_only = a.prefetch_only ? 1u : 0u;
    const SkPlan sp = plan_gemv_sk(a, p.NU, epl, rpu);
    if (a.begin.on) {
        if (a.mode != GEMV_QKV_ROPE || !a.gamma || !a.begin.freqs || !a.begin.cs_out || (a.begin.emb && (!a.begin.tok || !a.begin.x_out)) ||
            a.begin.n_freq * 2 > BEGIN_CS_WORDS)
            return hipErrorInvalidValue;  // (gemv_begin_ok() says which launches take it)
        p.begin.on = 1; p.begin.emb = static_cast<const uint8_t *>(a.begin.emb); p.begin.emb_type = a.begin.emb_type; p.begin.emb_rows = a.begin.emb_rows;
        p.begin.tok = a.begin.tok; p.begin.x_out = a.begin.x_out; p.begin.freqs = a.begin.freqs; p.begin.cs_out = a.begin.cs_out;
        p.begin.n_freq = a.begin.n_freq; p.begin.epoch = a.begin.epoch;
    }
    if (a.argmax_part) {
        if (a.mode != GEMV_PLAIN || !a.argmax_out) return hipErrorInvalidValue;
        p.am.part_v = static_cast<float *>(a.argmax_part);
        p.am.part_i = reinterpret_cast<uint32_t *>(p.am.part_v + ARGMAX_FUSED_MAX_BLOCKS);
        p.am.ticket = p.am.part_i + ARGMAX_FUSED_MAX_BLOCKS;
        p.am.out_idx = a.argmax_out; p.am.pos_inc = a.argmax_pos_inc; p.am.ring = a.argmax_ring; p.am.ring_len = a.argmax_ring_len;
    }
    if (sp.ok) {
        p.KC = a.K / (64 * epl);
        if (a.argmax_part && sp.grid > ARGMAX_FUSED_MAX_BLOCKS) return hipErrorInvalidValue;
        static const char *sk_names[] = {"gemv_sk_plain", "gemv_sk_residual", "gemv_sk_qkv_rope", "gemv_sk_gateup"};
        NFAI_STAMP_SET(p, sk_names[a.mode & 3], sp.grid, sp.nw * 64);
        if (a.w_type == NFAI_F16) return dispatch_sk_mode<NFAI_F16>(p, sp, a.mode, s);
        return dispatch_sk_mode<NFAI_F32>(p, sp, a.mode, s);
    }
    const GemvPlan pl = plan_gemv(p.NU, a.K, epl, rpu, a.n_cu, a.gamma != nullptr, a.mode);
    if (!pl.ok || (a.argmax_part && pl.grid > ARGMAX_FUSED_MAX_BLOCKS)) return hipErrorInvalidValue;
    p.KC = (a.K + 64 * epl - 1) / (64 * epl);
    if (pl.lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    p.xd.S = 0;
    if (a.xcd_shares && !pl.guard && !a.prefetch_only && a.w_type == NFAI_F16 && (a.mode == GEMV_PLAIN || a.mode == GEMV_GATEUP) && pl.grid % 8 == 0 &&
        (p.NU + pl.upw - 1) / pl.upw >= 8u * pl.grid * (pl.block / 64)) {   // >= 8 groups per wave: a share moves a fraction of a wave's work (at
                                                                           // gate | up's 4 groups per wave a share is a whole group: measured slower)
        uint32_t off = 0;
        bool uniform = true, sane = true;
        for (int x = 0; x < 8; x++) {
            p.xd.n[x] = a.xcd_shares[x];
            p.xd.off[x] = (uint16_t)off;
            off += a.xcd_shares[x];
            uniform = uniform && a.xcd_shares[x] == a.xcd_shares[0];
            sane = sane && a.xcd_shares[x] >= 1 && a.xcd_shares[x] <= 1024;
        }
        if (sane && !uniform) p.xd.S = off;   // equal shares: the plain dealing is the same partition, one instantiation fewer in flight
    }
    {
        static const char *names[] = {"gemv_plain", "gemv_residual", "gemv_qkv_rope", "gemv_gateup"};
        NFAI_STAMP_SET(p, names[a.mode & 3], pl.grid, pl.block);
    }
    if (a.w_type == NFAI_F16) return dispatch_mode<NFAI_F16>(p, pl, a.mode, s);
    return dispatch_mode<NFAI_F32>(p, pl, a.mode, s);
}

// ---- per-XCD streaming shares ----------------------------------------------------------------------------------------------------
// A weight-streaming launch ends with its slowest XCD, and the XCDs of this part do not stream equally fast: with equal shares the
// workgroups of some blockIdx % 8 labels finish a gate | up launch 5-9 % later than others, an lm_head launch up to 20 % later
// (tools/stamps.py, profiles/round3_stamps_f16.json).  The probe is the same access pattern without the arithmetic: every
// wave streams an equal number of rows of a buffer far larger than the Infinity Cache with the GEMV's 16-byte non-temporal loads,
// every workgroup stamps the moment it is done; a label's share is then proportional to 1 / (its workgroups' mean time).
__global__ __launch_bounds__(512) void k_xcd_probe(const uint8_t *buf, uint64_t bytes_per_wg, unsigned long long *t_begin, unsigned long long *t_end,
                                                    uint32_t *xcc, uint32_t *sink)
{
    // rows of 6 KiB dealt round-robin over all waves of the launch, as k_gemv deals its row groups: every wave — and every XCD —
    // reads from all over the buffer, so a label's time says how fast its XCD streams, not where its share of the buffer lives
    constexpr uint32_t ROW = 6144, PIECES = ROW / 1024;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t lane = threadIdx.x & 63, nwaves = blockDim.x >> 6, gw = blockIdx.x * nwaves + (threadIdx.x >> 6), tw = gridDim.x * nwaves;
    const uint64_t rows = bytes_per_wg * gridDim.x / ROW;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (uint64_t r = gw; r + tw < rows; r += 2ull * tw) {   // two rows (12 loads per lane) in flight
        u32x4 v[2 * PIECES];
#pragma unroll
        for (int j = 0; j < (int)PIECES; j++) {
            v[j] = load_nt16(buf + r * ROW + j * 1024 + lane * 16);
            v[PIECES + j] = load_nt16(buf + (r + tw) * ROW + j * 1024 + lane * 16);
        }
#pragma unroll
        for (int j = 0; j < 2 * (int)PIECES; j++) acc ^= v[j];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u) sink[0] = 1;  // keeps the loads
    __syncthreads();
    if (threadIdx.x == 0) {
        t_begin[blockIdx.x] = t0;
        t_end[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id & 0xFu;
    }
}

const uint16_t *gemv_xcd_calibrate(Ctx *c)
{
    if (c->xcd_state == 1) return c->xcd_shares;
    if (c->xcd_state == 2) return nullptr;
    c->xcd_state = 2;
    if (const char *env = getenv("NFAI_XCD_DEAL")) {
        if (atoi(env) == 0) return nullptr;
    }
    const uint32_t n_cu = (uint32_t)c->prop.multiProcessorCount;
    if (n_cu % 8 != 0) return nullptr;
    const uint64_t per_wg = 3ull << 20, total = per_wg * n_cu;   // 768 MB on 256 CUs: three times the Infinity Cache, ~120 us per pass
    uint8_t *buf = nullptr;
    unsigned long long *tb = nullptr;
    if (hipMalloc(&buf, total) != hipSuccess || hipMalloc(&tb, (size_t)n_cu * 24 + 64) != hipSuccess) {
        (void)hipGetLastError();
        if (buf) hipFree(buf);
        return nullptr;
    }
    unsigned long long *te = tb + n_cu;
    uint32_t *xcc = reinterpret_cast<uint32_t *>(te + n_cu), *sink = xcc + n_cu;
    std::vector<unsigned long long> hb(n_cu), he(n_cu);
    std::vector<uint32_t> hx(n_cu);
    double sum_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int passes = 6;   // the first one warms up (page tables, clocks) and is dropped
    bool ok = true;
    for (int it = 0; it < passes && ok; it++) {
        k_xcd_probe<<<n_cu, 512, 0, c->stream>>>(buf, per_wg, tb, te, xcc, sink);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
             hipMemcpy(hb.data(), tb, n_cu * 8, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(he.data(), te, n_cu * 8, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(hx.data(), xcc, n_cu * 4, hipMemcpyDeviceToHost) == hipSuccess;
        if (!ok || it == 0) continue;
        unsigned long long first = ~0ull;
        for (uint32_t b = 0; b < n_cu; b++) first = std::min(first, hb[b]);
        for (uint32_t b = 0; b < n_cu; b++) sum_us[b & 7] += (double)(he[b] - first) * 0.01;   // s_memrealtime: 100 MHz
    }
    hipFree(buf);
    hipFree(tb);
    if (!ok) { (void)hipGetLastError(); return nullptr; }
    // a label must be one XCD for its share to mean anything: every workgroup of a label reports the same XCC id, the labels differ
    uint32_t label_xcc[8], seen = 0;
    for (uint32_t b = 0; b < n_cu; b++) {
        if (b < 8) { label_xcc[b] = hx[b]; seen |= 1u << hx[b]; }
        else if (hx[b] != label_xcc[b & 7]) return nullptr;
    }
    if (seen != 0xFFu) return nullptr;
    double rate[8], mean = 0.0;
    for (int x = 0; x < 8; x++) {
        const double us = sum_us[x] / ((passes - 1) * (n_cu / 8));
        if (!(us > 1.0)) return nullptr;
        c->xcd_probe_us[x] = (float)us;
        rate[x] = 1.0 / us;
        mean += rate[x] / 8;
    }
    static const int base = getenv("NFAI_XCD_BASE") ? atoi(getenv("NFAI_XCD_BASE")) : 40;   // groups per dealing round and label at the mean rate: 2.5 % steps
    // 1: shares proportional to the probe's rates.  Measured on the lm_head launch (3B fp16, V = 128,256; us per launch, two alternations on
    // one box, tools/ab_xcd.sh): equal shares 115.4 / 116.3; gain 1: 117.1 / 115.2; 1.5: 113.0 / 114.1; 2: 111.9 / 113.8; 3: 115.2 / 117.3 —
    // the GEMV's waves lose a little more on a slow XCD than the bare stream of the probe does
    static const double gain = getenv("NFAI_XCD_GAIN") ? atof(getenv("NFAI_XCD_GAIN")) : 1.75;
    for (int x = 0; x < 8; x++) {
        const double r = 1.0 + gain * (rate[x] / mean - 1.0);
        const long n = lround(base * std::min(1.25, std::max(0.75, r)));
        c->xcd_shares[x] = (uint16_t)std::max(1L, n);
    }
    c->xcd_state = 1;
    return c->xcd_shares;
}

// does the first q|k|v launch described by `a` take the per-token prologue (GemvArgs::Begin)?  The fp16 / fp32 GEMV kernels and the
// int8-MFMA K-quant kernel do; the VALU K-quant fall-back does not (the model then launches k_token_begin).
bool gemv_begin_ok(const GemvArgs &a)
{
    if (a.mode != GEMV_QKV_ROPE || !a.gamma) return false;
    if (a.w_type == NFAI_Q4_K_T16 || a.w_type == NFAI_Q6_K_T16 || a.w_type == NFAI_KQ_MIXED) return a.K % 256 == 0;
    return a.w_type == NFAI_F16 || a.w_type == NFAI_F32;  // both forms of the fp16 / fp32 GEMV carry it
}

}  // namespace nfai
