// kernels_gemv_kq.hip — GEMV on ggml K-quant weights (Q4_K, Q6_K): the decode path of the
// Q4_K_M configurations of BASELINE.json.  The reference cannot load these types at all
// (NFAI.GGUF/Parser.cs:111-114 throws "Unsupported data type"); the op is the same
// MatrixMultiplyShader (MatrixMultiplyShader.cs:255-289) on dequantised weights, with the same
// fused prologues/epilogues as kernels_gemv.hip.  Block layouts: ggml-common.h (restated in
// oracle/nfai_oracle.c; parity unpinned by the reference).
//
// Bound: HBM — 4.5 (Q4_K) / 6.5625 (Q6_K) bits per weight are streamed once per token, blocks go
// HBM -> VGPR and are dequantised in registers; LDS holds the activations (and their 16-sums).
//
// Lane mapping (both formats): 8 lanes share one 256-weight super-block, each lane owns 32 weights;
// a wave-instruction therefore covers 8 super-blocks = 2048 weights of one row ("chunk").
//   Q4_K (144 B, native layout, 16-byte aligned): lane i of a block loads the 16-byte header
//        (d, dmin, 12 scale bytes — same address for the 8 lanes, one fetch) and qs[16i..16i+16):
//        low nibbles = 16 weights of sub-block 2j, high nibbles = 16 weights of sub-block 2j+1
//        (j = i/2).  sum_k w*x = d*sc*(sum q*x) - dmin*m*(sum x): the 16-sums of x are precomputed
//        once per launch in LDS and shared by all rows.
//   Q6_K (210 B: NOT 16-byte aligned natively) is repacked at load into four planes
//        ql[N][nb][128] | qh[N][nb][64] | sc[N][nb][16] | d[N][nb] (same bytes, every access
//        naturally aligned): lane (n = half, t = 8-weight column group) loads 8 B of ql twice,
//        8 B of qh, the 8 scale bytes of its half and d.
// The activations are staged in LDS in the order the lanes consume them, so that every
// ds_read_b128 of a wave covers 64 consecutive 16-byte slots (conflict-free).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace nfai {

#define GLOBAL_AS __attribute__((address_space(1)))

struct KqParams {
    const uint8_t *W[3];       // per segment: Q4_K native rows / Q6_K plane base
    uint32_t seg_end[3];
    uint32_t seg_rows[3];      // rows of each segment (plane strides for Q6_K)
    const float *x;
    const float *gamma;
    float eps;
    uint32_t K, NB, KC, NU;    // NB = super-blocks per row, KC = ceil(NB / 8) chunks
    float *y;
    const float *res;
    void *kc, *vc;
    uint64_t kv_pos_stride, kv_head_stride;
    const float *rope_cs;
    uint32_t rope_dims, D;
    const uint32_t *pos;
    int kv_f16;
};

// ---- activation layout in LDS ------------------------------------------------------------------
// Q4_K: k = blk*256 + j*64 + hi*32 + half*16 + t*4 + e   (blk = 8c + s)
//       slot16 = ((((c*2 + hi)*4 + t)*8 + s)*4 + j)*2 + half        -> lane s*8 + j*2 + half
// Q6_K: k = blk*256 + n*128 + quarter*32 + t4*8 + p*4 + e
//       slot16 = ((((c*4 + quarter)*2 + p)*8 + s)*2 + n)*4 + t4      -> lane s*8 + n*4 + t4
template <int QT> __device__ __forceinline__ uint32_t kq_xs_index(uint32_t k)
{
    const uint32_t blk = k >> 8, c = blk >> 3, s = blk & 7, e = k & 3;
    if constexpr (QT == NFAI_Q4_K) {
        const uint32_t j = (k >> 6) & 3, hi = (k >> 5) & 1, half = (k >> 4) & 1, t = (k >> 2) & 3;
        return ((((((c * 2 + hi) * 4 + t) * 8 + s) * 4 + j) * 2 + half) << 2) + e;
    } else {
        const uint32_t n = (k >> 7) & 1, quarter = (k >> 5) & 3, t4 = (k >> 3) & 3, p = (k >> 2) & 1;
        return ((((((c * 4 + quarter) * 2 + p) * 8 + s) * 2 + n) * 4 + t4) << 2) + e;
    }
}

// unsigned byte n of a dword -> float in ONE instruction (hipcc otherwise emits v_bfe_u32 +
// v_cvt_f32_ubyte0 for the shift-and-mask pattern: +1 VALU op per weight in a VALU-heavy kernel)
__device__ __forceinline__ float ub0(uint32_t v) { float r; asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ float ub1(uint32_t v) { float r; asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ float ub2(uint32_t v) { float r; asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ float ub3(uint32_t v) { float r; asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(v)); return r; }

// 4 byte-weights x 4 activations, two packed fp32 FMAs (v_pk_fma_f32); acc holds two partial sums
__device__ __forceinline__ f32x2 dot4_u8(uint32_t q, f32x4 x, f32x2 acc)
{
    acc = __builtin_elementwise_fma(f32x2{ub0(q), ub1(q)}, f32x2{x[0], x[1]}, acc);
    acc = __builtin_elementwise_fma(f32x2{ub2(q), ub3(q)}, f32x2{x[2], x[3]}, acc);
    return acc;
}

// ---- per-lane block data ---------------------------------------------------------------------------
struct Q4Regs { u32x4 hdr, qs; };
struct Q6Regs { u32x2 qla, qlb, qh, sc; uint32_t d; };

__device__ __forceinline__ uint32_t byte_of(u32x4 v, uint32_t n)  // byte n (0..15) of a 16-byte register, n per lane
{
    const uint32_t w = n >> 2;
    const uint32_t d = w == 0 ? v[0] : (w == 1 ? v[1] : (w == 2 ? v[2] : v[3]));
    return (d >> ((n & 3) * 8)) & 0xFFu;
}

// row pointer helpers: segment lookup as in kernels_gemv.hip
template <int MODE>
__device__ __forceinline__ void kq_row(const KqParams &p, uint32_t unit, int sub, uint32_t &seg, uint32_t &row)
{
    if constexpr (MODE == GEMV_GATEUP) {
        seg = sub; row = unit;
    } else if constexpr (MODE == GEMV_QKV_ROPE) {
        const uint32_t r = unit * 2 + sub;
        if (r < p.seg_end[0]) { seg = 0; row = r; }
        else if (r < p.seg_end[1]) { seg = 1; row = r - p.seg_end[0]; }
        else { seg = 2; row = r - p.seg_end[1]; }
    } else {
        seg = 0; row = unit;
    }
}

__device__ __forceinline__ Q4Regs q4_load(const KqParams &p, uint32_t seg, uint32_t row, uint32_t blk, uint32_t i)
{
    const uint8_t *b = p.W[seg] + ((uint64_t)row * p.NB + blk) * 144;
    Q4Regs r;
    r.hdr = *reinterpret_cast<const GLOBAL_AS u32x4 *>((const GLOBAL_AS uint8_t *)b);
    r.qs = load_nt16(b + 16 + i * 16);
    return r;
}

__device__ __forceinline__ Q6Regs q6_load(const KqParams &p, uint32_t seg, uint32_t row, uint32_t blk, uint32_t n, uint32_t t4)
{
    const uint64_t nblk = (uint64_t)p.seg_rows[seg] * p.NB, bi = (uint64_t)row * p.NB + blk;
    const GLOBAL_AS uint8_t *base = (const GLOBAL_AS uint8_t *)p.W[seg];
    const GLOBAL_AS uint8_t *ql = base + bi * 128 + n * 64 + t4 * 8;
    const GLOBAL_AS uint8_t *qh = base + nblk * 128 + bi * 64 + n * 32 + t4 * 8;
    const GLOBAL_AS uint8_t *sc = base + nblk * 192 + bi * 16 + n * 8;
    const GLOBAL_AS uint16_t *d = reinterpret_cast<const GLOBAL_AS uint16_t *>(base + nblk * 208) + bi;
    Q6Regs r;
    r.qla = __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS u32x2 *>(ql));
    r.qlb = __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS u32x2 *>(ql + 32));
    r.qh = __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS u32x2 *>(qh));
    r.sc = *reinterpret_cast<const GLOBAL_AS u32x2 *>(sc);
    r.d = *d;
    return r;
}

// dot of this lane's 32 Q4_K weights with x; xs = chunk base in LDS, sx = 16-sums [hi][lane]
__device__ __forceinline__ float q4_dot(const Q4Regs &r, const float *xs, const float *sx, uint32_t lane, uint32_t j, float acc)
{
    const float d = h2f_lo(r.hdr[0]), dmin = h2f_hi(r.hdr[0]);
    // get_scale_min_k4 for sub-blocks 2j (low nibbles) and 2j+1 (high nibbles), branch-free:
    // scales[n] = header byte 4+n; with n3 = sb & 3:  lo = scales[n3], mid = scales[4+n3], hi = scales[8+n3]
    //   sb <  4: sc = lo & 63                          m = mid & 63
    //   sb >= 4: sc = (hi & 0xF) | ((lo >> 6) << 4)    m = (hi >> 4) | ((mid >> 6) << 4)
    float scv[2], mv[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t sb = 2 * j + h, sh = (sb & 3) * 8;
        const uint32_t lo = (r.hdr[1] >> sh) & 0xFFu, mid = (r.hdr[2] >> sh) & 0xFFu, hi = (r.hdr[3] >> sh) & 0xFFu;
        const bool low = sb < 4;
        const uint32_t sc = low ? (lo & 63u) : ((hi & 0xFu) | ((lo >> 6) << 4));
        const uint32_t m = low ? (mid & 63u) : ((hi >> 4) | ((mid >> 6) << 4));
        scv[h] = d * (float)sc;
        mv[h] = dmin * (float)m;
    }
    f32x2 qa = {0.f, 0.f}, qb = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const f32x4 xa = *reinterpret_cast<const f32x4 *>(xs + ((0 * 4 + t) * 64 + lane) * 4);  // hi = 0
        const f32x4 xb = *reinterpret_cast<const f32x4 *>(xs + ((1 * 4 + t) * 64 + lane) * 4);  // hi = 1
        qa = dot4_u8(r.qs[t] & 0x0F0F0F0Fu, xa, qa);
        qb = dot4_u8((r.qs[t] >> 4) & 0x0F0F0F0Fu, xb, qb);
    }
    acc = fmaf(scv[0], qa[0] + qa[1], acc);
    acc = fmaf(-mv[0], sx[lane], acc);
    acc = fmaf(scv[1], qb[0] + qb[1], acc);
    acc = fmaf(-mv[1], sx[64 + lane], acc);
    return acc;
}

// dot of this lane's 32 Q6_K weights (4 quarters x 8) with x:  sum (q-32)*x = sum q*x - 32*sum x
__device__ __forceinline__ float q6_dot(const Q6Regs &r, const float *xs, uint32_t lane, uint32_t t4, float acc)
{
    const float d = h2f_lo(r.d);
    float tot = 0.f;
#pragma unroll
    for (int quarter = 0; quarter < 4; quarter++) {
        // quarters 0,2 use ql[l], quarters 1,3 use ql[l+32]; low nibble for 0,1, high nibble for 2,3; qh bits 2*quarter
        const u32x2 ql = (quarter & 1) ? r.qlb : r.qla;
        f32x2 s2 = {0.f, 0.f}, xsum = {0.f, 0.f};
#pragma unroll
        for (int p2 = 0; p2 < 2; p2++) {
            const uint32_t lo = (quarter >= 2) ? ((ql[p2] >> 4) & 0x0F0F0F0Fu) : (ql[p2] & 0x0F0F0F0Fu);
            const uint32_t hi = (r.qh[p2] >> (2 * quarter)) & 0x03030303u;
            const uint32_t q = lo | (hi << 4);  // 4 unsigned 6-bit values
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(xs + ((quarter * 2 + p2) * 64 + lane) * 4);
            s2 = dot4_u8(q, xv, s2);
            xsum += f32x2{xv[0], xv[1]} + f32x2{xv[2], xv[3]};
        }
        const float s = fmaf(-32.0f, xsum[0] + xsum[1], s2[0] + s2[1]);
        // scale index within the half: l/16 + 2*quarter with l = 8*t4 .. 8*t4+7  ->  (t4 >> 1) + 2*quarter
        const uint32_t si = (t4 >> 1) + 2 * quarter;
        const uint32_t sw = si < 4 ? r.sc[0] : r.sc[1];
        const int sc = (int)(int8_t)((sw >> ((si & 3) * 8)) & 0xFFu);
        tot = fmaf((float)sc, s, tot);
    }
    return fmaf(d, tot, acc);
}

__device__ __forceinline__ void kq_kv_store(void *base, int f16, uint64_t idx, float v)
{
    if (f16) reinterpret_cast<_Float16 *>(base)[idx] = (_Float16)v;
    else reinterpret_cast<float *>(base)[idx] = v;
}

template <int MODE>
__device__ __forceinline__ void kq_epilogue(const KqParams &p, uint32_t unit, float a0, float a1)
{
    if constexpr (MODE == GEMV_PLAIN) {
        p.y[unit] = a0;
    } else if constexpr (MODE == GEMV_RESIDUAL) {
        p.y[unit] = p.res[unit] + a0;
    } else if constexpr (MODE == GEMV_GATEUP) {
        p.y[unit] = a1 * silu_ref(a0);
    } else {
        const uint32_t row = unit * 2;
        const uint32_t seg = row < p.seg_end[0] ? 0u : (row < p.seg_end[1] ? 1u : 2u);
        const uint32_t r = seg == 0 ? row : (seg == 1 ? row - p.seg_end[0] : row - p.seg_end[1]);
        const uint32_t head = r / p.D, d = r % p.D;
        float o0 = a0, o1 = a1;
        if (seg < 2 && d < p.rope_dims) {
            const float c = p.rope_cs[d], sn = p.rope_cs[d + 1];
            o0 = c * a0 - sn * a1;
            o1 = sn * a0 + c * a1;
        }
        if (seg == 0) {
            p.y[row] = o0;
            p.y[row + 1] = o1;
        } else {
            const uint64_t idx = (uint64_t)p.pos[0] * p.kv_pos_stride + (uint64_t)head * p.kv_head_stride + d;
            void *base = seg == 1 ? p.kc : p.vc;
            kq_kv_store(base, p.kv_f16, idx, o0);
            kq_kv_store(base, p.kv_f16, idx + 1, o1);
        }
    }
}

template <int QT, int MODE, int UPW, bool NORM>
__global__ __launch_bounds__(512) void k_gemv_kq(const KqParams p)
{
    constexpr int RPU = (MODE == GEMV_QKV_ROPE || MODE == GEMV_GATEUP) ? 2 : 1;
    constexpr int R = UPW * RPU;
    constexpr int XN = NORM ? 4 : 16;
    using Regs = typename std::conditional<QT == NFAI_Q4_K, Q4Regs, Q6Regs>::type;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t kpad = p.KC * 2048;
    float *xs = smem;               // kpad floats, permuted (kq_xs_index)
    float *sx = smem + kpad;        // Q4_K only: 16-sums, [chunk][hi][lane] = KC*128 floats
    float *red = sx + p.KC * 128;   // 16 floats

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * nwaves + wid, tw = gridDim.x * nwaves;
    const uint32_t u_begin = (uint32_t)(((uint64_t)p.NU * gw) / tw);
    const uint32_t u_end = (uint32_t)(((uint64_t)p.NU * (gw + 1)) / tw);
    const uint32_t ngroups = (u_end - u_begin + UPW - 1) / UPW;
    const uint32_t nsteps = ngroups * p.KC;
    const uint32_t s8 = lane >> 3, i8 = lane & 7;  // block slot within the chunk, lane within the block

    // ---- activations first (see kernels_gemv.hip) -------------------------------------------
    f32x4 xv[XN], gv[NORM ? XN : 1];
#pragma unroll
    for (int i = 0; i < XN; i++) {
        const uint32_t k = (threadIdx.x + i * blockDim.x) * 4;
        const uint32_t kk = min(k, p.K - 4);
        const f32x4 v = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.x + kk);
        xv[i] = k < p.K ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NORM) gv[i] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.gamma + kk);
    }

    // ---- weight loads of the first two steps ------------------------------------------------
    Regs bufA[R], bufB[R];
    uint32_t ig = 0, ic = 0;  // issue walker: unit group, chunk
    auto issue = [&](Regs (&buf)[R]) {
        const uint32_t blk = min(ic * 8 + s8, p.NB - 1);  // lanes past the last block re-read it with zero weight
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t u = min(min(u_begin + ig * UPW + r / RPU, u_end - 1), p.NU - 1);
            uint32_t seg, row;
            kq_row<MODE>(p, u, r % RPU, seg, row);
            if constexpr (QT == NFAI_Q4_K) buf[r] = q4_load(p, seg, row, blk, i8);
            else buf[r] = q6_load(p, seg, row, blk, i8 >> 2, i8 & 3);
        }
        if (++ic == p.KC) { ic = 0; ++ig; }
    };
    issue(bufA);
    issue(bufB);

    // ---- prologue: RMSNorm, permuted x -> LDS, 16-sums ------------------------------------------
    {
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
            ss = block_sum(ss, red);
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
                *reinterpret_cast<f32x4 *>(xs + kq_xs_index<QT>(k)) = v;
            }
        }
        __syncthreads();
        if constexpr (QT == NFAI_Q4_K) {
            // sx[c][hi][lane] = sum of the 16 activations that lane's (hi) nibbles multiply
            for (uint32_t e = threadIdx.x; e < p.KC * 128; e += blockDim.x) {
                const uint32_t c = e >> 7, hi = (e >> 6) & 1, ln = e & 63;
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(xs + c * 2048 + ((hi * 4 + t) * 64 + ln) * 4);
                    s += (v[0] + v[1]) + (v[2] + v[3]);
                }
                sx[e] = s;
            }
            __syncthreads();
        }
    }

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.f;
    uint32_t cg = 0, cc = 0;  // consume walker
    auto consume = [&](Regs (&buf)[R]) {
        const bool live = cc * 8 + s8 < p.NB;
#pragma unroll
        for (int r = 0; r < R; r++) {
            float a;
            if constexpr (QT == NFAI_Q4_K) a = q4_dot(buf[r], xs + cc * 2048, sx + cc * 128, lane, i8 >> 1, 0.f);
            else a = q6_dot(buf[r], xs + cc * 2048, lane, i8 & 3, 0.f);
            acc[r] += live ? a : 0.f;
        }
        if (cc == p.KC - 1) {
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]);
#pragma unroll
            for (int q = 0; q < UPW; q++) {
                const uint32_t u = u_begin + cg * UPW + q;
                if (lane == (uint32_t)q && u < u_end) kq_epilogue<MODE>(p, u, acc[q * RPU], RPU == 2 ? acc[q * RPU + RPU - 1] : 0.f);
            }
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = 0.f;
        }
        if (++cc == p.KC) { cc = 0; ++cg; }
    };

    uint32_t st = 0;
    for (; st + 3 < nsteps; st += 2) {
        consume(bufA);
        issue(bufA);
        consume(bufB);
        issue(bufB);
    }
    const uint32_t rem = nsteps - st;
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
}

// ---- Q6_K repack: native 210-byte blocks -> four aligned planes (same bytes) -----------------------
__global__ void k_repack_q6k(const uint8_t *src, uint8_t *dst, uint64_t nblk)
{
    const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblk) return;
    const uint32_t t = threadIdx.x & 63;
    const uint8_t *s = src + b * 210;
    dst[b * 128 + t] = s[t];
    dst[b * 128 + 64 + t] = s[64 + t];
    dst[nblk * 128 + b * 64 + t] = s[128 + t];
    if (t < 16) dst[nblk * 192 + b * 16 + t] = s[192 + t];
    if (t < 2) dst[nblk * 208 + b * 2 + t] = s[208 + t];
}

hipError_t launch_repack_q6k(const void *src, void *dst, uint64_t nblk, hipStream_t s)
{
    if (nblk == 0) return hipSuccess;
    k_repack_q6k<<<(uint32_t)((nblk + 3) / 4), 256, 0, s>>>(static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), nblk);
    return hipGetLastError();
}

// ---- dequantise one row to fp32 (embedding gather for quantised token_embd) --------------------------
template <int QT>
__global__ void k_embed_kq(const uint8_t *table, uint64_t n_rows, const uint32_t *tok, float *y, uint32_t E)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E) return;
    const uint32_t nb = E / 256, blk = k >> 8, kk = k & 255;
    const uint64_t row = tok[0];
    if constexpr (QT == NFAI_Q4_K) {
        const uint8_t *b = table + (row * nb + blk) * 144;
        const float d = (float)*reinterpret_cast<const _Float16 *>(b), dmin = (float)*reinterpret_cast<const _Float16 *>(b + 2);
        const uint8_t *scales = b + 4;
        const uint32_t sb = kk >> 5, l = kk & 31;
        uint32_t sc, m;
        if (sb < 4) { sc = scales[sb] & 63; m = scales[sb + 4] & 63; }
        else { sc = (scales[sb + 4] & 0xF) | ((scales[sb - 4] >> 6) << 4); m = (scales[sb + 4] >> 4) | ((scales[sb] >> 6) << 4); }
        const uint8_t q = b[16 + (sb >> 1) * 32 + l];
        const float qv = (float)((sb & 1) ? (q >> 4) : (q & 0xF));
        y[k] = d * (float)sc * qv - dmin * (float)m;
    } else {
        const uint64_t nblk = n_rows * nb, bi = row * nb + blk;
        const uint32_t n = kk >> 7, quarter = (kk >> 5) & 3, l = kk & 31;
        const uint8_t ql = table[bi * 128 + n * 64 + (quarter & 1) * 32 + l];
        const uint8_t qh = table[nblk * 128 + bi * 64 + n * 32 + l];
        const int8_t sc = (int8_t)table[nblk * 192 + bi * 16 + n * 8 + (l >> 4) + 2 * quarter];
        const float d = (float)reinterpret_cast<const _Float16 *>(table + nblk * 208)[bi];
        const int q = (int)(((quarter >= 2) ? (ql >> 4) : (ql & 0xF)) | (((qh >> (2 * quarter)) & 3) << 4)) - 32;
        y[k] = d * (float)sc * (float)q;
    }
}

hipError_t launch_embed_kq(const void *table, int type, uint64_t n_rows, const uint32_t *tok, float *y, uint32_t E, hipStream_t s)
{
    if (E % 256) return hipErrorInvalidValue;
    if (type == NFAI_Q4_K_T16 || type == NFAI_Q6_K_T16) return launch_embed_kqt(table, type, n_rows, tok, y, E, s);
    if (type == NFAI_Q4_K) k_embed_kq<NFAI_Q4_K><<<(E + 255) / 256, 256, 0, s>>>(static_cast<const uint8_t *>(table), n_rows, tok, y, E);
    else if (type == NFAI_Q6_K) k_embed_kq<NFAI_Q6_K><<<(E + 255) / 256, 256, 0, s>>>(static_cast<const uint8_t *>(table), n_rows, tok, y, E);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- dispatch ------------------------------------------------------------------------------------
template <int QT, int MODE, int UPW>
static hipError_t kq_launch(const KqParams &p, uint32_t grid, uint32_t block, size_t lds, hipStream_t s)
{
    if (p.gamma != nullptr) hipLaunchKernelGGL((k_gemv_kq<QT, MODE, UPW, true>), dim3(grid), dim3(block), lds, s, p);
    else hipLaunchKernelGGL((k_gemv_kq<QT, MODE, UPW, false>), dim3(grid), dim3(block), lds, s, p);
    return hipGetLastError();
}

template <int QT, int MODE>
static hipError_t kq_upw(const KqParams &p, int upw, uint32_t grid, uint32_t block, size_t lds, hipStream_t s)
{
    constexpr int RPU = (MODE == GEMV_QKV_ROPE || MODE == GEMV_GATEUP) ? 2 : 1;
    if (upw == 1) return kq_launch<QT, MODE, 1>(p, grid, block, lds, s);
    if (upw == 2) return kq_launch<QT, MODE, 2>(p, grid, block, lds, s);
    if constexpr (RPU == 1) {
        if (upw == 3) return kq_launch<QT, MODE, 3>(p, grid, block, lds, s);
        return kq_launch<QT, MODE, 4>(p, grid, block, lds, s);
    }
    return hipErrorInvalidValue;
}

template <int QT>
static hipError_t kq_mode(const KqParams &p, int mode, int upw, uint32_t grid, uint32_t block, size_t lds, hipStream_t s)
{
    switch (mode) {
        case GEMV_PLAIN: return kq_upw<QT, GEMV_PLAIN>(p, upw, grid, block, lds, s);
        case GEMV_RESIDUAL: return kq_upw<QT, GEMV_RESIDUAL>(p, upw, grid, block, lds, s);
        case GEMV_QKV_ROPE: return kq_upw<QT, GEMV_QKV_ROPE>(p, upw, grid, block, lds, s);
        case GEMV_GATEUP: return kq_upw<QT, GEMV_GATEUP>(p, upw, grid, block, lds, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_gemv_kq(const GemvArgs &a, hipStream_t s)
{
    if (a.w_type != NFAI_Q4_K && a.w_type != NFAI_Q6_K) return hipErrorInvalidValue;
    if (a.K == 0 || a.K % 256 != 0) return hipErrorInvalidValue;
    KqParams p{};
    const int rpu = (a.mode == GEMV_QKV_ROPE || a.mode == GEMV_GATEUP) ? 2 : 1;
    uint32_t total_rows = 0;
    for (int i = 0; i < 3; i++) {
        p.W[i] = reinterpret_cast<const uint8_t *>(a.W[i]);
        p.seg_rows[i] = a.seg_rows[i];
        total_rows += a.seg_rows[i];
    }
    p.seg_end[0] = a.seg_rows[0];
    p.seg_end[1] = a.seg_rows[0] + a.seg_rows[1];
    p.seg_end[2] = total_rows;
    if (a.mode == GEMV_QKV_ROPE) {
        if ((a.seg_rows[0] | a.seg_rows[1] | a.seg_rows[2] | a.D) & 1u) return hipErrorInvalidValue;
        p.NU = total_rows / 2;
    } else if (a.mode == GEMV_GATEUP) {
        if (a.seg_rows[0] != a.seg_rows[1]) return hipErrorInvalidValue;
        p.NU = a.seg_rows[0];
    } else {
        p.NU = a.seg_rows[0];
    }
    if (p.NU == 0) return hipSuccess;
    p.x = a.x; p.gamma = a.gamma; p.eps = a.eps; p.K = a.K;
    p.NB = a.K / 256;
    p.KC = (p.NB + 7) / 8;
    p.y = a.y; p.res = a.res; p.kc = a.kcache; p.vc = a.vcache;
    p.kv_pos_stride = a.kv_pos_stride; p.kv_head_stride = a.kv_head_stride;
    p.rope_cs = a.rope_cs; p.rope_dims = a.rope_dims; p.D = a.D; p.pos = a.pos_dev;
    p.kv_f16 = a.kv_type == NFAI_F16;
    // rows are short (K/256 * 144 or 210 bytes): many waves per CU, several rows per step
    const uint32_t n_cu = a.n_cu;
    uint32_t wpb = 8;
    const uint32_t cands[] = {8, 4, 6, 5, 7};
    bool exact = false;
    for (uint32_t c : cands)
        if (p.NU % (n_cu * c) == 0) { wpb = c; exact = true; break; }
    uint32_t grid = n_cu;
    if (!exact && p.NU < n_cu * 8) { grid = (p.NU + 7) / 8; wpb = 8; }
    if (grid == 0) grid = 1;
    static const int env_bpc = getenv("NFAI_KQ_BPC") ? atoi(getenv("NFAI_KQ_BPC")) : 0;   // sweep knobs
    static const int env_wpb = getenv("NFAI_KQ_WPB") ? atoi(getenv("NFAI_KQ_WPB")) : 0;
    if (env_wpb >= 1 && env_wpb <= 8 && p.NU >= n_cu * 16) wpb = (uint32_t)env_wpb;
    if (env_bpc >= 1 && env_bpc <= 4 && p.NU >= n_cu * 16) grid = n_cu * (uint32_t)env_bpc;
    const uint32_t xn = a.gamma ? 4 : 16;
    const uint32_t kpad = p.KC * 2048;
    while (wpb < 8 && (uint64_t)xn * wpb * 64 * 4 < kpad) wpb++;
    if ((uint64_t)xn * wpb * 64 * 4 < kpad) return hipErrorInvalidValue;
    const uint32_t upw_total = (p.NU + grid * wpb - 1) / (grid * wpb);
    int upw = 1;
    for (int c : {4, 3, 2, 1})  // <= 4 rows per step: two register buffers of 8-9 dwords per row must not spill
        if (c * rpu <= 4 && upw_total % c == 0) { upw = c; break; }
    const size_t lds = ((size_t)kpad + (size_t)p.KC * 128 + 16) * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    if (a.w_type == NFAI_Q4_K) return kq_mode<NFAI_Q4_K>(p, a.mode, upw, grid, wpb * 64, lds, s);
    return kq_mode<NFAI_Q6_K>(p, a.mode, upw, grid, wpb * 64, lds, s);
}

}  // namespace nfai
