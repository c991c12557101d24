// kqm.h — pieces of the int8-MFMA K-quant GEMV (kernels_gemv_kqm.hip: layout, numerics and operand roles are described there)
// shared with the fused attention + Wo launch (kernels_attn.hip).
#pragma once
#include "common.h"

namespace nfai {

#ifndef GLOBAL_AS
#define GLOBAL_AS __attribute__((address_space(1)))
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct KqmParams {
    const uint8_t *W[3];
    uint32_t seg_tiles[3];     // 16-row tiles per segment
    uint32_t seg_tile_end[3];  // running sum (QKV unit -> segment)
    const float *x;
    const float *gamma;
    float eps;
    uint32_t K, NB, NU, UB;
    uint32_t seg6;             // NFAI_KQ_MIXED: bit i set = segment i is Q6_K (else Q4_K)
    uint32_t rot;              // q|k|v: unit u works on tile (u + rot) mod NU
    float *y;
    const float *res;
    void *kc, *vc;
    uint64_t kv_pos_stride, kv_head_stride;
    const float *rope_cs;
    uint32_t rope_dims, D;
    const uint32_t *pos;
    int kv_f16;
    ArgmaxFused am;            // GEMV_PLAIN: first index of the largest output, taken in this launch (am.ticket == nullptr: off)
    BeginParams begin;         // GEMV_QKV_ROPE: the per-token prologue in this launch (begin.on == 0: off)
    NFAI_STAMP_PARAM
};

struct Q4T { u32x4 q0, q1, hdr; };

__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t bits) { return (a & mask) | bits; }

__device__ __forceinline__ float dpp_add8(float v)  // sum within aligned groups of 8 lanes
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}

// One step of a wave: the quants (2 x 1 KiB) and the 16 row headers (256 B) of super-block `blk` of 16-row tile `tile`; `base` = the
// tensor in the T16 layout, `n_tiles` its tiles, NB = super-blocks per row.
__device__ __forceinline__ Q4T q4t_load_raw(const uint8_t *base, uint64_t n_tiles, uint32_t NB, uint32_t tile, uint32_t blk, uint32_t lane)
{
    const uint64_t tb = (uint64_t)tile * NB + blk;
    const uint64_t nblk = n_tiles * 16 * NB;
    Q4T r;
    r.q0 = load_nt16(base + tb * 2048 + lane * 16);
    r.q1 = load_nt16(base + tb * 2048 + 1024 + lane * 16);
    r.hdr = load_nt16(base + nblk * 128 + tb * 256 + (lane & 15) * 16);
    return r;
}

__device__ __forceinline__ Q4T q4t_load(const KqmParams &p, uint32_t seg, uint32_t tile, uint32_t blk, uint32_t lane)
{
    return q4t_load_raw(p.W[seg], p.seg_tiles[seg], p.NB, tile, blk, lane);
}

// 64 weights of one lane (sub-blocks 2G: low nibbles, 2G+1: high nibbles) against the activations.
// af: this lane's four A fragments (slot 2n + hf; all zero for the lanes whose A rows are zero);
// sums = {SX[2G], SX[2G+1]} of this super-block.
__device__ __forceinline__ float q4t_dot(const Q4T &w, const i32x4 (&af)[4], f32x2 sums, uint32_t g)
{
    constexpr uint32_t M = 0x0F0F0F0Fu;
    i32x4 dlo = {0, 0, 0, 0}, dhi = {0, 0, 0, 0};
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        const u32x4 q = hf ? w.q1 : w.q0;
        const u32x4 blo = q & M, bhi = (q >> 4) & M;  // 16 weights each, one byte per weight, k-slot j = byte j
        dlo = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[0 * 2 + hf], __builtin_bit_cast(i32x4, blo), dlo, 0, 0, 0);
        dhi = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[1 * 2 + hf], __builtin_bit_cast(i32x4, bhi), dhi, 0, 0, 0);
    }
    // get_scale_min_k4 (ggml) for sub-blocks 2G and 2G+1, branch-free (see kernels_gemv_kq.hip)
    const float d = h2f_lo(w.hdr[0]), dmin = h2f_hi(w.hdr[0]);
    float scv[2], mv[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t sb = 2 * g + h, sh = (sb & 3) * 8;
        const uint32_t lo8 = (w.hdr[1] >> sh) & 0xFFu, mid = (w.hdr[2] >> sh) & 0xFFu, hi8 = (w.hdr[3] >> sh) & 0xFFu;
        const bool low = sb < 4;
        const uint32_t sc = low ? (lo8 & 63u) : ((hi8 & 0xFu) | ((lo8 >> 6) << 4));
        const uint32_t mn = low ? (mid & 63u) : ((hi8 >> 4) | ((mid >> 6) << 4));
        scv[h] = d * (float)sc;
        mv[h] = dmin * (float)mn;
    }
    // three signed base-256 digits of the fixed-point activations: sum q*x' = S0 + 256*S1 + 65536*S2 (integers, exact)
    const float vlo = fmaf((float)dlo[2], 65536.0f, fmaf((float)dlo[1], 256.0f, (float)dlo[0]));
    const float vhi = fmaf((float)dhi[2], 65536.0f, fmaf((float)dhi[1], 256.0f, (float)dhi[0]));
    float a = scv[0] * vlo;
    a = fmaf(-mv[0], sums[0], a);
    a = fmaf(scv[1], vhi, a);
    a = fmaf(-mv[1], sums[1], a);
    return a;
}

// Fixed-point staging of ONE 256-element super-block of the activation vector by one wave (lane holds elements 4*lane .. +3, after the
// optional RMSNorm): power-of-two scale so that |x * 2^S| < 2^22, three signed base-256 digits per element written as MFMA A
// fragments [blk][slot:4][G][digit][16 B] (Q4_K layout in xa, Q6_K layout in xa6), the sums of x' per scale group and 2^-S.
// Nothing written here is read by another wave.
template <bool HAS4, bool HAS6>
__device__ __forceinline__ void kqm_stage(const f32x4 v, const uint32_t blk, const uint32_t lane, uint8_t *xa, uint8_t *xa6, float *sums,
                                          float *sums6, float *scl)
{
    const uint32_t k = lane * 4;  // position inside the super-block
    const float am = wave_max(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    int S = 0;
    if (am > 0.f && am < 3.0e38f) S = 21 - ilogbf(am);  // |x * 2^S| < 2^22: a 24-bit signed integer after rounding
    S = max(-100, min(100, S));
    const float scale = ldexpf(1.0f, S);
    if (lane == 0) scl[blk] = ldexpf(1.0f, -S);
    uint32_t d0 = 0, d1 = 0, d2 = 0;  // digit planes of the four elements, one byte each
    float sx = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float vs = v[e] * scale;
        const int xi = (int)rintf(vs);
        const int b0 = (int)(int8_t)(xi & 0xFF);
        const int r1 = (xi - b0) >> 8;
        const int b1 = (int)(int8_t)(r1 & 0xFF);
        const int b2 = (r1 - b1) >> 8;
        d0 |= (uint32_t)(b0 & 0xFF) << (8 * e);
        d1 |= (uint32_t)(b1 & 0xFF) << (8 * e);
        d2 |= (uint32_t)(b2 & 0xFF) << (8 * e);
        sx += (float)xi;
    }
    // A fragments [blk][slot:4][G][digit][16 bytes]; a lane reads slot s at +256*s from its (G, digit) base.
    //   Q4_K: k = (2G+n)*32 + hf*16 + j, slot = 2n + hf, sums per sub-block of 32 -> [blk][G][n]
    //   Q6_K: k = n*128 + qd*32 + lh*16 + j, G = 2n + lh, slot = qd, sums per group of 16 -> [blk][G][qd]
    const uint32_t j = k & 15;
    // sums over aligned groups of 4 lanes (16 elements) and of 8 lanes (32 elements)
    sx += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sx), 0xB1, 0xF, 0xF, true));
    sx += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sx), 0x4E, 0xF, 0xF, true));
    const float sx16 = sx;
    const float sx32 = sx + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sx), 0x141, 0xF, 0xF, true));
    if constexpr (HAS4) {
        const uint32_t sb = (k >> 5) & 7, hf = (k >> 4) & 1, g = sb >> 1, slot = (sb & 1) * 2 + hf;
        uint8_t *frag = xa + (size_t)blk * 1024 + (slot * 4 + g) * 64 + j;
        *reinterpret_cast<uint32_t *>(frag) = d0;
        *reinterpret_cast<uint32_t *>(frag + 16) = d1;
        *reinterpret_cast<uint32_t *>(frag + 32) = d2;
        if ((lane & 7) == 0) sums[(blk * 4 + g) * 4 + (sb & 1)] = sx32;
    }
    if constexpr (HAS6) {
        const uint32_t g = ((k >> 7) & 1) * 2 + ((k >> 4) & 1), slot = (k >> 5) & 3;
        uint8_t *frag = xa6 + (size_t)blk * 1024 + (slot * 4 + g) * 64 + j;
        *reinterpret_cast<uint32_t *>(frag) = d0;
        *reinterpret_cast<uint32_t *>(frag + 16) = d1;
        *reinterpret_cast<uint32_t *>(frag + 32) = d2;
        if ((lane & 3) == 0) sums6[(blk * 4 + g) * 4 + slot] = sx16;
    }
}

}  // namespace nfai
