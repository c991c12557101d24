// kernels_prefill.hip — batched prompt ingestion on the matrix cores.
//
// The reference has no batched prefill: the prompt goes through the M = 1 GEMV path token by token
// (LlamaModel.cs:103-126; every MatrixMultiplyShader is built with inputRowCount = 1,
// TransformerBlock.cs:47-101).  Here T prompt tokens go through each block together: the seven
// projections become [T x K] x [N x K]^T GEMMs on v_mfma_f32_16x16x32_f16 (fp16 operands, fp32
// accumulate), attention is one launch per chunk (k_attn_prefill; or two batched GEMMs around a causal row softmax).  Oracle = the
// reference's token-by-token fp32 path; tolerance is the "stated fp16 tolerance" (activations are
// rounded to fp16 on the way into the MFMA): logits max|d| <= 5e-2, same argmax (tests).
//
// GEMM  C[M][N] (+R) = A[M][K] * B[N][K]^T      (B = GGUF weight matrix as stored: K contiguous)
//   k_gemm_f16_glds (the projections; round 3): operands global -> LDS by LDS-DMA, EIGHT waves per workgroup —
//     wide N (gate|up): 256 x 128 x 64 tiles, 4 x 2 waves of 64 x 64, three stages (one workgroup per CU);
//     narrow N (q|k|v, Wo, Wdown): 128 x {48, 64, 80, 96} tiles picked by rounds x (128 + BN), 4 x 1 waves of 32 x BN in TWO groups
//     that split every K tile's k-steps and add their accumulators in LDS at the end, BK = 128 with three stages or 64 with four;
//   epilogues: fp32 (+ residual), fp16, SiLU * up (fp16), RoPE + q / KV-cache / fp16 K / V^T stores (EPI_ROPE);
//   k_gemm_f16 (P.V of the unfused attention, odd shapes, the register-staged baseline) and k_gemm_kq (dequant-in-LDS, opt-in) below.
//   blockIdx -> (m tile, n tile) is XCD-aware: the m tiles that share a weight tile get ids that are equal mod 8 (same XCD, adjacent
//   dispatch slots), so a weight tile is read from HBM once and from that XCD's L2 by the other m tiles.
//   At 512 rows these GEMMs are bound by what a CU takes in and by the lock-step of a workgroup's K loop, not by MFMA or HBM
//   (22.5 % of the dense fp16 peak over a whole 512-token prefill at 3B, gate|up 890 TFLOP/s).
#include <stdlib.h>

#include "common.h"

namespace nfai {

#define GLOBAL_AS __attribute__((address_space(1)))

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct GemmParams {
    const _Float16 *A;      // [M][lda] fp16
    const _Float16 *B[3];   // up to three row segments of [rows][ldb] fp16 (q | k | v, gate | up) — one launch
    uint32_t seg_end[3];    // cumulative row ends of the segments (multiples of 64)
    void *C;                // [M][ldc] fp32 (EPI_F32) or fp16 (EPI_F16, EPI_SILU)
    const float *R;         // optional residual [M][ldc] (EPI_F32)
    uint32_t M, N, K, lda, ldb, ldc;
    uint64_t a_bs, b_bs, c_bs;   // batch strides (elements)
    uint32_t b_div;              // B batch index = batch / b_div (GQA: query heads share a kv head)
    float alpha;
    uint32_t causal;             // 0: dense.  1: C = scores [query t][key s], tiles with only s > causal_pos0 + t are skipped
                                 // (the softmax never reads them).  2: A = probabilities [t][s]: K tiles past the last
                                 // unmasked key of the tile's rows are skipped (they multiply zeros).
    uint32_t causal_pos0;
    NFAI_STAMP_PARAM
    GemmRope rope;               // EPI_ROPE
    uint32_t ksplit;             // > 1: blockIdx.z owns K tiles [z*KT/ksplit, (z+1)*KT/ksplit) and adds its product atomically
                                 // into C, which the host has initialised with the residual (or zeros)
};

enum { EPI_F32 = 0, EPI_F16 = 1, EPI_SILU = 2, EPI_ROPE = 3 };

// LDS tile rows are BK halves = CH chunks of 16 B; the chunk index is XORed with the row so that the
// 16 rows a ds_read_b128 fragment read touches at one k-chunk land in 16 different bank groups
template <int CH> __device__ __forceinline__ uint32_t lds_off(uint32_t row, uint32_t chunk)
{
    return row * (CH * 16) + ((chunk ^ (row & (CH - 1))) << 4);
}

// Block tile BM x BN x BK, WM x WN waves, each wave owns (BM/WM) x (BN/WN) = TM x TN MFMA tiles.
// Global -> VGPR -> LDS with a ring of three register sets: while tile kt is multiplied out of LDS, tile
// kt+1 sits in registers and tile kt+2 is in flight.
//   <128, 64, 4, 1>: 192-320 workgroups on the narrow projections (N = E), the attention GEMMs
//   <128, 128, 2, 2>: wave tile 64 x 64 (8 fragment reads per 16 MFMAs instead of 6 per 8) where N is wide
// EPI_SILU: the tile's columns are 32 gate | 32 up rows of the SAME 32 outputs per 64-column wave slice, so
// act = up * silu(gate) is formed in registers and written as fp16 (SiLUShader + ElementWiseMultiplicationShader
// fused into the GEMM: no fp32 gate/up round trip through HBM).
// Epilogue shared by the GEMM kernels: accumulators of one wave tile -> C (fp32 + residual / fp16 / SiLU*up fp16 / split-K atomics).
template <int BM, int BN, int WM, int WN, int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_store(f32x4 (&acc)[TM][TN], const GemmParams &p, uint32_t m0, uint32_t n0, uint32_t wm, uint32_t wn,
                                           uint32_t lane, uint32_t batch)
{
    // C/D layout of mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg.
    const uint32_t rbase = m0 + wm * (BM / WM) + (lane >> 4) * 4, cbase = wn * (BN / WN) + (lane & 15);
    if constexpr (EPI == EPI_SILU) {
        _Float16 *Ch = static_cast<_Float16 *>(p.C) + (uint64_t)batch * p.c_bs;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t row = rbase + i * 16 + r, col = n0 / 2 + wn * 32 + j * 16 + (lane & 15);
                    const float g = acc[i][j][r] * p.alpha, u = acc[i][j + 2][r] * p.alpha;
                    if (row < p.M) Ch[(uint64_t)row * p.ldc + col] = (_Float16)(u * silu_ref(g));
                }
    } else if constexpr (EPI == EPI_ROPE) {
        // RoPEShader.cs:249-262 on the accumulators + the stores of k_rope_store_tiles (same arithmetic, same order: the results are
        // bit-identical to GEMM -> fp32 -> that kernel).  A 16-column block of the tile lies in one head of one of q | k | v (segment
        // ends and D are multiples of 16), so everything column-dependent is wave-uniform per j; the partner of a rotated pair is
        // the neighbouring lane (columns 2i, 2i+1 <-> lanes 2i, 2i+1 of the block).
        const GemmRope &rp = p.rope;
        const uint32_t half = rp.D / 2, l15 = lane & 15;
        const uint32_t cwave = n0 + wn * (BN / WN);  // first column of the wave's slice: wave-uniform, like everything derived from it
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        // cos / sin of the lane's rows and column pairs, ALL requested before the first use (a branch or a use between the loads makes
        // hipcc wait for each one: dependent L2 round trips); pairs that are not rotated read a valid entry and ignore it
        f32x2 csv[TN][TM][4];
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const uint32_t c0 = cwave + j * 16;
            const uint32_t cc0 = c0 - (c0 < p.seg_end[0] ? 0u : (c0 < p.seg_end[1] ? p.seg_end[0] : p.seg_end[1]));
            const uint32_t pr = ((cc0 % rp.D) + l15) >> 1;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    csv[j][i][r] = *reinterpret_cast<const GLOBAL_AS f32x2 *>((const GLOBAL_AS float *)rp.cs + ((uint64_t)min(rbase + i * 16 + r, p.M - 1) * half + pr) * 2);
        }
        _Float16 *const qh = static_cast<_Float16 *>(rp.qh), *const kh = static_cast<_Float16 *>(rp.kh), *const vt = static_cast<_Float16 *>(rp.vt);
#pragma unroll
        for (int j = 0; j < TN; j++) {
            // scalars of the 16-column block (kept in SGPRs: with a lane-dependent column hipcc selects the cache pointer by a LOAD
            // from the kernel arguments and then waits vmcnt(0) — i.e. for every earlier store — in front of each store)
            const uint32_t c0 = __builtin_amdgcn_readfirstlane(cwave + j * 16);
            const uint32_t seg = c0 < p.seg_end[0] ? 0u : (c0 < p.seg_end[1] ? 1u : 2u);
            const uint32_t cc0 = c0 - (seg == 0 ? 0u : (seg == 1 ? p.seg_end[0] : p.seg_end[1]));
            const uint32_t head = cc0 / rp.D, dd = cc0 % rp.D + l15, cc = cc0 + l15;
            const bool rot = seg < 2 && (dd & ~1u) < rp.rope_dims;
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const uint32_t row0 = rbase + i * 16;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float a = acc[i][j][r] * p.alpha;
                    const float other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));
                    const f32x2 cs = csv[j][i][r];
                    const float ro = (dd & 1) ? (cs[1] * other + cs[0] * a) : (cs[0] * a - cs[1] * other);
                    o[r] = rot ? ro : a;
                }
                if (seg == 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (row0 + r < p.M) qh[(uint64_t)(row0 + r) * rp.H * rp.D + cc] = (_Float16)o[r];
                    continue;
                }
                // K / V rows of the cache (TransformerBlock's KV write) ...
                const uint64_t at0 = (uint64_t)(rp.pos0 + row0) * rp.pos_stride + (uint64_t)head * rp.head_stride + dd;
                if (seg == 1) {
                    if (rp.kv_f16) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (row0 + r < p.M) static_cast<_Float16 *>(rp.kc)[at0 + r * rp.pos_stride] = (_Float16)o[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (row0 + r < p.M) static_cast<float *>(rp.kc)[at0 + r * rp.pos_stride] = o[r];
                    }
                    // ... and the fp16 K rows the chunk's attention reads
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (row0 + r < p.M) kh[((uint64_t)head * rp.Spad + rp.pos0 + row0 + r) * rp.D + dd] = (_Float16)o[r];
                } else {
                    if (rp.kv_f16) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (row0 + r < p.M) static_cast<_Float16 *>(rp.vc)[at0 + r * rp.pos_stride] = (_Float16)o[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (row0 + r < p.M) static_cast<float *>(rp.vc)[at0 + r * rp.pos_stride] = o[r];
                    }
                    // V^T: the lane's four rows are four consecutive positions of one row of the transposed matrix
                    _Float16 *dst = vt + ((uint64_t)head * rp.D + dd) * rp.Spad + rp.pos0 + row0;
                    if (row0 + 3 < p.M && ((rp.pos0 + row0) & 3u) == 0 && (rp.Spad & 3u) == 0) {
                        *reinterpret_cast<h4 *>(dst) = h4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (row0 + r < p.M) dst[r] = (_Float16)o[r];
                    }
                }
            }
        }
    } else if constexpr (EPI == EPI_F16) {
        _Float16 *Ch = static_cast<_Float16 *>(p.C) + (uint64_t)batch * p.c_bs;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t row = rbase + i * 16 + r, col = n0 + cbase + j * 16;
                    if (row < p.M) Ch[(uint64_t)row * p.ldc + col] = (_Float16)(acc[i][j][r] * p.alpha);
                }
    } else {
        float *Cb = static_cast<float *>(p.C) + (uint64_t)batch * p.c_bs;
        if (p.ksplit > 1) {  // split-K: C already holds the residual (or zeros); every split adds its share
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const uint32_t row = rbase + i * 16 + r, col = n0 + cbase + j * 16;
                        if (row < p.M) __hip_atomic_fetch_add(Cb + ((uint64_t)row * p.ldc + col), acc[i][j][r] * p.alpha, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
            return;
        }
        const float *Rb = p.R ? p.R + (uint64_t)batch * p.c_bs : nullptr;
        // The residual is fetched for a whole row block first (rows clamped, no branch per element: a branch around
        // each load makes hipcc wait vmcnt(0) per element = dependent L2 round trips).
        if (Rb) {
#pragma unroll
            for (int i = 0; i < TM; i++) {
                float rv[TN][4];
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const uint32_t row = min(rbase + i * 16 + r, p.M - 1), col = n0 + cbase + j * 16;
                        rv[j][r] = *((const GLOBAL_AS float *)Rb + ((uint64_t)row * p.ldc + col));
                    }
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[i][j][r] = fmaf(acc[i][j][r], p.alpha, rv[j][r]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] *= p.alpha;
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t row = rbase + i * 16 + r, col = n0 + cbase + j * 16;
                    if (row < p.M) Cb[(uint64_t)row * p.ldc + col] = acc[i][j][r];
                }
    }
}

// KS > 1: KS wave groups share the tile and split every K tile's k-steps between them (twice the waves per SIMD to
// overlap fragment reads, MFMAs and staging when only ~one workgroup fits or exists per CU); their accumulators
// meet in LDS once, after the K loop.
// RING = register sets of staged tiles: RING-1 K tiles of global loads are in flight per workgroup.  A workgroup's
// stream rate is (bytes in flight) / (memory latency): with one workgroup per CU, 2 tiles x 24 KB over ~2500 cycles
// is ~19 B/clk/CU (measured on the Wo / Wdown GEMMs of 192 workgroups), so those get RING = 5.
template <int BM, int BN, int WM, int WN, int BK, int EPI, int KS, int RING>
__global__ __launch_bounds__(WM *WN *KS * 64) void k_gemm_f16(const GemmParams p)
{
    constexpr int NT = WM * WN * KS * 64;
    static_assert((BK / 32) % KS == 0, "k-steps of a tile must divide over the wave groups");
    constexpr int CH = BK / 8;                // 16-byte chunks per tile row
    constexpr int AN = BM * CH / NT;          // A chunks per thread per tile
    constexpr int BN_ = BN * CH / NT;         // B chunks per thread per tile
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int TILE_BYTES = (BM + BN) * BK * 2;
    static_assert(EPI != EPI_SILU || BN / WN == 64, "SiLU epilogue pairs columns inside a 64-wide wave slice");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    auto ldsA = [&](int buf) -> uint8_t * { return lds + buf * TILE_BYTES; };
    auto ldsB = [&](int buf) -> uint8_t * { return lds + buf * TILE_BYTES + BM * BK * 2; };

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % (WM * WN), ksg = (tid >> 6) / (WM * WN), wm = wave / WN, wn = wave % WN;
    const uint32_t tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
    uint32_t mt_i, nt_i;
    {
        const uint32_t id = blockIdx.x;
        if (tiles_n % 8 == 0) {
            const uint32_t xcd = id & 7, slot = id >> 3;
            nt_i = (slot / tiles_m) * 8 + xcd;
            mt_i = slot % tiles_m;
        } else {
            nt_i = id / tiles_m;
            mt_i = id % tiles_m;
        }
    }
    const uint32_t m0 = mt_i * BM, n0 = nt_i * BN, batch = blockIdx.y;
    if (p.causal == 1 && n0 > p.causal_pos0 + m0 + BM - 1) return;  // every (t, s) of this tile has s > pos0 + t
    const GLOBAL_AS uint8_t *Ab = (const GLOBAL_AS uint8_t *)(p.A + (uint64_t)batch * p.a_bs);
    // weight segment of this n tile (EPI_SILU: rows come from both segments, see b_row)
    const uint32_t seg = n0 < p.seg_end[0] ? 0u : (n0 < p.seg_end[1] ? 1u : 2u);
    const uint32_t nrow0 = n0 - (seg == 0 ? 0u : p.seg_end[seg - 1]);
    const uint64_t bbatch = (uint64_t)(batch / p.b_div) * p.b_bs;
    const GLOBAL_AS uint8_t *Bb = (const GLOBAL_AS uint8_t *)(p.B[seg] + bbatch);
    const GLOBAL_AS uint8_t *Bg = (const GLOBAL_AS uint8_t *)p.B[0], *Bu = (const GLOBAL_AS uint8_t *)p.B[1];

    u32x4 ra[RING][AN], rb[RING][BN_];
    auto load_tile = [&](u32x4 (&a)[AN], u32x4 (&b)[BN_], uint32_t kt) {
        const uint32_t k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < AN; i++) {
            const uint32_t c = tid + i * NT, row = c / CH, q = c % CH;
            const uint32_t gr = min(m0 + row, p.M - 1);
            a[i] = *reinterpret_cast<const GLOBAL_AS u32x4 *>(Ab + ((uint64_t)gr * p.lda + k0 + q * 8) * 2);
        }
#pragma unroll
        for (int i = 0; i < BN_; i++) {
            const uint32_t c = tid + i * NT, row = c / CH, q = c % CH;
            if constexpr (EPI == EPI_SILU) {
                // tile column `row`: 64-wide slice sl, inside it 32 gate rows then 32 up rows of outputs n0/2 + sl*32 ..
                const uint32_t sl = row >> 6, cc = row & 63, out = n0 / 2 + sl * 32 + (cc & 31);
                const GLOBAL_AS uint8_t *base = cc < 32 ? Bg : Bu;
                b[i] = *reinterpret_cast<const GLOBAL_AS u32x4 *>(base + ((uint64_t)out * p.ldb + k0 + q * 8) * 2);
            } else {
                b[i] = *reinterpret_cast<const GLOBAL_AS u32x4 *>(Bb + ((uint64_t)(nrow0 + row) * p.ldb + k0 + q * 8) * 2);
            }
        }
    };
    auto store_tile = [&](const u32x4 (&a)[AN], const u32x4 (&b)[BN_], int buf) {
#pragma unroll
        for (int i = 0; i < AN; i++) {
            const uint32_t c = tid + i * NT, row = c / CH, q = c % CH;
            *reinterpret_cast<u32x4 *>(ldsA(buf) + lds_off<CH>(row, q)) = a[i];
        }
#pragma unroll
        for (int i = 0; i < BN_; i++) {
            const uint32_t c = tid + i * NT, row = c / CH, q = c % CH;
            *reinterpret_cast<u32x4 *>(ldsB(buf) + lds_off<CH>(row, q)) = b[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int cur) {
#pragma unroll
        for (int kk = 0; kk < BK / 32 / KS; kk++) {
            const uint32_t ks = ksg * (BK / 32 / KS) + kk;
            const uint32_t chunk = ks * 4 + (lane >> 4);
            f16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) af[i] = *reinterpret_cast<const f16x8 *>(ldsA(cur) + lds_off<CH>(wm * (BM / WM) + i * 16 + (lane & 15), chunk));
#pragma unroll
            for (int j = 0; j < TN; j++) bf[j] = *reinterpret_cast<const f16x8 *>(ldsB(cur) + lds_off<CH>(wn * (BN / WN) + j * 16 + (lane & 15), chunk));
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    uint32_t KT = p.K / BK / p.ksplit;
    const uint32_t kt_base = blockIdx.z * KT;
    if (p.causal == 2) KT = min(KT, (p.causal_pos0 + m0 + BM + BK - 1) / BK);  // keys 0 .. pos0 + (last row of the tile)
#pragma unroll
    for (int r = 0; r < RING - 1; r++) load_tile(ra[r], rb[r], kt_base + min((uint32_t)r, KT - 1));
    store_tile(ra[0], rb[0], 0);
    __syncthreads();
    // iteration kt: [request tile kt+RING-1 into the free register set] [MFMAs on tile kt] [tile kt+1: regs -> LDS]
    // [barrier]; unrolled by RING so that every register-set index is a compile-time constant (no scratch).
    // Loads and LDS stores are NEVER conditional (past the end they repeat the last tile, an L2 hit): with
    // `if (kt + 2 < KT) load_tile(...)` hipcc's waitcnt pass takes the no-load path as the worst case and waits
    // vmcnt(0) before the LDS store of tile kt+1 — i.e. also for the tiles it has just requested: every K tile then
    // costs a full memory latency (seen in the ISA as vmcnt(11)..vmcnt(0) per iteration; now vmcnt(23)..vmcnt(12)).
    for (uint32_t kt0 = 0; kt0 < KT; kt0 += RING) {
#pragma unroll
        for (int r = 0; r < RING; r++) {
            const uint32_t kt = kt0 + r;
            load_tile(ra[(r + RING - 1) % RING], rb[(r + RING - 1) % RING], kt_base + min(kt + RING - 1, KT - 1));
            if (kt < KT) compute(kt & 1);
            store_tile(ra[(r + 1) % RING], rb[(r + 1) % RING], (kt + 1) & 1);
            __syncthreads();
        }
    }

    if constexpr (KS > 1) {
        // the last __syncthreads() of the loop has passed: the tile buffers are free.  Group g > 0 parks its
        // accumulators at [g-1][wave][i][j][lane] (16 B per lane: conflict-free), group 0 adds them up.
        f32x4 *park = reinterpret_cast<f32x4 *>(lds);
        static_assert((KS - 1) * WM * WN * TM * TN * 64 * 16 <= 2 * TILE_BYTES, "accumulator exchange must fit the tile buffers");
        if (ksg > 0) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) park[((((ksg - 1) * (WM * WN) + wave) * TM + i) * TN + j) * 64 + lane] = acc[i][j];
        }
        __syncthreads();
        if (ksg > 0) return;
#pragma unroll
        for (int g = 1; g < KS; g++)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] += park[((((g - 1) * (WM * WN) + wave) * TM + i) * TN + j) * 64 + lane];
    }
    gemm_store<BM, BN, WM, WN, EPI, TM, TN>(acc, p, m0, n0, wm, wn, lane, batch);
}

// ---- dequant-in-LDS GEMM for K-quant weights (T16 layouts of kernels_gemv_kqm.hip) ---------------------------------------
// C[M][N] (+R) = A[M][K] fp16 * W[N][K]^T with W in Q4_K / Q6_K blocks: the quantised bytes of a B tile (64 or 128 rows x 64 k:
// 2.3 - 6.6 KB instead of 8 - 16 KB of fp16) go global -> VGPR, are expanded to fp16 by the thread that loaded them
// (d * sc * q - dmin * m, resp. d * sc * (q - 32), rounded to fp16 exactly as a separate widening pass would) and written
// into the same XOR-swizzled LDS tile the fp16 kernel uses; A staging, MFMA loop and epilogues are those of k_gemm_f16.
// A 64-wide K tile is a quarter of a super-block: K tile kt -> super-block kt / 4, quarter c = kt % 4.
//   Q4_K: quarter c = lane group G = c of the T16 layout: sub-blocks 2c (low nibbles) and 2c+1 (high nibbles).
//         task (row, part 0..3) expands 16 weights: sub-block 2c + (part >> 1), bytes 16 (part & 1) .. +15 of the row's 32.
//   Q6_K: quarter c = half n = c >> 1, ggml quarters 2 (c & 1) + {0, 1}; task (row, part): quarter 2 (c & 1) + (part >> 1),
//         columns 16 (part & 1) .. +15 = lane group 2 n + (part & 1).
template <int QT> struct KqTask;
template <> struct KqTask<NFAI_Q4_K_T16> { u32x4 qs, hdr; };
template <> struct KqTask<NFAI_Q6_K_T16> { u32x4 ql, qh, sc; uint32_t d; };

template <int QT, int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void k_gemm_kq(const GemmParams p)
{
    constexpr int NT = 256, BK = 64, CH = 8, RING = 3;
    constexpr int AN = BM * CH / NT;       // A chunks per thread per tile
    constexpr int BT = BN * 4 / NT;        // B tasks (16 weights each) per thread per tile
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int TILE_BYTES = (BM + BN) * BK * 2;
    static_assert(WM * WN == 4 && BT >= 1, "256 threads");
    static_assert(EPI != EPI_SILU || BN / WN == 64, "SiLU epilogue pairs columns inside a 64-wide wave slice");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    auto ldsA = [&](int buf) -> uint8_t * { return lds + buf * TILE_BYTES; };
    auto ldsB = [&](int buf) -> uint8_t * { return lds + buf * TILE_BYTES + BM * BK * 2; };

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    const uint32_t tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
    uint32_t mt_i, nt_i;
    {
        const uint32_t id = blockIdx.x;
        if (tiles_n % 8 == 0) {
            const uint32_t xcd = id & 7, slot = id >> 3;
            nt_i = (slot / tiles_m) * 8 + xcd;
            mt_i = slot % tiles_m;
        } else {
            nt_i = id / tiles_m;
            mt_i = id % tiles_m;
        }
    }
    const uint32_t m0 = mt_i * BM, n0 = nt_i * BN, batch = blockIdx.y;
    const GLOBAL_AS uint8_t *Ab = (const GLOBAL_AS uint8_t *)p.A;
    const uint32_t NB = p.K / 256;
    // per task: weight tensor, row inside it, rows of that tensor (the header planes sit behind rows * NB blocks)
    const GLOBAL_AS uint8_t *wbase[BT];
    uint32_t wrow[BT], part[BT];
    uint64_t wnblk[BT];
#pragma unroll
    for (int i = 0; i < BT; i++) {
        const uint32_t c = tid + i * NT, row = c >> 2;
        part[i] = c & 3;
        if constexpr (EPI == EPI_SILU) {
            const uint32_t sl = row >> 6, cc = row & 63;
            wbase[i] = (const GLOBAL_AS uint8_t *)(cc < 32 ? p.B[0] : p.B[1]);
            wrow[i] = n0 / 2 + sl * 32 + (cc & 31);
            wnblk[i] = (uint64_t)p.seg_end[0] * NB;  // gate and up have N / 2 rows each
        } else {
            const uint32_t n = n0 + row;
            const uint32_t seg = n < p.seg_end[0] ? 0u : (n < p.seg_end[1] ? 1u : 2u);
            const uint32_t s0 = seg == 0 ? 0u : p.seg_end[seg - 1];
            wbase[i] = (const GLOBAL_AS uint8_t *)p.B[seg];
            wrow[i] = n - s0;
            wnblk[i] = (uint64_t)(p.seg_end[seg] - s0) * NB;
        }
    }

    u32x4 ra[RING][AN];
    KqTask<QT> rb[RING][BT];
    auto load_tile = [&](u32x4 (&a)[AN], KqTask<QT> (&b)[BT], uint32_t kt) {
        const uint32_t k0 = kt * BK, blk = kt >> 2, c4 = kt & 3;
#pragma unroll
        for (int i = 0; i < AN; i++) {
            const uint32_t c = tid + i * NT, row = c / CH, q = c % CH;
            const uint32_t gr = min(m0 + row, p.M - 1);
            a[i] = *reinterpret_cast<const GLOBAL_AS u32x4 *>(Ab + ((uint64_t)gr * p.lda + k0 + q * 8) * 2);
        }
#pragma unroll
        for (int i = 0; i < BT; i++) {
            const uint64_t tb = (uint64_t)(wrow[i] >> 4) * NB + blk;
            const uint32_t r = wrow[i] & 15;
            if constexpr (QT == NFAI_Q4_K_T16) {
                b[i].qs = *reinterpret_cast<const GLOBAL_AS u32x4 *>(wbase[i] + tb * 2048 + (part[i] & 1) * 1024 + (c4 * 16 + r) * 16);
                b[i].hdr = *reinterpret_cast<const GLOBAL_AS u32x4 *>(wbase[i] + wnblk[i] * 128 + tb * 256 + r * 16);
            } else {
                const uint32_t qd = 2 * (c4 & 1) + (part[i] >> 1), G = 2 * (c4 >> 1) + (part[i] & 1);
                b[i].ql = *reinterpret_cast<const GLOBAL_AS u32x4 *>(wbase[i] + tb * 3072 + (qd & 1) * 1024 + (G * 16 + r) * 16);
                b[i].qh = *reinterpret_cast<const GLOBAL_AS u32x4 *>(wbase[i] + tb * 3072 + 2048 + (G * 16 + r) * 16);
                b[i].sc = *reinterpret_cast<const GLOBAL_AS u32x4 *>(wbase[i] + wnblk[i] * 192 + tb * 256 + r * 16);
                b[i].d = *reinterpret_cast<const GLOBAL_AS uint16_t *>(wbase[i] + wnblk[i] * 208 + tb * 32 + r * 2);
            }
        }
    };
    auto store_tile = [&](const u32x4 (&a)[AN], const KqTask<QT> (&b)[BT], int buf, uint32_t kt) {
        const uint32_t c4 = kt & 3;
#pragma unroll
        for (int i = 0; i < AN; i++) {
            const uint32_t c = tid + i * NT, row = c / CH, q = c % CH;
            *reinterpret_cast<u32x4 *>(ldsA(buf) + lds_off<CH>(row, q)) = a[i];
        }
#pragma unroll
        for (int i = 0; i < BT; i++) {
            const uint32_t row = (tid + i * NT) >> 2, pt = part[i];
            f16x8 o[2];
            if constexpr (QT == NFAI_Q4_K_T16) {
                const float d = h2f_lo(b[i].hdr[0]), dmin = h2f_hi(b[i].hdr[0]);
                const uint32_t sb = 2 * c4 + (pt >> 1), sh = (sb & 3) * 8;
                const uint32_t lo8 = (b[i].hdr[1] >> sh) & 0xFFu, mid = (b[i].hdr[2] >> sh) & 0xFFu, hi8 = (b[i].hdr[3] >> sh) & 0xFFu;
                const bool low = sb < 4;
                const uint32_t sc = low ? (lo8 & 63u) : ((hi8 & 0xFu) | ((lo8 >> 6) << 4));
                const uint32_t mn = low ? (mid & 63u) : ((hi8 >> 4) | ((mid >> 6) << 4));
                const float d1 = d * (float)sc, m1 = dmin * (float)mn;
                const uint32_t nsh = (pt >> 1) * 4;  // high nibbles for the odd sub-block
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const uint32_t q = (b[i].qs[e >> 2] >> (8 * (e & 3) + nsh)) & 0xFu;
                    o[e >> 3][e & 7] = (_Float16)(d1 * (float)q - m1);
                }
            } else {
                const uint32_t qd = 2 * (c4 & 1) + (pt >> 1), si = 8 * (c4 >> 1) + (pt & 1) + 2 * qd;
                const uint32_t sw = si < 8 ? (si < 4 ? b[i].sc[0] : b[i].sc[1]) : (si < 12 ? b[i].sc[2] : b[i].sc[3]);
                const float dsc = h2f_lo(b[i].d) * (float)(int)(int8_t)((sw >> ((si & 3) * 8)) & 0xFFu);
                const uint32_t nsh = (qd >> 1) * 4;
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const uint32_t lb = (b[i].ql[e >> 2] >> (8 * (e & 3) + nsh)) & 0xFu;
                    const uint32_t hb = (b[i].qh[e >> 2] >> (8 * (e & 3) + 2 * qd)) & 3u;
                    o[e >> 3][e & 7] = (_Float16)(dsc * (float)((int)(lb | (hb << 4)) - 32));
                }
            }
            const uint32_t chunk0 = (pt >> 1) * 4 + (pt & 1) * 2;  // 16 consecutive k of the tile = two 16-byte chunks
            *reinterpret_cast<f16x8 *>(ldsB(buf) + lds_off<CH>(row, chunk0)) = o[0];
            *reinterpret_cast<f16x8 *>(ldsB(buf) + lds_off<CH>(row, chunk0 + 1)) = o[1];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int cur) {
#pragma unroll
        for (int ks = 0; ks < BK / 32; ks++) {
            const uint32_t chunk = ks * 4 + (lane >> 4);
            f16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) af[i] = *reinterpret_cast<const f16x8 *>(ldsA(cur) + lds_off<CH>(wm * (BM / WM) + i * 16 + (lane & 15), chunk));
#pragma unroll
            for (int j = 0; j < TN; j++) bf[j] = *reinterpret_cast<const f16x8 *>(ldsB(cur) + lds_off<CH>(wn * (BN / WN) + j * 16 + (lane & 15), chunk));
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    const uint32_t KT = p.K / BK;
#pragma unroll
    for (int r = 0; r < RING - 1; r++) load_tile(ra[r], rb[r], min((uint32_t)r, KT - 1));
    store_tile(ra[0], rb[0], 0, 0);
    __syncthreads();
    for (uint32_t kt0 = 0; kt0 < KT; kt0 += RING) {  // same ring as k_gemm_f16: nothing conditional except the MFMAs past the end
#pragma unroll
        for (int r = 0; r < RING; r++) {
            const uint32_t kt = kt0 + r;
            load_tile(ra[(r + RING - 1) % RING], rb[(r + RING - 1) % RING], min(kt + RING - 1, KT - 1));
            if (kt < KT) compute(kt & 1);
            store_tile(ra[(r + 1) % RING], rb[(r + 1) % RING], (kt + 1) & 1, min(kt + 1, KT - 1));
            __syncthreads();
        }
    }
    gemm_store<BM, BN, WM, WN, EPI, TM, TN>(acc, p, m0, n0, wm, wn, lane, batch);
}

template <int QT, int EPI>
static hipError_t gemm_kq_pick(const GemmParams &p, uint32_t n_cu, hipStream_t s)
{
    const uint64_t big_tiles = (uint64_t)((p.M + 127) / 128) * (p.N / 128);
    if (p.N % 128 == 0 && big_tiles >= (uint64_t)n_cu * 3 / 2) {
        constexpr int LDS = 2 * (128 + 128) * 64 * 2;
        hipLaunchKernelGGL((k_gemm_kq<QT, 128, 128, 2, 2, EPI>), dim3((uint32_t)big_tiles, 1, 1), dim3(256), LDS, s, p);
    } else {
        constexpr int LDS = 2 * (128 + 64) * 64 * 2;
        hipLaunchKernelGGL((k_gemm_kq<QT, 128, 64, 4, 1, EPI>), dim3(((p.M + 127) / 128) * (p.N / 64), 1, 1), dim3(256), LDS, s, p);
    }
    return hipGetLastError();
}

// W segments are T16 K-quant tensors of ONE type (a.b_type); fp16 A, fp32 C (+R) or the SiLU*up fp16 epilogue.
hipError_t launch_gemm_kq(const GemmArgs &a, hipStream_t s)
{
    if (a.M == 0 || a.N == 0) return hipSuccess;
    if (a.b_type != NFAI_Q4_K_T16 && a.b_type != NFAI_Q6_K_T16) return hipErrorInvalidValue;
    if (a.N % 64 != 0 || a.K % 256 != 0 || a.K == 0 || a.lda % 8 != 0 || (a.batch > 1)) return hipErrorInvalidValue;
    GemmParams p{};
    p.A = static_cast<const _Float16 *>(a.A);
    p.B[0] = static_cast<const _Float16 *>(a.B);
    p.B[1] = static_cast<const _Float16 *>(a.B1 ? a.B1 : a.B);
    p.B[2] = static_cast<const _Float16 *>(a.B2 ? a.B2 : a.B);
    p.seg_end[0] = a.B1 ? a.n0 : a.N;
    p.seg_end[1] = a.B2 ? a.n0 + a.n1 : a.N;
    p.seg_end[2] = a.N;
    p.C = a.C; p.R = a.R;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldb = a.K; p.ldc = a.ldc;
    p.b_div = 1; p.alpha = a.alpha; p.ksplit = 1;
    const uint32_t n_cu = a.n_cu ? a.n_cu : 256;
    if (a.epi == EPI_SILU) {
        if (!a.B1 || a.n0 * 2 != a.N || a.R || a.n0 % 64) return hipErrorInvalidValue;
        return a.b_type == NFAI_Q4_K_T16 ? gemm_kq_pick<NFAI_Q4_K_T16, EPI_SILU>(p, n_cu, s) : gemm_kq_pick<NFAI_Q6_K_T16, EPI_SILU>(p, n_cu, s);
    }
    if (a.epi != EPI_F32) return hipErrorInvalidValue;
    // a tile must not straddle a segment boundary (rows of a tile address one tensor's planes)
    const uint32_t gran = (p.N % 128 == 0 && (p.seg_end[0] | p.seg_end[1]) % 128 == 0) ? 128u : 64u;
    if ((p.seg_end[0] | p.seg_end[1]) % 64) return hipErrorInvalidValue;
    if (gran == 64) {
        constexpr int LDS = 2 * (128 + 64) * 64 * 2;
        const dim3 grid(((p.M + 127) / 128) * (p.N / 64), 1, 1);
        if (a.b_type == NFAI_Q4_K_T16) hipLaunchKernelGGL((k_gemm_kq<NFAI_Q4_K_T16, 128, 64, 4, 1, EPI_F32>), grid, dim3(256), LDS, s, p);
        else hipLaunchKernelGGL((k_gemm_kq<NFAI_Q6_K_T16, 128, 64, 4, 1, EPI_F32>), grid, dim3(256), LDS, s, p);
        return hipGetLastError();
    }
    return a.b_type == NFAI_Q4_K_T16 ? gemm_kq_pick<NFAI_Q4_K_T16, EPI_F32>(p, n_cu, s) : gemm_kq_pick<NFAI_Q6_K_T16, EPI_F32>(p, n_cu, s);
}

// ---- direct-to-LDS variants (128 x 128 x 64 tiles with 2 x 2 waves of 64 x 64; 128 x 64 x 64 with 4 x 1 waves) --------------
// Operands go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write pass); NST LDS stages, tile kt + NST - 1
// is requested while tile kt is multiplied.  One raw s_barrier per K tile:
//     s_waitcnt vmcnt(G (NST - 2))   my share of tile kt has landed (G = LDS-DMA instructions per wave and tile: 8 or 6) (tiles kt+1 .. kt+NST-2 may still be in flight)
//     s_barrier                      everybody's share has; everybody is past the MFMAs of tile kt-1
//     G x global_load_lds            tile kt+NST-1 -> the stage tile kt-1 just vacated (past the end: the last tile again,
//                                    into a stage nobody reads any more — the counts stay uniform, nothing is conditional)
//     MFMAs on tile kt
// An LDS-DMA instruction writes lane-linear (wave-uniform base + lane * 16 B): the XOR swizzle of the fragment reads is
// applied to the SOURCE address instead (LDS slot (row, c') receives global chunk c' ^ (row & 7)).  Plain __syncthreads()
// would drain the DMAs (its fence waits vmcnt(0)), hence the raw barrier + counted waits (MI355X guide, §5).
template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// PIPE: the fragments of k-step s+1 are read from LDS while the MFMAs of step s run (double-buffered registers, issue order
// pinned with sched_group_barrier).  Left alone hipcc reads a whole step, waits lgkmcnt(0) and then issues its MFMAs: with one
// wave per SIMD (the narrow GEMMs: one workgroup per CU) every step then exposes a full LDS latency.
// NSTB > NST ("roles"): the B operand (the weights: cold, read once from HBM) gets a deeper ring than the A operand (the
// activations: L2 hits), NSTB - 1 weight tiles in flight per workgroup.  vmcnt retires in issue order PER WAVE, so a wave that
// issued both operands would wait for its far-ahead weight tiles whenever it waits for the next activation tile: waves 0-1 issue
// only A, waves 2-3 only B, each with its own counted wait; the barrier joins them.  (In a prefill the 192-workgroup
// projections were found waiting on HBM latency, not on L2 / LDS / MFMA: profiles/round2_prefill_pmc.json — 1 TB/s on the
// memory side, MFMA 22 % busy.)
// KS = 2: two groups of WM x WN waves share the tile; group g multiplies the k-steps s with s % KS == g of every K tile, and the
// groups' accumulators meet in LDS after the K loop (the narrow projections have one workgroup per CU: a second wave per SIMD
// fills the first one's waits for LDS reads, its barrier and its LDS-DMA issue).  WM * WN = 8, KS = 1: eight waves, one tile each.
template <int BM, int BN, int WM, int WN, int EPI, int NST, int BK = 64, bool PIPE = false, int NSTB = NST, int KS = 1>
__global__ __launch_bounds__(WM *WN *KS * 64) void k_gemm_f16_glds(const GemmParams p)
{
    static_assert(NSTB >= NST, "the weight ring is at least as deep as the activation ring");
    constexpr int NWT = WM * WN, NW = NWT * KS;  // waves that own a tile position; waves of the workgroup
    static_assert(NW == 4 || NW == 8 || NW == 16, "four, eight or sixteen waves");
    constexpr bool ROLES = NSTB != NST;
    static_assert(!ROLES || NW == 4, "operand roles are written for four waves");
    static_assert(KS == 1 || (BK / 32) % KS == 0, "k-steps of a tile are dealt to the wave groups");
    constexpr int CH = BK / 8, RPI = 64 / CH, TM = BM / WM / 16, TN = BN / WN / 16;  // RPI = tile rows per LDS-DMA instruction
    // LDS-DMA instructions per wave and tile.  Without roles the tile's BM / RPI + BN / RPI instructions are dealt to the four
    // waves round-robin; when BN / RPI is not a multiple of four (BN = 80, 48: tile widths that give exactly 256 workgroups on the
    // 5120- and 3072-column projections at 512 rows) the last B instruction exists only for the first waves (b_last).  With roles
    // a wave pair shares one operand: instruction i of pair member w covers LDS rows (i * 2 + w) * RPI ...
    constexpr int NAI = BM / RPI, NBI = BN / RPI;
    constexpr int DEAL = ROLES ? 2 : NW;
    constexpr int AG = NAI / DEAL, BG = ROLES ? NBI / 2 : (NBI + NW - 1) / NW;
    constexpr bool B_EVEN = ROLES || NBI % NW == 0;
    static_assert(BM % (RPI * DEAL) == 0 && BN % RPI == 0 && (!ROLES || NBI % 2 == 0), "tile rows per LDS-DMA instruction");
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, B_BASE = NST * A_BYTES;  // LDS: NST A stages, then NSTB B stages
    static_assert(EPI != EPI_SILU || BN / WN == 64, "SiLU epilogue pairs columns inside a 64-wide wave slice");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typedef __attribute__((address_space(3))) uint8_t lds_u8;

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), kg = wave / NWT, wm = (wave % NWT) / WN, wn = wave % WN;
    const uint32_t tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
    uint32_t mt_i, nt_i;
    {
        const uint32_t id = blockIdx.x;
        if (tiles_n % 8 == 0) {
            const uint32_t xcd = id & 7, slot = id >> 3;
            nt_i = (slot / tiles_m) * 8 + xcd;
            mt_i = slot % tiles_m;
        } else {
            nt_i = id / tiles_m;
            mt_i = id % tiles_m;
        }
    }
    const uint32_t m0 = mt_i * BM, n0 = nt_i * BN, batch = blockIdx.y;
    if (p.causal == 1 && n0 > p.causal_pos0 + m0 + BM - 1) return;
    const GLOBAL_AS uint8_t *Ab = (const GLOBAL_AS uint8_t *)(p.A + (uint64_t)batch * p.a_bs);
    const GLOBAL_AS uint8_t *Bg = (const GLOBAL_AS uint8_t *)p.B[0], *Bu = (const GLOBAL_AS uint8_t *)p.B[1];
    const bool role_a = !ROLES || wave < 2, role_b = !ROLES || wave >= 2;  // wave-uniform
    const uint32_t dw = ROLES ? (wave & 1) : wave;                         // this wave's place in the deal
    const bool b_last = B_EVEN || (uint32_t)(BG - 1) * NW + wave < (uint32_t)NBI;

    // per-lane source rows: lane = (row % RPI) * CH + c'
    const uint32_t lrow = lane / CH, lc = lane % CH;
    static_assert(AG <= 8 && BG <= 8, "source pointer arrays");
    const GLOBAL_AS uint8_t *asrc[8], *bsrc[8];  // fixed bounds: with [AG] / [BG] the host pass of hipcc 7.2 silently drops the kernel's definition
#pragma unroll
    for (int i = 0; i < AG; i++) {
        const uint32_t row = (i * DEAL + dw) * RPI + lrow;
        asrc[i] = Ab + ((uint64_t)min(m0 + row, p.M - 1) * p.lda + (lc ^ (row & (CH - 1))) * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < BG; i++) {
        const uint32_t row = min((i * DEAL + dw) * RPI + lrow, (uint32_t)BN - 1);
        const uint32_t chunk = lc ^ (row & (CH - 1));
        if constexpr (EPI == EPI_SILU) {
            const uint32_t sl = row >> 6, cc = row & 63, out = n0 / 2 + sl * 32 + (cc & 31);
            bsrc[i] = (cc < 32 ? Bg : Bu) + ((uint64_t)out * p.ldb + chunk * 8) * 2;
        } else {
            // the row's segment (q | k | v are three tensors): per row, so a tile may straddle a segment boundary
            const uint32_t n = n0 + row;
            const uint32_t seg = n < p.seg_end[0] ? 0u : (n < p.seg_end[1] ? 1u : 2u);
            const uint32_t nrow = n - (seg == 0 ? 0u : p.seg_end[seg - 1]);
            bsrc[i] = (const GLOBAL_AS uint8_t *)(p.B[seg] + (uint64_t)(batch / p.b_div) * p.b_bs) + ((uint64_t)nrow * p.ldb + chunk * 8) * 2;
        }
    }
    auto issue_a = [&](uint32_t kt, uint32_t stage) {
        const uint32_t koff = kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < AG; i++) {
            lds_u8 *da = (lds_u8 *)(lds + stage * A_BYTES + (i * DEAL + dw) * 1024);
            __builtin_amdgcn_global_load_lds(asrc[i] + koff, da, 16, 0, 0);
        }
    };
    auto issue_b = [&](uint32_t kt, uint32_t stage) {
        const uint32_t koff = kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < BG; i++) {
            lds_u8 *db = (lds_u8 *)(lds + B_BASE + stage * B_BYTES + (i * DEAL + dw) * 1024);
            if (B_EVEN || i < BG - 1 || b_last) __builtin_amdgcn_global_load_lds(bsrc[i] + koff, db, 16, 0, 0);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint32_t KT = p.K / BK;
    if (p.causal == 2) KT = min(KT, (p.causal_pos0 + m0 + BM + BK - 1) / BK);
    // prologue: tiles 0 .. depth-2 of each ring (past the end: the last tile again — the counts stay uniform)
    if constexpr (ROLES) {
        if (role_a) {
#pragma unroll
            for (int s2 = 0; s2 < NST - 1; s2++) issue_a(min((uint32_t)s2, KT - 1), s2);
        } else {
#pragma unroll
            for (int s2 = 0; s2 < NSTB - 1; s2++) issue_b(min((uint32_t)s2, KT - 1), s2);
        }
    } else {
#pragma unroll
        for (int s2 = 0; s2 < NST - 1; s2++) {
            issue_a(min((uint32_t)s2, KT - 1), s2);
            issue_b(min((uint32_t)s2, KT - 1), s2);
        }
    }

    uint32_t cur = 0, fill = NST - 1;     // A: stage being multiplied, stage being refilled
    uint32_t curb = 0, fillb = NSTB - 1;  // B likewise
#ifdef NFAI_STAMPS
    // diagnostic build: shader-clock cycles this wave spends waiting for its tile (vmcnt), at the barrier, issuing, multiplying
    STAMP_DECL;
    unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_mul = 0, c0 = __builtin_amdgcn_s_memtime(), c1;
    const unsigned long long c_start = c0;
#define GEMM_TICK(acc_) do { __builtin_amdgcn_sched_barrier(0); c1 = __builtin_amdgcn_s_memtime(); acc_ += c1 - c0; c0 = c1; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GEMM_TICK(acc_) ((void)0)
#endif
    for (uint32_t kt = 0; kt < KT; kt++) {
        GEMM_TICK(c_mul);
        if constexpr (ROLES) {
            if (role_a) wait_vmcnt<AG * (NST - 2)>();
            else wait_vmcnt<BG * (NSTB - 2)>();
        } else {
            if (b_last) wait_vmcnt<(AG + BG) * (NST - 2)>();
            else wait_vmcnt<(AG + BG - 1) * (NST - 2)>();
        }
        GEMM_TICK(c_wait);
        __builtin_amdgcn_s_barrier();
        GEMM_TICK(c_bar);
        if (role_a) issue_a(min(kt + NST - 1, KT - 1), fill);
        if (role_b) issue_b(min(kt + NSTB - 1, KT - 1), fillb);
        GEMM_TICK(c_issue);
        const uint8_t *la = lds + cur * A_BYTES, *lb = lds + B_BASE + curb * B_BYTES;
        constexpr int KSTEPS = BK / 32;
        auto load_frags = [&](int ks, f16x8 (&a)[TM], f16x8 (&b)[TN]) {
            const uint32_t chunk = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < TM; i++) a[i] = *reinterpret_cast<const f16x8 *>(la + lds_off<CH>(wm * (BM / WM) + i * 16 + (lane & 15), chunk));
#pragma unroll
            for (int j = 0; j < TN; j++) b[j] = *reinterpret_cast<const f16x8 *>(lb + lds_off<CH>(wn * (BN / WN) + j * 16 + (lane & 15), chunk));
        };
        if constexpr (PIPE) {
            constexpr int KL = KSTEPS / KS;  // this wave group's k-steps of the tile: kg, kg + KS, ...
            const int k0 = KS > 1 ? (int)kg : 0;
            f16x8 af[2][TM], bf[2][TN];
            load_frags(k0, af[0], bf[0]);
#pragma unroll
            for (int ks = 0; ks < KL; ks++) {
                if (ks + 1 < KL) load_frags((ks + 1) * KS + k0, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
            }
            // issue order: reads of step 0; then, per step, one read of the next step behind every MFMA until it is complete
            static_assert(TM * TN >= TM + TN, "more MFMAs than fragment reads per step");
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
            for (int ks = 0; ks < KL; ks++) {
                if (ks + 1 < KL) {
#pragma unroll
                    for (int q = 0; q < TM + TN; q++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KSTEPS / KS; ks++) {
                f16x8 af[TM], bf[TN];
                load_frags(ks * KS + (KS > 1 ? (int)kg : 0), af, bf);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        cur = cur + 1 == NST ? 0 : cur + 1;
        fill = fill + 1 == NST ? 0 : fill + 1;
        curb = curb + 1 == NSTB ? 0 : curb + 1;
        fillb = fillb + 1 == NSTB ? 0 : fillb + 1;
    }
    GEMM_TICK(c_mul);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus DMAs of the last iterations must not outlive the workgroup's LDS
    if constexpr (KS > 1) {
        // every wave's DMAs have landed and every wave is past its last fragment read: the stages are free for the groups' sums
        static_assert((KS - 1) * BM * BN * 4 <= (NST * BM + NSTB * BN) * BK * 2, "the other groups' accumulators fit the stages");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        constexpr uint32_t SLAB = NWT * TM * TN * 64;  // f32x4 words per group
        f32x4 *slab = reinterpret_cast<f32x4 *>(lds) + (size_t)(wave % NWT) * TM * TN * 64 + lane;
        if (kg != 0) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) slab[(kg - 1) * SLAB + (i * TN + j) * 64] = acc[i][j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kg != 0) return;
#pragma unroll
        for (int g2 = 1; g2 < KS; g2++)  // fixed order: group 0 + group 1 + ...
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] += slab[(g2 - 1) * SLAB + (i * TN + j) * 64];
    }
    gemm_store<BM, BN, WM, WN, EPI, TM, TN>(acc, p, m0, n0, wm, wn, lane, batch);
#ifdef NFAI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long c_epi = 0;
    GEMM_TICK(c_epi);  // epilogue: group sums (KS > 1), residual loads, stores, until the stores are acknowledged
    _st.t[0] = c_start; _st.t[1] = c_wait; _st.t[2] = c_bar; _st.t[3] = c_issue; _st.t[4] = c_mul; _st.t[5] = c1; _st.t[6] = KT; _st.t[7] = c_epi;
    STAMP_FLUSH(p.stamps, blockIdx.x * NW + wave, 8);
#endif
#undef GEMM_TICK
}

template <int BM, int BN, int WM, int WN, int EPI, int NST, int BK = 64, bool PIPE = false, int NSTB = NST, int KS = 1>
static hipError_t gemm_launch_glds(const GemmParams &p, uint32_t batch, hipStream_t s)
{
    constexpr int NW = WM * WN * KS;
    constexpr int LDS = (NST * BM + NSTB * BN) * BK * 2;
    static_assert(LDS <= 160 * 1024, "LDS of one CU");
    if (p.K % BK) return hipErrorInvalidValue;
    auto kern = k_gemm_f16_glds<BM, BN, WM, WN, EPI, NST, BK, PIPE, NSTB, KS>;
    static bool attr_set = false;
    if (LDS > 64 * 1024 && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const uint32_t tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    GemmParams pp = p;
    NFAI_STAMP_SET(pp, "gemm_glds", tiles, NW * 64);
    hipLaunchKernelGGL(kern, dim3(tiles, batch, 1), dim3(NW * 64), LDS, s, pp);
    return hipGetLastError();
}

// (Round 4 built a producer / consumer form of this kernel — four loader waves issuing every LDS-DMA, the MFMA waves polling per-stage
// words in LDS, no s_barrier in the K loop — for VERDICT r3 item 7: correct, and 3 % SLOWER than the lock-step kernel on gate|up
// (61.2 against 59.5 us; profiles/round4_gemm_pc.txt).  Removed again; `git show 09bf429` has the code.)

template <int BM, int BN, int WM, int WN, int BK, int EPI, int KS = 1, int RING = 3>
static hipError_t gemm_launch(const GemmParams &p, uint32_t batch, hipStream_t s)
{
    constexpr int LDS = 2 * (BM + BN) * BK * 2;
    auto kern = k_gemm_f16<BM, BN, WM, WN, BK, EPI, KS, RING>;
    if (LDS > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
    }
    const uint32_t tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    hipLaunchKernelGGL(kern, dim3(tiles, batch, p.ksplit), dim3(WM * WN * KS * 64), LDS, s, p);
    return hipGetLastError();
}

template <int EPI>
static hipError_t gemm_pick(GemmParams &p, uint32_t batch, uint32_t n_cu, int variant, hipStream_t s)
{
    if (variant) {  // explicit configuration (tests, tools)
        if (((variant >= 2 && variant <= 4) || variant == 11 || variant == 18 || variant == 19 || (variant >= 21 && variant <= 25)) && p.N % 128 != 0) return hipErrorInvalidValue;
        if (((variant >= 12 && variant <= 15) || (variant >= 27 && variant <= 34) || variant == 36 || variant == 37 || variant == 40 || variant == 41 || variant == 45 || variant == 46 || variant == 47 || variant == 48) && (p.seg_end[0] % 16 || p.seg_end[1] % 16)) return hipErrorInvalidValue;
        switch (variant) {
            case 1: return gemm_launch<128, 64, 4, 1, 64, EPI>(p, batch, s);
            case 2: return gemm_launch<128, 128, 2, 2, 64, EPI>(p, batch, s);
            case 3: return gemm_launch_glds<128, 128, 2, 2, EPI, 2>(p, batch, s);
            case 4: return gemm_launch_glds<128, 128, 2, 2, EPI, 3>(p, batch, s);
            case 5: return gemm_launch_glds<128, 64, 4, 1, EPI, 2>(p, batch, s);
            case 6: return gemm_launch_glds<128, 64, 4, 1, EPI, 3>(p, batch, s);
            case 7: return gemm_launch_glds<128, 64, 4, 1, EPI, 4>(p, batch, s);
            case 8: return gemm_launch_glds<128, 64, 4, 1, EPI, 3, 64, true>(p, batch, s);    // + pipelined fragment reads
            case 9: return gemm_launch_glds<128, 64, 4, 1, EPI, 3, 128, true>(p, batch, s);   // BK 128, 3 stages (144 KB)
            case 10: return gemm_launch_glds<128, 64, 4, 1, EPI, 2, 128, true>(p, batch, s);  // BK 128, 2 stages
            case 11: return gemm_launch_glds<128, 128, 2, 2, EPI, 2, 64, true>(p, batch, s);
            case 12: if constexpr (EPI != EPI_SILU) return p.N % 80 ? hipErrorInvalidValue : gemm_launch_glds<128, 80, 4, 1, EPI, 3>(p, batch, s); else break;
            case 13: if constexpr (EPI != EPI_SILU) return p.N % 48 ? hipErrorInvalidValue : gemm_launch_glds<128, 48, 4, 1, EPI, 3>(p, batch, s); else break;
            case 14: if constexpr (EPI != EPI_SILU) return p.N % 80 ? hipErrorInvalidValue : gemm_launch_glds<128, 80, 4, 1, EPI, 4>(p, batch, s); else break;
            case 15: if constexpr (EPI != EPI_SILU) return p.N % 48 ? hipErrorInvalidValue : gemm_launch_glds<128, 48, 4, 1, EPI, 4>(p, batch, s); else break;
            case 16: return gemm_launch_glds<128, 64, 4, 1, EPI, 3, 64, false, 6>(p, batch, s);    // weights 5 tiles ahead (96 KB)
            case 17: return gemm_launch_glds<128, 64, 4, 1, EPI, 3, 64, false, 9>(p, batch, s);    // weights 8 tiles ahead (120 KB)
            case 18: return gemm_launch_glds<128, 128, 2, 2, EPI, 3, 64, false, 4>(p, batch, s);   // 128 x 128: weights 3 tiles ahead (112 KB)
            case 19: return gemm_launch_glds<128, 128, 2, 2, EPI, 3, 64, false, 6>(p, batch, s);   // 5 tiles ahead (144 KB)
            case 20: if constexpr (EPI != EPI_SILU) return p.N % 80 ? hipErrorInvalidValue : gemm_launch_glds<128, 80, 4, 1, EPI, 3, 64, false, 6>(p, batch, s); else break;
            case 21: return gemm_launch_glds<256, 128, 2, 2, EPI, 2>(p, batch, s);                             // four waves of 128 x 64 (96 KB)
            case 22: return gemm_launch_glds<256, 128, 2, 2, EPI, 3>(p, batch, s);                             // (144 KB)
            case 23: return gemm_launch_glds<256, 128, 4, 2, EPI, 2>(p, batch, s);                             // eight waves of 64 x 64
            case 24: return gemm_launch_glds<256, 128, 4, 2, EPI, 3>(p, batch, s);
            case 25: return gemm_launch_glds<128, 128, 2, 2, EPI, 3, 64, false, 3, 2>(p, batch, s);            // two wave groups split the k-steps
            case 26: return gemm_launch_glds<128, 64, 4, 1, EPI, 3, 64, false, 3, 2>(p, batch, s);
            case 27: if constexpr (EPI != EPI_SILU) return p.N % 80 ? hipErrorInvalidValue : gemm_launch_glds<128, 80, 4, 1, EPI, 3, 64, false, 3, 2>(p, batch, s); else break;
            case 28: if constexpr (EPI != EPI_SILU) return p.N % 48 ? hipErrorInvalidValue : gemm_launch_glds<128, 48, 4, 1, EPI, 3, 64, false, 3, 2>(p, batch, s); else break;
            case 29: if constexpr (EPI != EPI_SILU) return p.N % 48 ? hipErrorInvalidValue : gemm_launch_glds<128, 48, 4, 1, EPI, 4, 64, false, 4, 2>(p, batch, s); else break;
            case 31: if constexpr (EPI != EPI_SILU) return p.N % 48 ? hipErrorInvalidValue : gemm_launch_glds<128, 48, 4, 1, EPI, 5, 64, false, 5, 2>(p, batch, s); else break;
            case 32: if constexpr (EPI != EPI_SILU) return p.N % 48 || p.K % 128 ? hipErrorInvalidValue : gemm_launch_glds<128, 48, 4, 1, EPI, 3, 128, false, 3, 2>(p, batch, s); else break;
            case 33: if constexpr (EPI != EPI_SILU) return p.N % 48 || p.K % 128 ? hipErrorInvalidValue : gemm_launch_glds<128, 48, 4, 1, EPI, 3, 128, false, 3, 4>(p, batch, s); else break;
            case 34: if constexpr (EPI != EPI_SILU) return p.N % 80 ? hipErrorInvalidValue : gemm_launch_glds<128, 80, 4, 1, EPI, 4, 64, false, 4, 2>(p, batch, s); else break;
            case 35: return p.N % 128 ? hipErrorInvalidValue : gemm_launch_glds<256, 128, 4, 2, EPI, 3, 64, false, 3, 2>(p, batch, s);
            case 36: if constexpr (EPI != EPI_SILU) return p.N % 48 ? hipErrorInvalidValue : gemm_launch_glds<128, 48, 4, 1, EPI, 6, 64, false, 6, 2>(p, batch, s); else break;
            case 37: if constexpr (EPI != EPI_SILU) return p.N % 80 ? hipErrorInvalidValue : gemm_launch_glds<128, 80, 4, 1, EPI, 5, 64, false, 5, 2>(p, batch, s); else break;
            case 38: return p.N % 128 ? hipErrorInvalidValue : gemm_launch_glds<128, 128, 2, 2, EPI, 4, 64, false, 4, 2>(p, batch, s);
            case 39: return p.K % 128 ? hipErrorInvalidValue : gemm_launch_glds<128, 64, 4, 1, EPI, 3, 128, false, 3, 2>(p, batch, s);
            case 40: if constexpr (EPI != EPI_SILU) return p.N % 80 || p.K % 128 ? hipErrorInvalidValue : gemm_launch_glds<128, 80, 4, 1, EPI, 3, 128, false, 3, 2>(p, batch, s); else break;
            case 41: if constexpr (EPI != EPI_SILU) return p.N % 96 ? hipErrorInvalidValue : gemm_launch_glds<128, 96, 4, 1, EPI, 4, 64, false, 4, 2>(p, batch, s); else break;
            case 42: return gemm_launch_glds<128, 64, 4, 1, EPI, 4, 64, false, 4, 2>(p, batch, s);
            case 43: return p.N % 128 ? hipErrorInvalidValue : gemm_launch_glds<256, 128, 4, 2, EPI, 3, 64, true>(p, batch, s);
            case 44: return p.N % 128 ? hipErrorInvalidValue : gemm_launch_glds<256, 128, 4, 2, EPI, 2, 64, true>(p, batch, s);
            case 45: if constexpr (EPI != EPI_SILU) return p.N % 48 || p.K % 128 ? hipErrorInvalidValue : gemm_launch_glds<128, 48, 4, 1, EPI, 3, 128, true, 3, 2>(p, batch, s); else break;
            case 46: if constexpr (EPI != EPI_SILU) return p.N % 80 || p.K % 128 ? hipErrorInvalidValue : gemm_launch_glds<128, 80, 4, 1, EPI, 3, 128, true, 3, 2>(p, batch, s); else break;
            case 47: if constexpr (EPI != EPI_SILU) return p.N % 96 || p.K % 128 ? hipErrorInvalidValue : gemm_launch_glds<64, 96, 2, 2, EPI, 3, 128, false, 3, 2>(p, batch, s); else break;
            case 48: if constexpr (EPI != EPI_SILU) return p.N % 96 ? hipErrorInvalidValue : gemm_launch_glds<64, 96, 2, 2, EPI, 4, 64, false, 4, 2>(p, batch, s); else break;
            case 30: if constexpr (EPI != EPI_SILU) return p.N % 96 ? hipErrorInvalidValue : gemm_launch_glds<128, 96, 2, 2, EPI, 3, 64, false, 3, 2>(p, batch, s); else break;
        }
        return hipErrorInvalidValue;
    }
    // Short prompts (<= 64 rows): a 128-row tile stages 128 activation rows (the rows past M clamped: L2 hits, but the same bytes through
    // the CU's load path) beside 48-80 weight rows — the activations are most of what a workgroup takes in.  64-row tiles halve that
    // (four waves of 16 x 64, four stages of 16 KB).  NFAI_GEMM_M64_SHORT=0: the 128-row tiles at every row count.
    static const bool m64_short = !(getenv("NFAI_GEMM_M64_SHORT") && atoi(getenv("NFAI_GEMM_M64_SHORT")) == 0);
    if (m64_short && p.M <= 64 && p.causal == 0 && p.ksplit == 1) return gemm_launch_glds<64, 64, 4, 1, EPI, 4>(p, batch, s);
    // A dense product whose K range the caller split over the batch dimension (llama.hip: Wdown at >= 256 rows): 256 x 128 tiles — per
    // workgroup half the operand bytes of the 128 x 48 tiling this shape takes unsplit.
    if (batch > 1 && p.causal == 0 && p.ksplit == 1 && p.M >= 256 && p.N % 128 == 0 && ((p.M + 127) / 128) % 2 == 0)
        return gemm_launch_glds<256, 128, 4, 2, EPI, 3, 64, true>(p, batch, s);
    static const int env_big = getenv("NFAI_GEMM_BIG") ? atoi(getenv("NFAI_GEMM_BIG")) : 1;
    const uint64_t big_tiles = (uint64_t)((p.M + 127) / 128) * (p.N / 128) * batch;
    if constexpr (EPI == EPI_F32) {
        // Narrow outputs (N = E): too few 128 x 128 tiles for the chip, and the 128 x 64 configuration is LDS-bound
        // (~370 TFLOP/s).  Split K over workgroups instead: ~1.5+ workgroups per CU of the efficient configuration,
        // fp32 atomics into C, which is pre-set to the residual (or zero) by a device copy ahead of the launch.
        // Measured (3B, T = 512): the Wo/Wdown launches 49.5 -> 45.2 us, QKV 43.5 -> 44.6 us, plus the copy that
        // pre-sets C: 7.55 -> 7.75 ms per prefill.  Off by default (NFAI_GEMM_SPLITK=1 enables it).
        static const int env_sk = getenv("NFAI_GEMM_SPLITK") ? atoi(getenv("NFAI_GEMM_SPLITK")) : 0;
        const uint32_t KT = p.K / 64;
        if (env_sk && env_big && batch == 1 && p.N % 128 == 0 && p.M >= 128 && big_tiles < (uint64_t)n_cu * 3 / 2) {
            uint32_t ks = 0;
            for (uint32_t c : {2u, 3u, 4u, 6u, 8u})
                if (KT % c == 0 && KT / c >= 6 && big_tiles * c >= (uint64_t)n_cu * 3 / 2) { ks = c; break; }
            if (ks) {
                const size_t bytes = (size_t)p.M * p.ldc * sizeof(float);
                hipError_t e = hipSuccess;
                if (!p.R) e = hipMemsetAsync(p.C, 0, bytes, s);
                else if (static_cast<const void *>(p.R) != p.C) e = hipMemcpyAsync(p.C, p.R, bytes, hipMemcpyDeviceToDevice, s);
                if (e != hipSuccess) return e;
                p.ksplit = ks;
                p.R = nullptr;
                return gemm_launch<128, 128, 2, 2, 64, EPI_F32>(p, batch, s);
            }
        }
    }
    // wide N: direct-to-LDS staging with 2 stages (64 KB: two workgroups per CU).  Measured end to end (3B, T = 512, three
    // runs each): 7.20 ms against 7.37 ms for the register-staged 128 x 128 kernel and 8.0 ms for 3 stages (96 KB: one
    // workgroup per CU — occupancy beats prefetch depth here).  NFAI_GEMM_GLDS=0 selects the register-staged kernel.
    static const int env_glds = getenv("NFAI_GEMM_GLDS") ? atoi(getenv("NFAI_GEMM_GLDS")) : 2;
    // Round 3: eight waves on a 256 x 128 tile (wave tile 64 x 64 as before), three stages (144 KB, one workgroup per CU): per
    // CU the same eight waves as two 128 x 128 workgroups, but 48 KB instead of 64 KB of operands per 64-deep K tile through the
    // CU's load path, which is what bounds these kernels (tools/gemm_bench.py --cold, 512 rows: gate|up of 3B 67.3 -> 58.0 us,
    // of 1B 43.1 -> 41.5, of 8B 167.7 -> 166.6).  Needs an even number of 128-row blocks.  NFAI_GEMM_W8=0: the 128 x 128 form.
    static const int env_w8 = getenv("NFAI_GEMM_W8") ? atoi(getenv("NFAI_GEMM_W8")) : 1;
    if (env_big && env_w8 && env_glds == 2 && p.ksplit == 1 && batch == 1 && p.causal == 0 && p.N % 128 == 0 && ((p.M + 127) / 128) % 2 == 0 &&
        (uint64_t)((p.M + 255) / 256) * (p.N / 128) >= (uint64_t)n_cu * 3 / 4)
        return gemm_launch_glds<256, 128, 4, 2, EPI, 3, 64, true>(p, batch, s);  // + pipelined fragment reads: 61.1 -> 57.9 us
    if (env_big && p.N % 128 == 0 && big_tiles >= (uint64_t)n_cu * 3 / 2) {
        if (env_glds == 3 && p.ksplit == 1) return gemm_launch_glds<128, 128, 2, 2, EPI, 3>(p, batch, s);
        if (env_glds == 2 && p.ksplit == 1) return gemm_launch_glds<128, 128, 2, 2, EPI, 2>(p, batch, s);
        return gemm_launch<128, 128, 2, 2, 64, EPI>(p, batch, s);
    }
    // at most ~one workgroup per CU: nothing else hides latency, so twice the bytes in flight and half the barriers
    static const int env_bk = getenv("NFAI_GEMM_BK128") ? atoi(getenv("NFAI_GEMM_BK128")) : 0;  // measured: 3 % slower than BK = 64
    const uint64_t small_tiles = (uint64_t)((p.M + 127) / 128) * (p.N / 64) * batch;
    if (env_bk && p.K % 128 == 0 && small_tiles <= (uint64_t)n_cu * 3 / 2) return gemm_launch<128, 64, 4, 1, 128, EPI>(p, batch, s);
    static const int env_ks = getenv("NFAI_GEMM_KS") ? atoi(getenv("NFAI_GEMM_KS")) : 0;  // measured: no gain (the limit is bytes in flight)
    if (env_ks && small_tiles <= (uint64_t)n_cu * 3 / 2) return gemm_launch<128, 64, 4, 1, 64, EPI, 2>(p, batch, s);
    static const int env_m64 = getenv("NFAI_GEMM_M64") ? atoi(getenv("NFAI_GEMM_M64")) : 0;  // measured: 15 % slower than 128 x 64 on 192 workgroups
    if (env_m64 && small_tiles < n_cu && p.M > 64) return gemm_launch<64, 64, 2, 1, 64, EPI>(p, batch, s);  // fewer tiles than CUs: halve the M tile
    // narrow N (192-320 workgroups of 128 x 64): direct-to-LDS with 3 stages (72 KB).  Measured end to end, three runs
    // each: 7.06 ms against 7.18 ms register-staged; 2 stages 8.27 ms, 4 stages 7.42 ms.  0 = register staging.
    static const int env_glds_n = getenv("NFAI_GEMM_GLDS_NARROW") ? atoi(getenv("NFAI_GEMM_GLDS_NARROW")) : 3;
    // Tile width of the narrow configuration: the one that needs the fewest rounds of workgroups over the CUs, then the fewest
    // operand bytes per workgroup (cost = rounds x (BM + BN)).  At 512 rows: N = 5120 (q|k|v of 3B) -> 80 (256 workgroups instead
    // of 320 = 1.25 rounds: 43.2 -> 28.9 us with cold weights), N = 3072 -> 48 (256 instead of 192: 27.2 -> 24.0 us, K = 8192:
    // 64.5 -> 56.6 us), N = 6144 (q|k|v of 8B) -> 96.  tools/gemm_bench.py --cold; NFAI_GEMM_BN=64 keeps the fixed width.
    static const int env_bn = getenv("NFAI_GEMM_BN") ? atoi(getenv("NFAI_GEMM_BN")) : 0;
    if constexpr (EPI != EPI_SILU) {
        if (env_glds_n == 3 && p.ksplit == 1 && batch == 1 && p.causal == 0 && env_bn != 64) {
            const uint64_t tm = (p.M + 127) / 128;
            auto cost = [&](uint32_t bn) { return ((tm * (p.N / bn) + n_cu - 1) / n_cu) * (128 + bn); };
            uint32_t best = 64;
            uint64_t best_cost = cost(64);
            for (uint32_t bn : {48u, 80u, 96u}) {
                if (env_bn && (uint32_t)env_bn != bn) continue;
                if (p.N % bn == 0 && cost(bn) * 100 < best_cost * 97) { best = bn; best_cost = cost(bn); }
            }
            // Round 3: a second group of four waves on the same tile; the groups split every K tile's k-steps and add their
            // accumulators in LDS at the end (these launches have one workgroup per CU: the second wave per SIMD fills the first one's
            // waits).  BK = 128 where three stages fit the LDS, else four stages of 64 (3B at 512 rows: Wdown 55.3 -> 46.9 us, Wo 23.8
            // -> 21.4, q|k|v 29.0 -> 27.5; 8B: q|k|v 47.5 -> 37.8, Wo 35.3 -> 29.3, Wdown 116 -> 91; 1B: 17.3 -> 15.9, 18.3 -> 16.0,
            // 59.7 -> 47.4).  NFAI_GEMM_KS2=0: the four-wave form.
            static const int env_ks2 = getenv("NFAI_GEMM_KS2") ? atoi(getenv("NFAI_GEMM_KS2")) : 1;
            const bool bk128 = p.K % 128 == 0;
            if (env_ks2) {
                if (best == 48) return bk128 ? gemm_launch_glds<128, 48, 4, 1, EPI, 3, 128, false, 3, 2>(p, batch, s) : gemm_launch_glds<128, 48, 4, 1, EPI, 4, 64, false, 4, 2>(p, batch, s);
                if (best == 80) return bk128 ? gemm_launch_glds<128, 80, 4, 1, EPI, 3, 128, true, 3, 2>(p, batch, s) : gemm_launch_glds<128, 80, 4, 1, EPI, 4, 64, false, 4, 2>(p, batch, s);
                if (best == 96) return gemm_launch_glds<128, 96, 4, 1, EPI, 4, 64, false, 4, 2>(p, batch, s);
                return bk128 ? gemm_launch_glds<128, 64, 4, 1, EPI, 3, 128, false, 3, 2>(p, batch, s) : gemm_launch_glds<128, 64, 4, 1, EPI, 4, 64, false, 4, 2>(p, batch, s);
            }
            if (best == 48) return gemm_launch_glds<128, 48, 4, 1, EPI, 3>(p, batch, s);
            if (best == 80) return gemm_launch_glds<128, 80, 4, 1, EPI, 3>(p, batch, s);
            if (best == 96) return gemm_launch_glds<128, 96, 4, 1, EPI, 3>(p, batch, s);
        }
    }
    if (env_glds_n == 2 && p.ksplit == 1) return gemm_launch_glds<128, 64, 4, 1, EPI, 2>(p, batch, s);
    if (env_glds_n == 3 && p.ksplit == 1) return gemm_launch_glds<128, 64, 4, 1, EPI, 3>(p, batch, s);
    if (env_glds_n == 4 && p.ksplit == 1) return gemm_launch_glds<128, 64, 4, 1, EPI, 4>(p, batch, s);
    static const int env_ring = getenv("NFAI_GEMM_RING") ? atoi(getenv("NFAI_GEMM_RING")) : 3;  // measured: 5 is 5 % slower
    if (env_ring == 5 && small_tiles <= (uint64_t)n_cu * 3 / 2) return gemm_launch<128, 64, 4, 1, 64, EPI, 1, 5>(p, batch, s);
    return gemm_launch<128, 64, 4, 1, 64, EPI>(p, batch, s);
}

// (Round 4 also built a "skinny" GEMM for <= 16 rows — shaped like the decode GEMV: 16 weight rows per workgroup, eight waves splitting K,
// weights straight into the MFMA's B fragments — and a 64-row form of it with K splits and a ticketed combine.  The 64-row tiles below plus
// the K split of Wo / Wdown (llama.hip) are as fast or faster at every row count (1 / 8 / 16 tokens at 3B: 2.34 / 2.34 / 2.34 ms against
// 2.34 / 2.40 / 2.50), so both were removed: `git show 9ef9490:nfai_amd/csrc/kernels_prefill.hip`, profiles/round4_ingest_latency*.txt.)

// cos / sin of positions pos0 .. pos0+T-1 for the pairs below rope_dims (the arithmetic of k_rope_store_tiles: theta = freq * pos in
// fp32, cosf / sinf), once per prompt chunk: every block's q | k | v epilogue reads it.
__global__ void k_rope_table(const float *freqs, uint32_t pos0, uint32_t half, uint32_t rope_dims, float *cs)
{
    const uint32_t t = blockIdx.x, pr = threadIdx.x;
    if (pr >= half) return;
    float c = 1.f, sn = 0.f;
    if (pr * 2 < rope_dims) {
        const float theta = freqs[pr] * (float)(pos0 + t);
        c = cosf(theta);
        sn = sinf(theta);
    }
    *reinterpret_cast<f32x2 *>(cs + ((uint64_t)t * half + pr) * 2) = f32x2{c, sn};
}

hipError_t launch_rope_table(const float *freqs, uint32_t pos0, uint32_t T, uint32_t D, uint32_t rope_dims, float *cs, hipStream_t s)
{
    if (T == 0) return hipSuccess;
    if (D < 2 || D % 2 || D / 2 > 1024) return hipErrorInvalidValue;
    k_rope_table<<<T, (D / 2 + 63) / 64 * 64, 0, s>>>(freqs, pos0, D / 2, rope_dims, cs);
    return hipGetLastError();
}

// the q | k | v projection with the RoPE epilogue: the narrow configurations only (N = (H + 2 Hkv) D is a few thousand columns)
static hipError_t gemm_pick_rope(GemmParams &p, uint32_t n_cu, hipStream_t s)
{
    static const bool m64_short = !(getenv("NFAI_GEMM_M64_SHORT") && atoi(getenv("NFAI_GEMM_M64_SHORT")) == 0);
    if (m64_short && p.M <= 64)   // short prompts: 64-row tiles (gemm_pick)
        return p.N % 80 == 0 ? gemm_launch_glds<64, 80, 4, 1, EPI_ROPE, 4>(p, 1, s) : gemm_launch_glds<64, 64, 4, 1, EPI_ROPE, 4>(p, 1, s);
    const uint64_t tm = (p.M + 127) / 128;
    auto cost = [&](uint32_t bn) { return ((tm * (p.N / bn) + n_cu - 1) / n_cu) * (128 + bn); };
    uint32_t best = 64;
    uint64_t best_cost = cost(64);
    for (uint32_t bn : {48u, 80u, 96u})
        if (p.N % bn == 0 && cost(bn) * 100 < best_cost * 97) { best = bn; best_cost = cost(bn); }
    const bool bk128 = p.K % 128 == 0;
    if (best == 48) return bk128 ? gemm_launch_glds<128, 48, 4, 1, EPI_ROPE, 3, 128, false, 3, 2>(p, 1, s) : gemm_launch_glds<128, 48, 4, 1, EPI_ROPE, 4, 64, false, 4, 2>(p, 1, s);
    if (best == 80) return bk128 ? gemm_launch_glds<128, 80, 4, 1, EPI_ROPE, 3, 128, true, 3, 2>(p, 1, s) : gemm_launch_glds<128, 80, 4, 1, EPI_ROPE, 4, 64, false, 4, 2>(p, 1, s);
    if (best == 96) return gemm_launch_glds<128, 96, 4, 1, EPI_ROPE, 4, 64, false, 4, 2>(p, 1, s);
    return bk128 ? gemm_launch_glds<128, 64, 4, 1, EPI_ROPE, 3, 128, false, 3, 2>(p, 1, s) : gemm_launch_glds<128, 64, 4, 1, EPI_ROPE, 4, 64, false, 4, 2>(p, 1, s);
}

hipError_t launch_gemm_f16(const GemmArgs &a, hipStream_t s)
{
    if (a.M == 0 || a.N == 0) return hipSuccess;
    if (a.N % 64 != 0 || a.K % 64 != 0 || a.K == 0) return hipErrorInvalidValue;
    if (a.lda % 8 != 0 || a.ldb % 8 != 0 || a.a_f32) return hipErrorInvalidValue;
    GemmParams p{};
    p.A = static_cast<const _Float16 *>(a.A);
    p.B[0] = static_cast<const _Float16 *>(a.B);
    p.B[1] = static_cast<const _Float16 *>(a.B1 ? a.B1 : a.B);
    p.B[2] = static_cast<const _Float16 *>(a.B2 ? a.B2 : a.B);
    p.seg_end[0] = a.B1 ? a.n0 : a.N;
    p.seg_end[1] = a.B2 ? a.n0 + a.n1 : a.N;
    p.seg_end[2] = a.N;
    if ((p.seg_end[0] | p.seg_end[1]) % 128 && a.epi != EPI_SILU) {
        if ((p.seg_end[0] | p.seg_end[1]) % 64) return hipErrorInvalidValue;
    }
    p.C = a.C; p.R = a.R;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldb = a.ldb; p.ldc = a.ldc;
    p.a_bs = a.a_bs; p.b_bs = a.b_bs; p.c_bs = a.c_bs; p.b_div = a.b_div ? a.b_div : 1; p.alpha = a.alpha;
    p.ksplit = 1;
    p.causal = a.causal; p.causal_pos0 = a.causal_pos0;
    const uint32_t batch = a.batch ? a.batch : 1;
    const uint32_t n_cu = a.n_cu ? a.n_cu : 256;
    if (a.epi == EPI_ROPE) {
        const GemmRope &r = a.rope;
        if (!a.B1 || !a.B2 || batch != 1 || a.R || a.causal || !r.cs || !r.qh || !r.kh || !r.vt || !r.kc || !r.vc) return hipErrorInvalidValue;
        if (r.D == 0 || r.D % 16 || r.rope_dims % 2 || r.rope_dims > r.D || r.H == 0 || r.Hkv == 0) return hipErrorInvalidValue;
        if (a.n0 != r.H * r.D || a.n1 != r.Hkv * r.D || a.N != (r.H + 2 * r.Hkv) * r.D) return hipErrorInvalidValue;
        if ((p.seg_end[0] | p.seg_end[1] | a.N) % 16) return hipErrorInvalidValue;
        p.rope = r;
        return gemm_pick_rope(p, n_cu, s);
    }
    if (a.epi == EPI_SILU) {
        // N counts gate + up columns; the two segments must be equally long and the output is [M][N/2] fp16
        if (!a.B1 || a.n0 * 2 != a.N || a.R || batch != 1) return hipErrorInvalidValue;
        return gemm_pick<EPI_SILU>(p, batch, n_cu, a.variant, s);
    }
    // a segment boundary inside a 128-wide tile is not supported by the wide configuration: fall back
    if ((p.seg_end[0] | p.seg_end[1]) % 128) {
        if (a.epi == EPI_F16) return a.R ? hipErrorInvalidValue : gemm_launch<128, 64, 4, 1, 64, EPI_F16>(p, batch, s);
        return gemm_launch<128, 64, 4, 1, 64, EPI_F32>(p, batch, s);
    }
    if (a.epi == EPI_F16) return a.R ? hipErrorInvalidValue : gemm_pick<EPI_F16>(p, batch, n_cu, a.variant, s);
    return gemm_pick<EPI_F32>(p, batch, n_cu, a.variant, s);
}

// C = R + slab 0 + slab 1 + ... (fixed order): the combine of a projection whose K range was split over workgroups (short prompts,
// llama.hip prefill_chunk); n = M * N elements per slab, a multiple of 4.
__global__ __launch_bounds__(256) void k_sum_slabs(const float *slabs, uint32_t ks, uint64_t n, const float *R, float *C)
{
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 v = R ? *reinterpret_cast<const f32x4 *>(R + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (uint32_t z = 0; z < ks; z++) v += *reinterpret_cast<const f32x4 *>(slabs + (uint64_t)z * n + i);
    *reinterpret_cast<f32x4 *>(C + i) = v;
}

hipError_t launch_sum_slabs(const float *slabs, uint32_t ks, uint64_t n, const float *R, float *C, hipStream_t s)
{
    if (n == 0 || n % 4 || ks == 0) return hipErrorInvalidValue;
    k_sum_slabs<<<(uint32_t)((n / 4 + 255) / 256), 256, 0, s>>>(slabs, ks, n, R, C);
    return hipGetLastError();
}

// fp32 rows -> fp16 (the attention output on its way into the Wo GEMM)
__global__ void k_f32_to_f16(const float *x, _Float16 *y, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (_Float16)x[i];
}

hipError_t launch_f32_to_f16(const float *x, void *y_f16, uint64_t n, hipStream_t s)
{
    k_f32_to_f16<<<(uint32_t)((n + 255) / 256), 256, 0, s>>>(x, static_cast<_Float16 *>(y_f16), n);
    return hipGetLastError();
}

// ---- row-batched small ops ------------------------------------------------------------------------
// RMSNormShader over T rows, output fp16 (the next GEMM's A operand).  One block per row; rows of up to
// 256 * 4 * NV floats stay in registers between the two passes (16-byte loads, 8-byte fp16 stores).
template <int NV>
__global__ __launch_bounds__(256) void k_rmsnorm_rows_v(const float *x, const float *g, _Float16 *y, uint32_t E, float eps)
{
    __shared__ float red[16];
    const GLOBAL_AS float *xr = (const GLOBAL_AS float *)x + (uint64_t)blockIdx.x * E;
    const GLOBAL_AS float *gr = (const GLOBAL_AS float *)g;
    _Float16 *yr = y + (uint64_t)blockIdx.x * E;
    f32x4 v[NV], gv[NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const uint32_t k = (threadIdx.x + i * 256) * 4, kk = min(k, E - 4);
        const f32x4 a = *reinterpret_cast<const GLOBAL_AS f32x4 *>(xr + kk);
        gv[i] = *reinterpret_cast<const GLOBAL_AS f32x4 *>(gr + kk);
        v[i] = k < E ? a : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; e++) ss = fmaf(v[i][e], v[i][e], ss);
    }
    ss = block_sum(ss, red);
    const float rms = sqrtf(ss / (float)E + eps);
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const uint32_t k = (threadIdx.x + i * 256) * 4;
        if (k < E) {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (_Float16)((v[i][e] / rms) * gv[i][e]);
            *reinterpret_cast<f16x4 *>(yr + k) = o;
        }
    }
}

// The same behind a projection whose K range was split (llama.hip, Wdown at >= 256 rows): the row is residual + slab 0 + slab 1 + ...
// (the order of k_sum_slabs: bit-identical to the two separate launches), written out as the fp32 residual stream AND normalised.
template <int NV>
__global__ __launch_bounds__(256) void k_rmsnorm_rows_combine(const float *slabs, uint32_t ks, uint64_t slab_stride, const float *R, float *xo, const float *g,
                                                              _Float16 *y, uint32_t E, float eps)
{
    __shared__ float red[16];
    const uint64_t row = (uint64_t)blockIdx.x * E;
    const GLOBAL_AS float *gr = (const GLOBAL_AS float *)g;
    f32x4 v[NV], gv[NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const uint32_t k = (threadIdx.x + i * 256) * 4, kk = min(k, E - 4);
        f32x4 a = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)R + row + kk);
        gv[i] = *reinterpret_cast<const GLOBAL_AS f32x4 *>(gr + kk);
        for (uint32_t z = 0; z < ks; z++) a += *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)slabs + z * slab_stride + row + kk);
        v[i] = k < E ? a : f32x4{0.f, 0.f, 0.f, 0.f};
        if (k < E) *reinterpret_cast<f32x4 *>(xo + row + k) = a;
#pragma unroll
        for (int e = 0; e < 4; e++) ss = fmaf(v[i][e], v[i][e], ss);
    }
    ss = block_sum(ss, red);
    const float rms = sqrtf(ss / (float)E + eps);
    _Float16 *yr = y + row;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const uint32_t k = (threadIdx.x + i * 256) * 4;
        if (k < E) {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (_Float16)((v[i][e] / rms) * gv[i][e]);
            *reinterpret_cast<f16x4 *>(yr + k) = o;
        }
    }
}

hipError_t launch_rmsnorm_rows_combine(const float *slabs, uint32_t ks, const float *R, float *x_out, const float *g, void *y_f16, uint32_t T, uint32_t E, float eps,
                                       hipStream_t s)
{
    _Float16 *y = static_cast<_Float16 *>(y_f16);
    if (E % 4 || E < 4 || E > 4096 || ks == 0 || !R) return hipErrorInvalidValue;
    const uint64_t stride = (uint64_t)T * E;
    if (E <= 1024) k_rmsnorm_rows_combine<1><<<T, 256, 0, s>>>(slabs, ks, stride, R, x_out, g, y, E, eps);
    else if (E <= 2048) k_rmsnorm_rows_combine<2><<<T, 256, 0, s>>>(slabs, ks, stride, R, x_out, g, y, E, eps);
    else if (E <= 3072) k_rmsnorm_rows_combine<3><<<T, 256, 0, s>>>(slabs, ks, stride, R, x_out, g, y, E, eps);
    else k_rmsnorm_rows_combine<4><<<T, 256, 0, s>>>(slabs, ks, stride, R, x_out, g, y, E, eps);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_rmsnorm_rows(const float *x, const float *g, _Float16 *y, uint32_t E, float eps)
{
    __shared__ float red[16];
    const float *xr = x + (uint64_t)blockIdx.x * E;
    _Float16 *yr = y + (uint64_t)blockIdx.x * E;
    float ss = 0.f;
    for (uint32_t i = threadIdx.x; i < E; i += blockDim.x) ss = fmaf(xr[i], xr[i], ss);
    ss = block_sum(ss, red);
    const float rms = sqrtf(ss / (float)E + eps);
    for (uint32_t i = threadIdx.x; i < E; i += blockDim.x) yr[i] = (_Float16)((xr[i] / rms) * g[i]);
}

hipError_t launch_rmsnorm_rows(const float *x, const float *g, void *y_f16, uint32_t T, uint32_t E, float eps, hipStream_t s)
{
    _Float16 *y = static_cast<_Float16 *>(y_f16);
    if (E % 4 == 0 && E >= 4 && E <= 4096) {
        if (E <= 1024) k_rmsnorm_rows_v<1><<<T, 256, 0, s>>>(x, g, y, E, eps);
        else if (E <= 2048) k_rmsnorm_rows_v<2><<<T, 256, 0, s>>>(x, g, y, E, eps);
        else if (E <= 3072) k_rmsnorm_rows_v<3><<<T, 256, 0, s>>>(x, g, y, E, eps);
        else k_rmsnorm_rows_v<4><<<T, 256, 0, s>>>(x, g, y, E, eps);
    } else {
        k_rmsnorm_rows<<<T, 256, 0, s>>>(x, g, y, E, eps);
    }
    return hipGetLastError();
}

// RoPE on T rows of q (-> fp16) and k (-> KV cache rows pos0+t), v -> cache rows.  One thread per pair.
// kh / vt (optional): the fp16 K rows [Hkv][Spad][D] and V^T [Hkv][D][Spad] the attention GEMMs read are written here
// for the chunk's own positions, so k_kv_to_f16 only has to convert positions < pos0 (nothing for a first chunk).
__global__ void k_rope_store_rows(const float *q, const float *k, const float *v, _Float16 *qh, void *kc, void *vc, int kv_f16,
                                  uint64_t pos_stride, uint64_t head_stride, const float *freqs, uint32_t rope_dims, uint32_t H,
                                  uint32_t Hkv, uint32_t D, uint32_t pos0, uint32_t ld, _Float16 *kh, _Float16 *vt, uint32_t Spad)
{
    const uint32_t t = blockIdx.y, half = D / 2;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nq = H * half, nk = Hkv * half;
    if (idx >= nq + 2 * nk) return;
    const uint32_t pos = pos0 + t;
    const uint32_t which = idx < nq ? 0u : (idx < nq + nk ? 1u : 2u);
    const uint32_t li = which == 0 ? idx : (which == 1 ? idx - nq : idx - nq - nk);
    const uint32_t h = li / half, pair = (li % half) * 2;
    const float *src = (which == 0 ? q : (which == 1 ? k : v)) + (uint64_t)t * ld;  // q | k | v may be column blocks of one [T][ld] buffer
    const float a = src[h * D + pair], b = src[h * D + pair + 1];
    float o0 = a, o1 = b;
    if (which < 2 && pair < rope_dims) {
        const float theta = freqs[pair / 2] * (float)pos;
        const float c = cosf(theta), sn = sinf(theta);
        o0 = c * a - sn * b;
        o1 = sn * a + c * b;
    }
    if (which == 0) {
        qh[(uint64_t)t * H * D + h * D + pair] = (_Float16)o0;
        qh[(uint64_t)t * H * D + h * D + pair + 1] = (_Float16)o1;
    } else {
        void *base = which == 1 ? kc : vc;
        const uint64_t o = (uint64_t)pos * pos_stride + (uint64_t)h * head_stride + pair;
        if (kv_f16) {
            reinterpret_cast<_Float16 *>(base)[o] = (_Float16)o0;
            reinterpret_cast<_Float16 *>(base)[o + 1] = (_Float16)o1;
        } else {
            reinterpret_cast<float *>(base)[o] = o0;
            reinterpret_cast<float *>(base)[o + 1] = o1;
        }
        if (kh) {
            if (which == 1) {
                _Float16 *dst = kh + ((uint64_t)h * Spad + pos) * D + pair;
                dst[0] = (_Float16)o0;
                dst[1] = (_Float16)o1;
            } else {
                vt[((uint64_t)h * D + pair) * Spad + pos] = (_Float16)o0;
                vt[((uint64_t)h * D + pair + 1) * Spad + pos] = (_Float16)o1;
            }
        }
    }
}

// The same work in tiles of 32 positions x one head (workgroup = 256 threads): rows are read and written 8 bytes per thread along d
// (coalesced), and the fp16 V^T of the chunk — rows along the positions — goes through LDS so that it is written 32 bytes per
// thread along the positions instead of two bytes at a stride of Spad (the scattered stores were what the per-pair kernel above
// spent its 9.5 us per block on at 3B, T = 512; it stays for head sizes above 128 and as the reference form).
__global__ __launch_bounds__(256) void k_rope_store_tiles(const float *q, const float *k, const float *v, _Float16 *qh, void *kc, void *vc,
                                                          int kv_f16, uint64_t pos_stride, uint64_t head_stride, const float *freqs,
                                                          uint32_t rope_dims, uint32_t H, uint32_t Hkv, uint32_t D, uint32_t pos0, uint32_t T,
                                                          uint32_t ld, _Float16 *kh, _Float16 *vt, uint32_t Spad)
{
    __shared__ _Float16 vtile[128][32 + 8];  // [d][position of the tile], padded rows
    const uint32_t half = D / 2, rpp = 256 / half;  // rows per pass
    const uint32_t hh = blockIdx.y, t0 = blockIdx.x * 32, tid = threadIdx.x;
    const uint32_t which = hh < H ? 0u : (hh < H + Hkv ? 1u : 2u);
    const uint32_t h = which == 0 ? hh : (which == 1 ? hh - H : hh - H - Hkv);
    const float *src0 = which == 0 ? q : (which == 1 ? k : v);
    const uint32_t pr = tid % half, rr = tid / half, pair = pr * 2;
    const bool rot = which < 2 && pair < rope_dims;
    const float fr = rot ? freqs[pr] : 0.f;
    for (uint32_t r0 = 0; r0 < 32; r0 += rpp) {
        const uint32_t tl = r0 + rr, t = t0 + tl;
        if (rr < rpp && t < T) {
            const uint32_t pos = pos0 + t;
            const f32x2 ab = *reinterpret_cast<const f32x2 *>(src0 + (uint64_t)t * ld + h * D + pair);
            float o0 = ab[0], o1 = ab[1];
            if (rot) {  // RoPEShader.cs:249-262 (a per-chunk cos / sin table instead of cosf / sinf here was measured: no change)
                const float theta = fr * (float)pos;
                const float c = cosf(theta), sn = sinf(theta);
                o0 = c * ab[0] - sn * ab[1];
                o1 = sn * ab[0] + c * ab[1];
            }
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 oh = {(_Float16)o0, (_Float16)o1};
            if (which == 0) {
                *reinterpret_cast<h2 *>(qh + (uint64_t)t * H * D + h * D + pair) = oh;
            } else {
                void *base = which == 1 ? kc : vc;
                const uint64_t o = (uint64_t)pos * pos_stride + (uint64_t)h * head_stride + pair;
                if (kv_f16) *reinterpret_cast<h2 *>(reinterpret_cast<_Float16 *>(base) + o) = oh;
                else *reinterpret_cast<f32x2 *>(reinterpret_cast<float *>(base) + o) = f32x2{o0, o1};
                if (kh) {
                    if (which == 1) {
                        *reinterpret_cast<h2 *>(kh + ((uint64_t)h * Spad + pos) * D + pair) = oh;
                    } else {
                        vtile[pair][tl] = oh[0];
                        vtile[pair + 1][tl] = oh[1];
                    }
                }
            }
        }
    }
    if (which == 2 && kh) {
        __syncthreads();
        // V^T rows of the tile: thread = (d, half of the 32 positions); 16 positions = 32 bytes when the destination is 16-byte
        // aligned and the tile is full, else element by element
        const uint32_t d = tid / 2, hp = tid % 2;
        if (d < D) {
            _Float16 *dst = vt + ((uint64_t)h * D + d) * Spad + pos0 + t0 + hp * 16;
            const uint32_t nvalid = t0 + hp * 16 < T ? min(16u, T - (t0 + hp * 16)) : 0u;
            if (nvalid == 16 && ((pos0 + t0) % 8) == 0) {
                const u32x4 a = *reinterpret_cast<const u32x4 *>(&vtile[d][hp * 16]);
                const u32x4 b = *reinterpret_cast<const u32x4 *>(&vtile[d][hp * 16 + 8]);
                *reinterpret_cast<u32x4 *>(dst) = a;
                *reinterpret_cast<u32x4 *>(dst + 8) = b;
            } else {
                for (uint32_t i = 0; i < nvalid; i++) dst[i] = vtile[d][hp * 16 + i];
            }
        }
    }
}

hipError_t launch_rope_store_rows(const float *q, const float *k, const float *v, void *qh, void *kc, void *vc, int kv_f16,
                                  uint64_t pos_stride, uint64_t head_stride, const float *freqs, uint32_t rope_dims, uint32_t H,
                                  uint32_t Hkv, uint32_t D, uint32_t pos0, uint32_t T, uint32_t ld, void *kh, void *vt, uint32_t Spad,
                                  hipStream_t s)
{
    static const int env_tiles = getenv("NFAI_PREFILL_ROPE_TILES") ? atoi(getenv("NFAI_PREFILL_ROPE_TILES")) : 1;
    // tiles: D/2 pairs must divide the 256 threads of a workgroup, rows 8-byte aligned
    if (env_tiles && D <= 128 && D >= 8 && 256 % (D / 2) == 0 && D % 2 == 0 && ld % 2 == 0 && (pos_stride % 2) == 0 && (head_stride % 2) == 0) {
        k_rope_store_tiles<<<dim3((T + 31) / 32, H + 2 * Hkv), 256, 0, s>>>(q, k, v, static_cast<_Float16 *>(qh), kc, vc, kv_f16, pos_stride,
                                                                          head_stride, freqs, rope_dims, H, Hkv, D, pos0, T, ld,
                                                                          static_cast<_Float16 *>(kh), static_cast<_Float16 *>(vt), Spad);
        return hipGetLastError();
    }
    const uint32_t n = (H + 2 * Hkv) * D / 2;
    k_rope_store_rows<<<dim3((n + 255) / 256, T), 256, 0, s>>>(q, k, v, static_cast<_Float16 *>(qh), kc, vc, kv_f16, pos_stride,
                                                               head_stride, freqs, rope_dims, H, Hkv, D, pos0, ld,
                                                               static_cast<_Float16 *>(kh), static_cast<_Float16 *>(vt), Spad);
    return hipGetLastError();
}

// KV cache (fp32 or fp16, strided) -> fp16 K [Hkv][Spad][D] and V^T [Hkv][D][Spad] for positions < S
// (rows S..Spad-1 / columns S..Spad-1 are zero so the padded GEMMs see zeros).
// skip_lo..skip_hi-1: positions another kernel fills (the chunk's own rows, written by k_rope_store_rows)
__global__ void k_kv_to_f16(const void *kc, const void *vc, int kv_f16, uint64_t pos_stride, uint64_t head_stride, _Float16 *kh,
                            _Float16 *vt, uint32_t Hkv, uint32_t D, uint32_t S, uint32_t Spad, uint32_t skip_lo, uint32_t skip_hi)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = (uint64_t)Hkv * Spad * D;
    if (i >= n) return;
    const uint32_t d = i % D, sp = (i / D) % Spad, h = i / ((uint64_t)D * Spad);
    if (sp >= skip_lo && sp < skip_hi) return;
    float kvv = 0.f, vvv = 0.f;
    if (sp < S) {
        const uint64_t o = (uint64_t)sp * pos_stride + (uint64_t)h * head_stride + d;
        if (kv_f16) {
            kvv = (float)reinterpret_cast<const _Float16 *>(kc)[o];
            vvv = (float)reinterpret_cast<const _Float16 *>(vc)[o];
        } else {
            kvv = reinterpret_cast<const float *>(kc)[o];
            vvv = reinterpret_cast<const float *>(vc)[o];
        }
    }
    kh[i] = (_Float16)kvv;
    vt[((uint64_t)h * D + d) * Spad + sp] = (_Float16)vvv;
}

hipError_t launch_kv_to_f16(const void *kc, const void *vc, int kv_f16, uint64_t pos_stride, uint64_t head_stride, void *kh, void *vt,
                            uint32_t Hkv, uint32_t D, uint32_t S, uint32_t Spad, uint32_t skip_lo, uint32_t skip_hi, hipStream_t s)
{
    if (skip_lo == 0 && skip_hi >= Spad) return hipSuccess;  // a first chunk that fills the padded length exactly
    const uint64_t n = (uint64_t)Hkv * Spad * D;
    k_kv_to_f16<<<(uint32_t)((n + 255) / 256), 256, 0, s>>>(kc, vc, kv_f16, pos_stride, head_stride, static_cast<_Float16 *>(kh),
                                                            static_cast<_Float16 *>(vt), Hkv, D, S, Spad, skip_lo, skip_hi);
    return hipGetLastError();
}

// causal softmax of one score row: query t (absolute position pos0 + t) attends keys 0..pos0+t.
// scores [H][T][Spad] fp32 (unscaled dot products) -> P fp16, zero beyond the causal limit.
// AttentionSoftmaxShader.cs:148-177 semantics (max, exp(clamp(.,-80,80)), sum, 1/sum).
// Fast form: one wave per row, the row (Spad <= 256 * NV) in registers, wave reductions only.
template <int NV>
__global__ __launch_bounds__(256) void k_softmax_causal_rows_w(const float *sc, _Float16 *p, uint32_t rows, uint32_t T, uint32_t Spad,
                                                               uint32_t pos0, float scale)
{
    const uint32_t rowi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (rowi >= rows) return;
    const uint32_t t = rowi % T, n = pos0 + t + 1;
    const GLOBAL_AS float *row = (const GLOBAL_AS float *)sc + (uint64_t)rowi * Spad;
    _Float16 *prow = p + (uint64_t)rowi * Spad;
    f32x4 v[NV];
    float m = -1.0e38f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const uint32_t k = (lane + i * 64) * 4;
        v[i] = *reinterpret_cast<const GLOBAL_AS f32x4 *>(row + min(k, Spad - 4));
#pragma unroll
        for (int e = 0; e < 4; e++) {
            v[i][e] *= scale;
            if (k + e < n) m = fmaxf(m, v[i][e]);
        }
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const uint32_t k = (lane + i * 64) * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float ex = expf(fminf(fmaxf(v[i][e] - m, -80.f), 80.f));
            v[i][e] = k + e < n ? ex : 0.f;
            sum += v[i][e];
        }
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const uint32_t k = (lane + i * 64) * 4;
        if (k < Spad) {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (_Float16)(v[i][e] * inv);
            *reinterpret_cast<f16x4 *>(prow + k) = o;
        }
    }
}

__global__ __launch_bounds__(256) void k_softmax_causal_rows(const float *sc, _Float16 *p, uint32_t T, uint32_t Spad, uint32_t pos0,
                                                             float scale)
{
    __shared__ float red[16];
    const uint32_t t = blockIdx.x % T;
    const float *row = sc + (uint64_t)blockIdx.x * Spad;
    _Float16 *prow = p + (uint64_t)blockIdx.x * Spad;
    const uint32_t n = pos0 + t + 1;
    float m = -1.0e38f;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, row[i] * scale);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) sum += expf(fminf(fmaxf(row[i] * scale - m, -80.f), 80.f));
    sum = block_sum(sum, red);
    const float inv = 1.0f / sum;
    for (uint32_t i = threadIdx.x; i < Spad; i += blockDim.x)
        prow[i] = i < n ? (_Float16)(expf(fminf(fmaxf(row[i] * scale - m, -80.f), 80.f)) * inv) : (_Float16)0.f;
}

hipError_t launch_softmax_causal_rows(const float *sc, void *p_f16, uint32_t H, uint32_t T, uint32_t Spad, uint32_t pos0, float scale,
                                      hipStream_t s)
{
    _Float16 *p = static_cast<_Float16 *>(p_f16);
    const uint32_t rows = H * T;
    if (Spad % 4 == 0 && Spad >= 4 && Spad <= 2048) {
        const uint32_t grid = (rows + 3) / 4;
        if (Spad <= 256) k_softmax_causal_rows_w<1><<<grid, 256, 0, s>>>(sc, p, rows, T, Spad, pos0, scale);
        else if (Spad <= 512) k_softmax_causal_rows_w<2><<<grid, 256, 0, s>>>(sc, p, rows, T, Spad, pos0, scale);
        else if (Spad <= 1024) k_softmax_causal_rows_w<4><<<grid, 256, 0, s>>>(sc, p, rows, T, Spad, pos0, scale);
        else k_softmax_causal_rows_w<8><<<grid, 256, 0, s>>>(sc, p, rows, T, Spad, pos0, scale);
    } else {
        k_softmax_causal_rows<<<rows, 256, 0, s>>>(sc, p, T, Spad, pos0, scale);
    }
    return hipGetLastError();
}

// act = up * silu(gate) over T*F elements, fp16 out (SiLUShader + ElementWiseMultiplicationShader)
// ---- causal attention of a prompt chunk in one launch (scores, softmax, weighted V: AttentionScoreCalculationShader /
// AttentionSoftmaxShader / AttentionWeightedValueSumShader for T query rows at once; the reference runs them per token) -----------
// Replaces Q.K^T GEMM + row softmax + P.V GEMM (25 MB of scores and 12 MB of probabilities per block at T = 512 never exist).
// Workgroup = 64 query rows of one head, one wave per 16 rows; 32 keys per step:
//   S^T = K . Q^T    (mfma 16x16x32: A = 16 keys x 32 d, B = the wave's Q rows, held in registers)
//        -> lane (q, g) holds keys g*4 .. g*4+3 of both 16-key tiles for query q: exactly the eight k-slots of lane group g of a
//           B fragment if the MFMA's k index is READ as (g, j) -> key g*4 + j (j < 4), 16 + g*4 + (j - 4) (j >= 4);
//   O^T += V^T . P^T (A = V^T rows d in the same key order: two 8-byte reads per fragment; B = the probabilities, from registers)
// so the probabilities never leave the registers they were computed in.  The K and V^T tiles of a step (8 KB each at D = 128) go
// global -> LDS once per workgroup with LDS-DMA (a ring of ATTN_PF_NST stages, counted vmcnt + raw s_barrier as in k_gemm_f16_glds; the swizzle
// that makes the fragment reads conflict-free is applied to the per-lane SOURCE address) — a first version in which every wave
// loaded its own fragments from L2 took 49.9 us per launch at 3B, T = 512: bound by what a CU can take in, not by arithmetic.
// Online softmax per query (running max, rescale), the reference's exp(clamp(s - max, -80, 80)) with the running max and the
// hardware exponential (v_exp_f32: the probabilities are rounded to fp16 anyway); fp32 accumulation, as the GEMM path.
#ifndef ATTN_PF_NST
#define ATTN_PF_NST 3  // stages of the K / V^T ring (measured per launch at 3B, T = 512: 3 stages 22.2 us, 6 stages 22.6 us — the step is
                       // bound by its dependent chain S-MFMAs -> max -> exp -> P.V-MFMAs with one wave per SIMD, not by the tiles' latency)
#endif
// KG = 2: two wave groups per 16-row tile split every 64 keys of a step between them (each its own 32-key sub-tile of K and V^T) and
// merge their (max, sum, output) through LDS at the end: the dependent chain of a step is walked half as often per wave.
template <int D, int KG>
__global__ __launch_bounds__(256 * KG) void k_attn_prefill(const _Float16 *QH, const _Float16 *KH, const _Float16 *VT, _Float16 *O, uint32_t T,
                                                      uint32_t H, uint32_t G, uint32_t Spad, uint32_t pos0, float scale
#ifdef NFAI_STAMPS
                                                      , unsigned long long *stamps
#endif
)
{
#ifdef NFAI_STAMPS
    // diagnostic build: shader-clock cycles this wave spends waiting for its tile, at the barrier, issuing, in scores + softmax, in P.V
    unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_s = 0, c_pv = 0, c0 = __builtin_amdgcn_s_memtime(), c1;
    const unsigned long long c_start = c0;
#define APF_TICK(acc_) do { __builtin_amdgcn_sched_barrier(0); c1 = __builtin_amdgcn_s_memtime(); acc_ += c1 - c0; c0 = c1; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define APF_TICK(acc_) ((void)0)
#endif
    constexpr int KK = D / 32, DT = D / 16, NST = KG > 2 ? 2 : ATTN_PF_NST;  // four key groups: two stages of 4 x 16 KB (D = 128)
    constexpr int KROW = D * 2;                  // bytes of a key row in the K tile (D / 8 chunks of 16 B)
    constexpr int KCH = D / 8;                   // chunks per key row
    constexpr int K_BYTES = 32 * KROW, V_BYTES = D * 64, SUB = K_BYTES + V_BYTES, STAGE = SUB * KG;
    constexpr int KI = K_BYTES / 1024, VI = V_BYTES / 1024;  // LDS-DMA instructions per tile (1 KiB each)
    static_assert(KI % 4 == 0 && VI % 4 == 0, "instructions divide over the four waves");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    const uint32_t lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wave = wave_all & 3, kg = wave_all >> 2;  // query tile of the workgroup, key group
    const uint32_t qi = lane & 15, g = lane >> 4;
    const uint32_t qblocks = (T + 63) / 64;
    const uint32_t qb = qblocks - 1 - blockIdx.x;  // long rows first
    const uint32_t q0 = (qb * 4 + wave) * 16;
    const uint32_t h = blockIdx.y, kvh = h / G, HD = H * D;
    const uint32_t qrow = min(q0 + qi, T - 1);
    f16x8 qf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; kk++)
        qf[kk] = *reinterpret_cast<const GLOBAL_AS f16x8 *>((const GLOBAL_AS _Float16 *)QH + (uint64_t)qrow * HD + h * D + kk * 32 + g * 8);
    const GLOBAL_AS uint8_t *Kb = (const GLOBAL_AS uint8_t *)(KH + (uint64_t)kvh * Spad * D);
    const GLOBAL_AS uint8_t *Vb = (const GLOBAL_AS uint8_t *)(VT + (uint64_t)kvh * D * Spad);
    const uint32_t kmax = pos0 + min(qb * 64 + 63, T - 1) + 1;  // keys the workgroup's last row can see (exclusive)
    const uint32_t nsteps = (kmax + 32 * KG - 1) / (32 * KG);   // block-uniform; a step covers 32 keys per key group
    const uint32_t my_last = pos0 + q0 + qi;                    // last key of this lane's query

    // per-lane source offsets of this wave's LDS-DMA instructions (K: 4 keys per instruction, V^T: 16 d rows per instruction)
    static_assert(KI / 4 <= 2 && VI / 4 <= 2, "source offset arrays");
    uint32_t ksrc[2], vsrc[2];  // fixed bounds: with template-dependent bounds captured by a lambda the host pass of hipcc 7.2 drops the kernel's definition
#pragma unroll
    for (int i = 0; i < KI / 4; i++) {
        const uint32_t ins = i * 4 + wave, key = ins * (1024 / KROW) + lane / KCH, c = lane % KCH;
        ksrc[i] = key * KROW + ((c ^ (key & (KCH - 1) & 15)) * 16);
    }
#pragma unroll
    for (int i = 0; i < VI / 4; i++) {
        const uint32_t ins = i * 4 + wave, dd = ins * 16 + lane / 4, c = lane % 4;
        vsrc[i] = dd * Spad * 2 + ((c ^ ((dd >> 2) & 3)) * 16);
    }
    auto issue = [&](uint32_t st, uint32_t stage) {
        const uint32_t k0 = (min(st, nsteps - 1) * KG + kg) * 32;  // this key group's sub-tile
#pragma unroll
        for (int i = 0; i < KI / 4; i++)
            __builtin_amdgcn_global_load_lds(Kb + (uint64_t)k0 * KROW + ksrc[i], (lds_u8 *)(lds + stage * STAGE + kg * SUB + (i * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < VI / 4; i++)
            __builtin_amdgcn_global_load_lds(Vb + (uint64_t)k0 * 2 + vsrc[i], (lds_u8 *)(lds + stage * STAGE + kg * SUB + K_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
    };
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -1.0e30f, l = 0.f;  // (see the softmax below: masked scores are -1e38)
    typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s2 = 0; s2 < NST - 1; s2++) issue(s2, s2);
    uint32_t cur = 0, fill = NST - 1;
    APF_TICK(c_issue);
    for (uint32_t st = 0; st < nsteps; st++) {
        wait_vmcnt<(KI / 4 + VI / 4) * (NST - 2)>();
        APF_TICK(c_wait);
        __builtin_amdgcn_s_barrier();
        APF_TICK(c_bar);
        issue(st + NST - 1, fill);
        APF_TICK(c_issue);
        const uint8_t *lk = lds + cur * STAGE + kg * SUB, *lv = lk + K_BYTES;
        const uint32_t k0 = (st * KG + kg) * 32;
        // K fragments of both 16-key tiles first — eight 16-byte reads in flight, then the eight MFMAs, the two tiles' chains
        // alternating (left to itself hipcc read one fragment, waited lgkmcnt(0), multiplied, eight times in a row: the scores +
        // softmax part of a step took 1570 cycles per wave, tools/attn_pf_stamps.py)
        f16x8 kf[2][KK];
#pragma unroll
        for (int kt = 0; kt < 2; kt++) {
            const uint32_t key = kt * 16 + qi;
#pragma unroll
            for (int kk = 0; kk < KK; kk++)
                kf[kt][kk] = *reinterpret_cast<const f16x8 *>(lk + key * KROW + (((kk * 4 + g) ^ (key & (KCH - 1) & 15)) * 16));
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 sc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kk = 0; kk < KK; kk++)
#pragma unroll
            for (int kt = 0; kt < 2; kt++) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][kk], qf[kk], sc[kt], 0, 0, 0);
        // scale; the causal limit only where the sub-tile reaches past the first of the wave's 16 rows (wave-uniform)
        if (k0 + 31 <= pos0 + q0) {
#pragma unroll
            for (int kt = 0; kt < 2; kt++) sc[kt] *= scale;
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t key = k0 + kt * 16 + g * 4 + r;
                    sc[kt][r] = key <= my_last ? sc[kt][r] * scale : -1.0e38f;
                }
        }
        float mx = m;
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++) mx = fmaxf(mx, sc[kt][r]);
        {   // max over the four lane groups of a query (lanes q, q + 16, q + 32, q + 48) inside the vector ALU: v_permlane16_swap /
            // v_permlane32_swap hand every row its neighbours' values (gfx950); the shuffle form is two dependent trips through the LDS crossbar
            uint32_t b = __builtin_bit_cast(uint32_t, mx);
            auto r16 = __builtin_amdgcn_permlane16_swap(b, b, false, false);  // [0]: rows {0,0,2,2}, [1]: rows {1,1,3,3}
            mx = fmaxf(__builtin_bit_cast(float, (uint32_t)r16[0]), __builtin_bit_cast(float, (uint32_t)r16[1]));
            b = __builtin_bit_cast(uint32_t, mx);
            auto r32 = __builtin_amdgcn_permlane32_swap(b, b, false, false);  // [0]: rows {0,1,0,1}, [1]: rows {2,3,2,3}
            mx = fmaxf(__builtin_bit_cast(float, (uint32_t)r32[0]), __builtin_bit_cast(float, (uint32_t)r32[1]));
        }
        const float corr = __expf(fmaxf(m - mx, -80.f));  // 1 when nothing changes; e^-80 ~ 0 while nothing has been seen (acc and l are 0 then)
        f16x8 pf;
        float ls = 0.f;
        // A masked key carries -1e38 and the running maximum never falls below -1e30 (its initial value): its exponent clamps to -80
        // and e^-80 rounds to an fp16 ZERO — exactly what the reference's masked-out weight is — without a branch per key.
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float e = __expf(fminf(fmaxf(sc[kt][r] - mx, -80.f), 80.f));  // AttentionSoftmaxShader.cs:160-166
                const _Float16 eh = (_Float16)e;
                pf[kt * 4 + r] = eh;
                ls += (float)eh;  // the sum of what is actually multiplied
            }
        l = l * corr + ls;
        m = mx;
        APF_TICK(c_s);
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
            acc[dt] *= corr;
            const uint32_t dd = dt * 16 + qi, sw = (dd >> 2) & 3;
            const f16x4v v0 = *reinterpret_cast<const f16x4v *>(lv + dd * 64 + (((g >> 1) ^ sw) * 16) + (g & 1) * 8);
            const f16x4v v1 = *reinterpret_cast<const f16x4v *>(lv + dd * 64 + (((2 + (g >> 1)) ^ sw) * 16) + (g & 1) * 8);
            f16x8 va;
#pragma unroll
            for (int e = 0; e < 4; e++) { va[e] = v0[e]; va[4 + e] = v1[e]; }
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pf, acc[dt], 0, 0, 0);
        }
        cur = cur + 1 == NST ? 0 : cur + 1;
        fill = fill + 1 == NST ? 0 : fill + 1;
        APF_TICK(c_pv);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus DMAs of the last iterations must not outlive the workgroup's LDS
#ifdef NFAI_STAMPS
    {
        APF_TICK(c_wait);
        const uint32_t wg = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + wave_all;
        if (stamps && wg < STAMP_WAVES && lane == 0) {
            unsigned long long *o = stamps + (size_t)wg * STAMP_WORDS;
            o[0] = c_start; o[1] = c_wait; o[2] = c_bar; o[3] = c_issue; o[4] = c_s; o[5] = c_pv; o[6] = c1; o[7] = ((unsigned long long)nsteps << 32) | qb;
        }
    }
#endif
#undef APF_TICK
    if constexpr (KG >= 2) {
        // merge the key groups of a query tile through LDS (the ring is free now): groups 1.. hand (max, partial sum, output) over,
        // group 0 folds them in, in group order
        constexpr uint32_t MB = 2 + DT * 4;
        static_assert((KG - 1) * 256 * MB * 4 <= NST * STAGE, "merge records fit the ring");
        __syncthreads();
        float *mb = reinterpret_cast<float *>(lds) + (size_t)(wave * 64 + lane) * MB;
        if (kg != 0) {
            float *w = mb + (size_t)(kg - 1) * 256 * MB;
            w[0] = m;
            w[1] = l;
#pragma unroll
            for (int dt = 0; dt < DT; dt++)
#pragma unroll
                for (int r = 0; r < 4; r++) w[2 + dt * 4 + r] = acc[dt][r];
        }
        __syncthreads();
        if (kg != 0) return;
#pragma unroll
        for (int g2 = 1; g2 < KG; g2++) {
            const float *rd = mb + (size_t)(g2 - 1) * 256 * MB;
            const float m1 = rd[0], l1 = rd[1];
            const float M = fmaxf(m, m1);
            const float a0 = __expf(fmaxf(m - M, -80.f)), a1 = __expf(fmaxf(m1 - M, -80.f));
            l = l * a0 + l1 * a1;
            m = M;
#pragma unroll
            for (int dt = 0; dt < DT; dt++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[dt][r] = acc[dt][r] * a0 + rd[2 + dt * 4 + r] * a1;
        }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    if (q0 + qi < T) {
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
            f16x4v o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = (_Float16)(acc[dt][r] * inv);
            *reinterpret_cast<f16x4v *>(O + (uint64_t)(q0 + qi) * HD + h * D + dt * 16 + g * 4) = o;
        }
    }
}

hipError_t launch_attn_prefill(const void *qh, const void *kh, const void *vt, void *out_f16, uint32_t T, uint32_t H, uint32_t Hkv,
                               uint32_t D, uint32_t Spad, uint32_t pos0, hipStream_t s)
{
    if (T == 0) return hipSuccess;
    if (Hkv == 0 || H % Hkv || Spad % 32 || pos0 + T > Spad || (D != 64 && D != 128)) return hipErrorInvalidValue;
    const dim3 grid((T + 63) / 64, H);
    const float scale = 1.0f / sqrtf((float)D);  // AttentionScoreCalculationShader.cs:93
    // two key groups per query tile (512 threads) unless NFAI_PREFILL_ATTN_KG=1
    static const int env_kg = getenv("NFAI_PREFILL_ATTN_KG") ? atoi(getenv("NFAI_PREFILL_ATTN_KG")) : 2;
    const int kg = (env_kg == 1 || Spad % 64) ? 1 : ((env_kg == 4 && Spad % 128 == 0) ? 4 : 2);  // a step of KG key groups covers 32 KG keys
    const size_t lds = (size_t)(kg > 2 ? 2 : ATTN_PF_NST) * (32 * D * 2 + D * 64) * kg;
    const _Float16 *q = static_cast<const _Float16 *>(qh), *k = static_cast<const _Float16 *>(kh), *v = static_cast<const _Float16 *>(vt);
    _Float16 *o = static_cast<_Float16 *>(out_f16);
    const uint32_t G = H / Hkv;
#ifdef NFAI_STAMPS
#define NFAI_APF_LAUNCH(D_, KG_) k_attn_prefill<D_, KG_><<<grid, 256 * KG_, lds, s>>>(q, k, v, o, T, H, G, Spad, pos0, scale, stamp_next_slot("attn_prefill", grid.x * grid.y, 256 * KG_))
#else
#define NFAI_APF_LAUNCH(D_, KG_) k_attn_prefill<D_, KG_><<<grid, 256 * KG_, lds, s>>>(q, k, v, o, T, H, G, Spad, pos0, scale)
#endif
#define NFAI_APF(D_, KG_)                                                                                                                  \
    do {                                                                                                                                   \
        static bool attr_set = false;                                                                                                      \
        if (lds > 64 * 1024 && !attr_set) {                                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_attn_prefill<D_, KG_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                                                 \
            attr_set = true;                                                                                                               \
        }                                                                                                                                  \
        NFAI_APF_LAUNCH(D_, KG_);                                                                                                          \
    } while (0)
    if (D == 128) { if (kg == 4) NFAI_APF(128, 4); else if (kg == 2) NFAI_APF(128, 2); else NFAI_APF(128, 1); }
    else { if (kg == 4) NFAI_APF(64, 4); else if (kg == 2) NFAI_APF(64, 2); else NFAI_APF(64, 1); }
#undef NFAI_APF
#undef NFAI_APF_LAUNCH
    return hipGetLastError();
}

__global__ void k_silu_mul_rows(const float *gate, const float *up, _Float16 *act, uint64_t n, uint32_t F, uint32_t ld)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t o = (i / F) * ld + (i % F);  // gate | up may be column blocks of one [T][ld] buffer
    act[i] = (_Float16)(up[o] * silu_ref(gate[o]));
}

hipError_t launch_silu_mul_rows(const float *gate, const float *up, void *act_f16, uint32_t T, uint32_t F, uint32_t ld, hipStream_t s)
{
    const uint64_t n = (uint64_t)T * F;
    k_silu_mul_rows<<<(uint32_t)((n + 255) / 256), 256, 0, s>>>(gate, up, static_cast<_Float16 *>(act_f16), n, F, ld);
    return hipGetLastError();
}

// embedding rows for T tokens (device token ids) -> fp32 [T][E]
__global__ void k_embed_rows(const void *table, int type, const uint32_t *toks, float *x, uint32_t E)
{
    const uint32_t t = blockIdx.y, d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= E) return;
    const uint64_t row = (uint64_t)toks[t] * E;
    x[(uint64_t)t * E + d] = type == NFAI_F16 ? (float)reinterpret_cast<const _Float16 *>(table)[row + d]
                                              : reinterpret_cast<const float *>(table)[row + d];
}

hipError_t launch_embed_rows(const void *table, int type, const uint32_t *toks, float *x, uint32_t T, uint32_t E, hipStream_t s)
{
    k_embed_rows<<<dim3((E + 255) / 256, T), 256, 0, s>>>(table, type, toks, x, E);
    return hipGetLastError();
}


// ---- weight read-ahead for the prefill ----------------------------------------------------------------------------------------------
// A projection of the prefill reads its weight matrix exactly once; coming from HBM, the LDS-DMA pipeline of a 128-row tile waits on
// first-use latency at every K step (tools/gemm_bench.py: the same GEMMs run 15-40 % faster on a matrix that sits in the 256 MB
// Infinity Cache).  This launch, on a side stream while the PREVIOUS GEMM of the block computes, reads the next matrix through with
// default-policy loads so that it is on-die when its GEMM starts.  A pure hint: it writes nothing.
__global__ __launch_bounds__(256) void k_read_ahead(const u32x4 *w, uint64_t n16)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const u32x4 v = *((const __attribute__((address_space(1))) u32x4 *)(w + i));
        acc ^= v[0] ^ v[3];
    }
    asm volatile("" ::"v"(acc));
}

hipError_t launch_read_ahead(const void *w, uint64_t bytes, uint32_t n_cu, hipStream_t s)
{
    if (!w || bytes < 16) return hipSuccess;
    k_read_ahead<<<n_cu / 2 ? n_cu / 2 : 1, 256, 0, s>>>(static_cast<const u32x4 *>(w), bytes / 16);
    return hipGetLastError();
}

}  // namespace nfai
