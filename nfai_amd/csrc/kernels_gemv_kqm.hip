// kernels_gemv_kqm.hip — decode GEMV on ggml K-quant weights, second generation: the integer dot products
// of the quantised weights with fixed-point activations run on the matrix cores (v_mfma_i32_16x16x64_i8 used as
// 64 independent 16-wide dot products per instruction), the K-quant scales are applied in fp32.
// Same op as kernels_gemv_kq.hip (MatrixMultiplyShader.cs:255-289 on weights the reference cannot
// load at all, Parser.cs:111-114); that file stays as the path for row counts that are not a
// multiple of 16.
//
// Why: kernels_gemv_kq.hip spends ~2.8 VALU operations per weight and lane (cvt + fma per nibble,
// eight ds_read_b128 of activations per 32 weights) and runs at 1.8 TB/s on the gate/up matrix of
// Llama-3.2-3B — VALU/LDS-bound at a quarter of what HBM delivers.  Here four nibbles become four
// byte operands with ONE v_and_b32 (q & 0x0F0F0F0F; the high nibbles one shift more), the multiply-
// accumulate moves to MFMA, and the activations are read from LDS by 12 lanes per instruction
// instead of 64: 0.375 (Q4_K) / 0.8 (Q6_K) VALU operations per weight and lane.
//
// Numerics: each 256-element super-block of the activation vector (after the optional RMSNorm) is
// scaled by a power of two so that |x * 2^S| < 2^22, rounded to an integer and split into three
// signed base-256 digits; sum q * x' = S0 + 256 S1 + 65536 S2 is exact in int32, the K-quant scales
// (d * sc, dmin * m, the -32 offset of Q6_K via the group sums of x') are applied in fp32 exactly
// as ggml defines the block.  The only approximation is x' = round(x * 2^S): 24 bits relative to
// the largest |x| of the super-block.  Against the fp32 oracle the differences are at the level of
// fp32 summation-order noise (tests/test_gpu_kquant.py states the tolerance; max |dlogit| 1.5e-5 at 3B).
//
// HBM layout ("T16", repacked once at upload, same bytes): rows are grouped in tiles of 16.
//   Q4_K: plane 0  [tile][blk][h:2][lane:64][16 B]  lane = G*16 + r holds qs[32G+16h .. +16) of row 16*tile+r,
//                  i.e. lane group G owns sub-blocks 2G (low nibbles) and 2G+1 (high nibbles) of its row;
//         plane 1  [tile][blk][r:16][16 B]          d, dmin, 12 scale bytes of row 16*tile+r.
//   Q6_K: see k_repack_q6k_t16.
// Every wave-wide load is one contiguous kilobyte (quants) or 256 bytes (headers).
//
// Work split: a workgroup owns whole 16-row tiles ("units": one tile, or the gate and the up tile of
// the same rows); its waves split K (wave w owns super-blocks w*BPW .. w*BPW+BPW-1 of every tile and
// stages exactly those super-blocks of x: no workgroup barrier before the first dot product unless
// the RMSNorm sum needs one), partial sums meet in LDS every UB units.  MFMA operand roles:
// A = activations (rows 4G, 4G+1, 4G+2 of the 16x64 A matrix carry the three digits of the 16 k-slots
// of lane group G, zero elsewhere), B = weights (column r = row r of the tile), so D[4G + {0,1,2}][r]
// lands in registers 0..2 of lane (G, r): every lane receives the dot product of exactly the 16
// weights it unpacked.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kqm.h"

namespace nfai {

template <int MODE>
__device__ __forceinline__ void kqm_unit(const KqmParams &p, uint32_t u, int t, uint32_t &seg, uint32_t &tile)
{
    if constexpr (MODE == GEMV_GATEUP) {
        seg = t; tile = u;
    } else if constexpr (MODE == GEMV_QKV_ROPE) {
        u += p.rot;
        if (u >= p.NU) u -= p.NU;
        if (u < p.seg_tile_end[0]) { seg = 0; tile = u; }
        else if (u < p.seg_tile_end[1]) { seg = 1; tile = u - p.seg_tile_end[0]; }
        else { seg = 2; tile = u - p.seg_tile_end[1]; }
    } else {
        seg = 0; tile = u;
    }
}

struct Q6T { u32x4 qla, qlb, qh, sc; uint32_t d; };

__device__ __forceinline__ Q6T q6t_load(const KqmParams &p, uint32_t seg, uint32_t tile, uint32_t blk, uint32_t lane)
{
    const uint64_t tb = (uint64_t)tile * p.NB + blk;
    const uint64_t nblk = (uint64_t)p.seg_tiles[seg] * 16 * p.NB;
    const uint8_t *base = p.W[seg];
    Q6T r;
    r.qla = load_nt16(base + tb * 3072 + lane * 16);
    r.qlb = load_nt16(base + tb * 3072 + 1024 + lane * 16);
    r.qh = load_nt16(base + tb * 3072 + 2048 + lane * 16);
    r.sc = load_nt16(base + nblk * 192 + tb * 256 + (lane & 15) * 16);
    r.d = *reinterpret_cast<const GLOBAL_AS uint16_t *>((const GLOBAL_AS uint8_t *)base + nblk * 208 + tb * 32 + (lane & 15) * 2);
    return r;
}

// 64 weights of one lane: half n = G>>1 of the super-block, columns l = 16*(G&1) .. +15 of all four quarters
// (ggml dequantize_row_q6_K: quarter 0/2 = low/high nibbles of ql[l], quarter 1/3 = of ql[l+32], bits 2q..2q+1 of qh[l]);
// one MFMA per quarter = one 16-weight scale group.  sums = sum of x' over each of the four groups.
__device__ __forceinline__ float q6t_dot(const Q6T &w, const i32x4 (&af)[4], f32x4 sums, uint32_t g)
{
    constexpr uint32_t M4 = 0x0F0F0F0Fu, M2 = 0x30303030u;
    const float d = h2f_lo(w.d);
    float tot = 0.f;
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
        const u32x4 ql = (qd & 1) ? w.qlb : w.qla;
        const u32x4 lo4 = (qd >= 2) ? ((ql >> 4) & M4) : (ql & M4);
        const u32x4 hs = qd == 0 ? (w.qh << 4) : (qd == 1 ? (w.qh << 2) : (qd == 2 ? w.qh : (w.qh >> 2)));
        const u32x4 b = (hs & M2) | lo4;  // unsigned 6-bit value per byte
        const i32x4 dq = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[qd], __builtin_bit_cast(i32x4, b), i32x4{0, 0, 0, 0}, 0, 0, 0);
        const float v = fmaf((float)dq[2], 65536.0f, fmaf((float)dq[1], 256.0f, (float)dq[0]));
        // scales[8n + (G&1) + 2*qd] of the row
        const uint32_t si = 8 * (g >> 1) + (g & 1) + 2 * qd;
        const uint32_t sw = si < 8 ? (si < 4 ? w.sc[0] : w.sc[1]) : (si < 12 ? w.sc[2] : w.sc[3]);
        const int sc = (int)(int8_t)((sw >> ((si & 3) * 8)) & 0xFFu);
        tot = fmaf((float)sc, fmaf(-32.0f, sums[qd], v), tot);
    }
    return d * tot;
}

__device__ __forceinline__ void kqm_kv_store(void *base, int f16, uint64_t idx, float v)
{
    if (f16) reinterpret_cast<_Float16 *>(base)[idx] = (_Float16)v;
    else reinterpret_cast<float *>(base)[idx] = v;
}

// What the epilogue of unit u reads besides the dot products; loaded unconditionally (clamped) so that the
// first round's copy can be fetched at kernel start, a microsecond of latency off the tail of the short kernels.
struct KqmPre { float res; f32x2 cs; uint32_t pos; };

template <int MODE, bool BEGIN = false>
__device__ __forceinline__ KqmPre kqm_preload(const KqmParams &p, uint32_t u, uint32_t lane)
{
    KqmPre q;
    q.res = 0.f; q.cs = f32x2{1.f, 0.f}; q.pos = 0;
    const uint32_t r = lane & 15;
    if constexpr (MODE == GEMV_RESIDUAL) {
        q.res = *((const GLOBAL_AS float *)p.res + (u * 16 + r));
    } else if constexpr (MODE == GEMV_QKV_ROPE) {
        uint32_t seg, tile;
        kqm_unit<MODE>(p, u, 0, seg, tile);
        const uint32_t de = ((tile * 16 + r) % p.D) & ~1u;
        q.pos = *((const GLOBAL_AS uint32_t *)p.pos);
        if constexpr (!BEGIN)  // (first launch of a token: the epilogue takes cos/sin from the workgroup's own table in LDS, begin_bookkeeping)
            q.cs = *reinterpret_cast<const GLOBAL_AS f32x2 *>((const GLOBAL_AS float *)p.rope_cs + min(de, p.rope_dims - 2));
    }
    return q;
}

// lanes 0..15 hold the 16 rows of the unit (a0; a1 = up row for GATEUP)
template <int MODE>
__device__ __forceinline__ void kqm_epilogue(const KqmParams &p, uint32_t u, uint32_t lane, float a0, float a1, const KqmPre &pre, float &best_v,
                                             uint32_t &best_i)
{
    const uint32_t r = lane & 15;
    if constexpr (MODE == GEMV_PLAIN) {
        if (lane < 16) {
            p.y[u * 16 + r] = a0;
            if (topk_better(a0, u * 16 + r, best_v, best_i)) { best_v = a0; best_i = u * 16 + r; }  // lm_head + ArgMax in one launch
        }
    } else if constexpr (MODE == GEMV_RESIDUAL) {
        if (lane < 16) p.y[u * 16 + r] = pre.res + a0;
    } else if constexpr (MODE == GEMV_GATEUP) {
        if (lane < 16) p.y[u * 16 + r] = a1 * silu_ref(a0);
    } else {
        uint32_t seg, tile;
        kqm_unit<MODE>(p, u, 0, seg, tile);
        const uint32_t row = tile * 16 + r, head = row / p.D, dd = row % p.D;
        const float other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0xB1, 0xF, 0xF, true));
        float o = a0;
        const uint32_t de = dd & ~1u;  // the even element of the rotated pair
        if (seg < 2 && de < p.rope_dims) {
            const float c = pre.cs[0], sn = pre.cs[1];
            o = (dd & 1) ? (sn * other + c * a0) : (c * a0 - sn * other);
        }
        if (lane < 16) {
            if (seg == 0) p.y[row] = o;
            else kqm_kv_store(seg == 1 ? p.kc : p.vc, p.kv_f16, (uint64_t)pre.pos * p.kv_pos_stride + (uint64_t)head * p.kv_head_stride + dd, o);
        }
    }
}

// NS > 0 (1 or 2): no workgroup has more than NS steps — all of them are issued before the prologue and consumed
// in a straight line (the common case for the per-block matrices of a 3B model).  NS = 0: ping-pong over two buffers.
// (Measured and dropped: an s_barrier between the activation loads and the first weight loads, so that no wave's x
// queues behind another wave's weights — no effect; four steps in flight — slower, see the step list in DESIGN.md.)
// BEGIN (q|k|v with RMSNorm only): the first launch of a token — embedding row, cos/sin table and the token's bookkeeping in this launch.
// A template parameter and not a kernel argument: as a run-time branch the (dead) bookkeeping code sat between the activation loads and
// the first weight requests of EVERY q|k|v launch (stamps: weights requested at 1.16 us against 0.4 us in the other launches).
template <int QT, int MODE, int BPW, bool NORM, int NS, bool BEGIN = false>
__global__ __launch_bounds__(1024) void k_gemv_kqt(const KqmParams p)
{
    static_assert(!BEGIN || (MODE == GEMV_QKV_ROPE && NORM), "a token begins with a normed q|k|v launch");
    // QT = NFAI_KQ_MIXED (QKV only): Q4_K and Q6_K segments in one launch (Q4_K_M files keep attn_v in Q6_K on half of
    // the blocks); the activations are staged in both fragment layouts and every step branches on its segment's type.
    constexpr bool HAS4 = QT != NFAI_Q6_K_T16, HAS6 = QT != NFAI_Q4_K_T16, MIXED = HAS4 && HAS6;
    static_assert(!MIXED || MODE == GEMV_QKV_ROPE, "mixed encodings exist for the q|k|v launch only");
    using Regs = typename std::conditional<QT == NFAI_Q4_K_T16, Q4T, Q6T>::type;
    constexpr int R = MODE == GEMV_GATEUP ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t kpad = nw * BPW * 256;
    const uint32_t UB = p.UB;
    uint8_t *xa = smem;                                               // kpad * 4 bytes: A fragments [blk][slot][G][digit][16 B]
    uint8_t *xa6 = MIXED ? smem + (size_t)kpad * 4 : xa;             // second fragment layout (Q6_K) when both are needed
    float *sums = reinterpret_cast<float *>(xa6 + (size_t)kpad * 4);  // [blk][G][4]: sums of x' per scale group
    float *sums6 = MIXED ? sums + nw * BPW * 16 : sums;
    float *scl = sums6 + nw * BPW * 16;                               // [blk] 2^-S of the super-block's fixed-point scale
    float *red = scl + nw * BPW;                                      // [2][UB][R][nw][64]
    float *scal = red + 2 * UB * R * nw * 64;                       // 32 floats of reduction scratch

    const uint32_t nunits = (p.NU - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const uint32_t nsteps = nunits * R * BPW;  // step = (unit, tile of the unit, super-block of this wave)
    STAMP_DECL;
    STAMP(0);  // wave started

    // ---- activations first (vmcnt is in order: what the prologue needs must not queue behind weights).
    // Wave w loads, scales and stages exactly the super-blocks it will consume (w*BPW .. w*BPW+BPW-1): the fixed-point
    // scale is per super-block (a wave reduction, no LDS), and nothing staged here is read by another wave — the only
    // workgroup barrier of the prologue is the one the RMSNorm sum needs.
    f32x4 xv[BPW], gv[NORM ? BPW : 1];
#pragma unroll
    for (int i = 0; i < BPW; i++) {
        const uint32_t bw = wid * BPW + i;
        const uint32_t kk = min(bw, p.NB - 1) * 256 + lane * 4;
        f32x4 v;
        if (BEGIN && p.begin.emb) {  // block-uniform: the token's embedding row (first launch of a token)
            v = embed_load4(p.begin.emb, p.begin.emb_type, p.begin.emb_rows, p.begin.tok[0], kk, p.K);
            if (blockIdx.x == 0 && bw < p.NB) *reinterpret_cast<f32x4 *>(p.begin.x_out + kk) = v;  // the residual the later launches read
        } else {
            v = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.x + kk);
        }
        xv[i] = bw < p.NB ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NORM) gv[i] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.gamma + kk);
    }

    // keep the activation loads FIRST in program order (vmcnt retires in order: the prologue must not wait behind
    // weights): without this fence hipcc's scheduler interleaves the loads below with them (measured: -2.5 % tokens/s)
    __builtin_amdgcn_sched_barrier(0);
    const KqmPre pre0 = kqm_preload<MODE, BEGIN>(p, min(blockIdx.x + wid * gridDim.x, p.NU - 1), lane);  // epilogue inputs of round 0
    float *cs_lds = scal + 32 + 48;  // behind the fused ArgMax words
    if constexpr (BEGIN) begin_bookkeeping(p.begin, pre0.pos, cs_lds);
    // ---- weights of the first steps
    constexpr int NBUF = NS > 0 ? NS : 2;
    Regs buf[NBUF];
    uint32_t ist = 0;
    auto issue = [&](Regs &buf) {
        const uint32_t is = min(ist, nsteps - 1);  // past the end: the wave's own last step again
        const uint32_t ui = is / (R * BPW), tt = (is / BPW) % R, bi = is % BPW;
        const uint32_t u = blockIdx.x + ui * gridDim.x;
        const uint32_t blk = min(wid * BPW + bi, p.NB - 1);
        uint32_t seg, tile;
        kqm_unit<MODE>(p, u, tt, seg, tile);
        if constexpr (QT == NFAI_Q4_K_T16) {
            buf = q4t_load(p, seg, tile, blk, lane);
        } else if constexpr (QT == NFAI_Q6_K_T16) {
            buf = q6t_load(p, seg, tile, blk, lane);
        } else {
            if ((p.seg6 >> seg) & 1u) {
                buf = q6t_load(p, seg, tile, blk, lane);
            } else {
                const Q4T q = q4t_load(p, seg, tile, blk, lane);
                buf.qla = q.q0; buf.qlb = q.q1; buf.sc = q.hdr;
            }
        }
        ++ist;
    };
#pragma unroll
    for (int j = 0; j < NBUF; j++) issue(buf[j]);
    STAMP(1);  // activation loads and the first weight steps issued

    // ---- prologue: per-super-block power-of-two scale, three base-256 digits -> LDS, scale-group sums.
    // RMSNorm (RMSNormShader.cs:136-149: (x / rms) * g) is applied in two parts: the gains here, element by element (u = x * g), the
    // division by rms = sqrt(mean(x^2) + eps) on the finished dot products in the epilogue — it is one scalar per vector, and
    // waiting for it here meant a workgroup barrier on the slowest wave's x, whose request sits behind every other wave's first
    // weight requests in the CU's in-order memory queue (stamps, round 3: x staged at 2.7-3.4 us with the barrier, 0.8 us without;
    // 3B Q4_K_M 1026 -> 1088 tokens/s).  Every wave stages exactly the super-blocks it consumes, so nothing here waits for another
    // wave; its share of sum(x^2) goes to LDS and is combined in fixed wave order behind the barrier the cross-wave reduction needs
    // anyway.  The fixed-point rounding of u (24 bits relative to the super-block's maximum) dominates the two roundings the
    // reordering moves.
    {
        if constexpr (NORM) {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < BPW; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) ss = fmaf(xv[i][e], xv[i][e], ss);
            ss = wave_sum(ss);
            if (lane == 0) scal[wid] = ss;
        }
#pragma unroll
        for (int i = 0; i < BPW; i++) {
            const uint32_t blk = wid * BPW + i;
            f32x4 v = xv[i];
            if constexpr (NORM) {
                v[0] = v[0] * gv[i][0];
                v[1] = v[1] * gv[i][1];
                v[2] = v[2] * gv[i][2];
                v[3] = v[3] * gv[i][3];
            }
            kqm_stage<HAS4, HAS6>(v, blk, lane, xa, xa6, sums, sums6, scl);
        }
        // no barrier: every LDS word written above is read only by this wave (LDS operations of a wave execute in order)
    }
    STAMP(2);  // x normalised, scaled and staged as digits in LDS

    const uint32_t g = lane >> 4, ra = lane & 15;
    const bool a_live = (ra >> 2) == g && (ra & 3) < 3;  // A rows 4G, 4G+1, 4G+2 = digits 0, 1, 2 of lane group G
    const uint32_t a_off = g * 64 + (ra & 3) * 16;

    float acc = 0.f;
    float best_v = -INFINITY;
    uint32_t best_i = 0xFFFFFFFFu;
    uint32_t cst = 0, par = 0;
    auto consume = [&](Regs &buf) {
        const uint32_t ui = cst / (R * BPW), tt = (cst / BPW) % R, bi = cst % BPW;
        const uint32_t bc = wid * BPW + bi;
        const bool live = bc < p.NB;
        const uint32_t blk = min(bc, p.NB - 1);
        bool is6 = QT == NFAI_Q6_K_T16;
        if constexpr (MIXED) {
            uint32_t seg, tile;
            kqm_unit<MODE>(p, blockIdx.x + ui * gridDim.x, tt, seg, tile);
            is6 = (p.seg6 >> seg) & 1u;
        }
        float a;
        // A fragments: only the lanes that carry digits read LDS (exec-masked), the rest supply zeros
        const uint8_t *abase = (is6 ? xa6 : xa) + (size_t)blk * 1024 + a_off;
        i32x4 af[4] = {i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}};
        if (a_live) {
#pragma unroll
            for (int sl = 0; sl < 4; sl++) af[sl] = *reinterpret_cast<const i32x4 *>(abase + sl * 256);
        }
        const float inv_scale = scl[blk];
        if (is6) {
            if constexpr (HAS6) {
                const f32x4 sm = *reinterpret_cast<const f32x4 *>(sums6 + (blk * 4 + g) * 4);
                a = q6t_dot(buf, af, sm, g);
            }
        } else {
            if constexpr (HAS4) {
                const f32x4 sm = *reinterpret_cast<const f32x4 *>(sums + (blk * 4 + g) * 4);
                if constexpr (QT == NFAI_Q4_K_T16) a = q4t_dot(buf, af, f32x2{sm[0], sm[1]}, g);
                else a = q4t_dot(Q4T{buf.qla, buf.qlb, buf.sc}, af, f32x2{sm[0], sm[1]}, g);
            }
        }
        acc += live ? a * inv_scale : 0.f;
        ++cst;
        if (bi == BPW - 1) {
            const uint32_t slot = ui % UB;
            red[(((par * UB + slot) * R + tt) * nw + wid) * 64 + lane] = acc;
            acc = 0.f;
            if (tt == R - 1 && (slot == UB - 1 || ui == nunits - 1)) {
                __syncthreads();
                if (wid <= slot) {  // wave q finishes unit q of this round
                    const uint32_t uq = blockIdx.x + (ui - slot + wid) * gridDim.x;
                    KqmPre pre = ui == slot ? pre0 : kqm_preload<MODE, BEGIN>(p, uq, lane);
                    if constexpr (BEGIN) {
                        {  // the table this workgroup tabulated before its first step (barrier above)
                            uint32_t seg, tile;
                            kqm_unit<MODE>(p, uq, 0, seg, tile);
                            const uint32_t de = ((tile * 16 + (lane & 15)) % p.D) & ~1u;
                            pre.cs = *reinterpret_cast<const f32x2 *>(cs_lds + min(de, p.rope_dims - 2));
                        }
                    }
                    float rms = 1.f;
                    if constexpr (NORM) {  // every wave's share of sum(x^2) is in LDS (written before its first step, barrier above)
                        float tss = 0.f;
#pragma unroll
                        for (uint32_t i = 0; i < 16; i++) {  // fixed trip count and order: the LDS reads issue back to back
                            const float s_i = scal[min(i, nw - 1)];
                            tss += i < nw ? s_i : 0.f;
                        }
                        rms = sqrtf(tss / (float)p.K + p.eps);
                    }
                    float af[R];
#pragma unroll
                    for (int t2 = 0; t2 < R; t2++) {
                        const float *rp = red + (((par * UB + wid) * R + t2) * nw) * 64 + lane;
                        float sp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (uint32_t w = 0; w < 16; w++) {  // fixed trip count, four independent chains
                            const float v = rp[min(w, nw - 1) * 64];
                            sp[w & 3] += w < nw ? v : 0.f;
                        }
                        float s = (sp[0] + sp[1]) + (sp[2] + sp[3]);
                        af[t2] = rows4_sum(s);
                        if constexpr (NORM) af[t2] = af[t2] / rms;
                    }
                    kqm_epilogue<MODE>(p, uq, lane, af[0], af[R - 1], pre, best_v, best_i);
                }
                par ^= 1;
            }
        }
    };

    if constexpr (NS > 0) {
#pragma unroll
        for (int j = 0; j < NS; j++) {
            if ((uint32_t)j < nsteps) consume(buf[j]);
            if (j == 0) STAMP(3);  // first weight step landed, multiplied (and, with one step per unit, reduced)
        }
    } else {
        uint32_t st = 0;
#ifdef NFAI_STAMPS
        if (nsteps > 4) {
            consume(buf[0]);
            STAMP(3);
            issue(buf[0]);
            consume(buf[1]);
            issue(buf[1]);
            st = 2;
        } else {
            STAMP(3);
        }
#endif
        for (; st + 3 < nsteps; st += 2) {
            consume(buf[0]);
            issue(buf[0]);
            consume(buf[1]);
            issue(buf[1]);
        }
        const uint32_t rem = nsteps - st;
        if (rem == 3) {
            consume(buf[0]);
            issue(buf[0]);
            consume(buf[1]);
            consume(buf[0]);
        } else if (rem == 2) {
            consume(buf[0]);
            consume(buf[1]);
        } else if (rem == 1) {
            consume(buf[0]);
        }
    }
#ifdef NFAI_STAMPS
    STAMP(4);  // last step consumed, reductions and epilogue stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(5);  // stores acknowledged
    STAMP_FLUSH(p.stamps, blockIdx.x * nw + wid, 6);
#endif
    if constexpr (MODE == GEMV_PLAIN) {
        // SamplingUtils.ArgMax over the outputs in the same launch (block-uniform: a kernel argument); its 48 LDS words sit behind `scal`
        if (p.am.ticket != nullptr) argmax_fused_tail(best_v, best_i, p.am, reinterpret_cast<uint32_t *>(scal + 32));
    }
}

// ---- Q4_K repack: native 144-byte blocks (row-major) -> T16 planes (same bytes) ------------------------
__global__ void k_repack_q4k_t16(const uint8_t *src, uint8_t *dst, uint32_t n_tiles, uint32_t NB)
{
    const uint64_t tb = blockIdx.x;  // tile * NB + blk
    const uint32_t tile = (uint32_t)(tb / NB), blk = (uint32_t)(tb % NB), t = threadIdx.x;
    const uint64_t nblk = (uint64_t)n_tiles * 16 * NB;
    if (t < 128) {
        const uint32_t h = t >> 6, ln = t & 63, G = ln >> 4, r = ln & 15;
        const uint8_t *s = src + ((uint64_t)(tile * 16 + r) * NB + blk) * 144 + 16 + 32 * G + 16 * h;
        *reinterpret_cast<u32x4 *>(dst + tb * 2048 + h * 1024 + ln * 16) = *reinterpret_cast<const u32x4 *>(s);
    } else if (t < 144) {
        const uint32_t r = t - 128;
        const uint8_t *s = src + ((uint64_t)(tile * 16 + r) * NB + blk) * 144;
        *reinterpret_cast<u32x4 *>(dst + nblk * 128 + tb * 256 + r * 16) = *reinterpret_cast<const u32x4 *>(s);
    }
}

hipError_t launch_repack_q4k_t16(const void *native, void *tiled, uint64_t rows, uint64_t cols, hipStream_t s)
{
    if (rows == 0) return hipSuccess;
    if (rows % 16 || cols % 256) return hipErrorInvalidValue;
    const uint64_t nb = cols / 256, grid = rows / 16 * nb;
    if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
    k_repack_q4k_t16<<<(uint32_t)grid, 192, 0, s>>>(static_cast<const uint8_t *>(native), static_cast<uint8_t *>(tiled), (uint32_t)(rows / 16),
                                                    (uint32_t)nb);
    return hipGetLastError();
}

// ---- Q6_K repack: native 210-byte blocks (row-major) -> T16 planes (same bytes) ------------------------
//   plane 0  [tile][blk][piece:3][lane:64][16 B]  lane = G*16 + r, n = G>>1, lh = G&1:
//            piece 0 = ql[64n + 16lh ..+16), piece 1 = ql[64n + 32 + 16lh ..+16), piece 2 = qh[32n + 16lh ..+16)
//   plane 1  [tile][blk][r:16][16 B]  the 16 int8 scales      plane 2  [tile][blk][r:16] fp16 d
__global__ void k_repack_q6k_t16(const uint8_t *src, uint8_t *dst, uint32_t n_tiles, uint32_t NB)
{
    const uint64_t tb = blockIdx.x;
    const uint32_t tile = (uint32_t)(tb / NB), blk = (uint32_t)(tb % NB), t = threadIdx.x;
    const uint64_t nblk = (uint64_t)n_tiles * 16 * NB;
    // native blocks are only 2-byte aligned: byte copies
    for (uint32_t e = t; e < 3072; e += blockDim.x) {
        const uint32_t piece = e >> 10, ln = (e >> 4) & 63, b = e & 15, G = ln >> 4, r = ln & 15, n = G >> 1, lh = G & 1;
        const uint8_t *s = src + ((uint64_t)(tile * 16 + r) * NB + blk) * 210;
        const uint32_t off = piece == 0 ? 64 * n + 16 * lh : (piece == 1 ? 64 * n + 32 + 16 * lh : 128 + 32 * n + 16 * lh);
        dst[tb * 3072 + e] = s[off + b];
    }
    for (uint32_t e = t; e < 256; e += blockDim.x) {
        const uint32_t r = e >> 4, b = e & 15;
        dst[nblk * 192 + tb * 256 + e] = src[((uint64_t)(tile * 16 + r) * NB + blk) * 210 + 192 + b];
    }
    if (t < 32) dst[nblk * 208 + tb * 32 + t] = src[((uint64_t)(tile * 16 + (t >> 1)) * NB + blk) * 210 + 208 + (t & 1)];
}

hipError_t launch_repack_q6k_t16(const void *native, void *tiled, uint64_t rows, uint64_t cols, hipStream_t s)
{
    if (rows == 0) return hipSuccess;
    if (rows % 16 || cols % 256) return hipErrorInvalidValue;
    const uint64_t nb = cols / 256, grid = rows / 16 * nb;
    if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
    k_repack_q6k_t16<<<(uint32_t)grid, 256, 0, s>>>(static_cast<const uint8_t *>(native), static_cast<uint8_t *>(tiled), (uint32_t)(rows / 16),
                                                    (uint32_t)nb);
    return hipGetLastError();
}

__global__ void k_embed_q6t(const uint8_t *table, uint64_t n_rows, const uint32_t *tok, float *y, uint32_t E)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E) return;
    tok += blockIdx.y;
    y += (uint64_t)blockIdx.y * E;
    const uint32_t NB = E / 256, blk = k >> 8, kk = k & 255, n = kk >> 7, qd = (kk >> 5) & 3, l = kk & 31, lh = l >> 4, b = l & 15;
    const uint64_t row = tok[0], tile = row >> 4, r = row & 15, tb = tile * NB + blk, nblk = n_rows * NB;
    const uint32_t ln = (n * 2 + lh) * 16 + (uint32_t)r;
    const uint8_t ql = table[tb * 3072 + (qd & 1) * 1024 + ln * 16 + b];
    const uint8_t qh = table[tb * 3072 + 2048 + ln * 16 + b];
    const int8_t sc = (int8_t)table[nblk * 192 + tb * 256 + r * 16 + 8 * n + lh + 2 * qd];
    const float d = (float)reinterpret_cast<const _Float16 *>(table + nblk * 208 + tb * 32)[r];
    const int q = (int)(((qd >= 2) ? (ql >> 4) : (ql & 0xF)) | (((qh >> (2 * qd)) & 3) << 4)) - 32;
    y[k] = d * (float)sc * (float)q;
}

// ---- one row of a T16 table -> fp32 (embedding gather) ----------------------------------------------------
__global__ void k_embed_q4t(const uint8_t *table, uint64_t n_rows, const uint32_t *tok, float *y, uint32_t E)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E) return;
    tok += blockIdx.y;  // row t of a batch (prefill) -> y[t][E]
    y += (uint64_t)blockIdx.y * E;
    const uint32_t NB = E / 256, blk = k >> 8, kk = k & 255, sb = kk >> 5, l = kk & 31;
    const uint64_t row = tok[0], tile = row >> 4, r = row & 15, tb = tile * NB + blk, nblk = n_rows * NB;
    const uint8_t *hdr = table + nblk * 128 + tb * 256 + r * 16;
    const float d = (float)*reinterpret_cast<const _Float16 *>(hdr), dmin = (float)*reinterpret_cast<const _Float16 *>(hdr + 2);
    const uint8_t *scales = hdr + 4;
    uint32_t sc, m;
    if (sb < 4) { sc = scales[sb] & 63; m = scales[sb + 4] & 63; }
    else { sc = (scales[sb + 4] & 0xF) | ((scales[sb - 4] >> 6) << 4); m = (scales[sb + 4] >> 4) | ((scales[sb] >> 6) << 4); }
    const uint8_t q = table[tb * 2048 + (l >> 4) * 1024 + ((sb >> 1) * 16 + r) * 16 + (l & 15)];
    const float qv = (float)((sb & 1) ? (q >> 4) : (q & 0xF));
    y[k] = d * (float)sc * qv - dmin * (float)m;
}

hipError_t launch_embed_rows_kqt(const void *table, int type, uint64_t n_rows, const uint32_t *toks, float *y, uint32_t T, uint32_t E, hipStream_t s)
{
    if (E % 256 || n_rows % 16 || T == 0) return hipErrorInvalidValue;
    const dim3 grid((E + 255) / 256, T);
    if (type == NFAI_Q4_K_T16) k_embed_q4t<<<grid, 256, 0, s>>>(static_cast<const uint8_t *>(table), n_rows, toks, y, E);
    else if (type == NFAI_Q6_K_T16) k_embed_q6t<<<grid, 256, 0, s>>>(static_cast<const uint8_t *>(table), n_rows, toks, y, E);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_embed_kqt(const void *table, int type, uint64_t n_rows, const uint32_t *tok, float *y, uint32_t E, hipStream_t s)
{
    return launch_embed_rows_kqt(table, type, n_rows, tok, y, 1, E, s);
}

// ---- T16 tensor -> fp16 [rows][cols] (K-quant prefill: one block's matrices are widened into a scratch buffer and
// go through the fp16 MFMA GEMMs of kernels_prefill.hip; the decode path never dequantises to memory) -------------------
// One wave per (tile, super-block); lane (G, r) expands the 64 weights it also owns in the GEMV.  The wave's 16 x 256 fp16 go through
// LDS (rows padded by 16 bytes) so that every store instruction writes two whole 512-byte row pieces: stored straight from the
// lanes that expand them, an instruction wrote 64 separate 16-byte pieces in 64 different lines (13.2 us per tensor on average at
// 3B, a third of the K-quant prefill).
constexpr uint32_t DQ_ROW = 512 + 16;
template <int QT>
__global__ __launch_bounds__(256) void k_dequant_t16(const uint8_t *W, _Float16 *out, uint32_t n_tiles, uint32_t NB)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[4][16 * DQ_ROW];
    const uint64_t tb = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tb >= (uint64_t)n_tiles * NB) return;
    const uint32_t lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const uint32_t tile = (uint32_t)(tb / NB), blk = (uint32_t)(tb % NB);
    const uint64_t nblk = (uint64_t)n_tiles * 16 * NB;
    _Float16 *orow = reinterpret_cast<_Float16 *>(stage[threadIdx.x >> 6] + r * DQ_ROW);
    if constexpr (QT == NFAI_Q4_K_T16) {
        const u32x4 q0 = load_nt16(W + tb * 2048 + lane * 16), q1 = load_nt16(W + tb * 2048 + 1024 + lane * 16);
        const u32x4 hdr = load_nt16(W + nblk * 128 + tb * 256 + r * 16);
        const float d = h2f_lo(hdr[0]), dmin = h2f_hi(hdr[0]);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t sb = 2 * g + h, sh = (sb & 3) * 8;
            const uint32_t lo8 = (hdr[1] >> sh) & 0xFFu, mid = (hdr[2] >> sh) & 0xFFu, hi8 = (hdr[3] >> sh) & 0xFFu;
            const bool low = sb < 4;
            const uint32_t sc = low ? (lo8 & 63u) : ((hi8 & 0xFu) | ((lo8 >> 6) << 4));
            const uint32_t mn = low ? (mid & 63u) : ((hi8 >> 4) | ((mid >> 6) << 4));
            const float d1 = d * (float)sc, m1 = dmin * (float)mn;
#pragma unroll
            for (int c = 0; c < 4; c++) {  // 8 weights = bytes 8c .. 8c+7 of the 32-byte run
                const u32x4 q = c < 2 ? q0 : q1;
                const uint32_t wa = q[2 * (c & 1)], wb = q[2 * (c & 1) + 1];
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const uint32_t byte = ((e < 4 ? wa : wb) >> (8 * (e & 3))) & 0xFFu;
                    const uint32_t nib = h ? (byte >> 4) : (byte & 0xFu);
                    o[e] = (_Float16)(d1 * (float)nib - m1);
                }
                *reinterpret_cast<f16x8 *>(orow + sb * 32 + c * 8) = o;
            }
        }
    } else {
        const u32x4 qla = load_nt16(W + tb * 3072 + lane * 16), qlb = load_nt16(W + tb * 3072 + 1024 + lane * 16);
        const u32x4 qh = load_nt16(W + tb * 3072 + 2048 + lane * 16);
        const u32x4 scv = load_nt16(W + nblk * 192 + tb * 256 + r * 16);
        const float d = (float)*reinterpret_cast<const _Float16 *>(W + nblk * 208 + tb * 32 + r * 2);
        const uint32_t n = g >> 1, lh = g & 1;
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const u32x4 ql = (qd & 1) ? qlb : qla;
            const uint32_t si = 8 * n + lh + 2 * qd;
            const uint32_t sw = si < 8 ? (si < 4 ? scv[0] : scv[1]) : (si < 12 ? scv[2] : scv[3]);
            const float dsc = d * (float)(int)(int8_t)((sw >> ((si & 3) * 8)) & 0xFFu);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const uint32_t j = c * 8 + e;
                    const uint32_t lb = (ql[j >> 2] >> (8 * (j & 3))) & 0xFFu, hb = (qh[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                    const int q = (int)(((qd >= 2) ? (lb >> 4) : (lb & 0xFu)) | (((hb >> (2 * qd)) & 3u) << 4)) - 32;
                    o[e] = (_Float16)(dsc * (float)q);
                }
                *reinterpret_cast<f16x8 *>(orow + n * 128 + qd * 32 + lh * 16 + c * 8) = o;
            }
        }
    }
    // LDS operations of a wave execute in order: the row pieces written above are complete when these reads are served
    const uint8_t *st = stage[threadIdx.x >> 6];
    const uint64_t row_bytes = (uint64_t)NB * 512;
    uint8_t *obase = reinterpret_cast<uint8_t *>(out) + (uint64_t)tile * 16 * row_bytes + (uint64_t)blk * 512;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t row = 2 * i + (lane >> 5), piece = lane & 31;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(st + row * DQ_ROW + piece * 16);
        *reinterpret_cast<u32x4 *>(obase + row * row_bytes + piece * 16) = v;
    }
}

hipError_t launch_dequant_t16_f16(const void *W, int type, uint64_t rows, uint64_t cols, void *out_f16, hipStream_t s)
{
    if (rows == 0) return hipSuccess;
    if (rows % 16 || cols % 256) return hipErrorInvalidValue;
    const uint64_t n_tiles = rows / 16, nb = cols / 256, grid = (n_tiles * nb + 3) / 4;
    if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (type == NFAI_Q4_K_T16)
        k_dequant_t16<NFAI_Q4_K_T16><<<(uint32_t)grid, 256, 0, s>>>(static_cast<const uint8_t *>(W), static_cast<_Float16 *>(out_f16), (uint32_t)n_tiles, (uint32_t)nb);
    else if (type == NFAI_Q6_K_T16)
        k_dequant_t16<NFAI_Q6_K_T16><<<(uint32_t)grid, 256, 0, s>>>(static_cast<const uint8_t *>(W), static_cast<_Float16 *>(out_f16), (uint32_t)n_tiles, (uint32_t)nb);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- dispatch ------------------------------------------------------------------------------------------
template <int QT, int MODE, int BPW>
static hipError_t q4t_launch(const KqmParams &p, int ns, uint32_t grid, uint32_t block, size_t lds, hipStream_t s)
{
    auto go = [&](auto kern) {
        static bool attr_done = false;  // per instantiation
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, p);
        return hipGetLastError();
    };
    const bool norm = p.gamma != nullptr;
    if constexpr (MODE == GEMV_QKV_ROPE) {
        if (p.begin.on) {  // (launch_gemv_kqm: only with gamma)
            switch (ns) {
                case 1: return go(k_gemv_kqt<QT, MODE, BPW, true, 1, true>);
                case 2: return go(k_gemv_kqt<QT, MODE, BPW, true, 2, true>);
            }
            return go(k_gemv_kqt<QT, MODE, BPW, true, 0, true>);
        }
    }
    switch (ns) {
        case 1: return norm ? go(k_gemv_kqt<QT, MODE, BPW, true, 1>) : go(k_gemv_kqt<QT, MODE, BPW, false, 1>);
        case 2: return norm ? go(k_gemv_kqt<QT, MODE, BPW, true, 2>) : go(k_gemv_kqt<QT, MODE, BPW, false, 2>);
    }
    return norm ? go(k_gemv_kqt<QT, MODE, BPW, true, 0>) : go(k_gemv_kqt<QT, MODE, BPW, false, 0>);
}

template <int QT, int MODE>
static hipError_t q4t_bpw(const KqmParams &p, int bpw, uint32_t grid, uint32_t block, size_t lds, hipStream_t s)
{
    // steps of the busiest workgroup; all-upfront variants exist for 1, 2 and (<= 12 waves) 4 steps
    constexpr int R = MODE == GEMV_GATEUP ? 2 : 1;
    const uint32_t max_steps = (p.NU + grid - 1) / grid * R * bpw;
    static const int env_ns = getenv("NFAI_KQM_NS") ? atoi(getenv("NFAI_KQM_NS")) : 1;
    int ns = 0;
    if (env_ns) {
        if (max_steps == 1) ns = 1;
        else if (max_steps == 2) ns = 2;
    }
    if (bpw == 1) return q4t_launch<QT, MODE, 1>(p, ns, grid, block, lds, s);
    if (bpw == 2) return q4t_launch<QT, MODE, 2>(p, ns, grid, block, lds, s);
    if (bpw == 4) return q4t_launch<QT, MODE, 4>(p, ns, grid, block, lds, s);
    return q4t_launch<QT, MODE, 8>(p, ns, grid, block, lds, s);
}

hipError_t launch_gemv_kqm(const GemvArgs &a, hipStream_t s)
{
    if (a.w_type != NFAI_Q4_K_T16 && a.w_type != NFAI_Q6_K_T16 && a.w_type != NFAI_KQ_MIXED) return hipErrorInvalidValue;
    if (a.w_type == NFAI_KQ_MIXED && a.mode != GEMV_QKV_ROPE) return hipErrorInvalidValue;
    if (a.K == 0 || a.K % 256 != 0 || a.K > 32768) return hipErrorInvalidValue;
    KqmParams p{};
    uint32_t total_tiles = 0;
    for (int i = 0; i < 3; i++) {
        if (a.seg_rows[i] % 16) return hipErrorInvalidValue;
        p.W[i] = reinterpret_cast<const uint8_t *>(a.W[i]);
        p.seg_tiles[i] = a.seg_rows[i] / 16;
        total_tiles += p.seg_tiles[i];
        p.seg_tile_end[i] = total_tiles;
    }
    if (a.mode == GEMV_QKV_ROPE) {
        if (a.D & 1u) return hipErrorInvalidValue;
        p.NU = total_tiles;
    } else if (a.mode == GEMV_GATEUP) {
        if (a.seg_rows[0] != a.seg_rows[1]) return hipErrorInvalidValue;
        p.NU = p.seg_tiles[0];
    } else {
        p.NU = p.seg_tiles[0];
    }
    if (p.NU == 0) return hipSuccess;
    p.x = a.x; p.gamma = a.gamma; p.eps = a.eps; p.K = a.K;
    p.NB = a.K / 256;
    p.y = a.y; p.res = a.res; p.kc = a.kcache; p.vc = a.vcache;
    p.kv_pos_stride = a.kv_pos_stride; p.kv_head_stride = a.kv_head_stride;
    p.rope_cs = a.rope_cs; p.rope_dims = a.rope_dims; p.D = a.D ? a.D : 2; p.pos = a.pos_dev;
    p.kv_f16 = a.kv_type == NFAI_F16;
    p.seg6 = a.seg6_mask;
    if (a.begin.on) {
        if (a.mode != GEMV_QKV_ROPE || !a.gamma || !a.begin.freqs || !a.begin.cs_out || (a.begin.emb && (!a.begin.tok || !a.begin.x_out))) return hipErrorInvalidValue;
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
    static const int env_bpw = getenv("NFAI_KQM_BPW") ? atoi(getenv("NFAI_KQM_BPW")) : 0;  // sweep knob: super-blocks per wave
    int bpw = p.NB <= 16 ? 1 : (p.NB <= 32 ? 2 : (p.NB <= 64 ? 4 : 8));  // K <= 32768 (Llama-70B: ffn length 28672)
    if ((env_bpw == 2 || env_bpw == 4) && env_bpw > bpw && p.NB % env_bpw == 0) bpw = env_bpw;
    const uint32_t nw = (p.NB + bpw - 1) / bpw;
    // workgroups per CU.  One, except for q|k|v: its 320 tiles (3B) on 256 single workgroups leave 64 of them two tiles in a row; as 320
    // workgroups, two of them share 64 CUs instead (3B Q4_K_M: 8.97 against 9.19 us per launch; the lm_head loses 3 us with two)
    static const int env_bpc_all = getenv("NFAI_KQM_BPC") ? atoi(getenv("NFAI_KQM_BPC")) : 0;  // sweep knobs
    const int env_bpc = env_bpc_all ? env_bpc_all : (a.mode == GEMV_QKV_ROPE ? 2 : 1);
    static const int env_ub = getenv("NFAI_KQM_UB") ? atoi(getenv("NFAI_KQM_UB")) : 4;
    const uint32_t grid = min(p.NU, a.n_cu * (uint32_t)max(1, min(env_bpc, 8)));
    if (a.argmax_part && grid > ARGMAX_FUSED_MAX_BLOCKS) return hipErrorInvalidValue;
    const uint32_t upb = (p.NU + grid - 1) / grid;
    // q|k|v as one workgroup per tile with more tiles than CUs (3B: 320 on 256): workgroups b and b + n_cu share a CU.  Rotating the
    // tile numbering by NU - n_cu puts the FIRST tiles (q rows, Q4_K) on the shared CUs and the last ones — the v rows, Q6_K in half
    // of a Q4_K_M file's blocks: 40 KB per tile instead of 27 — on CUs of their own (3B Q4_K_M 1071 -> 1078-1088 tokens/s, 8B 646 -> 651).
    // NFAI_KQM_ROT=n forces a rotation, -1 none.
    static const int env_rot = getenv("NFAI_KQM_ROT") ? atoi(getenv("NFAI_KQM_ROT")) : 0;
    p.rot = 0;
    if (a.mode == GEMV_QKV_ROPE) {
        if (env_rot > 0 && (uint32_t)env_rot < p.NU) p.rot = (uint32_t)env_rot;
        else if (env_rot == 0 && grid == p.NU && p.NU > a.n_cu && p.NU <= 2 * a.n_cu) p.rot = p.NU - a.n_cu;
    }
    p.UB = min((uint32_t)max(1, min(env_ub, 8)), min(upb, nw));
    const int R = a.mode == GEMV_GATEUP ? 2 : 1;
    const size_t nlay = a.w_type == NFAI_KQ_MIXED ? 2 : 1;  // fragment layouts staged
    auto lds_bytes = [&](uint32_t ub) {
        return nlay * ((size_t)nw * bpw * 1024 + (size_t)nw * bpw * 64) + (size_t)nw * bpw * 4 + (size_t)2 * ub * R * nw * 256 + 128 + 192 + BEGIN_CS_WORDS * 4;  // + 48 words of the fused ArgMax + the cos/sin table of a token's first launch
    };
    while (p.UB > 1 && lds_bytes(p.UB) > 160 * 1024) p.UB--;  // very long K: fewer units per reduction round
    const size_t lds = lds_bytes(p.UB);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    {
        static const char *names[] = {"kqm_plain", "kqm_residual", "kqm_qkv_rope", "kqm_gateup"};
        NFAI_STAMP_SET(p, names[a.mode & 3], grid, nw * 64);
    }
    if (a.w_type == NFAI_KQ_MIXED) return q4t_bpw<NFAI_KQ_MIXED, GEMV_QKV_ROPE>(p, bpw, grid, nw * 64, lds, s);
    if (a.w_type == NFAI_Q4_K_T16) {
        switch (a.mode) {
            case GEMV_PLAIN: return q4t_bpw<NFAI_Q4_K_T16, GEMV_PLAIN>(p, bpw, grid, nw * 64, lds, s);
            case GEMV_RESIDUAL: return q4t_bpw<NFAI_Q4_K_T16, GEMV_RESIDUAL>(p, bpw, grid, nw * 64, lds, s);
            case GEMV_QKV_ROPE: return q4t_bpw<NFAI_Q4_K_T16, GEMV_QKV_ROPE>(p, bpw, grid, nw * 64, lds, s);
            case GEMV_GATEUP: return q4t_bpw<NFAI_Q4_K_T16, GEMV_GATEUP>(p, bpw, grid, nw * 64, lds, s);
        }
    } else {
        switch (a.mode) {
            case GEMV_PLAIN: return q4t_bpw<NFAI_Q6_K_T16, GEMV_PLAIN>(p, bpw, grid, nw * 64, lds, s);
            case GEMV_RESIDUAL: return q4t_bpw<NFAI_Q6_K_T16, GEMV_RESIDUAL>(p, bpw, grid, nw * 64, lds, s);
            case GEMV_QKV_ROPE: return q4t_bpw<NFAI_Q6_K_T16, GEMV_QKV_ROPE>(p, bpw, grid, nw * 64, lds, s);
            case GEMV_GATEUP: return q4t_bpw<NFAI_Q6_K_T16, GEMV_GATEUP>(p, bpw, grid, nw * 64, lds, s);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace nfai
