// kernels_engine.hip — the weight-streaming engine: Wo + residual -> RMSNorm + Wgate|Wup + SiLU*up -> Wdown + residual ->
// RMSNorm + (next block's) Wq|Wk|Wv + RoPE + KV write of one TransformerBlock (TransformerBlock.cs:150-181, then :129-141 of
// the next block) as ONE launch, one workgroup per CU, instead of four launches.
//
// Why: the per-launch stamps (profiles/round2_stamps_f16_five_launch.json) show ~1.5 us of idle HBM at every kernel boundary,
// a 1-3 us head in which a launch only requests bytes and a 1-2.6 us tail in which half of the CUs have already finished.
// Inside one launch the weight stream never stops: a LOADER wave per CU runs ahead of every dependency through a ring of
// 8-KiB LDS slots filled by LDS-DMA (global_load_lds_dwordx4, non-temporal: each byte is read once by one CU), CONSUMER waves
// multiply what has landed, and the activation vector of the next projection travels between the CUs as 8-byte
// {value, tag} granules written through to memory (one sc1 store each) that a CONTROL wave per CU sweeps with sc1 loads until
// every tag carries this token's epoch — the data is the flag: no grid barrier, no counter, nothing to reset
// (MI355X guide: rows ldsdma-fill, nt-weights, allgather, engine-vs-launches; Guideline 16, R2).
//
//   workgroup = 1 loader wave | 4 consumer waves | 1 control wave, one per CU, all co-resident (LDS: >= 120 KiB each)
//   piece     = 1 KiB = 512 fp16 weights of one matrix row: one LDS-DMA wave-instruction, one ds_read_b128 per consumer lane
//   unit      = two rows that finish together (rows 2u, 2u+1; gate row u + up row u; a RoPE pair) = 2*K/512 pieces, owned by
//               ONE consumer wave, so a row is reduced inside a wave (DPP) and its epilogue needs nobody else
//   op        = one projection; the CU owns a contiguous range of its units; the pieces of all ops form ONE flat sequence
//               through the ring, so the loader is already fetching the next projection while this one waits for its input
//   slot      = 8 pieces; loader -> consumers: full_gen[slot] behind a counted vmcnt; consumers -> loader: a release mark per consumer
//               after their reads are in registers (MI355X guide, ring-gemm: a FULL word per loader, a FREE word per consumer)
//
// Numerics: the same fp16-weight x fp32-activation FMAs and fp32 epilogues as kernels_gemv.hip (RMSNorm as RMSNormShader.cs:
// 136-149, SiLU as SiLUShader.cs:121-123, RoPE as RoPEShader.cs:249-262); only the summation tree differs.  Bit-reproducible:
// no atomics on the value path, every sum has a fixed order.
// Every wait is bounded: on a timeout the workgroup raises `err`, sets its abort word and every wave leaves its loops.
#include <stdlib.h>

#include "common.h"

namespace nfai {

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

typedef uint64_t GLOBAL_AS gu64;

constexpr int ENG_NC = 4;                   // consumer waves
constexpr int ENG_WAVES = ENG_NC + 2;       // + loader + control
constexpr int ENG_THREADS = ENG_WAVES * 64;
constexpr int ENG_SLOT = 8;                 // pieces per slot (8 KiB)
constexpr int ENG_MAX_OPS = 4;
constexpr uint32_t ENG_SPIN_CAP = 1u << 21; // x s_sleep(2): tens of milliseconds, then give up

enum { ENG_RESIDUAL = 0, ENG_GATEUP = 1, ENG_QKV = 2 };

struct EngOp {
    const uint8_t *W[3];
    uint32_t seg_end[3];   // ENG_QKV: cumulative row ends of q | k | v
    uint32_t K, NU, mode;
    uint32_t x_sel;        // activation vector in LDS: 0 = XA (normalised input), 1 = XB (attention output / ffn activation)
    uint64_t *g_out;       // granules of the outputs ([row] for RESIDUAL, [unit] for GATEUP), or null
    float *y_plain;        // plain copy of the outputs for the NEXT launch (RESIDUAL), or null
};

struct EngineParams {
    EngOp op[ENG_MAX_OPS];
    uint32_t n_ops;
    uint32_t E, F, HD;
    const float *att;        // [HD]  attention output of this block (plain, written by the previous launch)
    const float *x_in;       // [E]   the block's input (residual of Wo), plain
    const float *gamma_ffn;  // [E]
    const float *gamma_next; // [E]   attn_norm gain of the next block (op 3), or null
    float eps;
    uint64_t *g_h, *g_act, *g_x;   // granule vectors: E, F, E
    const uint32_t *epoch;
    // ENG_QKV epilogue
    float *q_out;
    void *kc, *vc;
    uint64_t kv_pos_stride, kv_head_stride;
    const float *rope_cs;
    uint32_t rope_dims, D;
    const uint32_t *pos;
    int kv_f16;
    uint32_t *err;
    uint32_t nslot;          // ring slots
    uint32_t xa_off, xr_off, xb_off, ring_off;  // byte offsets in LDS
    NFAI_STAMP_PARAM
};

// ---- LDS words, accessed with inline asm: the waitcnt pass must not see them, or it would make the loader wait for its
//      LDS-DMAs (pending LDS writes to the same array) before every flag access (MI355X guide, 5.7) -------------------------
__device__ __forceinline__ uint32_t lds_ld(uint32_t addr)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);  // every lane reads the same word: the value (and every branch on it) is wave-uniform
}
__device__ __forceinline__ void lds_st(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_add(uint32_t addr, uint32_t v) { asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

constexpr uint32_t W_ABORT = 0, W_XREADY = 4, W_DONE = 8, W_REL = 16, W_FULL = 64;  // byte offsets of the control words
// W_REL: one word per consumer wave (16-byte aligned block of ENG_NC words): slot sequence numbers [0, rel) are released by it.
// W_FULL: one word per ring slot: generation (+1) whose pieces have landed.

// the smallest of the four consumers' release marks (one ds_read_b128), wave-uniform
__device__ __forceinline__ uint32_t lds_ld_min4(uint32_t addr)
{
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    const uint32_t a = min(min(v[0], v[1]), min(v[2], v[3]));
    return __builtin_amdgcn_readfirstlane(a);
}

// bounded wait until the LDS word at `addr` is >= target; false on abort / timeout
__device__ __forceinline__ bool lds_wait_ge(uint32_t addr, uint32_t target, uint32_t *err, uint32_t code)
{
    for (uint32_t spins = 0;; spins++) {
        if ((int32_t)(lds_ld(addr) - target) >= 0) return true;
        if (lds_ld(W_ABORT) != 0) return false;
        if (spins > ENG_SPIN_CAP) {
            lds_st(W_ABORT, 1);
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

template <int N> __device__ __forceinline__ void eng_wait_vmcnt()
{
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
}

__device__ __forceinline__ void eng_unit_range(uint32_t NU, uint32_t &ub, uint32_t &ue)
{
    ub = (uint32_t)(((uint64_t)NU * blockIdx.x) / gridDim.x);
    ue = (uint32_t)(((uint64_t)NU * (blockIdx.x + 1)) / gridDim.x);
}

__device__ __forceinline__ const GLOBAL_AS uint8_t *eng_row(const EngOp &o, uint32_t unit, uint32_t sub)
{
    const uint64_t rb = (uint64_t)o.K * 2;
    if (o.mode == ENG_GATEUP) return (const GLOBAL_AS uint8_t *)o.W[sub] + (uint64_t)unit * rb;
    const uint32_t row = unit * 2 + sub;
    if (o.mode == ENG_QKV) {
        if (row < o.seg_end[0]) return (const GLOBAL_AS uint8_t *)o.W[0] + (uint64_t)row * rb;
        if (row < o.seg_end[1]) return (const GLOBAL_AS uint8_t *)o.W[1] + (uint64_t)(row - o.seg_end[0]) * rb;
        return (const GLOBAL_AS uint8_t *)o.W[2] + (uint64_t)(row - o.seg_end[1]) * rb;
    }
    return (const GLOBAL_AS uint8_t *)o.W[0] + (uint64_t)row * rb;
}

// LDS index of activation element k (the layout of kernels_gemv.hip: per 512-element chunk two 1-KiB planes, lane l's two
// float4 reads at l*16 bytes in each plane: conflict-free ds_read_b128)
__device__ __forceinline__ uint32_t eng_xs_index(uint32_t k)
{
    const uint32_t chunk = k >> 9, within = k & 511;
    return (chunk << 9) + (((within >> 2) & 1) << 8) + ((within >> 3) << 2) + (within & 3);
}

// ---- loader wave: the flat piece sequence of all ops into the ring -------------------------------------------------------
// Piece order inside an op: the CU's units are taken four at a time (a "quad": one unit per consumer wave); inside a quad the
// rows advance together, eight pieces at a time: for sub-row 0/1, for each group of <= 8 chunks, for consumer 0..3.  So the four
// consumers walk the ring side by side whatever the row length (a Wdown row of Llama-3.1-8B alone is 28 KiB).
//   offset of (sub, c0, w) inside a quad of nuq units = (sub * KC + c0) * nuq + w * min(8, KC - c0)
template <int AHEAD>
__device__ __forceinline__ void eng_loader(const EngineParams &p, uint8_t *lds, uint32_t lane)
{
    const uint32_t nslot = p.nslot, ring_pieces = nslot * ENG_SLOT;
    // position of the next piece (all wave-uniform): ring index rp, `within` pieces of the current slot issued
    uint32_t rp = 0, within = 0;
    uint32_t seq = 0;           // slot sequence number being filled
    uint32_t published = 0;     // slot sequence numbers [0, published) are marked full
    uint32_t ps = 0, pgen = 0;  // slot / generation of sequence number `published`
    LDS_AS uint8_t *ring = (LDS_AS uint8_t *)(lds + p.ring_off);
    bool ok = true;
    STAMP_DECL;
    STAMP(0);  // loader: start | last piece of op 0..3 issued (1..4) | everything landed (5)
    // n consecutive pieces of one row, split where they cross a slot boundary
    auto issue = [&](const GLOBAL_AS uint8_t *src, uint32_t n) {
        while (n && ok) {
            if (within == 0 && seq >= nslot) {  // a new slot: every consumer must have released its previous occupant (seq - nslot)
                const uint32_t need = seq - nslot + 1;
                for (uint32_t spins = 0; (int32_t)(lds_ld_min4(W_REL) - need) < 0; spins++) {
                    if (lds_ld(W_ABORT) != 0) { ok = false; break; }
                    if (spins > ENG_SPIN_CAP) {
                        lds_st(W_ABORT, 1);
                        if (lane == 0) __hip_atomic_fetch_or(p.err, 0x10u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (!ok) break;
            }
            const uint32_t run = min(n, (uint32_t)ENG_SLOT - within);
            for (uint32_t j = 0; j < run; j++) {
                __builtin_amdgcn_global_load_lds(src, ring + (rp + j) * 1024, 16, 0, 2);  // aux 2 = nt
                src += 1024;
            }
            n -= run;
            rp += run;
            within += run;
            if (within == ENG_SLOT) {  // slot `seq` issued completely: the slot AHEAD behind it has landed
                within = 0;
                if (rp == ring_pieces) rp = 0;
                if (seq >= (uint32_t)AHEAD) {
                    eng_wait_vmcnt<ENG_SLOT * AHEAD>();
                    lds_st(W_FULL + ps * 4, pgen + 1);
                    published++;
                    if (++ps == nslot) { ps = 0; pgen++; }
                }
                seq++;
            }
        }
    };
    for (uint32_t oi = 0; oi < p.n_ops && ok; oi++) {
        const EngOp &o = p.op[oi];
        uint32_t ub, ue;
        eng_unit_range(o.NU, ub, ue);
        const uint32_t KC = o.K >> 9, nu = ue - ub;
        for (uint32_t q0 = 0; q0 < nu && ok; q0 += ENG_NC) {
            const uint32_t nuq = min((uint32_t)ENG_NC, nu - q0);
            for (uint32_t sub = 0; sub < 2 && ok; sub++)
                for (uint32_t c0 = 0; c0 < KC && ok; c0 += 8) {
                    const uint32_t n = min(8u, KC - c0);
                    for (uint32_t w = 0; w < nuq; w++)  // the row base is wave-uniform arithmetic: recomputed, not kept in an array
                        issue(eng_row(o, ub + q0 + w, sub) + (uint64_t)c0 * 1024 + lane * 16, n);
                }
        }
#ifdef NFAI_STAMPS
        if (oi == 0) STAMP(1); else if (oi == 1) STAMP(2); else if (oi == 2) STAMP(3); else STAMP(4);
#endif
    }
    // drain: everything issued has landed; publish the remaining slots (the last one may be partial)
    eng_wait_vmcnt<0>();
    const uint32_t total = seq + (within ? 1u : 0u);
    for (; published < total; published++) {
        lds_st(W_FULL + ps * 4, pgen + 1);
        if (++ps == nslot) { ps = 0; pgen++; }
    }
    STAMP(5);
    STAMP_FLUSH(p.stamps, blockIdx.x * ENG_WAVES, 6);
}

__device__ __forceinline__ void eng_publish(uint64_t *g, uint32_t idx, uint32_t epoch, float v)
{
    __hip_atomic_store((gu64 *)g + idx, ((uint64_t)epoch << 32) | (uint64_t)__builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);  // one global_store_dwordx2 sc1: written through, value and tag together
}

// ---- consumer wave w: its units of every op ---------------------------------------------------------------------------
__device__ __forceinline__ void eng_consumer(const EngineParams &p, uint8_t *lds, uint32_t w, uint32_t lane, uint32_t epoch)
{
    const uint32_t nslot = p.nslot, ring_pieces = nslot * ENG_SLOT;
    const uint8_t *ring = lds + p.ring_off;
    const float *XR = reinterpret_cast<const float *>(lds + p.xr_off);
    uint32_t rel = 0;     // slot sequence numbers [0, rel) released by this wave
    uint32_t gbase = 0;   // first piece of the current op
    auto release_below = [&](uint32_t S) {  // this wave will not read slot sequence numbers < S again
        if (S > rel) {
            rel = S;
            lds_st(W_REL + w * 4, rel);
        }
    };
    const uint32_t pos = p.pos ? ((const GLOBAL_AS uint32_t *)p.pos)[0] : 0u;
    bool ok = true;
    STAMP_DECL;  // consumer: activation of op i seen (2i) | last unit of op i finished (2i + 1)
    for (uint32_t oi = 0; oi < p.n_ops && ok; oi++) {
        const EngOp &o = p.op[oi];
        uint32_t ub, ue;
        eng_unit_range(o.NU, ub, ue);
        const uint32_t KC = o.K >> 9, nu = ue - ub;
        // nothing before this wave's first piece of the op will be read by it again
        release_below((gbase + (w < nu ? w * min(8u, KC) : nu * 2 * KC)) / ENG_SLOT);
        if (!lds_wait_ge(W_XREADY, oi + 1, p.err, 0x20u)) { ok = false; break; }
        const float *xs = reinterpret_cast<const float *>(lds + (o.x_sel ? p.xb_off : p.xa_off));
#ifdef NFAI_STAMPS
        if (oi == 0) STAMP(0); else if (oi == 1) STAMP(2); else if (oi == 2) STAMP(4); else STAMP(6);
#endif
        for (uint32_t q0 = 0; q0 + w < nu && ok; q0 += ENG_NC) {
            const uint32_t nuq = min((uint32_t)ENG_NC, nu - q0), qbase = gbase + q0 * 2 * KC;
            const uint32_t u = ub + q0 + w;
            // what the epilogue reads from memory is requested now (RoPE pair)
            float cs0 = 1.f, cs1 = 0.f;
            uint32_t seg = 0, r = 0;
            if (o.mode == ENG_QKV) {
                const uint32_t row = u * 2;
                seg = row < o.seg_end[0] ? 0u : (row < o.seg_end[1] ? 1u : 2u);
                r = seg == 0 ? row : (seg == 1 ? row - o.seg_end[0] : row - o.seg_end[1]);
                const uint32_t d = min(r % p.D, max(p.rope_dims, 2u) - 2);
                const f32x2 cs = *reinterpret_cast<const GLOBAL_AS f32x2 *>((const GLOBAL_AS float *)p.rope_cs + d);
                cs0 = cs[0];
                cs1 = cs[1];
            }
            float acc[2] = {0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                float a = 0.f;
                for (uint32_t c0 = 0; c0 < KC && ok; c0 += 8) {
                    const uint32_t n = min(8u, KC - c0);
                    const uint32_t gfirst = qbase + ((uint32_t)sub * KC + c0) * nuq + w * n, glast = gfirst + n - 1;
                    release_below(gfirst / ENG_SLOT);
                    const uint32_t S = glast / ENG_SLOT, Sgen = S / nslot;
                    if (!lds_wait_ge(W_FULL + (S - Sgen * nslot) * 4, Sgen + 1, p.err, 0x40u)) { ok = false; break; }
                    const uint32_t rp0 = gfirst % ring_pieces;
#pragma unroll 4
                    for (uint32_t j = 0; j < n; j++) {
                        const uint32_t c = c0 + j;
                        uint32_t rp = rp0 + j;
                        rp = rp >= ring_pieces ? rp - ring_pieces : rp;
                        const u32x4 wv = *reinterpret_cast<const u32x4 *>(ring + rp * 1024 + lane * 16);
                        const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xs + (c << 9) + (lane << 2));
                        const f32x4 x1 = *reinterpret_cast<const f32x4 *>(xs + (c << 9) + 256 + (lane << 2));
                        a = dot8_f16(wv, x0, x1, a);
                    }
                }
                acc[sub] = a;
            }
            if (!ok) break;
            const float a0 = wave_sum(acc[0]), a1 = wave_sum(acc[1]);
            if (lane == 0) {
                if (o.mode == ENG_RESIDUAL) {
                    // host residual add of TransformerBlock.cs:153-158 / 176-180: input + projection
                    const float y0 = XR[2 * u] + a0, y1 = XR[2 * u + 1] + a1;
                    if (o.g_out) { eng_publish(o.g_out, 2 * u, epoch, y0); eng_publish(o.g_out, 2 * u + 1, epoch, y1); }
                    if (o.y_plain) { o.y_plain[2 * u] = y0; o.y_plain[2 * u + 1] = y1; }
                } else if (o.mode == ENG_GATEUP) {
                    const float v = a1 * silu_ref(a0);  // SiLUShader.cs:121-123, ElementWiseMultiplicationShader.cs:137
                    eng_publish(o.g_out, u, epoch, v);
                } else {
                    // RoPEShader.cs:249-262 on the pair (row, row+1); V rows are stored unrotated
                    const uint32_t head = r / p.D, d = r % p.D, row = u * 2;
                    float o0 = a0, o1 = a1;
                    if (seg < 2 && d < p.rope_dims) {
                        o0 = cs0 * a0 - cs1 * a1;
                        o1 = cs1 * a0 + cs0 * a1;
                    }
                    if (seg == 0) {
                        p.q_out[row] = o0;
                        p.q_out[row + 1] = o1;
                    } else {
                        const uint64_t idx = (uint64_t)pos * p.kv_pos_stride + (uint64_t)head * p.kv_head_stride + d;
                        void *base = seg == 1 ? p.kc : p.vc;
                        if (p.kv_f16) {
                            reinterpret_cast<_Float16 *>(base)[idx] = (_Float16)o0;
                            reinterpret_cast<_Float16 *>(base)[idx + 1] = (_Float16)o1;
                        } else {
                            reinterpret_cast<float *>(base)[idx] = o0;
                            reinterpret_cast<float *>(base)[idx + 1] = o1;
                        }
                    }
                }
            }
        }
        gbase += nu * 2 * KC;
#ifdef NFAI_STAMPS
        if (oi == 0) STAMP(1); else if (oi == 1) STAMP(3); else if (oi == 2) STAMP(5); else STAMP(7);
#endif
        if (lane == 0) lds_add(W_DONE, 1);  // this wave's outputs of the op are on their way
    }
    release_below((gbase + ENG_SLOT - 1) / ENG_SLOT);
    STAMP_FLUSH(p.stamps, blockIdx.x * ENG_WAVES + 1 + w, 8);
}

// ---- control wave: gathers ------------------------------------------------------------------------------------------------
// Sweep n granules (n % 128 == 0) at g[] until every tag == epoch; values -> dst[index(k)] in LDS.  Returns the sum of squares.
template <bool PERMUTE>
__device__ __forceinline__ bool eng_gather(const uint64_t *g, uint32_t n, uint32_t epoch, float *dst, uint32_t lane, uint32_t *err,
                                           float &ss_out)
{
    // 16-byte sc1 loads (two granules per lane): L1 is bypassed, every pass reads what has reached memory
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, (int)(n * 8), 0x00020000);
    float ss = 0.f;
    const uint32_t nloads = n / 128;  // per lane
    for (uint32_t l0 = 0; l0 < nloads; l0 += 8) {
        const uint32_t nl = min(8u, nloads - l0);
        u32x4 v[8];
        for (uint32_t spins = 0;; spins++) {
            bool okc = true;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t li = l0 + min((uint32_t)k, nl - 1);
                v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((li * 64 + lane) * 16), 0, 16));
            }
#pragma unroll
            for (int k = 0; k < 8; k++) okc = okc && v[k][1] == epoch && v[k][3] == epoch;
            if (__all(okc)) break;
            if (lds_ld(W_ABORT) != 0) return false;
            if (spins > (ENG_SPIN_CAP >> 5)) {  // a pass is a memory round trip (~2 us): the same tens of milliseconds
                lds_st(W_ABORT, 1);
                if (lane == 0) __hip_atomic_fetch_or(err, 0x80u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if ((uint32_t)k < nl) {
                const uint32_t e = ((l0 + k) * 64 + lane) * 2;  // element index of the first of the two granules
                const f32x4 vf = __builtin_bit_cast(f32x4, v[k]);  // whole-vector cast (on an ELEMENT lvalue hipcc 7.2 reads element 0)
                const float a = vf[0], b = vf[2];
                ss = fmaf(a, a, ss);
                ss = fmaf(b, b, ss);
                const uint32_t i = PERMUTE ? eng_xs_index(e) : e;  // e is even: the pair stays adjacent under the permutation
                *reinterpret_cast<f32x2 *>(dst + i) = f32x2{a, b};
            }
        }
    }
    ss_out = ss;
    return true;
}

// XA[perm(k)] = (XR[k] / rms) * gamma[k]   (RMSNormShader.cs:136-149)
__device__ __forceinline__ void eng_norm(const float *XR, const float *gamma, float *XA, uint32_t E, float ss, float eps, uint32_t lane)
{
    const float rms = sqrtf(wave_sum(ss) / (float)E + eps);
    for (uint32_t k = lane * 4; k < E; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(XR + k);
        const f32x4 gm = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)gamma + k);
        f32x4 o;
        o[0] = (v[0] / rms) * gm[0];
        o[1] = (v[1] / rms) * gm[1];
        o[2] = (v[2] / rms) * gm[2];
        o[3] = (v[3] / rms) * gm[3];
        *reinterpret_cast<f32x4 *>(XA + eng_xs_index(k)) = o;
    }
}

__device__ __forceinline__ void eng_control(const EngineParams &p, uint8_t *lds, uint32_t lane, uint32_t epoch)
{
    float *XA = reinterpret_cast<float *>(lds + p.xa_off), *XR = reinterpret_cast<float *>(lds + p.xr_off);
    float *XB = reinterpret_cast<float *>(lds + p.xb_off);
    STAMP_DECL;  // control: start | x of op 0 set | local consumers done with op 0 | h gathered | done op 1 | act gathered | done op 2 | x gathered
    STAMP(0);
    // op 0 (Wo + residual): attention output -> XB, block input -> XR; both are plain vectors of the previous launch
    for (uint32_t k = lane * 4; k < p.HD; k += 256)
        *reinterpret_cast<f32x4 *>(XB + eng_xs_index(k)) = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.att + k);
    for (uint32_t k = lane * 4; k < p.E; k += 256)
        *reinterpret_cast<f32x4 *>(XR + k) = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.x_in + k);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    lds_st(W_XREADY, 1);
    STAMP(1);
    float ss;
    // edge h = x + Wo.att: every CU's rows -> XR (raw, the residual of Wdown) and XA = RMSNorm(h) * ffn_norm
    if (!lds_wait_ge(W_DONE, ENG_NC * 1, p.err, 0x100u)) return;
    STAMP(2);
    if (!eng_gather<false>(p.g_h, p.E, epoch, XR, lane, p.err, ss)) return;
    eng_norm(XR, p.gamma_ffn, XA, p.E, ss, p.eps, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_st(W_XREADY, 2);
    STAMP(3);
    // edge act = up * silu(gate) -> XB
    if (!lds_wait_ge(W_DONE, ENG_NC * 2, p.err, 0x200u)) return;
    STAMP(4);
    if (!eng_gather<true>(p.g_act, p.F, epoch, XB, lane, p.err, ss)) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_st(W_XREADY, 3);
    STAMP(5);
    if (p.n_ops < 4) { STAMP_FLUSH(p.stamps, blockIdx.x * ENG_WAVES + 1 + ENG_NC, 6); return; }
    // edge x' = h + Wdown.act -> XA = RMSNorm(x') * attn_norm of the next block
    if (!lds_wait_ge(W_DONE, ENG_NC * 3, p.err, 0x400u)) return;
    STAMP(6);
    if (!eng_gather<false>(p.g_x, p.E, epoch, XR, lane, p.err, ss)) return;
    eng_norm(XR, p.gamma_next, XA, p.E, ss, p.eps, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_st(W_XREADY, 4);
    STAMP(7);
    STAMP_FLUSH(p.stamps, blockIdx.x * ENG_WAVES + 1 + ENG_NC, 8);
}

template <int AHEAD>
__global__ __launch_bounds__(ENG_THREADS) void k_engine(const EngineParams p)
{
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // control words start at zero: wave 0 clears them, everyone meets once (the only workgroup barrier of the kernel)
    if (wave == 0) {
        for (uint32_t i = lane; i < 64; i += 64) reinterpret_cast<uint32_t *>(lds)[i] = 0;
    }
    __syncthreads();
    const uint32_t epoch = ((const GLOBAL_AS uint32_t *)p.epoch)[0];
    if (wave == 0) eng_loader<AHEAD>(p, lds, lane);
    else if (wave <= ENG_NC) eng_consumer(p, lds, wave - 1, lane, epoch);
    else eng_control(p, lds, lane, epoch);
}

// ---- host side ----------------------------------------------------------------------------------------------------------
struct EngineArgs;  // common.h

hipError_t launch_engine(const EngineArgs &a, hipStream_t s)
{
    EngineParams p{};
    if (a.n_ops < 3 || a.n_ops > 4) return hipErrorInvalidValue;
    const uint32_t E = a.E, F = a.F, HD = a.HD;
    if (E % 512 || F % 512 || HD % 512 || E % 128 || F % 128) return hipErrorInvalidValue;
    p.n_ops = a.n_ops;
    p.E = E; p.F = F; p.HD = HD;
    // op 0: Wo + residual
    p.op[0].W[0] = static_cast<const uint8_t *>(a.Wo); p.op[0].K = HD; p.op[0].NU = E / 2; p.op[0].mode = ENG_RESIDUAL;
    p.op[0].x_sel = 1; p.op[0].g_out = a.g_h; p.op[0].y_plain = nullptr;
    // op 1: gate | up
    p.op[1].W[0] = static_cast<const uint8_t *>(a.Wgate); p.op[1].W[1] = static_cast<const uint8_t *>(a.Wup);
    p.op[1].K = E; p.op[1].NU = F; p.op[1].mode = ENG_GATEUP; p.op[1].x_sel = 0; p.op[1].g_out = a.g_act;
    // op 2: Wdown + residual
    p.op[2].W[0] = static_cast<const uint8_t *>(a.Wdown); p.op[2].K = F; p.op[2].NU = E / 2; p.op[2].mode = ENG_RESIDUAL;
    p.op[2].x_sel = 1; p.op[2].g_out = a.n_ops == 4 ? a.g_x : nullptr; p.op[2].y_plain = a.x_out;
    if (a.n_ops == 4) {
        const uint32_t rows = a.qkv_rows[0] + a.qkv_rows[1] + a.qkv_rows[2];
        if ((a.qkv_rows[0] | a.qkv_rows[1] | a.qkv_rows[2] | a.D) & 1u) return hipErrorInvalidValue;
        for (int i = 0; i < 3; i++) p.op[3].W[i] = static_cast<const uint8_t *>(a.Wqkv[i]);
        p.op[3].seg_end[0] = a.qkv_rows[0];
        p.op[3].seg_end[1] = a.qkv_rows[0] + a.qkv_rows[1];
        p.op[3].seg_end[2] = rows;
        p.op[3].K = E; p.op[3].NU = rows / 2; p.op[3].mode = ENG_QKV; p.op[3].x_sel = 0;
    }
    p.att = a.att; p.x_in = a.x_in; p.gamma_ffn = a.gamma_ffn; p.gamma_next = a.gamma_next; p.eps = a.eps;
    p.g_h = a.g_h; p.g_act = a.g_act; p.g_x = a.g_x; p.epoch = a.epoch;
    p.q_out = a.q_out; p.kc = a.kcache; p.vc = a.vcache; p.kv_pos_stride = a.kv_pos_stride; p.kv_head_stride = a.kv_head_stride;
    p.rope_cs = a.rope_cs; p.rope_dims = a.rope_dims; p.D = a.D; p.pos = a.pos_dev; p.kv_f16 = a.kv_type == NFAI_F16;
    p.err = a.err;
    // LDS: control words | XA (E) | XR (E) | XB (max(HD, F)) | ring
    const uint32_t xb = (HD > F ? HD : F);
    p.xa_off = 1024;
    p.xr_off = p.xa_off + E * 4;
    p.xb_off = p.xr_off + E * 4;
    p.ring_off = (p.xb_off + xb * 4 + 1023) & ~1023u;
    const uint32_t lds_cap = 160 * 1024;
    if (p.ring_off + 5 * ENG_SLOT * 1024 > lds_cap) return hipErrorInvalidValue;
    uint32_t nslot = (lds_cap - p.ring_off) / (ENG_SLOT * 1024);
    if (nslot > 16) nslot = 16;
    static const int env_slots = getenv("NFAI_ENGINE_SLOTS") ? atoi(getenv("NFAI_ENGINE_SLOTS")) : 0;
    if (env_slots >= 5 && (uint32_t)env_slots <= nslot) nslot = (uint32_t)env_slots;
    p.nslot = nslot;
    const uint32_t lds_bytes = p.ring_off + nslot * ENG_SLOT * 1024;
    // slots of loads kept in flight (x 8 KiB): the ring keeps >= 3 slots beyond them (being read / waiting / being released)
    int ahead = nslot >= 9 ? 6 : (nslot >= 7 ? 4 : 2);
    static const int env_ahead = getenv("NFAI_ENGINE_AHEAD") ? atoi(getenv("NFAI_ENGINE_AHEAD")) : 0;
    if ((env_ahead == 2 || env_ahead == 4 || env_ahead == 6) && (uint32_t)env_ahead + 3 <= nslot) ahead = env_ahead;
    NFAI_STAMP_SET(p, "engine", a.n_cu, ENG_THREADS);
    auto launch = [&](auto kern) -> hipError_t {
        static bool attr_set[8] = {false, false, false, false, false, false, false, false};
        if (!attr_set[ahead]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
            if (e != hipSuccess) return e;
            attr_set[ahead] = true;
        }
        hipLaunchKernelGGL(kern, dim3(a.n_cu), dim3(ENG_THREADS), lds_bytes, s, p);
        return hipGetLastError();
    };
    if (ahead == 6) return launch(k_engine<6>);
    if (ahead == 4) return launch(k_engine<4>);
    return launch(k_engine<2>);
}

}  // namespace nfai
