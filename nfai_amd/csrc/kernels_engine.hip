// kernels_engine.hip — the weight-streaming engine: Wo + residual -> RMSNorm + Wgate|Wup + SiLU*up -> Wdown + residual ->
// RMSNorm + (next block's) Wq|Wk|Wv + RoPE + KV write of one TransformerBlock (TransformerBlock.cs:150-181, then :129-141 of
// the next block) as ONE launch, one workgroup per CU, instead of four launches.
//
// Why: the per-launch stamps (profiles/round2_stamps_f16_five_launch.json) show ~1.5 us of idle HBM at every kernel boundary,
// a 1-3 us head in which a launch only requests bytes and a 1-2.6 us tail in which half of the CUs have already finished.
// Inside one launch the weight stream never stops: a LOADER wave per CU runs ahead of every dependency through a ring of
// 8-KiB LDS slots filled by LDS-DMA (global_load_lds_dwordx4, non-temporal: each byte is read once by one CU), CONSUMER waves
// multiply what has landed, and the activation vector of the next projection travels between the CUs as 8-byte
// {value, tag} granules written through to memory (one sc1 store each) that the consumer waves sweep with sc1 loads — each its
// share of the vector — until every tag carries this token's epoch: the data is the flag, no grid barrier, no counter, nothing
// to reset (MI355X guide: rows ldsdma-fill, nt-weights, allgather, engine-vs-launches; Guideline 16, R2).
//
//   workgroup = 1 loader wave + ENG_NC consumer waves, one workgroup per CU, all co-resident (LDS: >= 120 KiB each)
//   piece     = 1 KiB = 512 fp16 weights of one matrix row: one LDS-DMA wave-instruction, one ds_read_b128 per consumer lane
//   unit      = two rows that finish together (rows 2u, 2u+1; gate row u + up row u; a RoPE pair), owned by ONE consumer wave:
//               a row is reduced inside a wave (DPP), both rows share every activation read, the epilogue needs nobody else
//   op        = one projection; the CU owns a contiguous range of its units; the pieces of all ops form ONE flat sequence
//               through the ring, so the loader is already fetching the next projection while this one waits for its input
//   slot      = 8 pieces; loader -> consumers: one monotonic count of landed slots (behind a counted vmcnt); consumers ->
//               loader: a release mark per consumer wave (MI355X guide, ring-gemm: FULL / FREE words in LDS)
//   edge      = between two ops: every consumer wave gathers its share of the granule vector, the waves of the CU meet at an
//               LDS counter (sum of squares for the RMSNorm, then the normalised vector), and go on
//
// Numerics: the same fp16-weight x fp32-activation FMAs and fp32 epilogues as kernels_gemv.hip (RMSNorm as RMSNormShader.cs:
// 136-149, SiLU as SiLUShader.cs:121-123, RoPE as RoPEShader.cs:249-262); only the summation tree differs.  Bit-reproducible:
// no atomics on the value path, every sum has a fixed order.
// Every wait is bounded: on a timeout the workgroup raises `err`, sets its abort word and every wave leaves its loops.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace nfai {

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

typedef uint64_t GLOBAL_AS gu64;

constexpr int ENG_NC = 4;                   // consumer waves
constexpr int ENG_WAVES = ENG_NC + 1;       // + loader
constexpr int ENG_THREADS = ENG_WAVES * 64;
constexpr int ENG_SLOT = 8;                 // pieces per slot (8 KiB)
constexpr int ENG_MAX_OPS = 4;
constexpr int ENG_GL = 16;                  // 16-byte loads per lane and gather chunk
constexpr uint32_t ENG_SPIN_CAP = 1u << 21; // x s_sleep: tens of milliseconds, then give up

enum { ENG_RESIDUAL = 0, ENG_GATEUP = 1, ENG_QKV = 2 };

struct EngOp {
    const uint8_t *W[3];
    uint32_t seg_end[3];   // ENG_QKV: cumulative row ends of q | k | v
    uint32_t K, NU, mode;
    uint32_t x_sel;        // activation vector in LDS: 0 = XA (normalised input), 1 = XB (attention output / ffn activation)
    uint64_t *g_out;       // granules of the outputs ([row] for RESIDUAL, [unit] for GATEUP), or null
    float *y_plain;        // plain copy of the outputs for the NEXT launch (RESIDUAL), or null
};

struct EngineParams {   // by value in the kernel argument: constant loads the compiler may keep in SGPRs across the asm waits
    const EngOp *ops;    // [n_ops] in device memory: op i is picked with a run-time index (a run-time index into the by-value
                         // argument would make hipcc copy the whole block to scratch; scratch accesses count on vmcnt, which the
                         // loader counts by hand); each wave copies the op it works on into registers once
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
//      LDS-DMAs (pending LDS writes to the same array) before every flag access (MI355X guide, 5.7).  Every lane reads the
//      same word, so the value (and every branch on it) is made wave-uniform. -----------------------------------------------
__device__ __forceinline__ uint32_t lds_ld(uint32_t addr)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void lds_st(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_add(uint32_t addr, uint32_t v) { asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// the smallest of the consumers' release marks (ds_read_b128 per four of them)
__device__ __forceinline__ uint32_t lds_ld_min_rel(uint32_t addr)
{
    uint32_t m = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < ENG_NC; i += 4) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr + i * 4) : "memory");
        m = min(m, min(min(v[0], v[1]), min(v[2], v[3])));
    }
    return __builtin_amdgcn_readfirstlane(m);
}

// byte offsets of the control words (the first KiB of LDS)
constexpr uint32_t W_ABORT = 0;     // non-zero: a wait gave up, everybody leaves
constexpr uint32_t W_PUB = 4;       // loader: slot sequence numbers [0, pub) have landed
constexpr uint32_t W_ARRIVE = 8;    // consumers' meeting counter (monotonic)
constexpr uint32_t W_REL = 32;      // per consumer wave: slot sequence numbers [0, rel) will not be read by it again
constexpr uint32_t W_SS = 128;      // per consumer wave: its share of the sum of squares of the vector being gathered
static_assert(ENG_NC % 4 == 0 && ENG_NC <= 16, "release marks are read four at a time");

__device__ __forceinline__ bool eng_give_up(uint32_t *err, uint32_t code)
{
    lds_st(W_ABORT, 1);
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// bounded wait until the LDS word at `addr` is >= target; returns the value seen (>= target), or ~0 on abort / timeout
__device__ __forceinline__ bool lds_wait_ge(uint32_t addr, uint32_t target, uint32_t *err, uint32_t code, uint32_t &seen)
{
    for (uint32_t spins = 0;; spins++) {
        seen = lds_ld(addr);
        if ((int32_t)(seen - target) >= 0) return true;
        if (lds_ld(W_ABORT) != 0) return false;
        if (spins > ENG_SPIN_CAP) return eng_give_up(err, code);
        __builtin_amdgcn_s_sleep(1);
    }
}

template <int N> __device__ __forceinline__ void eng_wait_vmcnt()
{
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
}

// Op i of the launch as a register copy.  The table was written before the launch and nobody writes it: read through the
// CONSTANT address space, so that hipcc uses scalar loads (through a plain global pointer it emits a vector load and waits
// vmcnt(0) for it: a memory round trip on every op boundary, and in the loader a drain of the LDS-DMAs in flight).
__device__ __forceinline__ EngOp eng_load_op(const EngOp *ops, uint32_t i)
{
    typedef const uint32_t __attribute__((address_space(4))) cu32;
    static_assert(sizeof(EngOp) % 4 == 0, "copied as dwords");
    cu32 *src = (cu32 *)(ops + i);
    uint32_t tmp[sizeof(EngOp) / 4];
#pragma unroll
    for (uint32_t k = 0; k < sizeof(EngOp) / 4; k++) tmp[k] = src[k];
    EngOp o;
    __builtin_memcpy(&o, tmp, sizeof(o));  // not a uint32_t* view of the struct: that would be an aliasing violation (undefined pointers)
    return o;
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
// Piece order inside an op: the CU's units are taken ENG_NC at a time (a "round": one unit per consumer wave); inside a round
// the rows advance together, eight chunks at a time: for each group of <= 8 chunks, for consumer 0..NC-1: the group's pieces of
// the unit's first row, then of its second row.  So the consumers walk the ring side by side whatever the row length (a Wdown
// row of Llama-3.1-8B alone is 28 KiB), and a consumer finds the two rows of a chunk range next to each other.
//   offset of (c0, w) inside a round of nur units = c0 * 2 * nur + w * 2 * n,  n = min(8, KC - c0); second row at + n
template <int AHEAD>
__device__ __forceinline__ void eng_loader(const EngineParams &p, uint8_t *lds, uint32_t lane)
{
    const uint32_t nslot = p.nslot, ring_pieces = nslot * ENG_SLOT;
    uint32_t rp = 0, within = 0;  // ring index of the next piece; pieces of the current slot already issued
    uint32_t seq = 0;             // slot sequence number being filled
    uint32_t published = 0;       // slot sequence numbers [0, published) have been declared landed
    uint32_t min_rel = 0;         // last value seen of the consumers' smallest release mark
    LDS_AS uint8_t *ring = (LDS_AS uint8_t *)(lds + p.ring_off);
    bool ok = true;
    __builtin_amdgcn_s_setprio(3);  // the loader's few scalar / memory instructions go first on the SIMD it shares with a consumer
    STAMP_DECL;
    STAMP(0);  // loader: start | last piece of op 0..3 issued (1..4) | everything landed (5)
#ifdef NFAI_STAMPS
    unsigned long long t_free = 0, t_mem = 0;  // ticks spent waiting for a free slot (consumers) / for loads to land (memory)
#endif
    // n consecutive pieces of one row, split where they cross a slot boundary
    auto issue = [&](const GLOBAL_AS uint8_t *src, uint32_t n) {
        while (n && ok) {
            if (within == 0 && seq >= nslot && (int32_t)(min_rel - (seq - nslot + 1)) < 0) {
                // a new slot: every consumer must have released its previous occupant (sequence number seq - nslot)
#ifdef NFAI_STAMPS
                const unsigned long long ta = __builtin_amdgcn_s_memrealtime();
#endif
                for (uint32_t spins = 0;; spins++) {
                    min_rel = lds_ld_min_rel(W_REL);
                    if ((int32_t)(min_rel - (seq - nslot + 1)) >= 0) break;
                    if (lds_ld(W_ABORT) != 0) { ok = false; break; }
                    if (spins > ENG_SPIN_CAP) { ok = eng_give_up(p.err, 0x10u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#ifdef NFAI_STAMPS
                t_free += __builtin_amdgcn_s_memrealtime() - ta;
#endif
                if (!ok) break;
            }
            const uint32_t run = min(n, (uint32_t)ENG_SLOT - within);
#if 0  // one address pair per piece (kept for reference: 7 instructions per piece instead of ~3)
            for (uint32_t j = 0; j < run; j++) {
                __builtin_amdgcn_global_load_lds(src, ring + (rp + j) * 1024, 16, 0, 2);  // aux 2 = nt
                src += 1024;
            }
#else
            // the instruction's immediate offset applies to the global AND the LDS address: up to four pieces per address pair
            for (uint32_t j = 0; j < run; j += 4) {
                const GLOBAL_AS uint8_t *sp = src + (uint64_t)j * 1024;
                LDS_AS uint8_t *dst = ring + (rp + j) * 1024;
                const uint32_t m = run - j;
                __builtin_amdgcn_global_load_lds(sp, dst, 16, 0, 2);  // aux 2 = nt
                if (m > 1) __builtin_amdgcn_global_load_lds(sp, dst, 16, 1024, 2);
                if (m > 2) __builtin_amdgcn_global_load_lds(sp, dst, 16, 2048, 2);
                if (m > 3) __builtin_amdgcn_global_load_lds(sp, dst, 16, 3072, 2);
            }
            src += (uint64_t)run * 1024;
#endif
            n -= run;
            rp += run;
            within += run;
            if (within == ENG_SLOT) {  // slot `seq` issued completely: the slot AHEAD behind it has landed
                within = 0;
                if (rp == ring_pieces) rp = 0;
                if (seq >= (uint32_t)AHEAD) {
#ifdef NFAI_STAMPS
                    const unsigned long long ta = __builtin_amdgcn_s_memrealtime();
#endif
                    eng_wait_vmcnt<ENG_SLOT * AHEAD>();
#ifdef NFAI_STAMPS
                    t_mem += __builtin_amdgcn_s_memrealtime() - ta;
#endif
                    lds_st(W_PUB, ++published);
                }
                seq++;
            }
        }
    };
    for (uint32_t oi = 0; oi < p.n_ops && ok; oi++) {
        const EngOp o = eng_load_op(p.ops, oi);
        uint32_t ub, ue;
        eng_unit_range(o.NU, ub, ue);
        const uint32_t KC = o.K >> 9, nu = ue - ub;
        for (uint32_t r0 = 0; r0 < nu && ok; r0 += ENG_NC) {
            const uint32_t nur = min((uint32_t)ENG_NC, nu - r0);
            for (uint32_t c0 = 0; c0 < KC && ok; c0 += 8) {
                const uint32_t n = min(8u, KC - c0);
                for (uint32_t w = 0; w < nur; w++) {  // the row bases are wave-uniform arithmetic: recomputed, not kept in an array
                    issue(eng_row(o, ub + r0 + w, 0) + (uint64_t)c0 * 1024 + lane * 16, n);
                    issue(eng_row(o, ub + r0 + w, 1) + (uint64_t)c0 * 1024 + lane * 16, n);
                }
            }
        }
#ifdef NFAI_STAMPS
        if (oi == 0) STAMP(1); else if (oi == 1) STAMP(2); else if (oi == 2) STAMP(3); else STAMP(4);
#endif
    }
    // drain: everything issued has landed; the last slot may be partial
    eng_wait_vmcnt<0>();
    lds_st(W_PUB, seq + (within ? 1u : 0u));
#ifdef NFAI_STAMPS
    STAMP(5);
    _st.t[6] = t_free;
    _st.t[7] = t_mem;
    STAMP_FLUSH(p.stamps, blockIdx.x * ENG_WAVES, 8);
#endif
}

__device__ __forceinline__ void eng_publish(uint64_t *g, uint32_t idx, uint32_t epoch, float v)
{
    __hip_atomic_store((gu64 *)g + idx, ((uint64_t)epoch << 32) | (uint64_t)__builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);  // one global_store_dwordx2 sc1: written through, value and tag together
}

// ---- consumer waves ---------------------------------------------------------------------------------------------------
struct EngCons {  // per-wave state (all wave-uniform)
    uint32_t w, lane, nslot, ring_pieces;
    uint32_t rel = 0;      // slot sequence numbers [0, rel) released
    uint32_t pub = 0;      // last value seen of the loader's landed count
    uint32_t g_prev = 0, rp_prev = 0;  // a piece index whose ring position is known
    uint32_t meets = 0;    // meetings of the CU's consumer waves so far
    uint32_t *err;
    __device__ __forceinline__ void release_below(uint32_t S)
    {
        if (S > rel) {
            rel = S;
            lds_st(W_REL + w * 4, rel);
        }
    }
    __device__ __forceinline__ uint32_t ring_pos(uint32_t g)  // g >= g_prev
    {
        uint32_t rp = rp_prev + (g - g_prev);
        while (rp >= ring_pieces) rp -= ring_pieces;
        g_prev = g;
        rp_prev = rp;
        return rp;
    }
    __device__ __forceinline__ bool wait_landed(uint32_t glast)
    {
        const uint32_t S = glast / ENG_SLOT;
        if ((int32_t)(pub - (S + 1)) >= 0) return true;
        return lds_wait_ge(W_PUB, S + 1, err, 0x40u, pub);
    }
    // all consumer waves of the CU have reached this point (their LDS writes before it are visible after it)
    __device__ __forceinline__ bool meet()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_add(W_ARRIVE, 1);
        uint32_t seen;
        return lds_wait_ge(W_ARRIVE, ENG_NC * ++meets, err, 0x20u, seen);
    }
};

// One group of N chunks (c0 .. c0+N-1) of a unit: the two rows share every activation read.  Reads are batched four chunks
// at a time (16 LDS reads in flight, 64 VGPRs).
template <int N>
__device__ __forceinline__ void eng_group(const uint8_t *ring, uint32_t rp, uint32_t ring_pieces, const float *xs, uint32_t c0, uint32_t lane,
                                          float &a0, float &a1)
{
#pragma unroll
    for (int b = 0; b < N; b += 4) {
        constexpr int BMAX = 4;
        const int nb = N - b < BMAX ? N - b : BMAX;
        u32x4 w0[BMAX], w1[BMAX];
        f32x4 x0[BMAX], x1[BMAX];
#pragma unroll
        for (int j = 0; j < BMAX; j++) {
            if (j < nb) {
                uint32_t r0 = rp + b + j, r1 = rp + N + b + j;
                r0 = r0 >= ring_pieces ? r0 - ring_pieces : r0;
                r1 = r1 >= ring_pieces ? r1 - ring_pieces : r1;
                w0[j] = *reinterpret_cast<const u32x4 *>(ring + r0 * 1024 + lane * 16);
                w1[j] = *reinterpret_cast<const u32x4 *>(ring + r1 * 1024 + lane * 16);
                x0[j] = *reinterpret_cast<const f32x4 *>(xs + ((c0 + b + j) << 9) + (lane << 2));
                x1[j] = *reinterpret_cast<const f32x4 *>(xs + ((c0 + b + j) << 9) + 256 + (lane << 2));
            }
        }
#pragma unroll
        for (int j = 0; j < BMAX; j++) {
            if (j < nb) {
                a0 = dot8_f16(w0[j], x0[j], x1[j], a0);
                a1 = dot8_f16(w1[j], x0[j], x1[j], a1);
            }
        }
    }
}

// Gather this wave's share of an n-granule vector (n % (128 * ENG_NC) == 0): 16-byte sc1 loads (two granules per lane; L1 is
// bypassed, so every pass reads what has reached memory), load index L = j * ENG_NC + w, swept until every tag == epoch.
// MODE 0: raw values -> XR (linear) and the wave's sum of squares -> W_SS[w] (RMSNorm edges)   MODE 1: values -> dst permuted
template <int MODE>
__device__ __forceinline__ bool eng_gather(EngCons &c, const uint64_t *g, uint32_t n, uint32_t epoch, float *dst)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, (int)(n * 8), 0x00020000);
    float ss = 0.f;
    const uint32_t nloads = n / (128 * ENG_NC);  // per lane of this wave
    for (uint32_t l0 = 0; l0 < nloads; l0 += ENG_GL) {
        const uint32_t nl = min((uint32_t)ENG_GL, nloads - l0);
        u32x4 v[ENG_GL];
        for (uint32_t spins = 0;; spins++) {
            bool okc = true;
#pragma unroll
            for (int k = 0; k < ENG_GL; k++) {
                const uint32_t L = (l0 + min((uint32_t)k, nl - 1)) * ENG_NC + c.w;
                v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((L * 64 + c.lane) * 16), 0, 16));
            }
#pragma unroll
            for (int k = 0; k < ENG_GL; k++) okc = okc && v[k][1] == epoch && v[k][3] == epoch;
            if (__all(okc)) break;
            if (lds_ld(W_ABORT) != 0) return false;
            if (spins > (ENG_SPIN_CAP >> 5)) return eng_give_up(c.err, 0x80u);  // a pass is a memory round trip
            __builtin_amdgcn_s_sleep(4);
        }
#pragma unroll
        for (int k = 0; k < ENG_GL; k++) {
            if ((uint32_t)k < nl) {
                const uint32_t e = (((l0 + k) * ENG_NC + c.w) * 64 + c.lane) * 2;  // element index of the first of the two granules
                const f32x4 vf = __builtin_bit_cast(f32x4, v[k]);  // whole-vector cast (on an ELEMENT lvalue hipcc 7.2 reads element 0)
                const float a = vf[0], b = vf[2];
                ss = fmaf(a, a, ss);
                ss = fmaf(b, b, ss);
                const uint32_t i = MODE == 1 ? eng_xs_index(e) : e;  // e is even: the pair stays adjacent under the permutation
                *reinterpret_cast<f32x2 *>(dst + i) = f32x2{a, b};
            }
        }
    }
    if (MODE == 0) {
        ss = wave_sum(ss);
        lds_st(W_SS + c.w * 4, __builtin_bit_cast(uint32_t, ss));
    }
    return true;
}

// After the meeting that follows eng_gather<0>: XA[perm(k)] = (XR[k] / rms) * gamma[k] for this wave's share (RMSNormShader.cs:136-149).
// gm[] = the wave's gains, loaded at kernel start (same (j, lane) -> element mapping as the gather).
template <int NG>
__device__ __forceinline__ void eng_norm(const EngCons &c, const float *XR, const f32x2 (&gm)[NG], float *XA, uint32_t E, float eps)
{
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < ENG_NC; i++) tot += __builtin_bit_cast(float, lds_ld(W_SS + i * 4));  // fixed order: every wave, every CU, the same sum
    const float rms = sqrtf(tot / (float)E + eps);
    const uint32_t nloads = E / (128 * ENG_NC);
#pragma unroll
    for (int j = 0; j < NG; j++) {
        if ((uint32_t)j < nloads) {
            const uint32_t e = ((j * ENG_NC + c.w) * 64 + c.lane) * 2;
            const f32x2 v = *reinterpret_cast<const f32x2 *>(XR + e);
            f32x2 o;
            o[0] = (v[0] / rms) * gm[j][0];
            o[1] = (v[1] / rms) * gm[j][1];
            *reinterpret_cast<f32x2 *>(XA + eng_xs_index(e)) = o;
        }
    }
}

constexpr int ENG_NGAMMA = 8;   // gains per lane and wave: E <= 128 * ENG_NC * ENG_NGAMMA (4096 at four consumer waves)

__device__ __forceinline__ void eng_consumer(const EngineParams &p, uint8_t *lds, uint32_t w, uint32_t lane, uint32_t epoch)
{
    EngCons c;
    c.w = w; c.lane = lane; c.nslot = p.nslot; c.ring_pieces = p.nslot * ENG_SLOT; c.err = p.err;
    const uint8_t *ring = lds + p.ring_off;
    float *XA = reinterpret_cast<float *>(lds + p.xa_off), *XR = reinterpret_cast<float *>(lds + p.xr_off);
    float *XB = reinterpret_cast<float *>(lds + p.xb_off);
    STAMP_DECL;  // consumer: activation of op i in LDS (2i) | this wave's last unit of op i finished (2i + 1)
    // ---- requests first: this wave's share of the attention output and of the block input (plain vectors of the previous
    //      launch), its RMSNorm gains for both edges, the position ---------------------------------------------------------
    constexpr int NV = 4;  // float4 per lane of a plain vector: covers 4 * 64 * ENG_NC * NV elements (4096 at four consumer waves)
    f32x4 av[NV], xv[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const uint32_t k = ((i * ENG_NC + w) * 64 + lane) * 4;
        av[i] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.att + min(k, p.HD - 4));
        xv[i] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.x_in + min(k, p.E - 4));
    }
    f32x2 g1[ENG_NGAMMA], g2[ENG_NGAMMA];
    const uint32_t ngam = p.E / (128 * ENG_NC);
#pragma unroll
    for (int j = 0; j < ENG_NGAMMA; j++) {
        const uint32_t e = ((min((uint32_t)j, ngam - 1) * ENG_NC + w) * 64 + lane) * 2;
        g1[j] = *reinterpret_cast<const GLOBAL_AS f32x2 *>((const GLOBAL_AS float *)p.gamma_ffn + e);
        g2[j] = p.gamma_next ? *reinterpret_cast<const GLOBAL_AS f32x2 *>((const GLOBAL_AS float *)p.gamma_next + e) : f32x2{0.f, 0.f};
    }
    const uint32_t pos = p.pos ? ((const GLOBAL_AS uint32_t *)p.pos)[0] : 0u;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const uint32_t k = ((i * ENG_NC + w) * 64 + lane) * 4;
        if (k < p.HD) *reinterpret_cast<f32x4 *>(XB + eng_xs_index(k)) = av[i];
        if (k < p.E) *reinterpret_cast<f32x4 *>(XR + k) = xv[i];
    }
    bool ok = c.meet();
    uint32_t gbase = 0;   // first piece of the current op
    for (uint32_t oi = 0; oi < p.n_ops && ok; oi++) {
        const EngOp o = eng_load_op(p.ops, oi);
        uint32_t ub, ue;
        eng_unit_range(o.NU, ub, ue);
        const uint32_t KC = o.K >> 9, nu = ue - ub;
        const float *xs = o.x_sel ? XB : XA;
#ifdef NFAI_STAMPS
        if (oi == 0) STAMP(0); else if (oi == 1) STAMP(2); else if (oi == 2) STAMP(4); else STAMP(6);
#endif
        for (uint32_t r0 = 0; r0 + w < nu && ok; r0 += ENG_NC) {
            const uint32_t nur = min((uint32_t)ENG_NC, nu - r0), rbase = gbase + r0 * 2 * KC;
            const uint32_t u = ub + r0 + w;
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
            float a0 = 0.f, a1 = 0.f;
            for (uint32_t c0 = 0; c0 < KC && ok; c0 += 8) {
                const uint32_t n = min(8u, KC - c0);
                const uint32_t gfirst = rbase + c0 * 2 * nur + w * 2 * n;
                c.release_below(gfirst / ENG_SLOT);
                if (!c.wait_landed(gfirst + 2 * n - 1)) { ok = false; break; }
                const uint32_t rp = c.ring_pos(gfirst);
                switch (n) {
                    case 8: eng_group<8>(ring, rp, c.ring_pieces, xs, c0, lane, a0, a1); break;
                    case 7: eng_group<7>(ring, rp, c.ring_pieces, xs, c0, lane, a0, a1); break;
                    case 6: eng_group<6>(ring, rp, c.ring_pieces, xs, c0, lane, a0, a1); break;
                    case 5: eng_group<5>(ring, rp, c.ring_pieces, xs, c0, lane, a0, a1); break;
                    case 4: eng_group<4>(ring, rp, c.ring_pieces, xs, c0, lane, a0, a1); break;
                    case 3: eng_group<3>(ring, rp, c.ring_pieces, xs, c0, lane, a0, a1); break;
                    case 2: eng_group<2>(ring, rp, c.ring_pieces, xs, c0, lane, a0, a1); break;
                    default: eng_group<1>(ring, rp, c.ring_pieces, xs, c0, lane, a0, a1); break;
                }
            }
            if (!ok) break;
            a0 = wave_sum(a0);
            a1 = wave_sum(a1);
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
        if (!ok) break;
        gbase += nu * 2 * KC;
#ifdef NFAI_STAMPS
        if (oi == 0) STAMP(1); else if (oi == 1) STAMP(3); else if (oi == 2) STAMP(5); else STAMP(7);
#endif
        // nothing before this wave's first piece of the next op will be read by it again: the loader may run on while
        // this wave gathers
        if (oi + 1 < p.n_ops) {
            const EngOp nx = eng_load_op(p.ops, oi + 1);
            uint32_t nb, ne;
            eng_unit_range(nx.NU, nb, ne);
            const uint32_t nKC = nx.K >> 9, nnu = ne - nb;
            c.release_below((gbase + (w < nnu ? w * 2 * min(8u, nKC) : nnu * 2 * nKC)) / ENG_SLOT);
        } else {
            c.release_below((gbase + ENG_SLOT - 1) / ENG_SLOT);
            break;
        }
        // ---- edge: the next op's activation vector from every CU ------------------------------------------------------------
        if (oi == 0) {         // h = x + Wo.att -> XR (raw: the residual of Wdown), XA = RMSNorm(h) * ffn_norm
            ok = eng_gather<0>(c, p.g_h, p.E, epoch, XR) && c.meet();
            if (ok) { eng_norm<ENG_NGAMMA>(c, XR, g1, XA, p.E, p.eps); ok = c.meet(); }
        } else if (oi == 1) {  // act = up * silu(gate) -> XB
            ok = eng_gather<1>(c, p.g_act, p.F, epoch, XB) && c.meet();
        } else {               // x' = h + Wdown.act -> XA = RMSNorm(x') * attn_norm of the next block
            ok = eng_gather<0>(c, p.g_x, p.E, epoch, XR) && c.meet();
            if (ok) { eng_norm<ENG_NGAMMA>(c, XR, g2, XA, p.E, p.eps); ok = c.meet(); }
        }
    }
    STAMP_FLUSH(p.stamps, blockIdx.x * ENG_WAVES + 1 + w, 8);
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
    else eng_consumer(p, lds, wave - 1, lane, epoch);
}

// ---- host side ----------------------------------------------------------------------------------------------------------
size_t engine_params_bytes() { return (sizeof(EngOp) * ENG_MAX_OPS + 255) & ~(size_t)255; }

// Fill the launch's parameter block and copy it to `params_dev` (synchronous: call it outside stream capture, once per block).
hipError_t engine_plan(const EngineArgs &a, void *params_dev, EnginePlan &plan)
{
    EngineParams p{};
    EngOp ops[ENG_MAX_OPS] = {};
    if (a.n_ops < 3 || a.n_ops > 4 || !params_dev) return hipErrorInvalidValue;
    const uint32_t E = a.E, F = a.F, HD = a.HD;
    if (E % 512 || F % 512 || HD % 512 || E % (128 * ENG_NC) || F % (128 * ENG_NC)) return hipErrorInvalidValue;
    if (E > 128 * ENG_NC * ENG_NGAMMA || E > 4 * 64 * ENG_NC * 4 || HD > 4 * 64 * ENG_NC * 4) return hipErrorInvalidValue;  // per-lane register shares
    p.n_ops = a.n_ops;
    p.E = E; p.F = F; p.HD = HD;
    // op 0: Wo + residual
    ops[0].W[0] = static_cast<const uint8_t *>(a.Wo); ops[0].K = HD; ops[0].NU = E / 2; ops[0].mode = ENG_RESIDUAL;
    ops[0].x_sel = 1; ops[0].g_out = a.g_h; ops[0].y_plain = nullptr;
    // op 1: gate | up
    ops[1].W[0] = static_cast<const uint8_t *>(a.Wgate); ops[1].W[1] = static_cast<const uint8_t *>(a.Wup);
    ops[1].K = E; ops[1].NU = F; ops[1].mode = ENG_GATEUP; ops[1].x_sel = 0; ops[1].g_out = a.g_act;
    // op 2: Wdown + residual
    ops[2].W[0] = static_cast<const uint8_t *>(a.Wdown); ops[2].K = F; ops[2].NU = E / 2; ops[2].mode = ENG_RESIDUAL;
    ops[2].x_sel = 1; ops[2].g_out = a.n_ops == 4 ? a.g_x : nullptr; ops[2].y_plain = a.x_out;
    if (a.n_ops == 4) {
        const uint32_t rows = a.qkv_rows[0] + a.qkv_rows[1] + a.qkv_rows[2];
        if ((a.qkv_rows[0] | a.qkv_rows[1] | a.qkv_rows[2] | a.D) & 1u) return hipErrorInvalidValue;
        for (int i = 0; i < 3; i++) ops[3].W[i] = static_cast<const uint8_t *>(a.Wqkv[i]);
        ops[3].seg_end[0] = a.qkv_rows[0];
        ops[3].seg_end[1] = a.qkv_rows[0] + a.qkv_rows[1];
        ops[3].seg_end[2] = rows;
        ops[3].K = E; ops[3].NU = rows / 2; ops[3].mode = ENG_QKV; ops[3].x_sel = 0;
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
    plan.lds_bytes = p.ring_off + nslot * ENG_SLOT * 1024;
    // slots of loads kept in flight (x 8 KiB): the ring keeps >= 3 slots beyond them (being read / waiting / being released)
    int ahead = nslot >= 8 ? 4 : 2;  // measured: 2, 4 and 6 slots in flight stream at the same rate; 4 leaves the ring more room
    static const int env_ahead = getenv("NFAI_ENGINE_AHEAD") ? atoi(getenv("NFAI_ENGINE_AHEAD")) : 0;
    if ((env_ahead == 2 || env_ahead == 4 || env_ahead == 6) && (uint32_t)env_ahead + 3 <= nslot) ahead = env_ahead;
    plan.ahead = ahead;
    plan.n_cu = a.n_cu;
    plan.params_dev = params_dev;
    p.ops = static_cast<const EngOp *>(params_dev);
    NFAI_STAMP_SET(p, "engine", a.n_cu, ENG_THREADS);
    static_assert(sizeof(EngineParams) <= sizeof(plan.params), "EnginePlan::params holds the kernel argument");
    memcpy(plan.params, &p, sizeof(p));
    return hipMemcpy(params_dev, ops, sizeof(ops), hipMemcpyHostToDevice);
}

hipError_t launch_engine(const EnginePlan &plan, hipStream_t s)
{
    if (!plan.params_dev) return hipErrorInvalidValue;
    auto launch = [&](auto kern) -> hipError_t {
        static bool attr_set = false;  // one per instantiation of the lambda body = per kernel
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        EngineParams p;
        memcpy(&p, plan.params, sizeof(p));
        hipLaunchKernelGGL(kern, dim3(plan.n_cu), dim3(ENG_THREADS), plan.lds_bytes, s, p);
        return hipGetLastError();
    };
    if (plan.ahead == 6) return launch(k_engine<6>);
    if (plan.ahead == 4) return launch(k_engine<4>);
    return launch(k_engine<2>);
}

}  // namespace nfai
