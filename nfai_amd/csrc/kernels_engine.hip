// kernels_engine.hip — the weight-streaming engine: Wo + residual -> RMSNorm + Wgate|Wup + SiLU*up -> Wdown + residual ->
// RMSNorm + (next block's) Wq|Wk|Wv + RoPE + KV write of one TransformerBlock (TransformerBlock.cs:150-181, then :129-141 of
// the next block) as ONE launch, one workgroup per CU, instead of four launches.
//
// Why: the per-launch stamps (profiles/round2_stamps_f16_five_launch.json) show ~1.5 us of idle HBM at every kernel boundary,
// a 1-3 us head in which a launch only requests bytes and a 1-2.6 us tail in which half of the CUs have already finished.
// Inside one launch the weight stream never stops:
//   * STREAM waves (8 per CU) pull their rows HBM -> VGPR with 16-byte non-temporal loads, two steps of twelve 1-KiB loads in
//     flight each (the GEMV kernels' proven form: 6.6-7 TB/s; the weights never touch LDS).  A wave's steps form ONE flat
//     sequence over all projections of the launch, so the loads of the next projection are already in flight while the wave
//     waits for that projection's input: ~190 KiB per CU cross every dependency.
//   * The activation vector of the next projection travels between the CUs as 8-byte {value, tag} granules written through to
//     memory (one sc1 store each); CONTROL waves (2 per CU, no other memory traffic, so nothing queues in front of their loads)
//     sweep their share of the vector with sc1 loads until every tag carries this token's epoch — the data is the flag: no
//     grid barrier, no counter, nothing to reset — normalise it (RMSNorm) and put it into LDS, where the stream waves find it
//     behind an LDS word (MI355X guide: allgather, engine-vs-launches; Guideline 16, R2).
// An earlier form of this file fed the consumers through a ring of LDS slots filled by LDS-DMA (one loader wave per CU): alone
// that loader streams 6.4 TB/s (tools/ldsdma_bench.hip), but beside consumer waves reading fp32 activations and the ring from
// LDS its instructions issued at 64 ns per KiB instead of 40 (profiles/round2_stamps_engine_ldsdma_ring.json).
//
//   unit = two rows that finish together (rows 2u, 2u+1; gate row u + up row u; a RoPE pair), owned by ONE stream wave: a row
//          is reduced inside a wave (DPP), the epilogue needs nobody else
//   op   = one projection; the CU owns a contiguous range of its units, unit i of the range goes to stream wave i % 8
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

constexpr int ENG_NS = 8;                   // stream waves
constexpr int ENG_NCW = 4;                  // control waves
constexpr int ENG_WAVES = ENG_NS + ENG_NCW;
constexpr int ENG_THREADS = ENG_WAVES * 64;
constexpr int ENG_MAX_OPS = 4;
constexpr int ENG_GL = 8;                   // 16-byte loads per lane and gather chunk
constexpr int ENG_NGAMMA = 8;               // RMSNorm gains per lane of a control wave: E <= 128 * ENG_NCW * ENG_NGAMMA = 4096
constexpr int ENG_NV = 4;                   // float4 per lane of a control wave for a plain vector: <= 4 * 64 * ENG_NCW * ENG_NV = 4096
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
                         // argument would make hipcc copy the whole block to scratch); each wave copies the op it works on into
                         // registers once
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
    uint32_t xa_off, xr_off, xb_off, cs_off, part_off, cnt_off;  // byte offsets in LDS
    NFAI_STAMP_PARAM
};

// ---- LDS control words, accessed with inline asm (the polls must not be merged, hoisted or waited for by the compiler's
//      bookkeeping).  Every lane reads the same word, so the value (and every branch on it) is made wave-uniform. ----------
__device__ __forceinline__ uint32_t lds_ld(uint32_t addr)
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void lds_st(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_add(uint32_t addr, uint32_t v) { asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

// byte offsets of the control words (the first KiB of LDS)
constexpr uint32_t W_ABORT = 0;     // non-zero: a wait gave up, everybody leaves
constexpr uint32_t W_XREADY = 4;    // ops whose activation vector is in LDS (monotonic)
constexpr uint32_t W_DONE = 8;      // stream-wave completions, one per stream wave and op (monotonic)
constexpr uint32_t W_ARRIVE = 12;   // the control waves' meeting counter (monotonic)
constexpr uint32_t W_CTL_ISSUED = 16;  // control waves that have REQUESTED their first inputs (the stream waves' weight loads queue behind)
constexpr uint32_t W_SS = 32;       // per control wave: its share of the sum of squares of the vector being gathered

__device__ __forceinline__ bool eng_give_up(uint32_t *err, uint32_t code)
{
    lds_st(W_ABORT, 1);
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// bounded wait until the LDS word at `addr` is >= target; false on abort / timeout
__device__ __forceinline__ bool lds_wait_ge(uint32_t addr, uint32_t target, uint32_t *err, uint32_t code)
{
    for (uint32_t spins = 0;; spins++) {
        if ((int32_t)(lds_ld(addr) - target) >= 0) return true;
        if (lds_ld(W_ABORT) != 0) return false;
        if (spins > ENG_SPIN_CAP) return eng_give_up(err, code);
        __builtin_amdgcn_s_sleep(1);
    }
}

// Op i of the launch as a register copy.  The table was written before the launch and nobody writes it: read through the
// CONSTANT address space, so that hipcc uses scalar loads (through a plain global pointer it emits a vector load and waits
// vmcnt(0) for it: a memory round trip on every op boundary and a drain of the weight loads in flight).
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
    if (o.mode == ENG_GATEUP) return (const GLOBAL_AS uint8_t *)(sub ? o.W[1] : o.W[0]) + (uint64_t)unit * rb;
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

__device__ __forceinline__ void eng_publish(uint64_t *g, uint32_t idx, uint32_t epoch, float v)
{
    __hip_atomic_store((gu64 *)g + idx, ((uint64_t)epoch << 32) | (uint64_t)__builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);  // one global_store_dwordx2 sc1: written through, value and tag together
}

// ---- stream waves -------------------------------------------------------------------------------------------------------
// Work item = one STEP = U consecutive 1-KiB chunks of ONE row (U = K/512 of the projections from the embedding width: a whole
// row; the long rows of Wdown are cut into ceil(KC / U) segments, the last one padded with re-reads of its last chunk, which
// hit in cache), so the unrolled load / FMA code holds no per-piece bookkeeping.  The CU's items of an op — unit-major, then
// row, then segment — go to its eight stream waves round-robin, whatever the op's shape (Wdown of Llama-3.2-3B: 12 rows x 3
// segments = 36 items on a CU).  Every item ends in a partial sum in LDS and a count on the item's UNIT; the wave whose count
// completes the unit adds the partial sums in fixed order (bit-reproducible) and runs the epilogue.
//   item i of the op:  unit ul = i / (2 * nseg), row sub = (i / nseg) % 2, segment seg = i % nseg
// One cursor issues loads two steps ahead of the one that consumes; both are wave-uniform.
constexpr uint32_t ENG_MAX_UNITS = 64, ENG_MAX_SEG = 4;  // per CU and op (gate|up of Llama-3.1-8B: 56 units); Wdown row segments

struct EngCursor {
    uint32_t oi = 0, i = 0, nitems = 0, nseg = 1, KC = 1, ub = 0;
    uint32_t ul = 0, sub = 0, seg = 0;  // of item i
    bool end = false;
    EngOp o;
};

template <int U> __device__ __forceinline__ void eng_cursor_item(EngCursor &cu)
{
    const uint32_t q = cu.i / cu.nseg;
    cu.seg = cu.i - q * cu.nseg;
    cu.ul = q >> 1;
    cu.sub = q & 1;
}

// Enter op `oi` (or the first later op in which this wave owns an item).  `on_skip(op)` is called for every op the wave passes
// without work (the consuming cursor reports it done).
template <int U, typename F>
__device__ __forceinline__ void eng_cursor_enter(EngCursor &cu, const EngineParams &p, uint32_t oi, uint32_t s, F on_skip)
{
    for (;; oi++) {
        if (oi >= p.n_ops) { cu.end = true; cu.oi = oi; return; }
        cu.o = eng_load_op(p.ops, oi);
        uint32_t ue;
        eng_unit_range(cu.o.NU, cu.ub, ue);
        cu.KC = cu.o.K >> 9;
        cu.nseg = (cu.KC + U - 1) / U;
        cu.nitems = (ue - cu.ub) * 2 * cu.nseg;
        if (s < cu.nitems) break;
        on_skip(oi);
    }
    cu.oi = oi; cu.i = s;
    eng_cursor_item<U>(cu);
}

// ONE add per wave (lane 0), the value before it for every lane
__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t addr, uint32_t v, uint32_t lane)
{
    uint32_t r = 0;
    if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(v) : "memory");
    return __builtin_amdgcn_readfirstlane(r);
}

template <int U>
__device__ __forceinline__ void eng_stream(const EngineParams &p, uint8_t *lds, uint32_t s, uint32_t lane, uint32_t epoch)
{
    const float *XA = reinterpret_cast<const float *>(lds + p.xa_off), *XR = reinterpret_cast<const float *>(lds + p.xr_off);
    const float *XB = reinterpret_cast<const float *>(lds + p.xb_off), *CS = reinterpret_cast<const float *>(lds + p.cs_off);
    float *PART = reinterpret_cast<float *>(lds + p.part_off);   // [op][unit][row][segment] partial sums
    const uint32_t cnt_base = p.cnt_off;                          // [op][unit] arrival counts (zero at launch)
    const uint32_t pos = p.pos ? ((const GLOBAL_AS uint32_t *)p.pos)[0] : 0u;
    STAMP_DECL;  // stream wave: activation of op i seen (2i) | this wave's last item of op i finished (2i + 1)
    EngCursor ic, cc;  // issuing / consuming
    auto skip_quiet = [](uint32_t) {};
    auto skip_done = [&](uint32_t) { if (lane == 0) lds_add(W_DONE, 1); };
    eng_cursor_enter<U>(ic, p, 0, s, skip_quiet);
    eng_cursor_enter<U>(cc, p, 0, s, skip_done);
    if (cc.end) return;  // no item in any op (tiny models): the passes above reported every op done
    const GLOBAL_AS uint8_t *last_addr = eng_row(ic.o, ic.ub, 0) + lane * 16;  // a valid address for the surplus loads behind the end
    // A CU's memory requests are served in order: the control waves' first requests (the vectors op 0 waits for) go ahead of
    // the ~140 KiB of weights this CU's stream waves are about to request (stamps: 7 us -> first FMA otherwise).
    if (!lds_wait_ge(W_CTL_ISSUED, ENG_NCW, p.err, 0x800u)) return;

    // unconditional loads (behind the end of a row / of the sequence: re-reads of the last valid chunk): every
    // compiler-inserted vmcnt is an exact count
    auto issue = [&](u32x4 (&buf)[U]) {
        const uint32_t c0 = ic.seg * U;
        const uint32_t nvalid = ic.end ? 1u : min((uint32_t)U, ic.KC - c0);
        const GLOBAL_AS uint8_t *a = ic.end ? last_addr : eng_row(ic.o, ic.ub + ic.ul, ic.sub) + lane * 16 + (uint64_t)c0 * 1024;
#pragma unroll
        for (int j = 0; j < U; j++) buf[j] = load_nt16((const void *)(a + (uint64_t)min((uint32_t)j, nvalid - 1) * 1024));
        last_addr = a;
        if (!ic.end) {
            ic.i += ENG_NS;
            if (ic.i < ic.nitems) eng_cursor_item<U>(ic);
            else eng_cursor_enter<U>(ic, p, ic.oi + 1, s, skip_quiet);
        }
    };

    bool ok = true;
    bool fresh_op = true;   // the consuming cursor has not yet checked that its op's activation vector is in LDS
    const float *xs = XA;
    // the unit's epilogue, by the wave whose item completed it; r0, r1 = the two row sums
    auto finish_unit = [&](float r0, float r1, uint32_t ul) {
        const EngOp &o = cc.o;
        const uint32_t u = cc.ub + ul;
        if (lane == 0) {
            if (o.mode == ENG_RESIDUAL) {
                // host residual add of TransformerBlock.cs:153-158 / 176-180: input + projection
                const float y0 = XR[2 * u] + r0, y1 = XR[2 * u + 1] + r1;
                if (o.g_out) { eng_publish(o.g_out, 2 * u, epoch, y0); eng_publish(o.g_out, 2 * u + 1, epoch, y1); }
                if (o.y_plain) { o.y_plain[2 * u] = y0; o.y_plain[2 * u + 1] = y1; }
            } else if (o.mode == ENG_GATEUP) {
                const float v = r1 * silu_ref(r0);  // SiLUShader.cs:121-123, ElementWiseMultiplicationShader.cs:137
                eng_publish(o.g_out, u, epoch, v);
            } else {
                // RoPEShader.cs:249-262 on the pair (row, row+1); V rows are stored unrotated
                const uint32_t row = u * 2;
                const uint32_t seg = row < o.seg_end[0] ? 0u : (row < o.seg_end[1] ? 1u : 2u);
                const uint32_t r = seg == 0 ? row : (seg == 1 ? row - o.seg_end[0] : row - o.seg_end[1]);
                const uint32_t head = r / p.D, d = r % p.D;
                float o0 = r0, o1 = r1;
                if (seg < 2 && d < p.rope_dims) {
                    const float c0 = CS[d], s0 = CS[d + 1];  // [pair][2]: cos, sin of the current position (k_token_begin's table)
                    o0 = c0 * r0 - s0 * r1;
                    o1 = s0 * r0 + c0 * r1;
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
    };
    auto consume = [&](u32x4 (&buf)[U]) {
        if (cc.end || !ok) return;  // surplus loads behind the end of the sequence
        if (fresh_op) {
            // the first item of an op: its activation vector must be in LDS (the control waves flag it)
            ok = lds_wait_ge(W_XREADY, cc.oi + 1, p.err, 0x20u);
            if (!ok) return;
            fresh_op = false;
            xs = cc.o.x_sel ? XB : XA;
#ifdef NFAI_STAMPS
            if (cc.oi == 0) STAMP(0); else if (cc.oi == 1) STAMP(2); else if (cc.oi == 2) STAMP(4); else STAMP(6);
#endif
        }
        const uint32_t c0 = cc.seg * U;
        const uint32_t nvalid = min((uint32_t)U, cc.KC - c0);
        const float *xc = xs + (c0 << 9) + (lane << 2);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < U; j++) {
            if ((uint32_t)j < nvalid) {  // wave-uniform; nothing but LDS reads and FMAs inside
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xc + (j << 9));
                const f32x4 x1 = *reinterpret_cast<const f32x4 *>(xc + (j << 9) + 256);
                acc = dot8_f16(buf[j], x0, x1, acc);
            }
        }
        acc = wave_sum(acc);
        // partial sum of (unit, row, segment) -> LDS; count the item on its unit
        const uint32_t slot = ((cc.oi * ENG_MAX_UNITS + cc.ul) * 2 + cc.sub) * ENG_MAX_SEG + cc.seg;
        if (lane == 0) PART[slot] = acc;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t arrived = lds_add_rtn(cnt_base + (cc.oi * ENG_MAX_UNITS + cc.ul) * 4, 1, lane);
        if (arrived + 1 == 2 * cc.nseg) {  // this item completed its unit: every partial sum of the unit is in LDS
            const float *pu = PART + (cc.oi * ENG_MAX_UNITS + cc.ul) * 2 * ENG_MAX_SEG;
            float r0 = 0.f, r1 = 0.f;
            for (uint32_t g = 0; g < cc.nseg; g++) { r0 += pu[g]; r1 += pu[ENG_MAX_SEG + g]; }  // fixed order
            finish_unit(r0, r1, cc.ul);
        }
        cc.i += ENG_NS;
        if (cc.i < cc.nitems) {
            eng_cursor_item<U>(cc);
        } else {  // that was this wave's last item of the op
#ifdef NFAI_STAMPS
            if (cc.oi == 0) STAMP(1); else if (cc.oi == 1) STAMP(3); else if (cc.oi == 2) STAMP(5); else STAMP(7);
#endif
            if (lane == 0) lds_add(W_DONE, 1);
            fresh_op = true;
            eng_cursor_enter<U>(cc, p, cc.oi + 1, s, skip_done);
        }
    };

    // three register sets in rotation, two steps of loads in flight behind the one being multiplied (k_gemv's ping-pong, one deeper)
    u32x4 bufA[U], bufB[U], bufC[U];
    issue(bufA);
    issue(bufB);
    issue(bufC);
    while (!cc.end && ok) {
        consume(bufA);
        issue(bufA);
        consume(bufB);
        issue(bufB);
        consume(bufC);
        issue(bufC);
    }
    STAMP_FLUSH(p.stamps, blockIdx.x * ENG_WAVES + s, 8);
}

// ---- control waves ------------------------------------------------------------------------------------------------------
struct EngCtl {
    uint32_t cw, lane, meets = 0;
    uint32_t *err;
    // both control waves have reached this point (their LDS writes before it are visible after it)
    __device__ __forceinline__ bool meet()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_add(W_ARRIVE, 1);
        return lds_wait_ge(W_ARRIVE, ENG_NCW * ++meets, err, 0x200u);
    }
};

// Gather this wave's share of an n-granule vector (n % (128 * ENG_NCW) == 0): 16-byte sc1 loads (two granules per lane; L1 is
// bypassed, so every pass reads what has reached memory), load index L = j * ENG_NCW + cw, swept until every tag == epoch.
// MODE 0: raw values -> XR (linear) and the wave's sum of squares -> W_SS[cw] (RMSNorm edges)   MODE 1: values -> dst permuted
template <int MODE>
__device__ __forceinline__ bool eng_gather(EngCtl &c, const uint64_t *g, uint32_t n, uint32_t epoch, float *dst)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, (int)(n * 8), 0x00020000);
    float ss = 0.f;
    const uint32_t nloads = n / (128 * ENG_NCW);  // per lane of this wave
    for (uint32_t l0 = 0; l0 < nloads; l0 += ENG_GL) {
        const uint32_t nl = min((uint32_t)ENG_GL, nloads - l0);
        u32x4 v[ENG_GL];
        for (uint32_t spins = 0;; spins++) {
            bool okc = true;
#pragma unroll
            for (int k = 0; k < ENG_GL; k++) {
                const uint32_t L = (l0 + min((uint32_t)k, nl - 1)) * ENG_NCW + c.cw;
                v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((L * 64 + c.lane) * 16), 0, 16));
            }
#pragma unroll
            for (int k = 0; k < ENG_GL; k++) okc = okc && v[k][1] == epoch && v[k][3] == epoch;
            if (__all(okc)) break;
            if (lds_ld(W_ABORT) != 0) return false;
            if (spins > (ENG_SPIN_CAP >> 5)) return eng_give_up(c.err, 0x80u);  // a pass is a memory round trip
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int k = 0; k < ENG_GL; k++) {
            if ((uint32_t)k < nl) {
                const uint32_t e = (((l0 + k) * ENG_NCW + c.cw) * 64 + c.lane) * 2;  // element index of the first of the two granules
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
        lds_st(W_SS + c.cw * 4, __builtin_bit_cast(uint32_t, ss));
    }
    return true;
}

// After the meeting that follows eng_gather<0>: XA[perm(k)] = (XR[k] / rms) * gamma[k] for this wave's share (RMSNormShader.cs:136-149).
// gm[] = the wave's gains, loaded at kernel start (same (j, lane) -> element mapping as the gather).
__device__ __forceinline__ void eng_norm(const EngCtl &c, const float *XR, const f32x2 (&gm)[ENG_NGAMMA], float *XA, uint32_t E, float eps)
{
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < ENG_NCW; i++) tot += __builtin_bit_cast(float, lds_ld(W_SS + i * 4));  // fixed order: every wave, every CU, the same sum
    const float rms = sqrtf(tot / (float)E + eps);
    const uint32_t nloads = E / (128 * ENG_NCW);
#pragma unroll
    for (int j = 0; j < ENG_NGAMMA; j++) {
        if ((uint32_t)j < nloads) {
            const uint32_t e = ((j * ENG_NCW + c.cw) * 64 + c.lane) * 2;
            const f32x2 v = *reinterpret_cast<const f32x2 *>(XR + e);
            f32x2 o;
            o[0] = (v[0] / rms) * gm[j][0];
            o[1] = (v[1] / rms) * gm[j][1];
            *reinterpret_cast<f32x2 *>(XA + eng_xs_index(e)) = o;
        }
    }
}

__device__ __forceinline__ void eng_control(const EngineParams &p, uint8_t *lds, uint32_t cw, uint32_t lane, uint32_t epoch)
{
    EngCtl c;
    c.cw = cw; c.lane = lane; c.err = p.err;
    float *XA = reinterpret_cast<float *>(lds + p.xa_off), *XR = reinterpret_cast<float *>(lds + p.xr_off);
    float *XB = reinterpret_cast<float *>(lds + p.xb_off), *CS = reinterpret_cast<float *>(lds + p.cs_off);
    STAMP_DECL;  // control wave: start | x of op 0 in LDS | stream waves done op 0 | h gathered + normalised | done op 1 | act gathered | done op 2 | x' gathered + normalised
    STAMP(0);
    // ---- requests first: this wave's share of the attention output and of the block input (plain vectors of the previous
    //      launch), its RMSNorm gains for both edges, the cos/sin table ---------------------------------------------------
    f32x4 av[ENG_NV], xv[ENG_NV];
#pragma unroll
    for (int i = 0; i < ENG_NV; i++) {
        const uint32_t k = ((i * ENG_NCW + cw) * 64 + lane) * 4;
        av[i] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.att + min(k, p.HD - 4));
        xv[i] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)p.x_in + min(k, p.E - 4));
    }
    f32x2 g1[ENG_NGAMMA], g2[ENG_NGAMMA];
    const uint32_t ngam = p.E / (128 * ENG_NCW);
#pragma unroll
    for (int j = 0; j < ENG_NGAMMA; j++) {
        const uint32_t e = ((min((uint32_t)j, ngam - 1) * ENG_NCW + cw) * 64 + lane) * 2;
        g1[j] = *reinterpret_cast<const GLOBAL_AS f32x2 *>((const GLOBAL_AS float *)p.gamma_ffn + e);
        g2[j] = p.gamma_next ? *reinterpret_cast<const GLOBAL_AS f32x2 *>((const GLOBAL_AS float *)p.gamma_next + e) : f32x2{0.f, 0.f};
    }
    f32x2 csv = f32x2{1.f, 0.f};
    if (p.rope_cs && cw == 0 && lane * 2 < p.D) csv = *reinterpret_cast<const GLOBAL_AS f32x2 *>((const GLOBAL_AS float *)p.rope_cs + lane * 2);
    __builtin_amdgcn_sched_barrier(0);  // every request above is issued before the stream waves are let go
    if (lane == 0) lds_add(W_CTL_ISSUED, 1);
#pragma unroll
    for (int i = 0; i < ENG_NV; i++) {
        const uint32_t k = ((i * ENG_NCW + cw) * 64 + lane) * 4;
        if (k < p.HD) *reinterpret_cast<f32x4 *>(XB + eng_xs_index(k)) = av[i];
        if (k < p.E) *reinterpret_cast<f32x4 *>(XR + k) = xv[i];
    }
    if (cw == 0 && lane * 2 < p.D) *reinterpret_cast<f32x2 *>(CS + lane * 2) = csv;
    if (!c.meet()) return;
    if (cw == 0) lds_st(W_XREADY, 1);
    STAMP(1);
    for (uint32_t k = 0; k + 1 < p.n_ops; k++) {
        // edge k -> k + 1: every stream wave of this CU has sent its outputs of op k; then the whole vector, from every CU
        if (!lds_wait_ge(W_DONE, ENG_NS * (k + 1), p.err, 0x100u)) return;
#ifdef NFAI_STAMPS
        if (k == 0) STAMP(2); else if (k == 1) STAMP(4); else STAMP(6);
#endif
        bool ok;
        if (k == 0) {         // h = x + Wo.att -> XR (raw: the residual of Wdown), XA = RMSNorm(h) * ffn_norm
            ok = eng_gather<0>(c, p.g_h, p.E, epoch, XR) && c.meet();
            if (ok) { eng_norm(c, XR, g1, XA, p.E, p.eps); ok = c.meet(); }
        } else if (k == 1) {  // act = up * silu(gate) -> XB
            ok = eng_gather<1>(c, p.g_act, p.F, epoch, XB) && c.meet();
        } else {              // x' = h + Wdown.act -> XA = RMSNorm(x') * attn_norm of the next block
            ok = eng_gather<0>(c, p.g_x, p.E, epoch, XR) && c.meet();
            if (ok) { eng_norm(c, XR, g2, XA, p.E, p.eps); ok = c.meet(); }
        }
        if (!ok) return;
        if (cw == 0) lds_st(W_XREADY, k + 2);
#ifdef NFAI_STAMPS
        if (k == 0) STAMP(3); else if (k == 1) STAMP(5); else STAMP(7);
#endif
    }
    STAMP_FLUSH(p.stamps, blockIdx.x * ENG_WAVES + ENG_NS + cw, 8);
}

template <int U>
__global__ __launch_bounds__(ENG_THREADS) void k_engine(const EngineParams p)
{
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // control words start at zero: wave 0 clears them, everyone meets once (the only workgroup barrier of the kernel)
    if (wave == 0) reinterpret_cast<uint32_t *>(lds)[lane] = 0;
    for (uint32_t i = threadIdx.x; i < ENG_MAX_OPS * ENG_MAX_UNITS; i += ENG_THREADS) reinterpret_cast<uint32_t *>(lds + p.cnt_off)[i] = 0;
    __syncthreads();
    const uint32_t epoch = ((const GLOBAL_AS uint32_t *)p.epoch)[0];
    if (wave < ENG_NS) eng_stream<U>(p, lds, wave, lane, epoch);
    else eng_control(p, lds, wave - ENG_NS, lane, epoch);
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
    if (E % 512 || F % 512 || HD % 512 || E % (128 * ENG_NCW) || F % (128 * ENG_NCW)) return hipErrorInvalidValue;
    if (E > 128 * ENG_NCW * ENG_NGAMMA || E > 4 * 64 * ENG_NCW * ENG_NV || HD > 4 * 64 * ENG_NCW * ENG_NV || a.D > 128) return hipErrorInvalidValue;  // per-lane register shares
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
    // LDS: control words | XA (E) | XR (E) | XB (max(HD, F)) | cos/sin table of the position
    const uint32_t xb = (HD > F ? HD : F);
    p.xa_off = 1024;
    p.xr_off = p.xa_off + E * 4;
    p.xb_off = p.xr_off + E * 4;
    p.cs_off = p.xb_off + xb * 4;
    p.part_off = p.cs_off + 1024;                                                   // [op][unit][row][segment] floats
    p.cnt_off = p.part_off + ENG_MAX_OPS * ENG_MAX_UNITS * 2 * ENG_MAX_SEG * 4;     // [op][unit] arrival counts
    plan.lds_bytes = p.cnt_off + ENG_MAX_OPS * ENG_MAX_UNITS * 4;
    // per CU and op: at most ENG_MAX_UNITS units, rows of at most ENG_MAX_SEG steps
    {
        const uint32_t U = E / 512;
        for (uint32_t i = 0; i < a.n_ops; i++)
            if ((ops[i].NU + a.n_cu - 1) / a.n_cu > ENG_MAX_UNITS || ((ops[i].K >> 9) + U - 1) / U > ENG_MAX_SEG) return hipErrorInvalidValue;
    }
    if (plan.lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    // step = U chunks of one row: U = 512-weight chunks of a row of the projections from the embedding width
    if (E != HD || (E / 512 != 1 && E / 512 != 2 && E / 512 != 4 && E / 512 != 6 && E / 512 != 8)) return hipErrorInvalidValue;
    plan.ahead = (int)(E / 512);
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
    EngineParams p;
    memcpy(&p, plan.params, sizeof(p));
    auto launch = [&](auto kern) -> hipError_t {
        static bool attr_set = false;  // one per kernel (the lambda body is instantiated per kernel)
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3(plan.n_cu), dim3(ENG_THREADS), plan.lds_bytes, s, p);
        return hipGetLastError();
    };
    if (plan.ahead == 1) return launch(k_engine<1>);
    if (plan.ahead == 2) return launch(k_engine<2>);
    if (plan.ahead == 4) return launch(k_engine<4>);
    if (plan.ahead == 6) return launch(k_engine<6>);
    return launch(k_engine<8>);
}

}  // namespace nfai
