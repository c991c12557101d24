// Micro-benchmark (round 4): the MLP half of a decode block — gate|up GEMV + SiLU*up, then Wdown GEMV + residual — as TWO launches
// (the product's structure: a kernel boundary between them) against ONE launch in which every wave requests its whole share of Wdown
// right after its last gate|up request (Wdown's share of a CU, 196 KB at 3B, fits the registers of its eight waves: 24 x 16 bytes per
// lane), publishes its activations as 8-byte {value, tag} granules and polls the K-slice it needs while those weights arrive.
// Hypothesis: the weight stream never stops across the dependency, so the launch costs (bytes / 6.6 TB/s) + one head + one hand-off
// instead of two heads, two tails and a boundary.  Shapes: E x F of Llama-3.2-3B (3072 x 8192) or 1B (2048 x 8192); fp16 weights,
// fp32 activations; every block of the chain reads its own weights (cold, as in a token); chain captured in a hipGraph.
// RESULT (profiles/round4_mlp_fused_bench.txt): the hypothesis is WRONG as built — 43.0 against 26.9 us per block at 3B, 34.8 against 19.9 at 1B:
// 2,048 waves waiting on each other inside a launch cost far more than the boundary.  Timing experiment only (the two forms' outputs were not
// bit-identical and that was not tracked down); nothing of it is in the product.
//   hipcc --offload-arch=gfx950 -O3 tools/mlp_fused_bench.hip -o tools/bin/mlp_fused_bench && tools/bin/mlp_fused_bench [E] [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define GLOBAL_AS __attribute__((address_space(1)))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int F = 8192, WAVES = 4, GRID = 512;          // 2 workgroups per CU, 4 waves each: 2048 waves = 4 gate|up row pairs per wave
constexpr int KCH = (F / WAVES) / 512;                    // Wdown: a wave's K slice = F / 4 = 2048 = 4 chunks of 512
constexpr unsigned SPIN_CAP = 1u << 16;

__device__ __forceinline__ float wave_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    const int iv = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48)));
}
__device__ __forceinline__ float h2f_lo(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[0]; }
__device__ __forceinline__ float h2f_hi(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[1]; }
__device__ __forceinline__ float dot8(u32x4 w, f32x4 a, f32x4 b, float acc)
{
    acc = fmaf(h2f_lo(w[0]), a[0], acc); acc = fmaf(h2f_hi(w[0]), a[1], acc); acc = fmaf(h2f_lo(w[1]), a[2], acc); acc = fmaf(h2f_hi(w[1]), a[3], acc);
    acc = fmaf(h2f_lo(w[2]), b[0], acc); acc = fmaf(h2f_hi(w[2]), b[1], acc); acc = fmaf(h2f_lo(w[3]), b[2], acc); acc = fmaf(h2f_hi(w[3]), b[3], acc);
    return acc;
}
__device__ __forceinline__ u32x4 ldw(const unsigned char *p) { return __builtin_nontemporal_load((const GLOBAL_AS u32x4 *)p); }

struct Args {
    const unsigned char *Wg, *Wu, *Wd;   // [F][E], [F][E], [E][F] fp16
    const float *x;                       // [E] block input (also the residual)
    float *act;                           // [F] plain activations (two-launch form)
    unsigned long long *gran;             // [F] {value, tag} granules (fused form)
    float *y;                             // [E] block output
    const unsigned *epoch;
    unsigned blk, E;
    unsigned *err;
    unsigned long long *stamps;           // [GRID][8] s_memrealtime marks of wave 0 of every workgroup (fused form, block 0 only)
    unsigned *done;                       // [8][32] arrival counters, one per blockIdx % 8 label on its own 128-byte line (never reset: epoch * 64 arrivals each)
};

// ---- phase 1: gate|up rows of this wave (pairs dealt round-robin over all waves), x in LDS; act -> plain store and / or granule ---------
template <int EC, bool GRAN>   // EC = E / 512 chunks per row
__device__ __forceinline__ void gateup_phase(const Args &a, float *xs, unsigned tag)
{
    const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6, gw = blockIdx.x * WAVES + wid, tw = GRID * WAVES;
    for (unsigned k = threadIdx.x * 4; k < a.E; k += blockDim.x * 4) *reinterpret_cast<f32x4 *>(xs + k) = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)a.x + k);
    const unsigned npairs = F / tw;   // 4
    u32x4 bufA[2][EC], bufB[2][EC];
    auto issue = [&](u32x4 (&buf)[2][EC], unsigned pr) {
        const unsigned u = min(pr, npairs - 1) * tw + gw;
#pragma unroll
        for (int c = 0; c < EC; c++) {
            buf[0][c] = ldw(a.Wg + ((size_t)u * a.E + c * 512 + lane * 8) * 2);
            buf[1][c] = ldw(a.Wu + ((size_t)u * a.E + c * 512 + lane * 8) * 2);
        }
    };
    issue(bufA, 0);
    __syncthreads();
    issue(bufB, 1);
    auto consume = [&](u32x4 (&buf)[2][EC], unsigned pr) {
        float g = 0.f, up = 0.f;
#pragma unroll
        for (int c = 0; c < EC; c++) {
            const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xs + c * 512 + lane * 8), x1 = *reinterpret_cast<const f32x4 *>(xs + c * 512 + lane * 8 + 4);
            g = dot8(buf[0][c], x0, x1, g);
            up = dot8(buf[1][c], x0, x1, up);
        }
        g = wave_sum(g);
        up = wave_sum(up);
        if (lane == 0) {
            const unsigned u = pr * tw + gw;
            const float v = up * (g / (1.0f + __expf(-g)));
            a.act[u] = v;
            if constexpr (GRAN)
                __hip_atomic_store((GLOBAL_AS unsigned long long *)(a.gran + u), ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    static_assert(F / (GRID * WAVES) == 4, "four row pairs per wave: the ping-pong below is written out");
    consume(bufA, 0);
    issue(bufA, 2);
    consume(bufB, 1);
    issue(bufB, 3);
    consume(bufA, 2);
    consume(bufB, 3);
}

template <int EC>
__global__ __launch_bounds__(256) void k_gu(const Args a)
{
    extern __shared__ float xs[];
    gateup_phase<EC, false>(a, xs, 0);
}

// ---- Wdown: workgroup = RW rows, wave w = K slice [w * 2048, +2048): weights requested at once, x slice in registers -----------------
template <int RW>
__device__ __forceinline__ void down_issue(const Args &a, u32x4 (&wv)[RW][KCH])
{
    const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6, row0 = blockIdx.x * RW;
#pragma unroll
    for (int r = 0; r < RW; r++)
#pragma unroll
        for (int c = 0; c < KCH; c++) wv[r][c] = ldw(a.Wd + ((size_t)(row0 + r) * F + wid * (F / WAVES) + c * 512 + lane * 8) * 2);
}
template <int RW>
__device__ __forceinline__ void down_finish(const Args &a, u32x4 (&wv)[RW][KCH], f32x4 (&xa)[KCH][2], float (*part)[WAVES])
{
    const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6, row0 = blockIdx.x * RW;
#pragma unroll
    for (int r = 0; r < RW; r++) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < KCH; c++) acc = dot8(wv[r][c], xa[c][0], xa[c][1], acc);
        acc = wave_sum(acc);
        if (lane == 0) part[r][wid] = acc;
    }
    __syncthreads();
    if (threadIdx.x < RW) {
        float s = part[threadIdx.x][0];
#pragma unroll
        for (int w = 1; w < WAVES; w++) s += part[threadIdx.x][w];
        a.y[row0 + threadIdx.x] = a.x[row0 + threadIdx.x] + s;
    }
}

template <int RW>
__global__ __launch_bounds__(256) void k_down(const Args a)
{
    __shared__ float part[RW][WAVES];
    const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    f32x4 xa[KCH][2];
#pragma unroll
    for (int c = 0; c < KCH; c++) {
        const unsigned k = wid * (F / WAVES) + c * 512 + lane * 8;
        xa[c][0] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)a.act + k);
        xa[c][1] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)a.act + k + 4);
    }
    u32x4 wv[RW][KCH];
    down_issue<RW>(a, wv);
    down_finish<RW>(a, wv, xa, part);
}

// ---- both in one launch ---------------------------------------------------------------------------------------------------------------
template <int EC, int RW>
__global__ __launch_bounds__(256) void k_mlp_fused(const Args a)
{
    extern __shared__ float xs[];
    __shared__ float part[RW][WAVES];
    const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned tag = a.epoch[0] * 64u + a.blk + 1u;
    gateup_phase<EC, true>(a, xs, tag);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    // this wave's whole share of Wdown: requested before anything of the hand-off is looked at
    u32x4 wv[RW][KCH];
    down_issue<RW>(a, wv);
    // arrival: my granule stores are older than the RW * KCH weight requests just issued (vmcnt retires in order), so "at most that many
    // outstanding" means the stores are acknowledged; the workgroup meets; ONE lane adds to the label's counter (MI355X guide, valid
    // hand-off forms: sc1 stores, every storing wave's wait, barrier, one agent-scope add)
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(RW * KCH) : "memory");
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(a.done + (blockIdx.x & 7u) * 32u + a.blk * 256u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // wait until all eight labels have their 64 arrivals of this replay: one lane polls eight words, the wave sleeps in between
    const unsigned target = a.epoch[0] * (GRID / 8);
    // ONE wave per workgroup polls (eight words, long sleeps: a poller beside streaming waves costs them bandwidth); the others wait at the barrier
    if (wid == 0) {
        bool ok = false;
        for (unsigned spin = 0; spin < SPIN_CAP; spin++) {
            unsigned lo = 0xFFFFFFFFu;
            if (lane < 8) lo = __hip_atomic_load(a.done + lane * 32u + a.blk * 256u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = lo >= target;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(16);
        }
        if (!__all(ok) && lane == 0) a.err[0] = 1;
    }
    __syncthreads();
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    // the K slice of the activations this wave multiplies: 8 granules (64 bytes) per lane and chunk, ONE sweep of sc1 loads
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.gran, 0, F * 8, 0x00020000);
    u32x4 gv[KCH][4];
#pragma unroll
    for (int c = 0; c < KCH; c++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            gv[c][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((wid * (F / WAVES) + c * 512 + lane * 8 + q * 2) * 8), 0, 16));
    bool tags = true;
#pragma unroll
    for (int c = 0; c < KCH; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) tags = tags && gv[c][q][1] == tag && gv[c][q][3] == tag;
    if (!__all(tags) && lane == 0) a.err[0] = 2;
    const unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
    f32x4 xa[KCH][2];
#pragma unroll
    for (int c = 0; c < KCH; c++) {
        xa[c][0] = f32x4{__builtin_bit_cast(float, gv[c][0][0]), __builtin_bit_cast(float, gv[c][0][2]), __builtin_bit_cast(float, gv[c][1][0]), __builtin_bit_cast(float, gv[c][1][2])};
        xa[c][1] = f32x4{__builtin_bit_cast(float, gv[c][2][0]), __builtin_bit_cast(float, gv[c][2][2]), __builtin_bit_cast(float, gv[c][3][0]), __builtin_bit_cast(float, gv[c][3][2])};
    }
    down_finish<RW>(a, wv, xa, part);
    if (a.blk == 0 && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *st = a.stamps + blockIdx.x * 8;
        st[0] = t0; st[1] = t1; st[2] = t2; st[3] = t3; st[4] = t4; st[5] = __builtin_amdgcn_s_memrealtime();
    }
}

__global__ void k_epoch(unsigned *e) { e[0] += 1; }
__global__ void k_fill(unsigned *p, size_t n, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // two small fp16 values (|w| about 2e-3 .. 8e-3: the chain of blocks stays finite): sign | exponent | mantissa
        const unsigned lo = (h & 0x8000u) | (0x1800u + ((h >> 3) & 0x07FFu)), hi = ((h >> 16) & 0x8000u) | (0x1800u + ((h >> 19) & 0x07FFu));
        p[i] = lo | (hi << 16);
    }
}

template <int EC, int RW>
static void run(unsigned E, int blocks, int reps)
{
    const size_t wg = (size_t)F * E * 2, wd = (size_t)E * F * 2, per = 2 * wg + wd;
    unsigned char *W;
    CK(hipMalloc(&W, per * blocks));
    k_fill<<<4096, 256>>>((unsigned *)W, per * blocks / 4, 12345u);
    float *x, *xb[2], *act;
    unsigned long long *gran;
    unsigned *epoch, *err, *done;
    unsigned long long *stamps;
    CK(hipMalloc(&stamps, GRID * 64));
    CK(hipMalloc(&xb[0], E * 4)); CK(hipMalloc(&xb[1], E * 4)); CK(hipMalloc(&act, F * 4)); CK(hipMalloc(&gran, (size_t)F * 8 * blocks));
    CK(hipMalloc(&epoch, 256)); CK(hipMalloc(&err, 256)); CK(hipMalloc(&x, E * 4)); CK(hipMalloc(&done, (size_t)blocks * 1024)); CK(hipMemset(done, 0, (size_t)blocks * 1024));
    CK(hipMemset(gran, 0, (size_t)F * 8 * blocks)); CK(hipMemset(epoch, 0, 256)); CK(hipMemset(err, 0, 256));
    std::vector<float> hx(E);
    for (unsigned i = 0; i < E; i++) hx[i] = 0.5f * (float)((int)((i * 2654435761u) >> 20) - 2048) / 2048.0f;
    CK(hipMemcpy(x, hx.data(), E * 4, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<float> out[2];
    double us[2];
    for (int fused = 0; fused < 2; fused++) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipMemset(epoch, 0, 256));   // the arrival counters count from this variant's first replay
        CK(hipMemset(done, 0, (size_t)blocks * 1024));
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        k_epoch<<<1, 1, 0, s>>>(epoch);
        CK(hipMemcpyAsync(xb[0], x, E * 4, hipMemcpyDeviceToDevice, s));
        for (int b = 0; b < blocks; b++) {
            Args a;
            a.Wg = W + per * b; a.Wu = a.Wg + wg; a.Wd = a.Wu + wg;
            a.x = xb[b & 1]; a.y = xb[(b + 1) & 1]; a.act = act; a.gran = gran + (size_t)F * b; a.epoch = epoch; a.blk = (unsigned)b; a.E = E; a.err = err; a.done = done; a.stamps = stamps;
            if (fused) {
                k_mlp_fused<EC, RW><<<GRID, 256, E * 4, s>>>(a);
            } else {
                k_gu<EC><<<GRID, 256, E * 4, s>>>(a);
                k_down<RW><<<GRID, 256, 0, s>>>(a);
            }
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us[fused] = ms * 1e3 / reps / blocks;
        out[fused].resize(E);
        CK(hipMemcpy(out[fused].data(), xb[blocks & 1], E * 4, hipMemcpyDeviceToHost));
        unsigned herr = 0;
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("E=%u F=%d blocks=%d %s: %.2f us per block (%.0f GB/s)%s\n", E, F, blocks, fused ? "ONE launch (fused)  " : "two launches        ", us[fused],
               per / us[fused] / 1e3, herr == 1 ? "  [a poll gave up!]" : (herr == 2 ? "  [a granule carried the wrong tag!]" : ""));
    }
    {   // where the fused launch of block 0 spends its time (last replay): marks of wave 0 of every workgroup, relative to the earliest start
        std::vector<unsigned long long> st(GRID * 8);
        CK(hipMemcpy(st.data(), stamps, GRID * 64, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull;
        for (int b = 0; b < GRID; b++) first = st[b * 8] < first ? st[b * 8] : first;
        const char *names[6] = {"start", "gate|up done", "stores acknowledged + barrier", "all arrivals seen", "K slice gathered", "end"};
        for (int k = 0; k < 6; k++) {
            std::vector<double> v(GRID);
            for (int b = 0; b < GRID; b++) v[b] = (double)(st[b * 8 + k] - first) * 0.01;
            std::sort(v.begin(), v.end());
            printf("   %-32s median %6.2f us   p90 %6.2f   max %6.2f\n", names[k], v[GRID / 2], v[GRID * 9 / 10], v[GRID - 1]);
        }
    }
    int ndiff = 0;
    double maxd = 0;
    for (unsigned i = 0; i < E; i++) { if (out[0][i] != out[1][i]) ndiff++; const double d = fabs((double)out[0][i] - out[1][i]); if (d > maxd) maxd = d; }
    printf("differing outputs: %d of %u, max |d| = %g\n", ndiff, E, maxd);
    const bool same = memcmp(out[0].data(), out[1].data(), E * 4) == 0;
    printf("outputs %s (y[0] = %g, y[%u] = %g)\n", same ? "bit-identical" : "DIFFER", out[0][0], E - 1, out[0][E - 1]);
}

int main(int argc, char **argv)
{
    const unsigned E = argc > 1 ? atoi(argv[1]) : 3072;
    const int blocks = argc > 2 ? atoi(argv[2]) : 28;
    if (E == 3072) run<6, 6>(E, blocks, 20);
    else if (E == 2048) run<4, 4>(E, blocks, 20);
    else { fprintf(stderr, "E must be 3072 or 2048\n"); return 1; }
    return 0;
}
