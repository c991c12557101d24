// Micro-benchmark: the decode GEMV as MANY SHORT WORKGROUPS that split K over their waves, against the product's form (one or two
// long-running workgroups per CU whose waves walk whole rows behind a workgroup-wide activation prologue).
//   workgroup = NW waves; wave w owns K-slice w (KS = K / NW elements): its slice of x (fp32) lives in REGISTERS, no LDS, no barrier
//   in front of the stream; the workgroup owns RW consecutive rows; every wave requests its slice of all RW rows at once (RW * KS / 512
//   16-byte loads per lane), multiplies as they land, reduces each row over the wave (DPP) and the waves meet once in LDS at the end.
//   RMSNorm is applied as (sum_k w_k * (x_k * g_k)) / rms: rms needs all of x, the dot products do not wait for it.
// Heads and tails of different workgroups overlap on a CU (several are resident), which a persistent wave cannot do with itself.
// Each launch of the chain reads its own copy of the weights (cold, as in a token).  Output: us per launch, GB/s.
//   hipcc --offload-arch=gfx950 -O3 tools/gemv_sk_bench.hip -o tools/bin/gemv_sk_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define GLOBAL_AS __attribute__((address_space(1)))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

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

// NW waves, each owning CH chunks of 512 K-elements (one 16-byte load per lane per chunk and row), RW rows per workgroup
template <int NW, int CH, int RW, bool NORM>
__global__ __launch_bounds__(NW * 64) void k_gemv_sk(const unsigned char *W, const float *x, const float *gamma, float *y, unsigned N, unsigned K, float eps)
{
    __shared__ float part[RW][NW], ssq[NW];
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned row0 = blockIdx.x * RW;
    const unsigned k0 = (w * CH) * 512 + lane * 8;  // this lane's 8 elements of chunk c: k0 + c * 512
    // activations first (vmcnt retires in order), then every weight request of the workgroup at once
    f32x4 xa[CH][2], ga[CH][2];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        xa[c][0] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)x + k0 + c * 512);
        xa[c][1] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)x + k0 + c * 512 + 4);
        if constexpr (NORM) {
            ga[c][0] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)gamma + k0 + c * 512);
            ga[c][1] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)gamma + k0 + c * 512 + 4);
        }
    }
    u32x4 wv[RW][CH];
#pragma unroll
    for (int r = 0; r < RW; r++)
#pragma unroll
        for (int c = 0; c < CH; c++)
            wv[r][c] = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *)(W + ((size_t)min(row0 + r, N - 1) * K + k0 + c * 512) * 2));
    if constexpr (NORM) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    ss = fmaf(xa[c][h][e], xa[c][h][e], ss);
                    xa[c][h][e] *= ga[c][h][e];
                }
        ss = wave_sum(ss);
        if (lane == 0) ssq[w] = ss;
    }
#pragma unroll
    for (int r = 0; r < RW; r++) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const u32x4 q = wv[r][c];
            acc = fmaf(h2f_lo(q[0]), xa[c][0][0], acc); acc = fmaf(h2f_hi(q[0]), xa[c][0][1], acc);
            acc = fmaf(h2f_lo(q[1]), xa[c][0][2], acc); acc = fmaf(h2f_hi(q[1]), xa[c][0][3], acc);
            acc = fmaf(h2f_lo(q[2]), xa[c][1][0], acc); acc = fmaf(h2f_hi(q[2]), xa[c][1][1], acc);
            acc = fmaf(h2f_lo(q[3]), xa[c][1][2], acc); acc = fmaf(h2f_hi(q[3]), xa[c][1][3], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) part[r][w] = acc;
    }
    __syncthreads();
    if (threadIdx.x < RW && row0 + threadIdx.x < N) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NW; i++) s += part[threadIdx.x][i];
        if constexpr (NORM) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < NW; i++) t += ssq[i];
            s = s / sqrtf(t / (float)K + eps);
        }
        y[row0 + threadIdx.x] = s;
    }
}

// the same with ROUNDS groups of RW rows per workgroup (x loaded once per workgroup, two groups of weight requests in flight)
template <int NW, int CH, int RW, bool NORM>
__global__ __launch_bounds__(NW * 64) void k_gemv_skr(const unsigned char *W, const float *x, const float *gamma, float *y, unsigned N, unsigned K, float eps, unsigned rounds)
{
    __shared__ float part[128][NW], ssq[NW];
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned row0 = blockIdx.x * RW * rounds;
    const unsigned k0 = (w * CH) * 512 + lane * 8;
    f32x4 xa[CH][2], ga[CH][2];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        xa[c][0] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)x + k0 + c * 512);
        xa[c][1] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)x + k0 + c * 512 + 4);
        if constexpr (NORM) {
            ga[c][0] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)gamma + k0 + c * 512);
            ga[c][1] = *reinterpret_cast<const GLOBAL_AS f32x4 *>((const GLOBAL_AS float *)gamma + k0 + c * 512 + 4);
        }
    }
    u32x4 wa[RW][CH], wb[RW][CH];
    auto issue = [&](u32x4 (&buf)[RW][CH], unsigned rd) {
        const unsigned r0 = row0 + min(rd, rounds - 1) * RW;
#pragma unroll
        for (int r = 0; r < RW; r++)
#pragma unroll
            for (int c = 0; c < CH; c++)
                buf[r][c] = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *)(W + ((size_t)min(r0 + r, N - 1) * K + k0 + c * 512) * 2));
    };
    issue(wa, 0);
    issue(wb, 1);
    if constexpr (NORM) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int e = 0; e < 4; e++) { ss = fmaf(xa[c][h][e], xa[c][h][e], ss); xa[c][h][e] *= ga[c][h][e]; }
        ss = wave_sum(ss);
        if (lane == 0) ssq[w] = ss;
    }
    auto consume = [&](u32x4 (&buf)[RW][CH], unsigned rd) {
#pragma unroll
        for (int r = 0; r < RW; r++) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const u32x4 q = buf[r][c];
                acc = fmaf(h2f_lo(q[0]), xa[c][0][0], acc); acc = fmaf(h2f_hi(q[0]), xa[c][0][1], acc);
                acc = fmaf(h2f_lo(q[1]), xa[c][0][2], acc); acc = fmaf(h2f_hi(q[1]), xa[c][0][3], acc);
                acc = fmaf(h2f_lo(q[2]), xa[c][1][0], acc); acc = fmaf(h2f_hi(q[2]), xa[c][1][1], acc);
                acc = fmaf(h2f_lo(q[3]), xa[c][1][2], acc); acc = fmaf(h2f_hi(q[3]), xa[c][1][3], acc);
            }
            acc = wave_sum(acc);
            if (lane == 0 && rd < rounds) part[rd * RW + r][w] = acc;
        }
    };
    for (unsigned rd = 0; rd < rounds; rd += 2) {  // unconditional, clamped refills (exact vmcnt bookkeeping)
        consume(wa, rd);
        issue(wa, rd + 2);
        consume(wb, rd + 1);
        issue(wb, rd + 3);
    }
    __syncthreads();
    const unsigned nrows = RW * rounds;
    if (threadIdx.x < nrows && row0 + threadIdx.x < N) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NW; i++) s += part[threadIdx.x][i];
        if constexpr (NORM) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < NW; i++) t += ssq[i];
            s = s / sqrtf(t / (float)K + eps);
        }
        y[row0 + threadIdx.x] = s;
    }
}

template <int NW, int CH, int RW, bool NORM>
static float runr(const unsigned char *W, size_t wstride, int copies, const float *x, const float *g, float *y, unsigned N, unsigned K, int chain, unsigned rounds)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (N + RW * rounds - 1) / (RW * rounds);
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < chain; i++)
            hipLaunchKernelGGL((k_gemv_skr<NW, CH, RW, NORM>), dim3(grid), dim3(NW * 64), 0, 0, W + (size_t)(i % copies) * wstride, x, g, y, N, K, 1e-5f, rounds);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1000.f / chain;
}

template <int NW, int CH, int RW, bool NORM>
static float run(const unsigned char *W, size_t wstride, int copies, const float *x, const float *g, float *y, unsigned N, unsigned K, int chain)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (N + RW - 1) / RW;
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < chain; i++)
            hipLaunchKernelGGL((k_gemv_sk<NW, CH, RW, NORM>), dim3(grid), dim3(NW * 64), 0, 0, W + (size_t)(i % copies) * wstride, x, g, y, N, K, 1e-5f);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1000.f / chain;
}

int main()
{
    const size_t pool = 3ull << 30;   // weights cycle through 3 GB: every launch of a chain reads cold data
    unsigned char *W; CK(hipMalloc(&W, pool)); CK(hipMemset(W, 0x3c, pool));
    float *x, *g, *y;
    CK(hipMalloc(&x, 65536 * 4)); CK(hipMalloc(&g, 65536 * 4)); CK(hipMalloc(&y, 262144 * 4));
    std::vector<float> hx(65536, 0.5f);
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    struct S { const char *name; unsigned N, K; bool norm; } shapes[] = {
        {"3B q|k|v  5120x3072 norm", 5120, 3072, true}, {"3B Wo     3072x3072", 3072, 3072, false}, {"3B gate|up 16384x3072 norm", 16384, 3072, true},
        {"3B Wdown  3072x8192", 3072, 8192, false}, {"1B q|k|v  3072x2048 norm", 3072, 2048, true}, {"1B Wo     2048x2048", 2048, 2048, false},
        {"1B gate|up 16384x2048 norm", 16384, 2048, true}, {"1B Wdown  2048x8192", 2048, 8192, false}};
    const int chain = 56;
    for (const S &s : shapes) {
        const size_t bytes = (size_t)s.N * s.K * 2, stride = (bytes + 4095) & ~(size_t)4095;
        const int copies = (int)(pool / stride);
        printf("%-30s %6.1f MB:", s.name, bytes / 1e6);
#define RUN(NW, CH, RW)                                                                                                            \
    if (s.K == (unsigned)(NW) * (CH) * 512) {                                                                                       \
        const float us = s.norm ? run<NW, CH, RW, true>(W, stride, copies, x, g, y, s.N, s.K, chain) : run<NW, CH, RW, false>(W, stride, copies, x, g, y, s.N, s.K, chain); \
        printf("  [%dw x %dch, %2d rows] %6.2f us %5.2f TB/s", NW, CH, RW, us, bytes / us / 1e6);                                  \
    }
        RUN(6, 1, 4) RUN(6, 1, 8) RUN(6, 1, 16) RUN(3, 2, 8) RUN(3, 2, 16)        // K = 3072
#define RUNR(NW, CH, RW, GRIDS)                                                                                                    \
    if (s.K == (unsigned)(NW) * (CH) * 512) {                                                                                       \
        const unsigned rounds = (s.N + (GRIDS) * (RW) - 1) / ((GRIDS) * (RW));                                                      \
        if (rounds * (RW) <= 128) {                                                                                                 \
        const float us = s.norm ? runr<NW, CH, RW, true>(W, stride, copies, x, g, y, s.N, s.K, chain, rounds) : runr<NW, CH, RW, false>(W, stride, copies, x, g, y, s.N, s.K, chain, rounds); \
        printf("  [%dw x %dch, %d rows x %u rounds, %d wgs] %6.2f us %5.2f TB/s", NW, CH, RW, rounds, (int)((s.N + rounds * (RW) - 1) / (rounds * (RW))), us, bytes / us / 1e6); }                            \
    }
        RUNR(6, 1, 4, 256) RUNR(6, 1, 4, 512) RUNR(6, 1, 8, 256) RUNR(6, 1, 8, 512) RUNR(6, 1, 4, 1024)
        RUNR(8, 2, 4, 256) RUNR(8, 2, 4, 512) RUNR(4, 1, 4, 256) RUNR(4, 1, 4, 512) RUNR(4, 1, 8, 512)
        RUN(8, 2, 4) RUN(8, 2, 8) RUN(4, 4, 4) RUN(4, 4, 8) RUN(16, 1, 8)           // K = 8192
        RUN(4, 1, 4) RUN(4, 1, 8) RUN(4, 1, 16) RUN(2, 2, 8) RUN(2, 2, 16)        // K = 2048
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
