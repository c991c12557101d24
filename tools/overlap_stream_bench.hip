// Micro-benchmark 2: a chain of weight-STREAMING launches (the shape of the decode GEMVs) with the dependency between neighbours
// carried (a) by the stream order, as the product does today, or (b) by per-workgroup flags in memory, the launches dealt round-robin
// over 2-3 streams so that launch i+1 is resident, with its first NBUF steps of weights requested, while launch i still streams.
// Element i:  request NBUF steps -> wait for all flags of element i-1 (one wave, one 16-byte sc1 load per lane per 256 producers)
//             -> read the 12 KB vector element i-1 wrote (sc1 loads) into LDS -> stream the rest of its weights (ping-pong over the
//             register buffers) -> every workgroup writes its slice of the output vector (sc1), drains, raises its flag (sc1).
// Every wait is bounded.  Output: us per launch and TB/s over the chain, per (bytes per launch, variant).
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_stream_bench.hip -o tools/bin/overlap_stream_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define GLOBAL_AS __attribute__((address_space(1)))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int U = 8;           // 1-KiB loads per wave and step
constexpr int XN = 3072;       // floats in the handed-over vector

struct P {
    const unsigned char *w;    // this launch's weights
    unsigned long long bytes;  // multiple of grid * waves * U KiB
    const float *x_in;         // vector of the previous element (nullptr: none)
    float *x_out;
    const unsigned *flags_in;  // [n_prod] (nullptr: do not wait)
    unsigned n_prod;
    unsigned *flags_out;       // [gridDim.x]
    unsigned tag;
    unsigned *err;
    unsigned long long *stamps;  // [grid][4]: start, flags seen, x in LDS, end (wave 0 of each workgroup)
};

__device__ __forceinline__ u32x4 ld_nt(const unsigned char *p) { return __builtin_nontemporal_load((const GLOBAL_AS u32x4 *)p); }

template <int NBUF>
__global__ void __launch_bounds__(512) k_stream(const P p)
{
    __shared__ __attribute__((aligned(16))) float xs[XN];
    const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned gw = blockIdx.x * nw + wid, tw = gridDim.x * nw;
    const unsigned long long per_wave = p.bytes / tw;
    const unsigned nsteps = (unsigned)(per_wave / (U * 1024));
    const unsigned char *base = p.w + (unsigned long long)gw * per_wave + lane * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), t1 = 0, t2 = 0;
    u32x4 buf[NBUF][U];
    unsigned issued = 0;
#pragma unroll
    for (int b = 0; b < NBUF; b++) {
        const unsigned st = min(issued, nsteps - 1);
#pragma unroll
        for (int j = 0; j < U; j++) buf[b][j] = ld_nt(base + ((unsigned long long)st * U + j) * 1024);
        issued++;
    }
    // ---- dependency ------------------------------------------------------------------------------------------------------
    if (p.flags_in) {
        if (wid == 0) {
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                // one 4-byte sc1 load per lane per 64 producers (wide atomic loads are not available as builtins)
                for (unsigned f = lane; f < p.n_prod; f += 64)
                    ok = ok && __hip_atomic_load(&p.flags_in[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.tag;
                if (__all(ok)) break;
                if (++spins > (1u << 18)) { if (lane == 0) *p.err = 1; break; }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __syncthreads();
    }
    t1 = __builtin_amdgcn_s_memrealtime();
    if (p.x_in) {
        for (unsigned k = threadIdx.x; k < XN; k += blockDim.x) xs[k] = __hip_atomic_load(&p.x_in[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        for (unsigned k = threadIdx.x; k < XN; k += blockDim.x) xs[k] = 1.0f;
    }
    __syncthreads();
    t2 = __builtin_amdgcn_s_memrealtime();
    // ---- stream -------------------------------------------------------------------------------------------------------------
    float acc = 0.f;
    unsigned done = 0;
    auto consume = [&](u32x4 (&b)[U], unsigned st) {
#pragma unroll
        for (int j = 0; j < U; j++) {
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(&xs[((st * U + j) * 256 + lane * 4) % XN]);
            acc = fmaf(__builtin_bit_cast(float, b[j][0] & 0x3F800000u), xv[0], acc);
            acc = fmaf(__builtin_bit_cast(float, b[j][1] & 0x3F800000u), xv[1], acc);
            acc = fmaf(__builtin_bit_cast(float, b[j][2] & 0x3F800000u), xv[2], acc);
            acc = fmaf(__builtin_bit_cast(float, b[j][3] & 0x3F800000u), xv[3], acc);
        }
    };
    while (done + NBUF <= nsteps) {
#pragma unroll
        for (int b = 0; b < NBUF; b++) {
            consume(buf[b], done);
            done++;
            const unsigned st = min(issued, nsteps - 1);   // unconditional, clamped (exact vmcnt bookkeeping)
#pragma unroll
            for (int j = 0; j < U; j++) buf[b][j] = ld_nt(base + ((unsigned long long)st * U + j) * 1024);
            issued++;
        }
    }
#pragma unroll
    for (int b = 0; b < NBUF; b++)
        if (done < nsteps) { consume(buf[b], done); done++; }
    // ---- publish -----------------------------------------------------------------------------------------------------------
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const unsigned per_wg = XN / gridDim.x;  // outputs per workgroup (>= 1 for grid <= 3072)
    if (lane == 0 && wid < per_wg)
        __hip_atomic_store(&p.x_out[blockIdx.x * per_wg + wid], acc * 1e-30f + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&p.flags_out[blockIdx.x], p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p.stamps[blockIdx.x * 4 + 0] = t0;
        p.stamps[blockIdx.x * 4 + 1] = t1;
        p.stamps[blockIdx.x * 4 + 2] = t2;
        p.stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int NBUF> static void launch(const P &p, int grid, int block, hipStream_t s)
{
    hipLaunchKernelGGL(k_stream<NBUF>, dim3(grid), dim3(block), 0, s, p);
}

int main(int argc, char **argv)
{
    const int N = 32;  // chain length; every element has its own weights (N x bytes must exceed the 256 MB Infinity Cache)
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const size_t max_bytes = 104857600;
    unsigned char *w;
    CK(hipMalloc(&w, max_bytes * N));
    CK(hipMemset(w, 0x3c, max_bytes * N));
    float *x;
    unsigned *flags, *err;
    unsigned long long *stamps;
    const int max_grid = 1024;
    CK(hipMalloc(&x, sizeof(float) * XN * (N + 1)));
    CK(hipMalloc(&flags, sizeof(unsigned) * max_grid * (N + 1)));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&stamps, sizeof(unsigned long long) * 4 * max_grid * N));
    CK(hipMemset(flags, 0, sizeof(unsigned) * max_grid * (N + 1)));
    CK(hipMemset(err, 0, 4));
    hipStream_t s[3];
    for (auto &q : s) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej[3];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    for (auto &q : ej) CK(hipEventCreateWithFlags(&q, hipEventDisableTiming));
    unsigned tag = 0;
    std::vector<unsigned long long> hs(4 * (size_t)max_grid * N);

    struct Shape { const char *name; size_t bytes; int bpc, wpb; };
    const Shape shapes[] = {{"Wo 18.9 MB", 18874368, 1, 8}, {"Wo 18.9 MB", 18874368, 1, 4}, {"q|k|v 31.5 MB", 31457280, 1, 8}, {"q|k|v 31.5 MB", 31457280, 2, 4},
                            {"Wdown 50.3 MB", 50331648, 1, 8}, {"Wdown 50.3 MB", 50331648, 2, 4}, {"gate|up 100.7 MB", 100663296, 1, 8}, {"gate|up 100.7 MB", 100663296, 2, 4}, {"gate|up 100.7 MB", 100663296, 2, 8}};
    {
        hipFuncAttributes fa;
        CK(hipFuncGetAttributes(&fa, (const void *)k_stream<2>));
        printf("k_stream<2>: %d VGPRs, k_stream<4>: ", fa.numRegs);
        CK(hipFuncGetAttributes(&fa, (const void *)k_stream<4>));
        printf("%d VGPRs\n", fa.numRegs);
    }
    printf("%s, %d CUs; chain of %d launches, each its own weights; time per launch (us) and TB/s\n", prop.gcnArchName, prop.multiProcessorCount, N);
    for (const Shape &sh : shapes) {
        for (int nbuf : {2, 4}) {
            for (int nstreams = 1; nstreams <= 3; nstreams++) {
                for (int flagged = (nstreams == 1 ? 0 : 1); flagged <= 1; flagged++) {
                    const int grid = prop.multiProcessorCount * sh.bpc, block = sh.wpb * 64;
                    const size_t quantum = (size_t)grid * sh.wpb * U * 1024;
                    const size_t bytes = sh.bytes / quantum * quantum;
                    float best = 1e30f;
                    for (int rep = 0; rep < 6; rep++) {
                        tag++;
                        CK(hipDeviceSynchronize());
                        CK(hipEventRecord(e0, s[0]));
                        if (nstreams > 1) {
                            CK(hipEventRecord(ef, s[0]));
                            for (int k = 1; k < nstreams; k++) CK(hipStreamWaitEvent(s[k], ef, 0));
                        }
                        for (int i = 0; i < N; i++) {
                            P p{};
                            p.w = w + (size_t)i * max_bytes;
                            p.bytes = bytes;
                            p.x_in = i ? x + (size_t)(i - 1) * XN : nullptr;
                            p.x_out = x + (size_t)i * XN;
                            p.flags_in = (flagged && i) ? flags + (size_t)(i - 1) * max_grid : nullptr;
                            p.n_prod = grid;
                            p.flags_out = flags + (size_t)i * max_grid;
                            p.tag = tag;
                            p.err = err;
                            p.stamps = stamps + (size_t)i * 4 * max_grid;
                            if (nbuf == 2) launch<2>(p, grid, block, s[i % nstreams]);
                            else launch<4>(p, grid, block, s[i % nstreams]);
                        }
                        CK(hipGetLastError());
                        for (int k = 1; k < nstreams; k++) { CK(hipEventRecord(ej[k], s[k])); CK(hipStreamWaitEvent(s[0], ej[k], 0)); }
                        CK(hipEventRecord(e1, s[0]));
                        CK(hipDeviceSynchronize());
                        float ms;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        best = std::min(best, ms);
                    }
                    unsigned herr = 0;
                    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
                    // medians over launches 4.. of: own span, start -> flags seen, flags seen -> x in LDS, previous end -> this end
                    std::vector<double> span, wait, xin, period;
                    for (int i = 4; i < N; i++) {
                        unsigned long long a0 = ~0ull, a3 = 0, b3 = 0;
                        double w1 = 0, w2 = 0;
                        for (int b = 0; b < grid; b++) {
                            const unsigned long long *q = &hs[((size_t)i * max_grid + b) * 4], *r = &hs[((size_t)(i - 1) * max_grid + b) * 4];
                            a0 = std::min(a0, q[0]); a3 = std::max(a3, q[3]); b3 = std::max(b3, r[3]);
                            w1 += (double)(q[1] - q[0]); w2 += (double)(q[2] - q[1]);
                        }
                        span.push_back((a3 - a0) / 100.0); wait.push_back(w1 / grid / 100.0); xin.push_back(w2 / grid / 100.0);
                        period.push_back(((double)a3 - (double)b3) / 100.0);
                    }
                    auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
                    printf("%-18s %dx%dw NBUF %d  %d stream%s %-7s %7.2f us/launch  %5.2f TB/s | span %6.2f  start->flags %5.2f  flags->x %4.2f  end-to-end period %6.2f  err %u\n",
                           sh.name, sh.bpc, sh.wpb, nbuf, nstreams, nstreams > 1 ? "s" : " ", flagged ? "flags" : "inorder", best * 1000.0 / N, bytes / (best * 1e-3 / N) / 1e12,
                           med(span), med(wait), med(xin), med(period), herr);
                    fflush(stdout);
                }
            }
        }
    }
    return 0;
}
