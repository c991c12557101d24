// Micro-benchmark: does a poll through the SCALAR memory path overtake a CU's outstanding VECTOR loads?  A CU's vector memory requests
// return in order, whichever wave issued them: a hand-off poll behind 196 KB of prefetched weights waits ~4-5 us.
// Here every workgroup (one per CU, 8 waves) requests KB_PER_WAVE KiB of cold weights per wave (nt loads into registers, dropped), then
// wave 0 reads one word that has long been set
//   V: with a vector load (global_load_dword sc1),   S: with a scalar load (s_load_dword glc),
// and the time from issuing that read to having its value (s_memrealtime, 100 MHz) is reported, beside the time until the wave's own
// weight loads have landed.
//   hipcc --offload-arch=gfx950 -O3 tools/scalar_poll_bench.hip -o tools/bin/scalar_poll_bench && tools/bin/scalar_poll_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))

template <int NLOADS, int MODE>   // NLOADS 1-KiB loads per wave; MODE 0: vector poll, 1: scalar poll, 2: no weights + vector poll (reference latency)
__global__ void __launch_bounds__(512) k_poll(const unsigned char *w, size_t stride_wg, const unsigned *flag, unsigned long long *out, unsigned *sink)
{
    const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const GLOBAL_AS unsigned char *src = (const GLOBAL_AS unsigned char *)w + (size_t)blockIdx.x * stride_wg + (size_t)wave * NLOADS * 1024 + lane * 16;
    u32x4 v[NLOADS];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (MODE != 2) {
#pragma unroll
        for (int j = 0; j < NLOADS; j++) v[j] = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *)(src + j * 1024));
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = 0, t2 = 0;
    unsigned got = 0;
    if (wave == 0) {
        t1 = __builtin_amdgcn_s_memrealtime();
        if (MODE == 1) {
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(got) : "s"(flag) : "memory");
        } else {
            unsigned g;
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(flag + (lane & 0)) : "memory");
            got = __builtin_amdgcn_readfirstlane(g);
        }
        t2 = __builtin_amdgcn_s_memrealtime();
    }
    unsigned acc = got;
    if (MODE != 2) {
#pragma unroll
        for (int j = 0; j < NLOADS; j++) acc += v[j].x ^ v[j].w;
    }
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    if (wave == 0 && lane == 0) { out[blockIdx.x * 4 + 0] = t1 - t0; out[blockIdx.x * 4 + 1] = t2 - t1; out[blockIdx.x * 4 + 2] = t3 - t0; out[blockIdx.x * 4 + 3] = got; }
    if (acc == 0x12345u) *sink = acc;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nb = prop.multiProcessorCount;
    const size_t stride = 8 * 24 * 1024;   // 192 KiB per workgroup
    unsigned char *w; unsigned *flag, *sink; unsigned long long *out; unsigned char *flush;
    CK(hipMalloc(&w, stride * nb)); CK(hipMalloc(&flag, 256)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&out, nb * 32)); CK(hipMalloc(&flush, 1u << 30));
    CK(hipMemset(w, 1, stride * nb));
    unsigned one = 1; CK(hipMemcpy(flag, &one, 4, hipMemcpyHostToDevice));
    auto run = [&](const char *name, auto kern) {
        std::vector<double> issue, poll, land;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipMemset(flush, rep, 1u << 30));   // weights cold again
            hipLaunchKernelGGL(kern, dim3(nb), dim3(512), 0, 0, w, stride, flag, out, sink);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(nb * 4);
            CK(hipMemcpy(h.data(), out, nb * 32, hipMemcpyDeviceToHost));
            if (rep == 0) continue;
            for (int b = 0; b < nb; b++) { issue.push_back(h[b * 4] * 0.01); poll.push_back(h[b * 4 + 1] * 0.01); land.push_back(h[b * 4 + 2] * 0.01); if (h[b * 4 + 3] != 1) { printf("wrong flag value\n"); exit(1); } }
        }
        auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto p90 = [](std::vector<double> &v) { return v[v.size() * 9 / 10]; };
        printf("%-58s requests issued %5.2f us | poll %5.2f us (p90 %5.2f) | own weights landed %5.2f us\n", name, med(issue), med(poll), p90(poll), med(land));
    };
    run("no weights, vector poll (sc1)", k_poll<1, 2>);
    run("24 KiB per wave (192 KiB per CU) in flight, VECTOR poll", k_poll<24, 0>);
    run("24 KiB per wave (192 KiB per CU) in flight, SCALAR poll", k_poll<24, 1>);
    run("8 KiB per wave (64 KiB per CU) in flight, VECTOR poll", k_poll<8, 0>);
    run("8 KiB per wave (64 KiB per CU) in flight, SCALAR poll", k_poll<8, 1>);
    return 0;
}
