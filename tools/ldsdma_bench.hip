// Micro-benchmark: what one CU's LDS-DMA weight stream delivers on gfx950, by loader waves per CU, slots in flight and cache
// policy — the feed of kernels_engine.hip's ring, without consumers (the ring is overwritten as soon as a slot has landed).
//   hipcc --offload-arch=gfx950 -O3 tools/ldsdma_bench.hip -o gpurun_out/ldsdma_bench && gpurun_out/ldsdma_bench
// Also: the same bytes by plain global_load_dwordx4 into registers (the GEMV kernels' way), same geometry.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if constexpr (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
}

// each loader wave streams `bytes_per_wave` contiguous bytes starting at its own base, 8 pieces (8 KiB) per slot,
// INFLIGHT pieces outstanding; its part of the ring = INFLIGHT + 8 pieces
template <int INFLIGHT, int NT>
__global__ void k_dma(const unsigned char* w, size_t bytes_per_wave, unsigned* sink)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const GLOBAL_AS unsigned char* src = (const GLOBAL_AS unsigned char*)w + ((size_t)blockIdx.x * nw + wave) * bytes_per_wave + lane * 16;
    constexpr unsigned RING = INFLIGHT + 8;
    LDS_AS unsigned char* ring = (LDS_AS unsigned char*)(lds + wave * RING * 1024);
    const unsigned npieces = (unsigned)(bytes_per_wave >> 10);
    unsigned rp = 0;
    for (unsigned i = 0; i < npieces; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            __builtin_amdgcn_global_load_lds(src, ring + (rp + j) * 1024, 16, 0, NT ? 2 : 0);
            src += 1024;
        }
        rp += 8;
        if (rp + 8 > RING) rp = 0;
        wait_vmcnt<INFLIGHT>();
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (threadIdx.x == 0 && lds[5] == 123 && lds[777] == 99) *sink = 1;
}

template <int U>
__global__ void k_reg(const unsigned char* w, size_t bytes_per_wave, unsigned* sink)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const GLOBAL_AS unsigned char* src = (const GLOBAL_AS unsigned char*)w + ((size_t)blockIdx.x * nw + wave) * bytes_per_wave + lane * 16;
    const unsigned npieces = (unsigned)(bytes_per_wave >> 10);
    unsigned acc = 0;
    u32x4 a[U], b[U];
#pragma unroll
    for (int j = 0; j < U; j++) a[j] = __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(src + j * 1024));
#pragma unroll
    for (int j = 0; j < U; j++) b[j] = __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(src + (U + j) * 1024));
    unsigned i = 2 * U;
    for (; i + 2 * U <= npieces; i += 2 * U) {
#pragma unroll
        for (int j = 0; j < U; j++) acc += a[j].x ^ a[j].w;
#pragma unroll
        for (int j = 0; j < U; j++) a[j] = __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(src + (size_t)(i + j) * 1024));
#pragma unroll
        for (int j = 0; j < U; j++) acc += b[j].x ^ b[j].w;
#pragma unroll
        for (int j = 0; j < U; j++) b[j] = __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(src + (size_t)(i + U + j) * 1024));
    }
#pragma unroll
    for (int j = 0; j < U; j++) acc += a[j].x ^ b[j].w;
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nb = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, nb);
    const size_t total = (size_t)3 << 30;  // 3 GiB: far beyond the 256 MiB Infinity Cache
    unsigned char* w;
    unsigned* sink;
    CK(hipMalloc(&w, total));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(w, 1, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int waves, size_t lds, auto kern) {
        const size_t per_wave = (total / ((size_t)nb * waves)) & ~(size_t)(16 * 1024 - 1);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(nb), dim3(waves * 64), lds, 0, w, per_wave, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double bytes = (double)per_wave * waves * nb;
        printf("%-44s %7.3f ms  %6.2f TB/s  %6.1f GB/s per CU\n", name, best, bytes / (best * 1e-3) / 1e12, bytes / nb / (best * 1e-3) / 1e9);
    };
#define DMA(W, INF, NT) run("lds-dma " #W " loader wave(s), " #INF " KiB in flight each, nt=" #NT, W, (size_t)W * (INF + 8) * 1024, k_dma<INF, NT>)
    DMA(1, 8, 1); DMA(1, 16, 1); DMA(1, 32, 1); DMA(1, 48, 1); DMA(1, 56, 1);
    DMA(1, 32, 0); DMA(1, 56, 0);
    DMA(2, 16, 1); DMA(2, 32, 1); DMA(2, 48, 1);
    DMA(4, 16, 1); DMA(4, 24 + 8, 1);
    DMA(2, 32, 0);
    run("registers, 4 waves x 2x8 KiB in flight", 4, 0, k_reg<8>);
    run("registers, 8 waves x 2x8 KiB in flight", 8, 0, k_reg<8>);
    run("registers, 8 waves x 2x12 KiB in flight", 8, 0, k_reg<12>);
    run("registers, 4 waves x 2x12 KiB in flight", 4, 0, k_reg<12>);
    run("registers, 2 waves x 2x12 KiB in flight", 2, 0, k_reg<12>);
    run("registers, 1 wave x 2x12 KiB in flight", 1, 0, k_reg<12>);
    return 0;
}
