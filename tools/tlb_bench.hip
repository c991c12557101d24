// Micro-benchmark: first-byte latency of a short read (an attention slice's 16 KB of K rows) after ~6.4 GB of other data has
// streamed through the chip (one decode token's weights), by how the small buffer was allocated:
//   a) 56 separate hipMallocs of 2.65 MB (the product: one K and one V cache per block)     b) one 150 MB allocation, carved
//   c) as (a) but read again at once (TLB and caches warm)
// Per workgroup: s_memrealtime from wave start to "16 KB landed in registers" (256 threads x 4 x 16 B), nt loads.
//   hipcc --offload-arch=gfx950 -O3 tools/tlb_bench.hip -o tools/bin/tlb_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))

__global__ __launch_bounds__(256) void k_read(const unsigned char *base, size_t head_stride, size_t slice_bytes, unsigned *lat, unsigned *sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned kvh = blockIdx.x % 8, split = blockIdx.x / 8;
    const unsigned char *p = base + kvh * head_stride + split * slice_bytes + threadIdx.x * 16;
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *)(p + j * 4096));
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    asm volatile("" ::"v"(acc));
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) lat[blockIdx.x * 4 + (threadIdx.x >> 6)] = (unsigned)(t1 - t0);
    if (acc == 0x12345) *sink = acc;
}

__global__ void k_stream(const u32x4 *w, size_t n, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = __builtin_nontemporal_load((const GLOBAL_AS u32x4 *)(w + i));
        acc ^= v.x;
    }
    if (acc == 0x12345) *sink = acc;
}

int main()
{
    const int L = 28;
    const size_t kvb = (size_t)648 * 8 * 128 * 4;  // one cache of the benchmark model
    std::vector<unsigned char *> sep(2 * L);
    for (auto &p : sep) { CK(hipMalloc(&p, kvb)); CK(hipMemset(p, 1, kvb)); }
    unsigned char *slab;
    const size_t kvb_al = (kvb + 255) / 256 * 256;
    CK(hipMalloc(&slab, kvb_al * 2 * L)); CK(hipMemset(slab, 1, kvb_al * 2 * L));
    const size_t WB = 6400ull << 20;
    u32x4 *w; CK(hipMalloc(&w, WB)); CK(hipMemset(w, 2, WB));
    unsigned *lat, *sink; CK(hipMalloc(&lat, 4 * 4 * 256 * 2 * L)); CK(hipMalloc(&sink, 4));
    std::vector<unsigned> h(4 * 256 * 2 * L);
    const size_t head_stride = (size_t)648 * 128 * 4, slice = 31 * 128 * 4;  // head-major [Hkv][C][D], 31 positions per slice
    for (int variant = 0; variant < 3; variant++) {
        std::vector<double> all;
        for (int rep = 0; rep < 4; rep++) {
            if (variant != 2 || rep == 0) { k_stream<<<1024, 256>>>(w, WB / 16, sink); CK(hipDeviceSynchronize()); }
            for (int i = 0; i < 2 * L; i++) {
                const unsigned char *b = variant == 1 ? slab + (size_t)i * kvb_al : sep[i];
                k_read<<<152, 256>>>(b, head_stride, slice, lat + (size_t)i * 4 * 256, sink);
                if (variant == 2) k_read<<<152, 256>>>(b, head_stride, slice, lat + (size_t)i * 4 * 256, sink);  // the second, warm read is what is kept
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), lat, h.size() * 4, hipMemcpyDeviceToHost));
            if (rep == 0) continue;
            for (int i = 0; i < 2 * L; i++) for (int b = 0; b < 152 * 4; b++) all.push_back(h[(size_t)i * 4 * 256 + b] / 100.0);
        }
        std::sort(all.begin(), all.end());
        printf("%-58s wave start -> 16 KB landed: med %.2f  p10 %.2f  p90 %.2f  p99 %.2f  max %.2f us\n",
               variant == 0 ? "56 separate 2.65 MB allocations, after 6.4 GB of traffic" : variant == 1 ? "one 150 MB allocation, after 6.4 GB of traffic" : "separate allocations, read twice back to back (warm)",
               all[all.size() / 2], all[all.size() / 10], all[all.size() * 9 / 10], all[all.size() * 99 / 100], all.back());
    }
    return 0;
}
