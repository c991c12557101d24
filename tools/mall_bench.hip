// Micro-benchmark: what does a decode launch gain when (part of) its weights already sit in the 256 MiB Infinity Cache?
// A streaming read shaped like the decode GEMV (16 B per lane, 8 loads in flight per lane, 512 workgroups of 256 threads, nt or default
// policy) over a buffer of the decode shapes (31 / 50 / 100 MB), timed with events
//   cold         after a 1 GiB write to another buffer,
//   touched f    after a second kernel has read the first fraction f of the same bytes (default policy or nt) behind that write.
// The "touch" stands for a latency-bound launch (attention) whose idle waves request the next launch's weights and drop them.
//   hipcc --offload-arch=gfx950 -O3 tools/mall_bench.hip -o gpurun_out/mall_bench && gpurun_out/mall_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) k_stream(const f4 *__restrict__ w, size_t n16, float *out)
{
    // workgroup b owns the contiguous share [b, b+1) * n16 / grid, read in steps of 256 lanes x 8 loads
    const size_t per = n16 / gridDim.x, base = per * blockIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per; i += 256 * 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < per) v[u] = NT ? __builtin_nontemporal_load(&w[base + j]) : w[base + j];
            else v[u] = f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 12345.678f) out[blockIdx.x] = s;      // never true for the fill below: keeps the loads
}

__global__ void k_fill(f4 *p, size_t n16, float v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = f4{v, v, v, v};
}

int main()
{
    const size_t flush_bytes = 1ull << 30;
    f4 *flush, *w; float *out;
    CK(hipMalloc(&flush, flush_bytes)); CK(hipMalloc(&w, 128ull << 20)); CK(hipMalloc(&out, 4096 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_fill<<<2048, 256, 0, s>>>(w, (128ull << 20) / 16, 1.0f);
    const size_t sizes[] = {31457280, 50331648, 100663296};
    const float fracs[] = {0.f, 0.25f, 0.5f, 1.0f};
    for (int main_nt = 1; main_nt >= 0; --main_nt)
        for (int touch_nt = 0; touch_nt < 2; ++touch_nt)
            for (size_t bytes : sizes)
                for (float f : fracs) {
                    if (f == 0.f && touch_nt) continue;
                    std::vector<float> us;
                    for (int rep = 0; rep < 12; ++rep) {
                        k_fill<<<2048, 256, 0, s>>>(flush, flush_bytes / 16, (float)rep);
                        const size_t tn = (size_t)(bytes / 16 * f) / 512 * 512;
                        if (tn) { if (touch_nt) k_stream<true><<<512, 256, 0, s>>>(w, tn, out); else k_stream<false><<<512, 256, 0, s>>>(w, tn, out); }
                        CK(hipEventRecord(e0, s));
                        if (main_nt) k_stream<true><<<512, 256, 0, s>>>(w, bytes / 16, out); else k_stream<false><<<512, 256, 0, s>>>(w, bytes / 16, out);
                        CK(hipEventRecord(e1, s));
                        CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (rep >= 2) us.push_back(ms * 1000.f);
                    }
                    std::sort(us.begin(), us.end());
                    const float med = us[us.size() / 2];
                    printf("main %-7s touch %-7s %6.1f MB touched %.2f : %6.2f us  (%5.2f TB/s)\n", main_nt ? "nt" : "default", f == 0.f ? "-" : (touch_nt ? "nt" : "default"),
                           bytes / 1e6, f, med, bytes / (med * 1e-6) / 1e12);
                }
    return 0;
}
