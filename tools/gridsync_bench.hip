// Micro-benchmark: cost of an in-kernel grid-wide barrier on gfx950 (one workgroup per CU, all co-resident).
// Decides whether a persistent per-token decode kernel can beat five launches per TransformerBlock
// (a kernel boundary costs ~3.4 us of idle HBM).  Every spin is bounded.
//   hipcc --offload-arch=gfx950 -O3 tools/gridsync_bench.hip -o gpurun_out/gridsync_bench && gpurun_out/gridsync_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int SPIN_CAP = 1 << 22;

// variant 0: one monotonic counter, thread 0 of each block arrives + polls
__global__ void __launch_bounds__(512) k_flat(unsigned* ctr, int iters, unsigned* err) {
    for (int i = 0; i < iters; ++i) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(i + 1) * gridDim.x;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > SPIN_CAP) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
}

// variant 1: with payload — every block publishes a value before the barrier and reads its neighbour's after it
__global__ void __launch_bounds__(512) k_payload(unsigned* ctr, unsigned* data, int iters, unsigned* err) {
    const int nb = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        if (threadIdx.x < 64)
            __hip_atomic_store(&data[blockIdx.x * 64 + threadIdx.x], (unsigned)(i * 7 + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);  // stores issued by this wave have left
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(i + 1) * nb;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > SPIN_CAP) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int src = (blockIdx.x + 37) % nb;
            const unsigned v = __hip_atomic_load(&data[src * 64 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (unsigned)(i * 7 + src)) *err = 2;
        }
    }
}

// variant 2: like 1, but each wave also has 8 x 16-byte streaming loads in flight across the barrier
// (the megakernel's "next phase's weights are already on their way" situation); the poll sits behind them in vmcnt order
__global__ void __launch_bounds__(512) k_stream(unsigned* ctr, const u32x4* w, size_t wn, int iters, unsigned* err, unsigned* sink) {
    const int nb = gridDim.x;
    unsigned acc = 0;
    size_t base = ((size_t)blockIdx.x * 512 + threadIdx.x);
    for (int i = 0; i < iters; ++i) {
        u32x4 r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __builtin_nontemporal_load(&w[(base + (size_t)(i * 8 + j) * nb * 512) % wn]);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(i + 1) * nb;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > SPIN_CAP) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// same streaming without any barrier: the bandwidth ceiling for variant 2
__global__ void __launch_bounds__(512) k_stream_nosync(const u32x4* w, size_t wn, int iters, unsigned* sink) {
    const int nb = gridDim.x;
    unsigned acc = 0;
    size_t base = ((size_t)blockIdx.x * 512 + threadIdx.x);
    for (int i = 0; i < iters; ++i) {
        u32x4 r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __builtin_nontemporal_load(&w[(base + (size_t)(i * 8 + j) * nb * 512) % wn]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void k_empty() {}

// The dependency skeleton of one decode GEMV launch without its weights: every workgroup reads the 12 KB activation
// vector the previous launch wrote, reduces it (RMSNorm-like), and writes its 12 outputs.  What a launch costs when the
// weight stream is free: the floor of the per-kernel "fixed cost" in DESIGN.md 5.1.
__global__ void __launch_bounds__(512) k_chain(const float* __restrict__ x, float* __restrict__ y, int n) {
    __shared__ float red[8];
    float s = 0.f;
    for (int i = threadIdx.x; i < n / 4; i += blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < 8; i++) tot += red[i];
    const int per = n / gridDim.x;
    if (threadIdx.x < per) y[blockIdx.x * per + threadIdx.x] = x[blockIdx.x * per + threadIdx.x] * 0.5f + tot * 1e-9f;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nb = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, nb);
    unsigned *ctr, *err, *data, *sink;
    u32x4* w;
    const size_t wn = (size_t)1 << 26;  // 1 GiB of uint4
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&data, (size_t)nb * 64 * 4));
    CK(hipMalloc(&w, wn * 16));
    CK(hipMemset(w, 1, wn * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int iters, auto launch, double bytes_per_iter) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(ctr, 0, 4)); CK(hipMemset(err, 0, 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            launch(iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h = 0; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
            printf("%-16s iters=%5d  %.3f us/iter  err=%u", name, iters, ms * 1e3 / iters, h);
            if (bytes_per_iter > 0) printf("  %.2f TB/s", bytes_per_iter * iters / (ms * 1e-3) / 1e12);
            printf("\n");
        }
    };
    run("empty launches", 1000, [&](int it) { for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k_empty, dim3(nb), dim3(512), 0, 0); }, 0);
    run("flat", 2000, [&](int it) { hipLaunchKernelGGL(k_flat, dim3(nb), dim3(512), 0, 0, ctr, it, err); }, 0);
    run("payload", 2000, [&](int it) { hipLaunchKernelGGL(k_payload, dim3(nb), dim3(512), 0, 0, ctr, data, it, err); }, 0);
    {
        float *xa, *xb;
        CK(hipMalloc(&xa, 3072 * 4)); CK(hipMalloc(&xb, 3072 * 4));
        CK(hipMemset(xa, 0, 3072 * 4)); CK(hipMemset(xb, 0, 3072 * 4));
        run("chain (eager)", 1000, [&](int it) { for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k_chain, dim3(nb), dim3(512), 0, 0, (i & 1) ? xb : xa, (i & 1) ? xa : xb, 3072); }, 0);
        hipStream_t cs; CK(hipStreamCreate(&cs));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_chain, dim3(nb), dim3(512), 0, cs, (i & 1) ? xb : xa, (i & 1) ? xa : xb, 3072);
        CK(hipStreamEndCapture(cs, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, cs)); CK(hipStreamSynchronize(cs));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, cs));
            for (int k = 0; k < 5; ++k) CK(hipGraphLaunch(ge, cs));
            CK(hipEventRecord(e1, cs));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-16s iters=%5d  %.3f us/iter\n", "chain (graph)", 1000, ms * 1e3 / 1000);
        }
    }
    const double bpi = (double)nb * 512 * 8 * 16;
    run("stream+sync", 2000, [&](int it) { hipLaunchKernelGGL(k_stream, dim3(nb), dim3(512), 0, 0, ctr, w, wn, it, err, sink); }, bpi);
    run("stream nosync", 2000, [&](int it) { hipLaunchKernelGGL(k_stream_nosync, dim3(nb), dim3(512), 0, 0, w, wn, it, sink); }, bpi);
    return 0;
}
