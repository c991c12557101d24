// Micro-benchmark: can consecutive decode launches OVERLAP on gfx950, with the dependency carried by a flag in memory instead
// of the stream order?  A chain of N kernels, each of which (a) records when its workgroups start, (b) waits until the previous
// kernel of the chain has published its counter, (c) "works" for a fixed time, (d) publishes its own counter.  With ordinary
// same-stream launches (b) is satisfied on entry and the chain costs N x (work + boundary); if launch i+1 can be resident while
// launch i works, its start-up is hidden and the chain costs N x work + one start-up.
//   variant 0: one stream, ordinary launches              variant 1: one stream, hipExtLaunchKernel(..., hipExtAnyOrderLaunch)
//   variant 2: S streams round-robin, no events           variant 3/4: variant 1 / 2 captured into a hipGraph and replayed
// Every wait is bounded; a wait that gives up sets *err.
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_bench.hip -o gpurun_out/overlap_bench && gpurun_out/overlap_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Stamp { unsigned long long start, seen, end; };

// ctr[i] = arrivals of chain element i (one per workgroup).  Element i waits for ctr[i-1] == expect.
__global__ void __launch_bounds__(256) k_link(unsigned *ctr, int i, unsigned expect_prev, unsigned work_ticks, Stamp *st, unsigned *err)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && i > 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(&ctr[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect_prev) {
            if (++spins > (1u << 20)) { *err = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t1 < work_ticks) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&ctr[i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st[(size_t)i * gridDim.x + blockIdx.x] = Stamp{t0, t1, t2};
    }
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 40;          // chain length
    const unsigned work_us = argc > 2 ? atoi(argv[2]) : 8;
    const int grid = argc > 3 ? atoi(argv[3]) : 256;
    const int reps = 20;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; chain of %d launches x %u us work, grid %d x 256 threads\n", prop.gcnArchName, prop.multiProcessorCount, N, work_us, grid);
    unsigned *ctr, *err;
    Stamp *st;
    CK(hipMalloc(&ctr, N * sizeof(unsigned)));
    CK(hipMalloc(&err, sizeof(unsigned)));
    CK(hipMalloc(&st, sizeof(Stamp) * N * grid));
    CK(hipMemset(err, 0, sizeof(unsigned)));
    std::vector<Stamp> h(N * (size_t)grid);
    hipStream_t s[4];
    for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej[4];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    for (auto &x : ej) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));

    auto enqueue_chain = [&](int variant, int nstreams, unsigned expect) {
        for (int i = 0; i < N; i++) {
            unsigned work_ticks = work_us * 100;
            void *args[] = {&ctr, &i, &expect, &work_ticks, &st, &err};
            if (variant == 1)
                CK(hipExtLaunchKernel((const void *)k_link, dim3(grid), dim3(256), args, 0, s[0], nullptr, nullptr, hipExtAnyOrderLaunch));
            else
                CK(hipLaunchKernel((const void *)k_link, dim3(grid), dim3(256), args, 0, s[variant == 2 ? i % nstreams : 0]));
        }
    };
    auto report = [&](const char *name, float ms_total, int runs) {
        CK(hipMemcpy(h.data(), st, sizeof(Stamp) * N * grid, hipMemcpyDeviceToHost));
        // of the last run: per launch, first start / last end; overlap = launches whose first workgroup started before the
        // previous launch's last workgroup ended
        int overlapped = 0;
        double gap_sum = 0, head_sum = 0;
        unsigned long long chain0 = ~0ull, chain1 = 0;
        std::vector<unsigned long long> s0(N, ~0ull), s1(N, 0), seen1(N, 0), e1v(N, 0);
        for (int i = 0; i < N; i++)
            for (int b = 0; b < grid; b++) {
                const Stamp &x = h[(size_t)i * grid + b];
                s0[i] = std::min(s0[i], x.start); s1[i] = std::max(s1[i], x.start);
                seen1[i] = std::max(seen1[i], x.seen); e1v[i] = std::max(e1v[i], x.end);
                chain0 = std::min(chain0, x.start); chain1 = std::max(chain1, x.end);
            }
        for (int i = 1; i < N; i++) {
            if (s0[i] < e1v[i - 1]) overlapped++;
            gap_sum += ((double)seen1[i] - (double)e1v[i - 1]) / 100.0;   // previous launch finished -> every workgroup of this one working
            head_sum += ((double)s1[i] - (double)s0[i]) / 100.0;
        }
        unsigned herr = 0;
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("%-44s %8.2f us/chain = %6.2f us/launch (work %u)  in-kernel chain %7.2f us  overlapped starts %d/%d  end->all-working %.2f us  start skew %.2f us  err %u\n",
               name, ms_total * 1000.0 / runs, ms_total * 1000.0 / runs / N, work_us, (chain1 - chain0) / 100.0, overlapped, N - 1, gap_sum / (N - 1), head_sum / (N - 1), herr);
        fflush(stdout);
    };

    // eager variants
    for (int variant = 0; variant <= 2; variant++) {
        for (int nstreams = 2; nstreams <= (variant == 2 ? 4 : 2); nstreams++) {
            float ms = 0;
            for (int r = 0; r < reps + 1; r++) {
                CK(hipMemsetAsync(ctr, 0, N * sizeof(unsigned), s[0]));
                CK(hipStreamSynchronize(s[0]));
                if (r == reps) { CK(hipEventRecord(e0, s[0])); }
                if (variant == 2) {  // the other streams start behind the memset
                    CK(hipEventRecord(ef, s[0]));
                    for (int k = 1; k < nstreams; k++) CK(hipStreamWaitEvent(s[k], ef, 0));
                }
                enqueue_chain(variant, nstreams, (unsigned)grid);
                if (variant == 2)
                    for (int k = 1; k < nstreams; k++) { CK(hipEventRecord(ej[k], s[k])); CK(hipStreamWaitEvent(s[0], ej[k], 0)); }
                if (r == reps) { CK(hipEventRecord(e1, s[0])); }
                CK(hipDeviceSynchronize());
            }
            CK(hipEventElapsedTime(&ms, e0, e1));
            char name[96];
            snprintf(name, sizeof name, variant == 0 ? "eager, one stream" : variant == 1 ? "eager, one stream, hipExtAnyOrderLaunch" : "eager, %d streams round-robin", nstreams);
            report(name, ms, 1);
        }
    }
    // graph variants: capture one chain (counters are cumulative across replays: expect grows, so each replay gets its own launch of
    // the memset inside the graph)
    for (int variant = 2; variant >= 1; variant--) {  // the capture of the any-order launch last: it may be refused
        for (int nstreams = 2; nstreams <= (variant == 2 ? 3 : 2); nstreams++) {
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
            CK(hipMemsetAsync(ctr, 0, N * sizeof(unsigned), s[0]));
            if (variant == 2) {
                CK(hipEventRecord(ef, s[0]));
                for (int k = 1; k < nstreams; k++) CK(hipStreamWaitEvent(s[k], ef, 0));
            }
            enqueue_chain(variant, nstreams, (unsigned)grid);
            if (variant == 2)
                for (int k = 1; k < nstreams; k++) { CK(hipEventRecord(ej[k], s[k])); CK(hipStreamWaitEvent(s[0], ej[k], 0)); }
            CK(hipStreamEndCapture(s[0], &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int r = 0; r < 3; r++) CK(hipGraphLaunch(ge, s[0]));
            CK(hipStreamSynchronize(s[0]));
            CK(hipEventRecord(e0, s[0]));
            for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s[0]));
            CK(hipEventRecord(e1, s[0]));
            CK(hipStreamSynchronize(s[0]));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            char name[96];
            snprintf(name, sizeof name, variant == 1 ? "graph of hipExtAnyOrderLaunch chain" : "graph of %d-stream chain", nstreams);
            report(name, ms, reps);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
