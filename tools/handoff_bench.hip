// Micro-benchmark of the attention slices' hand-off (kernels_attn.hip): Hkv x nsplit workgroups; each publishes its G*(D+2) partial
// words as {value, tag} granules and then polls, for its share of the kv head's outputs, the granules of ALL slices of that kv head.
// Question: the slices of a kv head have equal blockIdx % 8, i.e. (observed round-robin placement) share an XCD and its L2.  Does a
// PLAIN store (stays in that L2) + sc1 load (served by the L2) hand over faster than the write-through sc1 store the product uses
// (which drops the line from the L2, so every reader goes to the memory side)?
//   mode 0: sc1 stores -> buffer A; poll A (the product's protocol)
//   mode 1: plain stores -> A and sc1 stores -> B; poll A for `fast` sweeps, then B (never depends on placement)
//   mode 2: plain stores -> A only; poll A (placement-dependent: diagnostic)
// Output per mode: median / p10 / p90 over workgroups x launches of (publish issued -> all granules of the share seen), share of
// polls that finished on the fast copy, and the XCC ids seen per kv head.
//   hipcc --offload-arch=gfx950 -O3 tools/handoff_bench.hip -o tools/bin/handoff_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int HKV = 8, G = 3, D = 128, ROW = D + 2, NS_MAX = 32, BLOCK = 256, GD = G * D;

struct Out { unsigned long long t_pub, t_seen; unsigned fast, xcc; };

__global__ __launch_bounds__(BLOCK) void k_handoff(unsigned long long *A, unsigned long long *B, unsigned nsplit, unsigned tag, int mode, int fast_sweeps, Out *out, unsigned *err, int pub_style, int poll_style, const unsigned char *kv)
{
    const unsigned kvh = blockIdx.x % HKV, split = blockIdx.x / HKV, tid = threadIdx.x, lane = tid & 63;
    if (split >= nsplit) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // stand-in for the slice's own work: a few hundred ns that differ per block
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 20 + (blockIdx.x * 7) % 30) __builtin_amdgcn_s_sleep(1);
    unsigned long long *ga = A + (size_t)kvh * NS_MAX * G * ROW, *gb = B + (size_t)kvh * NS_MAX * G * ROW;
    if (kv) {  // the slice's own K / V rows first (32 KB per block, nt), as the product
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        unsigned acc = 0;
        for (int j = 0; j < 8; j++) {
            const u32x4 v = __builtin_nontemporal_load((const __attribute__((address_space(1))) u32x4 *)(kv + ((size_t)blockIdx.x * 8 + j) * 4096 + tid * 16));
            acc ^= v.x ^ v.w;
        }
        asm volatile("" ::"v"(acc));
    }
    if (pub_style == 1) {
        // the product's form: thread t owns output elements 2t, 2t+1 of the G*D sums: two 8-byte stores, 16 bytes apart between lanes
        for (unsigned pi = tid; pi * 2 < (unsigned)GD; pi += BLOCK) {
            const unsigned e = pi * 2, g = e / D, d = e % D;
            unsigned long long *row = ga + ((size_t)split * G + g) * ROW;
            __hip_atomic_store(row + d, ((unsigned long long)tag << 32) | e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(row + d + 1, ((unsigned long long)tag << 32) | (e + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < (unsigned)G) {
            unsigned long long *row = ga + ((size_t)split * G + tid) * ROW;
            __hip_atomic_store(row + D, ((unsigned long long)tag << 32) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(row + D + 1, ((unsigned long long)tag << 32) | 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else
    for (unsigned e = tid; e < (unsigned)G * ROW; e += BLOCK) {
        const unsigned long long v = ((unsigned long long)tag << 32) | (e + split);
        unsigned long long *pa = ga + (size_t)split * G * ROW + e, *pb = gb + (size_t)split * G * ROW + e;
        if (mode == 0) __hip_atomic_store(pa, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {
            *(volatile unsigned long long *)pa = v;
            if (mode == 1) __hip_atomic_store(pb, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned long long t_pub = __builtin_amdgcn_s_memrealtime();
    const unsigned e0 = GD * split / nsplit, e1 = GD * (split + 1) / nsplit, ne = e1 - e0, nitems = ne * nsplit;
    unsigned off[2];
    for (int it = 0; it < 2; it++) {
        const unsigned i = min(tid + it * BLOCK, nitems - 1), s2 = i / ne, e = e0 + (i - s2 * ne);
        off[it] = (s2 * G + e / D) * ROW + e % D;
    }
    __shared__ unsigned sh_fast;
    if (tid == 0) sh_fast = 1;
    __syncthreads();
    bool ok = false;
    unsigned used_fast = 1;
    if (poll_style == 1) {
        // the product's sweep: this thread's element in ALL 32 slice slots (clamped), 8-byte loads, plus the stats of one head per wave
        const unsigned e = min(e0 + tid, e1 - 1), g = e / D, d = e % D;
        for (unsigned spins = 0; spins < (1u << 14); spins++) {
            unsigned long long a[NS_MAX];
#pragma unroll
            for (int s2 = 0; s2 < NS_MAX; s2++) a[s2] = __hip_atomic_load(ga + ((size_t)min((unsigned)s2, nsplit - 1) * G + g) * ROW + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned gh = min(tid >> 6, (unsigned)G - 1);
            const unsigned long long s0 = __hip_atomic_load(ga + ((size_t)min(lane, nsplit - 1) * G + gh) * ROW + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long s1 = __hip_atomic_load(ga + ((size_t)min(lane, nsplit - 1) * G + gh) * ROW + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (unsigned)(s0 >> 32) == tag && (unsigned)(s1 >> 32) == tag;
#pragma unroll
            for (int s2 = 0; s2 < NS_MAX; s2++) ok = ok && (unsigned)(a[s2] >> 32) == tag;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    } else
    for (unsigned spins = 0; spins < (1u << 14); spins++) {
        const bool from_b = mode == 1 && spins >= (unsigned)fast_sweeps;
        const unsigned long long *src = from_b ? gb : ga;
        unsigned long long v0 = __hip_atomic_load(src + off[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long v1 = __hip_atomic_load(src + off[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = (unsigned)(v0 >> 32) == tag && (unsigned)(v1 >> 32) == tag;
        if (__all(ok)) { used_fast = from_b ? 0u : 1u; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    if (!__all(ok) && lane == 0) *err = 1;
    if (lane == 0 && !used_fast) sh_fast = 0;
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = Out{t_pub, __builtin_amdgcn_s_memrealtime(), sh_fast, xcc};
}

// something streaming beside it (the fused launch has Wo weights in flight): optional second kernel on another stream
int main(int argc, char **argv)
{
    const unsigned nsplit = argc > 1 ? atoi(argv[1]) : 19;
    const int launches = 300;
    unsigned long long *A, *B;
    const size_t words = (size_t)HKV * NS_MAX * G * ROW;
    CK(hipMalloc(&A, words * 8)); CK(hipMalloc(&B, words * 8));
    CK(hipMemset(A, 0, words * 8)); CK(hipMemset(B, 0, words * 8));
    Out *out; unsigned *err;
    const int grid = HKV * nsplit;
    CK(hipMalloc(&out, sizeof(Out) * grid * launches)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    // a big buffer touched between launches so that nothing of the exchange stays cached by accident
    char *junk; const size_t JB = 512ull << 20; CK(hipMalloc(&junk, JB));
    std::vector<Out> h((size_t)grid * launches);
    unsigned tag = 1;
    for (int variant = 0; variant < 6; variant++) {
        const int mode = 0, fast = 2;
        const int pub_style = variant == 5 ? 1 : (variant >> 0) & 1, poll_style = variant == 5 ? 1 : (variant >> 1) & 1, with_kv = variant >= 4;
        if (variant == 5 && false) continue;
        {
            if (variant == 4 || variant == 5) {}

            for (int l = 0; l < launches; l++) {
                if (l % 50 == 0) CK(hipMemsetAsync(junk, l, JB, 0));
                hipLaunchKernelGGL(k_handoff, dim3(grid), dim3(BLOCK), 0, 0, A, B, nsplit, tag++, mode, fast, out + (size_t)l * grid, err, pub_style, poll_style, with_kv ? (const unsigned char *)junk + (size_t)(l % 64) * (8u << 20) : nullptr);
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), out, sizeof(Out) * h.size(), hipMemcpyDeviceToHost));
            unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            std::vector<double> dt; size_t nfast = 0;
            for (size_t i = (size_t)grid * 20; i < h.size(); i++) { dt.push_back((h[i].t_seen - h[i].t_pub) / 100.0); nfast += h[i].fast; }
            std::sort(dt.begin(), dt.end());
            // per launch: slowest block (what the next phase waits for)
            std::vector<double> worst;
            for (int l = 20; l < launches; l++) { double w = 0; for (int b = 0; b < grid; b++) { const Out &o = h[(size_t)l * grid + b]; w = std::max(w, (o.t_seen - o.t_pub) / 100.0); } worst.push_back(w); }
            std::sort(worst.begin(), worst.end());
            unsigned xmask[HKV] = {};
            for (int b = 0; b < grid; b++) xmask[b % HKV] |= 1u << (h[(size_t)(launches - 1) * grid + b].xcc & 15);
            printf("pub_style %d poll_style %d kv_loads %d | mode %d fast_sweeps %d nsplit %u: publish -> all seen  med %.2f  p10 %.2f  p90 %.2f us | slowest block per launch med %.2f p90 %.2f | fast-copy hits %.1f %% | err %u | xcc masks per kv head:",
                   variant == 5 ? 1 : pub_style, variant == 5 ? 1 : poll_style, with_kv, mode, fast, nsplit, dt[dt.size() / 2], dt[dt.size() / 10], dt[dt.size() * 9 / 10], worst[worst.size() / 2], worst[worst.size() * 9 / 10], 100.0 * nfast / dt.size(), herr);
            for (int k = 0; k < HKV; k++) printf(" %x", xmask[k]);
            printf("\n");
            CK(hipMemset(err, 0, 4));
        }
    }
    return 0;
}
