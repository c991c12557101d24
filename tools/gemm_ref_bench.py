#!/usr/bin/env python3
"""What the vendor library (hipBLASLt through torch.matmul) reaches on the prefill GEMM shapes of Llama-3.2-3B,
T = 512 — a yardstick for kernels_prefill.hip, not part of the product path (which never calls it)."""
import torch

def bench(M, N, K, iters=50):
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    b = torch.randn(N, K, device="cuda", dtype=torch.float16)
    for _ in range(5):
        torch.matmul(a, b.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, b.t())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"M={M} N={N} K={K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    for T in (512, 2048):
        for (N, K) in ((5120, 3072), (3072, 3072), (16384, 3072), (3072, 8192)):
            bench(T, N, K)
