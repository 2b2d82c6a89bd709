#!/usr/bin/env python3
"""Yardstick only (not a product path): the library GEMM (torch.matmul -> hipBLASLt) on the shapes of the 1x1 convolutions."""
import sys, torch
for M, N, K in [(51200, 512, 512), (51200, 512, 1024), (51200, 512, 768), (51200, 256, 512), (51200, 256, 256), (204800, 256, 256), (204800, 256, 512), (204800, 256, 384), (204800, 128, 256)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): c = a @ w.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"M={M} N={N} K={K}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s")
