"""How fast does the vendor library run the encoder's GEMM shapes?  (A yardstick for kernels_encoder.hip, not a product path.)
    python tools/micro/hipblaslt_gemm_ref.py        # on the GPU box
Prints µs per call and TFLOP/s for C[M,N] = A[M,K] · W[N,K]^T in bf16 with fp32 accumulation (torch.matmul -> hipBLASLt)."""
import torch

M = 64 * 1500
SHAPES = [("QKV", 1152, 384), ("O-proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536), ("cross-K/V (8 mats)", 3072, 384)]
dev = torch.device("cuda:0")
for name, N, K in SHAPES:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    print(f"{name:22s} M={M} N={N} K={K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
