"""Time of the 256^2 NT GEMM vs K at fixed M x N (GPU box): separates the per-launch fixed cost (prologue, epilogue store
burst, wave quantisation) from the marginal main-loop rate; modes: full kernel, MFMA-only ablation, hipBLASLt.

    python tools/gemm_kscan.py [N]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

M = 64 * 257
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib().op_gemm_set_tile(2)
out = torch.empty(M, N, **bf)
for K in (128, 512, 1024, 1536, 3072, 4608, 6144):
    x = torch.randn(M, K, **bf)
    w = torch.randn(N, K, **bf) * 0.02
    row = []
    for abl in (0, 5):
        hip.lib().op_gemm_set_tile(10 + abl)
        row.append(timeit(lambda: hip.gemm_nt(x, [w], out=out, splitk=False), iters=30))
    hip.lib().op_gemm_set_tile(10)
    row.append(timeit(lambda: torch.matmul(x, w.t()), iters=30))
    print("N=%d K=%5d full %.4f  mfma-only %.4f  hipblaslt %.4f ms | TF %.0f %.0f %.0f" % (
        N, K, *row, *[2.0 * M * N * K / r / 1e9 for r in row]), flush=True)
