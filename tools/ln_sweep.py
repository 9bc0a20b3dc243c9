"""Sweep the LayerNorm persistent-grid caps at ONE-PEACE-4B shapes (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

M = 64 * 257
bf = dict(dtype=torch.bfloat16, device="cuda")
for cols in (1536, 6144):
    x, dy = torch.randn(M, cols, **bf), torch.randn(M, cols, **bf)
    w, b = torch.ones(cols, **bf), torch.zeros(cols, **bf)
    y, mean, rstd = hip.layernorm_fwd(x, w, b)
    for blocks in (256, 512, 1024, 2048, 4096):
        hip.lib().op_layernorm_set_grid(blocks, blocks)
        tf = timeit(lambda: hip.layernorm_fwd(x, w, b), iters=50)
        tb = timeit(lambda: hip.layernorm_bwd(dy, x, w, b, mean, rstd), iters=50)
        tba = timeit(lambda: hip.layernorm_bwd(dy, x, w, b, mean, rstd, add=dy), iters=50)
        print("cols %d blocks %4d: fwd %.4f ms (%.0f GB/s)  bwd %.4f ms (%.0f GB/s)  bwd+add %.4f ms (%.0f GB/s)" % (
            cols, blocks, tf, 4.0 * M * cols / tf / 1e6, tb, 6.0 * M * cols / tb / 1e6, tba, 8.0 * M * cols / tba / 1e6), flush=True)

hip.lib().op_layernorm_set_grid(512, 512)
M, F = 64 * 257, 6144
h0, h1, dy = torch.randn(M, F, **bf), torch.randn(M, F, **bf), torch.randn(M, F, **bf)
w, b = torch.ones(F, **bf), torch.zeros(F, **bf)
_, mean, rstd = hip.layernorm_fwd(h0, w, b)
for blocks in (512, 1024, 2048):
    hip.lib().op_layernorm_set_grid(512, blocks)
    t = timeit(lambda: hip.ln_geglu_bwd(dy, h0, h1, w, mean, rstd), iters=30)
    print("ln_geglu_bwd blocks %d: %.4f ms (%.0f GB/s algorithmic)" % (blocks, t, 10.0 * M * F / t / 1e6))
hip.lib().op_layernorm_set_grid(512, 512)
t = timeit(lambda: hip.geglu_bwd(dy, h0, h1), iters=30)
print("geglu_bwd: %.4f ms (%.0f GB/s)" % (t, 10.0 * M * F / t / 1e6))
