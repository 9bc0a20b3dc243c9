"""hipBLASLt (through torch) vs the hand-written kernels on the backward GEMM shapes (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip, ops  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

H, Fd = 1536, 6144
bf = dict(dtype=torch.bfloat16, device="cuda")
for M in (64 * 257, 16000, 4096):
    for (Mo, No) in ((Fd, H), (H, Fd), (H, H)):
        dy, x = torch.randn(M, Mo, **bf), torch.randn(M, No, **bf)
        grad = torch.zeros(Mo, No, **bf)
        fl = 2.0 * M * Mo * No
        t_tn = timeit(lambda: ops.wgrad(dy, x, out=grad, accumulate=True), iters=20)
        t_bl = timeit(lambda: torch.addmm(grad, dy.t(), x, out=grad), iters=20)
        t_bl0 = timeit(lambda: torch.mm(dy.t(), x), iters=20)
        print("wgrad M=%5d %4dx%4d: ours(acc) %.4f ms %5.0f TF | hipBLASLt addmm %.4f ms %5.0f TF | mm %.4f ms %5.0f TF" % (
            M, Mo, No, t_tn, fl / t_tn / 1e9, t_bl, fl / t_bl / 1e9, t_bl0, fl / t_bl0 / 1e9), flush=True)
    for (N, K) in ((H, H), (H, 3 * H), (H, Fd), (Fd, H)):
        a, w = torch.randn(M, K, **bf), torch.randn(N, K, **bf) * 0.02   # dgrad: dX = dY @ W  (W^T stored [N_out=K_in...])
        wt = w.t().contiguous()                                            # what ops._transposed keeps: [K, N] -> NT operand [N, K]
        fl = 2.0 * M * N * K
        t_o = timeit(lambda: hip.gemm_nt(a, [w]), iters=20)
        t_b = timeit(lambda: torch.mm(a, wt), iters=20)     # NN on the untransposed weight (no transpose pass needed)
        print("dgrad M=%5d N=%4d K=%4d: ours %.4f ms %5.0f TF | hipBLASLt NN %.4f ms %5.0f TF" % (
            M, N, K, t_o, fl / t_o / 1e9, t_b, fl / t_b / 1e9), flush=True)
