"""Round 4: the out-proj + residual launch of the lock-step pass (M = 73 088, N = K = 1536; 3x the output bytes of a bare GEMM:
residual read, output and branch output written) -- `gemm256v` (one tile per workgroup, all CUs reach their epilogue together) against
the persistent `gemm256p` launched with ONE problem (tiles of a CU drift apart: epilogues overlap other CUs' main loops).

    python tools/resid_launch_ab.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib()
H = 1536
for M in (73088, 58368, 32896):
    x, w, b, g = torch.randn(M, H, **bf), torch.randn(H, H, **bf) * 0.03, torch.randn(H, **bf), torch.randn(H, **bf)
    res, out, y = torch.randn(M, H, **bf), torch.empty(M, H, **bf), torch.empty(M, H, **bf)
    ps = torch.rand(M, device="cuda")
    out2, y2 = torch.empty_like(out), torch.empty_like(y)
    one = lambda: hip.gemm_nt(x, [w], [b], epilogue=hip.EPI_RESID, resid=res, gamma=g, rowscale=ps, rows_per_sample=1, h0=y, out=out)  # noqa: E731
    per = lambda: hip.gemm_nt_grouped([x], [w], biases=[b], outs=[out2], epilogue=hip.EPI_RESID, h0s=[y2], resids=[res], gammas=[g],  # noqa: E731
                                      rowscales=[ps], rows_per_sample=[1])
    bare = lambda: hip.gemm_nt(x, [w], out=out, splitk=False)  # noqa: E731
    assert per() is not None
    one()
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(y, y2)
    fl = 2.0 * M * H * H
    for _ in range(2):
        t1, t2, t3 = timeit(one, iters=50, warmup=5), timeit(per, iters=50, warmup=5), timeit(bare, iters=50, warmup=5)
        print("M=%6d  gemm256v + residual %.4f ms %5.0f TF/s | persistent (one problem) %.4f ms %5.0f TF/s | bare GEMM %.4f ms %5.0f TF/s" % (
            M, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, t3, fl / t3 / 1e9), flush=True)
