"""Ours vs hipBLASLt (torch.mm, NT layout) on the step's NT GEMM shapes at M = 128 x 257; run under
`rocprofv3 --kernel-trace --output-format csv` to get hipBLASLt's kernel names (they spell out its tile / wave / LDS design).

    python tools/blas_nt_names.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

bf = dict(dtype=torch.bfloat16, device="cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 257
for N, K in ((4608, 1536), (1536, 1536), (6144, 1536), (1536, 6144), (1536, 4608)):
    a, w = torch.randn(M, K, **bf), torch.randn(N, K, **bf) * 0.02
    out = torch.empty(M, N, **bf)
    fl = 2.0 * M * N * K
    t_o = timeit(lambda: hip.gemm_nt(a, [w], out=out, splitk=False), iters=20)
    t_b = timeit(lambda: torch.mm(a, w.t(), out=out), iters=20)
    print("NT M=%5d N=%4d K=%4d: ours %.4f ms %5.0f TF | hipBLASLt %.4f ms %5.0f TF" % (
        M, N, K, t_o, fl / t_o / 1e9, t_b, fl / t_b / 1e9), flush=True)
