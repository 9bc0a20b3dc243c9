import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip
from tools.bench_ops import timeit
bf = dict(dtype=torch.bfloat16, device="cuda")
for (N, K) in ((1536, 6144), (1536, 1536), (4608, 1536), (6144, 1536)):
    for M in (32768, 32896):
        x, w, b = torch.randn(M, K, **bf), torch.randn(N, K, **bf) * 0.02, torch.randn(N, **bf)
        row = []
        for knob in (50, 51):
            hip.lib().op_gemm_set_tile(knob)
            row.append(timeit(lambda: hip.gemm_nt(x, [w], [b]), iters=20))
        print("N=%d K=%d M=%d: no-split %.4f ms  tail-split %.4f ms" % (N, K, M, row[0], row[1]), flush=True)
hip.lib().op_gemm_set_tile(51)
