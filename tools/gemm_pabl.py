"""Ablations of the persistent kernel's tile boundary (tune bits 12-14 with sched 6): K-scan at N = 1536, M = 32768."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip
from tools.bench_ops import timeit
bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib(); T = hip.TUNE
M = 32768
for N in (1536, 4608):
    for K in (256, 1536):
        x = torch.randn(M, K, **bf); w = torch.randn(N, K, **bf) * 0.03; o = torch.empty(M, N, **bf)
        fn = lambda: hip.gemm_nt(x, [w], out=o, splitk=False)
        res = {}
        for rnd in range(3):
            for name, sched, abl in (("w", 7, 0), ("v3", 3, 0), ("p1", 5, 0), ("p1-noepi", 5, 1), ("p", 6, 0), ("p-noepi", 6, 1)):
                T.reset(); T.tile_mode = 2; T.fullline = 3; T.sched = sched; T.ablation = abl
                res[name] = min(res.get(name, 1e9), timeit(fn, iters=50, warmup=5))
            T.reset()
            res["blas"] = min(res.get("blas", 1e9), timeit(lambda: torch.matmul(x, w.t(), out=o), iters=50, warmup=5))
        print("N=%d K=%5d " % (N, K) + "  ".join("%s %.4f" % kv for kv in res.items()), flush=True)
