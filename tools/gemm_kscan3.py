"""Round 3 K-scan: time vs K of the four-wave kernels (w, v3, persistent) and hipBLASLt at M = 32768 (exact rounds), to separate
the per-tile fixed cost from the per-K-tile slope.  NS=1536,4608"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip
from tools.bench_ops import timeit
bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib(); T = hip.TUNE
M = int(os.environ.get("M", "32768"))
for N in [int(v) for v in os.environ.get("NS", "1536,4608").split(",")]:
    rows = []
    for K in (256, 512, 1024, 1536, 3072, 6144):
        x = torch.randn(M, K, **bf); w = torch.randn(N, K, **bf) * 0.03; o = torch.empty(M, N, **bf)
        fn = lambda: hip.gemm_nt(x, [w], out=o, splitk=False)
        res = {}
        for rnd in range(3):
            for kd in ("w", 3, 5, 6):
                T.reset(); T.tile_mode = 2; T.fullline = 3; T.sched = 7 if kd == "w" else kd
                res[str(kd)] = min(res.get(str(kd), 1e9), timeit(fn, iters=50, warmup=5))
            T.reset()
            res["blas"] = min(res.get("blas", 1e9), timeit(lambda: torch.matmul(x, w.t(), out=o), iters=50, warmup=5))
        rows.append((K, res))
        print("N=%d K=%5d " % (N, K) + "  ".join("%s %.4f" % kv for kv in res.items()), flush=True)
    for kd in rows[0][1]:
        (k1, r1), (k2, r2) = rows[3], rows[5]
        slope = (r2[kd] - r1[kd]) / (k2 - k1)
        print("  %-4s slope %.1f ns/K  intercept %.1f us (from K=1536, 6144); tiles %d = %.2f rounds" % (
            kd, slope * 1e6, (r1[kd] - slope * k1) * 1e3, (M // 256) * (N // 256), (M // 256) * (N // 256) / 256))
