"""Weight-gradient (TN) GEMM dW = dy^T x on the merged launches of the lock-step step: eight-wave vs four-wave flavour
(tune: fullline 1 / 3), bit-equality and time; the planner's own choice (fullline 2) for reference.   ITERS=30"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

bf = dict(dtype=torch.bfloat16, device="cuda")
IT = int(os.environ.get("ITERS", "30"))
hip.lib()
T = hip.TUNE
for K, M, N in ((32896, 12288, 1536), (32896, 1536, 6144), (73216, 4608, 1536), (73216, 1536, 1536), (32000, 12288, 1536), (8320, 12288, 1536),
                (8320, 1536, 6144)):
    dy, x = torch.randn(K, M, **bf), torch.randn(K, N, **bf)
    outs, ts = {}, {}
    for fl in (1, 3, 2):
        T.reset()
        T.fullline = fl
        acc = torch.zeros(M, N, **bf)
        outs[fl] = hip.gemm_tn(dy, x, acc, True).clone()
        ts[fl] = min(timeit(lambda: hip.gemm_tn(dy, x, acc, True), iters=IT, warmup=5) for _ in range(2))
    T.reset()
    flops = 2.0 * K * M * N
    print("K=%6d  dW %5d x %4d  identical=%s  eight waves %.4f ms (%4.0f TF/s)  four waves %.4f ms (%4.0f TF/s)  planner %.4f ms" % (
        K, M, N, torch.equal(outs[1], outs[3]), ts[1], flops / ts[1] / 1e9, ts[3], flops / ts[3] / 1e9, ts[2]), flush=True)
    del dy, x
    torch.cuda.empty_cache()
