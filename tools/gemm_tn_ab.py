"""Weight-gradient (TN) GEMM dW = dy^T x at the headline batch: eight-wave vs four-wave flavour (tune bits 2-3), bit-equality and time."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip
from tools.bench_ops import timeit

bf = dict(dtype=torch.bfloat16, device="cuda")
IT = int(os.environ.get("ITERS", "50"))
L = hip.lib()
for K in (128 * 257, 128 * 64):
    for M, N in ((1536, 1536), (4608, 1536), (6144, 1536), (1536, 6144)):
        dy, x = torch.randn(K, M, **bf), torch.randn(K, N, **bf)
        outs, ts = {}, {}
        for fl in (21, 23):
            L.op_gemm_set_tile(fl)
            acc = torch.zeros(M, N, **bf)
            outs[fl] = (hip.gemm_tn(dy, x).clone(), hip.gemm_tn(dy, x, acc, True).clone())
            ts[fl] = timeit(lambda: hip.gemm_tn(dy, x, acc, True), iters=IT, warmup=10)
        L.op_gemm_set_tile(22)
        same = all(torch.equal(a, b) for a, b in zip(outs[21], outs[23]))
        fl_ = 2.0 * K * M * N
        print("K=%6d  dW %4d x %4d  bit-identical=%s  eight waves %.3f ms (%.0f TF/s)  four waves %.3f ms (%.0f TF/s)  %+.1f%%" % (
            K, M, N, same, ts[21], fl_ / ts[21] / 1e9, ts[23], fl_ / ts[23] / 1e9, 100 * (ts[21] / ts[23] - 1)), flush=True)
