"""Merged against separate weight-gradient / input-gradient launches of a layer at the headline batch (image pass: 32 896 tokens)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip, ops
from tools.bench_ops import timeit
bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib()
H, F = 1536, 6144
for N in (32896, 32000, 8192):
    x = torch.randn(N, H, **bf)
    dh = torch.randn(N, 2 * F, **bf); dh0c, dh1c = dh[:, :F].contiguous(), dh[:, F:].contiguous()
    dqkv = torch.randn(N, 3 * H, **bf)
    g01 = torch.zeros(2 * F, H, **bf); gq = torch.zeros(3 * H, H, **bf)
    w01t = torch.randn(H, 2 * F, **bf) * 0.02; w0t, w1t = w01t[:, :F].contiguous(), w01t[:, F:].contiguous()
    r = {}
    r["wgrad w0,w1 separate (contiguous dh)"] = timeit(lambda: (ops.wgrad(dh0c, x, out=g01[:F], accumulate=True), ops.wgrad(dh1c, x, out=g01[F:], accumulate=True)), iters=20)
    r["wgrad w0,w1 separate (strided halves)"] = timeit(lambda: (ops.wgrad(dh[:, :F], x, out=g01[:F], accumulate=True), ops.wgrad(dh[:, F:], x, out=g01[F:], accumulate=True)), iters=20)
    r["wgrad w0|w1 merged"] = timeit(lambda: ops.wgrad(dh, x, out=g01, accumulate=True), iters=20)
    r["wgrad q,k,v separate"] = timeit(lambda: [ops.wgrad(dqkv[:, i * H:(i + 1) * H], x, out=gq[i * H:(i + 1) * H], accumulate=True) for i in range(3)], iters=20)
    r["wgrad q|k|v merged"] = timeit(lambda: ops.wgrad(dqkv, x, out=gq, accumulate=True), iters=20)
    def two():
        o = hip.gemm_nt(dh0c, [w0t])
        hip.gemm_nt(dh1c, [w1t], out=o, epilogue=hip.EPI_RESID, resid=o)
        return o
    r["dgrad 2 x K=6144 chained"] = timeit(two, iters=20)
    r["dgrad K=12288"] = timeit(lambda: hip.gemm_nt(dh, [w01t]), iters=20)
    print("tokens %d" % N)
    for k, v in r.items():
        print("   %-42s %.4f ms" % (k, v), flush=True)
