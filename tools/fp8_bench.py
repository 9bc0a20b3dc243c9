"""fp8 vs bf16 FFN forward GEMMs (GPU box): the long-sequence shapes of BASELINE configs[4] and the headline image stream.
Round 5: the up-projection in its training form (PLAIN wi_0|wi_1 launch, N = 2F; GELU / gate / LayerNorm(F) in op_ln_geglu_fwd) and the
down-projection + residual on gemm256f8_kernel (256 x 256, four waves) against the 128 x 128 fp8 kernel (tune bit 0) and the bf16 kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

H, Fd = 1536, 6144
bf = dict(dtype=torch.bfloat16, device="cuda")
for M in (64 * 785, 32 * 1025, 128 * 257):
    x, xf = torch.randn(M, H, **bf), torch.randn(M, Fd, **bf)
    w0, w1, w2 = torch.randn(Fd, H, **bf) * 0.02, torch.randn(Fd, H, **bf) * 0.02, torch.randn(H, Fd, **bf) * 0.02
    w01 = torch.cat([w0, w1], 0)
    bias, gamma = torch.randn(H, **bf), torch.rand(H, **bf)
    hh, y, out = torch.empty(M, 2 * Fd, **bf), torch.empty(M, H, **bf), torch.empty(M, H, **bf)
    xq, xs = hip.quant_fp8_rows(x)
    fq, fs = hip.quant_fp8_rows(xf)
    (w01q, w01s), (w2q, w2s) = hip.quant_fp8_rows(w01), hip.quant_fp8_rows(w2)
    t = {}

    def both(name, fn, flops):
        for small in (0, 1):
            hip.TUNE.fp8_small = small
            try:
                t["%s fp8 %s" % (name, "128x128" if small else "256x256 four waves")] = (timeit(fn, iters=20, warmup=5), flops)
            finally:
                hip.TUNE.fp8_small = 0

    t["up-proj (plain, N=12288) bf16"] = (timeit(lambda: hip.gemm_nt(x, [w0, w1], n_seg=Fd, N=2 * Fd, out=hh), iters=20, warmup=5), 4.0 * M * Fd * H)
    both("up-proj (plain, N=12288)", lambda: hip.gemm_nt_fp8(xq, xs, [w01q], [w01s], out=hh), 4.0 * M * Fd * H)
    t["down-proj + residual bf16"] = (timeit(lambda: hip.gemm_nt(xf, [w2], [bias], epilogue=hip.EPI_RESID, resid=x, gamma=gamma, h0=y, out=out), iters=20, warmup=5), 2.0 * M * Fd * H)
    both("down-proj + residual", lambda: hip.gemm_nt_fp8(fq, fs, [w2q], [w2s], bias=bias, epilogue=hip.EPI_RESID, resid=x, gamma=gamma, h0=y, out=out), 2.0 * M * Fd * H)
    t["quant x[1536]"] = (timeit(lambda: hip.quant_fp8_rows(x)), 0)
    t["quant g[6144]"] = (timeit(lambda: hip.quant_fp8_rows(xf)), 0)
    for k, (ms, fl) in t.items():
        print("M=%d %-44s %.4f ms %s" % (M, k, ms, ("%.0f TF/s (%.3f of 5 PF)" % (fl / ms / 1e9, fl / ms / 1e9 / 5000)) if fl else ""), flush=True)
