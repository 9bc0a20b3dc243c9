"""fp8 vs bf16 FFN GEMMs at the long-sequence shapes of BASELINE configs[4] (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402
H, Fd = 1536, 6144
bf = dict(dtype=torch.bfloat16, device="cuda")
for M in (64 * 785, 32 * 1025):
    x, xf = torch.randn(M, H, **bf), torch.randn(M, Fd, **bf)
    w0, w1, w2 = torch.randn(Fd, H, **bf) * 0.02, torch.randn(Fd, H, **bf) * 0.02, torch.randn(H, Fd, **bf) * 0.02
    bias, gamma = torch.randn(H, **bf), torch.rand(H, **bf)
    h0, h1, y = torch.empty(M, Fd, **bf), torch.empty(M, Fd, **bf), torch.empty(M, H, **bf)
    xq, xs = hip.quant_fp8_rows(x)
    fq, fs = hip.quant_fp8_rows(xf)
    (w0q, w0s), (w1q, w1s), (w2q, w2s) = hip.quant_fp8_rows(w0), hip.quant_fp8_rows(w1), hip.quant_fp8_rows(w2)
    t = {}
    t["geglu bf16"] = (timeit(lambda: hip.gemm_nt(x, [w0, w1], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1)), 4.0 * M * Fd * H)
    t["geglu fp8"] = (timeit(lambda: hip.gemm_nt_fp8(xq, xs, [w0q, w1q], [w0s, w1s], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1)), 4.0 * M * Fd * H)
    t["ffn2 bf16"] = (timeit(lambda: hip.gemm_nt(xf, [w2], [bias], epilogue=hip.EPI_RESID, resid=x, gamma=gamma, h0=y)), 2.0 * M * Fd * H)
    t["ffn2 fp8"] = (timeit(lambda: hip.gemm_nt_fp8(fq, fs, [w2q], [w2s], bias=bias, epilogue=hip.EPI_RESID, resid=x, gamma=gamma, h0=y)), 2.0 * M * Fd * H)
    t["quant x[1536]"] = (timeit(lambda: hip.quant_fp8_rows(x)), 0)
    t["quant g[6144]"] = (timeit(lambda: hip.quant_fp8_rows(xf)), 0)
    for k, (ms, fl) in t.items():
        print("M=%d %-14s %.4f ms %s" % (M, k, ms, ("%.0f TF/s" % (fl / ms / 1e9)) if fl else ""), flush=True)
