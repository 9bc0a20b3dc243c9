"""Round 3: the GeGLU up-projection + inner LayerNorm of the FFN, two ways, same process:

  fused  : gemm_nt(EPI_GEGLU) writing g, h0, h1  +  layernorm_fwd(g)
  split  : plain two-segment gemm_nt writing h0 | h1  +  ln_geglu_fwd(h0, h1)

    python tools/geglu_split_bench.py         MS=32896,8320  ITERS=30
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

bf = dict(dtype=torch.bfloat16, device="cuda")
IT = int(os.environ.get("ITERS", "30"))
H, F = 1536, 6144
torch.manual_seed(0)
hip.lib()
for M in [int(v) for v in os.environ.get("MS", "32896,16512,8320").split(",")]:
    x = torch.randn(M, H, **bf)
    w0, w1 = torch.randn(F, H, **bf) * 0.03, torch.randn(F, H, **bf) * 0.03
    lw, lb = torch.randn(F, **bf), torch.randn(F, **bf)
    h0, h1, g, y = (torch.empty(M, F, **bf) for _ in range(4))
    hh = torch.empty(M, 2 * F, **bf)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    L = hip.lib()

    def ln_g():
        hip._check(L.op_layernorm_fwd(hip.ptr(g), hip.ptr(lw), hip.ptr(lb), hip.ptr(y), hip.ptr(mean), hip.ptr(rstd), M, F, 1e-5, 0,
                                      hip.DT_BF16, None, hip.stream()), "ln")

    fused_gemm = lambda: hip.gemm_nt(x, [w0, w1], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1, out=g)
    split_gemm = lambda: hip.gemm_nt(x, [w0, w1], n_seg=F, N=2 * F, out=hh)
    split_ln = lambda: hip.ln_geglu_fwd(hh[:, :F], hh[:, F:], lw, lb, out=y, mean=mean, rstd=rstd)
    t = {}
    for _ in range(3):
        for k, fn in (("fused gemm", fused_gemm), ("ln(g)", ln_g), ("split gemm", split_gemm), ("ln_geglu_fwd", split_ln)):
            t[k] = min(t.get(k, 1e9), timeit(fn, iters=IT, warmup=5))
    fl = 4.0 * M * F * H
    print("M=%6d  fused: gemm %.4f ms (%4.0f TF) + ln %.4f = %.4f | split: gemm %.4f ms (%4.0f TF) + ln_geglu %.4f (%.0f GB/s) = %.4f" % (
        M, t["fused gemm"], fl / t["fused gemm"] / 1e9, t["ln(g)"], t["fused gemm"] + t["ln(g)"], t["split gemm"], fl / t["split gemm"] / 1e9,
        t["ln_geglu_fwd"], 3.0 * M * F * 2 / t["ln_geglu_fwd"] / 1e6, t["split gemm"] + t["ln_geglu_fwd"]), flush=True)
