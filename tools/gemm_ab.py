"""A/B of the BK = 32 and BK = 64 (full-line) flavours of the 256^2 NT GEMM at ONE-PEACE-4B shapes (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

H, Fd = 1536, 6144
bf = dict(dtype=torch.bfloat16, device="cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hip.lib().op_gemm_set_tile(2)
for M in (64 * 257, 16000, 4096):
    x, xf = torch.randn(M, H, **bf), torch.randn(M, Fd, **bf)
    wq = [torch.randn(H, H, **bf) * 0.02 for _ in range(3)]
    w0, w1, w2 = torch.randn(Fd, H, **bf) * 0.02, torch.randn(Fd, H, **bf) * 0.02, torch.randn(H, Fd, **bf) * 0.02
    bias, gamma = torch.randn(H, **bf), torch.rand(H, **bf)
    x3, w3 = torch.cat([x, x, x], 1), torch.cat(wq, 1)
    ps = (torch.rand(M // 64, device="cuda") > 0.2).float()
    cases = {
        "qkv": (lambda: hip.gemm_nt(x, wq, [bias, None, bias], n_seg=H, N=3 * H), 2.0 * M * 3 * H * H),
        "geglu": (lambda: hip.gemm_nt(x, [w0, w1], epilogue=hip.EPI_GEGLU), 4.0 * M * Fd * H),
        "ffn2_resid": (lambda: hip.gemm_nt(xf, [w2], [bias], epilogue=hip.EPI_RESID, resid=x, gamma=gamma, rowscale=ps,
                                           rows_per_sample=64), 2.0 * M * H * Fd),
        "proj_1536": (lambda: hip.gemm_nt(x, [wq[0]], [bias]), 2.0 * M * H * H),
        "dgrad_4608": (lambda: hip.gemm_nt(x3, [w3], splitk=False), 2.0 * M * H * 3 * H),
    }
    for name, (fn, flops) in cases.items():
        outs, ts = [], []
        for mode in (20, 21, 20, 21):
            hip.lib().op_gemm_set_tile(mode)
            outs.append(fn().float())
            ts.append(timeit(fn, iters=30))
        worst = 0.0
        for _ in range(reps):  # race screen: repeated launches must reproduce the BK = 32 result bit for bit
            worst = max(worst, (fn().float() - outs[0]).abs().max().item())
        err = (outs[0] - outs[1]).abs().max().item()
        print("M=%5d %-11s BK32 %.4f/%.4f ms %6.0f TF | BK64 %.4f/%.4f ms %6.0f TF | max diff %.3g (repeat %.3g)" % (
            M, name, ts[0], ts[2], flops / min(ts[0], ts[2]) / 1e9, ts[1], ts[3], flops / min(ts[1], ts[3]) / 1e9, err, worst),
            flush=True)
hip.lib().op_gemm_set_tile(20)
hip.lib().op_gemm_set_tile(0)
