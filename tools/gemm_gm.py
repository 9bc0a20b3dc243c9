"""Sweep the L2 tile-group depth (M-tiles per group) of the 256^2 kernels at b=128 shapes (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip, ops  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

H, Fd = 1536, 6144
bf = dict(dtype=torch.bfloat16, device="cuda")
M = 128 * 257
x, xf = torch.randn(M, H, **bf), torch.randn(M, Fd, **bf)
wq = [torch.randn(H, H, **bf) * 0.02 for _ in range(3)]
w0, w1, w2 = torch.randn(Fd, H, **bf) * 0.02, torch.randn(Fd, H, **bf) * 0.02, torch.randn(H, Fd, **bf) * 0.02
bias, gamma = torch.randn(H, **bf), torch.rand(H, **bf)
x3, w3 = torch.cat([x, x, x], 1), torch.cat(wq, 1)
grad = torch.zeros(Fd, H, **bf)
gradh = torch.zeros(H, H, **bf)
cases = {
    "qkv": (lambda: hip.gemm_nt(x, wq, [bias, None, bias], n_seg=H, N=3 * H), 2.0 * M * 3 * H * H),
    "geglu": (lambda: hip.gemm_nt(x, [w0, w1], epilogue=hip.EPI_GEGLU), 4.0 * M * Fd * H),
    "ffn2_resid": (lambda: hip.gemm_nt(xf, [w2], [bias], epilogue=hip.EPI_RESID, resid=x, gamma=gamma), 2.0 * M * H * Fd),
    "proj_1536": (lambda: hip.gemm_nt(x, [wq[0]], [bias]), 2.0 * M * H * H),
    "dgrad_4608": (lambda: hip.gemm_nt(x3, [w3], splitk=False), 2.0 * M * H * 3 * H),
    "dgrad_F_to_H": (lambda: hip.gemm_nt(xf, [w2], splitk=False), 2.0 * M * H * Fd),
    "dgrad_H_to_F": (lambda: hip.gemm_nt(x, [w0], splitk=False), 2.0 * M * H * Fd),
    "wgrad_FxH": (lambda: ops.wgrad(xf, x, out=grad, accumulate=True), 2.0 * M * H * Fd),
    "wgrad_HxH": (lambda: ops.wgrad(x, x, out=gradh, accumulate=True), 2.0 * M * H * H),
}
for name, (fn, flops) in cases.items():
    row = []
    for gm in (1, 2, 4, 8, 16, 32):
        hip.lib().op_gemm_set_tile(40 + gm)
        row.append(timeit(fn, iters=20))
    print("%-13s " % name + "  ".join("gm%-2d %.4f ms %5.0f TF" % (g, t, flops / t / 1e9) for g, t in zip((1, 2, 4, 8, 16, 32), row)), flush=True)
hip.lib().op_gemm_set_tile(40)
