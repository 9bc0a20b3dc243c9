"""Round 3: time the headline NT shapes with whatever library ONEPEACE_HIP_LIB names (A/B of compile-time variants of csrc/gemm.hip:
build them with `DEFS=... python tools/gemm_timeline.py build`, copy the .so, run this once per copy in the same gpurun call).

    ONEPEACE_HIP_LIB=/path/to/variant.so python tools/gemm_lib_ab.py [label]      ITERS=40 ROUNDS=3 M=32896
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

bf = dict(dtype=torch.bfloat16, device="cuda")
IT, ROUNDS, M = int(os.environ.get("ITERS", "40")), int(os.environ.get("ROUNDS", "3")), int(os.environ.get("M", "32896"))
H, F = 1536, 6144
torch.manual_seed(0)
hip.lib()
x, xf, x3 = torch.randn(M, H, **bf), torch.randn(M, F, **bf), torch.randn(M, 3 * H, **bf)
wq = [torch.randn(H, H, **bf) * 0.03 for _ in range(3)]
bq = [torch.randn(H, **bf), None, torch.randn(H, **bf)]
w0, w1 = torch.randn(F, H, **bf) * 0.03, torch.randn(F, H, **bf) * 0.03
w2, w2t, w3t = torch.randn(H, F, **bf) * 0.02, torch.randn(F, H, **bf) * 0.02, torch.randn(H, 3 * H, **bf) * 0.02
b2, gamma, res = torch.randn(H, **bf), torch.randn(H, **bf), torch.randn(M, H, **bf)
ps = torch.rand(M // 257 + 1, device="cuda")
y, o_h, o_f, o_ff, o_q = torch.empty(M, H, **bf), torch.empty(M, H, **bf), torch.empty(M, F, **bf), torch.empty(M, 2 * F, **bf), torch.empty(M, 3 * H, **bf)
cases = [
    ("qkv", lambda: hip.gemm_nt(x, wq, bq, n_seg=H, N=3 * H, out=o_q), 2.0 * M * 3 * H * H),
    ("up 2seg", lambda: hip.gemm_nt(x, [w0, w1], n_seg=F, N=2 * F, out=o_ff), 4.0 * M * F * H),
    ("out-proj resid", lambda: hip.gemm_nt(x, [wq[0]], [b2], epilogue=hip.EPI_RESID, resid=res, gamma=gamma, rowscale=ps, rows_per_sample=257, h0=y, out=o_h), 2.0 * M * H * H),
    ("down resid", lambda: hip.gemm_nt(xf, [w2], [b2], epilogue=hip.EPI_RESID, resid=res, gamma=gamma, rowscale=ps, rows_per_sample=257, h0=y, out=o_h), 2.0 * M * H * F),
    ("dgrad K6144", lambda: hip.gemm_nt(xf, [w2], out=o_h, splitk=False), 2.0 * M * H * F),
    ("dgrad N6144", lambda: hip.gemm_nt(x, [w2t], out=o_f, splitk=False), 2.0 * M * F * H),
    ("dgrad K4608", lambda: hip.gemm_nt(x3, [w3t], out=o_h, splitk=False), 2.0 * M * H * 3 * H),
]
if os.environ.get("TN", "1") != "0":  # weight-gradient launches of the lock-step step (accumulate into bf16 gradients)
    KA = 73216
    dyf, dyq, xa, gf = torch.randn(M, 2 * F, **bf), torch.randn(KA, 3 * H, **bf), torch.randn(KA, H, **bf), torch.randn(M, F, **bf)
    a_w01, a_w2, a_qkv, a_o = torch.zeros(2 * F, H, **bf), torch.zeros(H, F, **bf), torch.zeros(3 * H, H, **bf), torch.zeros(H, H, **bf)
    cases += [
        ("tn w01", lambda: hip.gemm_tn(dyf, x, a_w01, True), 2.0 * M * 2 * F * H),
        ("tn w2", lambda: hip.gemm_tn(x, gf, a_w2, True), 2.0 * M * F * H),
        ("tn qkv", lambda: hip.gemm_tn(dyq, xa, a_qkv, True), 2.0 * KA * 3 * H * H),
        ("tn out", lambda: hip.gemm_tn(xa, xa, a_o, True), 2.0 * KA * H * H),
    ]
best = {}
for _ in range(ROUNDS):
    for name, fn, fl in cases:
        best[name] = min(best.get(name, 1e9), timeit(fn, iters=IT, warmup=5))
print("%-14s " % (sys.argv[1] if len(sys.argv) > 1 else "lib") + "  ".join("%s %.4f" % (n, best[n]) for n, _, _ in cases) + "  | sum %.4f" % sum(best.values()), flush=True)
