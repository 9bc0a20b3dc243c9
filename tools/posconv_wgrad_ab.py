"""Round 4: weight gradient of the audio adapter's grouped positional convolution (16 groups of 96 channels, kernel 19, 128 x 268 rows):
16 launches with split-K + folds (rounds 1-3) against ONE grouped persistent launch (op_gemm_tn_grouped with 16 problems).

    python tools/posconv_wgrad_ab.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip, ops  # noqa: E402
from one_peace_amd.audio_ops import _as_rows  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib()
G, cg, k, B, Ts = 16, 96, 19, 128, 268
R = B * Ts
xg = torch.randn(G, R + k, cg, **bf)
dyr = torch.randn(G, R, cg, **bf)
dw1, dw2 = torch.empty(G, cg, k * cg, **bf), torch.empty(G, cg, k * cg, **bf)
probs = [(dyr[g], _as_rows(xg[g], R, k * cg, cg, 0), dw2[g], False) for g in range(G)]


def loop():
    for g in range(G):
        ops.wgrad(dyr[g], _as_rows(xg[g], R, k * cg, cg, 0), out=dw1[g])


assert hip.gemm_tn_grouped(probs)
loop()
torch.cuda.synchronize()
err = float((dw1.float() - dw2.float()).norm() / dw1.float().norm())
fl = 2.0 * G * R * cg * k * cg
for _ in range(2):
    t1, t2 = timeit(loop, iters=20, warmup=3), timeit(lambda: hip.gemm_tn_grouped(probs), iters=20, warmup=3)
    print("16 launches (split-K + folds) %.4f ms %5.0f TF/s | one grouped launch %.4f ms %5.0f TF/s | relative difference of the results %.2e" % (
        t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, err), flush=True)
