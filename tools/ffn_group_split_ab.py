"""Wave quantisation of the N = 1536 launches of a lock-step layer: 286 row tiles x 6 column tiles = 1716 tiles = 6.70 rounds of 256 CUs
(text 32 + image 129 + audio 125 row tiles).  A/B per launch, same process:
  grouped3      the three modality problems as ONE persistent launch (production): 7 tile times
  img+aud|txt   image + audio grouped (254 row tiles x 6 = 1524 tiles = 5.95 rounds) + the text problem as its own launch (planner's choice)
  img+aud|txt128  ... the text problem forced onto 128 x 128 tiles (768 tiles on 512 slots)
and for the shared-weight launches (out-proj, dgrads over all 73 088 rows): whole rounds (65 536 rows) + the last 7 552 rows on 128 x 128 tiles.
    python tools/ffn_group_split_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib()
T = hip.TUNE
H, F = 1536, 6144
Ms = (32896, 32000, 8192)  # image, audio, text


def run(name, fns, flops, rounds=3):
    best = {k: 1e9 for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            T.reset()
            best[k] = min(best[k], timeit(f, iters=20, warmup=3))
    T.reset()
    print("%-34s " % name + "  ".join("%s %.4f ms (%.0f TF)" % (k, v, flops / v / 1e9) for k, v in best.items()), flush=True)


def forced128(fn):
    def g():
        T.tile_mode = 1
        fn()
        T.tile_mode = 0
    return g


for K, label in ((F, "down-proj + residual K=6144"), (2 * F, "FFN input gradient K=12288")):
    A = [torch.randn(m, K, **bf) for m in Ms]
    W = [torch.randn(H, K, **bf) * 0.02 for _ in Ms]
    out = [torch.empty(m, H, **bf) for m in Ms]
    res = [torch.randn(m, H, **bf) for m in Ms]
    b2 = [torch.randn(H, **bf) for _ in Ms]
    gam = [torch.randn(H, **bf) for _ in Ms]
    resid = K == F
    kw = lambda idx: (dict(biases=[b2[i] for i in idx], epilogue=hip.EPI_RESID, resids=[res[i] for i in idx], gammas=[gam[i] for i in idx])  # noqa: E731
                      if resid else {})

    def grouped(idx):
        assert hip.gemm_nt_grouped([A[i] for i in idx], [W[i] for i in idx], outs=[out[i] for i in idx], **kw(idx)) is not None

    def single(i):
        if resid:
            hip.gemm_nt(A[i], [W[i]], [b2[i]], epilogue=hip.EPI_RESID, resid=res[i], gamma=gam[i], out=out[i])
        else:
            hip.gemm_nt(A[i], [W[i]], out=out[i])
    run(label, {"grouped3": lambda: grouped((0, 1, 2)),
                "img+aud|txt": lambda: (grouped((0, 1)), single(2)),
                "img+aud|txt128": lambda: (grouped((0, 1)), forced128(lambda: single(2))())}, 2.0 * sum(Ms) * K * H)

Mall = sum(Ms)
for K, label, resid in ((H, "out-proj + residual K=1536", True), (H, "out-proj input gradient K=1536", False), (3 * H, "q|k|v input gradient K=4608", False)):
    A = torch.randn(Mall, K, **bf)
    W = torch.randn(H, K, **bf) * 0.02
    out = torch.empty(Mall, H, **bf)
    res, b, gam = torch.randn(Mall, H, **bf), torch.randn(H, **bf), torch.randn(H, **bf)

    def part(lo, hi):
        if resid:
            hip.gemm_nt(A[lo:hi], [W], [b], epilogue=hip.EPI_RESID, resid=res[lo:hi], gamma=gam, out=out[lo:hi])
        else:
            hip.gemm_nt(A[lo:hi], [W], out=out[lo:hi])
    run(label, {"one launch": lambda: part(0, Mall),
                "65536 + 7552": lambda: (part(0, 65536), part(65536, Mall)),
                "65536 + 7552 on 128^2": lambda: (part(0, 65536), forced128(lambda: part(65536, Mall))())}, 2.0 * Mall * K * H)
