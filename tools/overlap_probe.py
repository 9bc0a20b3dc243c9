"""Round 5 experiment: can the deferred weight-gradient launch of a layer run BESIDE the next layer's backward chain?

The grouped weight-gradient launch (4 ms per layer, 22 % of the step) has no consumer until the optimiser step, and a third of the backward
chain it follows is HBM-bound (LayerNorm / GeGLU / residual passes) or VALU-bound (attention): MFMA work and HBM work could overlap.  A
four-wave GEMM workgroup owns its CU (all registers, all LDS), so the two kinds cannot share a CU -- the side launch is given a limited
number of workgroups (its tickets make any count correct) and the chain's kernels flow onto the CUs that are left.

    python tools/overlap_probe.py [--nwg 64,96,128,160,192]   ->  chain alone, weight gradients alone, both on two streams
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nwg", default="64,96,128,160,192,256")
ap.add_argument("--layers", type=int, default=6)
ap.add_argument("--chain", default="full", choices=["full", "elementwise", "gemm"])
args = ap.parse_args()
bf = dict(dtype=torch.bfloat16, device="cuda")
dev = torch.device("cuda")
hip.lib()
H, F = 1536, 6144
rows = {"all": 73088, "img": 32896, "aud": 32000, "txt": 8192}
names = [("all", 3 * H, H), ("all", H, H)]
for m in ("img", "aud", "txt"):
    names += [(m, 2 * F, H), (m, H, F)]
probs = [(torch.randn(rows[m], o, **bf), torch.randn(rows[m], i, **bf), torch.zeros(o, i, **bf), True) for m, o, i in names]
wflops = sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for a, b, _, _ in probs)

# ---- one layer's backward chain (what runs between two weight-gradient launches), on fresh buffers of the headline sizes ----
N = rows["all"]
x, dx, dx2 = torch.randn(N, H, **bf), torch.randn(N, H, **bf), torch.empty(N, H, **bf)
w_h, b_h = torch.ones(H, **bf), torch.zeros(H, **bf)
mean, rstd = torch.zeros(N, device=dev), torch.ones(N, device=dev)
segs = [("img", 0), ("aud", rows["img"]), ("txt", rows["img"] + rows["aud"])]
hh = torch.randn(N, 2 * F, **bf)
dgln = torch.randn(N, F, **bf)
dh = torch.empty(N, 2 * F, **bf)
w_f = torch.ones(F, **bf)
w2t = [torch.randn(F, H, **bf) * 0.02 for _ in segs]        # dgrad down-proj: [N_out=F, K=H]
w01t = [torch.randn(H, 2 * F, **bf) * 0.02 for _ in segs]   # dgrad up-proj:   [N_out=H, K=2F]
wot, wqkvt = torch.randn(H, H, **bf) * 0.03, torch.randn(H, 3 * H, **bf) * 0.02
dqkv = torch.randn(N, 3 * H, **bf)
gam = torch.ones(H, **bf)
lnw = torch.zeros(F, **bf)
tw, tb = torch.zeros(H, **bf), torch.zeros(H, **bf)


def chain(kind):
    ew, gm = kind in ("full", "elementwise"), kind in ("full", "gemm")
    for m, r0 in segs:
        r = slice(r0, r0 + rows[m])
        if ew:
            hip.resid_bwd(dx[r], None, gam, None, 1, dbias=tb, accumulate=True)
        if gm:
            hip.gemm_nt(dx[r], [w2t[0]], out=dgln[r])
        if ew:
            hip.ln_geglu_bwd(dgln[r], hh[r, :F], hh[r, F:], w_f, mean[r], rstd[r], dw=lnw, db=lnw, accumulate=True, dh0=dh[r, :F], dh1=dh[r, F:])
    if gm:
        hip.gemm_nt_grouped([dh[slice(r0, r0 + rows[m])] for m, r0 in segs], w01t, outs=[dx2[slice(r0, r0 + rows[m])] for m, r0 in segs])
    if ew:
        hip.layernorm_bwd(dx2, x, w_h, b_h, mean, rstd, add=dx, dw=tw, db=tb, accumulate=True)
        hip.resid_bwd(dx, None, gam, None, 1, dbias=tb, accumulate=True)
    if gm:
        hip.gemm_nt(dx, [wot], out=dx2)
    if ew:
        hip.layernorm_bwd(dx2, x, w_h, b_h, mean, rstd, dw=tw, db=tb, accumulate=True)
    if gm:
        hip.gemm_nt(dqkv, [wqkvt], out=dx2)
    if ew:
        hip.layernorm_bwd(dx2, x, w_h, b_h, mean, rstd, add=dx, dw=tw, db=tb, accumulate=True)


def wall(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


side = torch.cuda.Stream()
L = args.layers


def only_chain():
    for _ in range(L):
        chain(args.chain)


def only_wgrad(nwg=0):
    for _ in range(L):
        assert hip.gemm_tn_grouped(probs, tune=nwg)


def both(nwg):
    for _ in range(L):
        with torch.cuda.stream(side):
            assert hip.gemm_tn_grouped(probs, tune=nwg)
        chain(args.chain)


for f in (only_chain, only_wgrad):
    f()
both(128)
tc, tw_ = wall(only_chain) / L, wall(only_wgrad) / L
print("per layer: chain (%s) alone %.3f ms; weight gradients alone (256 workgroups) %.3f ms = %.0f TF/s; one after the other %.3f ms" % (
    args.chain, tc, tw_, wflops / tw_ / 1e9, tc + tw_), flush=True)
for nwg in [int(v) for v in args.nwg.split(",")]:
    ta = wall(lambda: only_wgrad(nwg)) / L
    tb_ = wall(lambda: both(nwg)) / L
    print("  side launch with %3d workgroups: alone %.3f ms; beside the chain %.3f ms per layer (%+.1f %% against one after the other)" % (
        nwg, ta, tb_, 100.0 * (tb_ / (tc + tw_) - 1)), flush=True)
