"""Grouped (three modality FFNs in one launch) against three separate launches, b = 128 shapes; single-problem p / p1 / v3 too."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip
from tools.bench_ops import timeit
bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib(); T = hip.TUNE
H, F = 1536, 6144
Ms = (8192, 32896, 32000)
xs = [torch.randn(m, H, **bf) for m in Ms]
gs = [torch.randn(m, F, **bf) for m in Ms]
w0 = [torch.randn(F, H, **bf) * 0.03 for _ in Ms]; w1 = [torch.randn(F, H, **bf) * 0.03 for _ in Ms]
w2 = [torch.randn(H, F, **bf) * 0.02 for _ in Ms]; w2t = [torch.randn(F, H, **bf) * 0.02 for _ in Ms]
b2 = [torch.randn(H, **bf) for _ in Ms]; gam = [torch.randn(H, **bf) for _ in Ms]
res = [torch.randn(m, H, **bf) for m in Ms]
h0 = [torch.empty(m, F, **bf) for m in Ms]; h1 = [torch.empty(m, F, **bf) for m in Ms]; og = [torch.empty(m, F, **bf) for m in Ms]
oh = [torch.empty(m, H, **bf) for m in Ms]; y = [torch.empty(m, H, **bf) for m in Ms]


def sep_geglu():
    for i in range(3):
        hip.gemm_nt(xs[i], [w0[i], w1[i]], epilogue=hip.EPI_GEGLU, h0=h0[i], h1=h1[i], out=og[i])


def sep_ffn2():
    for i in range(3):
        hip.gemm_nt(gs[i], [w2[i]], [b2[i]], epilogue=hip.EPI_RESID, resid=res[i], gamma=gam[i], h0=y[i], out=oh[i], splitk=False)


def sep_dgrad_w2():  # dgln = dy2 @ W2 (N = F, K = H)
    for i in range(3):
        hip.gemm_nt(xs[i], [w2t[i]], out=og[i], splitk=False)


def sep_dgrad_w0():  # dxln2 = dh0 @ W0 (N = H, K = F)
    for i in range(3):
        hip.gemm_nt(gs[i], [w2[i]], out=oh[i], splitk=False)


cases = {
    "down-proj+resid": (sep_ffn2, lambda: hip.gemm_nt_grouped(gs, w2, biases=b2, outs=oh, epilogue=hip.EPI_RESID, h0s=y, resids=res, gammas=gam), 2.0 * sum(Ms) * F * H),
    "dgrad N=6144 K=1536": (sep_dgrad_w2, lambda: hip.gemm_nt_grouped(xs, w2t, outs=og), 2.0 * sum(Ms) * F * H),
    "dgrad N=1536 K=6144": (sep_dgrad_w0, lambda: hip.gemm_nt_grouped(gs, w2, outs=oh), 2.0 * sum(Ms) * F * H),
}
for name, (sep, grp, flops) in cases.items():
    r = {}
    for rnd in range(3):
        T.reset(); r["separate (auto)"] = min(r.get("separate (auto)", 1e9), timeit(sep, iters=20, warmup=3))
        T.reset(); T.sched = 3; r["separate v3"] = min(r.get("separate v3", 1e9), timeit(sep, iters=20, warmup=3))
        T.reset(); T.sched = 7; r["grouped one-tile"] = min(r.get("grouped one-tile", 1e9), timeit(grp, iters=20, warmup=3))
        T.reset(); T.sched = 6; r["grouped persistent"] = min(r.get("grouped persistent", 1e9), timeit(grp, iters=20, warmup=3))
    T.reset()
    print("%-22s " % name + "  ".join("%s %.4f ms (%.0f TF)" % (k, v, flops / v / 1e9) for k, v in r.items()), flush=True)
