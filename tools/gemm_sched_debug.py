"""Where do the gemm256v schedules differ from the eight-wave kernel?  (debug helper, GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip
bf = dict(dtype=torch.bfloat16, device="cuda")
torch.manual_seed(0)
hip.lib()
T = hip.TUNE


def flavour(kind):
    T.reset(); T.tile_mode = 2
    if kind == "e": T.fullline = 1
    else:
        T.fullline = 3; T.sched = 7 if kind == "w" else int(kind)


def report(name, ref, out):
    d = (ref.float() - out.float())
    bad = d != 0
    nb = int(bad.sum())
    if nb == 0:
        return "same"
    idx = bad.nonzero()
    rows, cols = idx[:, 0], idx[:, 1]
    tiles = sorted(set(zip((rows // 256).tolist(), (cols // 256).tolist())))
    inrow = sorted(set((rows % 256 // 16).tolist()))
    incol = sorted(set((cols % 256 // 16).tolist()))
    return "%d differ (max %.3g, ref max %.3g) tiles %d e.g. %s; 16-row blocks in tile %s; 16-col blocks %s; rows%%16 %s" % (
        nb, float(d.abs().max()), float(ref.float().abs().max()), len(tiles), tiles[:6], inrow, incol, sorted(set((rows % 16).tolist()))[:16])


for (M, N, K) in ((256, 256, 128), (256, 256, 192), (300, 512, 256), (5000, 1536, 1536), (70000, 512, 192), (32768, 1536, 6144), (32896, 4608, 1536)):
    x = torch.randn(M, K, **bf)
    w = torch.randn(N, K, **bf) * 0.05
    b = torch.randn(N, **bf)
    res = torch.randn(M, N, **bf)
    for epi in ("bias", "resid"):
        def fn():
            if epi == "bias":
                return hip.gemm_nt(x, [w], [b], splitk=False)
            return hip.gemm_nt(x, [w], [b], epilogue=hip.EPI_RESID, resid=res, splitk=False)
        flavour("e"); ref = fn().clone(); torch.cuda.synchronize()
        for kd in ("w", 1, 3, 6):
            flavour(kd)
            outs = [fn().clone() for _ in range(3)]
            torch.cuda.synchronize()
            rep = all(torch.equal(outs[0], o) for o in outs[1:])
            print("M=%d N=%d K=%d %-5s sched %s: %s | repeatable=%s" % (M, N, K, epi, kd, report("", ref, outs[0]), rep), flush=True)
T.reset()

# grouped launch (three problems with their own weights) against three plain launches
print("--- grouped", flush=True)
H, F = 1536, 6144
for Ms in ((8192, 32896, 32000), (300, 257, 64), (2048, 4096, 1000)):
    xs = [torch.randn(m, H, **bf) for m in Ms]
    w0 = [torch.randn(F, H, **bf) * 0.03 for _ in Ms]
    w1 = [torch.randn(F, H, **bf) * 0.03 for _ in Ms]
    T.reset(); T.tile_mode = 2; T.fullline = 1
    ref = []
    for x, a, b in zip(xs, w0, w1):
        h0, h1 = torch.empty(x.shape[0], F, **bf), torch.empty(x.shape[0], F, **bf)
        ref.append((hip.gemm_nt(x, [a, b], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1), h0, h1))
    T.reset()
    h0s = [torch.empty(x.shape[0], F, **bf) for x in xs]
    h1s = [torch.empty(x.shape[0], F, **bf) for x in xs]
    outs = hip.gemm_nt_grouped(xs, list(zip(w0, w1)), epilogue=hip.EPI_GEGLU, h0s=h0s, h1s=h1s)
    torch.cuda.synchronize()
    ok = outs is not None and all(torch.equal(o, r[0]) and torch.equal(a, r[1]) and torch.equal(b, r[2]) for o, a, b, r in zip(outs, h0s, h1s, ref))
    print("GeGLU grouped", Ms, "identical:", ok, flush=True)
    # down-projection + residual
    gs = [r[0] for r in ref]
    w2 = [torch.randn(H, F, **bf) * 0.02 for _ in Ms]
    b2 = [torch.randn(H, **bf) for _ in Ms]
    gam = [torch.randn(H, **bf) for _ in Ms]
    res = [torch.randn(m, H, **bf) for m in Ms]
    rps = [64, 257, 250] if Ms[0] == 8192 else [1, 1, 1]
    ps = [torch.rand(m // r + 1, device="cuda") for m, r in zip(Ms, rps)]
    T.reset(); T.tile_mode = 2; T.fullline = 1
    ref2 = []
    for i in range(3):
        y = torch.empty(Ms[i], H, **bf)
        ref2.append((hip.gemm_nt(gs[i], [w2[i]], [b2[i]], epilogue=hip.EPI_RESID, resid=res[i], gamma=gam[i], rowscale=ps[i], rows_per_sample=rps[i], h0=y, splitk=False), y))
    T.reset()
    ys = [torch.empty(m, H, **bf) for m in Ms]
    outs = hip.gemm_nt_grouped(gs, w2, biases=b2, epilogue=hip.EPI_RESID, h0s=ys, resids=res, gammas=gam, rowscales=ps, rows_per_sample=rps)
    torch.cuda.synchronize()
    ok = outs is not None and all(torch.equal(o, r[0]) and torch.equal(y, r[1]) for o, y, r in zip(outs, ys, ref2))
    print("down-proj + residual grouped", Ms, "identical:", ok, flush=True)
