"""Round 3: where (which output, which rows / columns) a four-wave GEMM flavour differs from the eight-wave kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

bf = dict(dtype=torch.bfloat16, device="cuda")
torch.manual_seed(0)
hip.lib()
T = hip.TUNE


def flavour(kind):
    T.reset()
    T.tile_mode = 2
    if kind == "e":
        T.fullline = 1
    else:
        T.fullline, T.sched = 3, int(kind)


SHAPES = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(300, 512, 128), (1024, 1536, 1536), (32896, 1536, 256)]
for (M, N, K) in SHAPES:
    x = torch.randn(M, K, **bf)
    w = torch.randn(N, K, **bf) * 0.05
    b, gamma, res = torch.randn(N, **bf), torch.randn(N, **bf), torch.randn(M, N, **bf)
    ps = torch.rand(M // 2 + 1, device="cuda")
    cases = {
        "plain": lambda: (hip.gemm_nt(x, [w], splitk=False),),
        "bias": lambda: (hip.gemm_nt(x, [w], [b], splitk=False),),
        "resid": lambda: hip_resid(),
    }

    def hip_resid():
        y = torch.empty(M, N, **bf)
        o = hip.gemm_nt(x, [w], [b], epilogue=hip.EPI_RESID, resid=res, gamma=gamma, rowscale=ps, rows_per_sample=2, h0=y)
        return o, y

    for name, fn in cases.items():
        flavour("e")
        ref = [t.clone() for t in fn()]
        for kd in (3, 6):
            flavour(kd)
            out = fn()
            torch.cuda.synchronize()
            for i, (a, c) in enumerate(zip(ref, out)):
                bad = (a != c)
                if bad.any():
                    idx = bad.nonzero()
                    print("M=%d N=%d K=%d %s sched %d output %d: %d differ; rows %d..%d cols %d..%d; first %s ref %s got %s; max |diff| %.4g" % (
                        M, N, K, name, kd, i, int(bad.sum()), int(idx[:, 0].min()), int(idx[:, 0].max()), int(idx[:, 1].min()),
                        int(idx[:, 1].max()), idx[0].tolist(), a[tuple(idx[0])].item(), c[tuple(idx[0])].item(),
                        float((a.float() - c.float()).abs().max())))
                else:
                    print("M=%d N=%d K=%d %s sched %d output %d: identical" % (M, N, K, name, kd, i))
T.reset()
