"""Round 3: time op_ln_geglu_fwd / op_ln_geglu_bwd at the image stream's FFN shape with whatever library ONEPEACE_HIP_LIB names
(grid-cap variants: -DOP_LN_GEGLU_BLOCKS_FWD=... -DOP_LN_GEGLU_BLOCKS_BWD=...).  Buffers rotate so that every launch streams from HBM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

M, F = 128 * 257, 6144
bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib()
n = 3
hh = [torch.randn(M, 2 * F, **bf) for _ in range(n)]
dy = [torch.randn(M, F, **bf) for _ in range(n)]
dd = [torch.empty(M, 2 * F, **bf) for _ in range(n)]
w, b = torch.ones(F, **bf), torch.zeros(F, **bf)
y = torch.empty(M, F, **bf)
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
tw, tb = torch.zeros(F, **bf), torch.zeros(F, **bf)


def timeit_rot(fn, iters=24):
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % n)
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / iters


tf = min(timeit_rot(lambda i: hip.ln_geglu_fwd(hh[i][:, :F], hh[i][:, F:], w, b, out=y, mean=mean, rstd=rstd)) for _ in range(3))
tbk = min(timeit_rot(lambda i: hip.ln_geglu_bwd(dy[i], hh[i][:, :F], hh[i][:, F:], w, mean, rstd, dw=tw, db=tb, accumulate=True,
                                                 dh0=dd[i][:, :F], dh1=dd[i][:, F:])) for _ in range(3))
print("%-10s ln_geglu_fwd %.4f ms (%.0f GB/s)   ln_geglu_bwd %.4f ms (%.0f GB/s)" % (
    sys.argv[1] if len(sys.argv) > 1 else "lib", tf, 6.0 * M * F / tf / 1e6, tbk, 10.0 * M * F / tbk / 1e6), flush=True)
