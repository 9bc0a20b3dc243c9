"""One NT launch shape of the headline step, six launches over three rotating A matrices (for rocprofv3 --pmc FETCH_SIZE passes:
tools/pmc_nt_shapes.sh).   python tools/nt_shape_run.py M N K [resid 0|1] [segments]      GM=<n>: M-tiles per L2 group (tune word);
TIME=1: 3 x 20 timed launches instead, prints the best average."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
resid = len(sys.argv) > 4 and sys.argv[4] == "1"
segs = int(sys.argv[5]) if len(sys.argv) > 5 else 1
bf = dict(dtype=torch.bfloat16, device="cuda")
xs = [torch.randn(M, K, **bf) for _ in range(3)]
ws = [torch.randn(N // segs, K, **bf) * K ** -0.5 for _ in range(segs)]
out = torch.empty(M, N, **bf)
r = torch.randn(M, N, **bf) if resid else None
g = torch.randn(N, **bf) if resid else None
if os.environ.get("GM"):
    hip.TUNE.gm = int(os.environ["GM"])


def launch(i):
    if resid:
        hip.gemm_nt(xs[i % 3], ws, epilogue=hip.EPI_RESID, resid=r, gamma=g, out=out, splitk=False)
    else:
        hip.gemm_nt(xs[i % 3], ws, out=out, n_seg=N // segs, N=N, splitk=False)


if os.environ.get("TIME"):
    best = 1e9
    for _ in range(3):
        for i in range(3):
            launch(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            launch(i)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print("%.4f" % best)
    sys.exit(0)
for i in range(6):
    if resid:
        hip.gemm_nt(xs[i % 3], ws, epilogue=hip.EPI_RESID, resid=r, gamma=g, out=out, splitk=False)
    else:
        hip.gemm_nt(xs[i % 3], ws, out=out, n_seg=N // segs, N=N, splitk=False)
torch.cuda.synchronize()
