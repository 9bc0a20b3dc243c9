"""Run one GEMM configuration repeatedly (for rocprofv3 --pmc passes).
python tools/gemm_probe.py qkv|geglu|ffn2|wgrad [tile] [iters]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip

which = sys.argv[1] if len(sys.argv) > 1 else "qkv"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
M, H, F = (int(sys.argv[4]) if len(sys.argv) > 4 else 64) * 257, 1536, 6144
bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib().op_gemm_set_tile(tile)
x, xf = torch.randn(M, H, **bf), torch.randn(M, F, **bf)
if which == "qkv":
    ws = [torch.randn(H, H, **bf) * 0.02 for _ in range(3)]
    b = torch.randn(H, **bf)
    out = torch.empty(M, 3 * H, **bf)
    fn = lambda: hip.gemm_nt(x, ws, [b, None, b], out=out, n_seg=H, N=3 * H)
elif which == "geglu":
    w0, w1 = torch.randn(F, H, **bf) * 0.02, torch.randn(F, H, **bf) * 0.02
    out = torch.empty(M, F, **bf)
    fn = lambda: hip.gemm_nt(x, [w0, w1], out=out, epilogue=hip.EPI_GEGLU)
elif which == "wgrad":
    dy = torch.randn(M, F, **bf)
    fn = lambda: hip.gemm_tn(dy, x)
else:
    w2 = torch.randn(H, F, **bf) * 0.02
    b = torch.randn(H, **bf)
    out = torch.empty(M, H, **bf)
    fn = lambda: hip.gemm_nt(xf, [w2], [b], out=out, epilogue=hip.EPI_RESID, resid=x)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
