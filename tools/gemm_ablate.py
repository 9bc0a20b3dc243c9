import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip
from tools.bench_ops import timeit
M, H = int(os.environ.get("ABL_B", "64")) * 257, 1536
bf = dict(dtype=torch.bfloat16, device="cuda")
x = torch.randn(M, H, **bf)
w = torch.randn(3 * H, H, **bf) * 0.02
out = torch.empty(M, 3 * H, **bf)
hip.lib().op_gemm_set_tile(2)
for abl, name in ((0, "full"), (1, "no-mfma"), (2, "no-global-loads"), (3, "activations through LDS only (weights neither staged nor read)"), (4, "mfma+barriers only"), (5, "mfma only"), (6, "LDS-DMA + barriers only"), (0, "full")):
    hip.lib().op_gemm_set_tile(10 + abl)
    ms = timeit(lambda: hip.gemm_nt(x, [w], out=out), iters=30)
    print(name, "%.3f ms" % ms, "%.0f TF-equiv" % (2.0 * M * 3 * H * H / ms / 1e9))

hip.lib().op_gemm_set_tile(10)
xz, wz = torch.zeros_like(x), torch.zeros_like(w)
ms = timeit(lambda: hip.gemm_nt(xz, [wz], out=out), iters=30)
print("full, zero operands", "%.3f ms" % ms, "%.0f TF" % (2.0 * M * 3 * H * H / ms / 1e9))
ms = timeit(lambda: torch.matmul(x, w.t()), iters=30)
print("hipBLASLt (torch.matmul)", "%.3f ms" % ms, "%.0f TF" % (2.0 * M * 3 * H * H / ms / 1e9))
