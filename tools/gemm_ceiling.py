"""What bounds the 256x256 NT kernel at the headline batch: component ablations of the BK = 32 kernel (tune bits 12-14) with
long, warmed-up timing loops, at M = 64 / 128 samples x 257 tokens and at an M that fills every round exactly."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip
from tools.bench_ops import timeit

bf = dict(dtype=torch.bfloat16, device="cuda")
IT = int(os.environ.get("ITERS", "200"))
VARIANTS = ((0, "full (BK=32 flavour)"), (5, "MFMAs only (no loads, no LDS, no barriers)"), (4, "MFMAs + barriers"),
            (2, "MFMAs + fragment ds_reads (no global loads)"), (3, "activations through LDS only (weights neither staged nor read)"),
            (1, "LDS-DMA + fragment reads, no MFMA"), (6, "LDS-DMA + barriers only"), (0, "full again"))
SHAPES = ((64 * 257, 4608, 1536), (128 * 256, 4608, 1536), (128 * 257, 4608, 1536), (128 * 256, 1536, 1536))
if os.environ.get("SHAPES") == "k":  # fixed cost per tile vs K: same 3.00-round launch at K = 512 ... 6144
    SHAPES = tuple((128 * 256, 1536, k) for k in (512, 1024, 1536, 3072, 6144))
for M, N, K in SHAPES:
    x = torch.randn(M, K, **bf)
    w = torch.randn(N, K, **bf) * 0.02
    out = torch.empty(M, N, **bf)
    fl = 2.0 * M * N * K
    tiles = -(-M // 256) * (N // 256)
    print("## M=%d N=%d K=%d: %d tiles = %.2f rounds of 256 CUs" % (M, N, K, tiles, tiles / 256.0))
    hip.lib().op_gemm_set_tile(0)
    hip.lib().op_gemm_set_tile(22)
    ms = timeit(lambda: hip.gemm_nt(x, [w], out=out), iters=IT, warmup=50)
    print("| production dispatch | %.3f ms | %.0f TF/s |" % (ms, fl / ms / 1e9))
    hip.lib().op_gemm_set_tile(2)   # force the 256 x 256 tile ...
    hip.lib().op_gemm_set_tile(21)  # ... BK = 64 full-line flavour
    ms = timeit(lambda: hip.gemm_nt(x, [w], out=out), iters=IT, warmup=20)
    print("| full (BK=64 full-line flavour) | %.3f ms | %.0f TF/s |" % (ms, fl / ms / 1e9))
    hip.lib().op_gemm_set_tile(20)  # BK = 32 flavour: the one that carries the ablations
    for abl, name in VARIANTS:
        hip.lib().op_gemm_set_tile(10 + abl)
        ms = timeit(lambda: hip.gemm_nt(x, [w], out=out), iters=IT, warmup=20)
        print("| %s | %.3f ms | %.0f TF/s-equivalent |" % (name, ms, fl / ms / 1e9))
    hip.lib().op_gemm_set_tile(10)
    xz, wz = torch.zeros_like(x), torch.zeros_like(w)
    ms = timeit(lambda: hip.gemm_nt(xz, [wz], out=out), iters=IT, warmup=20)
    print("| full, zero operands | %.3f ms | %.0f |" % (ms, fl / ms / 1e9))
    ms = timeit(lambda: torch.matmul(x, w.t()), iters=IT, warmup=20)
    print("| hipBLASLt (torch.matmul) | %.3f ms | %.0f |" % (ms, fl / ms / 1e9))
    hip.lib().op_gemm_set_tile(0)
    hip.lib().op_gemm_set_tile(22)
