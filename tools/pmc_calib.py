"""Known-traffic GEMM for calibrating FETCH_SIZE / WRITE_SIZE on the 256^2 NT kernels (run under rocprofv3 --pmc).
A [32896, 6144] bf16 (404 MB, larger than the 256 MB Infinity Cache) is read exactly once by a launch with ONE column of
output tiles (N = 256); W is 3 MB; C is 16.8 MB.   python tools/pmc_calib.py <fullline 0|1>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

full = int(sys.argv[1]) if len(sys.argv) > 1 else 1
M, N, K = 128 * 257, 256, 6144
bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib().op_gemm_set_tile(2)
hip.lib().op_gemm_set_tile(20 + full)
xs = [torch.randn(M, K, **bf) for _ in range(3)]
w = torch.randn(N, K, **bf)
out = torch.empty(M, N, **bf)
for i in range(6):
    hip.gemm_nt(xs[i % 3], [w], out=out, splitk=False)
torch.cuda.synchronize()
