"""A few attention forward / backward(+dBias) launches at one shape, for rocprofv3 --pmc passes (tools/pmc_attn.sh).

    python tools/attn_probe.py [S] [B] [iters]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 250
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
H, heads = 1536, 24
bf = dict(dtype=torch.bfloat16, device="cuda")
Spad = hip.attn_spad(S)
qkv = torch.randn(B * S, 3 * H, **bf)
bias = torch.randn(heads, S, Spad, **bf)
biasT = torch.randn(heads, S, Spad, **bf)
frag = hip.attn_bias_pack(bias, S)  # round 2: fragment-major copy -> resident forward kernel, matrix-pipe bias add
q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
out, lse = hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias, None, Spad, want_lse=True, bias_frag=frag)
dout = torch.randn_like(out)
for _ in range(iters):
    hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias, None, Spad, want_lse=True, bias_frag=frag)
    hip.attn_bwd(q, k, v, 3 * H, dout, out, lse, B, S, heads, 0.125, bias, biasT, None, Spad, want_dbias=True, bias_frag=frag)
torch.cuda.synchronize()
