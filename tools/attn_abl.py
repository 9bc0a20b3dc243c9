"""Timing ablations of the resident attention forward kernel (tools only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402
H, heads, B = 1536, 24, 64
bf = dict(dtype=torch.bfloat16, device="cuda")
for S in (257, 320, 250):
    Spad = hip.attn_spad(S)
    qkv = torch.randn(B * S, 3 * H, **bf)
    bias = torch.randn(heads, S, Spad, **bf)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    frag = hip.attn_bias_pack(bias, S)
    for name, mode in (("streaming", 0), ("resident", 1), ("res no-staging", 1 | 2), ("res no-compute", 1 | 4), ("res neither", 1 | 6)):
        hip.lib().op_attn_set_resident(mode)
        for use_bias in (True, False):
            tf = timeit(lambda: hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias if use_bias else None, None, Spad, want_lse=True,
                                             bias_frag=frag if use_bias else None), iters=20)
            print("S=%d %-22s bias=%d: %.4f ms" % (S, name, use_bias, tf), flush=True)
hip.lib().op_attn_set_resident(1)
