"""Attention forward / backward timings at ONE-PEACE-4B shapes (GPU box).  python tools/attn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

H, heads = 1536, 24
bf = dict(dtype=torch.bfloat16, device="cuda")
for (B, S) in ((64, 257), (64, 250), (64, 64), (64, 327)):
    Spad = hip.attn_spad(S)
    qkv = torch.randn(B * S, 3 * H, **bf)
    bias = torch.randn(heads, S, Spad, **bf)
    biasT = bias.transpose(1, 2).contiguous() if S == Spad else torch.randn(heads, S, Spad, **bf)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    out, lse = hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias, None, Spad, want_lse=True)
    dout = torch.randn_like(out)
    fl = 4.0 * B * heads * S * S * 64
    tf = timeit(lambda: hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias, None, Spad, want_lse=True), iters=20)
    tb0 = timeit(lambda: hip.attn_bwd(q, k, v, 3 * H, dout, out, lse, B, S, heads, 0.125, bias, biasT, None, Spad), iters=20)
    tb1 = timeit(lambda: hip.attn_bwd(q, k, v, 3 * H, dout, out, lse, B, S, heads, 0.125, bias, biasT, None, Spad,
                                      want_dbias=True), iters=20)
    print("B=%d S=%d: fwd %.4f ms (%.0f TF)  bwd %.4f ms (%.0f TF)  bwd+dbias %.4f ms (dbias part %.4f)" % (
        B, S, tf, fl / tf / 1e9, tb0, 2.5 * fl / tb0 / 1e9, tb1, tb1 - tb0), flush=True)
