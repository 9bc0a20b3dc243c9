"""Attention forward / backward timings at ONE-PEACE-4B shapes (GPU box), resident-K/V kernels vs streaming kernels.

    python tools/attn_bench.py [B]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

H, heads = 1536, 24
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bf = dict(dtype=torch.bfloat16, device="cuda")
for S in [int(v) for v in os.environ.get("SS", "257,250,64,320,327").split(",")]:
    Spad = hip.attn_spad(S)
    qkv = torch.randn(B * S, 3 * H, **bf)
    bias = torch.randn(heads, S, Spad, **bf)
    biasT = bias.transpose(1, 2).contiguous() if S == Spad else torch.randn(heads, S, Spad, **bf)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    fl = 4.0 * B * heads * S * S * 64
    algo_bytes = 4 * B * S * H * 2  # q, k, v read once + out written once
    frag = hip.attn_bias_pack(bias, S)
    for res in (1, 0):
        hip.lib().op_attn_set_resident(res)
        out, lse = hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias, None, Spad, want_lse=True, bias_frag=frag)
        dout = torch.randn_like(out)
        tf = timeit(lambda: hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias, None, Spad, want_lse=True, bias_frag=frag), iters=20)
        fr = frag if res else None  # "streaming" arm = the round-1 softmax code of the backward kernels as well
        tb0 = timeit(lambda: hip.attn_bwd(q, k, v, 3 * H, dout, out, lse, B, S, heads, 0.125, bias, biasT, None, Spad, bias_frag=fr), iters=20)
        tb1 = timeit(lambda: hip.attn_bwd(q, k, v, 3 * H, dout, out, lse, B, S, heads, 0.125, bias, biasT, None, Spad,
                                          want_dbias=True, bias_frag=fr), iters=20)
        print("B=%d S=%d %-9s: fwd %.4f ms (%.0f TF, %.2f TB/s algorithmic)  bwd %.4f ms (%.0f TF)  bwd+dbias %.4f ms" % (
            B, S, "resident" if res else "streaming", tf, fl / tf / 1e9, algo_bytes / tf / 1e9, tb0, 2.5 * fl / tb0 / 1e9, tb1),
            flush=True)
    hip.lib().op_attn_set_resident(1)
