"""Round 3: merged dQ + dBias kernel, batch chunks chosen by round quantisation vs round 2's ">= 768 workgroups" rule.
Kernel launches only (slab buffers allocated and zeroed outside the timed region).   python tools/attn_chunks_ab.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

bf = dict(dtype=torch.bfloat16, device="cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
heads, H = 24, 1536
torch.manual_seed(0)
L = hip.lib()
for S in (257, 250, 65):
    Spad = hip.attn_spad(S)
    qkv = torch.randn(B * S, 3 * H, **bf)
    bias = torch.randn(heads, S, Spad, **bf)
    biasT = torch.randn(heads, S, Spad, **bf)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    frag = hip.attn_bias_pack(bias, S)
    out, lse = hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias, None, Spad, want_lse=True, bias_frag=frag)
    dout = torch.randn_like(out)
    dqkv = torch.empty(B * S, 3 * H, **bf)
    delta = torch.empty(B, heads, Spad, dtype=torch.float32, device="cuda")
    res, ref = {}, None
    forced = [int(v) for v in os.environ.get("CHUNKS", "").split(",") if v]
    for r2 in ([1, 0, 1, 0] if not forced else [-c for c in forced] * 2):
        hip.TUNE.dbias_chunks_r2 = 1 if r2 == 1 else 0
        hip.TUNE.dbias_chunks = -r2 if r2 < 0 else 0
        n = L.op_attn_bwd_dbias_slabs(B, S, heads, hip.TUNE.attn_bwd())
        slabs = torch.zeros(n, heads, S, Spad, dtype=torch.float32, device="cuda")
        fn = lambda: hip.attn_bwd_launch(q, k, v, 3 * H, dout, bias, biasT, None, lse, delta, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:],
                                         3 * H, slabs, B, S, Spad, heads, 0.125, frag, out=out)
        fn()
        if ref is None:
            ref = dqkv.clone()
        assert torch.equal(dqkv, ref)
        key = "round-2 rule (%d chunks)" % n if r2 == 1 else ("library rule (%d chunks)" % n if r2 == 0 else "%d chunks" % n)
        res[key] = min(res.get(key, 1e9), timeit(fn, iters=20))
    hip.TUNE.dbias_chunks_r2 = hip.TUNE.dbias_chunks = 0
    print("B=%d S=%d dK/dV + dQ/dBias kernels: " % (B, S) + "   ".join("%s %.4f ms" % kv for kv in res.items()) + "   (dq/dk/dv identical)", flush=True)
