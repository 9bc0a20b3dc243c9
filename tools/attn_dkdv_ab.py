"""Round 3: dK/dV kernel with 128 keys per workgroup (two 16-key blocks per wave, 243 VGPRs, 2 waves/SIMD) vs 64 keys (one block,
164 VGPRs, 3 waves/SIMD).  python tools/attn_dkdv_ab.py [B]"""
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
hip.lib()
for S in (257, 250, 65, 320, 785):
    Bs = B if S < 400 else B // 4
    Spad = hip.attn_spad(S)
    qkv = torch.randn(Bs * S, 3 * H, **bf)
    bias = torch.randn(heads, S, Spad, **bf)
    biasT = torch.randn(heads, S, Spad, **bf)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    frag = hip.attn_bias_pack(bias, S) if S <= 320 else None
    out, lse = hip.attn_fwd(q, k, v, 3 * H, Bs, S, heads, 0.125, bias, None, Spad, want_lse=True, bias_frag=frag)
    dout = torch.randn_like(out)
    res, ref = {}, None
    for keys in (1, 2, 1, 2):
        hip.TUNE.dkdv_keys = keys
        fn = lambda: hip.attn_bwd(q, k, v, 3 * H, dout, out, lse, Bs, S, heads, 0.125, bias, biasT, None, Spad, want_dbias=True, bias_frag=frag)
        d, _ = fn()
        if ref is None:
            ref = d.clone()
        same = torch.equal(d, ref)
        key = "%d keys/workgroup" % (128 if keys == 1 else 64)
        res[key] = min(res.get(key, 1e9), timeit(fn, iters=20))
        assert same, "dq/dk/dv differ"
    hip.TUNE.dkdv_keys = 0
    print("B=%d S=%d backward + dBias: " % (Bs, S) + "   ".join("%s %.4f ms" % kv for kv in res.items()) + "   (identical)", flush=True)
