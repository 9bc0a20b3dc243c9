"""Round 3: resident forward kernel at S = 256 / 257 / 272 with 9-wave workgroups (one 16-query block per wave) against 8- / 6- /
5-wave workgroups whose waves loop over their blocks.  Bit-equality first, then time.   python tools/attn_waves_ab.py [B]"""
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
for S in [int(v) for v in os.environ.get("SS", "256,257,272,250").split(",")]:
    Spad = hip.attn_spad(S)
    qkv = torch.randn(B * S, 3 * H, **bf)
    bias = torch.randn(heads, S, Spad, **bf)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    frag = hip.attn_bias_pack(bias, S)
    fn = lambda: hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias, None, Spad, want_lse=True, bias_frag=frag)
    hip.TUNE.attn_waves = 10  # one block per wave (the round-2 shape)
    ref, lse = fn()
    ref, lse = ref.clone(), lse.clone()
    line = []
    for nw in (10, 0, 8, 6, 5, 4):
        hip.TUNE.attn_waves = nw
        o, l = fn()
        same = torch.equal(o, ref) and torch.equal(l[:, :, :S], lse[:, :, :S])
        t = min(timeit(fn, iters=20) for _ in range(3))
        line.append("%s %.4f%s" % ("auto" if nw == 0 else "<=%d" % nw, t, "" if same else " (DIFFERENT)"))
    hip.TUNE.attn_waves = 0
    print("B=%d S=%d forward ms by waves per workgroup: " % (B, S) + "  ".join(line), flush=True)
