"""Round 4: persistent attention forward kernel against the resident one, same process, interleaved (B = 128 by default).

    python tools/attn_pers_ab.py [B]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

H, heads = 1536, 24
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bf = dict(dtype=torch.bfloat16, device="cuda")
for S, use_pad in ((257, False), (250, True), (256, False), (197, False)):
    Spad = hip.attn_spad(S)
    qkv = torch.randn(B * S, 3 * H, **bf)
    bias = torch.randn(heads, S, Spad, **bf)
    pad = None
    if use_pad:
        pad = torch.zeros(B, Spad, dtype=torch.uint8, device="cuda")
        pad[:, S:] = 1
        pad[1::3, S - 20:] = 1
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    fl = 4.0 * B * heads * S * S * 64
    frag = hip.attn_bias_pack(bias, S)
    fn = lambda: hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias, pad, Spad, want_lse=True, bias_frag=frag)  # noqa: E731
    res = {}
    for rnd in range(2):
        for pers in (1, 0):
            hip.TUNE.attn_pers = pers
            res.setdefault(pers, []).append(timeit(fn, iters=30, warmup=5))
    hip.TUNE.attn_pers = 1
    tp, tr = min(res[1]), min(res[0])
    print("B=%d S=%d pad=%d: fwd persistent %.4f ms (%.0f TF/s, %.2f TB/s of q,k,v,out)  resident %.4f ms   %+.1f %%   (runs %s | %s)" % (
        B, S, int(use_pad), tp, fl / tp / 1e9, 8.0 * B * S * H / tp / 1e9, tr, 100.0 * (tp / tr - 1),
        " ".join("%.4f" % x for x in res[1]), " ".join("%.4f" % x for x in res[0])), flush=True)
    # backward (dQ + dBias kernel persistent or rounds 1-3; the dK/dV kernel is the same in both arms)
    biasT = torch.zeros_like(bias)
    biasT[..., :S] = bias[..., :S].transpose(1, 2)
    out, lse = fn()
    dout = torch.randn_like(out)
    dqkv = torch.empty(B * S, 3 * H, **bf)
    dbias = hip.attn_dbias_buffer(B, S, heads, Spad, "cuda")
    delta = torch.empty(B, heads, Spad, dtype=torch.float32, device="cuda")
    bw = lambda: hip.attn_bwd_launch(q, k, v, 3 * H, dout, bias, biasT, pad, lse, delta, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:], 3 * H,  # noqa: E731
                                     dbias, B, S, Spad, heads, 0.125, frag, out=out)
    resb = {}
    for rnd in range(2):
        for pers in (1, 0):
            hip.TUNE.attn_pers_bwd = pers
            resb.setdefault(pers, []).append(timeit(bw, iters=20, warmup=3))
    hip.TUNE.attn_pers_bwd = 1
    tp, tr = min(resb[1]), min(resb[0])
    print("              bwd (dQ + dBias + dK/dV): persistent dQ %.4f ms (%.0f TF/s)  rounds 1-3 %.4f ms   %+.1f %%   (runs %s | %s)" % (
        tp, 2.5 * fl / tp / 1e9, tr, 100.0 * (tp / tr - 1), " ".join("%.4f" % x for x in resb[1]), " ".join("%.4f" % x for x in resb[0])), flush=True)
    # persistent dK / dV kernel (round 4) against the rounds 1-3 kernel, both behind the persistent dQ kernel
    resk, outk = {}, {}
    for rnd in range(2):
        for pk in (1, 0):
            hip.TUNE.attn_pers_dkdv = pk
            resk.setdefault(pk, []).append(timeit(bw, iters=20, warmup=3))
            outk[pk] = dqkv.clone()
    hip.TUNE.attn_pers_dkdv = 1
    same = torch.equal(outk[1].view(B, S, 3 * H)[:, :min(S, 256)], outk[0].view(B, S, 3 * H)[:, :min(S, 256)])
    print("              bwd with the persistent dK / dV kernel %.4f ms, rounds 1-3 dK / dV %.4f ms   %+.1f %%   keys 0..255 bit-identical: %s, max |difference| %.3e" % (
        min(resk[1]), min(resk[0]), 100.0 * (min(resk[1]) / min(resk[0]) - 1), same, float((outk[1].float() - outk[0].float()).abs().max())), flush=True)
    # dK/dV: a trailing block of <= 16 keys split over the waves by queries (round 4) against one wave running it (bit-identical?)
    if S % 128 and S % 128 <= 16:
        resl, outs = {}, {}
        for rnd in range(2):
            for lone in (1, 0):
                hip.TUNE.attn_lone_keys = lone
                resl.setdefault(lone, []).append(timeit(bw, iters=20, warmup=3))
                outs[lone] = dqkv.clone()
        hip.TUNE.attn_lone_keys = 1
        err = float((outs[1].float() - outs[0].float()).abs().max())
        print("              bwd with the lone key block split by queries %.4f ms, on one wave %.4f ms   %+.1f %%   max |difference| of dq|dk|dv %.3e" % (
            min(resl[1]), min(resl[0]), 100.0 * (min(resl[1]) / min(resl[0]) - 1), err), flush=True)
