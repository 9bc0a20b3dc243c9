"""Fused attention backward (op_attn_bwd_fused + op_attn_bwd_delta) against the dQ + dBias / dK + dV kernel pair (op_attn_bwd) at the
headline shapes, interleaved in one process:   python tools/attn_fused_ab.py [B] [iters]   (B = 128, heads = 24)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    heads, dev = 24, torch.device("cuda")
    H = heads * 64
    g = torch.Generator(device=dev).manual_seed(0)
    print("# B = %d, heads = %d; ms per call, median of %d (min); fused = op_attn_bwd_delta + op_attn_bwd_fused" % (B, heads, iters))
    for S, num_rel, use_pad in ((257, 964, False), (250, 1026, True), (250, 1026, False), (200, 1026, True)):
        Spad = hip.attn_spad(S)
        qkv = torch.randn(B * S, 3 * H, generator=g, device=dev).to(torch.bfloat16)
        do = torch.randn(B * S, H, generator=g, device=dev).to(torch.bfloat16)
        table = (0.5 * torch.randn(num_rel, heads, generator=g, device=dev)).to(torch.bfloat16)
        ii = torch.arange(S, device=dev)
        bucket = ((ii[:, None] - ii[None, :]) + 512).clamp(0, num_rel - 1).to(torch.int32).contiguous()
        bias = hip.relpos_bias_build(table, bucket, S, Spad)
        biasT = hip.relpos_bias_build(table, bucket, S, Spad, transposed=True)
        frag = hip.attn_bias_pack(bias, S)
        bpack = hip.attn_bucket_pack(bucket)
        pad = None
        if use_pad:
            pad = torch.ones(B, Spad, dtype=torch.uint8, device=dev)
            pad[:, :S] = 0
            pad[1::3, S - 20:S] = 1
        out, lse = hip.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], 3 * H, B, S, heads, 0.125, bias, pad, Spad, bias_frag=frag)
        dqkv = torch.empty(B * S, 3 * H, dtype=torch.bfloat16, device=dev)
        delta = torch.empty(B, heads, Spad, dtype=torch.float32, device=dev)
        dtable = torch.zeros(num_rel, heads, dtype=torch.float32, device=dev)
        dbias = hip.attn_dbias_buffer(B, S, heads, Spad, dev)

        def fused():
            hip._check(hip.lib().op_attn_bwd_delta(hip.ptr(do), hip.ptr(out), do.stride(0), hip.ptr(delta), B, S, Spad, heads, hip.stream()), "delta")
            assert hip.attn_bwd_fused(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], 3 * H, do, lse, delta, biasT, bpack, pad, dqkv[:, :H],
                                      dqkv[:, H:2 * H], dqkv[:, 2 * H:], 3 * H, dtable, B, S, Spad, heads, 0.125)

        def pair():
            hip.attn_bwd_launch(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], 3 * H, do, bias, biasT, pad, lse, delta, dqkv[:, :H], dqkv[:, H:2 * H],
                                dqkv[:, 2 * H:], 3 * H, dbias, B, S, Spad, heads, 0.125, frag, out=out)
        res = {}
        for name, fn in (("fused", fused), ("pair", pair), ("fused", fused), ("pair", pair)):
            ts = []
            for i in range(iters + 3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                if i >= 3:
                    ts.append(e0.elapsed_time(e1))
            res.setdefault(name, []).extend(ts)
        line = "S = %3d pad %d:" % (S, int(use_pad))
        for name in ("pair", "fused"):
            ts = sorted(res[name])
            line += "   %s %.4f (%.4f)" % (name, ts[len(ts) // 2], ts[0])
        mp, mf = sorted(res["pair"])[len(res["pair"]) // 2], sorted(res["fused"])[len(res["fused"]) // 2]
        print(line + "   fused / pair = %.3f" % (mf / mp))


if __name__ == "__main__":
    main()
