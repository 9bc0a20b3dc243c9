"""Micro-benchmarks of the individual HIP kernels at ONE-PEACE-4B shapes (run on the GPU box).

    python tools/bench_ops.py [--out gpurun_out/ops_bench.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/ops_bench.json")
    ap.add_argument("--tokens", type=int, default=64 * 257)
    args = ap.parse_args()
    res = []
    M, H, Fd = args.tokens, 1536, 6144
    bf = dict(dtype=torch.bfloat16, device=DEV)
    x = torch.randn(M, H, **bf)
    xf = torch.randn(M, Fd, **bf)
    wq = [torch.randn(H, H, **bf) * 0.02 for _ in range(3)]
    w0, w1 = torch.randn(Fd, H, **bf) * 0.02, torch.randn(Fd, H, **bf) * 0.02
    w2 = torch.randn(H, Fd, **bf) * 0.02
    bias = torch.randn(H, **bf)
    gamma = torch.rand(H, **bf)

    def rec(name, ms, flops=None, bytes_=None):
        r = dict(name=name, ms=ms)
        if flops:
            r["tflops"] = flops / ms / 1e9
        if bytes_:
            r["gbps"] = bytes_ / ms / 1e6
        res.append(r)
        print(json.dumps(r), flush=True)

    for glds, tile in ((1, 1), (1, 2), (0, 1)):
        hip.lib().op_gemm_set_staging(glds)
        hip.lib().op_gemm_set_tile(tile)
        tag = ("glds" if glds else "reg") + ("_t256" if tile == 2 else "_t128")
        out = torch.empty(M, 3 * H, **bf)
        rec("gemm_qkv_%s" % tag, timeit(lambda: hip.gemm_nt(x, wq, [bias, None, bias], out=out, n_seg=H, N=3 * H)),
            flops=2.0 * M * 3 * H * H)
        outg = torch.empty(M, Fd, **bf)
        rec("gemm_geglu_%s" % tag, timeit(lambda: hip.gemm_nt(x, [w0, w1], out=outg, epilogue=hip.EPI_GEGLU)),
            flops=4.0 * M * Fd * H)
        outr = torch.empty(M, H, **bf)
        rec("gemm_ffn2_resid_%s" % tag,
            timeit(lambda: hip.gemm_nt(xf, [w2], [bias], out=outr, epilogue=hip.EPI_RESID, resid=x, gamma=gamma)),
            flops=2.0 * M * H * Fd)
    hip.lib().op_gemm_set_staging(1)
    hip.lib().op_gemm_set_tile(0)
    # weight-gradient GEMMs: transpose-read TN kernel vs transposes + NT kernel
    from one_peace_amd import ops
    for (Mo, No) in ((3 * H, H), (Fd, H), (H, Fd), (H, H)):
        dy = torch.randn(M, Mo, **bf)
        xx = torch.randn(M, No, **bf)
        rec("wgrad_tn_%dx%d" % (Mo, No), timeit(lambda: hip.gemm_tn(dy, xx)), flops=2.0 * M * Mo * No)
        rec("wgrad_transpose_nt_%dx%d" % (Mo, No), timeit(lambda: hip.gemm_nt(ops._t_pad(dy), [ops._t_pad(xx)])),
            flops=2.0 * M * Mo * No)
    # torch (hipBLASLt) reference point for the same GEMM
    wcat = torch.cat([w0, w1], 0)
    rec("torch_matmul_geglu_shape", timeit(lambda: torch.matmul(x, wcat.t())), flops=4.0 * M * Fd * H)

    ln_w, ln_b = torch.ones(H, **bf), torch.zeros(H, **bf)
    rec("ln_fwd_1536", timeit(lambda: hip.layernorm_fwd(x, ln_w, ln_b)), bytes_=4.0 * M * H)
    lw, lb = torch.ones(Fd, **bf), torch.zeros(Fd, **bf)
    rec("ln_fwd_6144", timeit(lambda: hip.layernorm_fwd(xf, lw, lb)), bytes_=4.0 * M * Fd)
    y, mean, rstd = hip.layernorm_fwd(x, ln_w, ln_b)
    rec("ln_bwd_1536", timeit(lambda: hip.layernorm_bwd(x, x, ln_w, ln_b, mean, rstd)), bytes_=6.0 * M * H)
    rec("transpose_MxF", timeit(lambda: hip.transpose(xf)), bytes_=4.0 * M * Fd)
    rec("geglu_bwd", timeit(lambda: hip.geglu_bwd(xf, xf, xf)), bytes_=10.0 * M * Fd)

    for (B, S, heads) in ((64, 257, 24), (64, 64, 24), (64, 250, 24), (8, 1025, 24)):
        qkv = torch.randn(B * S, 3 * H, **bf)
        Spad = ((S + 63) // 64) * 64
        bias_t = torch.randn(heads, S, Spad, **bf)
        o = torch.empty(B * S, H, **bf)
        rec("attn_fwd_B%d_S%d" % (B, S),
            timeit(lambda: hip.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], 3 * H, B, S, heads, 0.125, bias_t, None,
                                        Spad, out=o)), flops=4.0 * B * heads * S * S * 64)
        q4 = qkv.view(B, S, 3, heads, 64)
        qq, kk, vv = (q4[:, :, i].transpose(1, 2) for i in range(3))
        rec("torch_sdpa_B%d_S%d" % (B, S),
            timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qq, kk, vv, attn_mask=bias_t[None, :, :, :S])),
            flops=4.0 * B * heads * S * S * 64)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
