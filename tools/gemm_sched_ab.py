"""Round 3: the instruction schedules of gemm256v_kernel (tune bits 20-22 = 1..5) against gemm256w_kernel (7), the eight-wave
full-line kernel and hipBLASLt (torch.matmul), same process, interleaved rounds, random N(0,1) data.

    python tools/gemm_sched_ab.py [out.json]        MS=32768,32896  ITERS=40  ROUNDS=3

Every schedule must be bit-identical to the eight-wave kernel on every epilogue (checked before it is timed).
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

bf = dict(dtype=torch.bfloat16, device="cuda")
IT = int(os.environ.get("ITERS", "40"))
ROUNDS = int(os.environ.get("ROUNDS", "3"))
SCHEDS = [int(v) for v in os.environ.get("SCHEDS", "1,3,6").split(",")]
torch.manual_seed(0)
hip.lib()
T = hip.TUNE
H, F = 1536, 6144
results = []


def flavour(kind):
    """kind: 'e' eight waves, 'w' gemm256w, int n gemm256v schedule n."""
    T.reset()
    T.tile_mode = 2
    if kind == "e":
        T.fullline = 1
    else:
        T.fullline = 3
        T.sched = 7 if kind == "w" else int(kind)


def as_tuple(v):
    return v if isinstance(v, (tuple, list)) else (v,)


def case(name, M, N, K, fn, flops, blas=None):
    flavour("e")
    ref = [t.clone() for t in as_tuple(fn())]
    kinds = ["e", "w"] + SCHEDS
    same = {}
    for kd in kinds[1:]:
        flavour(kd)
        out = as_tuple(fn())
        torch.cuda.synchronize()
        same[str(kd)] = all(torch.equal(a, b) for a, b in zip(ref, out))
    best = {str(kd): 1e9 for kd in kinds}
    for _ in range(ROUNDS):
        for kd in kinds:
            flavour(kd)
            best[str(kd)] = min(best[str(kd)], timeit(fn, iters=IT, warmup=5))
        if blas is not None:
            best["blas"] = min(best.get("blas", 1e9), timeit(blas, iters=IT, warmup=5))
    T.reset()
    row = dict(name=name, M=M, N=N, K=K, ms=best, tflops={k: flops / v / 1e9 for k, v in best.items()}, bit_identical=same)
    results.append(row)
    print("%-30s M=%6d N=%5d K=%5d | " % (name, M, N, K) + "  ".join("%s %.4f (%4.0f)" % (k, v, flops / v / 1e9) for k, v in best.items())
          + " | identical: " + ("all" if all(same.values()) else str(same)), flush=True)


for M in [int(v) for v in os.environ.get("MS", "32768,32896").split(",")]:
    x = torch.randn(M, H, **bf)
    xf = torch.randn(M, F, **bf)
    x3 = torch.randn(M, 3 * H, **bf)
    wqkv = [torch.randn(H, H, **bf) * 0.03 for _ in range(3)]
    wcat = torch.cat(wqkv, 0)
    bq = [torch.randn(H, **bf), None, torch.randn(H, **bf)]
    w0, w1 = torch.randn(F, H, **bf) * 0.03, torch.randn(F, H, **bf) * 0.03
    w01 = torch.cat([w0, w1], 0)
    w2 = torch.randn(H, F, **bf) * 0.02
    w2t = torch.randn(F, H, **bf) * 0.02
    w3t = torch.randn(H, 3 * H, **bf) * 0.02
    b2, gamma = torch.randn(H, **bf), torch.randn(H, **bf)
    res = torch.randn(M, H, **bf)
    ps = torch.rand(M // 2 + 1, device="cuda")
    h0, h1, y = torch.empty(M, F, **bf), torch.empty(M, F, **bf), torch.empty(M, H, **bf)
    o_qkv, o_h, o_f = torch.empty(M, 3 * H, **bf), torch.empty(M, H, **bf), torch.empty(M, F, **bf)
    case("qkv (3 segments, bias)", M, 3 * H, H, lambda: hip.gemm_nt(x, wqkv, bq, n_seg=H, N=3 * H, out=o_qkv), 2.0 * M * 3 * H * H,
         lambda: torch.matmul(x, wcat.t(), out=o_qkv))
    case("out-proj + residual K=1536", M, H, H,
         lambda: (hip.gemm_nt(x, [wqkv[0]], [b2], epilogue=hip.EPI_RESID, resid=res, gamma=gamma, rowscale=ps, rows_per_sample=2,
                              h0=y, out=o_h), y), 2.0 * M * H * H, lambda: torch.matmul(x, wqkv[0].t(), out=o_h))
    case("GeGLU up-projection", M, F, H, lambda: (hip.gemm_nt(x, [w0, w1], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1, out=o_f), h0, h1),
         4.0 * M * F * H, lambda: torch.matmul(x, w01.t()))
    case("down-proj + residual K=6144", M, H, F,
         lambda: hip.gemm_nt(xf, [w2], [b2], epilogue=hip.EPI_RESID, resid=res, gamma=gamma, rowscale=ps, rows_per_sample=2, h0=y,
                             out=o_h), 2.0 * M * H * F, lambda: torch.matmul(xf, w2.t(), out=o_h))
    case("dgrad N=1536 K=6144", M, H, F, lambda: hip.gemm_nt(xf, [w2], out=o_h, splitk=False), 2.0 * M * H * F,
         lambda: torch.matmul(xf, w2.t(), out=o_h))
    case("dgrad N=6144 K=1536", M, F, H, lambda: hip.gemm_nt(x, [w2t], out=o_f, splitk=False), 2.0 * M * F * H,
         lambda: torch.matmul(x, w2t.t(), out=o_f))
    case("dgrad N=1536 K=4608", M, H, 3 * H, lambda: hip.gemm_nt(x3, [w3t], out=o_h, splitk=False), 2.0 * M * H * 3 * H,
         lambda: torch.matmul(x3, w3t.t(), out=o_h))
    del x, xf, x3, h0, h1, y, o_qkv, o_h, o_f
    torch.cuda.empty_cache()

if len(sys.argv) > 1:
    os.makedirs(os.path.dirname(sys.argv[1]) or ".", exist_ok=True)
    json.dump(results, open(sys.argv[1], "w"), indent=1)
