"""Round 4: the weight gradients of one lock-step layer at the headline batch -- eight launches with split-K slabs and folds (round 3)
against ONE grouped persistent launch without split-K (op_gemm_tn_grouped), same process, interleaved.

    python tools/wgrad_grouped_bench.py [--nwg 256,248,...] [--subsets]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nwg", default="0")
ap.add_argument("--subsets", action="store_true")
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib()
H, F = 1536, 6144
rows = {"all": 73088, "img": 32896, "aud": 32000, "txt": 8192}
names = [("q|k|v", "all", 3 * H, H), ("out-proj", "all", H, H)]
for m in ("img", "aud", "txt"):
    names += [("%s wi_0|wi_1" % m, m, 2 * F, H), ("%s wo" % m, m, H, F)]
probs = []
for nm, m, o, i in names:
    probs.append((torch.randn(rows[m], o, **bf), torch.randn(rows[m], i, **bf), torch.zeros(o, i, **bf), True))
flops = sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for a, b, _, _ in probs)


def separate(ps=probs):
    for a, b, c, _ in ps:
        hip.gemm_tn(a, b, c, True)


def grouped(ps=probs, tune=0):
    assert hip.gemm_tn_grouped(ps, tune=tune)


for rnd in range(2):
    t_sep = timeit(separate, iters=args.iters)
    print("round %d  separate (split-K + folds): %.3f ms  %.0f TF/s" % (rnd, t_sep, flops / t_sep / 1e9), flush=True)
    for nwg in [int(v) for v in args.nwg.split(",")]:
        t = timeit(lambda: grouped(tune=nwg), iters=args.iters)
        print("round %d  grouped nwg=%-4d             : %.3f ms  %.0f TF/s  (%+.1f %%)" % (rnd, nwg, t, flops / t / 1e9, 100.0 * (t / t_sep - 1)), flush=True)
print("per problem, separate launches:")
for (nm, *_), q in zip(names, probs):
    t = timeit(lambda: hip.gemm_tn(q[0], q[1], q[2], True), iters=args.iters)
    f = 2.0 * q[0].shape[0] * q[0].shape[1] * q[1].shape[1]
    print("   %-18s K=%6d  %.3f ms  %.0f TF/s" % (nm, q[0].shape[0], t, f / t / 1e9), flush=True)
if args.subsets:
    for label, idx in (("attention only", [0, 1]), ("FFNs only", [2, 3, 4, 5, 6, 7]), ("image FFN only", [2, 3]), ("all but text", [0, 1, 2, 3, 4, 5])):
        ps = [probs[i] for i in idx]
        f = sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for a, b, _, _ in ps)
        ts, tg = timeit(lambda: separate(ps), iters=args.iters), timeit(lambda: grouped(ps), iters=args.iters)
        print("%-16s separate %.3f ms (%.0f TF/s)   grouped %.3f ms (%.0f TF/s)" % (label, ts, f / ts / 1e9, tg, f / tg / 1e9), flush=True)
