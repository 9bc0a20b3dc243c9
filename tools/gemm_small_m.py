"""Small-M GEMMs (batch 1-8 feature extraction: M = 257 ... 2056) with COLD weights, timed under hipGraph replay so the
host launch path is out of the picture: 40 launches over 40 different weight tensors per replay (one forward of the 4B
encoder touches each weight once; 40 x >= 4.7 MB is re-streamed from HBM every replay).

Separates the fixed cost of a launch (K = 64: two K-steps) from the per-K-step cost, and compares the planner's choice
(tile + split-K + fold) with forced tiles, the unsplit kernel and hipBLASLt (torch.matmul).

    python tools/gemm_small_m.py [M ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

COPIES = 40
SPLIT_SCAN = bool(os.environ.get("SPLIT_SCAN"))  # also time forced K-split counts (op_gemm_set_tile(60 + s))


def replay_time(fns, reps=5):
    """fns: list of zero-arg launchers; returns microseconds per launch under graph replay."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for f in fns:
            f()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / len(fns) * 1e3


def main():
    Ms = [int(v) for v in sys.argv[1:]] or [257, 2056]
    bf = dict(dtype=torch.bfloat16, device="cuda")
    lib = hip.lib()
    shapes = [("k-scan", 1536, 64), ("k-scan", 1536, 256), ("k-scan", 1536, 512), ("out-proj", 1536, 1536),
              ("q|k|v", 4608, 1536), ("w2", 1536, 6144)]
    for M in Ms:
        for name, N, K in shapes:
            x = torch.randn(M, K, **bf)
            ws = [torch.randn(N, K, **bf) * 0.02 for _ in range(COPIES)]
            bias = torch.randn(N, **bf)
            out = torch.empty(M, N, **bf)
            row = {}
            for label, tile, split in (("auto", 0, True), ("unsplit", 0, False), ("128 unsplit", 1, False),
                                       ("256 unsplit", 2, False)):
                lib.op_gemm_set_tile(tile)
                row[label] = replay_time([lambda w=w: hip.gemm_nt(x, [w], [bias], out=out, splitk=split) for w in ws])
            lib.op_gemm_set_tile(0)
            if SPLIT_SCAN and K >= 1536:
                for sp in (2, 3, 4, 6, 8):
                    lib.op_gemm_set_tile(60 + sp)
                    row["s%d" % sp] = replay_time([lambda w=w: hip.gemm_nt(x, [w], [bias], out=out) for w in ws])
                lib.op_gemm_set_tile(60)
            row["hipblaslt"] = replay_time([lambda w=w: torch.addmm(bias, x, w.t(), out=out) for w in ws])
            floor = max(2.0 * N * K / 8e12, 2.0 * M * N * K / 2.5e15) * 1e6
            print("M=%5d %-8s N=%5d K=%5d | %s | floor %.1f us" % (
                M, name, N, K, "  ".join("%s %.1f" % kv for kv in row.items()), floor), flush=True)
            del ws


if __name__ == "__main__":
    main()
