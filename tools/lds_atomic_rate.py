"""LDS atomic throughput on this GPU (libonepeace_probe.so: op_probe_lds_atomic): cycles per wave-level instruction of a CU with 8 waves
issuing conflict-free ds_add_f32 / ds_add_u32 / ds_write_b32.   python tools/lds_atomic_rate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

P = hip.probe_lib()
dev = torch.device("cuda")
wgs, iters = 256, 2000
out = torch.empty(wgs * 512, dtype=torch.float32, device=dev)
clk = torch.zeros(wgs, dtype=torch.int64, device=dev)
print("# %d workgroups x 8 waves, %d x 16 wave-level operations per wave, conflict-free (lane = bank)" % (wgs, iters))
for mode, name in ((2, "ds_write_b32"), (1, "ds_add_u32"), (0, "ds_add_f32")):
    for rep in range(2):
        hip._check_probe(P.op_probe_lds_atomic(hip.ptr(out), hip.ptr(clk), wgs, iters, mode, hip.stream()), "op_probe_lds_atomic")
        torch.cuda.synchronize()
    c = clk.double().cpu()
    per = c / (iters * 16 * 8)
    print("%-14s %8.1f shader cycles per wave-level instruction of the CU (min %.1f max %.1f) = %.2f lanes per cycle" % (
        name, float(per.mean()), float(per.min()), float(per.max()), 64.0 / float(per.mean())))
