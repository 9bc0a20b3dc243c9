"""Peak memory / allocator retries of a bench run: python tools/mem_probe.py <side 0|1> <batch>"""
import os, runpy, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_peace_amd import ops
ops.USE_WGRAD_STREAM = bool(int(sys.argv[1]))
b = sys.argv[2]
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-profile", "--batch", b]
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
st = torch.cuda.memory_stats()
print("side", ops.USE_WGRAD_STREAM, "batch", b, "peak alloc GiB %.1f reserved GiB %.1f retries %d ooms %d" % (
    st["allocated_bytes.all.peak"] / 2**30, st["reserved_bytes.all.peak"] / 2**30, st["num_alloc_retries"], st["num_ooms"]))
