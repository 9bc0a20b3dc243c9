"""Peak memory / allocator retries of a bench run (GPU box):  python tools/mem_probe.py <batch>"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

b = sys.argv[1] if len(sys.argv) > 1 else "128"
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-profile", "--batch", b]
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
st = torch.cuda.memory_stats()
print("batch", b, "peak alloc GiB %.1f reserved GiB %.1f retries %d ooms %d" % (
    st["allocated_bytes.all.peak"] / 2**30, st["reserved_bytes.all.peak"] / 2**30, st["num_alloc_retries"], st["num_ooms"]))
