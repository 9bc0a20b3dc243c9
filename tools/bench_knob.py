"""Run bench.py with op_gemm_set_tile knobs applied first (A/B inside one GPU session):  python tools/bench_knob.py 44 -- --steps 4"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

sep = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
for m in sys.argv[1:sep]:
    hip.lib().op_gemm_set_tile(int(m))
sys.argv = ["bench.py"] + sys.argv[sep + 1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
