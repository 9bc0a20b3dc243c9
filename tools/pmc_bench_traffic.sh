#!/bin/bash
# HBM traffic of the GEMM family over a bench.py run, per launch (run on the GPU box):
#   tools/pmc_bench_traffic.sh <out.json>
# (round 2: counters collected for the GEMM kernels only -- --kernel-include-regex -- which cuts a pass from minutes to seconds)
# Two separate --pmc passes (FETCH_SIZE, WRITE_SIZE do not fit one pass), kernel-trace only, plus the two calibration runs.
out=$1; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c /tmp/pmcc0_$c /tmp/pmcc1_$c
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm|splitk" --output-format csv -d /tmp/pmcb_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-profile --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm|splitk" --output-format csv -d /tmp/pmcc0_$c -o p -- python $R/tools/pmc_calib.py 0 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm|splitk" --output-format csv -d /tmp/pmcc1_$c -o p -- python $R/tools/pmc_calib.py 1 > /dev/null 2>&1
done
python - "$out" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = "gemm" if "gemm" in k else "other"
        key = (fam, "wtn" if "gemm256w_tn" in k else ("tn" if "gemm256_tn" in k else ("w" if "gemm256w" in k else ("b" if "gemm256b" in k else ("a" if "gemm256_kernel" in k else ("s" if "gemm_nt" in k else "-"))))))
        agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
    return agg
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    b = load("/tmp/pmcb_" + c); c0 = load("/tmp/pmcc0_" + c); c1 = load("/tmp/pmcc1_" + c)
    res[c] = {
        "bench_gemm_sum_kb": sum(v[0] for k, v in b.items() if k[0] == "gemm"),
        "bench_gemm_launches": sum(v[1] for k, v in b.items() if k[0] == "gemm"),
        "bench_by_kernel_kb": {k[1]: [v[0], v[1]] for k, v in b.items() if k[0] == "gemm"},
        "calib_bk32_kb_per_launch": sum(v[0] for k, v in c0.items() if k[0] == "gemm") / max(1, sum(v[1] for k, v in c0.items() if k[0] == "gemm")),
        "calib_bk64_kb_per_launch": sum(v[0] for k, v in c1.items() if k[0] == "gemm") / max(1, sum(v[1] for k, v in c1.items() if k[0] == "gemm")),
    }
res["calib_true_read_bytes"] = 32896 * 6144 * 2 + 256 * 6144 * 2
res["calib_true_write_bytes"] = 32896 * 256 * 2
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps(res, indent=1))
PY
