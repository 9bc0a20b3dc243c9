#!/bin/bash
# HBM traffic of the GEMM family over a bench.py run, per launch (run on the GPU box):
#   tools/pmc_bench_traffic.sh <out.json>
# (round 2: counters collected for the GEMM kernels only -- --kernel-include-regex -- which cuts a pass from minutes to seconds)
# Two separate --pmc passes (FETCH_SIZE, WRITE_SIZE do not fit one pass), kernel-trace only, plus the two calibration runs.
out=$1; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c /tmp/pmcc0_$c /tmp/pmcc1_$c
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm|splitk" --output-format csv -d /tmp/pmcb_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-profile --no-cpu-baseline --no-skip-leg --no-power-probe > /dev/null 2>&1
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
        key = (fam, "gtn" if "gemm256w_tn_grouped" in k else "wtn" if "gemm256w_tn" in k else ("tn" if "gemm256_tn" in k else ("v" if "gemm256v" in k else ("p" if "gemm256p" in k else (
            "w" if "gemm256w" in k else ("b" if "gemm256b" in k else ("a" if "gemm256_kernel" in k else ("s" if "gemm_nt" in k else "-"))))))))
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
json.dump(res, open(sys.argv[1].replace(".json", "_raw.json"), "w"), indent=1)
# processed form (what bench.py's roofline.traffic quotes): counter unit KB; FETCH_SIZE under-reports 16 B/lane reads (guide: x2),
# corrected by the calibration launch that reads a 404 MB matrix exactly once; WRITE_SIZE is exact
fc = res["calib_true_read_bytes"] / (res["FETCH_SIZE"]["calib_bk64_kb_per_launch"] * 1024.0)
wc = res["calib_true_write_bytes"] / (res["WRITE_SIZE"]["calib_bk64_kb_per_launch"] * 1024.0)
n = res["FETCH_SIZE"]["bench_gemm_launches"]
rd = res["FETCH_SIZE"]["bench_gemm_sum_kb"] * 1024.0 * fc / n
wr = res["WRITE_SIZE"]["bench_gemm_sum_kb"] * 1024.0 * wc / n
out = {
    "what": "HBM-side bytes per GEMM-family launch (all gemm* kernels; split-K folds are not counted as launches), one bench.py step",
    "command": "tools/pmc_bench_traffic.sh: rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --kernel-include-regex 'gemm|splitk' -- "
               "python bench.py --steps 1 --warmup 1 --no-profile --no-cpu-baseline --no-skip-leg --no-power-probe (two separate passes + two calibration launches each)",
    "round": 6, "config": 3, "per_gpu_batch": 128, "n_gpus": 1, "gemm_launches": n,
    "fetch_size_correction": fc, "write_size_correction": wc,
    "calibration": "tools/pmc_calib.py: A[32896,6144] bf16 read exactly once (404 MB > 256 MB Infinity Cache)",
    "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr,
    "by_kernel_read_bytes_per_launch": {k: v[0] * 1024.0 * fc / v[1] for k, v in res["FETCH_SIZE"]["bench_by_kernel_kb"].items()},
    "by_kernel_write_bytes_per_launch": {k: v[0] * 1024.0 * wc / v[1] for k, v in res["WRITE_SIZE"]["bench_by_kernel_kb"].items()},
    "by_kernel_launches": {k: v[1] for k, v in res["FETCH_SIZE"]["bench_by_kernel_kb"].items()},
    "kernel_keys": "s = gemm_nt_kernel (128x128), a / b = gemm256 / gemm256b_kernel, v = gemm256v_kernel (four waves, K > 2048), "
                   "p = gemm256p_kernel (persistent: grouped launches and, since round 5, single problems with K <= 2048), tn / wtn = gemm256_tn_kernel / "
                   "gemm256w_tn_kernel, gtn = gemm256w_tn_grouped_kernel (all weight gradients of a layer; round 5: wave queues)",
    "note": "counter bytes include Infinity-Cache hits (MI355X_MICROARCH.md); weight panels re-fetched per M-tile group are mostly such hits",
}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
