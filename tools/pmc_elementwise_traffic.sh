#!/bin/bash
# HBM-side bytes and durations of the elementwise kernels (LayerNorm family, GeGLU passes, residual backward, AdamW) over ONE headline step:
#   tools/pmc_elementwise_traffic.sh <out.txt>        (run on the GPU box; two --pmc passes + the kernel trace of each)
# Counter unit KB; FETCH_SIZE is corrected by the factor tools/pmc_calib.py measured for 16-byte-per-lane reads (x 1.90, profiles/r4_gemm_hbm_traffic.json).
out=$1; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmce_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "ln_|resid_bwd|adamw|colsum" --output-format csv -d /tmp/pmce_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-profile --no-cpu-baseline --no-skip-leg --no-power-probe > /dev/null 2>&1
done
python - "$out" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        m = re.search(r"(ln_geglu_bwd|ln_geglu_fwd|ln_bwd|ln_fwd|resid_bwd|adamw_groups|colsum_partial)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][0] += float(r["Counter_Value"]); agg[m.group(1)][1] += 1
    return agg
def durations(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    agg = defaultdict(float)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(ln_geglu_bwd|ln_geglu_fwd|ln_bwd|ln_fwd|resid_bwd|adamw_groups|colsum_partial)", r["Kernel_Name"])
        if m:
            agg[m.group(1)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return agg
rd, wr, ms = load("/tmp/pmce_FETCH_SIZE", "FETCH_SIZE"), load("/tmp/pmce_WRITE_SIZE", "WRITE_SIZE"), durations("/tmp/pmce_FETCH_SIZE")
with open(sys.argv[1], "w") as o:
    o.write("# both steps of the run (warm-up + timed) together; bytes = counter KB x 1024 (reads x 1.90: 16-byte-per-lane correction)\n")
    for k in sorted(ms, key=lambda k: -ms[k]):
        rb, wb = rd[k][0] * 1024 * 1.90, wr[k][0] * 1024
        o.write("%-16s launches %4d  %8.2f ms  read %7.1f GB  written %7.1f GB  -> %.2f TB/s of counter bytes\n" % (
            k, rd[k][1], ms[k], rb / 1e9, wb / 1e9, (rb + wb) / (ms[k] * 1e-3) / 1e12))
print(open(sys.argv[1]).read())
PY
