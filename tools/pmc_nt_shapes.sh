#!/bin/bash
# FETCH_SIZE per launch of the headline step's NT launch shapes against their algorithmic read bytes (run on the GPU box):
#   tools/pmc_nt_shapes.sh <out.txt>      (correction factor of the counter from tools/pmc_calib.py, as tools/pmc_bench_traffic.sh)
out=$1; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
run() {  # name M N K resid segs
  rm -rf /tmp/pmcs
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "gemm" --output-format csv -d /tmp/pmcs -o p -- python $R/tools/nt_shape_run.py $2 $3 $4 $5 $6 > /dev/null 2>&1
  python - "$1" $2 $3 $4 $5 "$CAL" <<'PY'
import csv, glob, sys
name, M, N, K, resid, cal = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == "1", float(sys.argv[6])
f = glob.glob("/tmp/pmcs/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
kb = sum(float(r["Counter_Value"]) for r in rows) / 6.0  # per LOGICAL launch (nt_shape_run.py makes six): a tail-rows split is two kernels
kern = "%d kernels per launch" % (len(rows) // 6)
algo = 2 * (M * K + N * K + (M * N if resid else 0))
got = kb * 1024 * cal
print("%-34s M=%6d N=%5d K=%5d  %-24s read %7.1f MB per launch, algorithmic %7.1f MB: %.2f x" % (name, M, N, K, kern, got / 1e6, algo / 1e6, got / algo))
PY
}
rm -rf /tmp/pmcc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "gemm" --output-format csv -d /tmp/pmcc -o p -- python $R/tools/pmc_calib.py 1 > /dev/null 2>&1
CAL=$(python - <<'PY'
import csv, glob
f = glob.glob("/tmp/pmcc/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
kb = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
print((32896 * 6144 * 2 + 256 * 6144 * 2) / (kb * 1024.0))
PY
)
{
echo "# FETCH_SIZE (x $CAL, calibrated on a launch that reads 404 MB exactly once) per launch, A matrices rotating (cold), round 6 HEAD"
run "q|k|v"                      73088 4608 1536 0 3
run "out-proj + residual"        73088 1536 1536 1 1
run "out-proj input gradient"    73088 1536 1536 0 1
run "q|k|v input gradient"       73088 1536 4608 0 1
run "FFN up-projection (image)"  32896 12288 1536 0 2
run "FFN down input gradient"    32896 6144 1536 0 1
run "FFN down-projection + resid" 32896 1536 6144 1 1
run "FFN up input gradient"      32896 1536 12288 0 1
} > $out
cat $out
