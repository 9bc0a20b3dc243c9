# Final state of round 2 (four-wave GEMM flavours, TN transpose reads through asm): every bench mode + kernel trace of the headline.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2final2; mkdir -p $OUT
cd $R
python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_config1.json 2>/dev/null
python bench.py --config 2 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_config2.json 2>/dev/null
python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_config4_448.json 2>/dev/null
python bench.py --config 4 --fp8 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_config4_448_fp8.json 2>/dev/null
python bench.py --config 4 --res 512 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_config4_512.json 2>/dev/null
python bench.py --config 4 --res 512 --fp8 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_config4_512_fp8.json 2>/dev/null
python bench.py --objective pretrain-vl --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_pretrain_vl.json 2>/dev/null
for b in 16 32 64; do python bench.py --steps 6 --warmup 2 --batch $b --no-cpu-baseline --no-profile > $OUT/bench_config3_b$b.json 2>/dev/null; done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_r2f2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2f2 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $(find /tmp/prof_r2f2 -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python $R/tools/trace_summary.py $(find /tmp/prof_r2f2 -name "*kernel_trace.csv" | head -1) $OUT/bench_last_step.json 1 > $OUT/trace_summary.txt 2>&1
cd $R
python - <<'PY'
import json, glob, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r2final2")
for f in sorted(glob.glob(out + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        print("%-34s ms %.1f  samples/s %.2f  b %s  gemm_frac %s fp8 %s" % (os.path.basename(f), d["ms_per_step"], d["value"], d["config"]["per_gpu_batch"],
              ("%.3f" % r["frac"]) if r else "-", (r.get("fp8_gemm") or {}).get("frac") if r else None))
        if "sweep" in d["config"]: print("   sweep", d["config"]["sweep"])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
head -24 $OUT/trace_summary.txt
