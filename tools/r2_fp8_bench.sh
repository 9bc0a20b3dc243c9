mkdir -p gpurun_out/r2cfg
timeout 600 python bench.py --config 4 --fp8 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2cfg/bench_config4_fp8.json 2> gpurun_out/r2cfg/bench_config4_fp8.err
echo rc=$?; tail -3 gpurun_out/r2cfg/bench_config4_fp8.err
python - <<'PY'
import json
for f in ("bench_config4", "bench_config4_fp8"):
    try:
        d = json.loads(open("gpurun_out/r2cfg/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "ms/step %.1f" % d["ms_per_step"], "samples/s %.2f" % d["value"], "loss", d["config"]["final_loss"], "roofline", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["roofline"].items() if k in ("achieved", "frac", "fp8_gemm", "gemm_share_of_step")})
    except Exception as e:
        print(f, "ERR", e)
PY
