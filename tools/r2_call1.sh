set -x
mkdir -p gpurun_out/r2c1
python tools/attn_bench.py > gpurun_out/r2c1/attn_bench.txt 2>&1
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2c1/bench_prof.json 2> gpurun_out/r2c1/bench_prof.err
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/r2c1/bench_noprof.json 2> gpurun_out/r2c1/bench_noprof.err
tail -c 600 gpurun_out/r2c1/bench_prof.json; tail -c 300 gpurun_out/r2c1/bench_noprof.json; cat gpurun_out/r2c1/attn_bench.txt
