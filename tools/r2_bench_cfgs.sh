mkdir -p gpurun_out/r2cfg
for c in 1 2 4; do
  timeout 600 python bench.py --config $c --steps 4 --warmup 1 > gpurun_out/r2cfg/bench_config$c.json 2> gpurun_out/r2cfg/bench_config$c.err
  echo "config $c rc=$?"; tail -c 1500 gpurun_out/r2cfg/bench_config$c.json | head -c 1500; echo; tail -3 gpurun_out/r2cfg/bench_config$c.err
done
