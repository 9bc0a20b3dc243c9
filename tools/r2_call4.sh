mkdir -p gpurun_out/r2c4
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r2c4/pytest_gpu.txt 2>&1
tail -8 gpurun_out/r2c4/pytest_gpu.txt
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2c4/bench.json 2> gpurun_out/r2c4/bench.err
tail -c 900 gpurun_out/r2c4/bench.json
