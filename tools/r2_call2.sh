mkdir -p gpurun_out/r2c2
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > gpurun_out/r2c2/pytest_attn.txt 2>&1
tail -5 gpurun_out/r2c2/pytest_attn.txt
timeout 300 python tools/attn_bench.py 64 > gpurun_out/r2c2/attn_bench_b64.txt 2>&1
timeout 300 python tools/attn_bench.py 128 > gpurun_out/r2c2/attn_bench_b128.txt 2>&1
cat gpurun_out/r2c2/attn_bench_b64.txt gpurun_out/r2c2/attn_bench_b128.txt
