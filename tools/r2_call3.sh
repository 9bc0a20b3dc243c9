mkdir -p gpurun_out/r2c3
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > gpurun_out/r2c3/pytest_attn.txt 2>&1
tail -15 gpurun_out/r2c3/pytest_attn.txt
timeout 300 python tools/attn_abl.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c3/attn_abl.txt
cat gpurun_out/r2c3/attn_abl.txt
timeout 300 python tools/attn_bench.py 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c3/attn_bench_b128.txt
cat gpurun_out/r2c3/attn_bench_b128.txt
