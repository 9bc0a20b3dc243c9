OUT=$GRAFT_REPO_ROOT/gpurun_out/r2prof_vl
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_vl
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vl -o bench -- python $GRAFT_REPO_ROOT/bench.py --objective pretrain-vl --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/bench.json 2> $OUT/bench.err
ST=$(find /tmp/prof_vl -name "*kernel_stats.csv" | head -1)
cp $ST $OUT/kernel_stats.csv
head -25 $OUT/kernel_stats.csv | cut -c1-200
