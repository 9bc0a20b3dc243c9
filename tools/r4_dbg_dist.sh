#!/bin/bash
# debug: where do the two ranks of the one-device gloo step hang?  (SIGABRT after 100 s -> faulthandler prints every thread's stack)
O=gpurun_out/r4dbg; mkdir -p $O
export ONEPEACE_DIST_BACKEND=gloo ONEPEACE_SINGLE_DEVICE_DEBUG=1 HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT ${1:-100} python bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 --layers 3 --check-replicas --no-profile > $O/rank$r.out 2> $O/rank$r.err &
done
wait
for r in 0 1; do echo "== rank $r"; tail -5 $O/rank$r.out; grep -v "^  File.*site-packages\|^  File.*dist-packages" $O/rank$r.err | tail -60; done
