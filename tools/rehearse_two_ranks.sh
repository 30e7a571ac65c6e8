#!/bin/bash
# The N-rank flow of bench.py rehearsed with TWO ranks on ONE GPU over gloo (HSTU_DIST_BACKEND=gloo): every collective, barrier and
# rank-0-only section of the real multi-GPU run, without a second device.  A watchdog dumps all stacks if it stalls.
mkdir -p gpurun_out/r05_rehearse
export HSTU_DIST_BACKEND=gloo HSTU_BENCH_WATCHDOG=${WATCHDOG:-120}
timeout ${LIMIT:-300} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 10 --warmup 3 --users-per-gpu 2048 $* > gpurun_out/r05_rehearse/out.json 2> gpurun_out/r05_rehearse/err.txt
echo rc=$?
grep -n "File \"/root/repo\|File \".*bench.py\|Thread 0x\|Current thread\|most recent call first" gpurun_out/r05_rehearse/err.txt | head -60 | cut -c1-220
tail -1 gpurun_out/r05_rehearse/out.json | cut -c1-300
