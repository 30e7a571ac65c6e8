#!/bin/bash
mkdir -p gpurun_out/r3
export HSTU_DIST_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --no-extra --layer-steps 3 --users-per-gpu 2048 > gpurun_out/r3/bench_n2_gloo.json 2> gpurun_out/r3/bench_n2_gloo.err
echo rc=$?
tail -3 gpurun_out/r3/bench_n2_gloo.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_n2_gloo.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')})
print(json.dumps(d.get('rccl'))[:1200])
print(d['layer']['ms_per_step'], d['layer'].get('allreduce_bytes'))
PY
unset HSTU_DIST_BACKEND
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 2 --warmup 1 --no-extra --no-layer --no-cpu 2>&1 | grep -i "RuntimeError\|has no GPU" | head -3
