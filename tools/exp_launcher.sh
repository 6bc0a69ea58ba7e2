#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/exp_launcher.log
: > $O
A="--steps 30 --warmup 5 --no-extras --no-cpu-baseline"
pick() { python3 -c "import json,sys; r=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], r['value'], r['ms_per_step'], r.get('value_stress'))" $1 "$2" >> $O 2>&1; }
python3 bench.py $A > gpurun_out/l0.json 2>/dev/null; pick gpurun_out/l0.json "standalone"
OMP_NUM_THREADS=1 python3 bench.py $A > gpurun_out/l1.json 2>/dev/null; pick gpurun_out/l1.json "standalone OMP_NUM_THREADS=1"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $A --no-dist-init > gpurun_out/l2.json 2>/dev/null; pick gpurun_out/l2.json "torchrun, no init_process_group"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 $A > gpurun_out/l3.json 2>/dev/null; pick gpurun_out/l3.json "torchrun + RCCL"
OMP_NUM_THREADS=8 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 $A > gpurun_out/l4.json 2>/dev/null; pick gpurun_out/l4.json "torchrun + RCCL, OMP_NUM_THREADS=8"
cat $O
