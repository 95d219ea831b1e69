#!/bin/bash
# the N > 1 harness with one rank (RCCL communicator of one rank): weak-scaling headline and the C4 strong-scaling line
cd /root/repo; mkdir -p gpurun_out
PVI_FORCE_PARALLEL=1 timeout 900 python bench.py --gpus 1 > gpurun_out/r03_bench_world1.json 2> gpurun_out/r03_bench_world1.err; echo "rc=$?"
tail -c 1500 gpurun_out/r03_bench_world1.json; tail -5 gpurun_out/r03_bench_world1.err
PVI_FORCE_PARALLEL=1 timeout 900 python bench.py --gpus 1 --workload c4 > gpurun_out/r03_bench_world1_c4.json 2> gpurun_out/r03_bench_world1_c4.err; echo "rc=$?"
tail -c 1500 gpurun_out/r03_bench_world1_c4.json; tail -5 gpurun_out/r03_bench_world1_c4.err
MASTER_ADDR=127.0.0.1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 > gpurun_out/r03_bench_torchrun1.json 2> gpurun_out/r03_bench_torchrun1.err; echo "rc=$?"; tail -c 300 gpurun_out/r03_bench_torchrun1.json
