#!/bin/bash
# the N > 1 harness with one rank (RCCL communicator of one rank), the sharded GPU tests, and the per-rank slab kernel times
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shard or ranks or rccl or sharded or two_ranks or base_class_with" 2>&1 | tail -4
PVI_FORCE_PARALLEL=1 timeout 900 python bench.py --gpus 1 > gpurun_out/r04_bench_world1.json 2> gpurun_out/r04_bench_world1.err; echo "rc=$?"
wc -c gpurun_out/r04_bench_world1.json; cat gpurun_out/r04_bench_world1.json; grep -v "full record" gpurun_out/r04_bench_world1.err | tail -5
PVI_FORCE_PARALLEL=1 timeout 900 python bench.py --gpus 1 --workload c4 > gpurun_out/r04_bench_world1_c4.json 2> gpurun_out/r04_bench_world1_c4.err; echo "rc=$?"
cat gpurun_out/r04_bench_world1_c4.json
MASTER_ADDR=127.0.0.1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 > gpurun_out/r04_bench_torchrun1.json 2> gpurun_out/r04_bench_torchrun1.err; echo "rc=$?"; tail -c 400 gpurun_out/r04_bench_torchrun1.json
L=gpurun_out/r04_slab_times.log; : > $L
timeout 600 python tools/slab_time.py c4 8 0 3 >> $L 2>&1
timeout 600 python tools/slab_time.py c4 4 1 >> $L 2>&1
timeout 600 python tools/slab_time.py c4 2 0 >> $L 2>&1
timeout 600 python tools/slab_time.py c4 1 0 >> $L 2>&1
timeout 600 python tools/slab_time.py c3w 8 3 >> $L 2>&1
cat $L
