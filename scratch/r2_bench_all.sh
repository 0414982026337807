#!/bin/bash
# round 2: the bench line, the C3 workload, the in-process multi-GPU path with logical shards, and a 2-rank rehearsal
set -o pipefail
out=gpurun_out/r2_bench
mkdir -p $out
timeout -k 10 300 python bench.py > $out/c2.json 2> $out/c2.err || exit 1
timeout -k 10 300 python bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline > $out/c3.json 2> $out/c3.err || exit 1
timeout -k 10 300 python bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --node-shards 3 > $out/c3_node3.json 2> $out/c3_node3.err || exit 1
SW_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --workload c3 --targets 200000 --no-cpu-baseline > $out/c3_rehearsal2.json 2> $out/c3_rehearsal2.err || exit 1
SW_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 2 --warmup 1 --targets 50000 --no-cpu-baseline > $out/c2_rehearsal2.json 2> $out/c2_rehearsal2.err || exit 1
for f in c2 c3 c3_node3 c3_rehearsal2 c2_rehearsal2; do echo "== $f"; tail -1 $out/$f.json | cut -c1-700; done
