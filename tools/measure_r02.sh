#!/bin/bash
# Round-2 measurement sweep behind profiles/r02_{bench,configs,heads,control_loop,training_step,scaling_dry}.json and
# r02_train_*_top_kernels.txt (GPU box, one gpurun call, from the repo root).
cd "$(dirname "$0")/.."
O=gpurun_out/measure_r02; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
for a in "--batch 4 --samples 128" "--height 512 --width 512" "--samples 256" "--samples 32"; do
  echo "ARGS $a"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-precisions $a 2>/dev/null | tail -1
done > $O/configs.txt
# the N>1 code path on one GPU: RCCL initialised with one rank, and rank 0's shard of an 8-way strong split
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-precisions --force-dist 2>/dev/null | tail -1 > $O/force_dist.json
for n in 2 4 8; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-precisions --simulate-world $n 2>/dev/null | tail -1
done > $O/simulate_world.txt
python tools/bench_heads.py > $O/heads.txt 2>&1
python tools/bench_control.py > $O/control.txt 2>&1
NJF_PROFILE=1 python tools/bench_train.py action > $O/train_action.txt 2>&1
NJF_PROFILE=1 python tools/bench_train.py perception > $O/train_perception.txt 2>&1
tail -c 300 $O/bench.json; tail -2 $O/heads.txt; tail -2 $O/control.txt; tail -1 $O/train_action.txt; tail -1 $O/train_perception.txt
