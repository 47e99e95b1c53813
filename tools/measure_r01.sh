#!/bin/bash
# Round-1 measurement sweep behind profiles/r01_{bench,configs,heads,control_loop,training_step}.json (GPU box, one gpurun call).
cd "$(dirname "$0")/.."
O=gpurun_out/measure_r01; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
for a in "--batch 4 --samples 128" "--height 512 --width 512" "--samples 256" "--samples 32"; do
  echo "ARGS $a"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline $a 2>/dev/null | tail -1
done > $O/configs.txt
python tools/bench_heads.py > $O/heads.txt 2>&1
python tools/bench_control.py > $O/control.txt 2>&1
(python tools/bench_train.py action; python tools/bench_train.py perception) 2>&1 | grep training_step_ms > $O/train.txt
tail -c 300 $O/bench.json; tail -2 $O/heads.txt; tail -2 $O/control.txt; cat $O/train.txt
