#!/bin/bash
# Final training-step table of round 6 (HEAD, ABI v20): SURVEY's C4 shard (1 x 8,192 rays, 64 + 64) and the reference's default training
# configuration (7 x 256 rays, 256 + 256), both Jacobian heads, both modes, torch matmul precision highest / high.
# Output: gpurun_out/r06_training_final.txt (one line per run)
O=gpurun_out; OUT=$O/r06_training_final.txt; : > $OUT
for shape in "--scenes 1 --rays 8192" "--samples 256"; do
  for dec in jacobian_mlp jacobian_transformer; do
    for mode in action perception; do
      for mp in highest high; do
        line=$(timeout 300 python tools/bench_train.py --mode $mode --decoder $dec $shape --matmul-precision $mp --steps 40 --warmup 8 --start-step 20000 2>/dev/null | tail -1)
        echo "$shape | $dec | $mode | $mp | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms |", d["value"], "rays/s")' 2>/dev/null || echo FAILED)" | tee -a $OUT
      done
    done
  done
done
