#!/bin/bash
# Within-one-box A/B of experiment builds (build/libnjf_ablate_*.so, compiled with -DNJF_ABLATE_*).
# Results of ablated builds are numerically meaningless; only kernel_ms is read.
for v in "" "$@"; do
  if [ -z "$v" ]; then unset NJF_HIP_LIB; name=baseline; else export NJF_HIP_LIB=$PWD/build/libnjf_ablate_$v.so; name=$v; fi
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-precisions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['kernel_ms'])"
done
