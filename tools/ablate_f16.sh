#!/bin/bash
# Within-one-box A/B of the plain-fp16 kernels: build/libnjf_f16dev.so (-DNJF_DEV_ONLY_PREC=3) against experiment builds
# build/libnjf_ablate_f16_<v>.so (additionally -DNJF_ABLATE_<V>; their results are numerically meaningless, only kernel_ms is read)
# and any other variant library named on the command line (build/libnjf_<name>.so).
#   bash tools/ablate_f16.sh [--args "<bench args>"] f16dev ablate_f16_gather ...
cd "$(dirname "$0")/.."
ARGS=""
export NJF_AUTO_RANGE_CHECK=0   # the single-precision libraries have no exact-fp32 path for the range guard
if [ "$1" = "--args" ]; then ARGS="$2"; shift 2; fi
for v in "$@"; do
  export NJF_HIP_LIB=$PWD/build/libnjf_$v.so
  [ -f "$NJF_HIP_LIB" ] || { echo "$v: missing"; continue; }
  timeout 200 python bench.py --precision f16 --steps 8 --warmup 2 --no-cpu-baseline --no-other-precisions $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernel_ms'])"
done
unset NJF_HIP_LIB NJF_AUTO_RANGE_CHECK
