#!/bin/bash
# Within-one-box A/B of plain-fp16 experiment libraries (round 6): build/libnjf_<name>.so, kernel_ms of bench.py --precision f16.
#   bash tools/ablate_r06.sh [--args "<bench args>"] f16dev stag10k ...
cd "$(dirname "$0")/.."
ARGS=""
export NJF_AUTO_RANGE_CHECK=0
if [ "$1" = "--args" ]; then ARGS="$2"; shift 2; fi
for v in "$@"; do
  export NJF_HIP_LIB=$PWD/build/libnjf_$v.so
  [ -f "$NJF_HIP_LIB" ] || { echo "$v: missing"; continue; }
  timeout 300 python tools/diag/bench_with_counts.py --precision f16 --steps 8 --warmup 2 --no-cpu-baseline --no-other-precisions $ARGS 2>/tmp/err_$v.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernel_ms'], (d.get('frame_digest') or {}).get('sha256','')[:16])"
  grep "\[stagger\]" /tmp/err_$v.txt
done
unset NJF_HIP_LIB NJF_AUTO_RANGE_CHECK
