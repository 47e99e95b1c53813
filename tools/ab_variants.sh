#!/bin/bash
# A/B of kernel variants inside ONE gpurun call: build/libnjf_<variant>.so (built here on the CPU box) against the
# shipped library.  For every variant: output checksums (must equal the first variant's) and bench.py kernel times.
#   bash tools/ab_variants.sh base bufdma ...      ("shipped" = neural-jacobian-field_amd/libnjf_hip.so)
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ "$v" = shipped ]; then unset NJF_HIP_LIB; else export NJF_HIP_LIB=$PWD/build/libnjf_$v.so; fi
  sums=$(python tools/ab_checksum.py 2>/dev/null | tr '\n' ' ')
  line=$(python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], 'f32:', d['other_precision']['ms_per_step'], d['other_precision']['render_kernel_ms'])")
  echo "$v | $sums| $line"
done
