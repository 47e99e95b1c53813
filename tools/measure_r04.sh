#!/bin/bash
# Round-4 measurement sweep (GPU box, one gpurun call per step list, from the repo root).
#   bash tools/measure_r04.sh <step> [<step> ...]
# Outputs land in gpurun_out/measure_r04/; the ones that are evidence are copied to profiles/r04_* by hand.
cd "$(dirname "$0")/.."
O=gpurun_out/measure_r04; mkdir -p $O
for STEP in "$@"; do
case $STEP in
probe)     # VERDICT r03 "next" #4: MFMA || VALU co-issue on one SIMD
  timeout 120 build/probe_coissue > $O/probe_coissue.txt 2>&1; echo "probe rc=$?"; cat $O/probe_coissue.txt ;;
tests)
  NJF_MARGINS_OUT=$PWD/$O/r04_parity_margins.json timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
  echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt ;;
bench)
  timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json ;;
dist)      # the N > 1 code on ONE GPU: RCCL initialised with a single rank (collective for real), eager and graph
  for extra in "" "--graph"; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-dist $extra 2>>$O/dist.err | tail -1 > $O/force_dist${extra:+_graph}.json
    python -c "import sys,json; d=json.load(open('$O/force_dist${extra:+_graph}.json')); print('force-dist $extra', d['ms_per_step'], d['value_default_precision']['ms_per_step'], d['rccl']['backend'], d['rccl']['devices'][0].get('pci_bus_id'), d['rccl']['rank_step_ms'])"
  done
  for n in 2 4 8; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --simulate-world $n --graph 2>>$O/dist.err | tail -1
  done > $O/simulate_world.txt
  python - <<PY
import json
for l in open('$O/simulate_world.txt'):
    d = json.loads(l); print('sim', d['config']['rays_per_gpu'], d['ms_per_step'], d['value_default_precision']['ms_per_step'], d['kernel_ms'])
PY
  ;;
train)
  for m in action perception; do
    NJF_PROFILE=1 timeout 600 python tools/bench_train.py --mode $m --force-dist > $O/train_$m.json 2> $O/train_$m.txt; echo "train $m rc=$?"
    tail -1 $O/train_$m.json | cut -c1-600; tail -20 $O/train_$m.txt
  done ;;
configs)
  for a in "--batch 4 --samples 128" "--height 512 --width 512" "--samples 256" "--samples 32"; do
    echo "ARGS $a"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline $a 2>/dev/null | tail -1
  done > $O/configs.txt ;;
stream)    # HBM table of the streaming kernels: HIP events with warm and with COLD inputs, and rocprofv3 per-dispatch durations by shape
  timeout 300 python tools/stream_kernels.py --json $O/stream_hip_events_warm.json > $O/stream_hip_events_warm.txt 2>&1; tail -13 $O/stream_hip_events_warm.txt
  timeout 300 python tools/stream_kernels.py --cold --json $O/stream_hip_events_cold.json > $O/stream_hip_events_cold.txt 2>&1; tail -13 $O/stream_hip_events_cold.txt
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/stream_trace -- python $OLDPWD/tools/stream_kernels.py --launches 20 --cold > $OLDPWD/$O/stream_trace.log 2>&1)
  TR=$(find $O/stream_trace -name '*kernel_trace.csv' | head -1)
  [ -n "$TR" ] && python tools/stream_kernels.py --trace $TR --json $O/stream_rocprof.json | tee $O/stream_rocprof.txt
  ST=$(find $O/stream_trace -name '*kernel_stats.csv' | head -1); [ -n "$ST" ] && cp $ST $O/stream_kernel_stats.csv
  [ -n "$TR" ] && cp $TR $O/stream_kernel_trace.csv; rm -rf $O/stream_trace ;;
profile)   # rocprofv3 kernel stats + PMC passes of the default bench, headline (f32) and default-precision (f16f6) modes
  for p in f32 f16f6; do bash tools/profile_r04.sh $p > $O/profile_$p.log 2>&1; done
  ls gpurun_out/prof_r04_f32 gpurun_out/prof_r04_f16f6 ;;
ablate)    # experiment builds (build/libnjf_ablate_<v>.so, -DNJF_ABLATE_<V>): kernel times only, results are garbage
  for prec in f32 f16f6; do
    for v in "" gather barrier dma dmabarrier pe afrag; do
      if [ -z "$v" ]; then unset NJF_HIP_LIB; name=baseline; else export NJF_HIP_LIB=$PWD/build/libnjf_ablate_$v.so; name=$v; fi
      [ -n "$v" ] && [ ! -f "$NJF_HIP_LIB" ] && continue
      timeout 200 python bench.py --precision $prec --steps 8 --warmup 2 --no-cpu-baseline --no-other-precisions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$prec', '$name', d['ms_per_step'], d['kernel_ms'])"
    done
  done | tee $O/ablate.txt
  unset NJF_HIP_LIB ;;
mix)       # VERDICT r03 "next" #5: the first n wide layers of the (f16x2) proposal net on the fp6-corrected product form.
           # (round 5: the kernel switch -DNJF_PROPOSAL_MIX and the host splice it needed were removed from the product -- ADVICE r04;
           #  the experiment is reproducible from commit b0f5543, its result is profiles/r04_proposal_layer_mix_gpu.txt)
  for n in 0 2 3 5; do
    if [ $n = 0 ]; then unset NJF_HIP_LIB NJF_PROPOSAL_MIX; else export NJF_HIP_LIB=$PWD/build/libnjf_mix$n.so NJF_PROPOSAL_MIX=$n; fi
    timeout 300 python bench.py --precision f16f6 --no-other-precisions --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/mix$n.json
    python - <<PY
import json
d = json.load(open('$O/mix$n.json')); p = d['parity_on_bench_frame']['f16f6']
print('mix$n', 'ms', d['ms_per_step'], d['kernel_ms'], {k: (v['err'], v['limit'], v['ok'], v['truth']['ratio_max']) for k, v in p.items()})
PY
    NJF_MARGINS_OUT=$PWD/$O/mix${n}_margins.json timeout 600 python -m pytest tests/test_hip_parity.py tests/test_properties_gpu.py -m gpu -q -k "f16f6 or full_size or default" 2>&1 | tail -4
  done 2>&1 | tee $O/mix.txt
  unset NJF_HIP_LIB NJF_PROPOSAL_MIX ;;
diag)
  NJF_DIAG_DETAIL=1 timeout 500 python tools/diag/diag_perception.py > $O/diag_perception.txt 2>&1; grep -v "^    elem" $O/diag_perception.txt | head -40 ;;
*) echo "unknown step $STEP" ;;
esac
done
