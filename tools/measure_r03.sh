#!/bin/bash
# Round-3 measurement sweep (GPU box, one gpurun call, from the repo root): full GPU suite with the margins table, bench
# line, HBM table of the streaming kernels (HIP events + rocprofv3 averages), other configs, scaling dry run.
# Outputs land in gpurun_out/measure_r03/; the ones that are evidence are copied to profiles/r03_* by hand.
cd "$(dirname "$0")/.."
O=gpurun_out/measure_r03; mkdir -p $O
export NJF_MARGINS_OUT=$PWD/$O/r03_parity_margins.json
STEP=${1:-all}
if [ "$STEP" = all ] || [ "$STEP" = tests ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
fi
if [ "$STEP" = all ] || [ "$STEP" = bench ]; then
  timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json
fi
if [ "$STEP" = all ] || [ "$STEP" = stream ]; then
  timeout 300 python tools/stream_kernels.py --json $O/stream_hip_events.json > $O/stream_hip_events.txt 2>&1; cat $O/stream_hip_events.txt | tail -9
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/stream_trace -- python $OLDPWD/tools/stream_kernels.py --launches 20 > $OLDPWD/$O/stream_trace.log 2>&1)
  ST=$(find $O/stream_trace -name '*kernel_stats.csv' | head -1)
  [ -n "$ST" ] && cp $ST $O/stream_kernel_stats.csv && python tools/stream_kernels.py --stats $O/stream_kernel_stats.csv --json $O/stream_rocprof.json | tee $O/stream_rocprof.txt
fi
if [ "$STEP" = all ] || [ "$STEP" = configs ]; then
  for a in "--batch 4 --samples 128" "--height 512 --width 512" "--samples 256" "--samples 32"; do
    echo "ARGS $a"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-precisions $a 2>/dev/null | tail -1
  done > $O/configs.txt
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-precisions --force-dist 2>/dev/null | tail -1 > $O/force_dist.json
  for n in 2 4 8; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precisions --simulate-world $n 2>/dev/null | tail -1
  done > $O/simulate_world.txt
  tail -c 400 $O/simulate_world.txt
fi
if [ "$STEP" = scaling ]; then   # the N > 1 code on one GPU: RCCL with one rank, and rank 0's shard of an 8-way split, eager / graph / round-2 step
  for extra in "" "--graph" "--legacy-step"; do
    echo "FORCE-DIST $extra"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precisions --force-dist $extra 2>$O/fd.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d.get('step',{}).get('c_abi_launches_per_step'))"
    for n in 2 4 8; do
      echo "SIM $n $extra"; python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-precisions --simulate-world $n $extra 2>>$O/fd.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['kernel_ms'])"
    done
  done > $O/scaling_dry.txt 2>&1
  cat $O/scaling_dry.txt
fi
if [ "$STEP" = train ]; then
  NJF_PROFILE=1 python tools/bench_train.py action > $O/train_action.txt 2>&1; tail -22 $O/train_action.txt
  NJF_PROFILE=1 python tools/bench_train.py perception > $O/train_perception.txt 2>&1; tail -22 $O/train_perception.txt
fi
if [ "$STEP" = extras ]; then
  python tools/bench_heads.py > $O/heads.txt 2>&1; tail -2 $O/heads.txt
  python tools/bench_control.py > $O/control.txt 2>&1; tail -2 $O/control.txt
fi
if [ "$STEP" = profile ]; then   # rocprofv3 kernel stats + PMC passes of the default bench (tools/profile_r03.sh), headline and fp32 modes
  for p in f16f6 f32; do bash tools/profile_r03.sh $p > $O/profile_$p.log 2>&1; done
  ls gpurun_out/prof_r03_f16f6 gpurun_out/prof_r03_f32
fi
