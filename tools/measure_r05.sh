#!/bin/bash
# Round-5 measurement sweep (GPU box, one gpurun call per step list, from the repo root).
#   bash tools/measure_r05.sh <step> [<step> ...]
# Outputs land in gpurun_out/measure_r05/; the ones that are evidence are copied to profiles/r05_* by hand.
cd "$(dirname "$0")/.."
O=gpurun_out/measure_r05; mkdir -p $O
for STEP in "$@"; do
case $STEP in
tests)     # the whole -m gpu suite with the committed margins table
  NJF_MARGINS_OUT=$PWD/$O/r05_parity_margins.json timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
  echo "pytest rc=$?"; tail -8 $O/pytest_gpu.txt ;;
newtests)  # round 5's new test files only
  timeout 1800 python -m pytest tests/test_training_fixed_bins_gpu.py tests/test_rccl_gpu.py -m gpu -q -x > $O/pytest_new.txt 2>&1
  echo "pytest rc=$?"; tail -30 $O/pytest_new.txt ;;
bench)
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3000 $O/bench.json ;;
smoke)
  timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -c 1500 $O/smoke.txt ;;
c5)        # BASELINE config 5 (512 x 512, A = 8 bench weights): frame times of every precision, headline protocol
  timeout 600 python bench.py --height 512 --width 512 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
  echo "c5 rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1])
print('f32', d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])
for k,v in d['other_precisions'].items(): print(k, v['ms_per_step'], v['kernel_ms'], v['roofline_frac'])" ;;
configs)
  for a in "--batch 4 --samples 128" "--samples 256" "--samples 32"; do
    echo "ARGS $a"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline $a 2>/dev/null | tail -1
  done > $O/configs.txt ;;
profile_c5)   # the "rocprof roofline report" BASELINE config 5 names: kernel stats + PMC of the 512 x 512 frame, three precisions
  bash tools/profile_r05.sh f16 r05_c5 full "--height 512 --width 512" > $O/profile_c5_f16.log 2>&1
  bash tools/profile_r05.sh f16f6 r05_c5 short "--height 512 --width 512" > $O/profile_c5_f16f6.log 2>&1
  bash tools/profile_r05.sh f32 r05_c5 short "--height 512 --width 512" > $O/profile_c5_f32.log 2>&1
  ls gpurun_out/prof_r05_c5_f16 gpurun_out/prof_r05_c5_f16f6 gpurun_out/prof_r05_c5_f32 ;;
profile_c2)   # the C2 bench: the new plain-fp16 kernels in full, the headline and the default for their durations + traffic
  bash tools/profile_r05.sh f16 r05 full > $O/profile_f16.log 2>&1
  bash tools/profile_r05.sh f32 r05 short > $O/profile_f32.log 2>&1
  bash tools/profile_r05.sh f16f6 r05 short > $O/profile_f16f6.log 2>&1
  ls gpurun_out/prof_r05_f16 gpurun_out/prof_r05_f32 gpurun_out/prof_r05_f16f6 ;;
train)     # SURVEY's C4 as written: one rank of an 8-way split of a C2-shaped batch = ONE scene x 8,192 rays, 64 + 64 samples,
           # fwd + bwd + bucket all-reduce (RCCL, one rank) + Adam, with the top-kernel table; and the reference batch shape
  for m in action perception; do
    NJF_PROFILE=1 timeout 600 python tools/bench_train.py --mode $m --force-dist --scenes 1 --rays 8192 --start-step 20000 > $O/train_c4_$m.json 2> $O/train_c4_$m.txt
    echo "train c4 $m rc=$?"; tail -1 $O/train_c4_$m.json | cut -c1-400; grep -A14 "steady-state" $O/train_c4_$m.txt
    timeout 600 python tools/bench_train.py --mode $m --force-dist --start-step 20000 > $O/train_ref_$m.json 2>> $O/train_ref.err
    tail -1 $O/train_ref_$m.json | cut -c1-300
  done ;;
spills)    # A/B: training instantiations WITHOUT the 16 action-feature accumulators (build/libnjf_train_noaf.so, -DNJF_TRAIN_NO_AF: 45-68
           # spilled VGPRs) against the shipped ones (104-119), same box, C4 shard.  (Round 5 ran it the other way round -- the library
           # then shipped was the no-AF one, build/libnjf_train_af.so the round-4 form -- and kept the faster: profiles/r05_spills_ab.txt)
  for v in shipped train_noaf shipped train_noaf; do
    if [ $v = shipped ]; then unset NJF_HIP_LIB; else export NJF_HIP_LIB=$PWD/build/libnjf_$v.so; fi
    for m in action perception; do
      timeout 300 python tools/bench_train.py --mode $m --scenes 1 --rays 8192 --start-step 20000 --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$m', d['ms_per_step'])"
    done
  done | tee $O/spills_ab.txt
  unset NJF_HIP_LIB ;;
patch)
  timeout 600 python tools/bench_patch_render.py > $O/patch_render.json 2> $O/patch_render.err; echo "patch rc=$?"; cat $O/patch_render.json ;;
abinit)
  timeout 600 python tools/ab_reference_init.py > $O/ab_reference_init.json 2> $O/ab_reference_init.err; echo "abinit rc=$?"
  python -c "
import json; d=json.load(open('$O/ab_reference_init.json')); [print(k, v['summary']) for k, v in d['regimes'].items()]" ;;
heads)
  timeout 300 python tools/bench_heads.py > $O/heads.json 2>/dev/null; cat $O/heads.json
  timeout 300 python tools/bench_control.py > $O/control.json 2>/dev/null; cat $O/control.json ;;
*) echo "unknown step $STEP" ;;
esac
done
