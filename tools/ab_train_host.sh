#!/bin/bash
# One-box A/B of the action-mode training step's HOST side (round 6): the tree of the previous commit (build/old_tree, made with
# `git archive <rev> __graft_entry__.py tools/bench_train.py neural-jacobian-field_amd neural_jacobian_field_amd include | tar -x -C
# build/old_tree` + the built library) against the working tree, with and without the frozen encoder's HIP graph
# (NJF_ENCODER_GRAPH=0).  Reference default training configuration (7 scenes x 256 rays, 256 + 256 samples) and SURVEY's C4 shard,
# both Jacobian heads, torch matmul precision highest / high.  Output: gpurun_out/r06_ab_train_host.txt
O=gpurun_out; mkdir -p $O; OUT=$O/r06_ab_train_host.txt; : > $OUT
run() {  # label, tree, env, args...
  local label=$1 tree=$2 envs=$3; shift 3
  local line
  line=$(env $envs timeout 600 python $tree/tools/bench_train.py "$@" 2>>$O/r06_ab_train_host.err | tail -1)
  echo "$label | $* | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  loss", d["final_loss"])' 2>/dev/null || echo FAILED)" | tee -a $OUT
}
STEPS="--steps 40 --warmup 8 --start-step 20000"
for cfg in "--decoder jacobian_transformer --samples 256" "--decoder jacobian_mlp --samples 256" "--decoder jacobian_transformer --scenes 1 --rays 8192" "--decoder jacobian_transformer"; do
  for mp in highest high; do
    run "old tree        " build/old_tree "NJF_X=0" --mode action $cfg --matmul-precision $mp $STEPS
    run "new, eager trunk" .              "NJF_ENCODER_GRAPH=0" --mode action $cfg --matmul-precision $mp $STEPS
    run "new, graph trunk" .              "NJF_X=0" --mode action $cfg --matmul-precision $mp $STEPS
  done
done
