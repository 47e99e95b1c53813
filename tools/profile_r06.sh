#!/bin/bash
# rocprofv3 recipe for the round-6 profiles (run on the GPU box through gpurun, from the repo root).
#   bash tools/profile_r06.sh <precision: f16|f16f6|f16x2|f32> [<tag> [<pmc: full|short|none> [<extra bench.py args>]]]
#     tag   r06 (default: the C2 bench), r06_c5 (BASELINE config 5: "--height 512 --width 512 --action-dim 6"), r06_transformer ("--decoder jacobian_transformer") (BASELINE config 5: pass "--height 512 --width 512" as extra args)
# The profiled command is the default bench (`python bench.py`, plus flags that only drop the untimed extras: the other
# precisions and the CPU baseline).  Kernel-trace/stats and each PMC group are separate runs (PMC is never combined with
# other trace domains).  Condensed into profiles/ by tools/summarize_profile.py <tag> <precision>.
set -u
PREC=${1:-f16}
TAG=${2:-r06}
PMC=${3:-full}
EXTRA=${4:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_$PREC
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --precision $PREC --no-other-precisions --no-cpu-baseline $EXTRA"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH --steps 20 --warmup 3 > $OUT/trace.log 2>&1
[ "$PMC" = none ] && { ls $OUT; exit 0; }
PMC_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU")
if [ "$PMC" = full ]; then
  PMC_GROUPS+=("TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM"
           "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max"
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum")
fi
i=0
for grp in "${PMC_GROUPS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- $BENCH --steps 2 --warmup 1 > $OUT/pmc$i.log 2>&1
done
ls $OUT
