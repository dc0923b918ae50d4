#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun; raw output under gpurun_out/prof_<tag>, condensed into profiles/ by
# scripts/summarize_prof.py).  usage: profile_scan.sh TAG D T B [extra bench.py flags ...]
#   1) rocprofv3 --kernel-trace --stats of the bench command
#   2) PMC passes, one counter group per run (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys / hip /
#      hsa traces); the program itself follows `--` (python3 bench.py ...), no wrapper in between.
TAG=${1:-r2_c3}; D=${2:-32}; T=${3:-16000}; B=${4:-1024}; shift 4 || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --bond-dim $D --T $T --batch-per-gpu $B $*"
echo "python3 bench.py $ARGS" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
         "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  NAME=$(echo $C | tr ' ' '+' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$NAME -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$NAME.log 2>&1 || echo "pmc pass $NAME failed"
done
python3 $ROOT/scripts/summarize_prof.py $OUT $TAG "$D" "$T" "$B" | tail -40
