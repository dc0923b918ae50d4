#!/bin/bash
# PMC counters of the kernels of one command, one counter group per run: pmc_kernel.sh TAG python3 /abs/path/script.py args...
# (raw csv under gpurun_out/pmc_<TAG>/<group>; summarise with scripts/pmc_kernel_summary.py)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/g$i -- "$@" > $OUT/g$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 $ROOT/scripts/pmc_kernel_summary.py $OUT
