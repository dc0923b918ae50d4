#!/bin/bash
# PMC passes over the pair kernels (bond dimension 128) at a reduced length; outputs under gpurun_out/prof_wave_pmc
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_wave_pmc
rm -rf $OUT; mkdir -p $OUT
export PROF_OUT=$OUT     # read by the summariser below (the working directory changes to /tmp)
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/scripts/time_kernels.py 32 16000 1024 2"
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  TAG=$(echo $C | tr ' ' '+' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$TAG -- $CMD > $OUT/pmc_$TAG.log 2>&1 || echo "pmc pass $TAG failed"
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["PROF_OUT"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/*/*counter_collection.csv"):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"].split("(")[0]
        if "wave" in n:
            per[(n, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (n, _, c), v in per.items():
        acc[n][c].append(v)
steps = 1024 * 15999.0    # wave-steps: 256 workgroups x 4 waves x N
for n, cs in acc.items():
    print("==", n)
    for c, v in sorted(cs.items()):
        a = sum(v) / len(v)
        print(f"   {c:28s} {a:16.0f}   per wave-step {a / steps:10.2f}")
PY
