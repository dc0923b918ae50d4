#!/bin/bash
# per-kernel launch times of one command: kstats.sh TAG python3 scripts/time_kernels.py ...   (rocprofv3 --kernel-trace --stats)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kstats_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT/run.log 2>&1
python3 -c "
import csv,glob,sys
for f in glob.glob(sys.argv[1]+\"/*/*kernel_stats.csv\"):
    for r in list(csv.DictReader(open(f)))[:8]: print(r[\"Name\"].split(\"(\")[0][-40:], r[\"Calls\"], round(float(r[\"AverageNs\"])/1e6,3), \"ms\")
" $OUT
