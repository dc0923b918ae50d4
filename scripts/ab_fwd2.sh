#!/bin/bash
# A/B of the one-wave and two-wave forward kernels in ONE box: parity subset with the new kernel, then timings.
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or oracle_parity or chunk_boundaries or config3 or config2 or variants_agree or single_clip or normalisation or more_clips or full_size" > gpurun_out/ab_parity.log 2>&1 || { tail -30 gpurun_out/ab_parity.log; exit 1; }
tail -3 gpurun_out/ab_parity.log
echo "--- one wave per clip"; CMPS_FWD1=1 timeout -k 10 200 python scripts/time_kernels.py
echo "--- two waves per clip"; timeout -k 10 200 python scripts/time_kernels.py
