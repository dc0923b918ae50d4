#!/bin/bash
# Diagnostic: sample GPU clock / power with rocm-smi while the step loop runs (CMPS_FWD1=1 selects the one-wave forward).
mkdir -p gpurun_out
run() {
  python scripts/time_kernels.py 32 16000 1024 60 > gpurun_out/poll_$1.txt 2>&1 &
  PID=$!
  sleep 6
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket" | tr '\n' ' '; echo; sleep 0.4; done
  wait $PID
  grep median gpurun_out/poll_$1.txt
}
echo "== two-wave forward"; run fwd2
echo "== one-wave forward"; CMPS_FWD1=1 run fwd1
