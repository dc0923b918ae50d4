#!/bin/bash
# A/B of library builds on ONE device, interleaved rounds in one call (devices differ by ~7 %: never compare across calls).
# usage: ab_libs.sh "D T B ROUNDS VARIANT RANK1" libA.so libB.so ...   (paths relative to the repo root)
ARGS=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    echo "== round $round $lib"
    CMPS_LIB=$(pwd)/$lib timeout -k 10 120 python scripts/time_kernels.py $ARGS 2>/dev/null | grep median
  done
done
