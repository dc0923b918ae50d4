#!/usr/bin/env python3
"""Robustness sweep on the GPU by hand: `python scripts/random_sweep.py SEED [family ...]` runs tests/_sweep.py with the full draw
counts (the test module tests/test_gpu_sweep.py runs fixed seeds with fewer draws and asserts the bars) and prints the worst case per
family and quantity, the time per family, and every record above its bar."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _sweep import FAMILIES, run_sweep, worst_by_kind  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
fams = sys.argv[2:] or list(FAMILIES)
records = []
for fam in fams:                       # one generator per family here, so a family can be re-run alone
    t0 = time.perf_counter()
    rec = run_sweep(seed * 1000 + list(FAMILIES).index(fam), families=[fam])
    print(f"# {fam}: {len(rec)} checks in {time.perf_counter() - t0:.1f} s")
    records += rec
for key, (err, bar, cfg) in worst_by_kind(records).items():
    print(f"{key:18s} worst {err:.2e} (bar {bar:.1e}) at {cfg}")
for fam, what, err, bar, cfg in records:
    if not err <= bar:
        print(f"ABOVE BAR: {fam} {what} {err:.3e} > {bar:.1e} at {cfg}")
