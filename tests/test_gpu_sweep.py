"""The random sweep under `-m gpu`: fixed seeds, every kernel family, asserted per-family bars (tests/_sweep.py).
Rounds 2-3 shipped a wrong Qbar in k_bwd_wave under 223 green tests; the sweep that found it was a hand-run script.  Seeds 14-20 are
the ones logged in profiles/r4_random_sweep*.log (same generator, fewer draws per family so that the module stays under ~90 s), 21-23
are new.  What the draws make visible: Q = -(delta_t sigma^2 / 2) R^dagger R of /root/reference/model.py:312 at sigma up to 1, R
scaled over 1.5 decades, amplitudes over three decades with silent stretches (the fp16 scales), sampling rates 3-100 kHz."""
import time

import pytest

from _sweep import run_sweep, worst_by_kind

pytestmark = pytest.mark.gpu

COUNTS = {"psi": 24, "wide": 6, "step": 2, "pair": 5, "rho": 12, "legacy": 4}


@pytest.mark.parametrize("seed", [14, 15, 16, 17, 18, 19, 20, 21, 22, 23])
def test_random_sweep(seed):
    t0 = time.perf_counter()
    records = run_sweep(seed, COUNTS)
    worst = worst_by_kind(records)
    print(f"seed {seed}: {len(records)} checks in {time.perf_counter() - t0:.1f} s")
    for key, (err, bar, cfg) in worst.items():
        print(f"  {key:18s} worst {err:.2e} (bar {bar:.1e}) at {cfg}")
    bad = [(fam, what, err, bar, cfg) for fam, what, err, bar, cfg in records if not err <= bar]
    assert not bad, bad
    assert {r[0] for r in records} == set(COUNTS)          # every family produced checks
