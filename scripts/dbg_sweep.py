import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from audio_mps_amd import HParams, PsiCMPS
from audio_mps_amd.scan import HipScan, unpack_grad
from oracle import c_oracle as C
from _util import make_audio, c_oracle_run, rel_inf
rng = np.random.default_rng(12)
for it in range(60):
    D = int(rng.integers(1, 33)); T = int(rng.integers(2, 700)); B = int(rng.integers(1, 14))
    sigma = float(10 ** rng.uniform(-4, 0)); rs = float(10 ** rng.uniform(-1.5, 0)); variant = int(rng.choice([1, 2]))
    A = float(10 ** rng.uniform(0, 2))
    amp = float(10 ** rng.uniform(-3, 0.3))
    r1 = int(rng.choice([2, 3, 4]))
    if (D, T, B) != (18, 501, 11):
        continue
    hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma, A=A)
    audio = (make_audio(B, T, hp.delta_t, it) * np.float32(amp)).astype(np.float32)
    if T > 40 and it % 3 == 0:
        audio[:, : T // 3] = 0.0
    print("it", it, "silent prefix", it % 3 == 0, "sigma", sigma, "rs", rs, "A", A, "amp", amp)
    for var, mode in ((2, 0), (2, 1), (2, 2), (2, 3), (1, 2)):
        m = PsiCMPS(hp, data_iterator=audio, seed=it, backend=HipScan(D, variant=var, rank1=mode))
        m.variables["Rx"] *= np.float32(rs); m.variables["Ry"] *= np.float32(rs)
        flat, _ = m.grad_sums(); g = unpack_grad(flat.cpu().numpy(), D)
        ref = c_oracle_run(m, audio, "f32"); gr = C.unpack_grad(ref["grad"], D)
        r64 = c_oracle_run(m, audio, "f64"); g64 = C.unpack_grad(r64["grad"], D)
        print("variant", var, "mode", mode, {k: f"{rel_inf(g[k], gr[k]):.1e}" for k in ("Rbar", "fbar", "psi0bar", "Abar")},
              "| f32 oracle vs f64:", {k: f"{rel_inf(gr[k], g64[k]):.1e}" for k in ("Rbar", "fbar", "psi0bar", "Abar")})
