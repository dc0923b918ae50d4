import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from audio_mps_amd import HParams, PsiCMPS, _capi
from audio_mps_amd.scan import HipScan, unpack_grad
from _util import *
from oracle import c_oracle as C

def run(D, T, B, sigma, variant, seed=0):
    hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma)
    audio = make_audio(B, T, hp.delta_t, seed)
    m = PsiCMPS(hp, data_iterator=audio, seed=seed, backend=HipScan(D, variant=variant))
    t0 = time.time()
    per = m.loss_per_clip()
    flat, _ = m.grad_sums()
    flat = flat.cpu().numpy()
    torch.cuda.synchronize(); t1 = time.time()
    o32 = c_oracle_run(m, audio, "f32"); o64 = c_oracle_run(m, audio, "f64")
    print(f"D={D} T={T} B={B} sigma={sigma} variant={variant} time={t1-t0:.3f}s")
    print("  loss hip", per[:3], "o32", o32["loss_per_clip"][:3])
    print("  loss rel err vs o32 (per clip max)", np.max(np.abs(per - o32["loss_per_clip"]) / np.abs(o32["loss_per_clip"])),
          " o32 vs o64", np.max(np.abs(o64["loss_per_clip"] - o32["loss_per_clip"]) / np.abs(o64["loss_per_clip"])),
          " hip vs o64", np.max(np.abs(o64["loss_per_clip"] - per) / np.abs(o64["loss_per_clip"])))
    g = unpack_grad(flat, D); g32 = C.unpack_grad(o32["grad"], D); g64 = C.unpack_grad(o64["grad"], D)
    for k in ("Rbar", "fbar", "psi0bar", "Abar", "loss_sum"):
        print(f"  {k}: hip-vs-o64 {rel_inf(g[k], g64[k]):.3e}  o32-vs-o64 {rel_inf(g32[k], g64[k]):.3e}  hip-vs-o32 {rel_inf(g[k], g32[k]):.3e}")

for variant in (1, 2):
    run(4, 256, 8, 1e-4, variant)
    run(7, 256, 8, 1.0, variant)
    run(16, 512, 16, 1e-4, variant)
    run(32, 1000, 32, 1e-4, variant)
run(64, 300, 8, 1e-4, 1)
run(128, 100, 4, 1e-4, 1)
