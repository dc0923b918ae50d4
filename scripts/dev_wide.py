"""Quick look at the wide kernels on the GPU: per-clip loss and gradient errors against the C oracle, timing."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _util import c_oracle_run, make_audio, rel_inf
from audio_mps_amd import HParams, PsiCMPS
from audio_mps_amd.scan import HipScan, unpack_grad
from oracle import c_oracle as C

D, T, B = (int(x) for x in (sys.argv[1:4] + ["64", "300", "4"][len(sys.argv[1:4]):]))
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 5
sigma = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-4
hp = HParams(minibatch_size=B, bond_dim=D, sigma=sigma)
audio = make_audio(B, T, hp.delta_t, 1)
m = PsiCMPS(hp, data_iterator=audio, seed=0, backend=HipScan(D, variant=variant))
m.variables["Rx"] *= np.float32(0.1 if sigma == 1.0 else (1.0 if D <= 64 else 0.35)); m.variables["Ry"] *= np.float32(0.1 if sigma == 1.0 else (1.0 if D <= 64 else 0.35))
per = m.loss_per_clip()
print("variant", m._get_backend().variant, "loss", per[:4])
nref = min(B, 16)
ref = c_oracle_run(m, audio[:nref], "f32", nthreads=16)
print("ref ", ref["loss_per_clip"][:4])
print("loss err", np.max(np.abs(per[:nref] - ref["loss_per_clip"]) / np.maximum(np.abs(ref["loss_per_clip"]), 1.0)))
if nref == B:
    flat = m.grad_sums()[0].cpu().numpy()
    g, gr = unpack_grad(flat, D), C.unpack_grad(ref["grad"], D)
    for k in ("Rbar", "fbar", "psi0bar", "Abar"):
        print(k, rel_inf(g[k], gr[k]), float(np.max(np.abs(gr[k]))))
be = m._get_backend()
d_audio = torch.from_numpy(audio).cuda()
for it in range(3):
    be.set_params(m.effective_params(), B, T, train=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    be.forward(d_audio, save_for_bwd=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    be.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"fwd {1e3*(t1-t0):.3f} ms  bwd {1e3*(t2-t1):.3f} ms  ({1e9*(t1-t0)/max(T-1,1):.0f} / {1e9*(t2-t1)/max(T-1,1):.0f} ns per step)")
