import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from audio_mps_amd import HParams, PsiCMPS
from audio_mps_amd.train import Trainer
from audio_mps_amd.data import damped_sine
for (D, T, B) in ((4, 256, 8), (16, 4096, 256), (32, 16000, 1024)):
    hp = HParams(minibatch_size=B, bond_dim=D)
    audio = torch.from_numpy(damped_sine(B, T, hp.delta_t, seed=1)).cuda()
    m = PsiCMPS(hp, seed=0)
    tr = Trainer(m, hp, device_step=True)
    for _ in range(3): tr.step(audio, sync=False, global_batch=B)
    torch.cuda.synchronize()
    def timed(fn, n=20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    eager = timed(lambda: tr.step(audio, sync=False, global_batch=B))
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            tr.step(audio, sync=False, global_batch=B)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            tr.step(audio, sync=False, global_batch=B)
        rep = timed(lambda: g.replay())
        print(f"D={D} T={T} B={B}: eager {eager:.4f} ms  graph replay {rep:.4f} ms")
    except Exception as e:
        print(f"D={D}: eager {eager:.4f} ms; capture failed: {type(e).__name__}: {str(e)[:200]}")
