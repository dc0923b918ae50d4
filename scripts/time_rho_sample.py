import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from audio_mps_amd import HParams, RhoCMPS
for rank in (4, 32):
    hp = HParams(minibatch_size=8, bond_dim=32, initial_rank=rank, sigma=0.05)
    m = RhoCMPS(hp, seed=2)
    n, length = 8, 2000
    noise = (0.05 * np.sqrt(hp.delta_t) * np.random.default_rng(0).standard_normal((length, n))).astype(np.float32)
    m.sample(n, length, noise=noise)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w = m.sample(n, length, noise=noise)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"rank {rank}: sample({n} paths x {length} steps) {dt*1e3:.2f} ms -> {dt/length*1e6:.2f} us/step")
