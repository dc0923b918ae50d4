#!/usr/bin/env python3
"""Times the forward (with / without stash) and reverse scan kernels with HIP events (interleaved rounds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from audio_mps_amd import HParams, PsiCMPS
from audio_mps_amd.scan import HipScan
from audio_mps_amd.data import damped_sine

D = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
hp = HParams(minibatch_size=B, bond_dim=D)
rng = np.random.default_rng(15345)
x = damped_sine(B, T, hp.delta_t, seed=3000)
audio = torch.from_numpy((x + 0.02 * rng.standard_normal(x.shape)).astype(np.float32)).cuda()
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
rank1 = int(sys.argv[6]) if len(sys.argv) > 6 else None      # cmps_set_option(CMPS_OPT_RANK1): 0 exact, 1 bf16x2, 2 bf16x3
be = HipScan(D, variant=variant, rank1=rank1)
if len(sys.argv) > 7:                                         # cmps_set_option(CMPS_OPT_WIDE_CHAIN): 0 VALU chain, 1 fp16 x 2 on the matrix cores
    be.set_wide_chain(int(sys.argv[7]))
if os.environ.get("CMPS_BWD_WAVES"):                          # cmps_set_option(CMPS_OPT_BWD_WAVES)
    from audio_mps_amd import _capi
    _capi.check(be._h, be._lib.cmps_set_option(be._h, _capi.CMPS_OPT_BWD_WAVES, int(os.environ["CMPS_BWD_WAVES"])))
m = PsiCMPS(hp, seed=0, backend=be)
be.set_params(m.effective_params(), B, T, train=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
res = {"fwd_nosave": [], "fwd_save": [], "bwd": []}
for r in range(rounds + 1):
    ev[0].record(); be.forward(audio, save_for_bwd=False)
    ev[1].record(); be.forward(audio, save_for_bwd=True)
    ev[2].record(); be.backward()
    ev[3].record(); torch.cuda.synchronize()
    if r:
        res["fwd_nosave"].append(ev[0].elapsed_time(ev[1])); res["fwd_save"].append(ev[1].elapsed_time(ev[2])); res["bwd"].append(ev[2].elapsed_time(ev[3]))
N = T - 1
for k, v in res.items():
    ms = float(np.median(v))
    print(f"{k:12s} median {ms:8.3f} ms  min {min(v):8.3f}  -> {ms*1e6/N:7.1f} ns/step")
print("loss", float(be._loss.mean()))
