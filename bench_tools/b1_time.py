#!/usr/bin/env python
"""Developer tool: single-member (BASELINE configs[1]) step timing."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyqg_generative_amd as qa
from pyqg_generative_amd import weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 64
nets, xs, ys = weights.load_npz(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'weights_gan.npz'), 'gan')
gen = qa.Generator('gan', nets, xs, ys)
e = qa.EnsembleEngine(nx=N, n_members=B, dt=14400.)
rs = np.random.RandomState(0)
e.set_q(rs.randn(B, 2, N, N) * 1e-6)
kw = dict(generator=gen, sampling='constant', nsteps_decor=1, seed=1)
e.step(20, **kw); torch.cuda.synchronize()
K = 400
t0 = time.perf_counter(); e.step(K, refresh_diag=False, **kw); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f'B={B}: {dt*1e6:.1f} us/step  {B/dt:.0f} member-steps/s')
