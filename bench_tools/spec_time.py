#!/usr/bin/env python
"""Developer tool: time the unparameterized spectral step kernel (per launch) for a given B, N."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyqg_generative_amd as qa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
e = qa.EnsembleEngine(nx=N, n_members=B, dt=14400. if N <= 64 else 3600.)
rs = np.random.RandomState(0)
e.set_q(rs.randn(B, 2, N, N) * 1e-6)
S = torch.as_tensor(rs.randn(B, 2, N, N) * 1e-12).cuda()
for forcing in (None, S):
    kw = dict(forcing=forcing, demean=True) if forcing is not None else {}
    e.step(20, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 200
    e.step(K, refresh_diag=False, **kw); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f'B={B} N={N} forcing={forcing is not None}: {dt*1e6:.1f} us/step  {B/dt/1e6:.3f} M member-steps/s  '
          f'{(468992 if forcing is not None else 337920) * (N/64)**2 * B / dt / 1e9:.0f} GB/s algorithmic')
