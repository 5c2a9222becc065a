"""Developer tool: time per step of runs of unparameterized 256 x 256 steps (XCD-resident kernel vs three launches).
python bench_tools/team_time.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import bench
import pyqg_generative_amd as qa

Bs = [int(a) for a in sys.argv[1:]] or [64]
for B in Bs:
    e = qa.EnsembleEngine(nx=256, n_members=B, device=0, dt=3600.)
    e.set_q(bench.eddy_like_q(list(range(B)), 256))
    e.step(10, refresh_diag=False)
    for K in (1, 2, 5, 23, 100):
        reps = max(1, 200 // K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            e.step(K, refresh_diag=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (reps * K)
        print(f'B={B} run of {K:3d}: {1e6 * dt:7.1f} us/step  {1e6 * dt / ((B + 7) // 8):6.2f} us per member-step per team '
              f'{B / dt:9.0f} member-steps/s', flush=True)
    e.close()
