"""Developer tool: phase timeline of the XCD-resident step kernel (library built with -DQGX_TEAM_STAMPS, QGX_LIB=...)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import pyqg_generative_amd as qa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
e = qa.EnsembleEngine(nx=256, n_members=B, device=0, dt=3600.)
e.set_q(bench.eddy_like_q(list(range(B)), 256))
for _ in range(3):
    e.step(10, refresh_diag=False)
    e.status()
e.close()
