#!/usr/bin/env python
"""Developer tool: create / use / destroy generators and engines repeatedly and report device-memory drift."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import pyqg_generative_amd as qa
from pyqg_generative_amd import weights
nets, xs, ys = weights.load_npz('tests/golden/weights_gan.npz', 'gan')
x = torch.randn((32, 4, 64, 64), device='cuda')
free0 = None
for i in range(40):
    g = qa.Generator('gan', nets, xs, ys)
    y = g.cnn_forward(x)
    e = qa.EnsembleEngine(nx=64, n_members=8, dt=14400.)
    e.set_q(np.random.randn(8, 2, 64, 64) * 1e-6)
    e.step(2, generator=g, sampling='AR1', nsteps_decor=1, seed=i)
    torch.cuda.synchronize()
    del g, e, y
    if i == 5:
        free0 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
print('free after 6 iterations', free0 >> 20, 'MiB; after 40', free1 >> 20, 'MiB; leaked', (free0 - free1) >> 20, 'MiB')
