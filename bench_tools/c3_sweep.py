#!/usr/bin/env python
"""Developer tool: step time of BASELINE configs[3]'s shard (96 x 96 jet + CVAE, 32 members) under generator options.
    python bench_tools/c3_sweep.py "h2_x96=1" "h2_x96=1,h2_w8_min96=256" ...
"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyqg_generative_amd as qa
import bench

N, B = int(os.environ.get('NX', 96)), int(os.environ.get('MEMBERS', 32))
gen, _ = bench.load_generator(os.environ.get('KIND', 'vae'), 0)
eng = qa.EnsembleEngine(nx=N, n_members=B, dt=bench.dt_of(N), **(bench.JET if N == 96 else {}))
eng.set_q(bench.eddy_like_q(np.arange(B), N))
kw = dict(generator=gen, sampling='constant', nsteps_decor=1, seed=1)
base = {}
for spec in ['base'] + sys.argv[1:] + ['base']:
    opts = dict(kv.split('=') for kv in spec.split(',')) if spec != 'base' else {}
    for k, v in opts.items():
        gen.set_option(k, int(v))
    eng.step(10, **kw)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.step(40, **kw)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40)
    print(f'{spec:40s} {1e6 * np.median(ts):8.1f} us/step  (min {1e6 * min(ts):.1f})')
    for k in opts:
        gen.set_option(k, {'h2_x96': 1, 'h2_w8_min96': 1024}.get(k, 0))
