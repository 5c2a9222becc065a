"""Developer tool: the online step (bench.OnlineLoop) with two and with four workgroups per member in the step kernel
(option siblings 0 / 1), interleaved in one process.   python bench_tools/sibtime.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
import pyqg_generative_amd as qa
for N, B, kind in [(64, 1, 'gan'), (64, 8, 'gan'), (64, 16, 'gan'), (64, 32, 'gan'), (64, 64, 'gan'), (96, 32, 'vae'), (96, 16, 'vae'), (48, 16, 'gan')]:
    dt = bench.dt_of(N)
    gen, _ = bench.load_generator(kind, 0)
    eng = qa.EnsembleEngine(nx=N, n_members=B, device=0, dt=dt)
    eng.set_q(bench.eddy_like_q(np.arange(B), N))
    loop = bench.OnlineLoop(eng, dt, dict(generator=gen, sampling='constant', nsteps_decor=1, seed=2024, member_offset=0))
    loop.run(100)
    K = 1000 if B <= 16 else 400
    out = {}
    for rnd in range(2):
        for sib in (0, 1):
            eng.set_option('siblings', sib)
            loop.run(40)
            out[sib] = min(out.get(sib, 1e9), bench.timed(lambda: loop.run(K)) / K)
    print(f'N={N} B={B}: two workgroups per member {1e6 * out[0]:.1f} us/step, four {1e6 * out[1]:.1f} ({out[0] / out[1]:.3f} x)', flush=True)
    eng.close()
