"""Developer tool: the bench's online loop (diagnostics cadence, snapshots, status) with the step kernel whole and as two
kernels, the forcing-independent half on a side stream under the generator (option split_adv), interleaved in one process;
and the bit-identity of the two on a fresh pair of engines.   python bench_tools/split_adv.py [NxB ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
import pyqg_generative_amd as qa
import pyqg_generative_amd._lib as L

CASES = [(int(a.split('x')[0]), int(a.split('x')[1]), 'vae' if a.startswith('96') else 'gan') for a in sys.argv[1:]] or \
    [(64, 1, 'gan'), (64, 4, 'gan'), (64, 16, 'gan'), (64, 32, 'gan'), (64, 64, 'gan'), (64, 128, 'gan'), (96, 32, 'vae'), (96, 8, 'vae'), (48, 16, 'gan')]
for N, B, kind in CASES:
    dt = bench.dt_of(N)
    gen, _ = bench.load_generator(kind, 0)
    # bit-identity first: 13 steps, white noise from the device stream, diagnostics cadence inside
    res = []
    for sa in (0, 1):
        e = qa.EnsembleEngine(nx=N, n_members=B, device=0, dt=dt)
        e.set_option('split_adv', sa)
        e.set_q(bench.eddy_like_q(np.arange(B), N))
        e.diag_config(0, 4)
        for chunk in (7, 1, 5):
            e.step(chunk, generator=gen, sampling='constant', nsteps_decor=1, seed=11, member_offset=3)
        res.append([e.get(f).clone() for f in (L.F_QH, L.F_S, L.F_Q)])
        e.close()
    same = all(torch.equal(a, b) for a, b in zip(*res))
    eng = qa.EnsembleEngine(nx=N, n_members=B, device=0, dt=dt)
    eng.set_q(bench.eddy_like_q(np.arange(B), N))
    loop = bench.OnlineLoop(eng, dt, dict(generator=gen, sampling='constant', nsteps_decor=1, seed=2024, member_offset=0))
    loop.run(100)
    K = 1000 if B <= 16 else 400
    out = {}
    for rnd in range(2):
        for sa in (0, 1):
            eng.set_option('split_adv', sa)
            loop.run(40)
            el = bench.timed(lambda: loop.run(K))
            out[sa] = el / K
    print(f'N={N} B={B}: whole {1e6 * out[0]:.1f} us/step, two kernels {1e6 * out[1]:.1f} us/step ({out[0] / out[1]:.3f} x); bit-identical {same}', flush=True)
    eng.close()
