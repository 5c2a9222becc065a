"""Developer tool: the bench's online loop (diagnostics every 6 steps, snapshots, status) with the ensemble on one stream and as
two halves on two streams, interleaved in one process.   python bench_tools/halves_cadence.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
import pyqg_generative_amd as qa

CASES = [(int(a.split('x')[0]), int(a.split('x')[1]), 'vae' if a.startswith('96') else 'gan') for a in sys.argv[1:]] or [(64, 128, 'gan'), (64, 64, 'gan'), (96, 32, 'vae')]
for N, B, kind in CASES:
    dt = bench.dt_of(N)
    gen, _ = bench.load_generator(kind, 0)
    eng = qa.EnsembleEngine(nx=N, n_members=B, device=0, dt=dt)
    eng.set_q(bench.eddy_like_q(np.arange(B), N))
    loop = bench.OnlineLoop(eng, dt, dict(generator=gen, sampling='constant', nsteps_decor=1, seed=2024, member_offset=0))
    if os.environ.get('QGX_HC_NODIAG'):          # developer switch of this tool: no time-averaged diagnostics in the loop
        eng.diag_config(0, 0)
    loop.run(100)
    for rnd in range(2):
        for streams in (1, 2):
            eng.set_option('streams', streams)
            loop.run(40)
            el = bench.timed(lambda: loop.run(500)) * 2
            print(f'N={N} B={B} streams={streams}: {1e3 * el:.3f} us/step  {B * 1000 / el:.0f} steps/s', flush=True)
    eng.close()
