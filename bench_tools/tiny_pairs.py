"""Developer tool: the online step of tiny ensembles with the generator option tiny_pairs 0 / 3 / 4 / 7 (which of the layer pairs (7,8), (5,6), (3,4) run as one\nlaunch on 2-row strips), interleaved in one process.   python bench_tools/tiny_pairs.py"""
import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
import pyqg_generative_amd as qa
import pyqg_generative_amd._lib as L
for N, B, kind in [(64, 1, 'gan'), (64, 2, 'gan'), (64, 4, 'gan')]:
    dt = bench.dt_of(N)
    gen, _ = bench.load_generator(kind, 0)
    eng = qa.EnsembleEngine(nx=N, n_members=B, device=0, dt=dt)
    eng.set_q(bench.eddy_like_q(np.arange(B), N))
    loop = bench.OnlineLoop(eng, dt, dict(generator=gen, sampling='constant', nsteps_decor=1, seed=2024, member_offset=0))
    loop.run(100)
    out = {}
    x = torch.randn((B, 4, N, N), dtype=torch.float32, device='cuda')
    ys = {}
    for rnd in range(2):
        for tp in (0, 3, 4, 7):
            gen.set_option('tiny_pairs', tp)
            ys[tp] = gen.cnn_forward(x).clone()
            loop.run(40)
            out[tp] = min(out.get(tp, 1e9), bench.timed(lambda: loop.run(1000)) / 1000)
    gen.set_option('tiny_pairs', 1)
    d = lambda a, b: float((a - b).abs().max() / a.abs().max())
    print(f'N={N} B={B}: no strips {1e6 * out[0]:.1f} us/step, (5,6)+(7,8) {1e6 * out[3]:.1f}, (3,4) {1e6 * out[4]:.1f}, all three {1e6 * out[7]:.1f}; max diff {d(ys[0], ys[4]):.1e} {d(ys[0], ys[7]):.1e}', flush=True)
    eng.close()
