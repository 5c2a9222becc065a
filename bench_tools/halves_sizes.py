"""Developer tool: the online step (bench.OnlineLoop) at 64 x 64 on one stream and as two half-ensembles on two streams
(option streams 1 / 2), interleaved in one process, for a range of ensemble sizes.   python bench_tools/halves_sizes.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
import pyqg_generative_amd as qa
SIZES = [int(a) for a in sys.argv[1:]] or [32, 48, 64, 80, 96, 112, 128, 160, 192, 256, 512, 1024]
for B in SIZES:
    N, kind = 64, 'gan'
    dt = bench.dt_of(N)
    gen, _ = bench.load_generator(kind, 0)
    eng = qa.EnsembleEngine(nx=N, n_members=B, device=0, dt=dt)
    eng.set_q(bench.eddy_like_q(np.arange(B), N))
    loop = bench.OnlineLoop(eng, dt, dict(generator=gen, sampling='constant', nsteps_decor=1, seed=2024, member_offset=0))
    loop.run(60)
    K = 300 if B <= 256 else 100
    out = {}
    for rnd in range(3):
        for st in (1, 2):
            eng.set_option('streams', st)
            loop.run(20)
            t = bench.timed(lambda: loop.run(K)) / K
            out[st] = min(out.get(st, 1e9), t)
    print(f'N={N} B={B}: one stream {1e6 * out[1]:.1f} us/step, two halves {1e6 * out[2]:.1f} ({out[1] / out[2]:.3f} x)', flush=True)
    eng.close()
