"""Developer tool: microseconds per online step (bench.OnlineLoop: diagnostics cadence, snapshots, status) for a few ensemble sizes; run once per
library build (QGX_LIB=...) on ONE box to compare two builds of the step kernel.   QGX_LIB=path/to/libqgx.so python bench_tools/steptime.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
import pyqg_generative_amd as qa
for N, B, kind in [(64, 1, "gan"), (64, 2, "gan"), (64, 4, "gan"), (64, 8, "gan"), (96, 2, "vae"), (48, 2, "gan")]:
    dt = bench.dt_of(N)
    gen, _ = bench.load_generator(kind, 0)
    eng = qa.EnsembleEngine(nx=N, n_members=B, device=0, dt=dt)
    eng.set_q(bench.eddy_like_q(np.arange(B), N))
    loop = bench.OnlineLoop(eng, dt, dict(generator=gen, sampling='constant', nsteps_decor=1, seed=2024, member_offset=0))
    loop.run(100)
    K = 1000 if B <= 16 else 400
    best = min(bench.timed(lambda: loop.run(K)) / K for _ in range(3))
    print(f'{os.environ.get("QGX_LIB", "default")[-14:]} N={N} B={B}: {1e6 * best:.1f} us/step', flush=True)
    eng.close()
