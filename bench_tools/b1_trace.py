"""Developer tool: wall time per step of the single-member online loop (BASELINE configs[1]) and of small ensembles;
run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations.   python bench_tools/b1_trace.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import pyqg_generative_amd as qa

Bs = [int(a) for a in sys.argv[1:]] or [1]
gen, _ = bench.load_generator('gan', 0)
for B in Bs:
    e = qa.EnsembleEngine(nx=64, n_members=B, device=0, dt=3600.)
    e.set_q(bench.eddy_like_q(list(range(B)), 64))
    kw = dict(generator=gen, sampling='constant', nsteps_decor=1, seed=2024, member_offset=0)
    e.step(200, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.step(2000, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'B={B}: {1e6 * dt / 2000:.1f} us/step  {B * 2000 / dt:.0f} member-steps/s', flush=True)
    e.close()
