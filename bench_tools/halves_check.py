"""Developer tool: an ensemble stepped in two halves on two streams (option `streams`) against the single-stream step —
bit-identity of the state and step time, interleaved.   python bench_tools/halves_check.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import pyqg_generative_amd as qa

for N, B, kind, dt in ((64, 128, 'gan', 3600.), (96, 32, 'vae', 7200.), (64, 32, 'gan', 3600.), (64, 1024, 'gan', 3600.), (64, 256, 'gan', 3600.)):
    gen, _ = bench.load_generator(kind, 0)
    q0 = bench.eddy_like_q(list(range(B)), N)
    res = {}
    for streams in (1, 2):
        e = qa.EnsembleEngine(nx=N, n_members=B, device=0, dt=dt)
        e.set_option('streams', streams)
        e.set_q(q0)
        e.diag_config(0, 6)
        kw = dict(generator=gen, sampling='AR1', nsteps_decor=3, seed=2024, member_offset=5)
        e.step(37, **kw)
        res[streams] = (e.get(qa._lib.F_Q).clone(), e.diag('KEspec').clone(), e.diag('ENSparamspec').clone(), e.status())
        e.close()
    dq = (res[1][0] - res[2][0]).abs().max().item()
    dd = max((res[1][i] - res[2][i]).abs().max().item() for i in (1, 2))
    print(f'N={N} B={B}: max|q_1stream - q_2streams| = {dq:.3e}, diagnostics {dd:.3e}, KE {float(abs(res[1][3][0] - res[2][3][0]).max()):.3e}', flush=True)
    e = qa.EnsembleEngine(nx=N, n_members=B, device=0, dt=dt)
    e.set_q(q0)
    kw = dict(generator=gen, sampling='constant', nsteps_decor=1, seed=2024, member_offset=0)
    e.step(40, **kw)
    K = 80 if B > 256 else 400
    for rnd in range(2):
        for streams in (1, 2):
            e.set_option('streams', streams)
            e.step(16, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.step(K, **kw)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / K
            print(f'   streams={streams}: {1e6 * t:.1f} us/step  {B / t:.0f} steps/s', flush=True)
    e.close()
