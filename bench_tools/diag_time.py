"""Developer tool: the small-grid diagnostics increment, one workgroup per member: work fields in registers (k_diag_small_reg,
option diag_reg = 2; as two workgroups per member: 3) against work fields in global memory (k_diag_small, 0) — time per increment (HIP events over a run whose every
step increments) and bit identity of the sixteen accumulators.
    python bench_tools/diag_time.py [members] [N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import pyqg_generative_amd as qa

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rs = np.random.RandomState(1)
q0 = rs.randn(B, 2, N, N) * 1e-6
S = torch.as_tensor(rs.randn(B, 2, N, N) * 1e-12, device='cuda')
names = ['KEspec', 'Ensspec', 'entspec', 'APEflux', 'KEflux', 'APEgenspec', 'KEfrictionspec', 'paramspec', 'paramspec_APEflux',
         'paramspec_KEflux', 'Dissspec', 'ENSDissspec', 'ENSflux', 'ENSgenspec', 'ENSfrictionspec', 'ENSparamspec']
res, t = {}, {}
for reg in (0, 2, 3):
    for every in (1, 1000000):
        e = qa.EnsembleEngine(nx=N, n_members=B, dt=14400.)
        e.set_option('diag_wide', 0)
        e.set_option('diag_reg', reg)
        e.set_q(q0)
        e.diag_config(0, every)
        e.step(5, forcing=S, refresh_diag=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        e.step(200, forcing=S, refresh_diag=False)
        b.record()
        torch.cuda.synchronize()
        t[reg, every] = a.elapsed_time(b) / 200 * 1e3
        if every == 1:
            res[reg] = [e.diag(n).clone() for n in names]
        e.close()
    print(f'N={N} B={B} diag_reg={reg}: step with an increment {t[reg, 1]:.1f} us, without {t[reg, 1000000]:.1f} us -> increment {t[reg, 1] - t[reg, 1000000]:.1f} us', flush=True)
print('accumulators bit-identical:', all(torch.equal(x, y) and torch.equal(x, z) for x, y, z in zip(res[0], res[2], res[3])))
for n, x, y in zip(names, res[0], res[3]):
    if not torch.equal(x, y):
        print(f'   {n}: max |diff| / max {float((x - y).abs().max() / x.abs().max()):.2e}')
