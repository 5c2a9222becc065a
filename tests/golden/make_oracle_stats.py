#!/usr/bin/env python
"""Generates tests/golden/oracle_stats_*.npz: ensemble statistics of long PARAMETERIZED runs of the CPU oracle
(oracle/qg_ref.py + oracle/gen_ref.py — this repo's restatement, NOT the reference itself: pyqg cannot be imported
here).  They are cached oracle output: a 64x64 eddy + CGAN member costs ~5 CPU-minutes and a 96x96 jet +
CVAE member ~35, too slow to recompute inside the GPU test suite, which compares the same statistics of a GPU ensemble
run with the same protocol (tests/test_gpu_statistics.py).  Members differ in initial condition and noise stream.

    python tests/golden/make_oracle_stats.py eddy64_gan  [n_members] [n_procs]
    python tests/golden/make_oracle_stats.py jet96_vae   [n_members] [n_procs]
"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

YEAR = 360 * 86400.
CASES = {
    # BASELINE configs[1]/[2]: eddy 64x64 + CGAN, sampling='constant' nsteps=1 (scripts/run_parameterized.py:50)
    'eddy64_gan': dict(kind='gan', nx=64, params=dict(dt=14400.), nsteps=12000, tave=8000, taveint=86400.),
    # BASELINE configs[3]: jet 96x96 + CVAE (tools/parameters.py:26-27,37), 10 years, averaged over the second half
    'jet96_vae': dict(kind='vae', nx=96, params=dict(dt=7200., rek=7e-8, delta=0.1, beta=1e-11),
                      nsteps=int(10 * YEAR / 7200.), tave=int(5 * YEAR / 7200.), taveint=86400.),
}
DIAGS = ('KEspec', 'Ensspec', 'APEgenspec', 'KEflux', 'APEflux', 'KEfrictionspec', 'paramspec')


def member(args):
    case, b = args
    import torch
    torch.set_num_threads(1)
    from oracle import qg_ref, gen_ref, samplers_ref
    c = CASES[case]
    d = np.load(os.path.join(ROOT, 'tests', 'golden', f'weights_{c["kind"]}.npz'))
    ora = gen_ref.GeneratorRef(c['kind'], [gen_ref.CNNWeights.from_npz_dict(d, 'net0_')], d['x_std'], d['y_std'])
    dt = c['params']['dt']
    m = qg_ref.QGModelRef(nx=c['nx'], tmax=dt * c['nsteps'], tavestart=dt * c['tave'], taveint=c['taveint'],
                          twrite=10 ** 9, **c['params'])
    m.sampling_type = 'constant'
    m.noise_sampler = samplers_ref.make_sampler('constant', 1)
    m.q_parameterization = gen_ref.ParameterizationRef(ora, rng=np.random.RandomState(50000 + b))
    qg_ref.set_initial_condition(m, np.random.RandomState(1000 + b))
    ke_t = []
    t0 = time.time()
    while m.t < m.tmax:
        m._step_forward()
        if m.tc % 500 == 0:
            ke_t.append(m._calc_ke())
            if b == 0 and m.tc % 5000 == 0:
                print(f'[{case}] member 0: step {m.tc}/{c["nsteps"]}, KE {ke_t[-1]:.3e}, {time.time() - t0:.0f} s', flush=True)
    out = {k: m.get_diagnostic(k) for k in DIAGS}
    out['ke_series'] = np.array(ke_t)
    out['cfl'] = m._calc_cfl()
    return out


def main():
    case = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else min(n, os.cpu_count() or 1)
    import multiprocessing as mp
    with mp.get_context('spawn').Pool(procs) as pool:
        res = pool.map(member, [(case, b) for b in range(n)])
    c = CASES[case]
    save = {'n_members': n, 'nsteps': c['nsteps'], 'tave': c['tave'], 'ke_series': np.stack([r['ke_series'] for r in res]),
            'cfl': np.array([r['cfl'] for r in res])}
    for k in DIAGS:
        a = np.stack([r[k] for r in res])
        save[k + '_mean'] = a.mean(0)
        save[k + '_std'] = a.std(0, ddof=1)          # member-to-member spread of the time means
    save['KEspec_members'] = np.stack([r['KEspec'] for r in res]).astype('float32')   # per member: sampling error of derived spectra
    # a run that blew up (the shipped, eddy-trained CVAE on the jet configuration does, in every member, between steps
    # 24,000 and 33,500) keeps only what is meaningful: the KE series up to the blow-up and the step it happened at
    ke = save['ke_series']
    bad = ~np.isfinite(ke)
    save['first_nonfinite_step'] = np.where(bad.any(1), (bad.argmax(1) + 1) * 500, -1)
    if bad.any():
        save = {k: v for k, v in save.items() if not (isinstance(v, np.ndarray) and v.dtype.kind == 'f' and
                                                       not np.isfinite(v).all() and k != 'ke_series')}
    path = os.path.join(ROOT, 'tests', 'golden', f'oracle_stats_{case}.npz')
    np.savez_compressed(path, **save)
    print('wrote', path, 'final KE per member', save['ke_series'][:, -1])


if __name__ == '__main__':
    main()
