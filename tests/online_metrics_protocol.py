#!/usr/bin/env python
"""Helper MODULE of tests/test_gpu_online_metrics.py (not an experiment script: the test imports `run`; it scores with
oracle/metrics_ref.py, and only tests/ may import the oracle): the reference's ONLINE METRICS (Google-Colab/online-simulations.ipynb cells 6, 11-14, 29-33) from runs
of this engine: a 256 x 256 reference run coarse-grained with Operator1 to 48 x 48 (the notebook's `eddy/48/hires-sharp`),
48 x 48 runs without parameterization (`lores`) and with the shipped CGAN / CVAE / GZ models (AR1, nsteps = 1), 20 years
each, and the distributional / spectral errors of oracle/metrics_ref.py.
    python tests/online_metrics_protocol.py [members per low-resolution case] [hires members] [cases]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyqg_generative_amd import _lib as L, weights
from pyqg_generative_amd.qgmodel import QGModel
from pyqg_generative_amd.models import CGANRegression, CVAERegression, MeanVarModel
from pyqg_generative_amd.tools.operators import Dev
from pyqg_generative_amd.tools.simulate import set_initial_condition
from pyqg_generative_amd.tools.stochastic_pyqg import stochastic_QGModel
from pyqg_generative_amd.tools.parameters import EDDY_PARAMS, YEAR
from oracle import metrics_ref

PUBLISHED = {'lores': (0.1888102207415551, 0.5053847264088392), 'gan': (0.03483027178018462, 0.22129903221577354),
             'vae': (0.04144719274852606, 0.21444646848027046), 'gz': (0.20885271399081487, 0.4818579043905941)}
SPECS = ('KEspec', 'KEflux', 'APEflux', 'APEgenspec', 'KEfrictionspec', 'paramspec_KEflux', 'paramspec_APEflux')


def run_members(m, coarse=None):
    """-> list of per-member dicts: q, u, v snapshots (T,2,n,n) float32 [coarse-grained to `coarse`] + time-mean spectra"""
    B = m.n_members
    snaps = {k: [] for k in 'quv'}
    for _ in m.run_with_snapshots(tsnapint=3600000.):
        for k, f in (('q', L.F_Q), ('u', L.F_U), ('v', L.F_V)):
            a = m._eng.get(f)
            if coarse:
                a = Dev.Operator1(a.reshape(-1, m.nx, m.nx), coarse).reshape(B, 2, coarse, coarse)
            snaps[k].append(a.to(torch.float32).cpu().numpy())
    out = []
    for b in range(B):
        r = {k: np.stack([s[b] for s in v]) for k, v in snaps.items()}
        for name in SPECS:
            if name.startswith('paramspec') and m.q_parameterization is None:
                continue
            a = m.get_diagnostic(name)
            r[name] = np.asarray(a[b] if B > 1 else a)
        out.append(r)
    return out


def experiment(n_lores=8, n_hires=2, cases=('lores', 'lores14400', 'gan', 'vae', 'gz'), verbose=False):
    """-> {case: array (n_lores * n_hires, 2) of (distributional, spectral) errors}"""
    t0 = time.time()
    hp = dict(EDDY_PARAMS.nx(256)._update({'tmax': 20 * YEAR}), log_level=0)
    mh = QGModel(n_members=n_hires, **hp)
    set_initial_condition(mh, seeds=range(900, 900 + n_hires))
    hires = run_members(mh, coarse=48)
    mh.close()
    Dev.close()
    refs = []
    for r in hires:
        t = {k: r[k] for k in 'quv'}
        t.update(metrics_ref.coarsegrain_reference_spectra({k: r[k] for k in SPECS if k in r}, 48, 'Operator1'))
        refs.append(t)
    if verbose:
        print(f'hires-sharp references: {n_hires} x 256^2, 20 years, {time.time() - t0:.0f} s; {refs[0]["q"].shape[0]} snapshots')
    lp = dict(EDDY_PARAMS.nx(48)._update({'tmax': 20 * YEAR, 'dt': 7200}), log_level=0)       # the notebook's cell 11
    out = {}
    for case in cases:
        t0 = time.time()
        if case == 'lores':
            m = QGModel(n_members=n_lores, **lp)
        elif case == 'lores14400':        # the published `eddy/48/lores` dataset: the HPC default time step of 48 x 48 runs
            m = QGModel(n_members=n_lores, **dict(EDDY_PARAMS.nx(48)._update({'tmax': 20 * YEAR}), log_level=0))
        else:
            nets, xs, ys = weights.load_npz(os.path.join(ROOT, 'tests', 'golden', f'weights_{case}.npz'), case)
            model = {'gan': CGANRegression, 'vae': CVAERegression, 'gz': MeanVarModel}[case].from_arrays(nets, xs, ys)
            m = stochastic_QGModel(dict(lp, parameterization=model), 'AR1', 1, n_members=n_lores, seed=31)
        set_initial_condition(m, seeds=range(100, 100 + n_lores))
        runs = run_members(m)
        m.close()
        sc = np.array([[f(metrics_ref.diagnostic_differences(r, ref)) for f in (metrics_ref.distrib_score, metrics_ref.spectral_score)]
                       for r in runs for ref in refs])
        out[case] = sc
        if verbose:
            pd, ps = PUBLISHED['lores' if case.startswith('lores') else case]
            print(f'{case:10s} ({time.time() - t0:4.0f} s): distributional error {sc[:, 0].mean():.4f} +- {sc[:, 0].std():.4f} '
                  f'[{sc[:, 0].min():.4f}, {sc[:, 0].max():.4f}] published {pd:.4f} | spectral error {sc[:, 1].mean():.4f} +- '
                  f'{sc[:, 1].std():.4f} [{sc[:, 1].min():.4f}, {sc[:, 1].max():.4f}] published {ps:.4f}')
    return out


def main():
    nlo = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nhi = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cases = sys.argv[3].split(',') if len(sys.argv) > 3 else ('lores', 'lores14400', 'gan', 'vae', 'gz')
    experiment(nlo, nhi, cases, verbose=True)


if __name__ == '__main__':
    main()
