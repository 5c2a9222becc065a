"""GPU: time-averaged spectral diagnostics (KE-spectrum parity metric) against the oracle, and the
isotropic spectrum built from them."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import qg_ref, spectral_ref


def _ic(N, seed):
    m = qg_ref.QGModelRef(nx=N)
    rs = np.random.RandomState(seed)
    q = rs.randn(2, N, N) * np.array([8e-6, 1e-6])[:, None, None]
    return np.fft.irfftn(np.fft.rfftn(q, axes=(-2, -1)) * (m.wv < 2. / 3. * m.kk[-1]), axes=(-2, -1)) * 3.0


@pytest.mark.parametrize('N', [64, 128])
def test_time_averaged_diagnostics_match_oracle(N):
    from pyqg_generative_amd.qgmodel import QGModel
    dt = 14400. if N == 64 else 7200.
    nsteps, B = 40, 2
    kw = dict(nx=N, dt=dt, tmax=dt * nsteps, tavestart=dt * 8, taveint=dt * 3, twrite=10000)
    m = QGModel(log_level=0, n_members=B, **kw)
    q0 = np.stack([_ic(N, 1), _ic(N, 2)])
    m.q = q0
    m.run()
    refs = []
    for b in range(B):
        r = qg_ref.QGModelRef(**kw)
        r.set_q(q0[b])
        r.run()
        refs.append(r)
    assert m.diagnostics_count == refs[0].diag_count == len([t for t in range(1, nsteps) if t >= 8 and t % 3 == 0])
    for name in ('KEspec', 'Ensspec', 'entspec', 'APEflux', 'KEflux', 'APEgenspec', 'KEfrictionspec',
                 'Dissspec', 'ENSDissspec', 'ENSflux', 'ENSgenspec', 'ENSfrictionspec'):
        got = m.get_diagnostic(name)
        for b in range(B):
            ref = refs[b].get_diagnostic(name)
            assert got[b].shape == ref.shape, name
            assert np.abs(got[b] - ref).max() <= 1e-9 * np.abs(ref).max(), (name, np.abs(got[b] - ref).max() / np.abs(ref).max())
    # isotropic KE spectrum (the parity metric of the reference's notebooks) and its ensemble mean
    ke = m.ensemble_mean_diagnostic('KEspec')
    kr, sp = spectral_ref.ke_spectrum(refs[0], ke, m.delta)
    ke_ref = np.mean([r.get_diagnostic('KEspec') for r in refs], axis=0)
    kr2, sp2 = spectral_ref.ke_spectrum(refs[0], ke_ref, m.delta)
    np.testing.assert_allclose(sp, sp2, rtol=1e-9)
    m.close()


def test_paramspec_and_dataset_export():
    from pyqg_generative_amd.qgmodel import QGModel, QParameterization
    from pyqg_generative_amd.tools.simulate import snapshot_dataset
    N, dt, nsteps = 64, 14400., 12

    class Damp(QParameterization):
        def __call__(self, mm):
            return -2e-7 * np.asarray(mm.q)
    kw = dict(nx=N, dt=dt, tmax=dt * nsteps, tavestart=dt * 2, taveint=dt * 2, twrite=10000)
    m = QGModel(log_level=0, parameterization=Damp(), **kw)
    r = qg_ref.QGModelRef(parameterization=lambda mm: -2e-7 * mm.q, **kw)
    q0 = _ic(N, 3)
    m.q = q0
    r.set_q(q0)
    m.run()
    r.run()
    got, ref = m.get_diagnostic('paramspec'), r.get_diagnostic('paramspec')
    assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max()
    # its available-potential / kinetic split (summed by the reference's total energy flux,
    # comparison_tools.py:174-176): against the oracle, and the budget identity APE + KE == total
    parts = 0.0
    for name in ('paramspec_APEflux', 'paramspec_KEflux'):
        g, rr = m.get_diagnostic(name), r.get_diagnostic(name)
        assert np.abs(g - rr).max() <= 1e-9 * np.abs(rr).max(), name
        parts = parts + g
    assert np.abs(parts - got).max() <= 1e-10 * np.abs(got).max()
    # the enstrophy budget and the filter's dissipation with a parameterization in the tendency (Dissspec uses the
    # tendency the step is about to take, forcing included): all sixteen keys of comparison_tools.py:222-225
    from pyqg_generative_amd._lib import DIAGS
    assert len(DIAGS) == 16
    for name in DIAGS:
        g, rr = m.get_diagnostic(name), r.get_diagnostic(name)
        assert g.shape == rr.shape and np.abs(g - rr).max() <= 1e-9 * np.abs(rr).max(), name
    ds = snapshot_dataset(m)           # one snapshot: pyqg's layout, every variable with a length-one time axis
    assert ds['KEspec'].shape == (1, 2, N, N // 2 + 1) and ds['KEflux'].shape == (1, N, N // 2 + 1)
    for key in ('Dissspec', 'ENSDissspec', 'ENSflux', 'ENSfrictionspec', 'ENSgenspec', 'ENSparamspec'):
        assert ds[key].shape == (1, N, N // 2 + 1), key
    assert ds['q'].shape == (1, 2, N, N)
    from pyqg_generative_amd.tools.simulate import concat_in_time
    full = concat_in_time([ds, ds])    # the run's dataset: spectra from the last snapshot, no time axis
    assert full['KEspec'].shape == (2, N, N // 2 + 1) and full['paramspec_KEflux'].shape == (N, N // 2 + 1)
    m.close()


@pytest.mark.parametrize('N,fused', [(128, 1), (128, 0), (256, 1)])
def test_large_grid_increment_with_a_forcing_matches_oracle(N, fused):
    """large grids: the three-launch increment (four packed fields through fused row / column kernels, xi eliminated through
    q = xi + F (p_2 - p_1)) and the composed one (option large_fused = 0: one launch per transform) with an external forcing
    in the tendency — all sixteen diagnostics against the oracle, and ph as pyqg keeps it"""
    import pyqg_generative_amd as qa
    import pyqg_generative_amd._lib as L
    dt, nsteps, B = 3600., 7, 2
    rs = np.random.RandomState(N)
    q0 = np.stack([_ic(N, 5), _ic(N, 6)])
    Ss = [rs.randn(B, 2, N, N) * np.array([7e-12, 2e-13])[None, :, None, None] for _ in range(nsteps)]
    e = qa.EnsembleEngine(nx=N, n_members=B, dt=dt)
    e.set_option('large_fused', fused)
    e.set_q(q0)
    e.diag_config(2, 2)
    refs = []
    for b in range(B):
        it = iter([s[b] for s in Ss])
        r = qg_ref.QGModelRef(nx=N, dt=dt, tavestart=2 * dt, taveint=2 * dt, parameterization=(lambda it: lambda mm: 0.5 * next(it))(it))
        r.set_q(q0[b])
        refs.append(r)
    for s in range(nsteps):
        e.step(1, forcing=torch.as_tensor(Ss[s]).cuda(), weight=0.5, demean=False)
        for r in refs:
            r._step_forward()
    assert e.diag_count == refs[0].diag_count == 3
    for name in L.DIAGS:
        got = e.diag(name).cpu().numpy()
        for b, r in enumerate(refs):
            ref = r.get_diagnostic(name)
            assert got[b].shape == ref.shape and np.abs(got[b] - ref).max() <= 1e-9 * np.abs(ref).max(), (name, b)
    e.close()
