"""CPU tests of the dataset surface the reference consumes: pyqg's ``to_dataset`` layout,
``drop_vars`` / ``concat_in_time`` (pyqg_generative/tools/simulate.py:16-60), the product ``calc_ispec`` /
``spectrum`` (tools/spectral_tools.py:7-180) and the accesses of the reference's online metrics
(tools/comparison_tools.py:116-195) — on ``xr_lite`` and, when it is importable, on real xarray.

The model state comes from the CPU oracle here (no GPU); tests/test_gpu_facade.py runs the same flow on
the device model.
"""
import numpy as np
import pytest

from conftest import golden
from oracle import qg_ref, spectral_ref
from pyqg_generative_amd import xarray_output
from pyqg_generative_amd.tools import xr_lite, spectral_tools, simulate

BACKENDS = [xr_lite]
try:
    import xarray
    BACKENDS.append(xarray)
except ImportError:
    pass


def _oracle_snapshots(nx=32, nsnap=3, every=4, tavestart_steps=5):
    """snapshots of a short oracle run; diagnostics start being averaged after `tavestart_steps`"""
    dt = 14400.
    m = qg_ref.QGModelRef(nx=nx, dt=dt, tavestart=tavestart_steps * dt, taveint=dt, tmax=1e9)
    qg_ref.set_initial_condition(m, np.random.RandomState(3))
    m.set_q(m.q * 30)
    snaps = []
    for s in range(nsnap):
        for _ in range(every):
            m._step_forward()
        fields = dict(q=m.q.copy(), u=m.u.copy(), v=m.v.copy(), p=m.ifft(m.ph), qh=m.qh.copy(), ph=m.ph.copy(),
                      ufull=m.u + m.Ubg[:, None, None], vfull=m.v.copy(), dqhdt=m.dqhdt.copy(),
                      dqdt=m.ifft(m.dqhdt), Ubg=m.Ubg, Qy=m.Qy)
        diags = {k: v.copy() for k, v in m.diag.items() if k in xarray_output.DIAGNOSTICS}
        snaps.append((m.t, m.tc, fields, diags))
    return m, snaps


@pytest.fixture(params=BACKENDS, ids=lambda b: b.__name__.split('.')[-1])
def xr(request, monkeypatch):
    monkeypatch.setattr(simulate, 'dataset_backend', lambda: request.param)
    return request.param


def _datasets(xr, m, snaps):
    out = []
    for t, tc, fields, diags in snaps:
        m.t, m.tc = t, tc
        out.append(xarray_output.model_to_dataset(m, fields=fields, diagnostics=diags, xr=xr))
    return out


def test_to_dataset_layout_is_pyqgs(xr):
    m, snaps = _oracle_snapshots()
    ds = _datasets(xr, m, snaps)[-1]
    N, NK = m.nx, m.nk
    assert ds['q'].dims == ('time', 'lev', 'y', 'x') and ds['q'].shape == (1, 2, N, N)
    assert ds['qh'].dims == ('time', 'lev', 'l', 'k') and ds['qh'].dtype == np.complex128
    assert ds['KEspec'].dims == ('time', 'lev', 'l', 'k') and ds['APEflux'].dims == ('time', 'l', 'k')
    assert ds['Ubg'].dims == ('lev',)
    for c in ('time', 'lev', 'lev_mid', 'x', 'y', 'l', 'k'):
        assert c in ds.coords
    np.testing.assert_array_equal(np.asarray(ds['k'].values), m.kk)          # dataset.ipynb cell 8: k 0 ... 2.011e-4
    np.testing.assert_array_equal(np.asarray(ds['l'].values), m.ll)
    np.testing.assert_array_equal(np.asarray(ds['x'].values), (np.arange(N) + 0.5) * m.L / N)
    assert float(np.asarray(ds['time'].values)[0]) == m.t
    for a in ('L', 'W', 'M', 'beta', 'delta', 'del2', 'dt', 'filterfac', 'nx', 'ny', 'nz', 'nk', 'nl', 'rd', 'rek',
              'taveint', 'tavestart', 'tc', 'tmax', 'twrite'):
        assert f'pyqg:{a}' in ds.attrs, a
    assert ds.attrs['pyqg:nx'] == N and ds.attrs['pyqg:nk'] == NK
    assert ds.attrs['title'].startswith('pyqg')
    # the early snapshot (taken before tavestart) has no diagnostics
    assert 'KEspec' not in _datasets(xr, m, snaps[:1])[0].keys()


def test_drop_vars_and_concat_in_time_follow_the_reference(xr):
    m, snaps = _oracle_snapshots()
    parts = [simulate.drop_vars(d) for d in _datasets(xr, m, snaps)]
    for d in parts:
        assert set(d.keys()) >= {'q', 'u', 'v', 'psi'} and 'p' not in d.keys()
        for gone in ('qh', 'ph', 'dqhdt', 'dqdt', 'ufull', 'vfull'):
            assert gone not in d.keys()
        assert d['q'].dtype == np.float32 and d['time'].attrs['units'] == 'days'
    ds = simulate.concat_in_time(parts)
    assert ds['q'].dims == ('time', 'lev', 'y', 'x') and ds['q'].shape[0] == len(snaps)
    np.testing.assert_allclose(np.asarray(ds['time'].values), [s[0] / 86400. for s in snaps], rtol=1e-6)
    # spectral statistics come from the LAST snapshot, without a time axis (simulate.py:54-56)
    assert ds['KEspec'].dims == ('lev', 'l', 'k') and ds['KEflux'].dims == ('l', 'k')
    np.testing.assert_allclose(np.asarray(ds['KEspec'].values), snaps[-1][3]['KEspec'].astype('float32'), rtol=1e-6)
    np.testing.assert_array_equal(np.asarray(ds['psi'].values)[-1], snaps[-1][2]['p'].astype('float32'))


def _online_metric_accesses(xr, ds1, ds2, T=2):
    """The accesses of diagnostic_differences_Perezhogin (comparison_tools.py:116-195) on two run datasets,
    with the product calc_ispec; returns the spectral RMSE dictionary."""
    if 'run' not in ds1.dims:
        ds1 = ds1.expand_dims('run')
    if 'run' not in ds2.dims:
        ds2 = ds2.expand_dims('run')
    out = {}
    for z in (0, 1):
        ts = slice(-T, None)
        a = np.asarray(ds1.isel(lev=z, time=ts)['q'].values).ravel()
        b = np.asarray(ds2.isel(lev=z, time=ts)['q'].values).ravel()
        out[f'distrib_q{z + 1}'] = (a.mean() - b.mean(), float(np.sqrt(np.mean(b ** 2))))

    def spectral_rmse(spec1, spec2):
        g1, g2 = spectral_tools._Grid(spec1.shape[-2]), spectral_tools._Grid(spec2.shape[-2])
        kr1, i1 = spectral_tools.calc_ispec(g1, spec1.values)
        kr2, i2 = spectral_tools.calc_ispec(g2, spec2.values)
        return float(np.sqrt(np.mean((i1 - i2) ** 2))), float(np.sqrt(np.mean(i2 ** 2)))

    for z in (0, 1):
        out[f'KEspec{z + 1}'] = spectral_rmse(ds1['KEspec'].isel(lev=z).mean('run'), ds2['KEspec'].isel(lev=z).mean('run'))

    def compute_Eflux(ds):
        tot = 0
        for spec in ('KEflux', 'APEflux', 'paramspec_KEflux', 'paramspec_APEflux'):
            if spec in ds.data_vars:
                tot = tot + ds[spec].mean('run')
        return tot
    out['Eflux'] = spectral_rmse(compute_Eflux(ds1), compute_Eflux(ds2))
    out['APEgenspec'] = spectral_rmse(ds1['APEgenspec'].mean('run'), ds2['APEgenspec'].mean('run'))
    return out


def test_online_metric_accesses_run_on_the_dataset(xr):
    m, snaps = _oracle_snapshots()
    ds = simulate.concat_in_time([simulate.drop_vars(d) for d in _datasets(xr, m, snaps)])
    same = _online_metric_accesses(xr, ds, ds)
    for k, (diff, scale) in same.items():
        assert diff == 0 and scale > 0, k
    # two runs stacked along 'run' (the reference concatenates member files: simulate.py:282)
    both = xr.concat([ds, ds], 'run')
    assert both['q'].dims[0] == 'run' and both['KEspec'].dims == ('run', 'lev', 'l', 'k')
    again = _online_metric_accesses(xr, both, ds)
    for k, (diff, scale) in again.items():
        assert abs(diff) <= 1e-12 * scale, k
    # the isotropic KE spectrum from the dataset == the oracle's (golden-pinned) binning of the same density
    kr, sp = spectral_tools.calc_ispec(spectral_tools._Grid(m.nx), ds['KEspec'].isel(lev=0).values)
    kr0, sp0 = spectral_ref.calc_ispec(m, snaps[-1][3]['KEspec'][0].astype('float32'))
    np.testing.assert_allclose(kr, kr0, rtol=1e-15)
    np.testing.assert_allclose(sp, sp0, rtol=1e-12)


def test_product_calc_ispec_matches_reference_golden():
    g = golden('ispec.npz')
    for N in (48, 64):
        grid = spectral_tools._Grid(N)
        for av in (True, False):
            for tr in (True, False):
                kr, ph = spectral_tools.calc_ispec(grid, g[f'dens_{N}'], averaging=av, truncate=tr)
                np.testing.assert_allclose(kr, g[f'kr_{N}_{int(av)}{int(tr)}'], rtol=1e-15)
                np.testing.assert_allclose(ph, g[f'ph_{N}_{int(av)}{int(tr)}'], rtol=1e-13)
        # stacked densities are binned in one pass; non-dimensional wavenumbers preserve the integral
        d = np.stack([g[f'dens_{N}'], 2 * g[f'dens_{N}']])
        kr, ph = spectral_tools.calc_ispec(grid, d)
        np.testing.assert_allclose(ph[1], 2 * ph[0], rtol=1e-15)
        krn, phn = spectral_tools.calc_ispec(grid, d, nd_wavenumber=True)
        np.testing.assert_allclose((phn * (krn[1] - krn[0])).sum(), (ph * (kr[1] - kr[0])).sum(), rtol=1e-13)


def test_spectrum_satisfies_parseval(xr):
    """the check the reference's spectrum.test performs (spectral_tools.py:19-44): with summation over
    half-open rings up to the outer circle the spectrum integrates to the variance of the field"""
    rs = np.random.RandomState(0)
    N = 32
    x = rs.randn(2, 3, 2, N, N)                    # (run, time, lev, y, x)
    x -= x.mean(axis=(-2, -1), keepdims=True)
    da = xr.DataArray(x, dims=['run', 'time', 'lev', 'y', 'x'])
    sp = spectral_tools.spectrum(type='power', averaging=False, truncate=False, time=slice(0, None))(da)
    assert sp.dims == ('lev', 'k')
    k = np.asarray(sp['k'].values)
    dk = np.sqrt(2) * 2 * np.pi / 1e6
    var = (x ** 2).mean(axis=(0, 1, 3, 4))
    # the outermost ring [kmax - dkr, kmax) is left out by the bin edges (np.arange(kmin, kmax - dkr, dkr))
    assert np.all(np.asarray(sp.values).sum(axis=1) * dk <= var * (1 + 1e-12))
    assert np.all(np.asarray(sp.values).sum(axis=1) * dk >= 0.97 * var)
    np.testing.assert_allclose(np.diff(k), dk, rtol=1e-12)
    cross = spectral_tools.spectrum(type='cross_layer', averaging=False, truncate=False, time=slice(0, None))(da)
    assert cross.dims == ('k',)
