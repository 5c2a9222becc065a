"""CPU: host-side logic of the package (no GPU compute): run configurations, samplers,
weight loading, the dataset container, ensemble sharding."""
import json
import os
import numpy as np
import pytest
from conftest import golden, GOLDEN


def test_parameters_match_reference_constants():
    from pyqg_generative_amd.tools import parameters as par
    with open(os.path.join(GOLDEN, 'parameters.json')) as f:
        c = json.load(f)
    assert par.YEAR == c['YEAR'] and par.DAY == c['DAY'] and par.ANDREW_1000_STEPS == c['ANDREW_1000_STEPS']
    assert dict(par.EDDY_PARAMS) == c['EDDY'] and dict(par.JET_PARAMS) == c['JET']
    for n, dt in c['dt'].items():
        assert par.EDDY_PARAMS.nx(int(n))['dt'] == dt
    jet96 = par.JET_PARAMS.nx(96)
    assert jet96['dt'] == 7200 and jet96['rek'] == 7e-8 and par.JET_PARAMS['nx'] == 64


def test_host_samplers_match_reference_sequences():
    from pyqg_generative_amd.tools.stochastic_pyqg import AR1_sampler, constant_sampler
    g = golden('samplers.npz')
    for kind, cls, ns in (('ar1', AR1_sampler, (1, 5, -1)), ('const', constant_sampler, (1, 3))):
        for n in ns:
            rs = np.random.RandomState(11)
            s = cls(n)
            for t in range(8):
                flag = s.update(lambda: rs.randn(6))
                assert bool(flag) == bool(g[f'{kind}_{n}_flags'][t])
                np.testing.assert_array_equal(s.noise, g[f'{kind}_{n}_seq'][t])


def test_weight_fixture_loader_and_scaler():
    from pyqg_generative_amd import weights
    from pyqg_generative_amd.tools.cnn_tools import ChannelwiseScaler
    nets, xs, ys = weights.load_npz(os.path.join(GOLDEN, 'weights_gz.npz'), 'gz')
    assert len(nets) == 2 and nets[0]['conv_w'][0].shape == (128, 2, 5, 5)
    assert nets[1]['conv_w'][7].shape == (2, 32, 3, 3) and len(nets[0]['bn_v']) == 7
    np.testing.assert_allclose(xs, [7.784383342368528e-06, 1.0471941322975908e-06], rtol=1e-7)
    np.testing.assert_allclose(ys, [7.60611105349307e-12, 1.656513061486578e-13], rtol=1e-7)
    sc = ChannelwiseScaler(xs)
    X = np.ones((1, 2, 3, 3), np.float32)
    assert sc.normalize(X).dtype == np.float32
    np.testing.assert_allclose(sc.denormalize(sc.normalize(X)), X, rtol=1e-6)
    syn, _, _ = weights.synthetic('gan', seed=1)
    assert syn[0]['conv_w'][1].shape == (64, 128, 5, 5)


def test_scaler_json_roundtrip(tmp_path):
    from pyqg_generative_amd.tools.cnn_tools import ChannelwiseScaler
    sc = ChannelwiseScaler([2.0, 4.0])
    sc.write('x_scale.json', folder=str(tmp_path))
    rd = ChannelwiseScaler().read('x_scale.json', folder=str(tmp_path))
    np.testing.assert_array_equal(rd.std, sc.std)
    from pyqg_generative_amd.weights import read_scaler_std
    np.testing.assert_array_equal(read_scaler_std(str(tmp_path / 'x_scale.json')), [2.0, 4.0])


def test_xr_lite_follows_xarray_semantics(tmp_path):
    """the stand-in used when xarray is absent: the xarray behaviours the snapshot flow relies on"""
    from pyqg_generative_amd.tools import xr_lite as xr
    def snap(t):
        return xr.Dataset({'q': (('time', 'lev', 'y', 'x'), np.full((1, 2, 4, 4), t, np.float32)),
                           'KEspec': (('lev', 'l', 'k'), np.full((2, 4, 3), t))},
                          coords={'time': (('time',), np.array([t], np.float32)), 'k': (('k',), np.arange(3.))})
    ds = xr.concat([snap(1.), snap(2.), snap(3.)], dim='time')
    # variables without the concat dimension are broadcast along it (xarray's data_vars='all')
    assert ds['q'].shape == (3, 2, 4, 4) and ds['KEspec'].dims == ('time', 'lev', 'l', 'k')
    assert ds['KEspec'].isel(time=-1).values[0, 0, 0] == 3.0
    np.testing.assert_array_equal(ds['time'].values, [1, 2, 3])
    assert ds.q.isel(time=-1).shape == (2, 4, 4)
    assert list(ds.keys()) == ['q', 'KEspec'] and 'k' in ds.variables and 'k' not in ds.keys()
    with pytest.raises(ValueError):
        ds['k'].isel(time=0)                               # missing dimensions raise, as in xarray
    ds2 = ds.rename({'q': 'psi'}).drop_vars('KEspec').astype('float32')
    assert 'psi' in ds2 and 'KEspec' not in ds2 and 'q' in ds
    stacked = xr.concat([ds, ds], 'run')                   # a new dimension goes first
    assert stacked['q'].dims == ('run', 'time', 'lev', 'y', 'x')
    assert (stacked['q'].mean('run') - ds['q']).values.max() == 0
    ds.to_netcdf(str(tmp_path / 'a.nc'))
    from scipy.io import netcdf_file
    with netcdf_file(str(tmp_path / 'a.nc'), 'r', mmap=False) as f:
        assert f.variables['q'].shape == (3, 2, 4, 4)


def test_shard_members_partitions_every_member_once():
    from pyqg_generative_amd.parallel import shard_members
    for total, world in ((1024, 8), (256, 8), (10, 4), (3, 8), (0, 2)):
        blocks = [shard_members(total, r, world) for r in range(world)]
        ids = [i for first, n in blocks for i in range(first, first + n)]
        assert ids == list(range(total))
        assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1
    assert shard_members(1024, 3, 8) == (384, 128)
    with pytest.raises(ValueError):
        shard_members(8, 8, 8)


def test_product_set_initial_condition_matches_reference_golden():
    """the product's own set_initial_condition (host numpy, tools/simulate.py) against the reference's output for the
    same global numpy seed (tests/golden/make_golden.py G7: np.random.seed(N), duck-typed model) — directly, not through
    a trajectory; and the per-member seeded form draws the same numbers from RandomState(seed)."""
    from pyqg_generative_amd.tools.simulate import set_initial_condition
    g = golden('initial_condition.npz')

    class _M:
        inverted = 0

        def _invert(self):
            self.inverted += 1
    for N in (48, 64, 96):
        m = _M()
        m.nx = m.ny = N
        m.L = 1e6
        dk = 2 * np.pi / m.L
        ll = dk * np.append(np.arange(0., N / 2), np.arange(-N / 2, 0.))
        kk = dk * np.arange(0., N // 2 + 1)
        kx, ly = np.meshgrid(kk, ll)
        m.wv = np.sqrt(kx ** 2 + ly ** 2)
        np.random.seed(N)
        set_initial_condition(m)                     # seeds=None: numpy's global stream, the reference's draw order
        assert m.q.shape == (2, N, N) and m.inverted == 1
        np.testing.assert_allclose(m.q[0], g[f'q1_{N}'], rtol=0, atol=1e-22)
        assert np.abs(m.q[1]).max() == 0
        m2 = _M()
        m2.__dict__.update(nx=N, ny=N, L=m.L, wv=m.wv, n_members=3)
        set_initial_condition(m2, seeds=[7, N, 9])   # member 1 draws from RandomState(N): the same field
        assert m2.q.shape == (3, 2, N, N)
        np.testing.assert_allclose(m2.q[1, 0], g[f'q1_{N}'], rtol=0, atol=1e-22)
        assert np.abs(m2.q[0, 0] - m2.q[1, 0]).max() > 1e-9
