"""CPU: the oracle restatements against the golden vectors captured from the
reference's own code (tests/golden/make_golden.py)."""
import json
import os
import numpy as np
import pytest
from conftest import golden, load_generator, GOLDEN

from oracle import gen_ref, samplers_ref, operators_ref, spectral_ref, qg_ref


@pytest.mark.parametrize('kind', ['gan', 'vae', 'gz'])
@pytest.mark.parametrize('N', [48, 64, 96])
def test_generator_matches_reference(kind, N):
    g = golden('generator.npz')
    gen = load_generator(kind)
    q = g[f'{kind}_{N}_q'].astype('float64')
    z = g[f'{kind}_{N}_z']
    Sraw = gen.predict_snapshot(q, z)
    S = gen_ref.demean(Sraw)
    ref_raw, ref = g[f'{kind}_{N}_Sraw'], g[f'{kind}_{N}_S']
    assert S.dtype == np.float64 and S.shape == (2, N, N)
    # same torch-CPU kernels underneath: agreement to f32 rounding of the stack
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    assert np.abs(Sraw - ref_raw).max() <= 2e-6 * np.abs(ref_raw).max()
    assert (np.abs(S - ref) / scale).max() <= 2e-6
    assert np.abs(S.mean(axis=(1, 2))).max() <= 1e-6 * scale.max()


@pytest.mark.parametrize('kind', ['gan', 'vae'])
@pytest.mark.parametrize('N', [48, 64, 96])
def test_generator_with_a_regression_net_matches_reference(kind, N):
    """regression != 'None' (cgan_regression.py:159-162, cvae_regression.py:133-136): Y += net_mean(X) before the
    de-normalisation; vectors from the reference's own classes (tests/golden/make_golden_regression.py)."""
    g = golden('generator_regression.npz')
    gen = load_generator(kind, regression=True)
    q = g[f'{kind}_{N}_q'].astype('float64')
    Sraw = gen.predict_snapshot(q, g[f'{kind}_{N}_z'])
    S = gen_ref.demean(Sraw)
    ref_raw, ref = g[f'{kind}_{N}_Sraw'], g[f'{kind}_{N}_S']
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    assert np.abs(Sraw - ref_raw).max() <= 2e-6 * np.abs(ref_raw).max()
    assert (np.abs(S - ref) / scale).max() <= 2e-6
    # and the regression net does matter in this fixture: without it the answer is far off
    plain = load_generator(kind).predict_snapshot(q, g[f'{kind}_{N}_z'])
    assert np.abs(plain - ref_raw).max() > 1e-2 * np.abs(ref_raw).max()
    if N == 64:                                   # predict_mean_snapshot (cgan_regression.py:164-171), M = 6 given latent fields
        mean6 = gen.predict_mean_snapshot(q, M=6, z=g[f'{kind}_64_mean6_z'])
        assert np.abs(mean6 - g[f'{kind}_64_mean6']).max() <= 2e-6 * np.abs(g[f'{kind}_64_mean6']).max()


def test_layer_activations_match_reference():
    g = golden('layers.npz')
    w = gen_ref.CNNWeights.from_npz_dict(golden('weights_gan.npz'), 'net0_')
    out, layers = gen_ref.cnn_forward(w, g['x'], return_layers=True)
    assert len(layers) == 8
    for i, a in enumerate(layers):
        ref = g[f'act{i}']
        assert a.shape == ref.shape
        assert np.abs(a - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())


def test_samplers_match_reference():
    g = golden('samplers.npz')
    for kind, cls, ns in (('ar1', samplers_ref.AR1SamplerRef, (1, 5, -1)),
                          ('const', samplers_ref.ConstantSamplerRef, (1, 3))):
        for n in ns:
            rs = np.random.RandomState(11)
            s = cls(n)
            for t in range(8):
                flag = s.update(lambda: rs.randn(6))
                assert bool(flag) == bool(g[f'{kind}_{n}_flags'][t])
                np.testing.assert_array_equal(s.noise, g[f'{kind}_{n}_seq'][t])


def test_operators_match_reference():
    g = golden('operators.npz')
    X = g['X']
    for key, val in (('cut_off_32', operators_ref.cut_off(X, 32)),
                     ('cut_off_48', operators_ref.cut_off(X, 48)),
                     ('clean_2h', operators_ref.clean_2h(X)),
                     ('interp_64_96', operators_ref.fft_interpolate(X, 64, 96)),
                     ('interp_64_32', operators_ref.fft_interpolate(X, 64, 32)),
                     ('interp_64_96_keep2h', operators_ref.fft_interpolate(X, 64, 96, truncate_2h=False)),
                     ('op5_32', operators_ref.Operator5(X, 32))):
        assert val.shape == g[key].shape, key
        assert np.abs(val - g[key]).max() <= 1e-13 * np.abs(g[key]).max(), key


def test_operator_notebook_identities():
    # notebooks/3-2-dealiasing.ipynb cells 17-26: exact re-gridding of a resolved wave
    # (grids np.linspace(0,2pi,n+1)[:-1]); cut_off(x,16) == fft_interpolate(x,64,16) exactly
    def wave(n):
        x = np.linspace(0, 2 * np.pi, n + 1)[:-1]
        X, Y = np.meshgrid(x, x)
        return np.cos(X) * np.sin(Y)
    Z = wave(16)
    assert np.linalg.norm(wave(24) - operators_ref.fft_interpolate(Z, 16, 24)) < 5e-14
    assert np.linalg.norm(wave(8) - operators_ref.fft_interpolate(Z, 16, 8)) < 5e-15
    assert np.linalg.norm(Z - operators_ref.fft_interpolate(Z, 16, 16)) < 5e-15
    rs = np.random.RandomState(0)
    X = rs.randn(64, 64)
    assert np.abs(operators_ref.cut_off(X, 16) - operators_ref.fft_interpolate(X, 64, 16)).max() == 0.0
    # cells 30-32: truncation commutes with the spectral divergence
    u, v = rs.randn(2, 64, 64), rs.randn(2, 64, 64)
    for f in (lambda a: operators_ref.fft_interpolate(a, 64, 8), lambda a: operators_ref.cut_off(a, 8)):
        lhs = f(operators_ref.divergence(u, v))
        rhs = operators_ref.divergence(f(u), f(v))
        assert np.linalg.norm(lhs - rhs) < 1e-17


def test_advection_invariants_and_sgs_identity():
    # notebooks/3-2-dealiasing.ipynb cells 9-13 and 48-51 on a synthetic hires PV field
    rs = np.random.RandomState(5)
    m = qg_ref.QGModelRef(nx=128)
    qh = m.fft(rs.randn(2, 128, 128) * np.array([8e-6, 1e-6])[:, None, None])
    qh *= (m.wv < 0.5 * m.wv.max() / np.sqrt(2))       # smooth, well resolved
    m.set_qh(qh)
    m._invert()
    psi = m.ifft(m.ph)
    norm = lambda a, b: (a * b).mean(axis=(-2, -1)) / (a.std(axis=(-2, -1)) * b.std(axis=(-2, -1)))
    for rule in ('2/3-rule', '3/2-rule'):
        d = operators_ref.advect(m.q, m.u, m.v, dealias=rule)
        assert np.abs(d.mean(axis=(-2, -1)) / d.std(axis=(-2, -1))).max() < 1e-14
        assert np.abs(norm(d, m.q)).max() < 1e-13        # enstrophy conserved
        assert np.abs(norm(d, psi)).max() < 1e-13        # energy conserved
    params = {}
    SGS, mf, mm = operators_ref.PV_subgrid_forcing(m.q, 64, operators_ref.Operator5, params, '3/2-rule')
    advf = -operators_ref.advect(mf.q, mf.u, mf.v, dealias='3/2-rule')
    adv = -operators_ref.cut_off(operators_ref.advect(mm.q, mm.u, mm.v, dealias='3/2-rule'), 64)
    assert np.linalg.norm(adv - (SGS + advf)) / np.linalg.norm(adv) < 1e-14


def test_ispec_matches_reference():
    g = golden('ispec.npz')
    for N in (48, 64):
        grid = qg_ref.QGModelRef(nx=N)
        for av in (True, False):
            for tr in (True, False):
                kr, ph = spectral_ref.calc_ispec(grid, g[f'dens_{N}'], averaging=av, truncate=tr)
                np.testing.assert_allclose(kr, g[f'kr_{N}_{int(av)}{int(tr)}'], rtol=1e-15)
                np.testing.assert_allclose(ph, g[f'ph_{N}_{int(av)}{int(tr)}'], rtol=1e-13)


def test_initial_condition_matches_reference():
    g = golden('initial_condition.npz')
    for N in (48, 64, 96):
        m = qg_ref.QGModelRef(nx=N)
        qg_ref.set_initial_condition(m, np.random.RandomState(N))
        np.testing.assert_allclose(m.q[0], g[f'q1_{N}'], rtol=0, atol=1e-22)
        assert np.abs(m.q[1]).max() == 0 and np.abs(g[f'q2_{N}']).max() == 0


def test_parameters_constants():
    with open(os.path.join(GOLDEN, 'parameters.json')) as f:
        c = json.load(f)
    assert c['YEAR'] == 360 * 86400 and c['ANDREW_1000_STEPS'] == 3600000
    assert c['dt'] == {'32': 14400, '48': 14400, '64': 14400, '96': 7200, '128': 7200,
                       '256': 3600, '512': 1800}
    assert c['JET']['rek'] == 7e-08 and c['JET']['delta'] == 0.1 and c['JET']['beta'] == 1e-11
