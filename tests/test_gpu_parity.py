"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (stated per test):
  * spectral core, float64: relative 1e-12 of the field maximum (FFT butterflies are
    ordered differently from pocketfft, so bit equality is not expected)
  * generator, float32 arithmetic: 2e-5 of the field maximum against the golden
    vectors captured from the reference (different summation order than MIOpen/oneDNN)
"""
import os
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from conftest import golden, load_generator, GOLDEN
from oracle import qg_ref, gen_ref, samplers_ref

F64_TOL = 1e-12


def _engine(N, B, **kw):
    import pyqg_generative_amd as qa
    return qa.EnsembleEngine(nx=N, n_members=B, **kw)


def _gpu_generator(kind):
    """kind 'gan' | 'vae' | 'gz', or 'gan+reg' | 'vae+reg': with GZ's net_mean as the regression net (regression != 'None'),
    the combination tests/golden/make_golden_regression.py ran through the reference"""
    import pyqg_generative_amd as qa
    from pyqg_generative_amd import weights
    reg = kind.endswith('+reg')
    kind = kind[:-4] if reg else kind
    nets, xs, ys = weights.load_npz(os.path.join(GOLDEN, f'weights_{kind}.npz'), kind,
                                    regression_npz=os.path.join(GOLDEN, 'weights_gz.npz') if reg else None)
    return qa.Generator(kind, nets, xs, ys)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _random_q(rs, B, N):
    return rs.randn(B, 2, N, N) * np.array([8e-6, 1e-6])[None, :, None, None]


def _eddy_like_q(rs, B, N):
    """smooth fields with eddy-like amplitudes (white noise band-limited to 2/3 Nyquist)"""
    m = qg_ref.QGModelRef(nx=N)
    q = _random_q(rs, B, N)
    qh = np.fft.rfftn(q, axes=(-2, -1)) * (m.wv < 2. / 3. * m.kk[-1])
    return np.fft.irfftn(qh, axes=(-2, -1)) * 3.0


@pytest.mark.parametrize('N', [32, 48, 64, 96, 128, 192, 256])
def test_q_qh_roundtrip_and_invert(N):
    import pyqg_generative_amd._lib as L
    B = 3
    rs = np.random.RandomState(N)
    q = _random_q(rs, B, N)            # white noise: exercises the Nyquist rows/columns too
    e = _engine(N, B)
    e.set_q(q)
    qh = e.get(L.F_QH).cpu().numpy()
    ref = np.fft.rfftn(q, axes=(-2, -1))
    assert _rel(qh, ref) < F64_TOL
    e.invert()
    ph, u, v = (e.get(f).cpu().numpy() for f in (L.F_PH, L.F_U, L.F_V))
    for b in range(B):
        m = qg_ref.QGModelRef(nx=N)
        m.set_q(q[b])
        m._invert()
        assert _rel(ph[b], m.ph) < F64_TOL
        assert _rel(u[b], m.u) < F64_TOL and _rel(v[b], m.v) < F64_TOL
    # qh setter refreshes q
    e.set_qh(ref)
    assert _rel(e.get(L.F_Q).cpu().numpy(), q) < F64_TOL
    # grid tables, bit for bit against the oracle's numpy expressions
    m = qg_ref.QGModelRef(nx=N)
    np.testing.assert_array_equal(e.table(L.T_KK), m.kk)
    np.testing.assert_array_equal(e.table(L.T_LL), m.ll)
    np.testing.assert_array_equal(e.table(L.T_WV2), m.wv2)
    np.testing.assert_array_equal(e.table(L.T_A), m.a)
    # exp(-23.6 x^4) with x^4 from libm pow vs numpy power: ulp differences amplified by the exponent
    np.testing.assert_allclose(e.table(L.T_FILTR), m.filtr, rtol=1e-12, atol=0)


@pytest.mark.parametrize('N,params', [(64, dict(dt=14400.)), (48, dict(dt=14400.)),
                                      (96, dict(dt=7200., rek=7e-8, delta=0.1, beta=1e-11)),
                                      (32, dict(dt=14400.)), (128, dict(dt=7200.)),
                                      (256, dict(dt=3600.)), (384, dict(dt=3600.))])
def test_unparameterized_steps_match_oracle(N, params):
    """configs[0] physics on the GPU: Euler -> AB2 -> AB3 start-up, filter, friction."""
    import pyqg_generative_amd._lib as L
    B, nsteps = (3, 12) if N <= 128 else (2, 6)
    rs = np.random.RandomState(100 + N)
    q0 = _eddy_like_q(rs, B, N)
    e = _engine(N, B, **params)
    e.set_q(q0)
    refs = []
    for b in range(B):
        m = qg_ref.QGModelRef(nx=N, **params)
        m.set_q(q0[b])
        refs.append(m)
    for s in range(nsteps):
        e.step(1)
        for m in refs:
            m._step_forward()
        qh = e.get(L.F_QH).cpu().numpy()
        q = e.get(L.F_Q).cpu().numpy()
        for b, m in enumerate(refs):
            assert _rel(qh[b], m.qh) < F64_TOL * (s + 1), (s, b)
            assert _rel(q[b], m.q) < F64_TOL * (s + 1), (s, b)
    # fields pyqg keeps from the last inversion + tendencies
    ph, u, v = (e.get(f).cpu().numpy() for f in (L.F_PH, L.F_U, L.F_V))
    dq = e.get(L.F_DQHDT).cpu().numpy()
    dqpp = e.get(L.F_DQHDT_PP).cpu().numpy()
    for b, m in enumerate(refs):
        assert _rel(ph[b], m.ph) < 1e-11 and _rel(u[b], m.u) < 1e-11 and _rel(v[b], m.v) < 1e-11
        assert _rel(dq[b], m.dqhdt_p) < 1e-10 and _rel(dqpp[b], m.dqhdt_pp) < 1e-10
    ke, cfl = e.status()
    for b, m in enumerate(refs):
        assert abs(ke[b] - m._calc_ke()) < 1e-11 * m._calc_ke()
        assert abs(cfl[b] - m._calc_cfl()) < 1e-11
    assert e.tc == nsteps


def test_many_steps_in_one_call_equals_single_steps():
    import pyqg_generative_amd._lib as L
    N, B = 64, 2
    q0 = _eddy_like_q(np.random.RandomState(5), B, N)
    e1, e2 = _engine(N, B, dt=14400.), _engine(N, B, dt=14400.)
    e1.set_q(q0)
    e2.set_q(q0)
    e1.step(25)
    for _ in range(25):
        e2.step(1)
    assert torch.equal(e1.get(L.F_QH), e2.get(L.F_QH))       # deterministic: bit-identical


@pytest.mark.parametrize('N', [64, 128, 256])
def test_external_forcing_matches_oracle_q_parameterization(N):
    import pyqg_generative_amd._lib as L
    B = 2
    rs = np.random.RandomState(9)
    q0 = _eddy_like_q(rs, B, N)
    Ss = [rs.randn(B, 2, N, N) * np.array([7e-12, 2e-13])[None, :, None, None] for _ in range(4)]
    e = _engine(N, B, dt=14400.)
    e.set_q(q0)
    refs = []
    for b in range(B):
        it = iter([s[b] for s in Ss])
        m = qg_ref.QGModelRef(nx=N, dt=14400., parameterization=(lambda it: lambda mm: 0.5 * next(it))(it))
        m.set_q(q0[b])
        refs.append(m)
    for s in range(4):
        e.step(1, forcing=torch.as_tensor(Ss[s]).cuda(), weight=0.5, demean=False)
        for m in refs:
            m._step_forward()
    qh = e.get(L.F_QH).cpu().numpy()
    for b, m in enumerate(refs):
        assert _rel(qh[b], m.qh) < 1e-11


@pytest.mark.parametrize('kind', ['gan', 'vae', 'gz'])
@pytest.mark.parametrize('N', [48, 64, 96])
def test_generator_matches_reference_golden(kind, N):
    g = golden('generator.npz')
    gen = _gpu_generator(kind)
    q = torch.as_tensor(g[f'{kind}_{N}_q'].astype('float64')[None]).cuda().contiguous()
    z = g[f'{kind}_{N}_z']
    z = torch.as_tensor(z.reshape(1, 2, N, N)).cuda().contiguous()
    S = gen.forward(q, z, demean=True).cpu().numpy()[0]
    Sraw = gen.forward(q, z, demean=False).cpu().numpy()[0]
    ref, ref_raw = g[f'{kind}_{N}_S'], g[f'{kind}_{N}_Sraw']
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    assert (np.abs(Sraw - ref_raw) / scale).max() < 2e-5
    assert (np.abs(S - ref) / scale).max() < 2e-5
    assert np.abs(S.mean(axis=(1, 2))).max() < 1e-14 * scale.max() * N * N


@pytest.mark.parametrize('kind', ['gan', 'vae', 'gz'])
@pytest.mark.parametrize('N', [48, 64, 96])
def test_generator_matches_reference_golden_on_the_ensemble_kernels(kind, N):
    """The same reference-generated vectors on the kernels an ENSEMBLE takes by default (the golden test above runs one
    member, i.e. the split-K 25-tap kernels): the golden member sits at three positions of a 40-member ensemble padded
    with other fields, so the 5x5 layer runs as the 1-D Winograd kernel wherever calibration admitted it at this size
    (k_convw2 / k_convw), the 3x3 layers as the fused pairs.  Same 2e-5 tolerance as the single member
    (parameterization.py:23-34, cgan_regression.py:157-162)."""
    g = golden('generator.npz')
    gen = _gpu_generator(kind)
    B = 40
    rs = np.random.RandomState(11)
    q0 = g[f'{kind}_{N}_q'].astype('float64')
    z0 = g[f'{kind}_{N}_z'].reshape(2, N, N)
    q = rs.randn(B, 2, N, N) * np.array([7.8e-6, 1.05e-6]).reshape(1, 2, 1, 1)
    z = rs.randn(B, 2, N, N).astype(z0.dtype)
    where = (0, 17, B - 1)
    for b in where:
        q[b], z[b] = q0, z0
    qd, zd = torch.as_tensor(q).cuda().contiguous(), torch.as_tensor(z).cuda().contiguous()
    k = gen.layer2_kernel(B, N)
    info = gen.wino_info(N)
    assert (k >= 3) == info['enabled'], (k, info)
    S = gen.forward(qd, zd, demean=True).cpu().numpy()
    Sraw = gen.forward(qd, zd, demean=False).cpu().numpy()
    ref, ref_raw = g[f'{kind}_{N}_S'], g[f'{kind}_{N}_Sraw']
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    errs = [(np.abs(Sraw[b] - ref_raw) / scale).max() for b in where]
    print(f'\n{kind} {N}x{N}, 40 members: layer 2 = {gen.LAYER2_KERNELS[k]} {info}; max error / max|S| {max(errs):.2e}')
    for b in where:
        assert (np.abs(Sraw[b] - ref_raw) / scale).max() < 2e-5
        assert (np.abs(S[b] - ref) / scale).max() < 2e-5
    assert np.array_equal(Sraw[where[0]], Sraw[where[1]]) and np.array_equal(Sraw[where[0]], Sraw[where[2]])


@pytest.mark.parametrize('kind', ['gan', 'vae'])
@pytest.mark.parametrize('N', [48, 64, 96])
def test_generator_with_a_regression_net_matches_reference_golden(kind, N):
    """regression != 'None' (cgan_regression.py:59-60,157-162, cvae_regression.py:49-50,131-136): S = y_std (G([x, z]) +
    net_mean(x)), the sum in float32.  Vectors from the reference's own classes (tests/golden/make_golden_regression.py), on
    one member (split-K kernels) and at three positions of a 40-member ensemble (the default ensemble kernels); same 2e-5
    tolerance as the generators without a regression net."""
    g = golden('generator_regression.npz')
    gen = _gpu_generator(kind + '+reg')
    q0 = g[f'{kind}_{N}_q'].astype('float64')
    z0 = g[f'{kind}_{N}_z'].reshape(2, N, N)
    ref, ref_raw = g[f'{kind}_{N}_S'], g[f'{kind}_{N}_Sraw']
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    B = 40
    rs = np.random.RandomState(12)
    q = rs.randn(B, 2, N, N) * np.array([7.8e-6, 1.05e-6]).reshape(1, 2, 1, 1)
    z = rs.randn(B, 2, N, N).astype('float32')
    where = (0, 23, B - 1)
    for b in where:
        q[b], z[b] = q0, z0
    for sl in (slice(0, 1), slice(0, B)):
        qd, zd = torch.as_tensor(q[sl]).cuda().contiguous(), torch.as_tensor(z[sl]).cuda().contiguous()
        S = gen.forward(qd, zd, demean=True).cpu().numpy()
        Sraw = gen.forward(qd, zd, demean=False).cpu().numpy()
        for b in [w for w in where if w < S.shape[0]]:
            assert (np.abs(Sraw[b] - ref_raw) / scale).max() < 2e-5
            assert (np.abs(S[b] - ref) / scale).max() < 2e-5
    # the regression net is not a rounding-level term of this fixture
    plain = _gpu_generator(kind).forward(qd[:1], zd[:1], demean=False).cpu().numpy()[0]
    assert (np.abs(plain - ref_raw) / scale).max() > 1e-2


def test_cnn_layers_match_reference_batched():
    """Batched raw CNN forward (B=5, N=32) against the oracle's torch-CPU restatement."""
    gen = _gpu_generator('gan')
    ora = load_generator('gan')
    rs = np.random.RandomState(2)
    x = rs.randn(5, 4, 32, 32).astype('float32')
    y = gen.cnn_forward(torch.as_tensor(x).cuda()).cpu().numpy()
    ref = gen_ref.cnn_forward(ora.nets[0], x)
    assert np.abs(y - ref).max() < 2e-5 * np.abs(ref).max()


def test_philox_noise_matches_oracle_stream():
    import ctypes as C
    from pyqg_generative_amd._lib import lib, check
    B, n = 3, 2 * 16 * 16
    z = torch.zeros((B, n), dtype=torch.float32, device='cuda')
    check(lib.qgx_noise_normal(C.c_void_p(z.data_ptr()), 0, B, n, 0x1234567890ABCDEF, 10, 7, 0.0, 1.0, None))
    torch.cuda.synchronize()
    z = z.cpu().numpy()
    for b in range(B):
        ref, _ = samplers_ref.philox_normal(0x1234567890ABCDEF, 10 + b, 7, n)
        assert np.abs(z[b] - ref).max() < 2e-5       # device logf/sincosf vs numpy: few ulp
    # AR1 update z <- a z + b xi in float32, double flavour, distribution
    zd = torch.zeros((1, 1 << 18), dtype=torch.float64, device='cuda')
    check(lib.qgx_noise_normal(C.c_void_p(zd.data_ptr()), 1, 1, 1 << 18, 1, 0, 0, 0.0, 1.0, None))
    torch.cuda.synchronize()
    s = zd.cpu().numpy().ravel()
    assert abs(s.mean()) < 0.01 and abs(s.std() - 1) < 0.01 and abs((s ** 4).mean() - 3) < 0.1


JET = dict(dt=7200., rek=7e-8, delta=0.1, beta=1e-11)      # tools/parameters.py:26-27,37


@pytest.mark.parametrize('kind,sampling,nd,N,B,nsteps,params', [
    ('gan', 'AR1', 1, 64, 2, 5, dict(dt=14400.)),
    ('vae', 'AR1', 3, 64, 2, 5, dict(dt=14400.)),
    ('gz', 'constant', 2, 64, 2, 5, dict(dt=14400.)),
    ('gan', 'constant', 1, 64, 2, 5, dict(dt=14400.)),
    ('gan', 'constant', 1, 64, 1, 5, dict(dt=14400.)),      # BASELINE configs[1] exactly: one member
    ('vae', 'constant', 1, 96, 32, 3, JET),                  # BASELINE configs[3]'s per-GPU shard
    ('vae', 'AR1', 2, 96, 3, 4, JET),
    ('gan', 'AR1', 1, 48, 2, 4, dict(dt=14400.)),           # the notebooks' resolution
    ('vae', 'AR1', -1, 64, 2, 4, dict(dt=14400.)),          # nsteps < 0: the first draw is frozen (stochastic_pyqg.py:42-47)
    ('gan+reg', 'AR1', 1, 64, 2, 4, dict(dt=14400.)),       # regression != 'None': S = y_std (G + net_mean) (cgan_regression.py:159-162)
    ('vae+reg', 'constant', 2, 96, 3, 4, JET),
], ids=lambda v: str(v) if not isinstance(v, dict) else ('jet' if 'rek' in v else 'eddy'))
def test_parameterized_steps_match_oracle_with_external_noise(kind, sampling, nd, N, B, nsteps, params):
    """configs[1] (64x64 eddy + CGAN, B=1) and configs[3]'s shard (96x96 jet + CVAE, B=32): the full
    online step = sampler + generator + de-mean + spectral step, with the white noise xi supplied
    externally so that both sides see identical draws."""
    import pyqg_generative_amd._lib as L
    rs = np.random.RandomState(77)
    q0 = _eddy_like_q(rs, B, N)
    gen = _gpu_generator(kind)
    ora = load_generator(kind)
    e = _engine(N, B, **params)
    e.set_q(q0)
    if kind == 'gz':
        xis = [rs.randn(B, 2, N, N) for _ in range(nsteps)]
    else:
        xis = [rs.randn(B, 1, 2, N, N).astype('float32') for _ in range(nsteps)]
    refs = []
    for b in range(B):
        it = iter([x[b] for x in xis])

        class _Rng:          # feeds the external draws to generate_latent_noise
            def __init__(self, it):
                self.it = it

            def randn(self, *shape):
                return next(self.it).astype('float64').reshape(shape)
        m = qg_ref.QGModelRef(nx=N, **params)
        m.sampling_type = sampling
        m.noise_sampler = samplers_ref.make_sampler(sampling, nd)
        m.q_parameterization = gen_ref.ParameterizationRef(ora, rng=_Rng(it))
        m.set_q(q0[b])
        refs.append(m)
    draws = 0
    worst_S = worst_q = 0.0
    for s in range(nsteps):
        # the oracle consumes one external draw per sampler refresh: every step for AR1,
        # every nd-th step for the constant sampler (stochastic_pyqg.py:62-71)
        xi = torch.as_tensor(np.ascontiguousarray(xis[draws].reshape(B, 2, N, N))).cuda()
        if sampling == 'AR1' or s % nd == 0:
            draws += 1
        e.step(1, generator=gen, sampling=sampling, nsteps_decor=nd, z_external=xi)
        for m in refs:
            m._step_forward()
        qh = e.get(L.F_QH).cpu().numpy()
        S = e.get(L.F_S).cpu().numpy()
        for b, m in enumerate(refs):
            sc = np.abs(m.PV_forcing).max(axis=(1, 2), keepdims=True)
            worst_S = max(worst_S, (np.abs(S[b] - m.PV_forcing) / sc).max())
            worst_q = max(worst_q, _rel(qh[b], m.qh))
            # the bound the generator alone is held to (measured here: 1e-6 ... 3e-6 of max|S|, profiles/r04_step_parity.txt)
            assert (np.abs(S[b] - m.PV_forcing) / sc).max() < 2e-5, (s, b)
            # the f32 generator difference enters qh scaled by dt*|S|/|q| ~ 1e-2 per step (measured: 3e-8 ... 1.5e-7)
            assert _rel(qh[b], m.qh) < 5e-7, (s, b)
    print(f'\n{kind} {sampling} {nd} {N} {B}: worst S error {worst_S:.2e} of max|S|, worst qh error {worst_q:.2e}')


@pytest.mark.parametrize('mode', ['f32', 'f16x3'])
def test_on_device_noise_run_is_reproducible_and_member_streams_differ(mode):
    """Bit-for-bit: run-to-run, and a shard of an ensemble against the whole ensemble.  Results are
    independent of the ensemble size as long as both sides run the same generator arithmetic (the
    automatic choice switches kernels with the number of resident members), so it is pinned here."""
    import pyqg_generative_amd._lib as L
    N, B = 64, 4
    q0 = _eddy_like_q(np.random.RandomState(1), 1, N).repeat(B, axis=0)
    gen = _gpu_generator('gan')
    if mode == 'f32':
        gen.set_option('precision', 0)
    else:
        gen.set_option('part_max_tiles', 0)       # no split-K for small member counts (other summation order)
    outs = []
    for trial in range(2):
        e = _engine(N, B, dt=14400.)
        e.set_q(q0)
        e.step(6, generator=gen, sampling='AR1', nsteps_decor=1, seed=42)
        outs.append(e.get(L.F_Q))
    assert torch.equal(outs[0], outs[1])
    q = outs[0].cpu().numpy()
    assert np.isfinite(q).all()
    assert np.abs(q[0] - q[1]).max() > 0          # identical IC, different noise stream per member
    # sharding: members 2..3 of a 4-member run == a 2-member run with member_offset=2
    e = _engine(N, 2, dt=14400.)
    e.set_q(q0[:2])
    e.step(6, generator=gen, sampling='AR1', nsteps_decor=1, seed=42, member_offset=2)
    assert torch.equal(e.get(L.F_Q), outs[0][2:])


def test_fused_step_noise_follows_pinned_philox_stream():
    """The sampler update folded into the generator's input kernel draws the same Philox stream as
    qgx_noise_normal / the oracle: z after steps 0 and 1 of an AR1(nsteps=2) run."""
    import pyqg_generative_amd._lib as L
    N, B, seed, off = 64, 3, 99, 5
    gen = _gpu_generator('vae')
    e = _engine(N, B, dt=14400.)
    e.set_q(_eddy_like_q(np.random.RandomState(0), B, N))
    e.step(1, generator=gen, sampling='AR1', nsteps_decor=2, seed=seed, member_offset=off)
    z0 = e.get(L.F_Z).cpu().numpy().reshape(B, -1)
    e.step(1, generator=gen, sampling='AR1', nsteps_decor=2, seed=seed, member_offset=off)
    z1 = e.get(L.F_Z).cpu().numpy().reshape(B, -1)
    a, b = np.float32(1 - 1 / 2), np.float32((1 / 2 * (2 - 1 / 2)) ** 0.5)
    for m in range(B):
        x0, _ = samplers_ref.philox_normal(seed, off + m, 0, 2 * N * N)
        x1, _ = samplers_ref.philox_normal(seed, off + m, 1, 2 * N * N)
        assert np.abs(z0[m] - x0).max() < 2e-5
        assert np.abs(z1[m] - (a * x0 + b * x1)).max() < 4e-5



@pytest.mark.parametrize('kind,N,B,sampling,nd', [('gan', 64, 4, 'AR1', 1), ('vae', 64, 1, 'constant', 1),
                                                  ('vae', 96, 3, 'AR1', 1), ('gan', 48, 5, 'constant', 3),
                                                  ('gan', 64, 2, 'AR1', 10), ('gan+reg', 64, 3, 'AR1', 1),
                                                  ('vae+reg', 96, 2, 'constant', 2)])
def test_generator_kernels_folded_into_the_step_kernel_change_nothing(kind, N, B, sampling, nd):
    """Layer-split small grids: the generator's output kernel rides in the step kernel's prologue and, for white-in-time
    Philox noise, the next step's input kernel in its epilogue (GenFuse).  Same arithmetic in the same order: the run is
    bit-identical to the one with separate kernels (option genfuse = 0), diagnostics cadence and range words included.
    Likewise the diagnostics increment: one kernel per member, its transforms spread over (member, transform) workgroups, or
    one launch per transform (diag.hip) — all sixteen accumulated diagnostics bit for bit."""
    import pyqg_generative_amd._lib as L
    q0 = _eddy_like_q(np.random.RandomState(7), B, N)
    gen = _gpu_generator(kind)
    res = []
    # default / separate generator kernels / one launch per transform / the increment's transforms as (member, transform)
    # workgroups (three launches) / as ONE workgroup per member with its work fields in registers (k_diag_small_reg, grids up to
    # 64 x 64: a quarter of the bytes) / ... with them in global memory (k_diag_small)
    # ... / the forcing's transform inside the layer's one workgroup instead of on a sibling workgroup (k_step_small PART 3)
    # (siblings = 2: with the cross-XCD publication, the path a pair placed on two XCDs takes)
    for opts in ({}, dict(genfuse=0), dict(siblings=0), dict(siblings=0, genfuse=0), dict(siblings=2), dict(diag_fused=0), dict(diag_wide=1), dict(diag_wide=0), dict(diag_wide=0, diag_reg=0),
                 dict(diag_wide=0, diag_reg=2), dict(diag_wide=0, diag_reg=3)):
        e = _engine(N, B, dt=dt_for(N))
        for opt, val in opts.items():
            e.set_option(opt, val)
        e.set_q(q0)
        e.diag_config(0, 4)
        for chunk in (7, 1, 5):
            e.step(chunk, generator=gen, sampling=sampling, nsteps_decor=nd, seed=11, member_offset=3)
        res.append([e.get(f).clone() for f in (L.F_QH, L.F_S, L.F_Z, L.F_Q, L.F_U, L.F_PH)] +
                   [e.diag(n).clone() for n in _lib_diags()] +
                   [torch.as_tensor(gen.range_read()[1]), torch.as_tensor(e.diag_count)])
        e.close()
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize('kind,N,B,sampling,nd', [('gan', 64, 3, 'AR1', 1), ('vae', 96, 4, 'constant', 2), ('gan+reg', 48, 2, 'AR1', 4)])
def test_step_kernel_as_two_kernels_on_two_streams_changes_nothing(kind, N, B, sampling, nd):
    """Option split_adv: the half of the step kernel that needs nothing of the forcing (inversion, advection products, their
    transform, the tendency without its forcing term) runs as a kernel of its own on a side stream under the generator's
    layers, the other half behind both.  Same arithmetic in the same order: bit-identical state, forcing, noise and
    diagnostics, on whole ensembles and on two half-ensembles (pyqg model.py::_step_forward order of operations)."""
    import pyqg_generative_amd._lib as L
    q0 = _eddy_like_q(np.random.RandomState(9), B, N)
    gen = _gpu_generator(kind)
    res = []
    for opts in ({}, dict(split_adv=1), dict(split_adv=1, genfuse=0), dict(streams=2), dict(split_adv=1, streams=2), dict(siblings=0), dict(siblings=0, streams=2)):
        e = _engine(N, B, dt=dt_for(N))
        for opt, val in opts.items():
            e.set_option(opt, val)
        e.set_q(q0)
        e.diag_config(0, 4)
        for chunk in (7, 1, 9):
            e.step(chunk, generator=gen, sampling=sampling, nsteps_decor=nd, seed=11, member_offset=3)
        res.append([e.get(f).clone() for f in (L.F_QH, L.F_S, L.F_Z, L.F_Q, L.F_U, L.F_PH)] + [e.diag(n).clone() for n in _lib_diags()])
        e.close()
    for ref, other in ((0, 1), (0, 2), (3, 4), (0, 5), (3, 6)):      # (the halves against the halves: they may take other generator kernels than the whole)
        for a, b in zip(res[ref], res[other]):
            assert torch.equal(a, b)


def test_full_size_step_members_are_independent():
    """BASELINE's single-GPU shard (128 members, 64 x 64, GAN): copies of four members spread over the
    ensemble, fed the same external noise, stay bit-identical through parameterized steps, and the
    first copies match the CPU oracle (size-independent property at the full benchmark size)."""
    import pyqg_generative_amd._lib as L
    N, B, nsteps = 64, 128, 3
    rs = np.random.RandomState(11)
    q4 = _eddy_like_q(rs, 4, N)
    xi4 = rs.randn(nsteps, 4, 2, N, N).astype('float32')
    gen = _gpu_generator('gan')
    e = _engine(N, B, dt=14400.)
    e.set_q(np.tile(q4, (B // 4, 1, 1, 1)))
    for s in range(nsteps):
        z = torch.as_tensor(np.tile(xi4[s], (B // 4, 1, 1, 1))).cuda()
        e.step(1, generator=gen, sampling='constant', nsteps_decor=1, z_external=z)
    qh = e.get(L.F_QH)
    for r in range(4):
        assert torch.equal(qh[r::4], qh[r:r + 1].expand(B // 4, -1, -1, -1)), r
    ora = load_generator('gan')
    for b in range(2):
        it = iter(xi4[:, b])
        class _Rng:
            def randn(self, *shape):
                return next(it).astype('float64').reshape(shape)
        m = qg_ref.QGModelRef(nx=N, dt=14400.)
        m.sampling_type = 'constant'
        m.noise_sampler = samplers_ref.make_sampler('constant', 1)
        m.q_parameterization = gen_ref.ParameterizationRef(ora, rng=_Rng())
        m.set_q(q4[b])
        for s in range(nsteps):
            m._step_forward()
        assert _rel(qh[b].cpu().numpy(), m.qh) < 5e-6


@pytest.mark.parametrize('case', range(14))
def test_randomised_configurations_match_oracle(case):
    """seeded sweep over grid size x member count (odd counts, counts around the kernel-selection thresholds) x
    generator kind x sampler: two online steps against the oracle, member 0, a middle member and the last member"""
    import pyqg_generative_amd._lib as L
    rs = np.random.RandomState(1000 + case)
    N = [32, 48, 64, 96, 128, 64, 96, 48, 64, 32, 64, 96, 48, 128][case]
    B = [1, 3, 5, 7, 2, 9, 13, 17, 33, 40, 6, 57, 11, 3][case]
    kind = ['gan', 'vae', 'gz'][case % 3]
    sampling, nd = [('AR1', 1), ('constant', 2), ('AR1', 4), ('constant', 1)][case % 4]
    params = JET if case % 5 == 0 else dict(dt=dt_for(N))
    nsteps = 2
    q0 = _eddy_like_q(rs, B, N)
    gen = _gpu_generator(kind)
    ora = load_generator(kind)
    e = _engine(N, B, **params)
    e.set_q(q0)
    shape = (B, 2, N, N) if kind == 'gz' else (B, 1, 2, N, N)
    xis = [rs.randn(*shape) if kind == 'gz' else rs.randn(*shape).astype('float32') for _ in range(nsteps)]
    members = sorted({0, B // 2, B - 1})
    refs = {}
    for b in members:
        it = iter([x[b] for x in xis])

        class _Rng:
            def __init__(self, it):
                self.it = it

            def randn(self, *shp):
                return next(self.it).astype('float64').reshape(shp)
        m = qg_ref.QGModelRef(nx=N, **params)
        m.sampling_type = sampling
        m.noise_sampler = samplers_ref.make_sampler(sampling, nd)
        m.q_parameterization = gen_ref.ParameterizationRef(ora, rng=_Rng(it))
        m.set_q(q0[b])
        refs[b] = m
    draws = 0
    for s in range(nsteps):
        xi = torch.as_tensor(np.ascontiguousarray(xis[draws].reshape(B, 2, N, N))).cuda()
        if sampling == 'AR1' or s % nd == 0:
            draws += 1
        e.step(1, generator=gen, sampling=sampling, nsteps_decor=nd, z_external=xi)
        for m in refs.values():
            m._step_forward()
    qh = e.get(L.F_QH).cpu().numpy()
    S = e.get(L.F_S).cpu().numpy()
    for b, m in refs.items():
        sc = np.abs(m.PV_forcing).max(axis=(1, 2), keepdims=True)
        assert (np.abs(S[b] - m.PV_forcing) / sc).max() < 5e-5, (N, B, kind, b)
        assert _rel(qh[b], m.qh) < 2e-6, (N, B, kind, b)
    assert gen.range_ok() is None


def _lib_diags():
    from pyqg_generative_amd._lib import DIAGS
    return DIAGS


def dt_for(N):
    return 14400. if N <= 64 else 7200.


@pytest.mark.parametrize('N', [64, 256])
def test_fields_after_unrefreshed_steps_belong_to_the_current_state(N):
    """steps with refresh_diag=False store no ph, u, v (every step of a 256 x 256 run kernel): reading u, v, ph or p
    afterwards inverts the CURRENT state first instead of handing out fields of an older one"""
    import pyqg_generative_amd._lib as L
    B = 2
    q0 = _eddy_like_q(np.random.RandomState(21), B, N)
    e = _engine(N, B, dt=3600.)
    e.set_q(q0)
    e.step(2)                                  # refreshed: fields of the inversion of step 2
    e.step(7, refresh_diag=False)
    u, p = e.get(L.F_U).cpu().numpy(), e.get(L.F_P).cpu().numpy()
    for b in range(B):
        m = qg_ref.QGModelRef(nx=N, dt=3600.)
        m.set_q(q0[b])
        for _ in range(9):
            m._step_forward()
        m._invert()                            # of the state after step 9
        assert _rel(u[b], m.u) < 1e-10
        assert _rel(p[b], m.ifft(m.ph)) < 1e-10
