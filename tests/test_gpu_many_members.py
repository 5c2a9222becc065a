"""GPU parity of the kernels an ensemble of MORE than 128 members per GPU takes on the small grids.

`spectral_small.hip` advances a member with two workgroups (one per layer, `k_step_small<NN, true>`, generator output /
input kernels folded in) while 2 B <= 256, and with ONE workgroup per member beyond that (`k_step_small<NN, false>`:
six packed transforms, separate generator output / input kernels; 1024 threads up to 256 members, 512 threads beyond
at N <= 64).  BASELINE configs[2] (1024 members) on 1, 2 or 4 GPUs and every strong-scaling run of it live on the second
form, as do `k_q_to_qh_small`, `k_qh_to_q_small`, `k_invert_small` at 512 threads and `k_diag_small` at B > 128.
Every test compares with the CPU oracle (member 0, a middle member, the last member) on identical inputs; tolerances as
in tests/test_gpu_parity.py: float64 spectral core 1e-12 x steps of the field maximum, float32-class generator 5e-5 of
the forcing maximum and 2e-6 on qh after a parameterized step.
"""
import os
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from conftest import load_generator, GOLDEN
from oracle import qg_ref, gen_ref, samplers_ref

F64_TOL = 1e-12
JET = dict(dt=7200., rek=7e-8, delta=0.1, beta=1e-11)      # tools/parameters.py:26-27,37


def _engine(N, B, **kw):
    import pyqg_generative_amd as qa
    return qa.EnsembleEngine(nx=N, n_members=B, **kw)


def _gpu_generator(kind):
    import pyqg_generative_amd as qa
    from pyqg_generative_amd import weights
    nets, xs, ys = weights.load_npz(os.path.join(GOLDEN, f'weights_{kind}.npz'), kind)
    return qa.Generator(kind, nets, xs, ys)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _eddy_like_q(rs, B, N):
    m = qg_ref.QGModelRef(nx=N)
    q = rs.randn(B, 2, N, N) * np.array([8e-6, 1e-6])[None, :, None, None]
    qh = np.fft.rfftn(q, axes=(-2, -1)) * (m.wv < 2. / 3. * m.kk[-1])
    return np.fft.irfftn(qh, axes=(-2, -1)) * 3.0


def _members(B):
    return sorted({0, B // 2, B - 1})


@pytest.mark.parametrize('N,B', [(64, 129), (64, 257), (48, 300), (96, 130), (32, 513)])
def test_unparameterized_steps_one_workgroup_per_member(N, B):
    """Euler -> AB2 -> AB3 start-up, filter, friction through the un-split step kernel (1024 threads: B = 129, 130;
    512 threads: B = 257, 300, 513), and the q / qh setters, the inversion and the status reduction at those shapes"""
    import pyqg_generative_amd._lib as L
    nsteps = 12
    params = JET if N == 96 else dict(dt=14400.)
    rs = np.random.RandomState(300 + N + B)
    q0 = _eddy_like_q(rs, B, N)
    e = _engine(N, B, **params)
    e.set_q(q0)
    assert _rel(e.get(L.F_QH).cpu().numpy(), np.fft.rfftn(q0, axes=(-2, -1))) < F64_TOL        # k_q_to_qh_small
    refs = {}
    for b in _members(B):
        m = qg_ref.QGModelRef(nx=N, **params)
        m.set_q(q0[b])
        refs[b] = m
    e.invert()                                                                                     # k_invert_small
    ph, u, v = (e.get(f).cpu().numpy() for f in (L.F_PH, L.F_U, L.F_V))
    for b, m in refs.items():
        m._invert()
        assert _rel(ph[b], m.ph) < F64_TOL and _rel(u[b], m.u) < F64_TOL and _rel(v[b], m.v) < F64_TOL
    for s in range(nsteps):
        e.step(1)
        for m in refs.values():
            m._step_forward()
        if s in (0, 1, 2, nsteps - 1):
            qh = e.get(L.F_QH).cpu().numpy()
            q = e.get(L.F_Q).cpu().numpy()
            for b, m in refs.items():
                assert _rel(qh[b], m.qh) < F64_TOL * (s + 1), (s, b)
                assert _rel(q[b], m.q) < F64_TOL * (s + 1), (s, b)
    ph, u, v = (e.get(f).cpu().numpy() for f in (L.F_PH, L.F_U, L.F_V))
    dq, dqpp = e.get(L.F_DQHDT).cpu().numpy(), e.get(L.F_DQHDT_PP).cpu().numpy()
    ke, cfl = e.status()
    for b, m in refs.items():
        assert _rel(ph[b], m.ph) < 1e-11 and _rel(u[b], m.u) < 1e-11 and _rel(v[b], m.v) < 1e-11
        assert _rel(dq[b], m.dqhdt_p) < 1e-10 and _rel(dqpp[b], m.dqhdt_pp) < 1e-10
        assert abs(ke[b] - m._calc_ke()) < 1e-11 * m._calc_ke() and abs(cfl[b] - m._calc_cfl()) < 1e-11
    # many steps in one call == single steps, bit for bit (history rotation by pointer)
    e2 = _engine(N, B, **params)
    e2.set_q(q0)
    e2.step(nsteps)
    assert torch.equal(e2.get(L.F_QH), e.get(L.F_QH))
    # qh setter -> q (k_qh_to_q_small)
    e.set_qh(np.fft.rfftn(q0, axes=(-2, -1)))
    assert _rel(e.get(L.F_Q).cpu().numpy(), q0) < F64_TOL


@pytest.mark.parametrize('kind,N,B,sampling,nd,nsteps,params', [
    ('gan', 64, 129, 'constant', 1, 3, dict(dt=14400.)),     # first member count of the un-split kernel
    ('gan', 64, 256, 'constant', 1, 2, dict(dt=14400.)),     # BASELINE configs[2] on 4 GPUs; last count at 1024 threads
    ('gan', 64, 257, 'AR1', 1, 2, dict(dt=14400.)),          # first count at 512 threads
    ('vae', 96, 129, 'constant', 1, 2, JET),                 # BASELINE configs[3] whole on 2 GPUs would be 128; one more
    ('gz', 48, 300, 'constant', 2, 3, dict(dt=14400.)),      # float64 noise, two nets, 512 threads
    ('vae', 64, 160, 'AR1', 3, 3, dict(dt=14400.)),          # coloured noise: the separate sampler-update kernel
], ids=lambda v: str(v) if not isinstance(v, dict) else ('jet' if 'rek' in v else 'eddy'))
def test_parameterized_steps_one_workgroup_per_member(kind, N, B, sampling, nd, nsteps, params):
    """the full online step (sampler + generator + de-mean + spectral step) with more than 128 resident members, white
    noise supplied externally so that both sides see identical draws; S and qh after every step"""
    import pyqg_generative_amd._lib as L
    rs = np.random.RandomState(4000 + N + B)
    q0 = _eddy_like_q(rs, B, N)
    gen = _gpu_generator(kind)
    ora = load_generator(kind)
    e = _engine(N, B, **params)
    e.set_q(q0)
    shape = (B, 2, N, N) if kind == 'gz' else (B, 1, 2, N, N)
    xis = [rs.randn(*shape) if kind == 'gz' else rs.randn(*shape).astype('float32') for _ in range(nsteps)]
    refs = {}
    for b in _members(B):
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
            assert (np.abs(S[b] - m.PV_forcing) / sc).max() < 5e-5, (s, b)
            assert _rel(qh[b], m.qh) < 2e-6, (s, b)
    assert gen.range_ok() is None


def test_strong_scaling_shard_sizes_reproduce_each_other():
    """BASELINE configs[2] is ONE 1024-member ensemble: a GPU holds 1024 / 512 / 256 / 128 of its members on 1 / 2 / 4 / 8
    GPUs, i.e. three different step kernels (512 threads, 1024 threads, layer split).  Members 250..259 of a 300-member
    shard, of a 150-member shard starting at member 150 and of a 10-member shard starting at 250 (on-device Philox noise
    keyed by the global member id) agree to float64 rounding after parameterized steps — the generator arithmetic is pinned
    (no split-K for the small shard), so the only difference is the butterfly order of the spectral kernels."""
    import pyqg_generative_amd._lib as L
    N, nsteps = 64, 4
    q0 = _eddy_like_q(np.random.RandomState(5), 300, N)
    gen = _gpu_generator('gan')
    gen.set_option('part_max_tiles', 0)
    gen.set_option('wino_min_tiles', 1)          # the same 5x5 kernel at every shard size
    outs = []
    for first, count in ((0, 300), (150, 150), (250, 10)):
        e = _engine(N, count, dt=14400.)
        e.set_q(q0[first:first + count])
        e.step(nsteps, generator=gen, sampling='constant', nsteps_decor=1, seed=42, member_offset=first)
        lo = 250 - first
        outs.append((e.get(L.F_QH)[lo:lo + 10].cpu().numpy(), e.get(L.F_S)[lo:lo + 10].cpu().numpy(),
                     e.get(L.F_Z)[lo:lo + 10].cpu().numpy()))
        e.close()
    for other in outs[1:]:
        np.testing.assert_array_equal(other[2], outs[0][2])          # the same Philox draws, bit for bit
        assert _rel(other[0], outs[0][0]) < 1e-8                     # float32 generator fed with q differing by 1e-16
        assert _rel(other[1], outs[0][1]) < 1e-5


@pytest.mark.parametrize('lsplit', [0, 1])
def test_layer_split_and_whole_member_kernels_agree(lsplit):
    """the same four members through `k_step_small<64, true>` (two workgroups per member) and `<64, false>` (one): single
    real fields packed as (x + 0 i) against pairs packed as (x + i y) — the same transforms, different rounding:
    1e-12 x steps; parameterized steps with the same external forcing"""
    import pyqg_generative_amd._lib as L
    N, B, nsteps = 64, 4, 10
    rs = np.random.RandomState(17)
    q0 = _eddy_like_q(rs, B, N)
    S = [torch.as_tensor(rs.randn(B, 2, N, N) * np.array([7e-12, 2e-13])[None, :, None, None]).cuda() for _ in range(nsteps)]
    res = []
    for forced in (lsplit, -1):
        e = _engine(N, B, dt=14400.)
        e.set_option('lsplit', forced)
        e.set_q(q0)
        for s in range(nsteps):
            e.step(1, forcing=S[s], weight=0.7, demean=True)
        res.append([e.get(f).cpu().numpy() for f in (L.F_QH, L.F_Q, L.F_DQHDT, L.F_U, L.F_PH)])
        e.close()
    for a, b in zip(*res):
        assert _rel(a, b) < F64_TOL * nsteps
    if lsplit == 1:          # the automatic choice at B = 4 IS the layer split: bit-identical
        for a, b in zip(*res):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('N,B,threads', [(64, 130, 0), (48, 260, 0), (64, 6, 512)])
def test_time_averaged_diagnostics_with_many_members(N, B, threads):
    """`k_diag_small` (always 1024 threads) between steps of the un-split kernel: pyqg's cadence (tavestart, taveint)
    through the QGModel facade, every exported diagnostic against the oracle for three members"""
    from pyqg_generative_amd.qgmodel import QGModel
    from pyqg_generative_amd._lib import DIAGS
    dt, nsteps = 14400., 20
    kw = dict(nx=N, dt=dt, tmax=dt * nsteps, tavestart=dt * 4, taveint=dt * 3, twrite=10000)
    m = QGModel(log_level=0, n_members=B, **kw)
    if threads:
        m._eng.set_option('lsplit', 0)
        m._eng.set_option('spec_threads', threads)
    q0 = _eddy_like_q(np.random.RandomState(N + B), B, N)
    m.q = q0
    m.run()
    refs = {}
    for b in _members(B):
        r = qg_ref.QGModelRef(**kw)
        r.set_q(q0[b])
        r.run()
        refs[b] = r
    assert m.diagnostics_count == refs[0].diag_count > 0
    for name in DIAGS:
        if name.startswith('paramspec') or name == 'ENSparamspec':
            continue                      # no parameterization in this run
        got = m.get_diagnostic(name)
        for b, r in refs.items():
            ref = r.get_diagnostic(name)
            assert got[b].shape == ref.shape, name
            assert np.abs(got[b] - ref).max() <= 1e-9 * np.abs(ref).max(), (name, b)
    m.close()


def test_parameterized_diagnostics_with_many_members():
    """paramspec and its split with a CGAN forcing at B = 140 (separate generator kernels, diagnostics increment with S
    before the step), external noise: against the oracle's time averages"""
    import pyqg_generative_amd._lib as L
    N, B, nsteps = 64, 140, 6
    rs = np.random.RandomState(99)
    q0 = _eddy_like_q(rs, B, N)
    gen = _gpu_generator('gan')
    ora = load_generator('gan')
    e = _engine(N, B, dt=14400.)
    e.set_q(q0)
    e.diag_config(2, 2)
    xis = rs.randn(nsteps, B, 1, 2, N, N).astype('float32')
    refs = {}
    for b in _members(B):
        it = iter(xis[:, b])

        class _Rng:
            def __init__(self, it):
                self.it = it

            def randn(self, *shp):
                return next(self.it).astype('float64').reshape(shp)
        m = qg_ref.QGModelRef(nx=N, dt=14400., tavestart=2 * 14400., taveint=2 * 14400.)
        m.sampling_type = 'constant'
        m.noise_sampler = samplers_ref.make_sampler('constant', 1)
        m.q_parameterization = gen_ref.ParameterizationRef(ora, rng=_Rng(it))
        m.set_q(q0[b])
        refs[b] = m
    for s in range(nsteps):
        e.step(1, generator=gen, sampling='constant', nsteps_decor=1,
               z_external=torch.as_tensor(np.ascontiguousarray(xis[s].reshape(B, 2, N, N))).cuda())
        for m in refs.values():
            m._step_forward()
    assert e.diag_count == refs[0].diag_count == 2
    for name in L.DIAGS:
        got = e.diag(name).cpu().numpy()
        for b, m in refs.items():
            ref = m.get_diagnostic(name)
            # the forcing enters in float32 arithmetic on both sides (different summation order): 5e-5 of its maximum
            tol = 2e-4 if ('param' in name or 'Diss' in name) else 1e-6      # Dissspec: the forcing is part of the tendency
            assert np.abs(got[b] - ref).max() <= tol * np.abs(ref).max(), (name, b)


@pytest.mark.parametrize('kind,N,B', [('gan', 64, 100), ('vae', 96, 40), ('gz', 32, 150)])
def test_four_workgroups_per_member_beyond_residency_are_the_same_step(kind, N, B):
    """Option `siblings` (k_step_small PART 3): the forcing's transform on a sibling workgroup, joined with its layer's
    inversion / advection chain by a flag in memory.  The automatic choice keeps all 4 B workgroups resident (4 B <= 256);
    forced beyond that — 400, 160 and 600 workgroups here, some of them waiting for a CU — the forcing blocks still come first
    in dispatch order and wait for nothing, so the step completes, bit-identical to the two-workgroup form, through the
    same-XCD publication (1) and through the cross-XCD one (2); time-averaged diagnostics included."""
    import pyqg_generative_amd._lib as L
    gen = _gpu_generator(kind)
    q0 = _eddy_like_q(np.random.RandomState(B), B, N)
    res = []
    for sib in (0, 1, 2):
        e = _engine(N, B, **(JET if N == 96 else dict(dt=14400. * 64 / N)))
        e.set_option('siblings', sib)
        e.set_q(q0)
        e.diag_config(0, 3)
        for chunk in (5, 1, 4):
            e.step(chunk, generator=gen, sampling='AR1', nsteps_decor=2, seed=5, member_offset=1)
        res.append([e.get(f).clone() for f in (L.F_QH, L.F_S, L.F_Q, L.F_DQHDT, L.F_Z)] + [e.diag(n).clone() for n in ('KEspec', 'paramspec', 'ENSflux')])
        # the latent noise comes back in the element type of the generator (GZ: float64), a contradicting request is refused
        assert res[-1][4].dtype == gen.noise_dtype
        with pytest.raises(ValueError):
            e.get(L.F_Z, noise_dtype=torch.float32 if kind == 'gz' else torch.float64)
        e.close()
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)
    assert bool(torch.isfinite(res[0][2]).all())


def test_an_ensemble_stepped_as_two_halves_on_two_streams_is_the_same_ensemble():
    """qgx_step advances a 96 x 96 ensemble of 16 ... 64 members as two halves on two internal streams (option `streams`:
    0 automatic, 1 never, 2 whenever even; qgx_step_streams): a half is the same model over a slice of the member-major arrays,
    the noise streams are keyed by the global member id, the diagnostics accumulators are per member — state, forcing and
    time-averaged diagnostics are BIT-identical to the one-stream step where the halves run the kernels the whole runs
    (96 x 96 / 32 members, 64 x 64 / 128), and equal to the float32 generator's rounding where an ensemble-size threshold separates them (48 x 48 / 20: split-K kernels
    for the halves — which also checks that the two halves' split-K buffers are separate)."""
    import pyqg_generative_amd as qa
    from pyqg_generative_amd import _lib as L, weights
    for N, B, kind, exact in ((96, 32, 'vae', True), (64, 128, 'gan', True), (48, 20, 'gan', False), (64, 14, 'gz', False), (32, 6, 'vae', False)):
        nets, xs, ys = weights.load_npz(os.path.join(GOLDEN, f'weights_{kind}.npz'), kind)
        gen = qa.Generator(kind, nets, xs, ys)
        rs = np.random.RandomState(N + B)
        q0 = rs.randn(B, 2, N, N) * 1e-6
        res = {}
        for streams in (1, 2):
            e = qa.EnsembleEngine(nx=N, n_members=B, dt=3600.)
            e.set_option('streams', streams)
            assert e.step_streams(gen) == streams
            e.set_q(q0)
            e.diag_config(2, 5)
            kw = dict(generator=gen, sampling='AR1', nsteps_decor=3, seed=11, member_offset=7)
            e.step(9, **kw)
            e.step(14, refresh_diag=False, **kw)            # a second call continues both halves' samplers and AB3 histories
            import torch
            zt = torch.float64 if kind == 'gz' else torch.float32       # the mean / variance model draws float64 noise
            res[streams] = [e.get(f, noise_dtype=zt).cpu().numpy() for f in (L.F_QH, L.F_S, L.F_Z, L.F_U, L.F_DQHDT_PP)] + \
                           [e.diag(n).cpu().numpy() for n in ('KEspec', 'paramspec', 'ENSparamspec', 'Dissspec')] + [e.tc, e.diag_count]
            e.close()
        assert res[1][-2:] == res[2][-2:] and res[1][-2] == 23 and res[1][-1] == 4
        for a, b in zip(res[1][:-2], res[2][:-2]):
            scale = np.abs(a).max()
            if exact:
                assert np.array_equal(a, b)
            else:
                assert np.abs(a - b).max() <= 2e-5 * scale      # float32 generator, other kernels at half the ensemble size
        # the automatic choice: the 96 x 96 grid with 16 ... 64 members
        e = qa.EnsembleEngine(nx=N, n_members=B, dt=3600.)
        assert e.step_streams(gen) == (2 if N == 96 else 1) and e.step_streams(None) == 1
        e.close()
        gen.close()


def test_a_failing_half_of_a_two_stream_step_joins_the_streams_and_marks_the_model_invalid():
    """qgx_step's error paths (model.hip): whatever fails after the fork, the caller's stream is joined with both internal
    streams, and a step that failed after one half advanced leaves the handle INVALID — later calls fail loudly instead of
    stepping from halves that stand at different steps.  The failure is injected by a hook of the A/B library (option
    'step_fault': that half refuses its second chunk of steps), hence the child process on libqgx_ab.so."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ab = os.path.join(root, 'pyqg_generative_amd', 'libqgx_ab.so')
    if not os.path.exists(ab):
        pytest.skip('libqgx_ab.so is not built (make -C pyqg_generative_amd/csrc ab)')
    r = subprocess.run([sys.executable, os.path.abspath(__file__), 'stepfault-child'], env=dict(os.environ, QGX_LIB=ab),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    print(out)
    for half in ('1', '2'):
        o = out[half]
        assert o['step_raised'] and 'injected failure' in o['step_error']
        assert o['next_step_raised'] and 'model state is invalid' in o['next_error'] and 'halves are at steps' in o['next_error']
        assert o['get_raised'] and o['stream_usable'] and o['fresh_model_ok']
    # the product library refuses the hook
    import pyqg_generative_amd as qa
    import pyqg_generative_amd._lib as L
    e = qa.EnsembleEngine(nx=96, n_members=16, dt=3600.)
    with pytest.raises(L.QgxError, match='A/B library only'):
        e.set_option('step_fault', 1)
    e.close()


def _stepfault_child():
    import json
    import torch
    import pyqg_generative_amd as qa
    from pyqg_generative_amd import _lib as L, weights
    nets, xs, ys = weights.load_npz(os.path.join(GOLDEN, 'weights_vae.npz'), 'vae')
    gen = qa.Generator('vae', nets, xs, ys)
    N, B = 96, 32
    q0 = np.random.RandomState(5).randn(B, 2, N, N) * 1e-6
    kw = dict(generator=gen, sampling='AR1', nsteps_decor=1, seed=3)
    out = {}
    for half in (1, 2):
        e = qa.EnsembleEngine(nx=N, n_members=B, dt=3600.)
        assert e.step_streams(gen) == 2
        e.set_q(q0)
        e.step(3, **kw)
        e.set_option('step_fault', half)
        o = dict(step_raised=False, next_step_raised=False, get_raised=False)
        try:
            e.step(20, **kw)                        # three chunks of 8: the second chunk of that half is refused
        except L.QgxError as ex:
            o['step_raised'], o['step_error'] = True, str(ex)
        e.set_option('step_fault', 0)
        try:
            e.step(1, **kw)
        except L.QgxError as ex:
            o['next_step_raised'], o['next_error'] = True, str(ex)
        try:
            e.get(L.F_QH)
        except L.QgxError:
            o['get_raised'] = True
        # the caller's stream was joined: it is usable and ordered
        t = torch.ones(1024, device='cuda')
        torch.cuda.synchronize()
        o['stream_usable'] = bool((t + 1).sum().item() == 2048)
        e.close()
        # a fresh handle is not affected
        f = qa.EnsembleEngine(nx=N, n_members=B, dt=3600.)
        f.set_q(q0)
        f.step(9, **kw)
        o['fresh_model_ok'] = bool(f.tc == 9 and np.isfinite(f.get(L.F_QH).cpu().numpy()).all())
        f.close()
        out[str(half)] = o
    print(json.dumps(out))


if __name__ == '__main__' and len(__import__('sys').argv) > 1 and __import__('sys').argv[1] == 'stepfault-child':
    _stepfault_child()
