"""CPU: invariants that guard the (unpinned) restatement of the pyqg 0.7.2 core."""
import numpy as np
from oracle import qg_ref


def test_grid_and_constants():
    m = qg_ref.QGModelRef(nx=64)
    assert m.qh.shape == (2, 64, 33)
    # wavenumber ordering: l = dl*[0..N/2-1, -N/2..-1], k = dk*[0..N/2]  
    assert m.ll[0] == 0 and m.ll[31] == 31 * m.dl and m.ll[32] == -32 * m.dl and m.ll[-1] == -m.dl
    assert m.kk[-1] == 32 * m.dk
    assert abs(m.x[0, 0] - 0.5 * m.dx) < 1e-9 and m.dx == 1e6 / 64
    assert np.isclose(m.F1, 15000. ** -2 / 1.25) and np.isclose(m.F2, 0.25 * m.F1)
    assert np.isclose(m.Qy1, 1.5e-11 + m.F1 * 0.025) and np.isclose(m.Qy2, 1.5e-11 - m.F2 * 0.025)
    assert m.filtr[0, 0] == 1.0 and m.filtr.min() < 1e-3 and (m.filtr <= 1).all()
    assert m.a[0, 0, 0, 0] == 0 and np.all(np.isfinite(m.a))


def test_q_setter_refreshes_qh_and_inversion_roundtrip():
    rs = np.random.RandomState(0)
    m = qg_ref.QGModelRef(nx=48)
    q = rs.randn(2, 48, 48) * 1e-6
    q -= q.mean(axis=(1, 2), keepdims=True)
    m.set_q(q)
    np.testing.assert_allclose(m.ifft(m.qh), q, atol=1e-20)
    m._invert()
    # q1 = lap psi1 + F1 (psi2 - psi1), q2 = lap psi2 + F2 (psi1 - psi2)
    q1h = -m.wv2 * m.ph[0] + m.F1 * (m.ph[1] - m.ph[0])
    q2h = -m.wv2 * m.ph[1] + m.F2 * (m.ph[0] - m.ph[1])
    np.testing.assert_allclose(q1h, m.qh[0], atol=1e-12 * np.abs(m.qh).max())
    np.testing.assert_allclose(q2h, m.qh[1], atol=1e-12 * np.abs(m.qh).max())
    # u = -psi_y, v = psi_x
    np.testing.assert_allclose(m.u, m.ifft(-m.il * m.ph), atol=1e-18)


def test_linear_baroclinic_growth_rate():
    """A single small-amplitude zonal wave must grow at the analytic Phillips-model rate
    of the linearised two-layer equations (independent of pyqg)."""
    N = 64
    m = qg_ref.QGModelRef(nx=N, dt=3600., filterfac=0., rek=5.787e-7)
    kx = 4                                        # k = 4 dk, l = 0
    k = m.kk[kx]
    k2 = k * k
    A = np.array([[-(k2 + m.F2), -m.F1], [-m.F2, -(k2 + m.F1)]]) / (k2 * (k2 + m.F1 + m.F2))
    Lmat = -1j * k * (np.diag(m.Ubg) + np.diag(m.Qy) @ A)
    Lmat[1, :] += m.rek * k2 * A[1, :]
    w, vec = np.linalg.eig(Lmat)
    i = np.argmax(w.real)
    sigma = w[i].real
    assert sigma > 0
    # initialise on the unstable eigenvector
    amp = 1e-12 * N * N
    qh = np.zeros((2, N, N // 2 + 1), complex)
    qh[:, 0, kx] = amp * vec[:, i]
    m.set_qh(qh)
    nsteps = 400
    e0 = np.abs(m.qh[:, 0, kx]).copy()
    for _ in range(nsteps):
        m._step_forward()
    growth = np.log(np.abs(m.qh[:, 0, kx]) / e0) / (nsteps * m.dt)
    np.testing.assert_allclose(growth, sigma, rtol=2e-4)


def test_eddy_run_reproduces_published_growth_and_saturation():
    """Soft pin from the reference's own logs (notebooks/3-2-dealiasing.ipynb:1412-1455, unseeded
    64x64 eddy run): KE grows x5.7-7.1 per 1000 steps during the linear stage, CFL=0.023 at rest,
    equilibrium KE 4.6e-4..5.4e-4."""
    m = qg_ref.QGModelRef(nx=64, dt=14400., tmax=14400. * 6000, twrite=1000)
    qg_ref.set_initial_condition(m, np.random.RandomState(0))
    ke = {}
    for t in m.run_with_snapshots(tsnapint=14400. * 1000):
        ke[m.tc] = m._calc_ke()
        if m.tc == 1000:
            assert abs(m.cfl - 0.023) < 5e-4
    assert np.isfinite(m.q).all() and m.cfl < 1
    assert 4.0 < ke[2000] / ke[1000] < 8.0 and 5.0 < ke[3000] / ke[2000] < 8.0
    assert 3.5e-4 < ke[6000] < 6.5e-4
    assert np.abs(m.q.mean(axis=(1, 2))).max() < 1e-19


def test_ab3_startup_and_history_rotation():
    m = qg_ref.QGModelRef(nx=32, dt=14400.)
    qg_ref.set_initial_condition(m, np.random.RandomState(1))
    levels = []
    for _ in range(4):
        levels.append(m.ablevel)
        prev = m.dqhdt_p
        m._step_forward()
        assert m.dqhdt_pp is prev
    assert levels == [0, 1, 2, 2]


def _unstable_mode(m, kx, ly):
    """most unstable eigenpair of the linearised two-layer operator at wavenumber (k, l) for the model's
    parameters: dq/dt = L q with L = -ik (U + Qy A(kappa^2)) + bottom drag on layer 2"""
    k, l = m.kk[kx], m.ll[ly]
    K2 = k * k + l * l
    A = np.array([[-(K2 + m.F2), -m.F1], [-m.F2, -(K2 + m.F1)]]) / (K2 * (K2 + m.F1 + m.F2))
    Lmat = -1j * k * (np.diag(m.Ubg) + np.diag(m.Qy) @ A)
    Lmat[1, :] += m.rek * K2 * A[1, :]
    w, vec = np.linalg.eig(Lmat)
    i = np.argmax(w.real)
    return w[i], vec[:, i]


def test_linear_growth_rates_oblique_modes_and_jet_parameters():
    """analytic Phillips growth (or decay) rate AND phase speed at several (k, l), l != 0 included, for the eddy and the
    jet parameter sets (tools/parameters.py:26-27,37) at the grid sizes of BASELINE's configs"""
    jet = dict(rek=7e-8, delta=0.1, beta=1e-11)
    for N, params, modes in ((64, {}, [(4, 0), (3, 2), (5, -3)]), (96, jet, [(4, 0), (6, 3), (3, -2)]),
                             (48, {}, [(3, 1)])):
        for kx, ly in modes:
            m = qg_ref.QGModelRef(nx=N, dt=1800., filterfac=0., **params)
            w, vec = _unstable_mode(m, kx, ly)
            qh = np.zeros((2, N, N // 2 + 1), complex)
            qh[:, ly, kx] = 1e-12 * N * N * vec
            m.set_qh(qh)
            nsteps = 300
            a0 = m.qh[:, ly, kx].copy()
            for _ in range(nsteps):
                m._step_forward()
            ratio = m.qh[:, ly, kx] / a0
            T = nsteps * m.dt
            np.testing.assert_allclose(np.log(np.abs(ratio)) / T, w.real, rtol=3e-4, err_msg=str((N, kx, ly)))
            # phase: exp(i Im(w) T), compared modulo 2 pi through the complex ratio
            np.testing.assert_allclose(ratio / np.abs(ratio), np.exp(1j * w.imag * T) * np.ones(2), atol=2e-3)
            # no other mode was excited (the step is linear in a single small wave)
            rest = np.abs(m.qh).copy()
            rest[:, ly, kx] = 0
            assert rest.max() < 1e-9 * np.abs(m.qh[:, ly, kx]).max()


def _hermitian_sum(x):
    """sum over the full wavenumber plane of a half-plane (l, k >= 0) real density"""
    w = np.full(x.shape[-1], 2.0)
    w[0] = w[-1] = 1.0
    return float((x * w).sum())


def test_energy_budget_of_the_diagnostics_closes():
    """The ten spectral diagnostics are a decomposition of the energy tendency: with
    E = -1/2 sum_k (H_k/H) <psi_k q_k>, dE/dt = -sum_k (H_k/H) Re(conj(psi_k) dq_k/dt) / M^2 summed over the
    plane must equal sum(KEflux + APEflux + APEgenspec + KEfrictionspec + paramspec) — to round-off, for the
    tendency the model actually steps with (before the filter).  The nonlinear fluxes only redistribute:
    sum(KEflux) = 0 on any field; sum(APEflux) = 0 once cubic products are alias-free (kappa < N/4)."""
    N = 64
    rs = np.random.RandomState(5)
    S = rs.randn(2, N, N) * np.array([7e-12, 2e-13])[:, None, None]
    for band, params in ((1. / 4., {}), (2. / 3., dict(rek=7e-8, delta=0.1, beta=1e-11))):
        m = qg_ref.QGModelRef(nx=N, dt=14400., parameterization=lambda mm: S, **params)
        q = rs.randn(2, N, N) * np.array([8e-6, 1e-6])[:, None, None]
        qh = np.fft.rfftn(q, axes=(-2, -1)) * (m.wv < band * m.kk[-1])
        m.set_qh(qh)
        m._invert()
        m._do_advection()
        m._do_friction()
        m._do_q_subgrid_parameterization()
        d = m._diag_functions()
        dEdt = -_hermitian_sum(((m.Hi / m.H)[:, None, None] * np.real(np.conj(m.ph) * m.dqhdt)).sum(0)) / m.M ** 2
        parts = {k: _hermitian_sum(d[k]) for k in ('KEflux', 'APEflux', 'APEgenspec', 'KEfrictionspec', 'paramspec')}
        scale = sum(abs(_hermitian_sum(np.abs(d[k]))) for k in parts)
        assert abs(sum(parts.values()) - dEdt) < 1e-12 * scale, (band, parts, dEdt)
        assert abs(parts['KEflux']) < 1e-12 * _hermitian_sum(np.abs(d['KEflux']))
        if band <= 0.25:
            assert abs(parts['APEflux']) < 1e-12 * _hermitian_sum(np.abs(d['APEflux']))
        assert parts['KEfrictionspec'] < 0                     # bottom drag only removes energy
        # the parameterization's APE / KE split sums to its total contribution, wavenumber by wavenumber
        np.testing.assert_allclose(d['paramspec_APEflux'] + d['paramspec_KEflux'], d['paramspec'],
                                   atol=1e-12 * np.abs(d['paramspec']).max())
        # KEspec / Ensspec are the spectra of what their names say (Parseval against the grid fields)
        ke_grid = 0.5 * ((m.u ** 2 + m.v ** 2).mean(axis=(1, 2)))
        np.testing.assert_allclose([0.5 * _hermitian_sum(d['KEspec'][z]) for z in (0, 1)], ke_grid, rtol=1e-10)
        np.testing.assert_allclose([_hermitian_sum(d['Ensspec'][z]) for z in (0, 1)], (m.q ** 2).mean(axis=(1, 2)), rtol=1e-10)


def test_enstrophy_budget_and_filter_dissipation_of_the_diagnostics_close():
    """The barotropic-enstrophy diagnostics are a decomposition of the tendency of Z = 1/2 sum_k (H_k/H) |qh_k|^2 / M^2
    WAVENUMBER BY WAVENUMBER: Re[sum_k H_k/H conj(qh_k) dqh_k/dt] / M^2 == ENSflux + ENSgenspec + ENSfrictionspec +
    ENSparamspec for the tendency the model steps with.  The nonlinear flux only redistributes (sum(ENSflux) = 0 once the
    cubic products are alias-free), bottom drag and the filter only remove: ENSDissspec <= 0 everywhere, and over a real
    time step the change of Z and of the energy E equals (budget terms) + the filter's share, i.e. what is left of
    Z^{n+1} - Z^n after the unfiltered AB update is exactly dt * ENSDissspec + the quadratic remainder
    |diss|^2 / 2 (an identity of the update, checked to round-off)."""
    N = 64
    rs = np.random.RandomState(6)
    S = rs.randn(2, N, N) * np.array([7e-12, 2e-13])[:, None, None]
    for band, params in ((1. / 4., {}), (0.95, dict(rek=7e-8, delta=0.1, beta=1e-11))):
        m = qg_ref.QGModelRef(nx=N, dt=14400., parameterization=lambda mm: S, **params)
        q = rs.randn(2, N, N) * np.array([8e-6, 1e-6])[:, None, None]
        m.set_qh(np.fft.rfftn(q, axes=(-2, -1)) * (m.wv < band * m.kk[-1]))
        if band <= 0.25:                       # alias-free cubic products: the enstrophy flux sums to zero
            m._invert()
            m._do_advection()
            f0 = m._diag_functions()['ENSflux']
            assert abs(_hermitian_sum(f0)) < 1e-12 * _hermitian_sum(np.abs(f0))
        for _ in range(3):                     # AB3 history in place (dqhdt_p, dqhdt_pp non-zero, ablevel 2)
            m._step_forward()
        m._invert()
        m._do_advection()
        m._do_friction()
        m._do_q_subgrid_parameterization()
        d = m._diag_functions()
        hr = (m.Hi / m.H)[:, None, None]
        dZdt = (hr * np.real(np.conj(m.qh) * m.dqhdt)).sum(0) / m.M ** 2
        total = d['ENSflux'] + d['ENSgenspec'] + d['ENSfrictionspec'] + d['ENSparamspec']
        scale = sum(np.abs(d[k]).max() for k in ('ENSflux', 'ENSgenspec', 'ENSfrictionspec', 'ENSparamspec'))
        np.testing.assert_allclose(total, dZdt, rtol=0, atol=1e-12 * scale)
        # the filter: unfiltered update u = qh + sum dt_i T_i, filtered qh' = f u, diss = (f - 1) u
        dt1, dt2, dt3 = 23. / 12. * m.dt, -16. / 12. * m.dt, 5. / 12. * m.dt
        unf = m.qh + dt1 * m.dqhdt + dt2 * m.dqhdt_p + dt3 * m.dqhdt_pp
        diss = (m.filtr - 1.0) * unf
        np.testing.assert_allclose(diss, m._dissipation_spectrum(), rtol=0, atol=0)
        assert (m.filtr <= 1.0).all()
        if band > 0.9:
            assert np.abs(diss).max() > 0
        # ENSDissspec * dt == Re[sum hr conj(qh) diss] / M^2; against the filtered state of a real step:
        # Z(f u) - Z(u) = Re[sum hr conj(u) diss] + |diss|^2/2 (per wavenumber), and conj(u) = conj(qh) + O(dt)
        qh_n, ph_n = m.qh.copy(), m.ph.copy()
        np.testing.assert_allclose(d['ENSDissspec'] * m.dt, (hr * np.real(np.conj(qh_n) * diss)).sum(0) / m.M ** 2,
                                   rtol=0, atol=1e-13 * max(np.abs(d['ENSDissspec']).max() * m.dt, 1e-300))
        np.testing.assert_allclose(d['Dissspec'] * m.dt, -(hr * np.real(np.conj(ph_n) * diss)).sum(0) / m.M ** 2,
                                   rtol=0, atol=1e-13 * max(np.abs(d['Dissspec']).max() * m.dt, 1e-300))
        m._forward_timestep()
        np.testing.assert_allclose(m.qh, m.filtr * unf, rtol=0, atol=1e-15 * np.abs(unf).max())
        Zf = 0.5 * (hr * np.abs(m.filtr * unf) ** 2).sum(0)
        Zu = 0.5 * (hr * np.abs(unf) ** 2).sum(0)
        ident = (hr * (np.real(np.conj(unf) * diss) + 0.5 * np.abs(diss) ** 2)).sum(0)
        np.testing.assert_allclose(Zf - Zu, ident, rtol=0, atol=1e-12 * np.abs(Zu).max())
        assert (Zf <= Zu).all()                                                # the filter only removes enstrophy


def test_published_48x48_log_is_reproduced():
    """Google-Colab/online-simulations.ipynb:318-347 (48 x 48, dt = 7200 s): CFL 0.009 while the flow is at
    rest, KE growing by x2.8-2.9 per 1000 steps late in the linear stage (the printed run carries the GAN
    parameterization, whose forcing is negligible at those amplitudes), saturation near step 9000-11000."""
    m = qg_ref.QGModelRef(nx=48, dt=7200., tmax=7200. * 8000, twrite=1000)
    qg_ref.set_initial_condition(m, np.random.RandomState(0))
    ke = {}
    for t in m.run_with_snapshots(tsnapint=7200. * 1000):
        ke[m.tc] = m._calc_ke()
        if m.tc == 1000:
            assert abs(m.cfl - 0.009) < 5e-4
    for a, b in ((4000, 5000), (5000, 6000), (6000, 7000)):
        assert 2.4 < ke[b] / ke[a] < 3.4, (a, b, ke[b] / ke[a])
    assert 1e-8 < ke[1000] < 5e-6          # published: 1.4e-7 .. 5.7e-7 at step 1000 (unseeded initial conditions)


def test_oracle_long_run_matches_the_references_published_dataset_checksums():
    """The DIRECT pin of this restatement against real pyqg output: the reference publishes the std of the coarse-grained
    PV and of the subgrid forcing of its dataset `eddy/64/sharp` (Google-Colab/dataset.ipynb cell 16; 300 runs of
    256 x 256, 10 years, Operator1, no dealiasing).  Eight oracle members run with that protocol are cached in
    tests/golden/oracle_forcing_dataset_stats.npz (per-member sums; ~10 CPU-minutes per member,
    tests/golden/make_oracle_forcing_stats.py).  Stated tolerance: 4 standard errors of the 8-member estimate + 0.3 %."""
    import os
    import pytest
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'oracle_forcing_dataset_stats.npz')
    if not os.path.exists(path):
        pytest.skip('tests/golden/oracle_forcing_dataset_stats.npz not generated')
    s = np.load(path)['sums']                       # columns: n_q, sum_q, sum_q2, n_f, sum_f, sum_f2, snapshots
    assert s.shape[0] >= 4 and (s[:, 6] == 86).all()
    for (n, s1, s2), published in (((0, 1, 2), 5.701264812550008e-06), ((3, 4, 5), 4.999136229013802e-12)):
        tot = s.sum(0)
        total = np.sqrt(tot[s2] / tot[n] - (tot[s1] / tot[n]) ** 2)
        per_run = np.sqrt(s[:, s2] / s[:, n] - (s[:, s1] / s[:, n]) ** 2)
        se = per_run.std(ddof=1) / np.sqrt(len(per_run))
        print(f'oracle {total:.6e} published {published:.6e} ({100 * (total / published - 1):+.3f} %), standard error {100 * se / published:.2f} %')
        assert abs(total - published) <= 4 * se + 0.003 * published
