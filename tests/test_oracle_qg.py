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
