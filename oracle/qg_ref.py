"""CPU restatement (numpy, float64) of the 2-layer pseudo-spectral QG core.

TEST INFRASTRUCTURE — see oracle/__init__.py.  PARITY UNPINNED at step level: the
arithmetic restated here lives in the third-party package ``pyqg`` (version 0.7.2 in
the reference's published logs, Google-Colab/online-simulations.ipynb:40,99), which
is neither vendored under /root/reference nor installable offline, and the reference
holds no per-step fixture of it.  What the reference does hold is two published
checksums of a dataset produced by real pyqg (Google-Colab/dataset.ipynb cell 16);
tests/test_oracle_qg.py holds a cached long run of this restatement to them
(tests/golden/make_oracle_forcing_stats.py), statistically.  The
restatement follows pyqg 0.7.2's published algorithm

    pyqg/model.py      _initialize_grid, _initialize_filter, _initialize_time,
                       _step_forward, run_with_snapshots, _print_status, spec_var
    pyqg/qg_model.py   _initialize_background, _initialize_inversion_matrix,
                       set_q1q2, _calc_cfl, _calc_ke, QG diagnostics
    pyqg/kernel.pyx    _invert, _do_advection, _do_friction,
                       _do_q_subgrid_parameterization, _forward_timestep

and is anchored on the reference's call sites
(pyqg_generative/tools/simulate.py:83,121,131-132,137-138,147-168;
tools/stochastic_pyqg.py:74-88; tools/operators.py:229-234,241-247;
models/parameterization.py:13-34) and on the invariants its notebooks check
(notebooks/3-2-dealiasing.ipynb cells 11-13).
"""
import numpy as np

pi = np.pi


class QGModelRef:
    """Single-member two-layer QG model; attribute names follow pyqg.QGModel."""

    def __init__(self, nx=64, ny=None, L=1e6, W=None, dt=7200., twrite=1000.,
                 tmax=1576800000., tavestart=315360000., taveint=86400.,
                 rek=5.787e-7, filterfac=23.6, beta=1.5e-11, rd=15000.0,
                 delta=0.25, H1=500, U1=0.025, U2=0.0,
                 parameterization=None, q_parameterization=None,
                 log_level=0, **unused):
        self.nz = 2
        self.nx = int(nx)
        self.ny = int(ny) if ny is not None else int(nx)
        self.L = float(L)
        self.W = float(W) if W is not None else float(L)
        self.dt = float(dt)
        self.twrite = int(twrite)
        self.tmax = float(tmax)
        self.tavestart = float(tavestart)
        self.taveint = float(taveint)
        self.rek = rek
        self.filterfac = filterfac
        self.beta = beta
        self.rd = rd
        self.delta = delta
        self.H1 = H1
        self.U1 = U1
        self.U2 = U2
        self.log_level = log_level
        # pyqg: `parameterization=` is dispatched on .parameterization_type;
        # the reference only ever passes q-parameterizations
        # (models/parameterization.py:13).
        self.q_parameterization = q_parameterization or parameterization

        self._initialize_grid()
        self._initialize_background()
        self._initialize_inversion_matrix()
        self._initialize_filter()
        self._initialize_time()

        nz, ny_, nx_, nl, nk = self.nz, self.ny, self.nx, self.nl, self.nk
        self.q = np.zeros((nz, ny_, nx_))
        self.qh = np.zeros((nz, nl, nk), complex)
        self.ph = np.zeros((nz, nl, nk), complex)
        self.u = np.zeros((nz, ny_, nx_))
        self.v = np.zeros((nz, ny_, nx_))
        self.dqhdt = np.zeros((nz, nl, nk), complex)
        self.dqhdt_p = np.zeros((nz, nl, nk), complex)
        self.dqhdt_pp = np.zeros((nz, nl, nk), complex)
        self.diag = {}
        self.diag_count = 0

    # ---- pyqg/model.py::_initialize_grid --------------------------------
    def _initialize_grid(self):
        self.x, self.y = np.meshgrid(
            np.arange(0.5, self.nx, 1.) / self.nx * self.L,
            np.arange(0.5, self.ny, 1.) / self.ny * self.W)
        self.nl = self.ny
        self.nk = self.nx // 2 + 1
        self.dk = 2. * pi / self.L
        self.dl = 2. * pi / self.W
        self.ll = self.dl * np.append(np.arange(0., self.nx / 2),
                                      np.arange(-self.nx / 2, 0.))
        self.kk = self.dk * np.arange(0., self.nk)
        self.k, self.l = np.meshgrid(self.kk, self.ll)
        self.ik = 1j * self.k
        self.il = 1j * self.l
        self._ik = self.ik[0, :].copy()
        self._il = self.il[:, 0].copy()
        self.dx = self.L / self.nx
        self.dy = self.W / self.ny
        self.M = self.nx * self.ny
        self.wv2 = self.k ** 2 + self.l ** 2
        self.wv = np.sqrt(self.wv2)
        iwv2 = self.wv2 != 0.
        self.wv2i = np.zeros_like(self.wv2)
        self.wv2i[iwv2] = self.wv2[iwv2] ** -1

    # ---- pyqg/qg_model.py::_initialize_background -----------------------
    def _initialize_background(self):
        self.H = self.H1 + self.H1 / self.delta  # Hi = [H1, H2], H2 = H1/delta
        self.Hi = np.array([self.H1, self.H1 / self.delta])
        self.Ubg = np.array([self.U1, self.U2])
        self.F1 = self.rd ** -2 / (1. + self.delta)
        self.F2 = self.delta * self.F1
        self.Qy1 = self.beta + self.F1 * (self.U1 - self.U2)
        self.Qy2 = self.beta - self.F2 * (self.U1 - self.U2)
        self.Qy = np.array([self.Qy1, self.Qy2])
        self._ikQy = 1j * (self.kk[np.newaxis, :] * self.Qy[:, np.newaxis])
        self.del1 = self.delta / (self.delta + 1.)
        self.del2 = (self.delta + 1.) ** -1

    # ---- pyqg/qg_model.py::_initialize_inversion_matrix -----------------
    def _initialize_inversion_matrix(self):
        a = np.zeros((2, 2, self.nl, self.nk))
        det = self.wv2 * (self.wv2 + self.F1 + self.F2)
        with np.errstate(divide='ignore', invalid='ignore'):
            det_inv = np.where(det != 0., det ** -1, 0.)
        a[0, 0] = -(self.wv2 + self.F2) * det_inv
        a[0, 1] = -self.F1 * det_inv
        a[1, 0] = -self.F2 * det_inv
        a[1, 1] = -(self.wv2 + self.F1) * det_inv
        self.a = a

    # ---- pyqg/model.py::_initialize_filter ------------------------------
    def _initialize_filter(self):
        cphi = 0.65 * pi
        wvx = np.sqrt((self.k * self.dx) ** 2. + (self.l * self.dy) ** 2.)
        filtr = np.exp(-self.filterfac * (wvx - cphi) ** 4.)
        filtr[wvx <= cphi] = 1.
        self.filtr = filtr

    def _initialize_time(self):
        self.t = 0.
        self.tc = 0
        self.ablevel = 0
        self.taveints = np.ceil(self.taveint / self.dt)

    # ---- FFT conventions of pyqg (numpy-compatible rfft2) ----------------
    @staticmethod
    def fft(x):
        return np.fft.rfftn(x, axes=(-2, -1))

    @staticmethod
    def ifft(xh):
        return np.fft.irfftn(xh, axes=(-2, -1))

    # ---- state setters (kernel.pyx property q / qh; qg_model.set_q1q2) ----
    def set_q(self, q):
        self.q = np.array(q, dtype='float64')
        self.qh = self.fft(self.q)

    def set_qh(self, qh):
        self.qh = np.array(qh, dtype=complex)
        self.q = self.ifft(self.qh)

    def set_q1q2(self, q1, q2):
        self.set_q(np.vstack([q1[np.newaxis], q2[np.newaxis]]))

    # ---- kernel.pyx::_invert -------------------------------------------
    def _invert(self):
        qh = self.qh
        self.ph = np.empty_like(qh)
        self.ph[0] = self.a[0, 0] * qh[0] + self.a[0, 1] * qh[1]
        self.ph[1] = self.a[1, 0] * qh[0] + self.a[1, 1] * qh[1]
        self.uh = -self.il * self.ph
        self.vh = self.ik * self.ph
        self.u = self.ifft(self.uh)
        self.v = self.ifft(self.vh)

    # ---- kernel.pyx::_do_advection -------------------------------------
    def _do_advection(self):
        self.uq = (self.u + self.Ubg[:, None, None]) * self.q
        self.vq = self.v * self.q
        self.uqh = self.fft(self.uq)
        self.vqh = self.fft(self.vq)
        self.dqhdt = -(self.ik * self.uqh + self.il * self.vqh
                       + self._ikQy[:, None, :] * self.ph)

    # ---- kernel.pyx::_do_friction --------------------------------------
    def _do_friction(self):
        if self.rek:
            self.dqhdt[-1] = self.dqhdt[-1] + self.rek * self.wv2 * self.ph[-1]

    # ---- kernel.pyx::_do_q_subgrid_parameterization ---------------------
    def _do_q_subgrid_parameterization(self):
        self.dq = np.asarray(self.q_parameterization(self), dtype='float64')
        self.dqh = self.fft(self.dq)
        self.dqhdt = self.dqhdt + self.dqh

    # ---- kernel.pyx::_forward_timestep ---------------------------------
    def _forward_timestep(self):
        if self.ablevel == 0:      # forward Euler
            dt1, dt2, dt3 = self.dt, 0.0, 0.0
            self.ablevel = 1
        elif self.ablevel == 1:    # AB2
            dt1, dt2, dt3 = 1.5 * self.dt, -0.5 * self.dt, 0.0
            self.ablevel = 2
        else:                      # AB3
            dt1 = 23. / 12. * self.dt
            dt2 = -16. / 12. * self.dt
            dt3 = 5. / 12. * self.dt
        qh_new = self.filtr * (self.qh + dt1 * self.dqhdt
                               + dt2 * self.dqhdt_p + dt3 * self.dqhdt_pp)
        self.dqhdt_pp = self.dqhdt_p
        self.dqhdt_p = self.dqhdt
        self.set_qh(qh_new)
        self.tc += 1
        self.t += self.dt

    # ---- model.py::_step_forward ----------------------------------------
    def _step_forward(self):
        self._invert()
        self._do_advection()
        self._do_friction()
        if self.q_parameterization is not None:
            self._do_q_subgrid_parameterization()
        self._calc_diagnostics()
        self._forward_timestep()
        self._print_status()

    def run_with_snapshots(self, tsnapstart=0., tsnapint=432000.):
        tsnapints = np.ceil(tsnapint / self.dt)
        while self.t < self.tmax:
            self._step_forward()
            if self.t >= tsnapstart and (self.tc % tsnapints) == 0:
                yield self.t

    def run(self):
        while self.t < self.tmax:
            self._step_forward()

    # ---- status: model.py::_print_status, qg_model._calc_cfl/_calc_ke ----
    def spec_var(self, ph):
        var_dens = 2. * np.abs(ph) ** 2 / self.M ** 2
        var_dens[..., 0] = var_dens[..., 0] / 2.
        var_dens[..., -1] = var_dens[..., -1] / 2.
        return var_dens.sum()

    def _calc_cfl(self):
        return np.abs(np.hstack([self.u + self.Ubg[:, None, None], self.v])
                      ).max() * self.dt / self.dx

    def _calc_ke(self):
        ke1 = .5 * self.Hi[0] * self.spec_var(self.wv * self.ph[0])
        ke2 = .5 * self.Hi[1] * self.spec_var(self.wv * self.ph[1])
        return (ke1 + ke2) / self.H

    def _print_status(self):
        if (self.tc % self.twrite) == 0:
            self.ke = self._calc_ke()
            self.cfl = self._calc_cfl()
            if self.log_level:
                print('Step: %i, Time: %3.2e, KE: %3.2e, CFL: %4.3f'
                      % (self.tc, self.t, self.ke, self.cfl))
            assert self.cfl < 1., 'CFL condition violated'

    # ---- time-averaged diagnostics (model.py/_qg_model.py) ---------------
    def _diag_functions(self):
        """Instantaneous spectral diagnostics of pyqg 0.7.2 used downstream by the
        reference (tools/comparison_tools.py:116-195; notebooks: KEspec)."""
        M2 = self.M ** 2
        d = {}
        d['KEspec'] = self.wv2 * np.abs(self.ph) ** 2 / M2
        d['Ensspec'] = np.abs(self.qh) ** 2 / M2
        d['entspec'] = np.abs(self.del1 * self.qh[0] + self.del2 * self.qh[1]) ** 2 / M2
        d['APEflux'] = self.rd ** -2 * self.del1 * self.del2 * np.real(
            (self.ph[0] - self.ph[1]) * np.conj(self._Jptpc_h())) / M2
        d['KEflux'] = np.real(self.del1 * self.ph[0] * np.conj(self._Jpxi_h()[0])
                              + self.del2 * self.ph[1] * np.conj(self._Jpxi_h()[1])) / M2
        d['APEgenspec'] = (self.U1 - self.U2) * self.rd ** -2 * self.del1 * self.del2 * np.real(
            1j * self.k * (self.del1 * self.ph[0] + self.del2 * self.ph[1])
            * np.conj(self.ph[0] - self.ph[1])) / M2
        d['KEfrictionspec'] = -self.rek * self.del2 * self.wv2 * np.abs(self.ph[1]) ** 2 / M2
        # ---- the barotropic-enstrophy budget and the filter's dissipation (pyqg model.py::_initialize_core_diagnostics:
        # Dissspec, ENSDissspec, ENSfrictionspec, ENSparamspec; qg_model.py::_initialize_model_diagnostics: ENSflux,
        # ENSgenspec — the remaining keys the reference reads, comparison_tools.py:222-225).  Each is
        # Re[sum_k Hk/H conj(qh_k) X_k] / M^2 for one term X_k of the PV tendency (energy: -conj(ph_k) in place of conj(qh_k)).
        hr = (self.Hi / self.H)[:, None, None]
        Jq = self._advect_h(self.q, self.u, self.v)           # model.py::_advect with the perturbation velocities
        d['ENSflux'] = -(hr * np.real(np.conj(self.qh) * Jq)).sum(axis=0) / M2
        d['ENSgenspec'] = -(hr * np.real(np.conj(self.qh) * self._ikQy[:, None, :] * self.ph)).sum(axis=0) / M2
        d['ENSfrictionspec'] = self.rek * self.Hi[-1] / self.H * self.wv2 * np.real(np.conj(self.qh[-1]) * self.ph[-1]) / M2
        diss = self._dissipation_spectrum()
        d['Dissspec'] = -(hr * np.real(np.conj(self.ph) * diss)).sum(axis=0) / self.dt / M2
        d['ENSDissspec'] = (hr * np.real(np.conj(self.qh) * diss)).sum(axis=0) / self.dt / M2
        d['EKE'] = 0.5 * (self.v ** 2).mean(axis=(-1, -2))
        d['EKEdiss'] = self.Hi[-1] / self.H * self.rek * (self.v[-1] ** 2 + self.u[-1] ** 2).mean()
        if self.q_parameterization is not None and hasattr(self, 'dqh'):
            d['paramspec'] = -np.real((self.Hi[:, None, None] / self.H
                                       * np.conj(self.ph) * self.dqh).sum(axis=0)) / M2
            # the two parts of the energy budget the reference sums into its total flux
            # (comparison_tools.py:174-176): with dph = a . dqh (the model's inversion of the
            # parameterization's PV tendency), APE part + KE part == paramspec identically
            dph = np.einsum('ij...,j...->i...', self.a, self.dqh)
            d['paramspec_APEflux'] = self.rd ** -2 * self.del1 * self.del2 * np.real(
                (self.ph[0] - self.ph[1]) * np.conj(dph[0] - dph[1])) / M2
            d['paramspec_KEflux'] = self.wv2 * np.real(self.del1 * self.ph[0] * np.conj(dph[0])
                                                       + self.del2 * self.ph[1] * np.conj(dph[1])) / M2
            d['ENSparamspec'] = (hr * np.real(np.conj(self.qh) * self.dqh)).sum(axis=0) / M2
        return d

    def _dissipation_spectrum(self):
        """model.py::_initialize_core_diagnostics::dissipation_spectrum: what the exponential filter removes from the
        AB update the coming _forward_timestep will make, (filtr - 1) (qh + dt1 dqhdt + dt2 dqhdt_p + dt3 dqhdt_pp), with
        the coefficients of the CURRENT ablevel (diagnostics are evaluated before the time step)."""
        if self.ablevel == 0:
            dt1, dt2, dt3 = self.dt, 0.0, 0.0
        elif self.ablevel == 1:
            dt1, dt2, dt3 = 1.5 * self.dt, -0.5 * self.dt, 0.0
        else:
            dt1, dt2, dt3 = 23. / 12. * self.dt, -16. / 12. * self.dt, 5. / 12. * self.dt
        return (self.filtr - 1.0) * (self.qh + dt1 * self.dqhdt + dt2 * self.dqhdt_p + dt3 * self.dqhdt_pp)

    def _advect_h(self, q, u, v):
        """-> spectral J(psi, q) in flux form (model.py::_advect)."""
        uq = u * q
        vq = v * q
        return self.ik * self.fft(uq) + self.il * self.fft(vq)

    def _Jptpc_h(self):
        p = self.ifft(self.ph)
        return -self._advect_h(p[0] - p[1],
                               self.del1 * self.u[0] + self.del2 * self.u[1],
                               self.del1 * self.v[0] + self.del2 * self.v[1])

    def _Jpxi_h(self):
        xi = self.ifft(-self.wv2 * self.ph)
        return self._advect_h(xi, self.u, self.v)

    def _calc_diagnostics(self):
        if (self.t >= self.dt) and (self.t >= self.tavestart) and \
                (self.tc % self.taveints == 0):
            self._increment_diagnostics()

    def _increment_diagnostics(self):
        d = self._diag_functions()
        n = self.diag_count
        for key, val in d.items():
            if key in self.diag:
                self.diag[key] = (self.diag[key] * n + val) / (n + 1)
            else:
                self.diag[key] = np.array(val, copy=True)
        self.diag_count = n + 1

    def get_diagnostic(self, name):
        return self.diag[name]


def set_initial_condition(m, rng=None):
    """Band-limited random IC of the reference (tools/simulate.py:147-168).

    The reference draws from numpy's *unseeded* global MT19937 stream with
    ``np.random.rand(ny,nx)`` then ``np.random.rand(1,nx)``; pass a
    ``np.random.RandomState`` to make the draw reproducible (same call order).
    """
    rng = rng if rng is not None else np.random
    q2d = 1e-7 * rng.rand(m.ny, m.nx)
    q2d -= q2d.mean(axis=(-2, -1), keepdims=True)
    q2d *= np.sqrt(m.nx * m.ny / 64 ** 2)
    q1d = 1e-6 * (np.ones((m.ny, 1)) * rng.rand(1, m.nx))
    q1d -= q1d.mean(axis=(-2, -1), keepdims=True)
    q1d *= np.sqrt(m.nx / 64)
    noise = q1d + q2d
    Xf = np.fft.rfftn(noise)
    noise = np.fft.irfftn(Xf * (m.wv < np.pi / (m.L / 32)))
    m.set_q1q2(noise, 0 * m.x)
    m._invert()
    return noise
