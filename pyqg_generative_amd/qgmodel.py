"""pyqg.QGModel-compatible facade over the device engine.

Mirrors the attribute/method surface the reference uses on ``pyqg.QGModel``
(q, u, v, ph, qh, p, ik, il, k, l, kk, ll, wv, filtr, dk, dl, dx,
dt, t, nx, ny, L, M, x, dqhdt, fft, ifft, _invert, set_q1q2, run_with_snapshots,
to_dataset ...; call sites pyqg_generative/tools/simulate.py:83,121,131-138,147-168,
tools/operators.py:229-246, tools/spectral_tools.py:142-152).

New relative to pyqg: ``n_members`` (B) members are advanced together.  With
``n_members == 1`` every array attribute has pyqg's shape ((2,N,N) / (2,N,N/2+1));
with B > 1 a leading member axis is added.  All arithmetic runs on the GPU through
libqgx.so; arrays cross to the host only when an attribute is read.
"""
import math
import numpy as np
import torch

from . import _lib
from .engine import EnsembleEngine, PYQG_DEFAULTS


class QParameterization:
    """Stand-in for pyqg.QParameterization: subclasses implement __call__(m) -> (nz,ny,nx)."""
    parameterization_type = 'q_parameterization'

    def __mul__(self, w):
        return WeightedParameterization(self, float(w))
    __rmul__ = __mul__


class WeightedParameterization(QParameterization):
    """``model_weight * parameterization`` (reference: tools/simulate.py:242)."""

    def __init__(self, param, weight):
        self.param, self.weight = param, weight

    def __call__(self, m):
        return self.weight * self.param(m)

    def __mul__(self, w):
        return WeightedParameterization(self.param, self.weight * float(w))
    __rmul__ = __mul__


def _unwrap(param):
    """-> (base parameterization, weight)"""
    w = 1.0
    while isinstance(param, WeightedParameterization):
        w *= param.weight
        param = param.param
    return param, w


class QGModel:
    def __init__(self, nx=64, ny=None, L=1e6, W=None, dt=7200., twrite=1000., tmax=1576800000.,
                 tavestart=315360000., taveint=86400., rek=5.787e-7, filterfac=23.6, beta=1.5e-11,
                 rd=15000.0, delta=0.25, H1=500, U1=0.025, U2=0.0, parameterization=None,
                 q_parameterization=None, log_level=1, n_members=1, device=0, seed=0, member_offset=0,
                 **unused):
        if ny is not None and ny != nx:
            raise ValueError('only square grids (ny == nx) are supported')
        if W is not None and W != L:
            raise ValueError('only square domains (W == L) are supported')
        self.nz, self.nx, self.ny = 2, int(nx), int(nx)
        self.L = self.W = float(L)
        self.dt, self.tmax = float(dt), float(tmax)
        self.twrite = int(twrite)
        self.tavestart, self.taveint = float(tavestart), float(taveint)
        self.rek, self.filterfac, self.beta, self.rd, self.delta = rek, filterfac, beta, rd, delta
        self.H1, self.U1, self.U2 = H1, U1, U2
        self.log_level = log_level
        self.ntd = 1                      # pyqg: FFTW threads; exported as the attribute pyqg:ntd
        self.n_members = int(n_members)
        self.seed, self.member_offset = int(seed), int(member_offset)
        self.q_parameterization = q_parameterization or parameterization
        self._eng = EnsembleEngine(nx=nx, n_members=n_members, device=device, L=L, dt=dt, rek=rek,
                                   delta=delta, beta=beta, rd=rd, U1=U1, U2=U2, H1=H1,
                                   filterfac=filterfac)
        self._init_grid()
        self.t = 0.
        self.taveints = math.ceil(self.taveint / self.dt)
        # pyqg averages diagnostics every taveints steps once t >= tavestart (model.py::_calc_diagnostics)
        self._eng.diag_config(math.ceil(self.tavestart / self.dt - 1e-9), self.taveints)
        # pyqg's default initial condition is overwritten by every reference call site
        # (simulate.py:85,128 set_initial_condition); start from rest.

    # ---- grid constants (computed by the library; exposed with pyqg's names) ----
    def _init_grid(self):
        e = self._eng
        N = self.nx
        self.nl, self.nk = N, N // 2 + 1
        self.kk, self.ll = e.table(_lib.T_KK), e.table(_lib.T_LL)
        self.dk = self.dl = 2. * math.pi / self.L
        self.k, self.l = np.meshgrid(self.kk, self.ll)
        self.ik, self.il = 1j * self.k, 1j * self.l
        self._ik, self._il = self.ik[0, :].copy(), self.il[:, 0].copy()
        self.dx = self.dy = self.L / N
        self.M = N * N
        self.wv2 = e.table(_lib.T_WV2)
        self.wv = np.sqrt(self.wv2)
        self.wv2i = np.zeros_like(self.wv2)
        nz = self.wv2 != 0
        self.wv2i[nz] = self.wv2[nz] ** -1
        self.filtr = e.table(_lib.T_FILTR)
        self.a = e.table(_lib.T_A)
        self.x, self.y = np.meshgrid(np.arange(0.5, N, 1.) / N * self.L, np.arange(0.5, N, 1.) / N * self.W)
        self.Hi = np.array([self.H1, self.H1 / self.delta])
        self.H = self.Hi.sum()
        self.Ubg = np.array([self.U1, self.U2])
        self.F1 = self.rd ** -2 / (1. + self.delta)
        self.F2 = self.delta * self.F1
        self.Qy1 = self.beta + self.F1 * (self.U1 - self.U2)
        self.Qy2 = self.beta - self.F2 * (self.U1 - self.U2)
        self.Qy = np.array([self.Qy1, self.Qy2])
        self.del1 = self.delta / (self.delta + 1.)
        self.del2 = (self.delta + 1.) ** -1

    # ---- state access ---------------------------------------------------------------
    def _host(self, field, **kw):
        a = self._eng.get(field, **kw).cpu().numpy()
        return a[0] if self.n_members == 1 else a

    def _bcast(self, a, tail):
        a = np.asarray(a)
        if a.shape == tail:
            a = np.array(np.broadcast_to(a, (self.n_members,) + tail))
        if a.shape != (self.n_members,) + tail:
            raise ValueError(f'expected shape {tail} or {(self.n_members,) + tail}, got {a.shape}')
        return np.ascontiguousarray(a)

    @property
    def q(self):
        return self._host(_lib.F_Q)

    @q.setter
    def q(self, value):      # kernel.pyx property q: also refreshes qh
        self._eng.set_q(self._bcast(np.asarray(value, dtype='float64'), (2, self.ny, self.nx)))

    @property
    def qh(self):
        return self._host(_lib.F_QH)

    @qh.setter
    def qh(self, value):
        self._eng.set_qh(self._bcast(np.asarray(value, dtype='complex128'), (2, self.nl, self.nk)))

    def set_q1q2(self, q1, q2, check=False):
        q1, q2 = np.asarray(q1, 'float64'), np.asarray(q2, 'float64')
        self.q = np.stack([q1, q2], axis=-3)

    ph = property(lambda self: self._host(_lib.F_PH))
    u = property(lambda self: self._host(_lib.F_U))
    v = property(lambda self: self._host(_lib.F_V))
    dqhdt = property(lambda self: self._host(_lib.F_DQHDT))
    dqhdt_p = property(lambda self: self._host(_lib.F_DQHDT_P))
    dqhdt_pp = property(lambda self: self._host(_lib.F_DQHDT_PP))
    PV_forcing = property(lambda self: self._host(_lib.F_S))
    ufull = property(lambda self: self.u + self.Ubg[:, None, None])
    vfull = property(lambda self: self.v)

    p = property(lambda self: self._host(_lib.F_P))      # irfft2(ph) on the device (pyqg: _calc_derived_fields)

    @property
    def uh(self):                                         # kernel.pyx::_invert: uh = -il ph, vh = ik ph
        return -self.il * self.ph

    @property
    def vh(self):
        return self.ik * self.ph

    @property
    def dqdt(self):
        return self.ifft(self.dqhdt)

    @property
    def tc(self):
        return self._eng.tc

    # device tensors for callers that want to stay on the GPU
    def q_device(self):
        return self._eng.get(_lib.F_Q)

    # ---- pyqg's m.fft / m.ifft over the last two axes, executed on the device ----------
    # numpy in -> numpy out (pyqg's contract); a CUDA tensor in -> a CUDA tensor out, no host round trip.  The transforms
    # run on the plans of tools.operators.Dev (an LRU cache keyed by grid size and field count), so calls inside loops
    # — the reference's `divergence` transforms field by field — neither allocate engines nor leak them.
    def fft(self, x):
        from .tools.operators import Dev
        on_dev = torch.is_tensor(x)
        xt = x.to(torch.float64) if on_dev else torch.as_tensor(np.asarray(x, dtype='float64'), device=self._eng.device)
        lead = tuple(xt.shape[:-2])
        out = Dev.rfft2(xt.reshape((-1, self.ny, self.nx))).reshape(lead + (self.nl, self.nk))
        return out if on_dev else out.cpu().numpy()

    def ifft(self, xh):
        from .tools.operators import Dev
        on_dev = torch.is_tensor(xh)
        xt = xh.to(torch.complex128) if on_dev else torch.as_tensor(np.asarray(xh, dtype='complex128'), device=self._eng.device)
        lead = tuple(xt.shape[:-2])
        out = Dev.irfft2(xt.reshape((-1, self.nl, self.nk))).reshape(lead + (self.ny, self.nx))
        return out if on_dev else out.cpu().numpy()

    # ---- dynamics -------------------------------------------------------------------
    def _invert(self):
        self._eng.invert()

    def _step_kwargs(self):
        """Translate the attached parameterization into qgx_step arguments."""
        param, weight = _unwrap(self.q_parameterization)
        gen = getattr(param, 'device_generator', None)
        if gen is None:
            return None, param, weight
        sampling = getattr(self, 'sampling_type', 'AR1')
        if sampling == 'deterministic':
            return None, param, weight          # host-driven predict_mean_snapshot path
        nsteps = self.noise_sampler.nsteps
        return dict(generator=gen(), sampling=sampling, nsteps_decor=nsteps, weight=weight,
                    seed=self.seed, member_offset=self.member_offset), param, weight

    def _advance(self, n, refresh_diag=True):
        if n <= 0:
            return
        if self.q_parameterization is None:
            self._eng.step(n, refresh_diag=refresh_diag)
        else:
            kw, param, weight = self._step_kwargs()
            if kw is not None:               # fused on-device plugin
                self._eng.step(n, refresh_diag=refresh_diag, **kw)
            else:                            # generic pyqg plugin: one host call per step
                for s in range(n):
                    dq = np.asarray(param(self), dtype='float64')
                    f = torch.as_tensor(self._bcast(dq, (2, self.ny, self.nx))).to(self._eng.device)
                    self._eng.step(1, forcing=f, weight=weight, demean=False,
                                   refresh_diag=refresh_diag and s == n - 1)
        self.t += n * self.dt

    def _step_forward(self):
        self._advance(1)
        self._print_status()

    def _steps_until(self, *periods):
        """largest run of steps that stops at the next multiple of any period (or tmax)"""
        tc = self.tc
        left = int(math.ceil((self.tmax - self.t) / self.dt - 1e-9))
        n = max(left, 1) if self.t < self.tmax else 0
        for p in periods:
            if p and p > 0:
                n = min(n, int(p - tc % p))
        return n

    def run_with_snapshots(self, tsnapstart=0., tsnapint=432000.):
        tsnapints = int(math.ceil(tsnapint / self.dt))
        while self.t < self.tmax:
            n = self._steps_until(tsnapints, self.twrite)
            self._advance(n)
            self._print_status()
            if self.t >= tsnapstart and (self.tc % tsnapints) == 0:
                yield self.t

    def run(self):
        while self.t < self.tmax:
            self._advance(self._steps_until(self.twrite))
            self._print_status()

    # ---- status (model.py::_print_status) -------------------------------------------
    def _calc_ke(self):
        ke, _ = self._eng.status()
        return ke[0] if self.n_members == 1 else ke

    def _calc_cfl(self):
        _, cfl = self._eng.status()
        return cfl[0] if self.n_members == 1 else cfl

    def _print_status(self):
        if (self.tc % self.twrite) == 0:
            ke, cfl = self._eng.status()
            self.ke, self.cfl = (ke[0], cfl[0]) if self.n_members == 1 else (ke, cfl)
            if self.log_level:
                print('Step: %i, Time: %3.2e, KE: %3.2e, CFL: %4.3f'
                      % (self.tc, self.t, float(np.mean(ke)), float(np.max(cfl))))
            assert np.all(cfl < 1.), 'CFL condition violated'

    # ---- time-averaged diagnostics (pyqg Model.get_diagnostic) ------------------------------
    diagnostic_names = tuple(_lib.DIAGS)

    def get_diagnostic(self, name):
        """Time mean of a spectral diagnostic, pyqg shape ((2,nl,nk) or (nl,nk)); a leading member axis
        when n_members > 1.  Raises if nothing has been averaged yet (t < tavestart)."""
        a = self._eng.diag(name).cpu().numpy()
        return a[0] if self.n_members == 1 else a

    def ensemble_mean_diagnostic(self, name, group=None):
        """Mean over ALL members of the job (all ranks): one all-reduce of the per-rank sum
        (reference: ds[spec].mean('run'), comparison_tools.py:167-168)."""
        from . import parallel
        local = self._eng.diag(name)
        return parallel.ensemble_mean(local.sum(0), self.n_members, group).cpu().numpy()

    @property
    def diagnostics_count(self):
        return self._eng.diag_count

    def _calc_derived_fields(self):
        pass    # pyqg caches p, xi, Jptpc ... here; this facade derives p / uh / vh / dqdt on access

    def to_dataset(self, variables=None):
        """pyqg's Model.to_dataset(): state (+ time-averaged diagnostics once t >= tavestart) in pyqg's
        xarray layout (xarray_output.py; reference call sites simulate.py:93,105,133,138).  ``variables``
        restricts the exported state fields (run_simulation exports only what survives drop_vars)."""
        from . import xarray_output
        self._eng.check_generators()      # a snapshot of a state corrupted by a range overflow must not be written
        names = xarray_output.VARIABLES if variables is None else variables
        return xarray_output.model_to_dataset(self, fields={n: getattr(self, n) for n in names})

    def close(self):
        self._eng.close()
