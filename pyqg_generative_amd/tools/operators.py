"""Coarse-graining operators, FFT re-gridding and the subgrid-forcing diagnostic, executed on
the GPU (reference: pyqg_generative/tools/operators.py:84-99 gauss_filter / model_filter,
:117-132 cut_off, :134-190 fft_interpolate, :192-202 clean_2h, :204-217 Operator1/2/4/5,
:219-236 apply_operator_to_model, :241-247 divergence, :249-268 advect, :283-287
PV_subgrid_forcing).

Public functions keep the reference's names and numpy-in / numpy-out calling convention
(2-D (Ny,Nx) or stacked (..., Ny,Nx) arrays); ``Dev`` holds the same operators on device
tensors so that a pipeline (e.g. generate_subgrid_forcing) never leaves the GPU.  All
arithmetic is in libqgx.so (qgx_rfft2 / qgx_irfft2 / qgx_spec_regrid / qgx_spec_div /
qgx_real_fma); torch only owns the buffers.
"""
import collections
import ctypes as C
import math
import numpy as np
import torch

from .. import _lib
from .._lib import lib, check
from ..engine import EnsembleEngine, _ptr, _stream
from ..qgmodel import QGModel

FILTER_2h_HARMONICS = True


class Dev:
    """Operators on device tensors of shape (M, N, N) float64 (M fields) / (M, N, N/2+1) complex128.

    FFT plans and inversion models are engines (device allocations) kept in ONE keyed LRU cache:
    ('fft', N, M, device) or ('inv', N, B, device, params).  ``Dev.close()`` frees everything;
    the least recently used engine is closed when more than ``MAX_PLANS`` are alive.  The cache must hold the WORKING
    SET of one snapshot of the reference's own forcing-dataset run or every lookup of its cyclic access pattern misses
    (run_forcing_datasets.py: Nc = [32, 48, 64, 96, 128] with the 3/2-rule keeps 15 keys live — transforms of 384, 32,
    48, 64, 96, 128, 72, 144, 192 and inversions of 256, 32, 48, 64, 96, 128); transform plans are state-less handles
    (``plan_only``: tables + work space), so 64 of them cost little."""
    L = 1e6
    MAX_PLANS = 64
    _plans = collections.OrderedDict()
    _tables = {}
    _PARAM_KEYS = ('rek', 'delta', 'beta', 'rd', 'U1', 'U2', 'H1', 'L')

    # ---- plumbing ---------------------------------------------------------------------------
    @classmethod
    def _engine(cls, key, make):
        e = cls._plans.get(key)
        if e is None:
            e = cls._plans[key] = make()
            while len(cls._plans) > cls.MAX_PLANS:
                _, old = cls._plans.popitem(last=False)
                old.close()
        else:
            cls._plans.move_to_end(key)
        return e

    @classmethod
    def plan(cls, N, M, device=0):
        return cls._engine(('fft', N, M, device),
                           lambda: EnsembleEngine(nx=N, n_members=M // 2, device=device, L=cls.L, plan_only=True))

    @classmethod
    def inversion_model(cls, N, B, device, pyqg_params):
        """the cached engine that inverts (B,2,N,N) PV fields with these physical parameters"""
        kw = {k: v for k, v in pyqg_params.items() if k in cls._PARAM_KEYS}
        return cls._engine(('inv', N, B, device, tuple(sorted(kw.items()))),
                           lambda: EnsembleEngine(nx=N, n_members=B, device=device, **kw))

    @classmethod
    def close(cls):
        for e in cls._plans.values():
            e.close()
        cls._plans.clear()
        cls._tables.clear()

    @classmethod
    def table(cls, N, name, device=0):
        """real (N,N/2+1) spectral tables on the device: 'filtr' (pyqg exponential filter),
        ('gauss', ratio) Gaussian filter of width ratio*dx (operators.py:87-90)."""
        key = (N, name, device)
        if key not in cls._tables:
            e = cls.plan(N, 2, device)
            if name == 'filtr':
                t = e.table(_lib.T_FILTR)
            elif name == 'sharp':
                # pyqg's exponential filter with filterfac = 1e20 (advect '2/3-rule', operators.py:253): a sharp
                # cut at 0.65 pi; same expression as model.py::_initialize_filter
                wvx = np.sqrt(e.table(_lib.T_WV2)) * (cls.L / N)
                with np.errstate(under='ignore'):
                    t = np.exp(-1e20 * (wvx - 0.65 * np.pi) ** 4.)
                t[wvx <= 0.65 * np.pi] = 1.
            else:
                _, ratio = name
                t = np.exp(-e.table(_lib.T_WV2) * (ratio * cls.L / N) ** 2 / 24)
            cls._tables[key] = torch.as_tensor(t).to(e.device).contiguous()
        return cls._tables[key]

    @staticmethod
    def _even(x):
        """pad the field count to an even number (fields are transformed as packed pairs)"""
        if x.shape[0] % 2 == 0:
            return x, x.shape[0]
        return torch.cat([x, torch.zeros_like(x[:1])]), x.shape[0]

    # ---- building blocks --------------------------------------------------------------------
    @classmethod
    def rfft2(cls, x):
        x, M = cls._even(x.contiguous())
        N = x.shape[-1]
        out = torch.empty((x.shape[0], N, N // 2 + 1), dtype=torch.complex128, device=x.device)
        check(lib.qgx_rfft2(cls.plan(N, x.shape[0], x.device.index or 0)._h, _ptr(x), _ptr(out), _stream()))
        return out[:M]

    @classmethod
    def irfft2(cls, xh):
        xh, M = cls._even(xh.contiguous())
        N = xh.shape[-2]
        out = torch.empty((xh.shape[0], N, N), dtype=torch.float64, device=xh.device)
        check(lib.qgx_irfft2(cls.plan(N, xh.shape[0], xh.device.index or 0)._h, _ptr(xh), _ptr(out), _stream()))
        return out[:M]

    @staticmethod
    def regrid(xh, N, scale=1.0, zero_src_2h=False, zero_dst_2h=False, filt=None):
        xh = xh.contiguous()
        M, n = xh.shape[0], xh.shape[-2]
        out = torch.empty((M, N, N // 2 + 1), dtype=torch.complex128, device=xh.device)
        check(lib.qgx_spec_regrid(_ptr(xh), _ptr(out), M, n, N, float(scale), int(zero_src_2h),
                                  int(zero_dst_2h), _ptr(filt), _stream()))
        return out

    @classmethod
    def spec_div(cls, ah, bh):
        ref = ah if ah is not None else bh
        out = torch.empty_like(ref)
        check(lib.qgx_spec_div(_ptr(ah), _ptr(bh), _ptr(out), ref.shape[0], ref.shape[-2], cls.L, _stream()))
        return out

    @staticmethod
    def mul(a, b, alpha=1.0, c=None, beta=0.0):
        """alpha * a * b + beta * c"""
        a = a.contiguous()
        out = torch.empty_like(a)
        check(lib.qgx_real_fma(_ptr(a), _ptr(b.contiguous()) if b is not None else None, _ptr(out),
                               a.numel(), float(alpha), _ptr(c.contiguous()) if c is not None else None,
                               float(beta), _stream()))
        return out

    # ---- the reference's operators ------------------------------------------------------------
    @classmethod
    def cut_off(cls, X, nc):
        if nc % 2:
            raise ValueError('nc must be even')
        N = X.shape[-1]
        return cls.irfft2(cls.regrid(cls.rfft2(X), nc, scale=1.0 / (N / nc) ** 2,
                                     zero_dst_2h=FILTER_2h_HARMONICS))

    @classmethod
    def spectral_filter(cls, X, name):
        N = X.shape[-1]
        return cls.irfft2(cls.regrid(cls.rfft2(X), N, filt=cls.table(N, name, X.device.index or 0)))

    @classmethod
    def gauss_filter(cls, X, nc):
        return cls.spectral_filter(X, ('gauss', X.shape[-1] / nc))

    @classmethod
    def model_filter(cls, X, nc=None):
        return cls.spectral_filter(X, 'filtr')

    @classmethod
    def clean_2h(cls, X):
        N = X.shape[-1]
        return cls.irfft2(cls.regrid(cls.rfft2(X), N, zero_dst_2h=True))

    @classmethod
    def fft_interpolate(cls, x, n, N, truncate_2h=True):
        if x.shape[-1] != n or x.shape[-2] != n:
            raise ValueError('Input variable must be n*n points')
        if n % 2 or N % 2:
            raise ValueError('Grid sizes (n,N) must be even')
        return cls.irfft2(cls.regrid(cls.rfft2(x), N, scale=(N / n) ** 2, zero_src_2h=truncate_2h,
                                     zero_dst_2h=truncate_2h))

    @classmethod
    def Operator1(cls, X, nc):
        # cut_off then the model's exponential filter: one transform pair, filter fused in the regrid
        N = X.shape[-1]
        return cls.irfft2(cls.regrid(cls.rfft2(X), nc, scale=1.0 / (N / nc) ** 2, zero_dst_2h=True,
                                     filt=cls.table(nc, 'filtr', X.device.index or 0)))

    @classmethod
    def Operator2(cls, X, nc):
        N = X.shape[-1]
        return cls.irfft2(cls.regrid(cls.rfft2(X), nc, scale=1.0 / (N / nc) ** 2, zero_dst_2h=True,
                                     filt=cls.table(nc, ('gauss', 2.0), X.device.index or 0)))

    @classmethod
    def Operator4(cls, X, nc):
        return cls.model_filter(cls.Operator2(X, nc))

    @classmethod
    def Operator5(cls, X, nc):
        return cls.cut_off(X, nc)

    @classmethod
    def divergence(cls, fx, fy):
        return cls.irfft2(cls.spec_div(cls.rfft2(fx), cls.rfft2(fy)))

    @classmethod
    def advect(cls, var, u, v, dealias='none'):
        if dealias == 'none':
            return cls.divergence(cls.mul(var, u), cls.mul(var, v))
        if dealias == '3/2-rule':
            n = u.shape[-1]
            N = int((n * 3) // 2)
            a, b, c = (cls.fft_interpolate(t, n, N) for t in (var, u, v))
            return cls.divergence(cls.fft_interpolate(cls.mul(a, b), N, n),
                                  cls.fft_interpolate(cls.mul(a, c), N, n))
        if dealias == '2/3-rule':
            a, b, c = (cls.spectral_filter(t, 'sharp') for t in (var, u, v))
            return cls.spectral_filter(cls.divergence(cls.mul(a, b), cls.mul(a, c)), 'sharp')
        raise ValueError('dealias should be none or 2/3-rule or 3/2-rule')

    @classmethod
    def velocities(cls, q, pyqg_params, return_engine=False):
        """(u, v) of PV fields q (B,2,N,N) by the model's inversion (apply_operator_to_model,
        operators.py:229-234 builds a fresh pyqg model per call; here one cached engine per grid)."""
        B, _, N, _ = q.shape
        e = cls.inversion_model(N, B, q.device.index or 0, pyqg_params)
        e.set_q(q)
        e.invert()
        u, v = e.get(_lib.F_U), e.get(_lib.F_V)
        return (u, v, e) if return_engine else (u, v)

    # ---- the same diagnostic kept in spectral space ------------------------------------------------
    # The reference composes PV_subgrid_forcing from numpy-level operators, each of which transforms to and from grid
    # space: per high-resolution field and operator 14 transforms of the 256 / 384 grids, 9 of them round trips that
    # cancel.  Below the high-resolution tendency is computed ONCE per snapshot for all operators and stays spectral
    # between the steps: 5 large transforms per field (3/2-rule) + the inversion.  Same operations in exact arithmetic;
    # the results agree with the composed form to rounding (tests/test_gpu_operators.py).
    _HAT_OPERATORS = ('Operator1', 'Operator2', 'Operator4', 'Operator5', 'cut_off')

    @classmethod
    def _operator_hat(cls, operator, Xh, nc):
        """operator applied to the spectrum Xh (M,N,N/2+1) -> spectrum on the nc grid, or None (no spectral form)"""
        N = Xh.shape[-2]
        dev = Xh.device.index or 0
        sc = 1.0 / (N / nc) ** 2
        name = getattr(operator, '__name__', '')
        if name == 'Operator1':
            return cls.regrid(Xh, nc, scale=sc, zero_dst_2h=True, filt=cls.table(nc, 'filtr', dev))
        if name == 'Operator2':
            return cls.regrid(Xh, nc, scale=sc, zero_dst_2h=True, filt=cls.table(nc, ('gauss', 2.0), dev))
        if name == 'Operator4':
            y = cls.regrid(Xh, nc, scale=sc, zero_dst_2h=True, filt=cls.table(nc, ('gauss', 2.0), dev))
            return cls.regrid(y, nc, filt=cls.table(nc, 'filtr', dev))
        if name in ('Operator5', 'cut_off'):
            if nc % 2:
                raise ValueError('nc must be even')
            return cls.regrid(Xh, nc, scale=sc, zero_dst_2h=FILTER_2h_HARMONICS)
        return None

    @classmethod
    def advect_hat(cls, qh, uh, vh, dealias='none'):
        """spectrum of div(u q, v q) from the spectra of q, u, v (each (M,n,n/2+1)); the arithmetic of advect()"""
        n = qh.shape[-2]
        if dealias == 'none':
            a, b, c = (cls.irfft2(t) for t in (qh, uh, vh))
            return cls.spec_div(cls.rfft2(cls.mul(a, b)), cls.rfft2(cls.mul(a, c)))
        if dealias == '3/2-rule':
            N = int((n * 3) // 2)
            up = lambda t: cls.irfft2(cls.regrid(t, N, scale=(N / n) ** 2, zero_src_2h=True, zero_dst_2h=True))
            down = lambda t: cls.regrid(cls.rfft2(t), n, scale=(n / N) ** 2, zero_src_2h=True, zero_dst_2h=True)
            a, b, c = up(qh), up(uh), up(vh)
            return cls.spec_div(down(cls.mul(a, b)), down(cls.mul(a, c)))
        if dealias == '2/3-rule':
            sharp = cls.table(n, 'sharp', qh.device.index or 0)
            a, b, c = (cls.irfft2(cls.regrid(t, n, filt=sharp)) for t in (qh, uh, vh))
            return cls.regrid(cls.spec_div(cls.rfft2(cls.mul(a, b)), cls.rfft2(cls.mul(a, c))), n, filt=sharp)
        raise ValueError('dealias should be none or 2/3-rule or 3/2-rule')

    @classmethod
    def hires_tendency_hat(cls, q, pyqg_params, dealias='none'):
        """(spectrum of q, spectrum of its advective tendency div(u q, v q)) of PV fields q (B,2,N,N): the part of the
        subgrid-forcing diagnostic that depends neither on the operator nor on the coarse resolution"""
        B, _, N, _ = q.shape
        e = cls.inversion_model(N, B, q.device.index or 0, pyqg_params)
        e.set_q(q)
        e.invert()
        flat = lambda t: t.reshape(-1, t.shape[-2], t.shape[-1])
        qh, ph = flat(e.get(_lib.F_QH)), flat(e.get(_lib.F_PH))
        vh = cls.spec_div(ph, None)                               # i k psi
        uh = cls.regrid(cls.spec_div(None, ph), N, scale=-1.0)    # -i l psi
        return qh, cls.advect_hat(qh, uh, vh, dealias)

    @classmethod
    def subgrid_forcing_from_hat(cls, qh, adv_hat, nc, operator, pyqg_params, dealias='none', return_psi=False):
        """the operator- and resolution-dependent rest: coarse PV, its velocities and tendency, the filtered
        high-resolution tendency.  qh, adv_hat: (2B,N,N/2+1) from hires_tendency_hat"""
        B = qh.shape[0] // 2
        flat = lambda t: t.reshape(-1, t.shape[-2], t.shape[-1])
        qf = cls.irfft2(cls._operator_hat(operator, qh, nc)).reshape(B, 2, nc, nc)
        uf, vf, coarse = cls.velocities(qf, pyqg_params, return_engine=True)
        psi = coarse.get(_lib.F_P) if return_psi else None        # before the engine is reused / evicted
        adv_c = cls.advect(flat(qf), flat(uf), flat(vf), dealias)
        adv_f = cls.irfft2(cls._operator_hat(operator, adv_hat, nc))
        forcing = cls.mul(adv_c, None, 1.0, adv_f, -1.0).reshape(B, 2, nc, nc)
        return (forcing, qf, uf, vf, psi) if return_psi else (forcing, qf, uf, vf)

    @classmethod
    def has_spectral_form(cls, operator):
        return getattr(operator, '__name__', '') in cls._HAT_OPERATORS

    @classmethod
    def PV_subgrid_forcing_multi(cls, q, nc, operators, pyqg_params, dealias='none', return_psi=False):
        """PV_subgrid_forcing for several operators at once: the inversion and the advection of the high-resolution
        fields are shared.  -> list of (forcing, qf, uf, vf [, psi_f]) in the order of `operators`, or None when one
        of them has no spectral form (callers then use the composed PV_subgrid_forcing)."""
        if not all(cls.has_spectral_form(op) for op in operators):
            return None
        qh, adv_hat = cls.hires_tendency_hat(q, pyqg_params, dealias)
        return [cls.subgrid_forcing_from_hat(qh, adv_hat, nc, op, pyqg_params, dealias, return_psi) for op in operators]

    @classmethod
    def PV_subgrid_forcing(cls, q, nc, operator, pyqg_params, dealias='none', return_psi=False, composed=False):
        """q: (B,2,N,N) device tensor.  -> (forcing, qf, uf, vf [, psi_f]) on the nc grid, each (B,2,nc,nc).
        composed=True: the reference's sequence of grid-space operators (operators.py:283-287) step by step."""
        if not composed:
            fast = cls.PV_subgrid_forcing_multi(q, nc, [operator], pyqg_params, dealias, return_psi)
            if fast is not None:
                return fast[0]
        B, _, N, _ = q.shape
        flat = lambda t: t.reshape(-1, t.shape[-2], t.shape[-1])
        u, v = cls.velocities(q, pyqg_params)
        qf = operator(flat(q), nc).reshape(B, 2, nc, nc)
        uf, vf, coarse = cls.velocities(qf, pyqg_params, return_engine=True)
        psi = coarse.get(_lib.F_P) if return_psi else None      # before the engine is reused / evicted
        adv_c = cls.advect(flat(qf), flat(uf), flat(vf), dealias)
        adv_f = operator(cls.advect(flat(q), flat(u), flat(v), dealias), nc)
        forcing = cls.mul(adv_c, None, 1.0, adv_f, -1.0).reshape(B, 2, nc, nc)
        return (forcing, qf, uf, vf, psi) if return_psi else (forcing, qf, uf, vf)


# ---- numpy-facing functions with the reference's names ------------------------------------------
def _np_wrap(fn):
    def wrapper(X, *args, **kw):
        X = np.asarray(X, dtype='float64')
        lead = X.shape[:-2]
        t = torch.as_tensor(np.ascontiguousarray(X.reshape((-1,) + X.shape[-2:]))).cuda()
        out = fn(t, *args, **kw).cpu().numpy()
        return out.reshape(lead + out.shape[-2:])
    wrapper.__name__ = fn.__name__
    return wrapper


cut_off = _np_wrap(Dev.cut_off)
gauss_filter = _np_wrap(Dev.gauss_filter)
model_filter = _np_wrap(Dev.model_filter)
clean_2h = _np_wrap(Dev.clean_2h)
fft_interpolate = _np_wrap(Dev.fft_interpolate)
Operator1 = _np_wrap(Dev.Operator1)
Operator2 = _np_wrap(Dev.Operator2)
Operator4 = _np_wrap(Dev.Operator4)
Operator5 = _np_wrap(Dev.Operator5)
_DEV_OF = {Operator1: Dev.Operator1, Operator2: Dev.Operator2, Operator4: Dev.Operator4,
           Operator5: Dev.Operator5, cut_off: Dev.cut_off}


def _dev(a):
    a = np.asarray(a, dtype='float64')
    return torch.as_tensor(np.ascontiguousarray(a.reshape((-1,) + a.shape[-2:]))).cuda()


def divergence(fx, fy):
    return Dev.divergence(_dev(fx), _dev(fy)).cpu().numpy().reshape(np.shape(fx))


def advect(var, u, v, dealias='none'):
    return Dev.advect(_dev(var), _dev(u), _dev(v), dealias).cpu().numpy().reshape(np.shape(var))


def ave_lev(arr, delta):
    """depth average with weights [delta/(1+delta), 1/(1+delta)] over the 'lev' axis (operators.py:12-29)"""
    w = np.array([delta / (1 + delta), 1 / (1 + delta)])
    if hasattr(arr, 'dims') and 'lev' in arr.dims:
        ax = arr.dims.index('lev')
        vals = np.moveaxis(np.asarray(arr.values), ax, 0)
        return np.tensordot(w, vals, axes=(0, 0))
    return arr


def apply_operator_to_model(q, nc, operator, pyqg_params):
    """-> a QGModel on the operator's grid holding the coarse-grained PV, inverted (operators.py:219-236)."""
    qf = operator(np.asarray(q, dtype='float64'), nc)
    params = dict(pyqg_params)
    params.update(nx=qf.shape[-1], log_level=0)
    params.pop('parameterization', None)
    m = QGModel(**params)
    m.q = qf
    m._invert()
    return m


def PV_subgrid_flux(q, nc, operator, pyqg_params):
    """Subgrid PV fluxes (operators.py:269-281): (uq_flux, vq_flux) = filtered-model product minus filtered product."""
    m = apply_operator_to_model(q, 1, lambda x, n: x, pyqg_params)
    mf = apply_operator_to_model(q, nc, operator, pyqg_params)
    return mf.u * mf.q - operator(m.u * m.q, nc), mf.v * mf.q - operator(m.v * m.q, nc)


def PV_subgrid_forcing(q, nc, operator, pyqg_params, dealias='none'):
    """q: (nlev,Ny,Nx) numpy.  -> (forcing (nlev,nc,nc), coarse model, fine model)."""
    dev_op = _DEV_OF.get(operator)
    if dev_op is None:
        raise NotImplementedError('operator has no device implementation')
    forcing, _, _, _ = Dev.PV_subgrid_forcing(_dev(q)[None], nc, dev_op, pyqg_params, dealias)
    m = apply_operator_to_model(q, 1, lambda x, n: x, pyqg_params)
    mf = apply_operator_to_model(q, nc, operator, pyqg_params)
    return forcing[0].cpu().numpy(), mf, m
