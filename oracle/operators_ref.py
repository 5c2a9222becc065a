"""CPU restatement (numpy float64) of the coarse-graining operators, the FFT
re-gridding and the 3/2-rule subgrid-forcing diagnostic.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Pinned by tests/golden/operators.npz
(outputs of the reference's own functions) and by the identities of
notebooks/3-2-dealiasing.ipynb cells 20-26, 31-32, 50-51.

Reference followed (pyqg_generative/tools/operators.py):
  :84-90   gauss_filter      X̂ · exp(-κ² (ratio·dx)² / 24)
  :92-99   model_filter      X̂ · filtr          (pyqg exponential filter)
  :117-132 cut_off           sharp spectral truncation to nc×nc, 2h harmonics zeroed
  :134-190 fft_interpolate   zero-pad / truncate between n×n and N×N grids
  :192-202 clean_2h
  :204-217 Operator1/2/4/5
  :219-236 apply_operator_to_model
  :241-247 divergence;  :249-268 advect;  :283-287 PV_subgrid_forcing
All functions take arrays whose last two axes are (y, x); leading axes are batch.
"""
import numpy as np
from .qg_ref import QGModelRef


def _grid(n, **kw):
    return QGModelRef(nx=n, **kw)


def _rfft2(x):
    return np.fft.rfftn(x, axes=(-2, -1))


def _irfft2(xh):
    return np.fft.irfftn(xh, axes=(-2, -1))


def cut_off(X, nc):
    if nc % 2:
        raise ValueError('nc must be even')
    N = X.shape[-1]
    h = nc // 2
    Xh = _rfft2(X)
    out = np.concatenate([Xh[..., :h, :h + 1], Xh[..., N - h:, :h + 1]], axis=-2) / (N / nc) ** 2
    out[..., h, 0] = 0
    out[..., :, h] = 0
    return _irfft2(out)


def gauss_filter(X, nc):
    N = X.shape[-1]
    g = _grid(N)
    width = (N / nc) * g.dx
    return _irfft2(_rfft2(X) * np.exp(-g.wv ** 2 * width ** 2 / 24))


def model_filter(X, nc=None):
    g = _grid(X.shape[-1])
    return _irfft2(_rfft2(X) * g.filtr)


def clean_2h(X):
    Xh = _rfft2(X)
    h = X.shape[-1] // 2
    Xh[..., h, 0] = 0
    Xh[..., :, h] = 0
    return _irfft2(Xh)


def fft_interpolate(x, n, N, truncate_2h=True):
    if x.shape[-1] != n or x.shape[-2] != n:
        raise ValueError('Input variable must be n*n points')
    if n % 2 or N % 2:
        raise ValueError('Grid sizes (n,N) must be even')
    h = min(n, N) // 2
    xh = _rfft2(x)
    if truncate_2h:
        xh[..., h, 0] = 0
    Xh = np.zeros(x.shape[:-2] + (N, N // 2 + 1), dtype=complex)
    Xh[..., :h, :h + 1] = xh[..., :h, :h + 1]
    Xh[..., N - h:, :h + 1] = xh[..., n - h:, :h + 1]
    if truncate_2h:
        Xh[..., h, 0] = 0
        Xh[..., :, h] = 0
    return _irfft2(Xh) * (N / n) ** 2


def Operator1(X, nc):
    return model_filter(cut_off(X, nc))


def Operator2(X, nc):
    return gauss_filter(cut_off(X, nc), nc // 2)


def Operator4(X, nc):
    return model_filter(Operator2(X, nc))


def Operator5(X, nc):
    return cut_off(X, nc)


def identity_operator(X, nc):
    return X


def apply_operator_to_model(q, nc, operator, pyqg_params):
    qf = operator(np.asarray(q, dtype='float64'), nc)
    params = dict(pyqg_params)
    params.update(nx=qf.shape[-1], log_level=0)
    m = QGModelRef(**params)
    m.set_q(qf)
    m._invert()
    return m


def divergence(fx, fy):
    g = _grid(fx.shape[-1])
    return _irfft2(_rfft2(fx) * g.ik) + _irfft2(_rfft2(fy) * g.il)


def advect(var, u, v, dealias='none'):
    if dealias == 'none':
        return divergence(var * u, var * v)
    if dealias == '2/3-rule':
        g = _grid(u.shape[-1], filterfac=1e+20)
        f = lambda a: _irfft2(_rfft2(a) * g.filtr)
        a, b, c = f(var), f(u), f(v)
        return f(divergence(a * b, a * c))
    if dealias == '3/2-rule':
        n = u.shape[-1]
        N = int((n * 3) // 2)
        a = fft_interpolate(var, n, N)
        b = fft_interpolate(u, n, N)
        c = fft_interpolate(v, n, N)
        return divergence(fft_interpolate(a * b, N, n), fft_interpolate(a * c, N, n))
    raise ValueError('dealias should be none or 2/3-rule or 3/2-rule')


def PV_subgrid_forcing(q, nc, operator, pyqg_params, dealias='none'):
    m = apply_operator_to_model(q, 1, identity_operator, pyqg_params)
    mf = apply_operator_to_model(q, nc, operator, pyqg_params)
    forcing = advect(mf.q, mf.u, mf.v, dealias) - operator(advect(m.q, m.u, m.v, dealias), nc)
    return forcing, mf, m
