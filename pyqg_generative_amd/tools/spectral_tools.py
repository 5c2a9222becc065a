"""Isotropic spectra of model output (reference: pyqg_generative/tools/spectral_tools.py:7-101
``spectrum``, :103-180 ``calc_ispec``).  ``calc_ispec`` is the binning behind the reference's
KE-spectrum metric: ``calc_ispec(m, 0.5 * ave_lev(KEspec, delta))`` (Google-Colab/online-simulations.ipynb
cell 25) and ``spectral_rmse`` (tools/comparison_tools.py:149-160).

Host-side numpy: the inputs are (nl, nk) time-averaged spectral densities that already left the GPU
with the dataset; one call bins a few thousand numbers.  The ring membership of every wavenumber is
computed once per (grid, options) and cached, and any number of stacked densities (..., nl, nk) are
reduced in one pass.
"""
import numpy as np

from .parameters import AVERAGE_SLICE_ANDREW

_RINGS = {}


def _rings(model, averaging, truncate, nfactor):
    """-> (left bin edges, bin width, list of flat index arrays into the (nl,nk) plane)"""
    wv = np.asarray(model.wv)
    key = (wv.shape, float(model.dk), float(model.dl), bool(averaging), bool(truncate), float(nfactor))
    hit = _RINGS.get(key)
    if hit is not None:
        return hit
    ll_max, kk_max = np.abs(model.ll).max(), np.abs(model.kk).max()
    kmax = min(ll_max, kk_max) if truncate else np.sqrt(ll_max ** 2 + kk_max ** 2)
    kmin = min(model.dk, model.dl)
    dkr = np.sqrt(model.dk ** 2 + model.dl ** 2) * nfactor
    left = np.arange(kmin, kmax - dkr, dkr)
    flat = wv.ravel()
    if averaging:       # closed rings: a wavenumber on a shared edge counts in both
        members = [np.flatnonzero((flat >= lo) & (flat <= lo + dkr)) for lo in left]
    else:               # half-open rings: a partition, so that Parseval's identity holds
        members = [np.flatnonzero((flat >= lo) & (flat < lo + dkr)) for lo in left]
    _RINGS[key] = (left, dkr, members)
    return _RINGS[key]


def calc_ispec(model, _var_dens, averaging=True, truncate=True, nd_wavenumber=False, nfactor=1):
    """Isotropic spectrum of a 2-D spectral density ``|fft|^2 / M^2`` on the model's (l, k) half-plane.

    model: anything with pyqg's ``wv, ll, kk, dk, dl``.  _var_dens: (..., nl, nk); leading axes are
    carried through.  averaging: mean over each ring times its circumference (a density estimate),
    else sum / bin width (Parseval holds: ``var = phr.sum() * dkr``).  truncate: stop at the inscribed
    circle of the wavenumber square, else the circumscribed one.  Returns (kr bin centres, phr)."""
    dens = np.array(_var_dens, dtype='float64', copy=True)
    dens[..., 0] /= 2          # the k = 0 and k = Nyquist columns have no conjugate twin in the half-plane
    dens[..., -1] /= 2
    left, dkr, members = _rings(model, averaging, truncate, nfactor)
    flat = dens.reshape(dens.shape[:-2] + (-1,))
    phr = np.zeros(dens.shape[:-2] + (left.size,))
    for i, idx in enumerate(members):
        if averaging:
            if idx.size:
                phr[..., i] = flat[..., idx].mean(axis=-1) * (left[i] + dkr / 2) * np.pi / (model.dk * model.dl)
        else:
            phr[..., i] = flat[..., idx].sum(axis=-1) / dkr
    phr *= 2
    kr = left + dkr / 2
    if nd_wavenumber:
        kmin = min(model.dk, model.dl)
        kr, phr = kr / kmin, phr * kmin
    return kr, phr


class _Grid:
    """pyqg's spectral grid of an nx x nx, L x L domain (what ``pyqg.QGModel(nx=..)`` is built for at
    spectral_tools.py:51 and comparison_tools.py:153-154) without creating a device model."""

    def __init__(self, nx, L=1e6):
        self.nx, self.L = int(nx), float(L)
        self.dk = self.dl = 2. * np.pi / L
        self.kk = self.dk * np.arange(0., nx / 2 + 1)
        self.ll = self.dl * np.append(np.arange(0., nx / 2), np.arange(-nx / 2, 0.))
        k, l = np.meshgrid(self.kk, self.ll)
        self.wv = np.sqrt(k ** 2 + l ** 2)


class spectrum:
    """Time/run-mean isotropic power / energy / co- / cross-layer spectra of real fields
    (run, time, lev, y, x); ``spectrum(type)(x [, y])`` -> DataArray (lev, k) or (k)."""

    def __init__(self, type='power', averaging=False, truncate=False, time=AVERAGE_SLICE_ANDREW):
        self.type, self.averaging, self.truncate, self.time = type, averaging, truncate, time

    def fft2d(self, _xarray):
        M = _xarray.shape[-1] * _xarray.shape[-2]
        x = np.asarray(_xarray.isel(time=self.time).values, dtype='float64')
        return np.fft.rfftn(x, axes=(-2, -1)) / M

    def isotropize(self, af2, *x, name, description, units):
        from .simulate import dataset_backend
        xr = dataset_backend()
        grid = _Grid(x[0].shape[-1])
        k, sp = calc_ispec(grid, af2, averaging=self.averaging, truncate=self.truncate)
        kc = xr.DataArray(k, dims=['k'], attrs={'long_name': 'isotropic wavenumber, $m^{-1}$'})
        attrs = {'long_name': name, 'description': description, 'units': units}
        if self.type == 'cross_layer':
            return xr.DataArray(sp, dims=['k'], coords={'k': kc}, attrs=attrs)
        return xr.DataArray(sp, dims=['lev', 'k'], coords={'lev': np.arange(1, sp.shape[0] + 1), 'k': kc}, attrs=attrs)

    def __call__(self, *_x, name='', description='', units=''):
        x = []
        for xx in _x:
            if 'run' not in xx.dims:
                xx = xx.expand_dims('run')
            if 'time' not in xx.dims:
                xx = xx.expand_dims('time')
                self.time = slice(0, 1)
            x.append(xx.transpose(*(['run', 'time'] + [d for d in xx.dims if d not in ('run', 'time')])))
        if self.type == 'power':
            af2 = np.abs(self.fft2d(x[0])) ** 2
        elif self.type == 'energy':
            af2 = np.abs(self.fft2d(x[0])) ** 2 / 2
        elif self.type == 'cospectrum':
            af2 = np.real(np.conj(self.fft2d(x[0])) * self.fft2d(x[1]))
        elif self.type == 'cross_layer':
            xf = self.fft2d(x[0])
            af2 = np.real(np.conj(xf[:, :, 0]) * xf[:, :, 1])
        else:
            raise ValueError(f'unknown spectrum type {self.type!r}')
        return self.isotropize(af2.mean(axis=(0, 1)), *x, name=name, description=description, units=units)
