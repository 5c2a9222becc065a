"""CPU restatement (numpy) of the isotropic-spectrum binning that defines the
KE-spectrum parity metric.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Pinned by tests/golden/ispec.npz.
Reference followed: pyqg_generative/tools/spectral_tools.py:103-180 (calc_ispec);
metric use: Google-Colab/online-simulations.ipynb cell 25,
``calc_ispec(m, 0.5*ave_lev(KEspec, delta))`` with tools/operators.py:12-29 (ave_lev).
"""
import numpy as np


def calc_ispec(grid, dens2d, averaging=True, truncate=True, nd_wavenumber=False, nfactor=1):
    """grid: object with ll, kk, dk, dl, wv.  dens2d: (..., nl, nk) spectral density.
    Returns (kr, phr) with phr shaped (..., nbins)."""
    d = np.array(dens2d, dtype='float64', copy=True)
    d[..., 0] /= 2
    d[..., -1] /= 2
    lmax = np.abs(grid.ll).max()
    kmax_ = np.abs(grid.kk).max()
    kmax = min(lmax, kmax_) if truncate else np.hypot(lmax, kmax_)
    kmin = min(grid.dk, grid.dl)
    dkr = np.sqrt(grid.dk ** 2 + grid.dl ** 2) * nfactor
    edges = np.arange(kmin, kmax - dkr, dkr)
    phr = np.zeros(d.shape[:-2] + (edges.size,))
    for i, lo in enumerate(edges):
        if averaging:
            ring = (grid.wv >= lo) & (grid.wv <= lo + dkr)
            if ring.sum() == 0:
                val = 0.
            else:
                val = d[..., ring].mean(axis=-1) * (lo + dkr / 2) * np.pi / (grid.dk * grid.dl)
        else:
            ring = (grid.wv >= lo) & (grid.wv < lo + dkr)
            val = d[..., ring].sum(axis=-1) / dkr
        phr[..., i] = 2 * val
    kr = edges + dkr / 2
    if nd_wavenumber:
        kr = kr / kmin
        phr = phr * kmin
    return kr, phr


def ave_lev(arr, delta):
    """Depth average with weights [delta/(1+delta), 1/(1+delta)] over axis 0."""
    w = np.array([delta / (1 + delta), 1 / (1 + delta)])
    return np.tensordot(w, arr, axes=(0, 0))


def ke_spectrum(grid, KEspec, delta):
    """Isotropic depth-averaged KE spectrum as plotted by the reference."""
    return calc_ispec(grid, 0.5 * ave_lev(KEspec, delta))
