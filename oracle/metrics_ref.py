"""CPU restatement (numpy / scipy) of the reference's ONLINE and OFFLINE METRICS: the distributional and spectral errors of a
low-resolution run against a coarse-grained high-resolution reference.

TEST INFRASTRUCTURE — see oracle/__init__.py.  The reference publishes these numbers for its shipped models
(Google-Colab/online-simulations.ipynb cells 29-33); tests/test_gpu_online_metrics.py reproduces them from runs of the
HIP engine.  Reference followed (paths relative to /root/reference/pyqg_generative):
  tools/comparison_tools.py:16-54     DISTRIB_KEYS / SPECTRAL_KEYS, distrib_score, spectral_score
  tools/comparison_tools.py:56-115    coarsegrain_reference_dataset (spectra cut to the coarse wavenumbers and multiplied
                                      by the squared filter transfer function)
  tools/comparison_tools.py:116-195   diagnostic_differences_Perezhogin
  tools/spectral_tools.py:103-180     calc_ispec (oracle/spectral_ref.py, golden-pinned)
  tools/computational_tools.py:38-84  subgrid_scores (offline metrics; Google-Colab/offline-analysis.ipynb cells 28-31)
and, for the derived features 'add(pow(u,2),pow(v,2))' and 'pow(curl(u,v),2)', the FeatureExtractor of the un-vendored
dependency pyqg_parameterization_benchmarks (utils.py): curl(u, v) = ddx(v) - ddy(u) with spectral derivatives on
pyqg's grid of the fields' resolution.

Inputs are plain arrays (no xarray): a run is a dict with
  q, u, v            (T, 2, N, N)   snapshots (the last `T_last` are used)
  KEspec             (2, N, N/2+1)  time-averaged spectra on pyqg's (l, k) half plane
  KEflux, APEflux, APEgenspec [, paramspec_KEflux, paramspec_APEflux]   (N, N/2+1)
"""
import numpy as np
from scipy.stats import wasserstein_distance

from .qg_ref import QGModelRef
from .spectral_ref import calc_ispec

DISTRIB_KEYS = [f'distrib_diff_{v}{z}' for v in ('q', 'u', 'v', 'KE', 'Ens') for z in (1, 2)]
SPECTRAL_KEYS = ['spectral_diff_KEspec1', 'spectral_diff_KEspec2', 'spectral_diff_KEflux', 'spectral_diff_APEflux',
                 'spectral_diff_APEgenspec', 'spectral_diff_KEfrictionspec', 'spectral_diff_Eflux']


def _curl(u, v):
    """ddx(v) - ddy(u), spectral, for fields (..., N, N) on the L = 1e6 m periodic domain"""
    g = QGModelRef(nx=u.shape[-1])
    return np.fft.irfftn(np.fft.rfftn(v, axes=(-2, -1)) * g.ik - np.fft.rfftn(u, axes=(-2, -1)) * g.il, axes=(-2, -1))


def _features(run, z, T_last):
    q, u, v = (np.asarray(run[k], dtype='float64')[-T_last:, z] for k in ('q', 'u', 'v'))
    return {'q': q, 'u': u, 'v': v, 'KE': u ** 2 + v ** 2, 'Ens': _curl(u, v) ** 2}


def twothirds_nyquist(g):
    return g.k[0][np.argwhere(np.array(g.filtr)[0] < 1)[0, 0]]


def spectral_rmse(spec1, spec2):
    g1, g2 = QGModelRef(nx=spec1.shape[-2]), QGModelRef(nx=spec2.shape[-2])
    kr1, i1 = calc_ispec(g1, spec1)
    kr2, i2 = calc_ispec(g2, spec2)
    nk = int((kr1 < min(twothirds_nyquist(g1), twothirds_nyquist(g2))).sum())
    return np.sqrt(np.mean((i1[:nk] - i2[:nk]) ** 2)), np.sqrt(np.mean(i2[:nk] ** 2))


def diagnostic_differences(run, target, T_last=128):
    """-> dict of normalised differences (difference / scale), keys as in the reference"""
    diff, scale = {}, {}
    for z in (0, 1):
        f1, f2 = _features(run, z, T_last), _features(target, z, T_last)
        for label in ('q', 'u', 'v', 'KE', 'Ens'):
            a, b = f1[label].ravel(), f2[label].ravel()
            diff[f'distrib_diff_{label}{z + 1}'] = wasserstein_distance(a, b)
            scale[f'distrib_diff_{label}{z + 1}'] = float(np.sqrt(np.mean(b ** 2)))
    for z in (0, 1):
        diff[f'spectral_diff_KEspec{z + 1}'], scale[f'spectral_diff_KEspec{z + 1}'] = \
            spectral_rmse(run['KEspec'][z], target['KEspec'][z])

    def eflux(r):
        return sum(np.asarray(r[k]) for k in ('KEflux', 'APEflux', 'paramspec_KEflux', 'paramspec_APEflux') if k in r)
    diff['spectral_diff_Eflux'], scale['spectral_diff_Eflux'] = spectral_rmse(eflux(run), eflux(target))
    diff['spectral_diff_APEgenspec'], scale['spectral_diff_APEgenspec'] = spectral_rmse(run['APEgenspec'], target['APEgenspec'])
    return {k: diff[k] / scale[k] for k in diff}


def distrib_score(d):
    return float(np.mean([v for k, v in d.items() if k in DISTRIB_KEYS]))


def spectral_score(d):
    return float(np.mean([v for k, v in d.items() if k in SPECTRAL_KEYS]))


def coarsegrain_reference_spectra(hires, resolution, operator='Operator1'):
    """spectral statistics of a high-resolution run -> the coarse grid's wavenumbers x (filter transfer function)^2
    (comparison_tools.py:90-114).  hires: dict name -> (2, N, N/2+1) or (N, N/2+1)."""
    n = resolution // 2
    g = QGModelRef(nx=resolution)
    if operator == 'Operator1':
        tf2 = g.filtr ** 2
    elif operator == 'Operator2':
        tf2 = np.exp(-(g.k ** 2 + g.l ** 2) * (2 * g.dx) ** 2 / 24) ** 2
    else:
        raise ValueError('operator must be Operator1 or Operator2')
    out = {}
    for name, a in hires.items():
        a = np.asarray(a)
        out[name] = np.concatenate((a[..., :n, :n + 1], a[..., -n:, :n + 1]), axis=-2) * tf2
    return out


# ---- offline metrics ---------------------------------------------------------------------------------------------
def power_spectrum_iso(x):
    """spectrum(type='power', averaging=False, truncate=False, time=slice(None)) of tools/spectral_tools.py:7-101 for
    x (run, time, lev, N, N): run/time-mean power per lev, binned by calc_ispec.  -> (kr, (lev, nbins))"""
    x = np.asarray(x, dtype='float64')
    N = x.shape[-1]
    af2 = (np.abs(np.fft.rfftn(x, axes=(-2, -1)) / (N * N)) ** 2).mean(axis=(0, 1))
    g = QGModelRef(nx=N)
    out = [calc_ispec(g, af2[z], averaging=False, truncate=False) for z in range(af2.shape[0])]
    return out[0][0], np.stack([o[1] for o in out])


def subgrid_scores(true, mean, gen):
    """tools/computational_tools.py:38-84 on arrays (run, time, lev, N, N): relative L2 errors of the conditional mean,
    of the power spectrum of one generated sample and of the spectrum of its residual, and the residual variance ratio.
    -> dict(L2_mean, L2_total, L2_residual, var_ratio (lev,))"""
    true, mean, gen = (np.asarray(a, dtype='float64') for a in (true, mean, gen))

    def L2(x, xt, axes):
        return float(((((x - xt) ** 2).mean(axes) / (xt ** 2).mean(axes)) ** 0.5).mean())     # per lev, then mean
    field_axes = (0, 1, 3, 4)
    out = {'L2_mean': L2(mean, true, field_axes)}
    _, sp_true = power_spectrum_iso(true)
    _, sp_gen = power_spectrum_iso(gen)
    out['L2_total'] = L2(sp_gen, sp_true, (1,))
    _, sp_true_res = power_spectrum_iso(true - mean)
    _, sp_gen_res = power_spectrum_iso(gen - mean)
    out['L2_residual'] = L2(sp_gen_res, sp_true_res, (1,))
    out['var_ratio'] = ((gen - mean) ** 2).mean(field_axes) / ((true - mean) ** 2).mean(field_axes)
    return out
