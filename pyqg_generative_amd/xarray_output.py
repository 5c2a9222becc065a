"""``QGModel.to_dataset()``: the model state and its time-averaged diagnostics as an xarray Dataset in
pyqg's layout (pyqg 0.7.2 ``xarray_output.model_to_dataset``; consumed by the reference at
pyqg_generative/tools/simulate.py:16-60,93,105,133,138, tools/comparison_tools.py:102-103 and
tools/spectral_tools.py:60; layout evidence: Google-Colab/dataset.ipynb cells 8 and 14).

* state variables carry a leading ``time`` axis of length one: ``q, u, v, ufull, vfull, p, dqdt``
  ``(time, lev, y, x)`` float64; ``qh, uh, vh, ph, dqhdt`` ``(time, lev, l, k)`` complex128;
  ``Ubg, Qy`` ``(lev,)``;
* time-averaged spectral diagnostics ``(time, lev, l, k)`` / ``(time, l, k)``, present only once
  averaging has started (the reference takes them from the last snapshot with ``.isel(time=-1)``,
  simulate.py:54-56);
* coordinates ``time`` [s], ``lev``, ``lev_mid``, ``x``, ``y``, ``l``, ``k``; global attributes
  ``pyqg:<name>`` + ``title`` + ``reference``.

Ensemble extension: with ``n_members > 1`` every variable gains a leading ``run`` dimension (the
reference's name for the member axis: ``xr.concat(ds, 'run')``, simulate.py:282), i.e. exactly the
layout of the reference's stored multi-run datasets ``(run, time, lev, y, x)`` / ``(run, lev, l, k)``.

This module is host-only (numpy in, Dataset out) so that the layout is testable without a GPU.
"""
import numpy as np

SPATIAL = ('time', 'lev', 'y', 'x')
SPECTRAL = ('time', 'lev', 'l', 'k')

# name -> (dims, units, long_name)
VARIABLES = {
    'q': (SPATIAL, 's^-1', 'potential vorticity in real space'),
    'u': (SPATIAL, 'm s^-1', 'zonal velocity anomaly'),
    'v': (SPATIAL, 'm s^-1', 'meridional velocity anomaly'),
    'ufull': (SPATIAL, 'm s^-1', 'zonal full velocities in real space'),
    'vfull': (SPATIAL, 'm s^-1', 'meridional full velocities in real space'),
    'qh': (SPECTRAL, 's^-1', 'potential vorticity in spectral space'),
    'uh': (SPECTRAL, 'm s^-1', 'zonal velocity anomaly in spectral space'),
    'vh': (SPECTRAL, 'm s^-1', 'meridional velocity anomaly in spectral space'),
    'ph': (SPECTRAL, 'm^2 s^-1', 'streamfunction in spectral space'),
    'Ubg': (('lev',), 'm s^-1', 'background zonal velocity'),
    'Qy': (('lev',), 'm^-1 s^-1', 'background potential vorticity gradient'),
    'p': (SPATIAL, 'm^2 s^-1', 'streamfunction in real space'),
    'dqhdt': (SPECTRAL, 's^-2', 'previous partial derivative of potential vorticity wrt. time in spectral space'),
    'dqdt': (SPATIAL, 's^-2', 'previous partial derivative of potential vorticity wrt. time in real space'),
}

COORDS = {
    'time': ('s', 'model time'),
    'lev': ('', 'vertical levels'),
    'lev_mid': ('', 'vertical level interface'),
    'x': ('m', 'real space grid points in the x direction'),
    'y': ('m', 'real space grid points in the y direction'),
    'l': ('m^-1', 'spectal space grid points in the l direction'),
    'k': ('m^-1', 'spectal space grid points in the k direction'),
}

# diagnostics: name -> (dims after time, units, long_name)
DIAGNOSTICS = {
    'KEspec': (('lev', 'l', 'k'), 'm^2 s^-2', 'kinetic energy spectrum'),
    'Ensspec': (('lev', 'l', 'k'), 's^-2', 'enstrophy spectrum'),
    'entspec': (('l', 'k'), '', 'barotropic enstrophy spectrum'),
    'APEflux': (('l', 'k'), 'm^2 s^-3', 'spectral flux of available potential energy'),
    'KEflux': (('l', 'k'), 'm^2 s^-3', 'spectral flux of kinetic energy'),
    'APEgenspec': (('l', 'k'), 'm^2 s^-3', 'the spectrum of the rate of generation of available potential energy'),
    'KEfrictionspec': (('l', 'k'), 'm^2 s^-3', 'total energy dissipation spectrum by bottom drag'),
    'paramspec': (('l', 'k'), 'm^2 s^-3', 'spectral contribution of subgrid parameterization to energy (if present)'),
    'paramspec_APEflux': (('l', 'k'), 'm^2 s^-3', 'total additional APE flux due to subgrid parameterization'),
    'paramspec_KEflux': (('l', 'k'), 'm^2 s^-3', 'total additional KE flux due to subgrid parameterization'),
    'Dissspec': (('l', 'k'), 'm^2 s^-3', 'Spectral contribution of filter dissipation to total energy'),
    'ENSDissspec': (('l', 'k'), 's^-3', 'Spectral contribution of filter dissipation to barotropic enstrophy'),
    'ENSflux': (('l', 'k'), 's^-3', 'barotropic enstrophy flux'),
    'ENSgenspec': (('l', 'k'), 's^-3', 'the spectrum of the rate of generation of barotropic enstrophy'),
    'ENSfrictionspec': (('l', 'k'), 's^-3', 'the spectrum of the rate of dissipation of barotropic enstrophy due to bottom friction'),
    'ENSparamspec': (('l', 'k'), 's^-3', 'Spectral contribution of subgrid parameterization to enstrophy'),
}

# model attributes exported as global attributes "pyqg:<name>"
ATTRIBUTES = ('beta', 'delta', 'del2', 'dt', 'filterfac', 'L', 'M', 'nk', 'nl', 'ntd', 'nx', 'ny', 'nz',
              'rd', 'rek', 'taveint', 'tavestart', 'tc', 'tmax', 'twrite', 'W')


def _backend():
    try:
        import xarray as xr
        return xr
    except ImportError:
        from .tools import xr_lite
        return xr_lite


def model_to_dataset(m, fields=None, diagnostics=None, xr=None):
    """m: a model object with pyqg's grid attributes (x, y, k, l, t, nz, Ubg, Qy and those named in
    ATTRIBUTES; ``n_members`` and ``member_offset`` optional).  fields: dict name -> host array for the
    names of VARIABLES (default: read ``getattr(m, name)``); arrays have pyqg's shapes, with a leading
    member axis when ``m.n_members > 1``.  diagnostics: dict name -> time-mean array (default
    ``m.get_diagnostic`` for every name once ``m.diagnostics_count > 0``)."""
    xr = xr or _backend()
    B = int(getattr(m, 'n_members', 1))
    run = ('run',) if B > 1 else ()
    if fields is None:
        fields = {name: getattr(m, name) for name in VARIABLES if hasattr(m, name)}
    if diagnostics is None:
        diagnostics = {}
        if getattr(m, 'diagnostics_count', 0) > 0:
            diagnostics = {name: m.get_diagnostic(name) for name in DIAGNOSTICS if name in m.diagnostic_names}

    def with_time(a, dims):
        """member axis first, then the length-one time axis (the stored layout of the reference's runs)"""
        a = np.asarray(a)
        if 'time' not in dims:
            return dims, a
        tail = len(dims) - 1
        if a.ndim == tail + 1 and B > 1:
            return run + dims, a[:, None]
        return dims, a[None]

    variables = {}
    for name, arr in fields.items():
        dims, units, long_name = VARIABLES[name]
        d, a = with_time(arr, dims)
        variables[name] = (d, a, {'units': units, 'long_name': long_name})
    for name, arr in diagnostics.items():
        dims, units, long_name = DIAGNOSTICS[name]
        d, a = with_time(arr, ('time',) + dims)
        variables[name] = (d, a, {'units': units, 'long_name': long_name})

    cvals = {'time': np.array([float(m.t)]), 'lev': np.arange(1, m.nz + 1), 'lev_mid': np.arange(1.5, m.nz + .5),
             'x': np.asarray(m.x)[0, :], 'y': np.asarray(m.y)[:, 0], 'l': np.asarray(m.l)[:, 0], 'k': np.asarray(m.k)[0, :]}
    coords = {name: ((name,), cvals[name], {'units': u, 'long_name': ln}) for name, (u, ln) in COORDS.items()}
    if B > 1:
        off = int(getattr(m, 'member_offset', 0))
        coords['run'] = (('run',), np.arange(off, off + B), {'long_name': 'ensemble member'})

    attrs = {f'pyqg:{a}': getattr(m, a) for a in ATTRIBUTES if hasattr(m, a)}
    attrs['title'] = 'pyqg: Python Quasigeostrophic Model'
    attrs['reference'] = 'https://pyqg.readthedocs.io/en/latest/index.html'
    return xr.Dataset(variables, coords=coords, attrs=attrs)
