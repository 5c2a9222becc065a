"""Online stepper: builds the model, sets the initial condition, drives the snapshot loop
(reference: pyqg_generative/tools/simulate.py:16-60 drop_vars / concat_in_time, :109-145
run_simulation, :147-168 set_initial_condition).

Ensemble extension: ``n_members`` members advance together on one GPU; datasets gain a
leading 'run' dimension (the reference's name for the member axis, simulate.py:284) when
n_members > 1.
"""
import numpy as np

from ..qgmodel import QGModel
from .parameters import ANDREW_1000_STEPS
from .stochastic_pyqg import stochastic_QGModel


def dataset_backend():
    try:
        import xarray as xr
        return xr
    except ImportError:
        from . import xr_lite
        return xr_lite


def set_initial_condition(m, seeds=None):
    """Band-limited random PV in the upper layer (power confined to the scales a 32x32 model
    resolves), zero in the lower layer; then m._invert().

    seeds: None -> numpy's global stream, exactly the reference's draw order (rand(ny,nx) then
    rand(1,nx), per member); else one integer seed per member (seed = member id).
    """
    B = getattr(m, 'n_members', 1)
    q = np.zeros((B, 2, m.ny, m.nx))
    for b in range(B):
        rng = np.random if seeds is None else np.random.RandomState(int(seeds[b]))
        q2d = 1e-7 * rng.rand(m.ny, m.nx)
        q2d -= q2d.mean(axis=(-2, -1), keepdims=True)
        q2d *= np.sqrt(m.nx * m.ny / 64 ** 2)
        q1d = 1e-6 * (np.ones((m.ny, 1)) * rng.rand(1, m.nx))
        q1d -= q1d.mean(axis=(-2, -1), keepdims=True)
        q1d *= np.sqrt(m.nx / 64)
        Xf = np.fft.rfftn(q1d + q2d)
        q[b, 0] = np.fft.irfftn(Xf * (m.wv < np.pi / (m.L / 32)))
    m.q = q[0] if B == 1 else q
    m._invert()


def snapshot_dataset(m):
    """The part of pyqg's Model.to_dataset() that survives drop_vars: q, u, v, psi as float32
    (time, [run,] lev, y, x) with pyqg's coordinates; time in days."""
    xr = dataset_backend()
    B = m.n_members
    lead = ('time',) if B == 1 else ('time', 'run')
    dims = lead + ('lev', 'y', 'x')
    f32 = lambda a: np.asarray(a, dtype='float32')[None]
    data = {'q': (dims, f32(m.q)), 'u': (dims, f32(m.u)), 'v': (dims, f32(m.v)), 'psi': (dims, f32(m.p))}
    coords = {'time': ('time', np.array([m.t / 86400.], dtype='float32')),
              'lev': ('lev', np.arange(1, 3)),
              'x': ('x', m.x[0, :].astype('float32')), 'y': ('y', m.y[:, 0].astype('float32'))}
    if B > 1:
        coords['run'] = ('run', np.arange(m.member_offset, m.member_offset + B))
    if xr.__name__.endswith('xr_lite'):
        ds = xr.Dataset(data, coords={k: xr.DataArray(v[1], [v[0]]) for k, v in coords.items()})
    else:
        ds = xr.Dataset({k: (v[0], v[1]) for k, v in data.items()}, coords=coords)
    # time-averaged spectral diagnostics, present once averaging has started (t >= tavestart)
    if m.diagnostics_count > 0:
        run = () if B == 1 else ('run',)
        for name in m.diagnostic_names:
            a = np.asarray(m.get_diagnostic(name), dtype='float32')
            dd = run + (('lev', 'l', 'k') if a.ndim - len(run) == 3 else ('l', 'k'))
            ds[name] = xr.DataArray(a, dd) if xr.__name__.endswith('xr_lite') else (dd, a)
    ds['time'].attrs['units'] = 'days'
    ds.attrs.update({'pyqg:nx': m.nx, 'pyqg:dt': m.dt, 'pyqg:rek': m.rek, 'pyqg:delta': m.delta,
                     'pyqg:beta': m.beta, 'pyqg:L': m.L, 'pyqg:rd': m.rd})
    return ds


def concat_in_time(datasets):
    xr = dataset_backend()
    return xr.concat(datasets, dim='time')


def run_simulation(pyqg_params, parameterization=None, q_init=None, sampling_freq=ANDREW_1000_STEPS,
                   n_members=1, seeds=None, device=0, seed=0, member_offset=0):
    """pyqg_params: dict of model parameters; parameterization: None or
    dict(self=<Parameterization>, sampling='AR1'|'constant'|'deterministic', nsteps=int);
    q_init: optional PV (nlev,ny,nx) or (B,nlev,ny,nx).  Returns a Dataset of snapshots taken
    every ``sampling_freq`` seconds of model time."""
    params = dict(pyqg_params)
    params['tmax'] = float(params['tmax'])
    eng = dict(n_members=n_members, device=device, seed=seed, member_offset=member_offset)
    if parameterization is None:
        m = QGModel(**params, **eng)
    else:
        params['parameterization'] = parameterization['self']
        m = stochastic_QGModel(params, parameterization['sampling'], parameterization['nsteps'], **eng)
    if q_init is not None:
        m.q = np.asarray(q_init, dtype='float64')
        m._invert()
        parts = [snapshot_dataset(m)]            # convenient to have the IC saved
    else:
        set_initial_condition(m, seeds)
        parts = []
    for _ in m.run_with_snapshots(tsnapint=sampling_freq):
        parts.append(snapshot_dataset(m))
    ds = concat_in_time(parts)
    ds.attrs['pyqg_params'] = str(pyqg_params)
    m.close()
    return ds


def generate_subgrid_forcing(Nc, pyqg_params, sampling_freq=ANDREW_1000_STEPS, n_members=1, seeds=None,
                             device=0, operators=('Operator2', 'Operator5')):
    """Forcing-dataset generation (reference: simulate.py:62-106): run the high-resolution model given
    by pyqg_params and, every ``sampling_freq`` seconds, coarse-grain the PV to each resolution in Nc
    with each operator and diagnose the subgrid forcing with 3/2-rule dealiasing.  Returns
    {'<Operator>-<nc>-dealias': Dataset(q_forcing_advection, q, u, v, psi  float32 (time,[run,]lev,y,x))}.
    The hires members, the coarse-graining and the forcing diagnostic all stay on the GPU."""
    import torch
    from .operators import Dev
    from .. import _lib
    xr = dataset_backend()
    params = dict(pyqg_params)
    params['tmax'] = float(params['tmax'])
    m = QGModel(**params, n_members=n_members, device=device)
    set_initial_condition(m, seeds)
    coarse_params = {k: v for k, v in params.items() if k in ('rek', 'delta', 'beta', 'rd', 'U1', 'U2', 'H1', 'L')}
    B = n_members
    lead = ('time',) if B == 1 else ('time', 'run')
    dims = lead + ('lev', 'y', 'x')
    out = {}
    for _ in m.run_with_snapshots(tsnapint=sampling_freq):
        qd = m.q_device()
        for opname in operators:
            dev_op = getattr(Dev, opname)
            for nc in Nc:
                forcing, qf, uf, vf = Dev.PV_subgrid_forcing(qd, nc, dev_op, coarse_params, '3/2-rule')
                plan = Dev._plans[next(k for k in Dev._plans if k[0] == 'inv' and k[1] == nc and k[2] == B)]
                ph = plan.get(_lib.F_PH).reshape(-1, nc, nc // 2 + 1)
                psi = Dev.irfft2(ph).reshape(B, 2, nc, nc)
                pack = lambda t: (t[0] if B == 1 else t).to(torch.float32).cpu().numpy()[None]
                data = {'q_forcing_advection': (dims, pack(forcing)), 'q': (dims, pack(qf)),
                        'u': (dims, pack(uf)), 'v': (dims, pack(vf)), 'psi': (dims, pack(psi))}
                xc = ((np.arange(nc) + 0.5) / nc * m.L).astype('float32')
                coords = {'time': np.array([m.t / 86400.], dtype='float32'), 'lev': np.arange(1, 3), 'x': xc, 'y': xc}
                if xr.__name__.endswith('xr_lite'):
                    ds = xr.Dataset(data, coords={k: xr.DataArray(v, [k]) for k, v in coords.items()})
                else:
                    ds = xr.Dataset(data, coords={k: (k, v) for k, v in coords.items()})
                ds['time'].attrs['units'] = 'days'
                out.setdefault(f'{opname}-{nc}-dealias', []).append(ds)
    for key in out:
        out[key] = xr.concat(out[key], 'time').assign_attrs({'pyqg_params': str(pyqg_params)})
    m.close()
    return out


def run_forecast(pyqg_params, parameterization, q_init, n_ens, sampling_freq=86400, device=0, seed=0):
    """Forecast mode (reference: simulate.py:254-293): n_ens members start from the SAME coarse-grained
    initial PV and differ only in the latent noise; the reference runs them one after another and averages
    with xarray, here they advance together.  Returns a Dataset holding q,u,v,psi of member 0 and the
    ensemble means q_mean,u_mean,v_mean,psi_mean (time,lev,y,x)."""
    xr = dataset_backend()
    ds = run_simulation(pyqg_params, parameterization, q_init=q_init, sampling_freq=sampling_freq,
                        n_members=n_ens, device=device, seed=seed)
    out = xr.Dataset(coords={k: ds[k] for k in ('time', 'lev', 'x', 'y')}) if not xr.__name__.endswith('xr_lite') \
        else xr.Dataset(coords={k: ds[k] for k in ('time', 'lev', 'x', 'y')})
    for var in ('q', 'u', 'v', 'psi'):
        a = np.asarray(ds[var].values)
        if n_ens == 1:
            a = a[:, None]
        dims = ('time', 'lev', 'y', 'x')
        first, mean = a[:, 0], a.mean(axis=1)
        if xr.__name__.endswith('xr_lite'):
            out[var] = xr.DataArray(first, dims)
            out[var + '_mean'] = xr.DataArray(mean, dims)
        else:
            out[var] = (dims, first)
            out[var + '_mean'] = (dims, mean)
    out.attrs.update(ds.attrs)
    return out
