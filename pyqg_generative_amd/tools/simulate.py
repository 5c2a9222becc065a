"""Online stepper: builds the model, sets the initial condition, drives the snapshot loop
(reference: pyqg_generative/tools/simulate.py:16-60 drop_vars / concat_in_time, :109-145
run_simulation, :147-168 set_initial_condition).

Ensemble extension: ``n_members`` members advance together on one GPU; datasets gain a
leading 'run' dimension (the reference's name for the member axis, simulate.py:284) when
n_members > 1.
"""
import numpy as np

from ..qgmodel import QGModel
from .parameters import ANDREW_1000_STEPS, DAY
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


SNAPSHOT_VARIABLES = ('q', 'u', 'v', 'p')       # the state fields that survive drop_vars


def drop_vars(ds):
    """Drop complex variables and the redundant real ones, convert float64 to float32, rename
    p -> psi, express time in days (reference: simulate.py:16-36)."""
    for key, var in ds.variables.items():
        if var.dtype == np.float64:
            ds[key] = var.astype(np.float32)
        elif var.dtype == np.complex128:
            ds = ds.drop_vars(key)
    for key in ('dqdt', 'ufull', 'vfull'):
        if key in ds.keys():
            ds = ds.drop_vars([key])
    if 'p' in ds.keys():
        ds = ds.rename({'p': 'psi'})
    if ds['time'].attrs.get('units') != 'days':
        ds['time'] = ds['time'].values / 86400
        ds['time'].attrs['units'] = 'days'
    return ds


def concat_in_time(datasets):
    """Snapshots -> one dataset (reference: simulate.py:39-60): state variables are concatenated along
    'time'; the spectral diagnostics are running time averages, so they are taken from the LAST snapshot
    (early snapshots, taken before tavestart, do not have them)."""
    xr = dataset_backend()
    last = datasets[-1]
    common = [k for k in datasets[0].keys() if all(k in d.keys() for d in datasets)]
    ds = xr.concat([d[common] for d in datasets], dim='time')
    for key in last.keys():
        if 'k' in last[key].dims:
            ds[key] = last[key].isel(time=-1)
    return drop_vars(ds)


def snapshot_dataset(m):
    """One snapshot as run_simulation stores it: ``drop_vars(m.to_dataset())`` (simulate.py:133,138),
    exporting only the fields drop_vars keeps."""
    return drop_vars(m.to_dataset(variables=SNAPSHOT_VARIABLES))


def run_simulation(pyqg_params, parameterization=None, q_init=None, sampling_freq=ANDREW_1000_STEPS,
                   n_members=1, seeds=None, device=0, seed=0, member_offset=0):
    """pyqg_params: dict of model parameters; parameterization: None or
    dict(self=<Parameterization>, sampling='AR1'|'constant'|'deterministic', nsteps=int);
    q_init: optional PV (nlev,ny,nx) or (B,nlev,ny,nx).  Returns a Dataset of snapshots taken
    every ``sampling_freq`` seconds of model time (reference: simulate.py:109-145)."""
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
                             device=0, operators=('Operator2', 'Operator5'), dealias='3/2-rule'):
    """Forcing-dataset generation (reference: simulate.py:62-106): run the high-resolution model given
    by pyqg_params and, every ``sampling_freq`` seconds, coarse-grain the PV to each resolution in Nc
    with each operator and diagnose the subgrid forcing with 3/2-rule dealiasing.  Returns
    {'<Operator>-<nc>-dealias': Dataset(q_forcing_advection, q, u, v, psi  float32 ([run,]time,lev,y,x))}.
    The hires members, the coarse-graining and the forcing diagnostic all stay on the GPU."""
    import torch
    from .operators import Dev
    xr = dataset_backend()
    params = dict(pyqg_params)
    params['tmax'] = float(params['tmax'])
    m = QGModel(**params, n_members=n_members, device=device)
    set_initial_condition(m, seeds)
    coarse_params = {k: v for k, v in params.items() if k in ('rek', 'delta', 'beta', 'rd', 'U1', 'U2', 'H1', 'L')}
    B = n_members
    dims = (('run',) if B > 1 else ()) + ('time', 'lev', 'y', 'x')
    pack = lambda t: (t[:, None] if B > 1 else t).to(torch.float32).cpu().numpy()      # length-one time axis
    out = {}
    for _ in m.run_with_snapshots(tsnapint=sampling_freq):
        qd = m.q_device()
        # the high-resolution inversion + advection does not depend on the operator or on nc: once per snapshot
        hat = Dev.hires_tendency_hat(qd, coarse_params, dealias) \
            if all(Dev.has_spectral_form(getattr(Dev, o)) for o in operators) else None
        for opname in operators:
            dev_op = getattr(Dev, opname)
            for nc in Nc:
                if hat is not None:
                    forcing, qf, uf, vf, psi = Dev.subgrid_forcing_from_hat(hat[0], hat[1], nc, dev_op, coarse_params,
                                                                            dealias, return_psi=True)
                else:
                    forcing, qf, uf, vf, psi = Dev.PV_subgrid_forcing(qd, nc, dev_op, coarse_params, dealias,
                                                                      return_psi=True)
                data = {'q_forcing_advection': (dims, pack(forcing)), 'q': (dims, pack(qf)),
                        'u': (dims, pack(uf)), 'v': (dims, pack(vf)), 'psi': (dims, pack(psi))}
                xc = ((np.arange(nc) + 0.5) / nc * m.L).astype('float32')
                coords = {'time': (('time',), np.array([m.t / 86400.], dtype='float32'), {'units': 'days'}),
                          'lev': (('lev',), np.arange(1, 3)), 'x': (('x',), xc), 'y': (('y',), xc)}
                if B > 1:
                    coords['run'] = (('run',), np.arange(m.member_offset, m.member_offset + B))
                out.setdefault(f'{opname}-{nc}' + ('-dealias' if dealias != 'none' else ''), []).append(
                    xr.Dataset(data, coords=coords))
    attrs = dict(m.to_dataset(variables=()).attrs)          # simulate.py:105: the hires model's pyqg:* attributes
    attrs['pyqg_params'] = str(pyqg_params)
    for key in out:
        out[key] = xr.concat(out[key], 'time').assign_attrs(attrs)
    m.close()
    Dev.close()
    return out


def forecast_statistics(ds, n_local, xr=None, group=None):
    """Member 0 and the ensemble mean over ALL members of the job of q, u, v, psi (reference:
    ``ds[var].isel(run=0)`` / ``ds[var].mean('run')``, simulate.py:284-290).  ``ds`` holds this rank's members along
    'run'.  Single process: a local mean.  Several ranks (torch.distributed initialised; members sharded in contiguous
    blocks, rank 0 first): ONE all-reduce of the per-rank partial sums of the four fields together
    (parallel.ensemble_mean — the path's only collective) and one broadcast of member 0 from rank 0, so that every rank
    returns the same dataset."""
    import torch
    import torch.distributed as dist
    from .. import parallel
    xr = xr or dataset_backend()
    names = ('q', 'u', 'v', 'psi')
    out = xr.Dataset(attrs=dict(ds.attrs))
    multi = dist.is_available() and dist.is_initialized()      # (a one-rank group takes the collective path as well)
    if not multi:
        for var in names:
            out[var] = ds[var].isel(run=0)
            out[var + '_mean'] = ds[var].mean('run')
        return out
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    run_axis = ds['q'].dims.index('run')
    local = torch.stack([torch.as_tensor(np.asarray(ds[v].values, dtype=np.float64)).sum(run_axis) for v in names]).to(dev)
    mean = parallel.ensemble_mean(local, n_local, group).cpu().numpy()
    first = torch.stack([torch.as_tensor(np.take(np.asarray(ds[v].values, dtype=np.float64), 0, axis=run_axis)) for v in names]).to(dev)
    dist.broadcast(first, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    first = first.cpu().numpy()
    dims = tuple(d for d in ds['q'].dims if d != 'run')
    for i, var in enumerate(names):
        out[var] = (dims, first[i].astype(ds[var].dtype))
        out[var + '_mean'] = (dims, mean[i].astype(ds[var].dtype))
    for c in dims:
        if c in ds.variables:
            out[c] = ds[c]
    return out


def run_forecast(pyqg_params, parameterization, q_init, n_ens, operator=None, sampling_freq=DAY, device=0, seed=0):
    """Forecast mode (reference: simulate.py:254-293): the initial PV is a snapshot of a high-resolution
    run, coarse-grained to the model's grid with ``operator`` ('Operator1' | 'Operator2' | 'Operator4' |
    'Operator5' or the function itself; None / failure to apply = ``q_init`` is used as it is, as the
    reference's try/except does, simulate.py:269-273); ``n_ens`` members start from it and differ only in
    the latent noise.  The reference runs them one after another and averages with xarray
    (simulate.py:279-290); here they advance together on the GPU — and, when torch.distributed is initialised
    (one process per GPU), each rank advances its contiguous block of the ``n_ens`` members (noise streams keyed by
    the global member id) and the mean is formed by one all-reduce (forecast_statistics).  Returns a Dataset
    holding q, u, v, psi of member 0 and the ensemble means q_mean, u_mean, v_mean, psi_mean, each (time, lev, y, x)."""
    import torch.distributed as dist
    from .. import parallel
    xr = dataset_backend()
    q_init = np.asarray(q_init, dtype='float64')
    nx = int(pyqg_params['nx'])
    if operator is not None and q_init.shape[-1] != nx:
        from . import operators as ops
        op = getattr(ops, operator) if isinstance(operator, str) else operator
        q_init = op(q_init, nx)
    first, n_local = 0, n_ens
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # checked on EVERY rank before anything is sharded: a rank that raised alone would leave the others in the all-reduce
        if n_ens < dist.get_world_size():
            raise ValueError(f'n_ens={n_ens} leaves some of the {dist.get_world_size()} ranks without a member')
        first, n_local = parallel.shard_members(n_ens, dist.get_rank(), dist.get_world_size())
    ds = run_simulation(pyqg_params, parameterization, q_init=q_init, sampling_freq=sampling_freq,
                        n_members=n_local, device=device, seed=seed, member_offset=first)[['q', 'u', 'v', 'psi']]
    if n_local == 1:
        ds = ds.expand_dims('run')
    return forecast_statistics(ds, n_local, xr)
