"""Labelled-array stand-in used ONLY when xarray is not installed (this build image).

It implements the slice of the xarray API that the reference's snapshot flow and online metrics
touch (pyqg_generative/tools/simulate.py:16-60, tools/comparison_tools.py:116-195,
tools/spectral_tools.py:7-101) with xarray's own semantics, so that the product code is written
once against ``xr.Dataset`` / ``xr.DataArray`` / ``xr.concat`` and runs on either backend:

* ``Dataset(data_vars, coords, attrs)`` with ``(dims, data[, attrs])`` tuples; ``ds.variables``
  holds coordinates AND data variables, while ``keys()`` / iteration / ``data_vars`` are the data
  variables only (xarray >= 0.11);
* ``ds[name]``, ``ds[[names]]``, ``ds.name``, ``ds[name] = DataArray | (dims, data) | ndarray``;
* ``isel``, ``mean``, ``sum``, ``expand_dims``, ``astype``, ``drop_vars``, ``rename``, ``copy``,
  ``assign_attrs``, ``dims``, arithmetic between arrays of identical dims (or with numpy operands
  broadcast against the trailing axes), positional ``__getitem__``;
* ``concat(objs, dim)`` along an existing or a new dimension; variables without ``dim`` are
  broadcast along it, as xarray's default ``data_vars='all'`` does;
* ``to_netcdf`` through scipy (classic format, real variables only).

With xarray importable ``tools.simulate.dataset_backend()`` returns the real package and this
module is unused.
"""
import numpy as np


def _is_int(i):
    return isinstance(i, (int, np.integer))


class DataArray:
    def __init__(self, data, coords=None, dims=None, attrs=None, name=None):
        if isinstance(data, DataArray):
            dims = dims or data.dims
            attrs = attrs if attrs is not None else data.attrs
            data = data.values
        self.values = np.asarray(data)
        if dims is None:
            dims = tuple(f'dim_{i}' for i in range(self.values.ndim))
        self.dims = (dims,) if isinstance(dims, str) else tuple(dims)
        if len(self.dims) != self.values.ndim:
            raise ValueError(f'dims {self.dims} do not match array of shape {self.values.shape}')
        self.attrs = dict(attrs or {})
        self.name = name
        self.coords = {}
        if isinstance(coords, dict):
            self.coords = {k: (v if isinstance(v, DataArray) else DataArray(v, dims=[k])) for k, v in coords.items()}
        elif coords is not None:                       # list aligned with dims
            for d, c in zip(self.dims, coords):
                self.coords[d] = c if isinstance(c, DataArray) else DataArray(c, dims=[d])

    dtype = property(lambda self: self.values.dtype)
    shape = property(lambda self: self.values.shape)
    size = property(lambda self: self.values.size)
    ndim = property(lambda self: self.values.ndim)
    data = property(lambda self: self.values)

    def _like(self, values, dims=None):
        dims = self.dims if dims is None else tuple(dims)
        co = {k: v for k, v in self.coords.items() if k in dims and v.shape == (values.shape[dims.index(k)],)}
        return DataArray(values, coords=co, dims=dims, attrs=self.attrs, name=self.name)

    def astype(self, dt):
        return self._like(self.values.astype(dt))

    def copy(self, deep=True):
        return self._like(self.values.copy() if deep else self.values)

    def isel(self, indexers=None, **idx):
        idx = dict(indexers or {}, **idx)
        missing = set(idx) - set(self.dims)
        if missing:
            raise ValueError(f'Dimensions {missing} do not exist. Expected one or more of {self.dims}')
        sl, dims = [], []
        for d in self.dims:
            i = idx.get(d, slice(None))
            sl.append(i)
            if not _is_int(i):
                dims.append(d)
        out = DataArray(self.values[tuple(sl)], dims=dims, attrs=self.attrs, name=self.name)
        for k, c in self.coords.items():
            if k in dims and k in idx:
                out.coords[k] = DataArray(c.values[idx[k]], dims=[k], attrs=c.attrs)
            elif k in dims:
                out.coords[k] = c
        return out

    def _reduce(self, fn, dim):
        dims = self.dims if dim is None else ((dim,) if isinstance(dim, str) else tuple(dim))
        ax = tuple(self.dims.index(d) for d in dims)
        return self._like(fn(self.values, axis=ax), [d for d in self.dims if d not in dims])

    def mean(self, dim=None):
        return self._reduce(np.mean, dim)

    def sum(self, dim=None):
        return self._reduce(np.sum, dim)

    def std(self, dim=None):
        return self._reduce(np.std, dim)

    def expand_dims(self, dim):
        return self._like(self.values[None], (dim,) + self.dims)

    def transpose(self, *dims):
        return self._like(self.values.transpose([self.dims.index(d) for d in dims]), dims)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        key = key if isinstance(key, tuple) else (key,)
        key = key + (slice(None),) * (self.ndim - len(key))
        return self.isel({d: k for d, k in zip(self.dims, key)})

    def __len__(self):
        return self.shape[0]

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def __float__(self):
        return float(self.values)

    def _binary(self, other, op, reflected=False):
        if isinstance(other, DataArray):
            if other.dims == self.dims:
                o = other.values
            elif set(other.dims) <= set(self.dims):          # broadcast by name
                o = other.values.reshape([other.shape[other.dims.index(d)] if d in other.dims else 1
                                          for d in self.dims])
            elif set(self.dims) <= set(other.dims):
                return other._binary(self, op, not reflected)
            else:
                raise ValueError(f'cannot align dims {self.dims} and {other.dims}')
        else:
            o = np.asarray(other)
        return self._like(op(o, self.values) if reflected else op(self.values, o))

    __add__ = lambda s, o: s._binary(o, np.add)
    __radd__ = lambda s, o: s._binary(o, np.add, True)
    __sub__ = lambda s, o: s._binary(o, np.subtract)
    __rsub__ = lambda s, o: s._binary(o, np.subtract, True)
    __mul__ = lambda s, o: s._binary(o, np.multiply)
    __rmul__ = lambda s, o: s._binary(o, np.multiply, True)
    __truediv__ = lambda s, o: s._binary(o, np.divide)
    __rtruediv__ = lambda s, o: s._binary(o, np.divide, True)
    __pow__ = lambda s, o: s._binary(o, np.power)
    __neg__ = lambda s: s._like(-s.values)

    def __repr__(self):
        return f'<xr_lite.DataArray {self.name or ""} {dict(zip(self.dims, self.shape))} {self.dtype}>'


def _as_variable(v, name=None, like=None):
    if isinstance(v, DataArray):
        return v
    if isinstance(v, tuple):
        return DataArray(v[1], dims=v[0], attrs=v[2] if len(v) > 2 else None, name=name)
    v = np.asarray(v)
    if like is not None and like.ndim == v.ndim:
        return DataArray(v, dims=like.dims, attrs=like.attrs, name=name)
    if v.ndim == 1 and name is not None:
        return DataArray(v, dims=[name], name=name)
    if v.ndim == 0:
        return DataArray(v, dims=(), name=name)
    raise ValueError(f'cannot infer dims of {name!r}')


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self._vars = {}          # data variables
        self._coords = {}
        self.attrs = dict(attrs or {})
        for k, v in (coords or {}).items():
            self._coords[k] = _as_variable(v, k)
        for k, v in (data_vars or {}).items():
            self[k] = v

    # ---- mapping surface (data variables only, like xarray >= 0.11) ----
    def keys(self):
        return self._vars.keys()

    def __iter__(self):
        return iter(self._vars)

    def __len__(self):
        return len(self._vars)

    def items(self):
        return self._vars.items()

    def __contains__(self, k):
        return k in self._vars or k in self._coords

    data_vars = property(lambda self: self._vars)
    coords = property(lambda self: self._coords)

    @property
    def variables(self):
        """coordinates and data variables (a fresh dict: assignments during iteration are safe)"""
        return {**self._coords, **self._vars}

    @property
    def dims(self):
        sizes = {}
        for a in self.variables.values():
            sizes.update(zip(a.dims, a.shape))
        return sizes

    sizes = dims

    def _with_coords(self, a):
        out = a._like(a.values)
        for d in a.dims:
            if d in self._coords and self._coords[d].shape == (a.shape[a.dims.index(d)],):
                out.coords[d] = self._coords[d]
        return out

    def __getitem__(self, k):
        if isinstance(k, (list, tuple)):
            out = Dataset(attrs=self.attrs)
            out._coords = dict(self._coords)
            out._vars = {n: self._vars[n] for n in k}
            return out
        if k in self._vars:
            a = self._with_coords(self._vars[k])
            a.name = k
            return a
        return self._coords[k]

    def __setitem__(self, k, v):
        old = self._vars.get(k, self._coords.get(k))
        v = _as_variable(v, k, like=old)
        v.name = k
        for ck, cv in v.coords.items():          # a DataArray brings its coordinates along
            if ck != k and ck not in self._coords:
                self._coords[ck] = cv
        if k in self._coords or (k not in self._vars and v.dims == (k,)):
            self._coords[k] = v
        else:
            self._vars[k] = v

    def __getattr__(self, k):
        if k.startswith('_'):
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    # ---- transformations ----
    def _map(self, fn, coords_too=True):
        out = Dataset(attrs=self.attrs)
        out._coords = {k: (fn(k, v) if coords_too else v) for k, v in self._coords.items()}
        out._vars = {k: fn(k, v) for k, v in self._vars.items()}
        return out

    def copy(self, deep=False):
        return self._map(lambda k, v: v.copy(deep=deep))

    def drop_vars(self, names):
        names = [names] if isinstance(names, str) else list(names)
        out = self.copy()
        for n in names:
            if n not in out._vars and n not in out._coords:
                raise ValueError(f'variable {n!r} not found')
            out._vars.pop(n, None)
            out._coords.pop(n, None)
        return out

    def rename(self, mapping):
        out = self.copy()
        for a, b in mapping.items():
            tgt = out._vars if a in out._vars else out._coords
            tgt[b] = tgt.pop(a)
        return out

    def assign_attrs(self, *a, **kw):
        out = self.copy()
        for d in a:
            out.attrs.update(d)
        out.attrs.update(kw)
        return out

    def astype(self, dt):
        return self._map(lambda k, v: v.astype(dt), coords_too=False)

    def isel(self, indexers=None, **idx):
        idx = dict(indexers or {}, **idx)
        missing = set(idx) - set(self.dims)
        if missing:
            raise ValueError(f'Dimensions {missing} do not exist. Expected one or more of {tuple(self.dims)}')
        out = self._map(lambda k, v: v.isel({d: i for d, i in idx.items() if d in v.dims}))
        return out

    def mean(self, dim=None):
        dims = None if dim is None else ((dim,) if isinstance(dim, str) else tuple(dim))
        out = Dataset(attrs=self.attrs)
        out._coords = {k: v for k, v in self._coords.items() if dims is not None and not set(v.dims) & set(dims)}
        out._vars = {k: v.mean(None if dims is None else [d for d in dims if d in v.dims]) for k, v in self._vars.items()}
        return out

    def expand_dims(self, dim):
        out = self.copy()
        out._vars = {k: v.expand_dims(dim) for k, v in self._vars.items()}
        return out

    def to_netcdf(self, path):
        from scipy.io import netcdf_file
        with netcdf_file(path, 'w', version=2) as f:
            for d, n in self.dims.items():
                f.createDimension(d, n)
            for k, a in self.variables.items():
                if np.iscomplexobj(a.values) or a.values.dtype == object:
                    continue
                vals = a.values.astype('int32') if a.values.dtype == np.int64 else a.values
                v = f.createVariable(k, vals.dtype.char, a.dims)
                v[...] = vals
                for ak, av in a.attrs.items():
                    setattr(v, ak, av)
            for ak, av in self.attrs.items():
                setattr(f, ak, av if isinstance(av, (int, float, str)) else str(av))

    def __repr__(self):
        return f'<xr_lite.Dataset dims={self.dims} data_vars={list(self._vars)}>'


def _concat_arrays(arrs, dim, n_each):
    first = arrs[0]
    if dim in first.dims:
        ax = first.dims.index(dim)
        return first._like(np.concatenate([a.values for a in arrs], axis=ax), first.dims)
    # broadcast along the new / missing dimension (xarray's data_vars='all')
    vals = np.concatenate([np.broadcast_to(a.values[None], (n,) + a.shape) for a, n in zip(arrs, n_each)], axis=0)
    return first._like(vals, (dim,) + first.dims)


def concat(objs, dim):
    objs = list(objs)
    if isinstance(objs[0], DataArray):
        return _concat_arrays(objs, dim, [1] * len(objs))
    first = objs[0]
    n_each = [d.dims.get(dim, 1) for d in objs]
    out = Dataset(attrs=dict(first.attrs))
    for k, c in first._coords.items():
        if dim in c.dims:
            out._coords[k] = _concat_arrays([d._coords[k] for d in objs], dim, n_each)
        else:
            out._coords[k] = c
    for k in first._vars:
        out._vars[k] = _concat_arrays([d._vars[k] for d in objs], dim, n_each)
    return out
