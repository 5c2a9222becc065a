"""Minimal labelled-array container used ONLY when xarray is not installed (this build
image): enough of the xarray.Dataset surface for run_simulation's snapshot flow
(reference: pyqg_generative/tools/simulate.py:16-60) and its tests.  With xarray
available ``dataset_backend()`` returns the real package and this file is unused.
"""
import numpy as np


class DataArray:
    def __init__(self, data, dims=None, coords=None, attrs=None, name=None):
        self.values = np.asarray(data)
        self.dims = tuple(dims) if dims is not None else tuple(f'dim_{i}' for i in range(self.values.ndim))
        self.attrs = dict(attrs or {})
        self.name = name

    dtype = property(lambda self: self.values.dtype)
    shape = property(lambda self: self.values.shape)

    def astype(self, dt):
        return DataArray(self.values.astype(dt), self.dims, attrs=self.attrs)

    def isel(self, **idx):
        sl, dims = [], []
        for d in self.dims:
            if d in idx:
                sl.append(idx[d])
                if not np.isscalar(idx[d]) and not isinstance(idx[d], (int, np.integer)):
                    dims.append(d)
            else:
                sl.append(slice(None))
                dims.append(d)
        return DataArray(self.values[tuple(sl)], dims, attrs=self.attrs)

    def mean(self, dim):
        ax = self.dims.index(dim)
        return DataArray(self.values.mean(axis=ax), [d for d in self.dims if d != dim], attrs=self.attrs)

    def __array__(self, dtype=None):
        return self.values if dtype is None else self.values.astype(dtype)


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.variables = {}
        self.coords = {}
        self.attrs = dict(attrs or {})
        for k, v in (coords or {}).items():
            self.coords[k] = v if isinstance(v, DataArray) else DataArray(np.asarray(v), [k])
        for k, v in (data_vars or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, tuple):
            v = DataArray(v[1], v[0], attrs=v[2] if len(v) > 2 else None)
        elif not isinstance(v, DataArray):
            v = DataArray(np.asarray(v), self[k].dims if k in self else None)
        if k in self.coords:
            self.coords[k] = v
        else:
            self.variables[k] = v

    def __getitem__(self, k):
        return self.variables[k] if k in self.variables else self.coords[k]

    def __contains__(self, k):
        return k in self.variables or k in self.coords

    def keys(self):
        return self.variables.keys()

    def __getattr__(self, k):
        try:
            return self.__getitem__(k)
        except KeyError:
            raise AttributeError(k)

    def drop_vars(self, names):
        names = [names] if isinstance(names, str) else list(names)
        out = self.copy()
        for n in names:
            out.variables.pop(n, None)
        return out

    def rename(self, mapping):
        out = self.copy()
        for a, b in mapping.items():
            out.variables[b] = out.variables.pop(a)
        return out

    def copy(self, deep=False):
        cp = (lambda a: DataArray(a.values.copy(), a.dims, attrs=dict(a.attrs))) if deep else (lambda a: a)
        out = Dataset(attrs=dict(self.attrs))
        out.variables = {k: cp(v) for k, v in self.variables.items()}
        out.coords = {k: cp(v) for k, v in self.coords.items()}
        return out

    def assign_attrs(self, *a, **kw):
        out = self.copy()
        for d in a:
            out.attrs.update(d)
        out.attrs.update(kw)
        return out

    def astype(self, dt):
        out = self.copy()
        out.variables = {k: v.astype(dt) for k, v in self.variables.items()}
        return out

    def to_netcdf(self, path):
        from scipy.io import netcdf_file
        with netcdf_file(path, 'w', version=2) as f:
            sizes = {}
            for a in list(self.variables.values()) + list(self.coords.values()):
                for d, n in zip(a.dims, a.shape):
                    sizes[d] = n
            for d, n in sizes.items():
                f.createDimension(d, n)
            for k, a in {**self.coords, **self.variables}.items():
                if np.iscomplexobj(a.values):
                    continue
                v = f.createVariable(k, a.values.dtype.char, a.dims)
                v[...] = a.values
                for ak, av in a.attrs.items():
                    setattr(v, ak, av)
            for ak, av in self.attrs.items():
                setattr(f, ak, str(av))


def concat(datasets, dim):
    first = datasets[0]
    out = Dataset(attrs=dict(first.attrs))
    for k, c in first.coords.items():
        if dim in c.dims:
            out.coords[k] = DataArray(np.concatenate([d.coords[k].values for d in datasets],
                                                     axis=c.dims.index(dim)), c.dims, attrs=c.attrs)
        else:
            out.coords[k] = c
    for k, a in first.variables.items():
        if dim in a.dims:
            out.variables[k] = DataArray(np.concatenate([d.variables[k].values for d in datasets],
                                                        axis=a.dims.index(dim)), a.dims, attrs=a.attrs)
        else:
            out.variables[k] = datasets[-1].variables[k]
    return out
