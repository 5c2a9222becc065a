"""Run configurations of the reference (pyqg_generative/tools/parameters.py:3-41): resolution ->
time step table, eddy / jet parameter sets, calendar constants."""

DAY = 86400
YEAR = 360 * DAY
ANDREW_1000_STEPS = 3600000

_DT_BY_NX = {2048: 1800, 1024: 600, 512: 1800, 256: 3600, 128: 7200, 96: 7200}


class ConfigurationDict(dict):
    def _update(self, d):
        out = ConfigurationDict(self)
        out.update(d)
        return out

    def nx(self, _nx):
        out = ConfigurationDict(self)
        out['nx'] = _nx
        if _nx in _DT_BY_NX:
            out['dt'] = _DT_BY_NX[_nx]
        elif _nx <= 64:
            out['dt'] = 14400
        else:
            raise ValueError(f'no time step is tabulated for nx={_nx}')
        return out


EDDY_PARAMS = ConfigurationDict({'nx': 64, 'dt': 3600 * 4, 'tmax': 10 * YEAR, 'tavestart': 5 * YEAR})
JET_PARAMS = ConfigurationDict({'nx': 64, 'dt': 3600 * 4, 'tmax': 10 * YEAR, 'tavestart': 5 * YEAR,
                                'rek': 7e-08, 'delta': 0.1, 'beta': 1e-11})

SAMPLE_SLICE = slice(-40, None)
AVERAGE_SLICE = slice(360 * 5 * DAY, None)
AVERAGE_SLICE_ANDREW = slice(44, None)
