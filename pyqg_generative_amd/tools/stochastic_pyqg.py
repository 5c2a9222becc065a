"""Latent-noise time samplers and the stochastic model wrapper
(reference: pyqg_generative/tools/stochastic_pyqg.py:3-88).

In a fused on-device run the sampler object only carries (kind, nsteps): the update
z <- a z + b xi and the recompute/skip decision run inside qgx_step.  The host
``update`` methods keep the reference's stand-alone behaviour for code that drives a
parameterization by hand.
"""
from ..qgmodel import QGModel


class noise_time_sampler:
    kind = None

    def __init__(self, nsteps):
        self.nsteps = nsteps

    def update(self, generate_noise):
        raise NotImplementedError


class AR1_sampler(noise_time_sampler):
    kind = 'AR1'

    def update(self, generate_noise):
        if not hasattr(self, 'noise'):
            self.noise = generate_noise()
            return True
        n = self.nsteps
        a, b = (1 - 1 / n, (1 / n * (2 - 1 / n)) ** 0.5) if n > 0 else (1, 0)
        self.noise = a * self.noise + b * generate_noise()
        return True


class constant_sampler(noise_time_sampler):
    kind = 'constant'

    def update(self, generate_noise):
        if hasattr(self, 'noise') and self.counter % self.nsteps != 0:
            self.counter += 1
            return False
        self.noise = generate_noise()
        self.counter = 1
        return True


class stochastic_QGModel(QGModel):
    def __init__(self, pyqg_params, sampling_type='AR1', nsteps=1, **engine_kw):
        super().__init__(**pyqg_params, **engine_kw)
        self.sampling_type = sampling_type
        if sampling_type == 'AR1':
            self.noise_sampler = AR1_sampler(nsteps)
        elif sampling_type == 'constant':
            self.noise_sampler = constant_sampler(nsteps)
        elif sampling_type == 'deterministic':
            pass
        else:
            raise ValueError('Unknown sampling type')
