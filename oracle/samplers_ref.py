"""CPU restatement of the latent-noise time samplers and the device RNG.

TEST INFRASTRUCTURE — see oracle/__init__.py.

Reference followed: pyqg_generative/tools/stochastic_pyqg.py:30-54 (AR1_sampler),
:56-72 (constant_sampler), :74-88 (stochastic_QGModel: sampler selection).
Pinned by tests/golden/samplers.npz.

``philox4x32_10`` / ``philox_normal`` restate the counter-based generator the
HIP engine uses for on-device latent noise (Salmon et al. 2011, Philox-4x32
with 10 rounds; Box-Muller on the 4 outputs).  The reference itself draws from
numpy's unseeded global stream (cgan_regression.py:154-155), so there is no
reference bit pattern to match: the oracle pins OUR stream (integer-exact) and
tests check the distribution.
"""
import numpy as np


class AR1SamplerRef:
    def __init__(self, nsteps):
        self.nsteps = nsteps

    def update(self, generate_noise):
        if hasattr(self, 'noise'):
            if self.nsteps > 0:
                a = 1 - 1 / self.nsteps
                b = (1 / self.nsteps * (2 - 1 / self.nsteps)) ** 0.5
            else:
                a = 1
                b = 0
            self.noise = a * self.noise + b * generate_noise()
        else:
            self.noise = generate_noise()
        return True


class ConstantSamplerRef:
    def __init__(self, nsteps):
        self.nsteps = nsteps

    def update(self, generate_noise):
        compute = True
        if hasattr(self, 'noise'):
            if self.counter % self.nsteps == 0:
                self.noise = generate_noise()
                self.counter = 1
            else:
                self.counter += 1
                compute = False
        else:
            self.noise = generate_noise()
            self.counter = 1
        return compute


def make_sampler(sampling_type, nsteps):
    if sampling_type == 'AR1':
        return AR1SamplerRef(nsteps)
    if sampling_type == 'constant':
        return ConstantSamplerRef(nsteps)
    if sampling_type == 'deterministic':
        return None
    raise ValueError('Unknown sampling type')


# ---------------------------------------------------------------- Philox
PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox-4x32-10.  All arguments uint32 arrays (broadcastable)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = PHILOX_M0 * c0.astype(np.uint64)
            p1 = PHILOX_M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(PHILOX_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u32_to_unit_open(u):
    """(0,1] float32 from uint32: (u >> 8 + 1) * 2^-24 — never 0, so log() is finite."""
    return ((u >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -24)


def philox_normal(seed, member, step, n_elem):
    """Standard normals for one member and one step.

    Counter layout (must match csrc/noise.hip): c0 = quad index (element//4),
    c1 = step (low 32 bits), c2 = member id, c3 = step >> 32; key = (seed lo, seed hi).
    Each Philox call yields 4 uint32 -> 2 Box-Muller pairs -> 4 normals, stored at
    elements 4*quad .. 4*quad+3.
    Returns float32 array of n_elem (n_elem must be a multiple of 4) and the raw uint32s.
    """
    assert n_elem % 4 == 0
    quad = np.arange(n_elem // 4, dtype=np.uint32)
    r = philox4x32_10(quad, np.uint32(step & 0xFFFFFFFF), np.uint32(member),
                      np.uint32((step >> 32) & 0xFFFFFFFF),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    raw = np.stack(r, axis=1).reshape(-1)
    u = u32_to_unit_open(raw).reshape(-1, 4)
    two_pi = np.float32(6.283185307179586)
    rad0 = np.sqrt(np.float32(-2.0) * np.log(u[:, 0]))
    rad1 = np.sqrt(np.float32(-2.0) * np.log(u[:, 2]))
    out = np.empty((n_elem // 4, 4), dtype=np.float32)
    out[:, 0] = rad0 * np.cos(two_pi * u[:, 1])
    out[:, 1] = rad0 * np.sin(two_pi * u[:, 1])
    out[:, 2] = rad1 * np.cos(two_pi * u[:, 3])
    out[:, 3] = rad1 * np.sin(two_pi * u[:, 3])
    return out.reshape(-1), raw
