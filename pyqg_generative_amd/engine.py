"""Thin object wrappers over the C ABI (include/qgx.h).

PyTorch tensors are used as device-memory containers and for the stream only;
all arithmetic happens inside libqgx.so.
"""
import ctypes as C
import numpy as np
import torch

from . import _lib
from ._lib import lib, check

PYQG_DEFAULTS = dict(L=1e6, dt=7200., rek=5.787e-7, delta=0.25, beta=1.5e-11, rd=15000.0,
                     U1=0.025, U2=0.0, H1=500., filterfac=23.6)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Generator:
    """Device-resident generator (CGAN G / CVAE decoder, each optionally with the regression net `net_mean` as a second
    net / GZ mean+var nets)."""
    KINDS = {'gan': _lib.GEN_GAN, 'vae': _lib.GEN_VAE, 'gz': _lib.GEN_GZ}

    def __init__(self, kind, nets, x_std, y_std, device=0):
        """nets: list of dicts with float32 numpy arrays
        conv_w[8], conv_b[8], bn_g[7], bn_b[7], bn_m[7], bn_v[7] (PyTorch layouts)."""
        self.kind = kind
        self.device = device
        self._h = C.c_void_p(0)
        keep = []
        arr = (_lib.qgx_cnn_weights * len(nets))()
        for n, net in enumerate(nets):
            w = arr[n]
            w.n_in = int(net['conv_w'][0].shape[1])
            w.n_out = int(net['conv_w'][7].shape[0])
            w.bn_eps = 1e-5
            for i in range(8):
                cw = np.ascontiguousarray(net['conv_w'][i], dtype=np.float32)
                cb = np.ascontiguousarray(net['conv_b'][i], dtype=np.float32)
                keep += [cw, cb]
                w.conv_w[i] = cw.ctypes.data
                w.conv_b[i] = cb.ctypes.data
            for i in range(7):
                for field, key in (('bn_gamma', 'bn_g'), ('bn_beta', 'bn_b'),
                                   ('bn_mean', 'bn_m'), ('bn_var', 'bn_v')):
                    a = np.ascontiguousarray(net[key][i], dtype=np.float32)
                    keep.append(a)
                    getattr(w, field)[i] = a.ctypes.data
        xs = (C.c_float * 2)(*[float(v) for v in np.asarray(x_std, np.float32).reshape(-1)])
        ys = (C.c_float * 2)(*[float(v) for v in np.asarray(y_std, np.float32).reshape(-1)])
        self.x_std = np.asarray(x_std, np.float32).reshape(-1)
        self.y_std = np.asarray(y_std, np.float32).reshape(-1)
        check(lib.qgx_generator_create(self.KINDS[kind], arr, len(nets), xs, ys, device, C.byref(self._h)))
        self.n_in = 2 if kind == 'gz' else 4

    @property
    def noise_dtype(self):
        return torch.float64 if self.kind == 'gz' else torch.float32

    # ---- f16x3 range guard ---------------------------------------------------------------------
    def info(self):
        """What the calibration at construction decided: dict(precision 0|3, ascale_log2, fold, layer_absmax)."""
        p, e, f = C.c_int(0), C.c_int(0), C.c_int(0)
        mx = (C.c_float * 10)()
        check(lib.qgx_generator_info(self._h, C.byref(p), C.byref(e), C.byref(f), mx))
        return dict(precision=p.value, ascale_log2=e.value, fold=f.value, layer_absmax=list(mx))

    def wino_info(self, N=None):
        """the 5x5 layer's 1-D Winograd form: dict(enabled, chosen_by_calibration, calibration_error, N) — at a grid size it
        is the default only if its outputs on calibration inputs OF THAT SIZE stayed within 1e-5 of the exact-f32 kernels' at
        construction (measured at each of 32, 48, 64, 96, 128).  N = None: the 64 x 64 entry"""
        en, au, err = C.c_int(0), C.c_int(0), C.c_float(0)
        if N is None:
            check(lib.qgx_generator_wino_info(self._h, C.byref(en), C.byref(au), C.byref(err)))
        else:
            check(lib.qgx_generator_wino_info_n(self._h, int(N), C.byref(en), C.byref(au), C.byref(err)))
        return dict(enabled=bool(en.value), chosen_by_calibration=bool(au.value), calibration_error=err.value, N=64 if N is None else int(N))

    LAYER2_KERNELS = ('exact-f32', '25-tap', '25-tap split-K', 'winograd', 'winograd, transform under the MFMAs')

    def layer2_kernel(self, B, N, inet=0):
        """index into LAYER2_KERNELS of the kernel the 5x5 layer takes for B members at N x N under the options in force"""
        k = C.c_int(0)
        check(lib.qgx_generator_layer2_kernel(self._h, int(inet), int(B), int(N), C.byref(k)))
        return k.value

    def range_read(self):
        """Synchronise and return (flags, input_absmax) of the range guard since the last read; clears them.
        flags bit l: conv layer l+1 stored an activation beyond the f16 range; bit 31: non-finite forcing."""
        fl, mx = C.c_uint(0), C.c_float(0)
        check(lib.qgx_generator_range_read(self._h, C.byref(fl), C.byref(mx), _stream()))
        return fl.value, mx.value

    def range_ok(self):
        """-> None if every value stayed inside the 16-bit window since the last check, else a description."""
        if self.precision == 0:          # exact-f32 kernels store no 16-bit activations: nothing to read, no synchronisation
            return None
        flags, in_max = self.range_read()
        if flags == 0 and in_max <= 65504.:
            return None
        layers = [l + 1 for l in range(8) if flags >> l & 1]
        return (f'f16x3 generator arithmetic left its range: overflow in conv layer(s) {layers}, '
                f'non-finite forcing: {bool(flags >> 31)}, largest |network input| {in_max:g}')

    def _guarded(self, launch):
        """Run `launch()`; if the 16-bit window was left, switch this generator to the exact-f32 kernels for good
        and run it again.  What the caller receives is inside the 2e-5 (of max|y|) the golden vectors are held to: float32
        error class (1-2e-6) from the 25-tap and exact-f32 kernels, 3-9e-6 where the Winograd form of the 5x5 layer was
        admitted by its calibration at this grid size (bound 1e-5, `wino_info(N)`; `set_option('wino', 0)` turns it off)."""
        out = launch()
        if self.check_range:
            why = self.range_ok()
            if why is not None:
                import warnings
                warnings.warn(why + '; switching this generator to the exact-f32 kernels', RuntimeWarning)
                self.set_option('precision', 0)
                out = launch()
        return out

    check_range = True

    def guarded_loop(self, body):
        """Run `body()` — a loop of many forward launches (Monte-Carlo sampling, minibatches) — with ONE range check
        after it instead of a device-to-host read and a stream synchronisation per launch; if the 16-bit window was left
        anywhere in the loop, switch to the exact-f32 kernels for good and run the whole loop again."""
        if not self.check_range:
            return body()
        self.check_range = False
        try:
            out = body()
            why = self.range_ok()
            if why is not None:
                import warnings
                warnings.warn(why + '; switching this generator to the exact-f32 kernels', RuntimeWarning)
                self.set_option('precision', 0)
                out = body()
        finally:
            del self.check_range            # back to the class default
        return out

    def forward(self, q, z, demean=True, out=None):
        """q: (B,2,N,N) float64 cuda; z: (B,2,N,N) float32 (float64 for gz) -> S (B,2,N,N) float64."""
        assert q.is_cuda and q.dtype == torch.float64 and q.is_contiguous() and q.dim() == 4
        assert z.is_cuda and z.dtype == self.noise_dtype and z.is_contiguous()
        B, _, N, _ = q.shape
        assert z.numel() == q.numel()
        S = out if out is not None else torch.empty_like(q)

        def launch():
            check(lib.qgx_generator_forward(self._h, _ptr(q), _ptr(z), _ptr(S), B, N, int(bool(demean)), _stream()))
            return S
        return self._guarded(launch)

    def cnn_forward(self, x, inet=0):
        """Raw AndrewCNN forward: x (B,n_in,N,N) float32 -> (B,2,N,N) float32."""
        n_in = 2 if (self.kind == 'gz' or inet == 1) else 4          # net 1 of a GAN / VAE generator: the regression net
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] == n_in
        B, _, N, _ = x.shape
        y = torch.empty((B, 2, N, N), dtype=torch.float32, device=x.device)

        def launch():
            check(lib.qgx_cnn_forward(self._h, inet, _ptr(x), _ptr(y), B, N, _stream()))
            return y
        return self._guarded(launch)

    def set_option(self, name, value):
        check(lib.qgx_generator_set_option(self._h, name.encode(), int(value)))
        if name in ('precision', 'auto'):
            self._precision = None

    @property
    def precision(self):
        """arithmetic of the kernels in use: 0 exact f32, 3 f16x3 (cached: asked after every guarded launch)"""
        if getattr(self, '_precision', None) is None:
            self._precision = self.info()['precision']
        return self._precision

    def profile(self, layer):
        """Bracket every launch of conv layer `layer` (0..7; -1 = off) with HIP events."""
        check(lib.qgx_generator_profile(self._h, int(layer)))

    def profile_read(self):
        """-> (summed kernel milliseconds, launches) since the last read."""
        ms, n = C.c_double(0), C.c_int64(0)
        check(lib.qgx_generator_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if self._h:
            lib.qgx_generator_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EnsembleEngine:
    """B independent two-layer QG members resident on one GPU."""

    def __init__(self, nx=64, n_members=1, device=0, plan_only=False, **params):
        """plan_only: an FFT plan of the grid for rfft2 / irfft2 (tools/operators.py::Dev) — tables and work space, no
        model state, a tenth of the device memory and of the creation time of a model"""
        cfg = _lib.qgx_config()
        cfg.plan_only = int(bool(plan_only))
        p = dict(PYQG_DEFAULTS)
        for k, v in params.items():
            if k not in p:
                raise TypeError(f'unknown model parameter {k!r}')
            p[k] = v
        cfg.nx, cfg.n_members, cfg.device = int(nx), int(n_members), int(device)
        for k, v in p.items():
            setattr(cfg, k, float(v))
        self.params = p
        self.N, self.NK, self.B = int(nx), int(nx) // 2 + 1, int(n_members)
        self.device = torch.device('cuda', device)
        self._h = C.c_void_p(0)
        self._generators = []           # generators used by step(): their range guard is read at status time
        check(lib.qgx_create(C.byref(cfg), C.byref(self._h)))

    # ---- tables -------------------------------------------------------------------
    def table(self, which):
        N, NK = self.N, self.NK
        shape = {_lib.T_FILTR: (N, NK), _lib.T_WV2: (N, NK), _lib.T_A: (2, 2, N, NK),
                 _lib.T_KK: (NK,), _lib.T_LL: (N,)}[which]
        out = np.empty(shape, dtype=np.float64)
        check(lib.qgx_get_table(self._h, which, out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- state --------------------------------------------------------------------
    def _real(self):
        return torch.empty((self.B, 2, self.N, self.N), dtype=torch.float64, device=self.device)

    def _spec(self):
        return torch.empty((self.B, 2, self.N, self.NK), dtype=torch.complex128, device=self.device)

    def get(self, field, noise_dtype=None):
        """copy of a state field.  F_Z (the latent noise) has the element type of the generator last stepped with — float32
        for GAN / VAE, float64 for GZ — which the library reports (qgx_field_bytes); a `noise_dtype` that contradicts it is
        refused rather than handed a buffer of the wrong size"""
        if field in (_lib.F_Q, _lib.F_U, _lib.F_V, _lib.F_S, _lib.F_P):
            out = self._real()
        elif field == _lib.F_Z:
            nbytes = int(lib.qgx_field_bytes(self._h, field))
            dtype = torch.float64 if nbytes == self.B * 2 * self.N * self.N * 8 else torch.float32
            if noise_dtype is not None and noise_dtype != dtype:
                raise ValueError(f'the latent noise of this model is {dtype}, not {noise_dtype}')
            out = torch.empty((self.B, 2, self.N, self.N), dtype=dtype, device=self.device)
        else:
            out = self._spec()
        check(lib.qgx_get(self._h, field, _ptr(out), _stream()))
        return out

    def set_q(self, q):
        q = torch.as_tensor(q, dtype=torch.float64, device=self.device).contiguous()
        assert tuple(q.shape) == (self.B, 2, self.N, self.N), q.shape
        check(lib.qgx_set_q(self._h, _ptr(q), _stream()))
        torch.cuda.current_stream().synchronize()     # q may be a temporary

    def set_qh(self, qh):
        qh = torch.as_tensor(qh, dtype=torch.complex128, device=self.device).contiguous()
        assert tuple(qh.shape) == (self.B, 2, self.N, self.NK), qh.shape
        check(lib.qgx_set_qh(self._h, _ptr(qh), _stream()))
        torch.cuda.current_stream().synchronize()

    def invert(self):
        check(lib.qgx_invert(self._h, _stream()))

    @property
    def tc(self):
        return int(lib.qgx_step_count(self._h))

    @property
    def run_kernel_state(self):
        """256 x 256 grids: 1 = unparameterized runs of steps execute as single persistent launches, -1 = not on this
        device (three launches per step), 0 = not decided yet / other grid"""
        return int(lib.qgx_run_kernel_state(self._h))

    def reset_time(self):
        check(lib.qgx_reset_time(self._h))

    def set_option(self, name, value):
        """kernel-path switch of this model (include/qgx.h::qgx_set_option): same results, different fusion / tiling"""
        check(lib.qgx_set_option(self._h, name.encode(), int(value)))

    def status(self):
        """-> (KE[B], CFL[B]) as pyqg's _print_status computes them (from the last inversion)."""
        out = torch.empty((self.B, 2), dtype=torch.float64, device=self.device)
        check(lib.qgx_status_ke_cfl(self._h, _ptr(out), _stream()))
        out = out.cpu().numpy()
        self.check_generators()
        return out[:, 0], out[:, 1]

    def check_generators(self):
        """The fused step cannot re-run a forcing after the fact: a generator that left its 16-bit window since the
        last check has corrupted the members' state, so this raises (status / snapshot cadence of the run loop)."""
        self._generators = [g for g in self._generators if g._h]        # a closed generator has nothing left to report
        for g in self._generators:
            why = g.range_ok() if g.check_range else None
            if why is not None:
                raise FloatingPointError(why + "; re-run with Generator.set_option('precision', 0)")

    # ---- time-averaged diagnostics --------------------------------------------------
    def diag_config(self, start_step, every):
        check(lib.qgx_diag_config(self._h, int(start_step), int(every)))

    @property
    def diag_count(self):
        return int(lib.qgx_diag_count(self._h))

    def diag_reset(self):
        check(lib.qgx_diag_reset(self._h))

    def diag(self, name):
        """time mean of diagnostic `name` per member: (B,2,N,NK) for KEspec/Ensspec else (B,N,NK) (device tensor)"""
        i = _lib.DIAGS.index(name)
        shape = (self.B, 2, self.N, self.NK) if i < 2 else (self.B, self.N, self.NK)
        out = torch.empty(shape, dtype=torch.float64, device=self.device)
        check(lib.qgx_diag_get(self._h, i, _ptr(out), _stream()))
        return out

    # ---- stepping -----------------------------------------------------------------
    def step(self, nsteps=1, generator=None, sampling='AR1', nsteps_decor=1, weight=1.0, seed=0,
             member_offset=0, z_external=None, forcing=None, demean=None, refresh_diag=True):
        p = None
        keep = []
        if generator is not None or forcing is not None:
            p = _lib.qgx_param()
            p.gen = generator._h if generator is not None else None
            if generator is not None and generator not in self._generators:
                self._generators.append(generator)
            p.sampling = {'AR1': _lib.SAMPLING_AR1, 'constant': _lib.SAMPLING_CONSTANT}[sampling]
            p.nsteps = int(nsteps_decor)
            p.weight = float(weight)
            p.seed = int(seed)
            p.member_offset = int(member_offset)
            if z_external is not None:
                assert z_external.is_cuda and z_external.is_contiguous()
                keep.append(z_external)
                p.z_external_dev = z_external.data_ptr()
            if forcing is not None:
                assert forcing.is_cuda and forcing.dtype == torch.float64 and forcing.is_contiguous()
                keep.append(forcing)
                p.forcing_dev = forcing.data_ptr()
            if demean is None:
                demean = generator is not None
            p.demean = int(bool(demean))
        check(lib.qgx_step(self._h, int(nsteps), C.byref(p) if p is not None else None,
                           int(bool(refresh_diag)), _stream()))
        if keep:
            torch.cuda.current_stream().synchronize()

    def step_streams(self, generator=None):
        """1 or 2: the internal streams `step` advances this ensemble on with `generator` attached (qgx_step_streams)"""
        if generator is None:
            return 1
        p = _lib.qgx_param()
        p.gen = generator._h
        return int(lib.qgx_step_streams(self._h, C.byref(p)))

    def close(self):
        if self._h:
            lib.qgx_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
