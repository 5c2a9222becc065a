#!/usr/bin/env python
"""Experiment (imports the oracle, so it lives under tests/): feasibility of evaluating generator layer 2 (128 -> 64 channels, 5x5, CIRCULAR padding) in Fourier
space — a per-wavenumber complex (64 x 128) matrix product, 12x fewer multiply-adds than the 25-tap stencil.

  * numerics (CPU, float32 FFTs + float32 products against a float64 direct convolution of the shipped GAN weights)
  * with --gpu: time of the three phases with LIBRARY kernels (torch.fft / complex bmm: rocFFT + rocBLAS) at the
    bench's shape (128 members, 64 x 64) as a bound on what hand-written kernels would have to beat; the product
    path stays the direct f16x3 MFMA kernel.
"""
import os
import sys
import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_ref


def weight_spectrum(W2, N):
    """rfft2 of the 5x5 kernels placed for cross-correlation with circular padding (float64)"""
    wk = np.zeros(W2.shape[:2] + (N, N))
    for ty in range(5):
        for tx in range(5):
            wk[:, :, (-(ty - 2)) % N, (-(tx - 2)) % N] = W2[:, :, ty, tx]
    return np.fft.rfft2(wk, axes=(-2, -1))


def main():
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'weights_gan.npz'))
    w = gen_ref.CNNWeights.from_npz_dict(d, 'net0_')
    N = 64
    W2, b2 = w.conv_w[1], w.conv_b[1]
    rs = np.random.RandomState(0)
    x = rs.randn(2, 4, N, N).astype('float32')
    x[:, :2] *= 1.5

    def conv(t, W, b, dt):
        tp = F.pad(t, (2, 2, 2, 2), mode='circular')
        return F.conv2d(tp, torch.as_tensor(W).to(dt), torch.as_tensor(b).to(dt))
    a1 = torch.relu(conv(torch.as_tensor(x).double(), w.conv_w[0], w.conv_b[0], torch.float64)).float()
    truth = conv(a1.double(), W2, b2, torch.float64).numpy()
    direct32 = conv(a1, W2, b2, torch.float32).numpy()
    wf = torch.as_tensor(weight_spectrum(W2, N).astype('complex64'))
    xf = torch.fft.rfft2(a1)
    y = torch.fft.irfft2(torch.einsum('oiyx,biyx->boyx', wf, xf), s=(N, N)).numpy() + b2[None, :, None, None]
    mx = np.abs(truth).max()
    print(f'float32 direct conv: max err / max|y| = {np.abs(direct32 - truth).max() / mx:.2e}')
    print(f'float32 FFT conv   : max err / max|y| = {np.abs(y - truth).max() / mx:.2e}')
    if '--gpu' in sys.argv:
        B = 128
        dev = 'cuda'
        a = torch.randn(B, 128, N, N, device=dev)
        wfd = wf.to(dev).permute(2, 3, 0, 1).contiguous()          # (64, 33, cout, cin)

        def timed(fn, n=20):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                out = fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3, out
        t_f, af = timed(lambda: torch.fft.rfft2(a))                 # (B, cin, 64, 33)
        afp = af.permute(2, 3, 1, 0).contiguous()                   # (64, 33, cin, B)
        t_g, yf = timed(lambda: torch.matmul(wfd, afp))             # (64, 33, cout, B)
        yfp = yf.permute(3, 2, 0, 1).contiguous()
        t_i, _ = timed(lambda: torch.fft.irfft2(yfp, s=(N, N)))
        print(f'library kernels, {B} members: rfft2 {t_f:.0f} us, per-wavenumber complex GEMM {t_g:.0f} us, irfft2 {t_i:.0f} us '
              f'(+ two layout permutes); direct f16x3 kernel: 444-468 us')


if __name__ == '__main__':
    main()
