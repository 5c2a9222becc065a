"""CPU: numerics of the two ways to evaluate generator layer 2 (128 -> 64 channels, 5x5, circular padding: 75 % of the
step's multiply-adds) with FEWER multiplications than the 25-tap stencil — the numbers DESIGN.md quotes for them.

  * Fourier space: a per-wavenumber complex (64 x 128) matrix product, 12x fewer multiply-adds.
  * Winograd / Toom-Cook F(2x2, 5x5): 36 multiplications per 2x2 outputs instead of 100 (2.78x fewer); transforms
    B^T d B of the input tile in float32, G g G^T of the weights in float64 on the host, operands split AFTER the
    transform into f16 hi/lo pairs (the f16x3 arithmetic of csrc/conv_half.hpp: hi*hi + hi*lo + lo*hi, float32
    accumulation), A^T m A of the products in float32.

Truth = float64 direct convolution of the same float32 parameters and inputs (oracle/gen_ref.py's weights from the
shipped fixtures; inputs = layer 1's real output).  Stated bound for a product path: the SAME 2e-5 of max|y| the golden
vectors are held to, for the whole 8-layer net; a single layer has to stay well inside it.
"""
import os
from fractions import Fraction

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from oracle import gen_ref


def _layer2_case(kind='gan', N=64, members=2, seed=0):
    d = np.load(os.path.join(GOLDEN, f'weights_{kind}.npz'))
    w = gen_ref.CNNWeights.from_npz_dict(d, 'net0_')
    W2, b2 = w.conv_w[1], w.conv_b[1]
    rs = np.random.RandomState(seed)
    x = rs.randn(members, w.n_in, N, N).astype('float32')
    x[:, :2] *= 1.5

    def conv(t, W, b, dt):
        tp = F.pad(t, (2, 2, 2, 2), mode='circular')
        return F.conv2d(tp, torch.as_tensor(W).to(dt), torch.as_tensor(b).to(dt))
    # layer 1's stored output as layer 2 sees it in the fused-BatchNorm ('fold') layout: the ReLU output
    a1 = torch.relu(conv(torch.as_tensor(x).double(), w.conv_w[0], w.conv_b[0], torch.float64)).float()
    # BatchNorm of layer 1 folded into layer 2's weights (exact under circular padding; conv.hip does the same)
    s = (w.bn_g[0] / np.sqrt(w.bn_v[0] + gen_ref.BN_EPS)).astype('float64')
    t = w.bn_b[0].astype('float64') - w.bn_m[0].astype('float64') * s
    Wf = W2.astype('float64') * s[None, :, None, None]
    bf = b2.astype('float64') + (W2.astype('float64') * t[None, :, None, None]).sum(axis=(1, 2, 3))
    truth = conv(a1.double(), Wf, bf, torch.float64).numpy()
    direct32 = conv(a1, Wf.astype('float32'), bf.astype('float32'), torch.float32).numpy()
    return a1, Wf, bf, truth, direct32


def _split_f16(x):
    """float32 tensor -> (hi, lo) float16 pair carried as float32 (x ~ hi + lo, 22 significant bits)"""
    hi = x.to(torch.float16).to(torch.float32)
    lo = (x - hi).to(torch.float16).to(torch.float32)
    return hi, lo


def test_fourier_space_layer2_is_float32_class():
    a1, Wf, bf, truth, direct32 = _layer2_case()
    N = a1.shape[-1]
    wk = np.zeros(Wf.shape[:2] + (N, N))
    for ty in range(5):
        for tx in range(5):
            wk[:, :, (-(ty - 2)) % N, (-(tx - 2)) % N] = Wf[:, :, ty, tx]
    wf = torch.as_tensor(np.fft.rfft2(wk, axes=(-2, -1)).astype('complex64'))
    y = torch.fft.irfft2(torch.einsum('oiyx,biyx->boyx', wf, torch.fft.rfft2(a1)), s=(N, N)).numpy() + bf[None, :, None, None]
    mx = np.abs(truth).max()
    e_direct, e_fft = np.abs(direct32 - truth).max() / mx, np.abs(y - truth).max() / mx
    assert e_direct < 5e-7 and e_fft < 1e-6, (e_direct, e_fft)          # measured 1.9e-7 / 2.1-2.3e-7


# ---- Toom-Cook / Winograd matrices in exact rational arithmetic ----------------------------------------------------
def toom_cook(m, r, points):
    """F(m, r): y = A^T [(G g) * (B^T d)] for n = m + r - 1 evaluation points, the last one at infinity.
    From the transposition of Toom-Cook polynomial multiplication: with V the n x n evaluation matrix of degree n-1
    polynomials, A^T = V_m^T, G = diag(1/f) V_r, B^T = diag(f) V^{-T}, f_j = prod_{l != j} (p_j - p_l) (the
    scaling that keeps B^T free of fractions for integer points).  -> (AT (m,n), G (n,r), BT (n,n)) as float64."""
    n = m + r - 1
    pts = [Fraction(p) for p in points]
    assert len(pts) == n - 1 and len(set(pts)) == n - 1

    def vander(cols):
        rows = [[p ** k for k in range(cols)] for p in pts]
        rows.append([Fraction(0)] * (cols - 1) + [Fraction(1)])
        return rows
    V = vander(n)
    f = []
    for j, p in enumerate(pts):
        v = Fraction(1)
        for l, q in enumerate(pts):
            if l != j:
                v *= p - q
        f.append(v)
    f.append(Fraction(1))
    # inverse of V by Gauss-Jordan in rationals
    M = [row[:] + [Fraction(int(i == j)) for j in range(n)] for i, row in enumerate(V)]
    for c in range(n):
        piv = next(i for i in range(c, n) if M[i][c] != 0)
        M[c], M[piv] = M[piv], M[c]
        pv = M[c][c]
        M[c] = [v / pv for v in M[c]]
        for i in range(n):
            if i != c and M[i][c] != 0:
                fac = M[i][c]
                M[i] = [a - fac * b for a, b in zip(M[i], M[c])]
    Vinv = [row[n:] for row in M]
    BT = [[f[j] * Vinv[k][j] for k in range(n)] for j in range(n)]           # diag(f) V^{-T}
    G = [[V_r / f[j] for V_r in vander(r)[j]] for j in range(n)]
    AT = [[vander(m)[j][i] for j in range(n)] for i in range(m)]
    tof = lambda A: np.array([[float(v) for v in row] for row in A])
    return tof(AT), tof(G), tof(BT)


def test_toom_cook_matrices_are_exact_algorithms():
    rs = np.random.RandomState(1)
    AT, G, BT = toom_cook(2, 3, (0, 1, -1))
    # the textbook F(2,3) up to the sign convention of the rows
    np.testing.assert_allclose(np.abs(BT), np.abs(np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]])))
    for m, r, pts in ((2, 3, (0, 1, -1)), (2, 5, (0, 1, -1, 2, -2)), (2, 5, (0, 1, -1, Fraction(1, 2), -2)), (4, 3, (0, 1, -1, 2, -2))):
        AT, G, BT = toom_cook(m, r, pts)
        g, d = rs.randn(r), rs.randn(m + r - 1)
        y = AT @ ((G @ g) * (BT @ d))
        ref = np.array([np.dot(g, d[i:i + r]) for i in range(m)])
        np.testing.assert_allclose(y, ref, rtol=0, atol=1e-12 * np.abs(ref).max())


def winograd_layer2(a1, Wf, bf, points, split=True, quantised_input=True):
    """layer 2 by F(2x2, 5x5) with the arithmetic of the planned kernel; -> y (B, 64, N, N) float32 and the largest
    transformed operands (for the f16 window)"""
    AT, G, BT = toom_cook(2, 5, points)
    B_, C, N, _ = a1.shape
    if quantised_input:                                     # activations arrive as f16 hi/lo pairs (22 bits)
        hi, lo = _split_f16(a1)
        a1 = hi + lo
    U = np.einsum('ik,ockl,jl->ocij', G, Wf, G)             # float64 on the host
    xp = F.pad(a1, (2, 3, 2, 3), mode='circular')           # tile t covers input rows 2t-2 .. 2t+3
    d = xp.unfold(2, 6, 2).unfold(3, 6, 2)[:, :, :N // 2, :N // 2]      # (B, C, T, T, 6, 6)
    BTt = torch.as_tensor(BT, dtype=torch.float32)
    V = torch.einsum('ik,bctukl,jl->bctuij', BTt, d, BTt)   # float32 input transform
    info = {'V_max': float(V.abs().max()), 'U_max': float(np.abs(U).max()), 'a_max': float(a1.abs().max())}
    if split:
        # per-position power-of-two scales: transformed weights to [2^13, 2^14), transformed inputs to <= 2^10
        su = 2.0 ** (13 - np.floor(np.log2(np.abs(U).max(axis=(0, 1)))))
        sv = 2.0 ** (10 - np.ceil(np.log2(V.abs().amax(dim=(0, 1, 2, 3)).numpy())))
        Uh, Ul = _split_f16(torch.as_tensor((U * su).astype('float32')))
        Vh, Vl = _split_f16(V * torch.as_tensor(sv, dtype=torch.float32))
        info['U_lo_subnormal_frac'] = float(((Ul != 0) & (Ul.abs() < 2.0 ** -14)).float().mean())
        info['V_lo_subnormal_frac'] = float(((Vl != 0) & (Vl.abs() < 2.0 ** -14)).float().mean())
        M = sum(torch.einsum('ocij,bctuij->botuij', a, b) for a, b in ((Uh, Vh), (Uh, Vl), (Ul, Vh)))
        M = M / torch.as_tensor(su * sv, dtype=torch.float32)
    else:
        M = torch.einsum('ocij,bctuij->botuij', torch.as_tensor(U.astype('float32')), V)
    ATt = torch.as_tensor(AT, dtype=torch.float32)
    Y = torch.einsum('ik,botukl,jl->botuij', ATt, M, ATt)   # (B, O, T, T, 2, 2)
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, -1, N, N) + torch.as_tensor(bf.astype('float32'))[None, :, None, None]
    return y.numpy(), info


POINT_SETS = {'0,1,-1,2,-2': (0, 1, -1, 2, -2), '0,1,-1,1/2,-1/2': (0, 1, -1, Fraction(1, 2), Fraction(-1, 2)),
              '0,1,-1,1/2,-2': (0, 1, -1, Fraction(1, 2), -2), '0,1,-1,2,-1/2': (0, 1, -1, 2, Fraction(-1, 2))}


@pytest.mark.parametrize('kind', ['gan', 'vae', 'gz'])
def test_winograd_f2x2_5x5_error_of_layer2(kind):
    """the measured error of one layer: float32 transforms with f32 products, and with the f16x3 split after the
    transform — against the float64 truth, next to the direct float32 convolution.  The best point set is what a kernel
    would use; the assertion is the bound DESIGN.md quotes."""
    a1, Wf, bf, truth, direct32 = _layer2_case(kind)
    mx = np.abs(truth).max()
    e_direct = np.abs(direct32 - truth).max() / mx
    res = {}
    for name, pts in POINT_SETS.items():
        y32, _ = winograd_layer2(a1, Wf, bf, pts, split=False, quantised_input=False)
        y16, info = winograd_layer2(a1, Wf, bf, pts, split=True)
        res[name] = (np.abs(y32 - truth).max() / mx, np.abs(y16 - truth).max() / mx, info)
    best = min(res, key=lambda k: res[k][1])
    print(f'{kind}: direct f32 {e_direct:.2e}; ' + '; '.join(f'{k}: f32 {v[0]:.2e} f16x3 {v[1]:.2e}' for k, v in res.items()))
    print('   best', best, res[best][2])
    assert res[best][1] < 5e-6, res            # one layer, against the whole net's 2e-5 budget
    assert res[best][2]['V_max'] < 65504 * 2 ** 10
