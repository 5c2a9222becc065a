// In-LDS mixed-radix complex FFT passes (float64) for gfx950.
//
// Forward transform = decimation-in-frequency, in place, natural order in ->
// digit-reversed order out.  Inverse = the transposed flow graph
// (decimation-in-time, conjugate twiddles), digit-reversed in -> natural out.
// Spectral-space code addresses coefficients through SpecDev::pos[], so no
// reordering pass is ever executed.  Radices 2, 3, 4, 6, 8, 12, 16.
//
// A "line" is one 1-D transform of length N living in LDS at
//   base + line * line_stride + e * elem_stride      (units: double2)
// Work items (line, butterfly) are dealt so that consecutive lanes take
// consecutive LINES: with the row stride padded to N+1 elements both the row
// passes (lines = rows, stride N+1) and the column passes (lines = columns,
// stride 1) are free of LDS bank conflicts.
#pragma once
#include <hip/hip_runtime.h>

namespace qgx {

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) {  // a * conj(b)
    return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ double2 cscale(double2 a, double s) { return make_double2(a.x * s, a.y * s); }
// multiply by -i (forward) / +i (inverse)
__device__ __forceinline__ double2 mul_mi(double2 a) { return make_double2(a.y, -a.x); }
__device__ __forceinline__ double2 mul_pi(double2 a) { return make_double2(-a.y, a.x); }

template <int R, bool FWD>
__device__ __forceinline__ void small_dft(double2 (&v)[R]) {
    if constexpr (R == 2) {
        double2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    } else if constexpr (R == 4) {
        double2 s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
        double2 s13 = cadd(v[1], v[3]), d13 = csub(v[1], v[3]);
        double2 r = FWD ? mul_mi(d13) : mul_pi(d13);
        v[0] = cadd(s02, s13);
        v[1] = cadd(d02, r);
        v[2] = csub(s02, s13);
        v[3] = csub(d02, r);
    } else if constexpr (R == 8) {
        // two interleaved radix-4 transforms (even / odd inputs) + one radix-2 stage with W8^k
        double2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
        small_dft<4, FWD>(e);
        small_dft<4, FWD>(o);
        const double h = 0.70710678118654752440;
        // W8^1 = (1 -+ i)/sqrt2, W8^2 = -+ i, W8^3 = (-1 -+ i)/sqrt2   (upper sign: forward)
        double2 t1, t2, t3;
        if (FWD) {
            t1 = make_double2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
            t2 = mul_mi(o[2]);
            t3 = make_double2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
        } else {
            t1 = make_double2(h * (o[1].x - o[1].y), h * (o[1].y + o[1].x));
            t2 = mul_pi(o[2]);
            t3 = make_double2(-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y));
        }
        v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
        v[1] = cadd(e[1], t1);   v[5] = csub(e[1], t1);
        v[2] = cadd(e[2], t2);   v[6] = csub(e[2], t2);
        v[3] = cadd(e[3], t3);   v[7] = csub(e[3], t3);
    } else if constexpr (R == 16) {
        // 16 = 4 x 4, decimation in time over the four interleaved length-4 sequences x[4m + r]:
        // F_r = DFT4; G_r[k1] = W16^(r k1) F_r[k1]; X[k1 + 4 k2] = DFT4 over r of G_r[k1]
        double2 f0[4] = {v[0], v[4], v[8], v[12]}, f1[4] = {v[1], v[5], v[9], v[13]};
        double2 f2[4] = {v[2], v[6], v[10], v[14]}, f3[4] = {v[3], v[7], v[11], v[15]};
        small_dft<4, FWD>(f0);
        small_dft<4, FWD>(f1);
        small_dft<4, FWD>(f2);
        small_dft<4, FWD>(f3);
        const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
        const double sg = FWD ? -1.0 : 1.0;
        // W16^j = (cos(pi j / 8), sg sin(pi j / 8))
        f1[1] = cmul(f1[1], make_double2(c1, sg * s1));                         // W^1
        f1[2] = cmul(f1[2], make_double2(h, sg * h));                           // W^2
        f1[3] = cmul(f1[3], make_double2(s1, sg * c1));                         // W^3
        f2[1] = cmul(f2[1], make_double2(h, sg * h));                           // W^2
        f2[2] = FWD ? mul_mi(f2[2]) : mul_pi(f2[2]);                            // W^4
        f2[3] = cmul(f2[3], make_double2(-h, sg * h));                          // W^6
        f3[1] = cmul(f3[1], make_double2(s1, sg * c1));                         // W^3
        f3[2] = cmul(f3[2], make_double2(-h, sg * h));                          // W^6
        f3[3] = cmul(f3[3], make_double2(-c1, -sg * s1));                       // W^9 = -W^1
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            double2 g[4] = {f0[k1], f1[k1], f2[k1], f3[k1]};
            small_dft<4, FWD>(g);
            v[k1] = g[0]; v[k1 + 4] = g[1]; v[k1 + 8] = g[2]; v[k1 + 12] = g[3];
        }
    } else if constexpr (R == 6) {
        // 6 = 2 x 3, decimation in time: E = DFT3(even), O = DFT3(odd); X[k] = E[k] + W6^k O[k], X[k+3] = E[k] - W6^k O[k]
        double2 e[3] = {v[0], v[2], v[4]}, o[3] = {v[1], v[3], v[5]};
        small_dft<3, FWD>(e);
        small_dft<3, FWD>(o);
        const double s3 = 0.86602540378443864676;            // sin(pi/3)
        // W6^1 = (1/2, -+s3), W6^2 = (-1/2, -+s3)   (upper sign: forward)
        const double2 w1 = make_double2(0.5, FWD ? -s3 : s3), w2 = make_double2(-0.5, FWD ? -s3 : s3);
        const double2 t1 = cmul(o[1], w1), t2 = cmul(o[2], w2);
        v[0] = cadd(e[0], o[0]); v[3] = csub(e[0], o[0]);
        v[1] = cadd(e[1], t1);   v[4] = csub(e[1], t1);
        v[2] = cadd(e[2], t2);   v[5] = csub(e[2], t2);
    } else if constexpr (R == 12) {
        // 12 = 4 x 3, decimation in time over the three interleaved length-4 sequences x[3m + r]:
        // F_r = DFT4; G_r[k1] = W12^(r k1) F_r[k1]; X[k1 + 4 k2] = DFT3 over r of G_r[k1]
        double2 f0[4] = {v[0], v[3], v[6], v[9]}, f1[4] = {v[1], v[4], v[7], v[10]}, f2[4] = {v[2], v[5], v[8], v[11]};
        small_dft<4, FWD>(f0);
        small_dft<4, FWD>(f1);
        small_dft<4, FWD>(f2);
        const double c30 = 0.86602540378443864676, sg = FWD ? -1.0 : 1.0;
        // W12^1 = (c30, sg/2), W12^2 = (1/2, sg c30), W12^3 = (0, sg), W12^4 = (-1/2, sg c30), W12^6 = -1
        f1[1] = cmul(f1[1], make_double2(c30, sg * 0.5));
        f1[2] = cmul(f1[2], make_double2(0.5, sg * c30));
        f1[3] = FWD ? mul_mi(f1[3]) : mul_pi(f1[3]);
        f2[1] = cmul(f2[1], make_double2(0.5, sg * c30));
        f2[2] = cmul(f2[2], make_double2(-0.5, sg * c30));
        f2[3] = make_double2(-f2[3].x, -f2[3].y);
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            double2 g[3] = {f0[k1], f1[k1], f2[k1]};
            small_dft<3, FWD>(g);
            v[k1] = g[0]; v[k1 + 4] = g[1]; v[k1 + 8] = g[2];
        }
    } else {  // R == 3
        const double c = -0.5, s = 0.86602540378443864676;   // cos, sin of 2 pi / 3
        double2 t = cadd(v[1], v[2]);
        double2 d = csub(v[1], v[2]);
        double2 m = make_double2(v[0].x + c * t.x, v[0].y + c * t.y);
        // FWD: w = exp(-2 pi i/3): y1 = m - i s d, y2 = m + i s d ; inverse swaps
        double2 isd = make_double2(-s * d.y, s * d.x);       // i * s * d
        v[0] = cadd(v[0], t);
        if (FWD) { v[1] = csub(m, isd); v[2] = cadd(m, isd); }
        else     { v[1] = cadd(m, isd); v[2] = csub(m, isd); }
    }
}

// One pass of radix R with current block size n over `nl` lines.
template <int R, bool FWD>
__device__ __forceinline__ void fft_pass(double2 *Z, int nl, int ls, int es, int n, int N,
                                         const double2 *__restrict__ tw) {
    const int sub = n / R;
    const int per_line = N / R;
    const int total = nl * per_line;
    const int tstride = N / n;
    for (int w = threadIdx.x; w < total; w += blockDim.x) {
        const int bb = w / nl;
        const int line = w - bb * nl;
        const int blk = bb / sub;
        const int b = bb - blk * sub;
        double2 *base = Z + line * ls + (blk * n + b) * es;
        const int step = sub * es;
        double2 v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) v[m] = base[m * step];
        // (when the block is one butterfly, sub == 1, every twiddle is 1: skip loads and multiplies)
        if constexpr (FWD) {
            small_dft<R, true>(v);
            if (sub > 1) {
#pragma unroll
                for (int m = 1; m < R; ++m) v[m] = cmul(v[m], tw[m * b * tstride]);
            }
        } else {
            if (sub > 1) {
#pragma unroll
                for (int m = 1; m < R; ++m) v[m] = cmulc(v[m], tw[m * b * tstride]);
            }
            small_dft<R, false>(v);
        }
#pragma unroll
        for (int m = 0; m < R; ++m) base[m * step] = v[m];
    }
}

template <bool FWD>
__device__ __forceinline__ void fft_pass_any(int R, double2 *Z, int nl, int ls, int es, int n, int N,
                                             const double2 *__restrict__ tw) {
    if (R == 8) fft_pass<8, FWD>(Z, nl, ls, es, n, N, tw);
    else if (R == 16) fft_pass<16, FWD>(Z, nl, ls, es, n, N, tw);
    else if (R == 12) fft_pass<12, FWD>(Z, nl, ls, es, n, N, tw);
    else if (R == 6) fft_pass<6, FWD>(Z, nl, ls, es, n, N, tw);
    else if (R == 4) fft_pass<4, FWD>(Z, nl, ls, es, n, N, tw);
    else if (R == 2) fft_pass<2, FWD>(Z, nl, ls, es, n, N, tw);
    else fft_pass<3, FWD>(Z, nl, ls, es, n, N, tw);
}

// 1-D transforms along `nl` lines; block-wide, ends with a barrier.
__device__ __forceinline__ void fft_lines_fwd(double2 *Z, int nl, int ls, int es, int N, int nrad,
                                              const int *rad, const double2 *tw) {
    int n = N;
    for (int p = 0; p < nrad; ++p) {
        fft_pass_any<true>(rad[p], Z, nl, ls, es, n, N, tw);
        n /= rad[p];
        __syncthreads();
    }
}
__device__ __forceinline__ void fft_lines_inv(double2 *Z, int nl, int ls, int es, int N, int nrad,
                                              const int *rad, const double2 *tw) {
    int n = 1;
    for (int p = nrad - 1; p >= 0; --p) {
        n *= rad[p];
        fft_pass_any<false>(rad[p], Z, nl, ls, es, n, N, tw);
        __syncthreads();
    }
}

// ---- compile-time plans: the same greedy radix order as the host's factor_radices (8, 4, 2, 3), with
// every size a constant so that the index arithmetic of the passes folds to shifts and multiplies
// (radix 12 = 4 x 3 and 6 = 2 x 3 finish 96 = 8 x 12 and 48 = 8 x 6 in two passes instead of three; the passes
// are LDS-bandwidth bound, so a pass less is a third of the transform time less)
// (radix 16 only on the large grids — n >= 128, and the 16 left over by it: 256 = 16 x 16, 128 = 16 x 8 in two
// passes instead of three; the LDS-resident small grids keep their tuned plans)
constexpr int pick_radix(int n) {
    return (n % 16 == 0 && (n >= 128 || n == 16)) ? 16 : n % 8 == 0 ? 8 : (n % 12 == 0 ? 12 : ((n % 6 == 0 && n % 4 != 0) ? 6 : (n % 4 == 0 ? 4 : (n % 2 == 0 ? 2 : 3))));
}

template <int N, int n>
__device__ __forceinline__ void fft_lines_fwd_t(double2 *Z, int nl, int ls, int es, const double2 *tw) {
    if constexpr (n > 1) {
        constexpr int R = pick_radix(n);
        fft_pass<R, true>(Z, nl, ls, es, n, N, tw);
        __syncthreads();
        fft_lines_fwd_t<N, n / R>(Z, nl, ls, es, tw);
    }
}
template <int N, int n>
__device__ __forceinline__ void fft_lines_inv_t(double2 *Z, int nl, int ls, int es, const double2 *tw) {
    if constexpr (n > 1) {
        constexpr int R = pick_radix(n);
        fft_lines_inv_t<N, n / R>(Z, nl, ls, es, tw);
        fft_pass<R, false>(Z, nl, ls, es, n, N, tw);
        __syncthreads();
    }
}
// NN > 0: compile-time grid size; NN == 0: run-time plan from SpecDev
template <int NN>
__device__ __forceinline__ void fft2d_fwd_x(double2 *Z, int N, int LD, int nrad, const int *rad, const double2 *tw) {
    if constexpr (NN > 0) {
        fft_lines_fwd_t<NN, NN>(Z, NN, NN + 1, 1, tw);
        fft_lines_fwd_t<NN, NN>(Z, NN, 1, NN + 1, tw);
    } else {
        fft_lines_fwd(Z, N, LD, 1, N, nrad, rad, tw);
        fft_lines_fwd(Z, N, 1, LD, N, nrad, rad, tw);
    }
}
template <int NN>
__device__ __forceinline__ void fft2d_inv_x(double2 *Z, int N, int LD, int nrad, const int *rad, const double2 *tw) {
    if constexpr (NN > 0) {
        fft_lines_inv_t<NN, NN>(Z, NN, 1, NN + 1, tw);
        fft_lines_inv_t<NN, NN>(Z, NN, NN + 1, 1, tw);
    } else {
        fft_lines_inv(Z, N, 1, LD, N, nrad, rad, tw);
        fft_lines_inv(Z, N, LD, 1, N, nrad, rad, tw);
    }
}

// full 2-D transforms of an N x N field with row stride LD (callers barrier BEFORE)
__device__ __forceinline__ void fft2d_fwd(double2 *Z, int N, int LD, int nrad, const int *rad,
                                          const double2 *tw) {
    fft_lines_fwd(Z, N, LD, 1, N, nrad, rad, tw);   // along x, lines = rows
    fft_lines_fwd(Z, N, 1, LD, N, nrad, rad, tw);   // along y, lines = columns
}
__device__ __forceinline__ void fft2d_inv(double2 *Z, int N, int LD, int nrad, const int *rad,
                                          const double2 *tw) {
    fft_lines_inv(Z, N, 1, LD, N, nrad, rad, tw);
    fft_lines_inv(Z, N, LD, 1, N, nrad, rad, tw);
}

}  // namespace qgx
