// Device kernels behind the coarse-graining / re-gridding / subgrid-forcing operators
// (reference: pyqg_generative/tools/operators.py:84-99 gauss_filter / model_filter,
// :117-132 cut_off, :134-190 fft_interpolate, :192-202 clean_2h, :241-247 divergence,
// :249-268 advect).  Every operator of the reference is a composition of
//   rfft2 -> [move the resolved block of modes between grids, zero the 2h harmonics,
//             scale, multiply by a real spectral filter] -> irfft2,
// a pointwise real product, and the spectral divergence ik*A + il*B.
#include "common.hpp"

namespace qgx {
int large_q_to_qh(qgx_model *m, const double *q, double2 *qh, hipStream_t st);
int large_qh_to_q(qgx_model *m, const double2 *qh, double *q, hipStream_t st);

// dst (M,N,N/2+1) <- src (M,n,n/2+1): rows [0,h) and the last h rows, columns [0,h], h = min(n,N)/2
__global__ void k_spec_regrid(const double2 *src, double2 *dst, int n, int N, double scale, int zero_src_2h,
                              int zero_dst_2h, const double *filt) {
    const int nk = n / 2 + 1, NK = N / 2 + 1, h = (n < N ? n : N) / 2;
    const int f = blockIdx.y;
    const double2 *s = src + (size_t)f * n * nk;
    double2 *o = dst + (size_t)f * N * NK;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < N * NK; idx += gridDim.x * blockDim.x) {
        const int J = idx / NK, I = idx - J * NK;
        double2 val = make_double2(0., 0.);
        int js = -1;
        if (J < h) js = J;
        else if (J >= N - h) js = n - (N - J);
        if (js >= 0 && I <= h) {
            val = s[(size_t)js * nk + I];
            // fft_interpolate zeroes xf[h,0] of the SOURCE before the copy (operators.py:155-159)
            if (zero_src_2h && js == h && I == 0) val = make_double2(0., 0.);
        }
        if (zero_dst_2h && ((J == h && I == 0) || I == h)) val = make_double2(0., 0.);
        double sc = scale;
        if (filt) sc *= filt[idx];
        o[idx] = make_double2(val.x * sc, val.y * sc);
    }
}

// out = ik * A + il * B  (or, with B == nullptr, ik*A; with A == nullptr, il*B)
__global__ void k_spec_div(const double2 *A, const double2 *B, double2 *out, int N, double dk) {
    const int NK = N / 2 + 1, f = blockIdx.y;
    const size_t o = (size_t)f * N * NK;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < N * NK; idx += gridDim.x * blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        const double kx = dk * (double)i, ly = dk * (double)(j < N / 2 ? j : j - N);
        double re = 0., im = 0.;
        if (A) { const double2 a = A[o + idx]; re += -kx * a.y; im += kx * a.x; }
        if (B) { const double2 b = B[o + idx]; re += -ly * b.y; im += ly * b.x; }
        out[o + idx] = make_double2(re, im);
    }
}

__global__ void k_real_axpby_mul(const double *a, const double *b, double *out, size_t n, double alpha,
                                 const double *c, double beta) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double v = alpha * a[i] * (b ? b[i] : 1.0);
        if (c) v += beta * c[i];
        out[i] = v;
    }
}
}  // namespace qgx

using namespace qgx;

extern "C" int qgx_rfft2(qgx_model *m, const double *x_dev, double *xh_dev, void *stream) {
    QGX_REQUIRE(m && x_dev && xh_dev, "qgx_rfft2: null argument");
    return m->small ? small_q_to_qh(m->d, m->opts, x_dev, (double2 *)xh_dev, (hipStream_t)stream)
                    : large_q_to_qh(m, x_dev, (double2 *)xh_dev, (hipStream_t)stream);
}

extern "C" int qgx_irfft2(qgx_model *m, const double *xh_dev, double *x_dev, void *stream) {
    QGX_REQUIRE(m && x_dev && xh_dev, "qgx_irfft2: null argument");
    return m->small ? small_qh_to_q(m->d, m->opts, (const double2 *)xh_dev, x_dev, (hipStream_t)stream)
                    : large_qh_to_q(m, (const double2 *)xh_dev, x_dev, (hipStream_t)stream);
}

extern "C" int qgx_spec_regrid(const double *src_dev, double *dst_dev, int nfields, int n, int N, double scale,
                               int zero_src_2h, int zero_dst_2h, const double *filter_dev, void *stream) {
    QGX_REQUIRE(src_dev && dst_dev && nfields > 0 && n >= 2 && N >= 2 && n % 2 == 0 && N % 2 == 0,
                "qgx_spec_regrid: bad argument (n=%d, N=%d must be even)", n, N);
    const int tot = N * (N / 2 + 1);
    dim3 grid((tot + 255) / 256 > 1024 ? 1024 : (tot + 255) / 256, nfields);
    hipLaunchKernelGGL(k_spec_regrid, grid, dim3(256), 0, (hipStream_t)stream, (const double2 *)src_dev,
                       (double2 *)dst_dev, n, N, scale, zero_src_2h, zero_dst_2h, filter_dev);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

extern "C" int qgx_spec_div(const double *ah_dev, const double *bh_dev, double *out_dev, int nfields, int N,
                            double L, void *stream) {
    QGX_REQUIRE(out_dev && (ah_dev || bh_dev) && nfields > 0 && N >= 2, "qgx_spec_div: bad argument");
    const int tot = N * (N / 2 + 1);
    dim3 grid((tot + 255) / 256 > 1024 ? 1024 : (tot + 255) / 256, nfields);
    hipLaunchKernelGGL(k_spec_div, grid, dim3(256), 0, (hipStream_t)stream, (const double2 *)ah_dev,
                       (const double2 *)bh_dev, (double2 *)out_dev, N, 2. * 3.14159265358979323846 / L);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

extern "C" int qgx_real_fma(const double *a_dev, const double *b_dev, double *out_dev, size_t n, double alpha,
                            const double *c_dev, double beta, void *stream) {
    QGX_REQUIRE(a_dev && out_dev && n > 0, "qgx_real_fma: bad argument");
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_real_axpby_mul, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0,
                       (hipStream_t)stream, a_dev, b_dev, out_dev, n, alpha, c_dev, beta);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}
