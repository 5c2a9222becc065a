// Multi-pass spectral path for grids whose member field does not fit one CU's LDS (N >= 128):
// the same packed-pair algorithm as spectral_small.hip, with the complex N x N work fields in
// global memory (L2 / Infinity-Cache resident for moderate B) and every 2-D FFT executed as a
// row-line kernel plus a column-line kernel that stage lines through LDS.
// Global fields are always in natural order; the digit-reversal of the in-place DIF/DIT passes
// is absorbed when a line is staged into / out of LDS.
//
// Restates pyqg 0.7.2 kernel.pyx::{_invert,_do_advection,_do_friction,
// _do_q_subgrid_parameterization,_forward_timestep} (reference call sites:
// pyqg_generative/tools/simulate.py:83-88 — the 256^2 forcing-dataset runs).
#include "common.hpp"
#include "fft_lds.hpp"

namespace qgx {

extern __shared__ __attribute__((aligned(16))) char lg_smem[];

__device__ __forceinline__ int neg_mod_l(int j, int N) { return j == 0 ? 0 : N - j; }

// ---- batched 1-D FFT along x (ALONG_Y = false) or y (true) of `nf` complex N x N fields -------
// field f lives at base + f * fstride (double2 units).  One workgroup transforms LPB lines.
template <bool FWD, bool ALONG_Y>
__global__ void k_lines_fft(SpecDev d, double2 *base, size_t fstride, int LPB) {
    double2 *L = reinterpret_cast<double2 *>(lg_smem);
    const int N = d.N, LD = N + 1;
    int *pos = reinterpret_cast<int *>(L + (size_t)LPB * LD);
    for (int t = threadIdx.x; t < N; t += blockDim.x) pos[t] = d.pos[t];
    const int groups = N / LPB;
    const int f = blockIdx.x / groups;
    const int l0 = (blockIdx.x - f * groups) * LPB;
    double2 *g = base + (size_t)f * fstride;
    __syncthreads();
    // stage in: element e of line l -> L[l*LD + (FWD ? e : pos[e])]
    for (int t = threadIdx.x; t < LPB * N; t += blockDim.x) {
        int l, e;
        size_t go;
        if (ALONG_Y) { e = t / LPB; l = t - e * LPB; go = (size_t)e * N + l0 + l; }
        else { l = t / N; e = t - l * N; go = (size_t)(l0 + l) * N + e; }
        L[l * LD + (FWD ? e : pos[e])] = g[go];
    }
    __syncthreads();
    if (FWD) fft_lines_fwd(L, LPB, LD, 1, N, d.nrad, d.rad, d.tw);
    else fft_lines_inv(L, LPB, LD, 1, N, d.nrad, d.rad, d.tw);
    for (int t = threadIdx.x; t < LPB * N; t += blockDim.x) {
        int l, e;
        size_t go;
        if (ALONG_Y) { e = t / LPB; l = t - e * LPB; go = (size_t)e * N + l0 + l; }
        else { l = t / N; e = t - l * N; go = (size_t)(l0 + l) * N + e; }
        g[go] = L[l * LD + (FWD ? pos[e] : e)];
    }
}

// ---- pointwise kernels (grid: x over elements, y over member) --------------------------------
__device__ __forceinline__ double2 invert_l(const SpecDev &d, int k, int idx, double2 q0, double2 q1) {
    const int sz = d.N * d.NK;
    const double a0 = d.a[(2 * k) * sz + idx], a1 = d.a[(2 * k + 1) * sz + idx];
    return make_double2(a0 * q0.x + a1 * q1.x, a0 * q0.y + a1 * q1.y);
}

__device__ __forceinline__ void pack_store_l(double2 *Z, int N, int j, int i, double2 Ah, double2 Bh, double s) {
    Z[(size_t)j * N + i] = make_double2((Ah.x - Bh.y) * s, (Ah.y + Bh.x) * s);
    if (i != 0 && 2 * i != N)
        Z[(size_t)neg_mod_l(j, N) * N + (N - i)] = make_double2((Ah.x + Bh.y) * s, (Bh.x - Ah.y) * s);
}

// zbuf[b][k] <- spectrum of (u_k + i v_k) / N^2 for both layers; optional ph store
__global__ void k_l_build_uv(SpecDev d, const double2 *qh, double2 *zbuf, double2 *ph_out) {
    const int N = d.N, NK = d.NK, sz = N * NK, b = blockIdx.y;
    const double2 *qh0 = qh + (size_t)b * 2 * sz, *qh1 = qh0 + sz;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < sz; idx += gridDim.x * blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        const double2 q0 = qh0[idx], q1 = qh1[idx];
        const double kx = d.kk[i], ly = d.ll[j];
        const bool selfc = (i == 0 || 2 * i == N);
        const int jm = neg_mod_l(j, N), idm = jm * NK + i;
        double2 q0m, q1m;
        if (selfc) { q0m = qh0[idm]; q1m = qh1[idm]; }
        for (int k = 0; k < 2; ++k) {
            const double2 ph = invert_l(d, k, idx, q0, q1);
            if (ph_out) ph_out[(size_t)b * 2 * sz + k * sz + idx] = ph;
            double2 uh = make_double2(ly * ph.y, -ly * ph.x);
            double2 vh = make_double2(-kx * ph.y, kx * ph.x);
            if (selfc) {
                const double2 pm = invert_l(d, k, idm, q0m, q1m);
                const double lm = d.ll[jm];
                const double2 um = make_double2(lm * pm.y, -lm * pm.x);
                const double2 vm = make_double2(-kx * pm.y, kx * pm.x);
                uh = make_double2(0.5 * (uh.x + um.x), 0.5 * (uh.y - um.y));
                vh = make_double2(0.5 * (vh.x + vm.x), 0.5 * (vh.y - vm.y));
            }
            pack_store_l(zbuf + ((size_t)b * 2 + k) * N * N, N, j, i, uh, vh, d.invN2);
        }
    }
}

// zbuf[b][0] <- spectrum of (A + i B) / N^2 from two half spectra (A = src[b][0], B = src[b][1])
__global__ void k_l_build_pair(SpecDev d, const double2 *src, double2 *zbuf) {
    const int N = d.N, NK = d.NK, sz = N * NK, b = blockIdx.y;
    const double2 *Ah = src + (size_t)b * 2 * sz, *Bh = Ah + sz;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < sz; idx += gridDim.x * blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        double2 a = Ah[idx], bb = Bh[idx];
        if (i == 0 || 2 * i == N) {
            const int idm = neg_mod_l(j, N) * NK + i;
            const double2 am = Ah[idm], bm = Bh[idm];
            a = make_double2(0.5 * (a.x + am.x), 0.5 * (a.y - am.y));
            bb = make_double2(0.5 * (bb.x + bm.x), 0.5 * (bb.y - bm.y));
        }
        pack_store_l(zbuf + (size_t)b * 2 * N * N, N, j, i, a, bb, d.invN2);
    }
}

// zbuf[b][0] <- (w*r[b][0]) + i (w*r[b][1])
__global__ void k_l_pack_real(SpecDev d, const double *r, double2 *zbuf, double w) {
    const int rz = d.N * d.N, b = blockIdx.y;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < rz; idx += gridDim.x * blockDim.x)
        zbuf[(size_t)b * 2 * rz + idx] = make_double2(w * r[(size_t)b * 2 * rz + idx], w * r[(size_t)b * 2 * rz + rz + idx]);
}

// r[b][0], r[b][1] <- Re, Im of zbuf[b][0]
__global__ void k_l_unpack_real(SpecDev d, const double2 *zbuf, double *r) {
    const int rz = d.N * d.N, b = blockIdx.y;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < rz; idx += gridDim.x * blockDim.x) {
        const double2 w = zbuf[(size_t)b * 2 * rz + idx];
        r[(size_t)b * 2 * rz + idx] = w.x;
        r[(size_t)b * 2 * rz + rz + idx] = w.y;
    }
}

// dst[b][0], dst[b][1] <- half spectra of the two real fields packed in zbuf[b][0] (after fwd FFT)
__global__ void k_l_unpack_pair(SpecDev d, const double2 *zbuf, double2 *dst, int zero_mean) {
    const int N = d.N, NK = d.NK, sz = N * NK, b = blockIdx.y;
    const double2 *Z = zbuf + (size_t)b * 2 * N * N;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < sz; idx += gridDim.x * blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        const double2 a = Z[(size_t)j * N + i];
        const double2 c = Z[(size_t)neg_mod_l(j, N) * N + neg_mod_l(i, N)];
        double2 s0 = make_double2(0.5 * (a.x + c.x), 0.5 * (a.y - c.y));
        double2 s1 = make_double2(0.5 * (a.y + c.y), -0.5 * (a.x - c.x));
        if (zero_mean && idx == 0) { s0 = make_double2(0., 0.); s1 = s0; }
        dst[(size_t)b * 2 * sz + idx] = s0;
        dst[(size_t)b * 2 * sz + sz + idx] = s1;
    }
}

// zbuf[b][k] (u + i v) -> ((u+U_k) q, v q); optional u, v store
__global__ void k_l_products(SpecDev d, double2 *zbuf, const double *q, double *u, double *v) {
    const int rz = d.N * d.N, b = blockIdx.y;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 2 * rz; idx += gridDim.x * blockDim.x) {
        const int k = idx / rz;
        const size_t o = (size_t)b * 2 * rz + idx;
        const double2 uv = zbuf[o];
        if (u) { u[o] = uv.x; v[o] = uv.y; }
        const double qv = q[o];
        zbuf[o] = make_double2((uv.x + d.U[k]) * qv, uv.y * qv);
    }
}

// spectral tendency + friction + forcing + AB3/filter for both layers from zbuf[b][k] (after fwd FFT)
__global__ void k_l_tendency(SpecDev d, StepArgs a, const double2 *zbuf) {
    const int N = d.N, NK = d.NK, sz = N * NK, b = blockIdx.y;
    const double2 *qh0 = a.qh_in + (size_t)b * 2 * sz, *qh1 = qh0 + sz;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < sz; idx += gridDim.x * blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        const int jm = neg_mod_l(j, N), im = neg_mod_l(i, N);
        const double2 q0 = qh0[idx], q1 = qh1[idx];
        const double kx = d.kk[i], ly = d.ll[j];
        for (int k = 0; k < 2; ++k) {
            const double2 *Z = zbuf + ((size_t)b * 2 + k) * N * N;
            const double2 A = Z[(size_t)j * N + i], C = Z[(size_t)jm * N + im];
            const double2 uqh = make_double2(0.5 * (A.x + C.x), 0.5 * (A.y - C.y));
            const double2 vqh = make_double2(0.5 * (A.y + C.y), -0.5 * (A.x - C.x));
            const double2 ph = invert_l(d, k, idx, q0, q1);
            const double kq = kx * d.Qy[k];
            double tx = (kx * uqh.y + ly * vqh.y + kq * ph.y);
            double ty = -(kx * uqh.x + ly * vqh.x + kq * ph.x);
            if (k == 1 && d.rek != 0.0) {
                const double f = d.rek * d.wv2[idx];
                tx += f * ph.x;
                ty += f * ph.y;
            }
            const size_t o = (size_t)b * 2 * sz + k * sz + idx;
            if (a.has_S) { const double2 s = a.dqh[o]; tx += s.x; ty += s.y; }
            const double2 p = a.dq_p[o], pp = a.dq_pp[o];
            const double2 qk = k == 0 ? q0 : q1;
            const double f = d.filtr[idx];
            a.dq_new[o] = make_double2(tx, ty);
            a.qh_out[o] = make_double2(f * (qk.x + a.dt1 * tx + a.dt2 * p.x + a.dt3 * pp.x),
                                       f * (qk.y + a.dt1 * ty + a.dt2 * p.y + a.dt3 * pp.y));
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------
static int lines_per_block(int N) {
    int lpb = 64;
    while (lpb > 1 && ((size_t)lpb * (N + 1) * 16 + (size_t)N * 4 > 72 * 1024 || N % lpb)) lpb /= 2;
    return lpb;
}
static size_t lines_lds(int N, int lpb) { return (((size_t)lpb * (N + 1) * 16 + (size_t)N * 4) + 15) & ~(size_t)15; }

int large_prepare(const SpecDev &d) {
    const int bytes = (int)lines_lds(d.N, lines_per_block(d.N));
    QGX_HIP(hipFuncSetAttribute((const void *)k_lines_fft<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QGX_HIP(hipFuncSetAttribute((const void *)k_lines_fft<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QGX_HIP(hipFuncSetAttribute((const void *)k_lines_fft<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QGX_HIP(hipFuncSetAttribute((const void *)k_lines_fft<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return QGX_OK;
}

// 2-D FFT of nf fields starting at base with stride fstride
template <bool FWD>
static int fft2d_large(const SpecDev &d, double2 *base, size_t fstride, int nf, hipStream_t st) {
    const int lpb = lines_per_block(d.N);
    const size_t lds = lines_lds(d.N, lpb);
    dim3 grid(nf * (d.N / lpb)), block(256);
    if (FWD) {
        hipLaunchKernelGGL((k_lines_fft<true, false>), grid, block, lds, st, d, base, fstride, lpb);
        hipLaunchKernelGGL((k_lines_fft<true, true>), grid, block, lds, st, d, base, fstride, lpb);
    } else {
        hipLaunchKernelGGL((k_lines_fft<false, true>), grid, block, lds, st, d, base, fstride, lpb);
        hipLaunchKernelGGL((k_lines_fft<false, false>), grid, block, lds, st, d, base, fstride, lpb);
    }
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

static dim3 pw_grid(const SpecDev &d, int n) { return dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), d.B); }

int large_q_to_qh(qgx_model *m, const double *q, double2 *qh, hipStream_t st) {
    const SpecDev &d = m->d;
    const size_t f2 = (size_t)2 * d.N * d.N;
    hipLaunchKernelGGL(k_l_pack_real, pw_grid(d, d.N * d.N), dim3(256), 0, st, d, q, m->zbuf, 1.0);
    int rc = fft2d_large<true>(d, m->zbuf, f2, d.B, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_l_unpack_pair, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, m->zbuf, qh, 0);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

int large_qh_to_q(qgx_model *m, const double2 *qh, double *q, hipStream_t st) {
    const SpecDev &d = m->d;
    const size_t f2 = (size_t)2 * d.N * d.N;
    hipLaunchKernelGGL(k_l_build_pair, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, qh, m->zbuf);
    int rc = fft2d_large<false>(d, m->zbuf, f2, d.B, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_l_unpack_real, pw_grid(d, d.N * d.N), dim3(256), 0, st, d, m->zbuf, q);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

int large_invert(qgx_model *m, hipStream_t st) {
    const SpecDev &d = m->d;
    const size_t f1 = (size_t)d.N * d.N;
    hipLaunchKernelGGL(k_l_build_uv, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, m->qh[m->cur_q], m->zbuf, m->ph);
    int rc = fft2d_large<false>(d, m->zbuf, f1, 2 * d.B, st);
    if (rc) return rc;
    // the product kernel doubles as the (u, v) unpacker; what it leaves in the scratch zbuf is unused
    hipLaunchKernelGGL(k_l_products, pw_grid(d, 2 * d.N * d.N), dim3(256), 0, st, d, m->zbuf, m->q, m->u, m->v);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

int large_step(qgx_model *m, const StepArgs &a, hipStream_t st) {
    const SpecDev &d = m->d;
    const size_t f1 = (size_t)d.N * d.N, f2 = 2 * f1;
    int rc;
    if (a.has_S) {
        hipLaunchKernelGGL(k_l_pack_real, pw_grid(d, d.N * d.N), dim3(256), 0, st, d, a.S, m->zbuf, a.weight);
        if ((rc = fft2d_large<true>(d, m->zbuf, f2, d.B, st))) return rc;
        hipLaunchKernelGGL(k_l_unpack_pair, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, m->zbuf, a.dqh, a.demean);
    }
    hipLaunchKernelGGL(k_l_build_uv, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, a.qh_in, m->zbuf,
                       a.diag ? a.ph : (double2 *)nullptr);
    if ((rc = fft2d_large<false>(d, m->zbuf, f1, 2 * d.B, st))) return rc;
    hipLaunchKernelGGL(k_l_products, pw_grid(d, 2 * d.N * d.N), dim3(256), 0, st, d, m->zbuf, a.q,
                       a.diag ? a.u : (double *)nullptr, a.diag ? a.v : (double *)nullptr);
    if ((rc = fft2d_large<true>(d, m->zbuf, f1, 2 * d.B, st))) return rc;
    hipLaunchKernelGGL(k_l_tendency, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, a, (const double2 *)m->zbuf);
    hipLaunchKernelGGL(k_l_build_pair, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, (const double2 *)a.qh_out, m->zbuf);
    if ((rc = fft2d_large<false>(d, m->zbuf, f2, d.B, st))) return rc;
    hipLaunchKernelGGL(k_l_unpack_real, pw_grid(d, d.N * d.N), dim3(256), 0, st, d, m->zbuf, a.q);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

}  // namespace qgx
