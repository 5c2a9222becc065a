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
#include "diag_acc.hpp"
#include <cstdlib>

namespace qgx {

constexpr int ZF = 3;   // complex N x N work fields per member in zbuf: layers 0,1 and one scratch (S / q pair)

extern __shared__ __attribute__((aligned(16))) char lg_smem[];

__device__ __forceinline__ int neg_mod_l(int j, int N) { return j == 0 ? 0 : N - j; }

// ---- batched 1-D FFT along x (ALONG_Y = false) or y (true) of `nf` complex N x N fields -------
// The fields are zbuf[b][k0 .. k0+nfpm) for every member b.  One workgroup transforms LPB lines.
template <bool FWD, bool ALONG_Y>
__global__ void k_lines_fft(SpecDev d, double2 *base, int k0, int nfpm, int LPB) {
    double2 *L = reinterpret_cast<double2 *>(lg_smem);
    const int N = d.N, LD = N + 1;
    int *pos = reinterpret_cast<int *>(L + (size_t)LPB * LD);
    double2 *twl = reinterpret_cast<double2 *>(pos + ((N + 3) & ~3));      // twiddle table in LDS, see lines_lds()
    for (int t = threadIdx.x; t < N; t += blockDim.x) { pos[t] = d.pos[t]; twl[t] = d.tw[t]; }
    const int groups = N / LPB;
    const int f = blockIdx.x / groups;
    const int l0 = (blockIdx.x - f * groups) * LPB;
    double2 *g = base + ((size_t)(f / nfpm) * ZF + k0 + f % nfpm) * N * N;
    __syncthreads();
    // stage in: element e of line l -> L[l*LD + (FWD ? e : pos[e])]
#pragma unroll 4
    for (int t = threadIdx.x; t < LPB * N; t += blockDim.x) {
        int l, e;
        size_t go;
        if (ALONG_Y) { e = t / LPB; l = t - e * LPB; go = (size_t)e * N + l0 + l; }
        else { l = t / N; e = t - l * N; go = (size_t)(l0 + l) * N + e; }
        L[l * LD + (FWD ? e : pos[e])] = g[go];
    }
    __syncthreads();
    if (FWD) fft_lines_fwd(L, LPB, LD, 1, N, d.nrad, d.rad, twl);
    else fft_lines_inv(L, LPB, LD, 1, N, d.nrad, d.rad, twl);
#pragma unroll 4
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
            pack_store_l(zbuf + ((size_t)b * ZF + k) * N * N, N, j, i, uh, vh, d.invN2);
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
        pack_store_l(zbuf + (size_t)b * ZF * N * N, N, j, i, a, bb, d.invN2);
    }
}

// zbuf[b][0] <- (w*r[b][0]) + i (w*r[b][1])
__global__ void k_l_pack_real(SpecDev d, const double *r, double2 *zbuf, double w) {
    const int rz = d.N * d.N, b = blockIdx.y;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < rz; idx += gridDim.x * blockDim.x)
        zbuf[(size_t)b * ZF * rz + idx] = make_double2(w * r[(size_t)b * 2 * rz + idx], w * r[(size_t)b * 2 * rz + rz + idx]);
}

// r[b][0], r[b][1] <- Re, Im of zbuf[b][0]
__global__ void k_l_unpack_real(SpecDev d, const double2 *zbuf, double *r) {
    const int rz = d.N * d.N, b = blockIdx.y;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < rz; idx += gridDim.x * blockDim.x) {
        const double2 w = zbuf[(size_t)b * ZF * rz + idx];
        r[(size_t)b * 2 * rz + idx] = w.x;
        r[(size_t)b * 2 * rz + rz + idx] = w.y;
    }
}

// dst[b][0], dst[b][1] <- half spectra of the two real fields packed in zbuf[b][0] (after fwd FFT)
__global__ void k_l_unpack_pair(SpecDev d, const double2 *zbuf, double2 *dst, int zero_mean, int field) {
    const int N = d.N, NK = d.NK, sz = N * NK, b = blockIdx.y;
    const double2 *Z = zbuf + ((size_t)b * ZF + field) * N * N;
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
        const size_t o = (size_t)b * 2 * rz + idx, oz = (size_t)b * ZF * rz + idx;
        const double2 uv = zbuf[oz];
        if (u) { u[o] = uv.x; v[o] = uv.y; }
        const double qv = q[o];
        zbuf[oz] = make_double2((uv.x + d.U[k]) * qv, uv.y * qv);
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
            const double2 *Z = zbuf + ((size_t)b * ZF + k) * N * N;
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

// ================================================================================================
// Fused step kernels: every pointwise phase rides in the row or column FFT kernel next to it, so one
// step is 5 launches (7 with a forcing) and ~19 MB of traffic per 256x256 member instead of 18 / ~39.
// Row kernels own MIRROR PAIRS of rows (j, N-j) [and (0, N/2)]: the Hermitian extension of a packed
// spectrum needs row -j to build row j, and unpacking row j needs the transformed row -j.
// ================================================================================================
__device__ __forceinline__ int pair_row(int p, int which, int N) {      // p in [0, N/2): rows of pair p
    if (p == 0) return which ? N / 2 : 0;
    return which ? N - p : p;
}

// MODE 0: lines = (row, layer k): spectrum of (u_k + i v_k)/N^2 built from qh, inverse FFT along x -> zbuf[b][k]
// MODE 1: lines = rows of the pair (A,B) = (src[b][0], src[b][1]) -> zbuf[b][2]
// MODE 3: MODE 0 and MODE 1 of the same qh in one pass (three lines per row): the unparameterized step derives
//         q = irfft2(qh) beside (u, v) instead of keeping a real-space copy of q between steps
// NN > 0: grid size, pairs per workgroup (PPWT) and thread count (NT) are compile-time constants: the index arithmetic
// folds, the FFT plan is unrolled and — above all — the staging loops unroll, so that all of a thread's global loads
// are in flight together (with run-time trip counts each thread waits for one 16-byte load at a time: 60-70 % of the
// wave cycles were waits).
template <int MODE, int NN = 0, int PPWT = 0, int NT = 0>
__global__ __launch_bounds__(NT > 0 ? NT : 1024) void k_l_rows_build_inv(SpecDev d, const double2 *src, double2 *zbuf,
                                                                         double2 *ph_out, int PPW_, int ZP) {
    constexpr int NF = MODE == 0 ? 2 : (MODE == 3 ? 3 : 1);
    constexpr bool CT = NN > 0;
    double2 *L = reinterpret_cast<double2 *>(lg_smem);
    const int N = CT ? NN : d.N, NK = N / 2 + 1, LD = N + 1, sz = N * NK;
    const int PPW = CT ? PPWT : PPW_;
    const int nthr = CT ? NT : (int)blockDim.x;
    const int nlines = PPW * 2 * NF;
    int *pos = reinterpret_cast<int *>(L + (size_t)nlines * LD);
    double2 *twl = reinterpret_cast<double2 *>(pos + ((N + 3) & ~3));      // twiddle table in LDS, see lines_lds()
    for (int t = threadIdx.x; t < N; t += nthr) { pos[t] = d.pos[t]; twl[t] = d.tw[t]; }
    const int groups = (N / 2) / PPW;
    const int b = blockIdx.x / groups, p0 = (blockIdx.x - b * groups) * PPW;
    const double2 *s0 = src + (size_t)b * 2 * sz, *s1 = s0 + sz;
    __syncthreads();
#pragma unroll
    for (int t = threadIdx.x; t < PPW * 2 * NK; t += nthr) {
        const int i = t % NK, r = t / NK;                 // r = local row: pair r>>1, member r&1
        const int j = pair_row(p0 + (r >> 1), r & 1, N), jm = neg_mod_l(j, N);
        const int rm = (p0 + (r >> 1)) == 0 ? r : (r ^ 1);   // local row holding row -j
        const int idx = j * NK + i, idm = jm * NK + i;
        const bool selfc = (i == 0 || 2 * i == N);
        const double2 q0 = s0[idx], q1 = s1[idx];
        double2 q0m = q0, q1m = q1;
        if (selfc) { q0m = s0[idm]; q1m = s1[idm]; }
        if constexpr (MODE == 0 || MODE == 3) {
            const double kx = d.kk[i], ly = d.ll[j], lm = d.ll[jm];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double2 ph = invert_l(d, k, idx, q0, q1);
                if (ph_out) ph_out[(size_t)b * 2 * sz + k * sz + idx] = ph;
                double2 uh = make_double2(ly * ph.y, -ly * ph.x);
                double2 vh = make_double2(-kx * ph.y, kx * ph.x);
                if (selfc) {
                    const double2 pm = invert_l(d, k, idm, q0m, q1m);
                    const double2 um = make_double2(lm * pm.y, -lm * pm.x);
                    const double2 vm = make_double2(-kx * pm.y, kx * pm.x);
                    uh = make_double2(0.5 * (uh.x + um.x), 0.5 * (uh.y - um.y));
                    vh = make_double2(0.5 * (vh.x + vm.x), 0.5 * (vh.y - vm.y));
                }
                L[(r * NF + k) * LD + pos[i]] = make_double2((uh.x - vh.y) * d.invN2, (uh.y + vh.x) * d.invN2);
                if (!selfc) L[(rm * NF + k) * LD + pos[N - i]] = make_double2((uh.x + vh.y) * d.invN2, (vh.x - uh.y) * d.invN2);
            }
        }
        if constexpr (MODE == 1 || MODE == 3) {
            constexpr int KQ = MODE == 3 ? 2 : 0;
            double2 a = q0, bb = q1;
            if (selfc) {
                a = make_double2(0.5 * (a.x + q0m.x), 0.5 * (a.y - q0m.y));
                bb = make_double2(0.5 * (bb.x + q1m.x), 0.5 * (bb.y - q1m.y));
            }
            L[(r * NF + KQ) * LD + pos[i]] = make_double2((a.x - bb.y) * d.invN2, (a.y + bb.x) * d.invN2);
            if (!selfc) L[(rm * NF + KQ) * LD + pos[N - i]] = make_double2((a.x + bb.y) * d.invN2, (bb.x - a.y) * d.invN2);
        }
    }
    __syncthreads();
    if constexpr (CT) fft_lines_inv_t<NN, NN>(L, nlines, LD, 1, twl);
    else fft_lines_inv(L, nlines, LD, 1, N, d.nrad, d.rad, twl);
#pragma unroll
    for (int t = threadIdx.x; t < nlines * N; t += nthr) {
        const int e = t % N, line = t / N;
        const int r = line / NF, k = line - r * NF;
        const int j = pair_row(p0 + (r >> 1), r & 1, N);
        zbuf[((size_t)b * ZF + (MODE == 1 ? 2 : k)) * ZP * N + (size_t)j * ZP + e] = L[line * LD + e];
    }
}

// rows of (w*S_1 + i w*S_2) -> forward FFT along x -> zbuf[b][2]
__global__ void k_l_rows_S(SpecDev d, const double *S, double2 *zbuf, double w, int LPB) {
    double2 *L = reinterpret_cast<double2 *>(lg_smem);
    const int N = d.N, LD = N + 1, rz = N * N;
    int *pos = reinterpret_cast<int *>(L + (size_t)LPB * LD);
    double2 *twl = reinterpret_cast<double2 *>(pos + ((N + 3) & ~3));      // twiddle table in LDS, see lines_lds()
    for (int t = threadIdx.x; t < N; t += blockDim.x) { pos[t] = d.pos[t]; twl[t] = d.tw[t]; }
    const int groups = N / LPB;
    const int b = blockIdx.x / groups, j0 = (blockIdx.x - b * groups) * LPB;
    const double *S0 = S + (size_t)b * 2 * rz, *S1 = S0 + rz;
#pragma unroll 4
    for (int t = threadIdx.x; t < LPB * N; t += blockDim.x) {
        const int e = t % N, l = t / N;
        const size_t o = (size_t)(j0 + l) * N + e;
        L[l * LD + e] = make_double2(w * S0[o], w * S1[o]);
    }
    __syncthreads();
    fft_lines_fwd(L, LPB, LD, 1, N, d.nrad, d.rad, twl);
    double2 *g = zbuf + ((size_t)b * ZF + 2) * N * N;
#pragma unroll 4
    for (int t = threadIdx.x; t < LPB * N; t += blockDim.x) {
        const int e = t % N, l = t / N;
        g[(size_t)(j0 + l) * N + e] = L[l * LD + pos[e]];
    }
}

// column tiles.  MODE 0: zbuf[b][k]: inverse along y, (u,v) -> ((u+U_k) q, v q), forward along y
//                MODE 1: zbuf[b][2]: forward along y (forcing pair)
//                MODE 2: zbuf[b][2]: inverse along y, real/imag parts -> q_1, q_2
//                MODE 3: zbuf[b][k]: inverse along y, (u_k, v_k) stored — the inversion alone (large_invert)
template <int MODE>
__global__ void k_l_cols(SpecDev d, double2 *zbuf, double *q, double *u, double *v, int CPB) {
    double2 *L = reinterpret_cast<double2 *>(lg_smem);
    const int N = d.N, LD = N + 1, rz = N * N;
    int *pos = reinterpret_cast<int *>(L + (size_t)CPB * LD);
    double2 *twl = reinterpret_cast<double2 *>(pos + ((N + 3) & ~3));      // twiddle table in LDS, see lines_lds()
    for (int t = threadIdx.x; t < N; t += blockDim.x) { pos[t] = d.pos[t]; twl[t] = d.tw[t]; }
    const int groups = N / CPB;
    const int nf = (MODE == 0 || MODE == 3) ? 2 : 1;
    const int f = blockIdx.x / groups, c0 = (blockIdx.x - f * groups) * CPB;
    const int b = f / nf, k = (MODE == 0 || MODE == 3) ? f - b * nf : 2;
    double2 *g = zbuf + ((size_t)b * ZF + k) * N * N;
    __syncthreads();
#pragma unroll 4
    for (int t = threadIdx.x; t < CPB * N; t += blockDim.x) {
        const int r = t / CPB, c = t - r * CPB;
        L[c * LD + (MODE == 1 ? r : pos[r])] = g[(size_t)r * N + c0 + c];
    }
    __syncthreads();
    if constexpr (MODE == 1) {
        fft_lines_fwd(L, CPB, LD, 1, N, d.nrad, d.rad, twl);
    } else {
        fft_lines_inv(L, CPB, LD, 1, N, d.nrad, d.rad, twl);
    }
    if constexpr (MODE == 3) {
        const size_t ro = (size_t)b * 2 * rz + (size_t)k * rz;
#pragma unroll 4
        for (int t = threadIdx.x; t < CPB * N; t += blockDim.x) {
            const int r = t / CPB, c = t - r * CPB;
            const double2 uv = L[c * LD + r];
            const size_t o = ro + (size_t)r * N + c0 + c;
            u[o] = uv.x; v[o] = uv.y;
        }
        return;
    }
    if constexpr (MODE == 0) {
        const double Uk = d.U[k];
        const size_t ro = (size_t)b * 2 * rz + (size_t)k * rz;
#pragma unroll 4
        for (int t = threadIdx.x; t < CPB * N; t += blockDim.x) {
            const int r = t / CPB, c = t - r * CPB;
            const double2 uv = L[c * LD + r];
            const size_t o = ro + (size_t)r * N + c0 + c;
            if (u) { u[o] = uv.x; v[o] = uv.y; }
            const double qv = q[o];
            L[c * LD + r] = make_double2((uv.x + Uk) * qv, uv.y * qv);
        }
        __syncthreads();
        fft_lines_fwd(L, CPB, LD, 1, N, d.nrad, d.rad, twl);
    }
    if constexpr (MODE == 2) {
        double *q0 = q + (size_t)b * 2 * rz, *q1 = q0 + rz;
#pragma unroll 4
        for (int t = threadIdx.x; t < CPB * N; t += blockDim.x) {
            const int r = t / CPB, c = t - r * CPB;
            const double2 w = L[c * LD + r];
            q0[(size_t)r * N + c0 + c] = w.x;
            q1[(size_t)r * N + c0 + c] = w.y;
        }
    } else {
#pragma unroll 4
        for (int t = threadIdx.x; t < CPB * N; t += blockDim.x) {
            const int r = t / CPB, c = t - r * CPB;
            g[(size_t)r * N + c0 + c] = L[c * LD + pos[r]];
        }
    }
}

// column tiles of all three work fields of a member (after k_l_rows_build_inv<3>): inverse along y of
// (u_1 + i v_1), (u_2 + i v_2) and (q_1 + i q_2); the advection products with q taken from LDS; forward along y of
// the two product fields -> zbuf[b][0..1].  No real-space q is read or written.
template <int NN = 0, int CPBT = 0, int NT = 0>
__global__ __launch_bounds__(NT > 0 ? NT : 1024) void k_l_cols3(SpecDev d, double2 *zbuf, double *u, double *v, int CPB_, int ZP) {
    constexpr bool CT = NN > 0;
    double2 *L = reinterpret_cast<double2 *>(lg_smem);
    const int N = CT ? NN : d.N, LD = N + 1, rz = N * N;
    const int CPB = CT ? CPBT : CPB_;
    const int nthr = CT ? NT : (int)blockDim.x;
    int *pos = reinterpret_cast<int *>(L + (size_t)3 * CPB * LD);
    double2 *twl = reinterpret_cast<double2 *>(pos + ((N + 3) & ~3));      // twiddle table in LDS, see lines_lds()
    for (int t = threadIdx.x; t < N; t += nthr) { pos[t] = d.pos[t]; twl[t] = d.tw[t]; }
    const int groups = N / CPB;
    const int b = blockIdx.x / groups, c0 = (blockIdx.x - b * groups) * CPB;
    const size_t fz = (size_t)ZP * N;                    // one work field, rows at pitch ZP (see large_step)
    double2 *g = zbuf + (size_t)b * ZF * fz;
    __syncthreads();
#pragma unroll
    for (int t = threadIdx.x; t < 3 * CPB * N; t += nthr) {
        const int k = t / (CPB * N), t2 = t - k * CPB * N;
        const int r = t2 / CPB, c = t2 - r * CPB;
        L[(k * CPB + c) * LD + pos[r]] = g[(size_t)k * fz + (size_t)r * ZP + c0 + c];
    }
    __syncthreads();
    if constexpr (CT) fft_lines_inv_t<NN, NN>(L, 3 * CPB, LD, 1, twl);
    else fft_lines_inv(L, 3 * CPB, LD, 1, N, d.nrad, d.rad, twl);
#pragma unroll
    for (int t = threadIdx.x; t < 2 * CPB * N; t += nthr) {
        const int k = t / (CPB * N), t2 = t - k * CPB * N;
        const int r = t2 / CPB, c = t2 - r * CPB;
        const double2 uv = L[(k * CPB + c) * LD + r];
        const double2 qq = L[(2 * CPB + c) * LD + r];
        const double qv = k == 0 ? qq.x : qq.y;
        if (u) {
            const size_t o = (size_t)b * 2 * rz + (size_t)k * rz + (size_t)r * N + c0 + c;
            u[o] = uv.x; v[o] = uv.y;
        }
        L[(k * CPB + c) * LD + r] = make_double2((uv.x + d.U[k]) * qv, uv.y * qv);
    }
    __syncthreads();
    if constexpr (CT) fft_lines_fwd_t<NN, NN>(L, 2 * CPB, LD, 1, twl);
    else fft_lines_fwd(L, 2 * CPB, LD, 1, N, d.nrad, d.rad, twl);
#pragma unroll
    for (int t = threadIdx.x; t < 2 * CPB * N; t += nthr) {
        const int k = t / (CPB * N), t2 = t - k * CPB * N;
        const int r = t2 / CPB, c = t2 - r * CPB;
        g[(size_t)k * fz + (size_t)r * ZP + c0 + c] = L[(k * CPB + c) * LD + pos[r]];
    }
}

// rows of zbuf[b][0..1] (pairs j, -j): forward FFT along x, unpack (uq_k, vq_k), tendency + friction + forcing,
// AB3 + filter -> dq_new, qh_out.  The forcing spectrum is read from zbuf[b][2] (already transformed).
template <int NN = 0, int PPWT = 0, int NT = 0>
__global__ __launch_bounds__(NT > 0 ? NT : 1024) void k_l_rows_fwd_tend(SpecDev d, StepArgs a, const double2 *zbuf, int PPW_, int ZP) {
    constexpr bool CT = NN > 0;
    double2 *L = reinterpret_cast<double2 *>(lg_smem);
    const int N = CT ? NN : d.N, NK = N / 2 + 1, LD = N + 1, sz = N * NK;
    const int PPW = CT ? PPWT : PPW_;
    const int nthr = CT ? NT : (int)blockDim.x;
    const int nlines = PPW * 4;
    int *pos = reinterpret_cast<int *>(L + (size_t)nlines * LD);
    double2 *twl = reinterpret_cast<double2 *>(pos + ((N + 3) & ~3));      // twiddle table in LDS, see lines_lds()
    for (int t = threadIdx.x; t < N; t += nthr) { pos[t] = d.pos[t]; twl[t] = d.tw[t]; }
    const int groups = (N / 2) / PPW;
    const int b = blockIdx.x / groups, p0 = (blockIdx.x - b * groups) * PPW;
#pragma unroll
    for (int t = threadIdx.x; t < nlines * N; t += nthr) {
        const int e = t % N, line = t / N;
        const int r = line >> 1, k = line & 1;
        const int j = pair_row(p0 + (r >> 1), r & 1, N);
        L[line * LD + e] = zbuf[((size_t)b * ZF + k) * ZP * N + (size_t)j * ZP + e];
    }
    __syncthreads();
    if constexpr (CT) fft_lines_fwd_t<NN, NN>(L, nlines, LD, 1, twl);
    else fft_lines_fwd(L, nlines, LD, 1, N, d.nrad, d.rad, twl);
    const double2 *qh0 = a.qh_in + (size_t)b * 2 * sz, *qh1 = qh0 + sz;
    const double2 *ZS = zbuf + ((size_t)b * ZF + 2) * ZP * N;
#pragma unroll
    for (int t = threadIdx.x; t < PPW * 2 * NK; t += nthr) {
        const int i = t % NK, r = t / NK;
        const int j = pair_row(p0 + (r >> 1), r & 1, N), jm = neg_mod_l(j, N), im = neg_mod_l(i, N);
        const int rm = (p0 + (r >> 1)) == 0 ? r : (r ^ 1);
        const int idx = j * NK + i;
        const double2 q0 = qh0[idx], q1 = qh1[idx];
        const double kx = d.kk[i], ly = d.ll[j];
        double2 s0 = make_double2(0., 0.), s1 = s0;
        if (a.has_S) {
            const double2 A = ZS[(size_t)j * ZP + i], C = ZS[(size_t)jm * ZP + im];
            s0 = make_double2(0.5 * (A.x + C.x), 0.5 * (A.y - C.y));
            s1 = make_double2(0.5 * (A.y + C.y), -0.5 * (A.x - C.x));
            if (a.demean && idx == 0) { s0 = make_double2(0., 0.); s1 = s0; }
            if (a.diag) { a.dqh[(size_t)b * 2 * sz + idx] = s0; a.dqh[(size_t)b * 2 * sz + sz + idx] = s1; }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double2 A = L[(r * 2 + k) * LD + pos[i]], C = L[(rm * 2 + k) * LD + pos[im]];
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
            const double2 s = k == 0 ? s0 : s1;
            tx += s.x; ty += s.y;
            const size_t o = (size_t)b * 2 * sz + k * sz + idx;
            const double2 p = a.dq_p[o], pp = a.dq_pp[o];
            const double2 qk = k == 0 ? q0 : q1;
            const double f = d.filtr[idx];
            a.dq_new[o] = make_double2(tx, ty);
            a.qh_out[o] = make_double2(f * (qk.x + a.dt1 * tx + a.dt2 * p.x + a.dt3 * pp.x),
                                       f * (qk.y + a.dt1 * ty + a.dt2 * p.y + a.dt3 * pp.y));
        }
    }
}

// ================================================================================================
// One increment of the time-averaged diagnostics on the large grids in THREE launches (model.py::_calc_diagnostics).
// Composed from the generic transforms (diag.hip) an increment at 256 x 256 is ~30 launches and 4.9 GB of traffic per
// 64 members: eight packed 2-D transforms, each a row kernel + a column kernel + a pack / unpack sweep, with the
// real-space fields written and re-read in between.  Here the FOUR packed fields an increment needs ride through the
// row / column kernels together, as (u, v, q) do in the unparameterized step:
//   rows:    from qh: spectra of (u_1 + i v_1), (u_2 + i v_2), (q_1 + i q_2), (p_1 + i p_2), inverse along x; psi stored
//   columns: inverse along y, the four product pairs, forward along y
//              (u_1 q_1, v_1 q_1), (u_2 q_2, v_2 q_2), (u_1 d, v_1 d), (u_2 d, v_2 d),  d = p_1 - p_2
//   rows:    forward along x, unpack, accumulate the sixteen diagnostics (diag_acc.hpp)
// The relative vorticity xi_k = irfft2(-K^2 psi_k) is not transformed: q_1 = xi_1 + F_1 (p_2 - p_1),
// q_2 = xi_2 + F_2 (p_1 - p_2), so u_k xi_k = u_k q_k +/- F_k u_k d, and the thickness-weighted products of APEflux are
// del_1 (u_1 d) + del_2 (u_2 d): four forward transforms carry the five product pairs of the composed form (rounding
// differs at 1e-14 relative: cancellation by at most 1 + F / K_min^2 ~ 90).
// ================================================================================================
constexpr int DZF = 4;       // complex work fields per member of the fused increment

template <int NN, int PPWT, int NT>
__global__ __launch_bounds__(NT) void k_l_rows_diag_inv(SpecDev d, const double2 *src, double2 *dz, double2 *ph_out, int ZP) {
    constexpr int NF = DZF, N = NN, NK = N / 2 + 1, LD = N + 1, sz = N * NK, PPW = PPWT, nlines = PPW * 2 * NF;
    double2 *L = reinterpret_cast<double2 *>(lg_smem);
    int *pos = reinterpret_cast<int *>(L + (size_t)nlines * LD);
    double2 *twl = reinterpret_cast<double2 *>(pos + ((N + 3) & ~3));
    for (int t = threadIdx.x; t < N; t += NT) { pos[t] = d.pos[t]; twl[t] = d.tw[t]; }
    constexpr int groups = (N / 2) / PPW;
    const int b = blockIdx.x / groups, p0 = (blockIdx.x - b * groups) * PPW;
    const double2 *s0 = src + (size_t)b * 2 * sz, *s1 = s0 + sz;
    __syncthreads();
#pragma unroll
    for (int t = threadIdx.x; t < PPW * 2 * NK; t += NT) {
        const int i = t % NK, r = t / NK;
        const int j = pair_row(p0 + (r >> 1), r & 1, N), jm = neg_mod_l(j, N);
        const int rm = (p0 + (r >> 1)) == 0 ? r : (r ^ 1);
        const int idx = j * NK + i, idm = jm * NK + i;
        const bool selfc = (i == 0 || 2 * i == N);
        const double2 q0 = s0[idx], q1 = s1[idx];
        double2 q0m = q0, q1m = q1;
        if (selfc) { q0m = s0[idm]; q1m = s1[idm]; }
        const double kx = d.kk[i], ly = d.ll[j], lm = d.ll[jm];
        double2 phk[2], phm[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double2 ph = invert_l(d, k, idx, q0, q1);
            phk[k] = ph; phm[k] = ph;
            ph_out[(size_t)b * 2 * sz + k * sz + idx] = ph;
            double2 uh = make_double2(ly * ph.y, -ly * ph.x);
            double2 vh = make_double2(-kx * ph.y, kx * ph.x);
            if (selfc) {
                const double2 pm = invert_l(d, k, idm, q0m, q1m);
                phm[k] = pm;
                const double2 um = make_double2(lm * pm.y, -lm * pm.x);
                const double2 vm = make_double2(-kx * pm.y, kx * pm.x);
                uh = make_double2(0.5 * (uh.x + um.x), 0.5 * (uh.y - um.y));
                vh = make_double2(0.5 * (vh.x + vm.x), 0.5 * (vh.y - vm.y));
            }
            L[(r * NF + k) * LD + pos[i]] = make_double2((uh.x - vh.y) * d.invN2, (uh.y + vh.x) * d.invN2);
            if (!selfc) L[(rm * NF + k) * LD + pos[N - i]] = make_double2((uh.x + vh.y) * d.invN2, (vh.x - uh.y) * d.invN2);
        }
#pragma unroll
        for (int f = 2; f < 4; ++f) {            // the pairs (q_1 + i q_2) and (p_1 + i p_2)
            double2 a = f == 2 ? q0 : phk[0], bb = f == 2 ? q1 : phk[1];
            if (selfc) {
                const double2 am = f == 2 ? q0m : phm[0], bm = f == 2 ? q1m : phm[1];
                a = make_double2(0.5 * (a.x + am.x), 0.5 * (a.y - am.y));
                bb = make_double2(0.5 * (bb.x + bm.x), 0.5 * (bb.y - bm.y));
            }
            L[(r * NF + f) * LD + pos[i]] = make_double2((a.x - bb.y) * d.invN2, (a.y + bb.x) * d.invN2);
            if (!selfc) L[(rm * NF + f) * LD + pos[N - i]] = make_double2((a.x + bb.y) * d.invN2, (bb.x - a.y) * d.invN2);
        }
    }
    __syncthreads();
    fft_lines_inv_t<NN, NN>(L, nlines, LD, 1, twl);
#pragma unroll
    for (int t = threadIdx.x; t < nlines * N; t += NT) {
        const int e = t % N, line = t / N;
        const int r = line / NF, f = line - r * NF;
        const int j = pair_row(p0 + (r >> 1), r & 1, N);
        dz[((size_t)b * DZF + f) * ZP * N + (size_t)j * ZP + e] = L[line * LD + e];
    }
}

template <int NN, int CPBT, int NT>
__global__ __launch_bounds__(NT) void k_l_cols_diag(SpecDev d, double2 *dz, int ZP) {
    constexpr int N = NN, LD = N + 1, CPB = CPBT;
    double2 *L = reinterpret_cast<double2 *>(lg_smem);
    int *pos = reinterpret_cast<int *>(L + (size_t)DZF * CPB * LD);
    double2 *twl = reinterpret_cast<double2 *>(pos + ((N + 3) & ~3));
    for (int t = threadIdx.x; t < N; t += NT) { pos[t] = d.pos[t]; twl[t] = d.tw[t]; }
    constexpr int groups = N / CPB;
    const int b = blockIdx.x / groups, c0 = (blockIdx.x - b * groups) * CPB;
    const size_t fz = (size_t)ZP * N;
    double2 *g = dz + (size_t)b * DZF * fz;
    __syncthreads();
#pragma unroll
    for (int t = threadIdx.x; t < DZF * CPB * N; t += NT) {
        const int k = t / (CPB * N), t2 = t - k * CPB * N;
        const int r = t2 / CPB, c = t2 - r * CPB;
        L[(k * CPB + c) * LD + pos[r]] = g[(size_t)k * fz + (size_t)r * ZP + c0 + c];
    }
    __syncthreads();
    fft_lines_inv_t<NN, NN>(L, DZF * CPB, LD, 1, twl);
#pragma unroll
    for (int t = threadIdx.x; t < CPB * N; t += NT) {
        const int r = t / CPB, c = t - r * CPB;
        const double2 uv1 = L[(0 * CPB + c) * LD + r], uv2 = L[(1 * CPB + c) * LD + r];
        const double2 qq = L[(2 * CPB + c) * LD + r], pp = L[(3 * CPB + c) * LD + r];
        const double dp = pp.x - pp.y;
        L[(0 * CPB + c) * LD + r] = make_double2(uv1.x * qq.x, uv1.y * qq.x);
        L[(1 * CPB + c) * LD + r] = make_double2(uv2.x * qq.y, uv2.y * qq.y);
        L[(2 * CPB + c) * LD + r] = make_double2(uv1.x * dp, uv1.y * dp);
        L[(3 * CPB + c) * LD + r] = make_double2(uv2.x * dp, uv2.y * dp);
    }
    __syncthreads();
    fft_lines_fwd_t<NN, NN>(L, DZF * CPB, LD, 1, twl);
#pragma unroll
    for (int t = threadIdx.x; t < DZF * CPB * N; t += NT) {
        const int k = t / (CPB * N), t2 = t - k * CPB * N;
        const int r = t2 / CPB, c = t2 - r * CPB;
        g[(size_t)k * fz + (size_t)r * ZP + c0 + c] = L[(k * CPB + c) * LD + pos[r]];
    }
}

template <int NN, int PPWT, int NT>
__global__ __launch_bounds__(NT) void k_l_rows_diag_acc(SpecDev d, DiagConst c, const double2 *dz, const double2 *qh, const double2 *ph,
                                                        const double2 *Sh, const double2 *dq_p, const double2 *dq_pp, DiagAcc acc, int ZP) {
    constexpr int NF = DZF, N = NN, NK = N / 2 + 1, LD = N + 1, sz = N * NK, PPW = PPWT, nlines = PPW * 2 * NF;
    double2 *L = reinterpret_cast<double2 *>(lg_smem);
    int *pos = reinterpret_cast<int *>(L + (size_t)nlines * LD);
    double2 *twl = reinterpret_cast<double2 *>(pos + ((N + 3) & ~3));
    for (int t = threadIdx.x; t < N; t += NT) { pos[t] = d.pos[t]; twl[t] = d.tw[t]; }
    constexpr int groups = (N / 2) / PPW;
    const int b = blockIdx.x / groups, p0 = (blockIdx.x - b * groups) * PPW;
#pragma unroll
    for (int t = threadIdx.x; t < nlines * N; t += NT) {
        const int e = t % N, line = t / N;
        const int r = line / NF, f = line - r * NF;
        const int j = pair_row(p0 + (r >> 1), r & 1, N);
        L[line * LD + e] = dz[((size_t)b * DZF + f) * ZP * N + (size_t)j * ZP + e];
    }
    __syncthreads();
    fft_lines_fwd_t<NN, NN>(L, nlines, LD, 1, twl);
    const double F1 = c.rdm2 * c.del2, F2 = c.rdm2 * c.del1;
#pragma unroll
    for (int t = threadIdx.x; t < PPW * 2 * NK; t += NT) {
        const int i = t % NK, r = t / NK;
        const int j = pair_row(p0 + (r >> 1), r & 1, N), im = neg_mod_l(i, N);
        const int rm = (p0 + (r >> 1)) == 0 ? r : (r ^ 1);
        const int idx = j * NK + i;
        double2 A[NF], Bv[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const double2 X = L[(r * NF + f) * LD + pos[i]], C = L[(rm * NF + f) * LD + pos[im]];
            A[f] = make_double2(0.5 * (X.x + C.x), 0.5 * (X.y - C.y));
            Bv[f] = make_double2(0.5 * (X.y + C.y), -0.5 * (X.x - C.x));
        }
        // (ub d, vb d) = del_1 (u_1 d) + del_2 (u_2 d);  (u_1 xi_1) = (u_1 q_1) + F_1 (u_1 d);  (u_2 xi_2) = (u_2 q_2) - F_2 (u_2 d)
        const double2 A3 = make_double2(c.del1 * A[2].x + c.del2 * A[3].x, c.del1 * A[2].y + c.del2 * A[3].y);
        const double2 B3 = make_double2(c.del1 * Bv[2].x + c.del2 * Bv[3].x, c.del1 * Bv[2].y + c.del2 * Bv[3].y);
        const double2 A4 = make_double2(A[0].x + F1 * A[2].x, A[0].y + F1 * A[2].y), B4 = make_double2(Bv[0].x + F1 * Bv[2].x, Bv[0].y + F1 * Bv[2].y);
        const double2 A5 = make_double2(A[1].x - F2 * A[3].x, A[1].y - F2 * A[3].y), B5 = make_double2(Bv[1].x - F2 * Bv[3].x, Bv[1].y - F2 * Bv[3].y);
        const size_t o = (size_t)b * 2 * sz + idx, o2 = (size_t)b * sz + idx;
        const double2 zero = make_double2(0., 0.);
        diag_accumulate_elem(d, c, acc, idx, i, j, o, o2, sz, qh[o], qh[o + sz], ph[o], ph[o + sz], A3, B3, A4, B4, A5, B5, Sh != nullptr,
                             Sh ? Sh[o] : zero, Sh ? Sh[o + sz] : zero, A[0], Bv[0], A[1], Bv[1], dq_p[o], dq_p[o + sz], dq_pp[o], dq_pp[o + sz]);
    }
}

// ---- host side ---------------------------------------------------------------------------------
static int lines_per_block(int N) {
    static const int forced = tune_env("QGX_LARGE_LPB", 0);
    if (forced > 0 && N % forced == 0) return forced;
    int lpb = 64;
    // <= 36 KB of LDS per workgroup: 4 workgroups per CU hide the global-memory latency of these
    // short kernels better than longer lines do (256x256, B=64: 360 us/step at 8 lines vs 456 at 16)
    while (lpb > 1 && ((size_t)lpb * (N + 1) * 16 + (size_t)N * 20 > 40 * 1024 || N % lpb)) lpb /= 2;
    return lpb;
}
// LDS of a line kernel: the lines (row stride N + 1), the digit-reversal table and the twiddle table (twiddles read
// from global memory put an L2 round trip into every butterfly of every pass: the kernels were 60-70 % wait)
static size_t lines_lds(int N, int lpb) { return (size_t)lpb * (N + 1) * 16 + (size_t)((N + 3) & ~3) * 4 + (size_t)N * 16; }

// tile shapes of the three-field kernels of the unparameterized step: columns per tile of k_l_cols3 (3 lines per
// column) and mirror pairs of rows per workgroup of k_l_rows_build_inv<3> (6 lines per pair)
static int cols3_cpb(int N) {
    static const int forced = tune_env("QGX_LARGE_C3", 0);
    int c = forced > 0 ? forced : 4;
    while (c > 1 && (N % c || lines_lds(N, 3 * c) > 160 * 1024 - 512)) c /= 2;
    return (N % c == 0 && lines_lds(N, 3 * c) <= 160 * 1024 - 512) ? c : 0;
}
static int rows3_ppw(int N) {
    static const int forced = tune_env("QGX_LARGE_P3", 0);
    const int p = forced > 0 ? forced : 1;
    return ((N / 2) % p == 0 && lines_lds(N, 6 * p) <= 160 * 1024 - 512) ? p : 1;
}

// Row pitch of the work fields between the row and the column kernels of the lazy path, in complex elements beyond N.
// A column tile touches N rows at one x offset: with the natural pitch (N * 16 B = 4 KiB at 256) every one of those
// accesses lands on the same few memory channels.
int large_zpad() {
    static const int pad = tune_env("QGX_LARGE_ZPAD", 8);
    return pad < 0 ? 0 : pad;
}

int large_prepare(const SpecDev &d) {
    // the attribute is a per-kernel cap shared by every model of the process (models of different N coexist:
    // hires run + coarse-grained forcing models), so it is set once to the most any grid may ask for
    (void)d;
    const int cap = 160 * 1024 - 512;
    const void *kernels[] = {(const void *)k_lines_fft<true, false>, (const void *)k_lines_fft<true, true>,
                             (const void *)k_lines_fft<false, false>, (const void *)k_lines_fft<false, true>,
                             (const void *)k_l_rows_build_inv<0>, (const void *)k_l_rows_build_inv<1>,
                             (const void *)k_l_rows_build_inv<3>, (const void *)k_l_rows_S, (const void *)k_l_cols<0>,
                             (const void *)k_l_cols<1>, (const void *)k_l_cols<2>, (const void *)k_l_cols<3>, (const void *)k_l_cols3<>,
                             (const void *)k_l_rows_fwd_tend<>,
#define QGX_L3(NN, P1, T1, C2, T2, P3, T3) (const void *)k_l_rows_build_inv<3, NN, P1, T1>, \
                   (const void *)k_l_cols3<NN, C2, T2>, (const void *)k_l_rows_fwd_tend<NN, P3, T3>
                             QGX_L3(128, 2, 512, 8, 1024, 4, 512), QGX_L3(256, 2, 512, 8, 1024, 4, 512),
                             QGX_L3(512, 1, 256, 4, 512, 2, 256),
#ifdef QGX_L3_SWEEP
                             (const void *)k_l_cols3<256, 8, 512>, (const void *)k_l_cols3<256, 8, 1024>, (const void *)k_l_cols3<256, 4, 512>,
                             (const void *)k_l_cols3<256, 4, 128>, (const void *)k_l_cols3<256, 2, 128>, (const void *)k_l_cols3<256, 2, 256>,
                             (const void *)k_l_rows_build_inv<3, 256, 2, 512>, (const void *)k_l_rows_build_inv<3, 256, 2, 256>,
                             (const void *)k_l_rows_build_inv<3, 256, 1, 128>, (const void *)k_l_rows_fwd_tend<256, 4, 512>,
                             (const void *)k_l_rows_fwd_tend<256, 4, 256>, (const void *)k_l_rows_fwd_tend<256, 2, 128>,
                             (const void *)k_l_rows_fwd_tend<256, 1, 128>, (const void *)k_l_rows_fwd_tend<256, 1, 256>,
#endif
                             };
#undef QGX_L3
    for (const void *f : kernels) QGX_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    return QGX_OK;
}

// 2-D FFT of the work fields [k0, k0+nfpm) of every member
template <bool FWD>
static int fft2d_large(const SpecDev &d, double2 *base, int k0, int nfpm, hipStream_t st) {
    const int lpb = lines_per_block(d.N);
    const size_t lds = lines_lds(d.N, lpb);
    dim3 grid(d.B * nfpm * (d.N / lpb)), block(256);
    if (FWD) {
        hipLaunchKernelGGL((k_lines_fft<true, false>), grid, block, lds, st, d, base, k0, nfpm, lpb);
        hipLaunchKernelGGL((k_lines_fft<true, true>), grid, block, lds, st, d, base, k0, nfpm, lpb);
    } else {
        hipLaunchKernelGGL((k_lines_fft<false, true>), grid, block, lds, st, d, base, k0, nfpm, lpb);
        hipLaunchKernelGGL((k_lines_fft<false, false>), grid, block, lds, st, d, base, k0, nfpm, lpb);
    }
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

static dim3 pw_grid(const SpecDev &d, int n) { return dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), d.B); }

// rfft2 / irfft2 of (B,2,N,N) fields.  Fused form: the real -> complex packing rides in the row kernel's staging and the
// complex -> real split in the column kernel's store (3 / 2 launches and no separate pack / unpack sweep of the work field
// instead of 4 launches); the unfused form remains for tile shapes the fused kernels do not cover.
static bool fused_transforms_ok(int N, int lpb) { return lpb >= 2 && lpb % 2 == 0 && (N / 2) % (lpb / 2) == 0 && N % lpb == 0; }

int large_q_to_qh(qgx_model *m, const double *q, double2 *qh, hipStream_t st) {
    const SpecDev &d = m->d;
    const int lpb = lines_per_block(d.N);
    const bool unfused = !m->opts.large_fused;
    if (!unfused && fused_transforms_ok(d.N, lpb)) {
        const size_t lds = lines_lds(d.N, lpb);
        hipLaunchKernelGGL(k_l_rows_S, dim3(d.B * (d.N / lpb)), dim3(256), lds, st, d, q, m->zbuf, 1.0, lpb);
        hipLaunchKernelGGL(k_l_cols<1>, dim3(d.B * (d.N / lpb)), dim3(256), lds, st, d, m->zbuf, (double *)nullptr,
                           (double *)nullptr, (double *)nullptr, lpb);
        hipLaunchKernelGGL(k_l_unpack_pair, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, m->zbuf, qh, 0, 2);
        QGX_HIP(hipGetLastError());
        return QGX_OK;
    }
    hipLaunchKernelGGL(k_l_pack_real, pw_grid(d, d.N * d.N), dim3(256), 0, st, d, q, m->zbuf, 1.0);
    int rc = fft2d_large<true>(d, m->zbuf, 0, 1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_l_unpack_pair, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, m->zbuf, qh, 0, 0);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

int large_qh_to_q(qgx_model *m, const double2 *qh, double *q, hipStream_t st) {
    const SpecDev &d = m->d;
    const int lpb = lines_per_block(d.N);
    const bool unfused = !m->opts.large_fused;
    if (!unfused && fused_transforms_ok(d.N, lpb)) {
        const size_t lds = lines_lds(d.N, lpb);
        hipLaunchKernelGGL(k_l_rows_build_inv<1>, dim3(d.B * ((d.N / 2) / (lpb / 2))), dim3(256), lds, st, d, qh, m->zbuf,
                           (double2 *)nullptr, lpb / 2, d.N);
        hipLaunchKernelGGL(k_l_cols<2>, dim3(d.B * (d.N / lpb)), dim3(256), lds, st, d, m->zbuf, q, (double *)nullptr,
                           (double *)nullptr, lpb);
        QGX_HIP(hipGetLastError());
        return QGX_OK;
    }
    hipLaunchKernelGGL(k_l_build_pair, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, qh, m->zbuf);
    int rc = fft2d_large<false>(d, m->zbuf, 0, 1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_l_unpack_real, pw_grid(d, d.N * d.N), dim3(256), 0, st, d, m->zbuf, q);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

int large_invert(qgx_model *m, hipStream_t st) {
    const SpecDev &d = m->d;
    {
        // fused form: (u, v) spectra built in the row kernel's staging (with psi stored), split into u, v in the column
        // kernel's store — 2 launches instead of 4
        const int lpb = lines_per_block(d.N);
        const bool unfused = !m->opts.large_fused;
        if (!unfused && lpb >= 4 && lpb % 4 == 0 && (d.N / 2) % (lpb / 4) == 0 && d.N % lpb == 0) {
            const size_t lds = lines_lds(d.N, lpb);
            hipLaunchKernelGGL(k_l_rows_build_inv<0>, dim3(d.B * ((d.N / 2) / (lpb / 4))), dim3(256), lds, st, d,
                               (const double2 *)m->qh[m->cur_q], m->zbuf, m->ph, lpb / 4, d.N);
            hipLaunchKernelGGL(k_l_cols<3>, dim3(d.B * 2 * (d.N / lpb)), dim3(256), lds, st, d, m->zbuf, (double *)nullptr,
                               m->u, m->v, lpb);
            QGX_HIP(hipGetLastError());
            return QGX_OK;
        }
    }
    hipLaunchKernelGGL(k_l_build_uv, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, m->qh[m->cur_q], m->zbuf, m->ph);
    int rc = fft2d_large<false>(d, m->zbuf, 0, 2, st);
    if (rc) return rc;
    // the product kernel doubles as the (u, v) unpacker; what it leaves in the scratch zbuf is unused
    hipLaunchKernelGGL(k_l_products, pw_grid(d, 2 * d.N * d.N), dim3(256), 0, st, d, m->zbuf, m->q, m->u, m->v);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

static int large_step_unfused(qgx_model *m, const StepArgs &a, hipStream_t st) {
    const SpecDev &d = m->d;
    int rc;
    if (a.has_S) {
        hipLaunchKernelGGL(k_l_pack_real, pw_grid(d, d.N * d.N), dim3(256), 0, st, d, a.S, m->zbuf, a.weight);
        if ((rc = fft2d_large<true>(d, m->zbuf, 0, 1, st))) return rc;
        hipLaunchKernelGGL(k_l_unpack_pair, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, m->zbuf, a.dqh, a.demean, 0);
    }
    hipLaunchKernelGGL(k_l_build_uv, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, a.qh_in, m->zbuf,
                       a.diag ? a.ph : (double2 *)nullptr);
    if ((rc = fft2d_large<false>(d, m->zbuf, 0, 2, st))) return rc;
    hipLaunchKernelGGL(k_l_products, pw_grid(d, 2 * d.N * d.N), dim3(256), 0, st, d, m->zbuf, a.q,
                       a.diag ? a.u : (double *)nullptr, a.diag ? a.v : (double *)nullptr);
    if ((rc = fft2d_large<true>(d, m->zbuf, 0, 2, st))) return rc;
    hipLaunchKernelGGL(k_l_tendency, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, a, (const double2 *)m->zbuf);
    hipLaunchKernelGGL(k_l_build_pair, pw_grid(d, d.N * d.NK), dim3(256), 0, st, d, (const double2 *)a.qh_out, m->zbuf);
    if ((rc = fft2d_large<false>(d, m->zbuf, 0, 1, st))) return rc;
    hipLaunchKernelGGL(k_l_unpack_real, pw_grid(d, d.N * d.N), dim3(256), 0, st, d, m->zbuf, a.q);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

// m->q <- irfft2 of the current qh if the lazy unparameterized path left it behind
int large_ensure_q(qgx_model *m, hipStream_t st) {
    if (!m->q_stale) return QGX_OK;
    int rc = large_qh_to_q(m, m->qh[m->cur_q], m->q, st);
    if (!rc) m->q_stale = false;
    return rc;
}

int large_step(qgx_model *m, const StepArgs &a, hipStream_t st) {
    const bool unfused = !m->opts.large_fused;
    const SpecDev &d = m->d;
    const int lpb = lines_per_block(d.N);
    if (unfused || lpb < 4 || (d.N / 2) % (lpb / 4)) return large_step_unfused(m, a, st);
    const size_t lds = lines_lds(d.N, lpb);
    const int B = d.B, N = d.N;
    // Unparameterized steps (the 256 x 256 forcing-dataset runs) keep NO real-space q between steps: q = irfft2(qh)
    // rides as a third field through the two kernels that transform (u, v) — 3 launches per step instead of 5, no q
    // write + read, no second read of qh_out; m->q is refreshed on demand (large_ensure_q: snapshots, generator).
    const bool eager = !m->opts.large_lazy_q;
    const int c3 = cols3_cpb(N), ppw = rows3_ppw(N);
    const int zp = N + large_zpad();
    if (!a.has_S && !eager && c3 > 0) {
        const size_t lds1 = lines_lds(N, 6 * ppw), lds2 = lines_lds(N, 3 * c3);
        const bool generic = !m->opts.large_specialised;      // run-time-N kernels
        double2 *const ph_o = a.diag ? a.ph : (double2 *)nullptr;
        double *const u_o = a.diag ? a.u : (double *)nullptr, *const v_o = a.diag ? a.v : (double *)nullptr;
        // compile-time specialisations; tiles (mirror pairs of rows / columns / pairs per workgroup) and thread counts
        // from a sweep at 256 x 256, 64 members (bench_tools/large_sweep.sh with a -DQGX_L3_SWEEP build)
#define QGX_L3(NN, P1, T1, C2, T2, P3, T3)                                                                             \
    {                                                                                                                 \
        hipLaunchKernelGGL((k_l_rows_build_inv<3, NN, P1, T1>), dim3(B * (NN / 2 / P1)), dim3(T1), lines_lds(NN, 6 * P1), \
                           st, d, a.qh_in, m->zbuf, ph_o, P1, zp);                                                    \
        hipLaunchKernelGGL((k_l_cols3<NN, C2, T2>), dim3(B * (NN / C2)), dim3(T2), lines_lds(NN, 3 * C2), st, d, m->zbuf, \
                           u_o, v_o, C2, zp);                                                                         \
        hipLaunchKernelGGL((k_l_rows_fwd_tend<NN, P3, T3>), dim3(B * (NN / 2 / P3)), dim3(T3), lines_lds(NN, 4 * P3), st, \
                           d, a, (const double2 *)m->zbuf, P3, zp);                                                   \
    }
#ifdef QGX_L3_SWEEP      // tuning build: tile / thread variants of the 256 x 256 kernels, picked by environment variables
        static const int v1 = tune_env("QGX_V1", 0), v2 = tune_env("QGX_V2", 0), v3 = tune_env("QGX_V3", 0);
        if (!generic && N == 256 && (v1 || v2 || v3)) {
#define QGX_K1(P, T) hipLaunchKernelGGL((k_l_rows_build_inv<3, 256, P, T>), dim3(B * (128 / P)), dim3(T), lines_lds(256, 6 * P), st, d, a.qh_in, m->zbuf, ph_o, P, zp)
#define QGX_K2(C, T) hipLaunchKernelGGL((k_l_cols3<256, C, T>), dim3(B * (256 / C)), dim3(T), lines_lds(256, 3 * C), st, d, m->zbuf, u_o, v_o, C, zp)
#define QGX_K3(P, T) hipLaunchKernelGGL((k_l_rows_fwd_tend<256, P, T>), dim3(B * (128 / P)), dim3(T), lines_lds(256, 4 * P), st, d, a, (const double2 *)m->zbuf, P, zp)
            if (v1 == 1) QGX_K1(2, 512); else if (v1 == 2) QGX_K1(2, 256); else if (v1 == 3) QGX_K1(1, 128); else QGX_K1(1, 256);
            if (v2 == 1) QGX_K2(8, 512); else if (v2 == 2) QGX_K2(8, 1024); else if (v2 == 3) QGX_K2(4, 512); else if (v2 == 4) QGX_K2(4, 128);
            else if (v2 == 5) QGX_K2(2, 128); else if (v2 == 6) QGX_K2(2, 256); else QGX_K2(4, 256);
            if (v3 == 1) QGX_K3(4, 512); else if (v3 == 2) QGX_K3(4, 256); else if (v3 == 3) QGX_K3(2, 128); else if (v3 == 4) QGX_K3(1, 128);
            else if (v3 == 5) QGX_K3(1, 256); else QGX_K3(2, 256);
        } else
#endif
        if (!generic && N == 256) QGX_L3(256, 2, 512, 8, 1024, 4, 512)
        else if (!generic && N == 128) QGX_L3(128, 2, 512, 8, 1024, 4, 512)
        else if (!generic && N == 512) QGX_L3(512, 1, 256, 4, 512, 2, 256)
        else {
            static const int t1 = tune_env("QGX_LARGE_T1", 256);   // tuning aids (A/B library): threads
            static const int t2 = tune_env("QGX_LARGE_T2", 256);
            static const int t3 = tune_env("QGX_LARGE_T3", 256);
            static const int p4 = tune_env("QGX_LARGE_P4", 0);     // row pairs of the tendency kernel
            const int ppw4 = p4 > 0 && (N / 2) % p4 == 0 ? p4 : lpb / 4;
            hipLaunchKernelGGL(k_l_rows_build_inv<3>, dim3(B * ((N / 2) / ppw)), dim3(t1), lds1, st, d, a.qh_in, m->zbuf,
                               ph_o, ppw, zp);
            hipLaunchKernelGGL(k_l_cols3<>, dim3(B * (N / c3)), dim3(t2), lds2, st, d, m->zbuf, u_o, v_o, c3, zp);
            hipLaunchKernelGGL(k_l_rows_fwd_tend<>, dim3(B * ((N / 2) / ppw4)), dim3(t3), lines_lds(N, 4 * ppw4), st, d, a,
                               (const double2 *)m->zbuf, ppw4, zp);
        }
#undef QGX_L3
        QGX_HIP(hipGetLastError());
        m->q_stale = true;
        return QGX_OK;
    }
    if (m->q_stale) {            // the eager path reads q^n from memory
        int rc = large_qh_to_q(m, a.qh_in, a.q, st);
        if (rc) return rc;
        m->q_stale = false;
    }
    if (a.has_S) {
        hipLaunchKernelGGL(k_l_rows_S, dim3(B * (N / lpb)), dim3(256), lds, st, d, a.S, m->zbuf, a.weight, lpb);
        hipLaunchKernelGGL(k_l_cols<1>, dim3(B * (N / lpb)), dim3(256), lds, st, d, m->zbuf, (double *)nullptr,
                           (double *)nullptr, (double *)nullptr, lpb);
    }
    hipLaunchKernelGGL(k_l_rows_build_inv<0>, dim3(B * ((N / 2) / (lpb / 4))), dim3(256), lds, st, d, a.qh_in, m->zbuf,
                       a.diag ? a.ph : (double2 *)nullptr, lpb / 4, N);
    hipLaunchKernelGGL(k_l_cols<0>, dim3(B * 2 * (N / lpb)), dim3(256), lds, st, d, m->zbuf, a.q,
                       a.diag ? a.u : (double *)nullptr, a.diag ? a.v : (double *)nullptr, lpb);
    hipLaunchKernelGGL(k_l_rows_fwd_tend<>, dim3(B * ((N / 2) / (lpb / 4))), dim3(256), lds, st, d, a,
                       (const double2 *)m->zbuf, lpb / 4, N);
    hipLaunchKernelGGL(k_l_rows_build_inv<1>, dim3(B * ((N / 2) / (lpb / 2))), dim3(256), lds, st, d,
                       (const double2 *)a.qh_out, m->zbuf, (double2 *)nullptr, lpb / 2, N);
    hipLaunchKernelGGL(k_l_cols<2>, dim3(B * (N / lpb)), dim3(256), lds, st, d, m->zbuf, a.q, (double *)nullptr,
                       (double *)nullptr, lpb);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}


// the three launches of one diagnostics increment (kernels above); false: no specialisation for this grid size
bool large_diag_fused_ok(const qgx_model *m) { return m->opts.large_fused && (m->N == 128 || m->N == 256 || m->N == 512); }

int large_diag_fused(qgx_model *m, const DiagConst &c, const double2 *qh, const double2 *Sh, const double2 *dq_p, const double2 *dq_pp,
                     const DiagAcc &acc, hipStream_t st) {
    const SpecDev &d = m->d;
    const int B = d.B, N = d.N, ZP = N + large_zpad();
    if (!m->dg_z) {
        QGX_HIP(hipMalloc((void **)&m->dg_z, (size_t)B * DZF * ZP * N * sizeof(double2)));
        const int cap = 160 * 1024 - 512;
#define QGX_DG(NN, P, T1, C, T2)                                                                                               \
        QGX_HIP(hipFuncSetAttribute((const void *)k_l_rows_diag_inv<NN, P, T1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
        QGX_HIP(hipFuncSetAttribute((const void *)k_l_cols_diag<NN, C, T2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));     \
        QGX_HIP(hipFuncSetAttribute((const void *)k_l_rows_diag_acc<NN, P, T1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        QGX_DG(128, 4, 512, 8, 1024) QGX_DG(256, 2, 512, 8, 1024) QGX_DG(512, 1, 512, 4, 1024)
#undef QGX_DG
    }
#define QGX_DG(NN, P, T1, C, T2)                                                                                               \
    {                                                                                                                          \
        hipLaunchKernelGGL((k_l_rows_diag_inv<NN, P, T1>), dim3(B * (NN / 2 / P)), dim3(T1), lines_lds(NN, 2 * DZF * P), st, d, qh,  \
                           m->dg_z, m->ph, ZP);                                                                                \
        hipLaunchKernelGGL((k_l_cols_diag<NN, C, T2>), dim3(B * (NN / C)), dim3(T2), lines_lds(NN, DZF * C), st, d, m->dg_z, ZP); \
        hipLaunchKernelGGL((k_l_rows_diag_acc<NN, P, T1>), dim3(B * (NN / 2 / P)), dim3(T1), lines_lds(NN, 2 * DZF * P), st, d, c, \
                           (const double2 *)m->dg_z, qh, (const double2 *)m->ph, Sh, dq_p, dq_pp, acc, ZP);                    \
    }
    if (N == 128) QGX_DG(128, 4, 512, 8, 1024)
    else if (N == 256) QGX_DG(256, 2, 512, 8, 1024)
    else QGX_DG(512, 1, 512, 4, 1024)
#undef QGX_DG
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

// ================================================================================================
// XCD-resident runs of unparameterized steps (256 x 256).
//
// The three-launch step above moves ~18 MB per member-step because every row <-> column exchange of the 2-D FFTs and
// the spectral state go through HBM.  Here ONE persistent workgroup per CU is launched; the 32 workgroups that find
// themselves on the same XCD (HW_REG_XCC_ID, read at run time — a fact about where they run, not an assumption about
// dispatch) form a TEAM that owns one member at a time:
//   * the member's spectral state (qh and the two tendencies of the AB3 history) lives in the team's REGISTERS for a
//     whole run of K steps — workgroup r owns the 8 rows of mirror pairs 4r..4r+3 and the 8 columns 8r..8r+7 — so HBM is
//     touched once per run (load + store of 3 x 1.06 MB per member), not once per step;
//   * the two exchanges of a step (3 fields after the inverse-x pass, 2 fields after the forward-y pass) go through ONE
//     3.2 MB buffer per team that stays in the XCD's 4 MB L2: plain stores (the vector L1 is write-through) drained with
//     vmcnt(0) before the team barrier, consumer loads with sc1 (L1 bypass).  Both ends share one L2 by construction;
//   * the forward-x pass + tendency + AB3 step of step n and the build + inverse-x pass of step n+1 touch the same rows
//     of the same workgroup, so a step costs TWO team barriers (per-XCD counter, bounded spins).
// A census at first use (8 teams x 32 workgroups, all resident) decides whether the path is available; otherwise, and
// for parameterized / diagnostic steps and other grid sizes, the three-launch step runs.
// ================================================================================================
struct TeamCtl {
    unsigned arrived, err, pad[30];
    unsigned team_n[8][32];      // [x][0]: workgroups registered on XCD x
    unsigned bar[8][32];         // [x][0]: arrive counter of team x
    unsigned long long stamps[32];   // -DQGX_TEAM_STAMPS: phase timeline of one workgroup (10 ns ticks)
};
#ifdef QGX_TEAM_STAMPS
#define TEAM_STAMP(i) do { if (tid == 0 && rank == 0 && x == 0 && b == x && s == 3) c->stamps[i] = wall_clock64(); } while (0)
#else
#define TEAM_STAMP(i) do { } while (0)
#endif

struct TeamArgs {
    const double2 *qh_src, *p_src, *pp_src;
    double2 *qh_dst, *p_dst, *pp_dst;
    double2 *X;                  // team x works in zbuf slot x: ZF fields of N rows at pitch ZP
    TeamCtl *ctl;
    int nsteps, ablevel0, ZP, census_only;
    int fault;                   // test hook (QGX_TEAM_FAULT): finish the run, then raise the time-out flag
    double c[3][3];              // (dt1, dt2, dt3) of AB level 0, 1, 2
};


__device__ __forceinline__ unsigned team_xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }

// exchange-buffer loads: buffer loads with the sc1 bit (served by L2, never by this CU's vector L1, whose lines other
// workgroups have rewritten since) issued through the compiler's builtin, so that IT places the waits (hand-written asm
// loads leave their destination registers unprotected until a later s_waitcnt)
typedef unsigned int team_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 team_ld(__amdgpu_buffer_rsrc_t rs, size_t elem) {
    const team_u4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(elem * sizeof(double2)), 0, 16);
    return make_double2(__longlong_as_double(((unsigned long long)w.y << 32) | w.x),
                        __longlong_as_double(((unsigned long long)w.w << 32) | w.z));
}

constexpr int TEAM_NT = 512, TEAM_WG = 32, TEAM_SPIN = 1 << 21;

// every wave's stores reach L2, the workgroup arrives ONCE on the team's counter, thread 0 waits for the team.
// (Measured alternatives: one arrival per wave = 1.9x the step time; per-workgroup epoch words in one 128-byte line
// polled by a whole wave instead of the counter = +2 %.)  false: timed out (flag raised)
__device__ __forceinline__ bool team_barrier(unsigned *ctr, unsigned target, TeamCtl *c, int *ok_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0, good = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > TEAM_SPIN) { atomicExch(&c->err, 1u); good = 0; break; }
        }
        *ok_lds = good;
    }
    __syncthreads();
    return *ok_lds != 0;
}

struct TeamState { double2 q0, q1, p0, p1, pp0, pp1; };
constexpr int TEAM_NTAB = 6;                                   // per-element tables in LDS: a00 a01 a10 a11 filtr wv2

template <int NN>
__global__ __launch_bounds__(TEAM_NT) void k_l_team_steps(SpecDev d, TeamArgs a) {
    constexpr int N = NN, NK = N / 2 + 1, LD = N + 1, sz = N * NK, NT = TEAM_NT;
    constexpr int RW = N / TEAM_WG;                  // rows (and columns) per workgroup: 8
    constexpr int HALF = N / 2;                      // columns 0..HALF-1 live in registers, column HALF in LDS
    constexpr int SL = RW * HALF / NT;               // register slots per thread: 2
    static_assert(RW * HALF % NT == 0 && RW % 2 == 0, "team tiling");
    double2 *L = reinterpret_cast<double2 *>(lg_smem);
    int *pos = reinterpret_cast<int *>(L + (size_t)3 * RW * LD);
    double2 *twl = reinterpret_cast<double2 *>(pos + N);
    double2 *qsc = twl + N;                                        // [RW][2 columns][2 layers]
    TeamState *nst = reinterpret_cast<TeamState *>(qsc + RW * 4);  // [RW]  state of the Nyquist column (i = N/2)
    double *tab = reinterpret_cast<double *>(nst + RW);            // [TEAM_NTAB][SL * NT + RW]  tables of my elements, by slot
    double *kkl = tab + TEAM_NTAB * (SL * NT + RW);                // [NK]
    double *lll = kkl + NK + 1;                                    // [RW]  l of my rows
    int *ipos = reinterpret_cast<int *>(lll + RW);                 // [N]  frequency held at a digit-reversed position
    int *flags = ipos + N;                                         // [0] barrier ok, [1] xcc, [2] rank, [3] teams ok
    const int tid = threadIdx.x;
    TeamCtl *c = a.ctl;

    // ---- census: who shares my XCD ----
    if (tid == 0) {
        const unsigned x = team_xcc_id();
        flags[1] = (int)x;
        flags[2] = (int)atomicAdd(&c->team_n[x][0], 1u);
        __hip_atomic_fetch_add(&c->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0, good = 1;
        while (__hip_atomic_load(&c->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > TEAM_SPIN) { atomicExch(&c->err, 2u); good = 0; break; }
        }
        if (good)
            for (int t = 0; t < 8; ++t)
                if (__hip_atomic_load(&c->team_n[t][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != TEAM_WG) good = 0;
        if (!good) atomicCAS(&c->err, 0u, 3u);
        flags[3] = good;
    }
    for (int t = tid; t < N; t += NT) { pos[t] = d.pos[t]; twl[t] = d.tw[t]; ipos[d.pos[t]] = t; }
    __syncthreads();
    if (!flags[3] || a.census_only) return;
    const int x = flags[1], rank = flags[2];
    unsigned *ctr = &c->bar[x][0];
    unsigned phase = 0;
    const int p0 = rank * (RW / 2), c0 = rank * RW;
    const size_t fz = (size_t)a.ZP * N;
    double2 *X = a.X + (size_t)x * ZF * fz;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, (int)(ZF * fz * sizeof(double2)), 0x00020000);
    const int ZP = a.ZP;

    // ---- element slots of this thread, tables of the rows this workgroup owns for the whole launch (LDS) ----
    // slot t = tid + e * NT owns column i = 16 a + b of local row rl, with a = t & 7, rl = (t >> 3) & 7, b = t >> 6: eight
    // consecutive lanes address eight consecutive digit-reversed positions pos[i] = 16 b + a (and N - i likewise), so the
    // scattered LDS accesses of the build / tendency phases run 128 contiguous bytes per eight lanes; the Nyquist column
    // (slot index SL * NT + rl) belongs to lanes 0..7 of wave 1, the self-conjugate column 0 to wave 0
    constexpr int NE = SL * NT + RW;
    static_assert(N == 256, "slot mapping is written for 256 = 16 x 16");
    int el_i[SL], el_r[SL], el_j[SL];
#pragma unroll
    for (int e = 0; e < SL; ++e) {
        const int t = tid + e * NT;
        el_i[e] = 16 * (t & 7) + (t >> 6); el_r[e] = (t >> 3) & 7;
        el_j[e] = pair_row(p0 + (el_r[e] >> 1), el_r[e] & 1, N);
        const int idx = el_j[e] * NK + el_i[e];
#pragma unroll
        for (int q = 0; q < 4; ++q) tab[q * NE + t] = d.a[(size_t)q * sz + idx];
        tab[4 * NE + t] = d.filtr[idx];
        tab[5 * NE + t] = d.wv2[idx];
    }
    const bool nyq = tid >= 64 && tid < 64 + RW;
    const int nr = tid - 64;                                   // local row of my Nyquist element
    if (nyq) {
        const int idx = pair_row(p0 + (nr >> 1), nr & 1, N) * NK + HALF, t = SL * NT + nr;
#pragma unroll
        for (int q = 0; q < 4; ++q) tab[q * NE + t] = d.a[(size_t)q * sz + idx];
        tab[4 * NE + t] = d.filtr[idx];
        tab[5 * NE + t] = d.wv2[idx];
    }
    for (int t = tid; t < NK; t += NT) kkl[t] = d.kk[t];
    if (tid < RW) lll[tid] = d.ll[pair_row(p0 + (tid >> 1), tid & 1, N)];
    __syncthreads();

    // one element of the build phase: spectra of (u_k + i v_k) and (q_1 + i q_2), Hermitian-extended, into the LDS lines
    auto build = [&](const TeamState &st, int i, int rl, int te) {
        const int rm = (p0 + (rl >> 1)) == 0 ? rl : (rl ^ 1);
        const bool selfc = (i == 0 || 2 * i == N);
        const double2 q0 = st.q0, q1 = st.q1;
        double2 q0m = q0, q1m = q1;
        const int tm = i == 0 ? (rm << 3) : SL * NT + rm;      // slot of the mirror row's element (self-conjugate columns)
        if (selfc) {
            const int ic = i == 0 ? 0 : 1;
            q0m = qsc[(rm * 2 + ic) * 2]; q1m = qsc[(rm * 2 + ic) * 2 + 1];
        }
        const double kx = kkl[i], ly = lll[rl], lm = lll[rm];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double a0 = tab[(2 * k) * NE + te], a1 = tab[(2 * k + 1) * NE + te];
            const double2 ph = make_double2(a0 * q0.x + a1 * q1.x, a0 * q0.y + a1 * q1.y);
            double2 uh = make_double2(ly * ph.y, -ly * ph.x);
            double2 vh = make_double2(-kx * ph.y, kx * ph.x);
            if (selfc) {
                const double m0 = tab[(2 * k) * NE + tm], m1 = tab[(2 * k + 1) * NE + tm];
                const double2 pm = make_double2(m0 * q0m.x + m1 * q1m.x, m0 * q0m.y + m1 * q1m.y);
                const double2 um = make_double2(lm * pm.y, -lm * pm.x);
                const double2 vm = make_double2(-kx * pm.y, kx * pm.x);
                uh = make_double2(0.5 * (uh.x + um.x), 0.5 * (uh.y - um.y));
                vh = make_double2(0.5 * (vh.x + vm.x), 0.5 * (vh.y - vm.y));
            }
            L[(rl * 3 + k) * LD + pos[i]] = make_double2((uh.x - vh.y) * d.invN2, (uh.y + vh.x) * d.invN2);
            if (!selfc) L[(rm * 3 + k) * LD + pos[N - i]] = make_double2((uh.x + vh.y) * d.invN2, (vh.x - uh.y) * d.invN2);
        }
        double2 A = q0, Bq = q1;
        if (selfc) {
            A = make_double2(0.5 * (A.x + q0m.x), 0.5 * (A.y - q0m.y));
            Bq = make_double2(0.5 * (Bq.x + q1m.x), 0.5 * (Bq.y - q1m.y));
        }
        L[(rl * 3 + 2) * LD + pos[i]] = make_double2((A.x - Bq.y) * d.invN2, (A.y + Bq.x) * d.invN2);
        if (!selfc) L[(rm * 3 + 2) * LD + pos[N - i]] = make_double2((A.x + Bq.y) * d.invN2, (Bq.x - A.y) * d.invN2);
    };

    // one element of the tendency phase: unpack (uq_k, vq_k) from the transformed lines, tendency, AB3 + filter, rotate
    auto tend = [&](TeamState &st, int i, int rl, int te, double dt1, double dt2, double dt3) {
        const int rm = (p0 + (rl >> 1)) == 0 ? rl : (rl ^ 1);
        const int im = i == 0 ? 0 : N - i;
        const double kx = kkl[i], ly = lll[rl];
        const double2 q0 = st.q0, q1 = st.q1;
        double2 tn[2], qn[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double2 A = L[(rl * 2 + k) * LD + pos[i]], C = L[(rm * 2 + k) * LD + pos[im]];
            const double2 uqh = make_double2(0.5 * (A.x + C.x), 0.5 * (A.y - C.y));
            const double2 vqh = make_double2(0.5 * (A.y + C.y), -0.5 * (A.x - C.x));
            const double a0 = tab[(2 * k) * NE + te], a1 = tab[(2 * k + 1) * NE + te];
            const double2 ph = make_double2(a0 * q0.x + a1 * q1.x, a0 * q0.y + a1 * q1.y);
            const double kq = kx * d.Qy[k];
            double tx = (kx * uqh.y + ly * vqh.y + kq * ph.y);
            double ty = -(kx * uqh.x + ly * vqh.x + kq * ph.x);
            if (k == 1 && d.rek != 0.0) {
                const double f = d.rek * tab[5 * NE + te];
                tx += f * ph.x;
                ty += f * ph.y;
            }
            tx += 0.0; ty += 0.0;                    // the (absent) forcing of k_l_rows_fwd_tend
            const double2 p = k == 0 ? st.p0 : st.p1, pp = k == 0 ? st.pp0 : st.pp1;
            const double2 qk = k == 0 ? q0 : q1;
            const double f = tab[4 * NE + te];
            tn[k] = make_double2(tx, ty);
            qn[k] = make_double2(f * (qk.x + dt1 * tx + dt2 * p.x + dt3 * pp.x),
                                 f * (qk.y + dt1 * ty + dt2 * p.y + dt3 * pp.y));
        }
        st.pp0 = st.p0; st.pp1 = st.p1;
        st.p0 = tn[0]; st.p1 = tn[1];
        st.q0 = qn[0]; st.q1 = qn[1];
    };

    for (int b = x; b < d.B; b += 8) {
        const size_t mo = (size_t)b * 2 * sz;
        TeamState S[SL];
#pragma unroll
        for (int e = 0; e < SL; ++e) {
            const size_t o = mo + (size_t)el_j[e] * NK + el_i[e];
            S[e] = TeamState{a.qh_src[o], a.qh_src[o + sz], a.p_src[o], a.p_src[o + sz], a.pp_src[o], a.pp_src[o + sz]};
        }
        if (nyq) {
            const size_t o = mo + (size_t)pair_row(p0 + (nr >> 1), nr & 1, N) * NK + HALF;
            nst[nr] = TeamState{a.qh_src[o], a.qh_src[o + sz], a.p_src[o], a.p_src[o + sz], a.pp_src[o], a.pp_src[o + sz]};
        }
        for (int s = 0; s <= a.nsteps; ++s) {
            // (addresses derive from an opaque copy of the thread id: hoisted out of the step loop they would pin ~80 registers)
            int tl = tid;
            asm volatile("" : "+v"(tl));
            if (s > 0) {
                // ---- rows: forward along x of the two product fields, tendency, AB3 step ----
                TEAM_STAMP(0);
                // first radix-16 pass of the forward transform on operands taken straight from the exchange buffer: task
                // (line, b) owns elements {b + 16 m} of its row, sixteen consecutive lanes read 256 contiguous bytes
                if (tl < 2 * RW * 16) {
                    const int b2 = tl & 15, line = tl >> 4;
                    const int j = pair_row(p0 + (line >> 2), (line >> 1) & 1, N);
                    const size_t ro = (size_t)(line & 1) * fz + (size_t)j * ZP + b2;
                    double2 v[16];
#pragma unroll
                    for (int m = 0; m < 16; ++m) v[m] = team_ld(xrs, ro + 16 * m);
                    small_dft<16, true>(v);
#pragma unroll
                    for (int m = 1; m < 16; ++m) v[m] = cmul(v[m], twl[m * b2]);
                    double2 *ln = L + line * LD + b2;
#pragma unroll
                    for (int m = 0; m < 16; ++m) ln[16 * m] = v[m];
                }
                __syncthreads();
                TEAM_STAMP(1);
                fft_pass<16, true>(L, 2 * RW, LD, 1, 16, N, twl);
                __syncthreads();
                TEAM_STAMP(2);
                const int lev = a.ablevel0 + s - 1 > 2 ? 2 : a.ablevel0 + s - 1;
                const double dt1 = a.c[lev][0], dt2 = a.c[lev][1], dt3 = a.c[lev][2];
#pragma unroll
                for (int e = 0; e < SL; ++e) tend(S[e], el_i[e], el_r[e], tid + e * NT, dt1, dt2, dt3);
                if (nyq) {
                    TeamState st = nst[nr];
                    tend(st, HALF, nr, SL * NT + nr, dt1, dt2, dt3);
                    nst[nr] = st;
                }
                TEAM_STAMP(3);
                if (s == a.nsteps) { __syncthreads(); break; }
            }
            // ---- rows: build the three packed spectra from qh, inverse along x, publish the rows ----
            // (the self-conjugate columns need q of their mirror row: published here, ONE barrier covers it and the
            // tendency phase's last reads of the lines the build phase overwrites)
#pragma unroll
            for (int e = 0; e < SL; ++e)
                if (el_i[e] == 0) { qsc[(el_r[e] * 2) * 2] = S[e].q0; qsc[(el_r[e] * 2) * 2 + 1] = S[e].q1; }
            if (nyq) { qsc[(nr * 2 + 1) * 2] = nst[nr].q0; qsc[(nr * 2 + 1) * 2 + 1] = nst[nr].q1; }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < SL; ++e) build(S[e], el_i[e], el_r[e], tid + e * NT);
            if (nyq) build(nst[nr], HALF, nr, SL * NT + nr);
            __syncthreads();
            TEAM_STAMP(4);
            fft_pass<16, false>(L, 3 * RW, LD, 1, 16, N, twl);
            __syncthreads();
            TEAM_STAMP(5);
            // second radix-16 pass of the inverse in registers, rows stored straight to the exchange buffer
            if (tl < 3 * RW * 16) {
                const int b2 = tl & 15, line = tl >> 4;
                const int rl = line / 3, f = line - rl * 3;
                const int j = pair_row(p0 + (rl >> 1), rl & 1, N);
                const double2 *ln = L + line * LD + b2;
                double2 v[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) v[m] = ln[16 * m];
#pragma unroll
                for (int m = 1; m < 16; ++m) v[m] = cmulc(v[m], twl[m * b2]);
                small_dft<16, false>(v);
                double2 *xo = X + (size_t)f * fz + (size_t)j * ZP + b2;
#pragma unroll
                for (int m = 0; m < 16; ++m) xo[16 * m] = v[m];
            }
            TEAM_STAMP(6);
            if (!team_barrier(ctr, ++phase * TEAM_WG, c, flags)) return;
            TEAM_STAMP(7);
            // ---- columns: inverse along y of the three fields, advection products, forward along y of two ----
            // 256 = 16 x 16: the first radix-16 pass of the inverse takes its operands straight from the exchange buffer
            // and the last pass of the forward transform stores straight to it; between them the inverse's second pass,
            // the products and the forward's first pass work on the same sixteen rows {b + 16 m} of a column in registers.
            // LDS sees 4 sweeps of a line instead of 12, the phase 3 barriers instead of 9.
            {
                const int cc = tl & 7, blk = (tl >> 3) & 15, k1 = tl >> 7;          // pass 1 / last pass: (field, column, block)
                double2 v[16];
                if (tl < 3 * RW * 16) {
#pragma unroll
                    for (int m = 0; m < 16; ++m) v[m] = team_ld(xrs, (size_t)k1 * fz + (size_t)ipos[16 * blk + m] * ZP + c0 + cc);
                    TEAM_STAMP(8);
                    small_dft<16, false>(v);
                    double2 *base = L + (k1 * RW + cc) * LD + 16 * blk;
#pragma unroll
                    for (int m = 0; m < 16; ++m) base[m] = v[m];
                }
                __syncthreads();
                TEAM_STAMP(9);
                // second pass of the inverse: tasks (field, column, b); the q field publishes its rows, (u, v) keep theirs
                const int b2 = tl & 15, c2 = (tl >> 4) & 7, k2 = tl >> 7;
                double2 *line2 = L + (k2 * RW + c2) * LD + b2;
                if (tl < 3 * RW * 16) {
#pragma unroll
                    for (int m = 0; m < 16; ++m) v[m] = line2[16 * m];
#pragma unroll
                    for (int m = 1; m < 16; ++m) v[m] = cmulc(v[m], twl[m * b2]);
                    small_dft<16, false>(v);
                    if (k2 == 2) {
#pragma unroll
                        for (int m = 0; m < 16; ++m) line2[16 * m] = v[m];
                    }
                }
                __syncthreads();
                TEAM_STAMP(10);
                if (tl < 2 * RW * 16) {
                    const double2 *ql = L + (2 * RW + c2) * LD + b2;
                    const double Uk = d.U[k2];
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        const double2 qq = ql[16 * m];
                        const double qv = k2 == 0 ? qq.x : qq.y;
                        v[m] = make_double2((v[m].x + Uk) * qv, v[m].y * qv);
                    }
                    small_dft<16, true>(v);
#pragma unroll
                    for (int m = 1; m < 16; ++m) v[m] = cmul(v[m], twl[m * b2]);
#pragma unroll
                    for (int m = 0; m < 16; ++m) line2[16 * m] = v[m];
                }
                __syncthreads();
                TEAM_STAMP(11);
                if (tl < 2 * RW * 16) {
                    const double2 *base = L + (k1 * RW + cc) * LD + 16 * blk;
#pragma unroll
                    for (int m = 0; m < 16; ++m) v[m] = base[m];
                    small_dft<16, true>(v);
#pragma unroll
                    for (int m = 0; m < 16; ++m) X[(size_t)k1 * fz + (size_t)ipos[16 * blk + m] * ZP + c0 + cc] = v[m];
                }
            }
            TEAM_STAMP(12);
            if (!team_barrier(ctr, ++phase * TEAM_WG, c, flags)) return;
            TEAM_STAMP(13);
        }
        // ---- the run is over for this member: state back to memory ----
#pragma unroll
        for (int e = 0; e < SL; ++e) {
            const size_t o = mo + (size_t)el_j[e] * NK + el_i[e];
            a.qh_dst[o] = S[e].q0; a.qh_dst[o + sz] = S[e].q1;
            a.p_dst[o] = S[e].p0; a.p_dst[o + sz] = S[e].p1;
            a.pp_dst[o] = S[e].pp0; a.pp_dst[o + sz] = S[e].pp1;
        }
        if (nyq) {
            const size_t o = mo + (size_t)pair_row(p0 + (nr >> 1), nr & 1, N) * NK + HALF;
            const TeamState st = nst[nr];
            a.qh_dst[o] = st.q0; a.qh_dst[o + sz] = st.q1;
            a.p_dst[o] = st.p0; a.p_dst[o + sz] = st.p1;
            a.pp_dst[o] = st.pp0; a.pp_dst[o + sz] = st.pp1;
        }
        __syncthreads();
    }
    if (a.fault && tid == 0 && blockIdx.x == 0) atomicExch(&c->err, 1u);
}

static size_t team_lds(int N) {
    const int RW = N / TEAM_WG;
    const int NK = N / 2 + 1;
    return (size_t)3 * RW * (N + 1) * 16 + (size_t)N * 4 + (size_t)N * 16 + (size_t)RW * 4 * 16 + (size_t)RW * sizeof(TeamState) +
           (size_t)TEAM_NTAB * (2 * TEAM_NT + RW) * 8 + (size_t)(NK + 1) * 8 + (size_t)RW * 8 + (size_t)N * 4 + 64;
}

// 1: the device runs 8 teams of 32 co-resident workgroups (census passed), 0: it does not.  Decided once per model.
int large_team_available(qgx_model *m, hipStream_t st) {
    if (!m->opts.team || m->N != 256 || m->B < 1) return 0;
    if (m->team_state != 0) return m->team_state > 0;
    m->team_state = -1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, m->cfg.device) != hipSuccess || prop.multiProcessorCount != 8 * TEAM_WG) return 0;
    if (hipMalloc(&m->team_ctl, sizeof(TeamCtl)) != hipSuccess) { m->team_ctl = nullptr; return 0; }
    if (hipFuncSetAttribute((const void *)k_l_team_steps<256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 512) != hipSuccess) return 0;
    TeamArgs a{};
    a.ctl = (TeamCtl *)m->team_ctl;
    a.census_only = 1;
    a.ZP = m->N + large_zpad();
    if (hipMemsetAsync(m->team_ctl, 0, sizeof(TeamCtl), st) != hipSuccess) return 0;
    hipLaunchKernelGGL((k_l_team_steps<256>), dim3(8 * TEAM_WG), dim3(TEAM_NT), team_lds(256), st, m->d, a);
    TeamCtl h;
    if (hipMemcpyAsync(&h, m->team_ctl, sizeof(TeamCtl), hipMemcpyDeviceToHost, st) != hipSuccess) return 0;
    if (hipStreamSynchronize(st) != hipSuccess) return 0;
    if (h.err == 0) m->team_state = 1;
    return m->team_state > 0;
}

// reads back the flag word of the pending run (one stream synchronisation): *flag = 0 if the run completed, else what
// stopped it (1: a team barrier timed out, 2: workgroups not co-resident, 3: census changed).  The caller
// (model.hip::team_settle) undoes and replays a flagged run.
int large_team_check(qgx_model *m, hipStream_t st, unsigned *flag) {
    *flag = 0;
    if (!m->team_pending) return QGX_OK;
    m->team_pending = false;
    TeamCtl h;
    QGX_HIP(hipMemcpyAsync(&h, m->team_ctl, sizeof(TeamCtl), hipMemcpyDeviceToHost, st));
    QGX_HIP(hipStreamSynchronize(st));
#ifdef QGX_TEAM_STAMPS
    {
        static const char *names[] = {"C load rows", "C fft fwd x", "C tendency", "A build", "A fft inv x", "A store rows", "barrier 1",
                                      "B load cols", "B fft inv y", "B products", "B fft fwd y", "B store cols", "barrier 2"};
        fprintf(stderr, "team step timeline (workgroup 0 of team 0, step 3; us):");
        for (int i = 0; i < 13; ++i) fprintf(stderr, " %s %.2f |", names[i], (double)(long long)(h.stamps[i + 1] - h.stamps[i]) / 100.0);
        fprintf(stderr, " total %.2f\n", (double)(long long)(h.stamps[13] - h.stamps[0]) / 100.0);
    }
#endif
    *flag = h.err;
    return QGX_OK;
}

// K consecutive unparameterized steps without diagnostics output; coef[level] = (dt1, dt2, dt3)
int large_team_steps(qgx_model *m, int K, int ablevel0, const double coef[3][3], const double2 *qh_src, double2 *qh_dst,
                     const double2 *p_src, const double2 *pp_src, double2 *p_dst, double2 *pp_dst, hipStream_t st) {
    QGX_REQUIRE(!m->team_pending, "large_team_steps: the previous run has not been settled");
    TeamArgs a{};
    a.qh_src = qh_src; a.qh_dst = qh_dst; a.p_src = p_src; a.pp_src = pp_src; a.p_dst = p_dst; a.pp_dst = pp_dst;
    a.X = m->zbuf; a.ctl = (TeamCtl *)m->team_ctl;
    a.nsteps = K; a.ablevel0 = ablevel0; a.ZP = m->N + large_zpad(); a.census_only = 0;
    a.fault = m->opts.team_fault;       // test hook (A/B library): the flag path of large_team_check
    m->opts.team_fault = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a.c[i][j] = coef[i][j];
    QGX_HIP(hipMemsetAsync(m->team_ctl, 0, sizeof(TeamCtl), st));
    hipLaunchKernelGGL((k_l_team_steps<256>), dim3(8 * TEAM_WG), dim3(TEAM_NT), team_lds(256), st, m->d, a);
    QGX_HIP(hipGetLastError());
    m->team_pending = true;
    m->team_stream = st;
    m->q_stale = true;
    return QGX_OK;
}

}  // namespace qgx
