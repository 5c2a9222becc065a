// LDS-resident spectral kernels: one workgroup advances one ensemble member.
//
// Restates pyqg 0.7.2 kernel.pyx::{_invert,_do_advection,_do_friction,
// _do_q_subgrid_parameterization,_forward_timestep} (call sites in the reference:
// pyqg_generative/tools/simulate.py:132,137,168; operators.py:233) for the
// ensemble-batched layout (B,2,N,N).  The twelve real 2-D FFTs of one step are
// executed as six complex N x N transforms of packed pairs
//   (u_k + i v_k), ((u_k+U_k) q_k + i v_k q_k), (S_1 + i S_2), (q_1 + i q_2)
// entirely inside one CU's LDS (N <= 96: N*(N+1)*16 B <= 149 KB).
#include "../pyqg_generative_amd/csrc/common.hpp"
#include "../pyqg_generative_amd/csrc/fft_lds.hpp"
#include <cstdlib>

namespace qgx {

__device__ __forceinline__ int neg_mod(int j, int N) { return j == 0 ? 0 : N - j; }

struct Grid {
    int N, NK, LD, nrad;
    const int *rad;     // registers/param space
    const int *pos;     // LDS copy
    const double2 *tw;
};

// half-spectra (A,B) of the real parts packed as A + iB, read out of the DIF-ordered field
__device__ __forceinline__ void unpack_pair(const double2 *Z, const Grid &g, int j, int i,
                                            double2 &A, double2 &Bv) {
    const int jm = neg_mod(j, g.N), im = neg_mod(i, g.N);
    const double2 a = Z[g.pos[j] * g.LD + g.pos[i]];
    const double2 b = cconj(Z[g.pos[jm] * g.LD + g.pos[im]]);
    A = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
    // -i/2 * (a - b)
    Bv = make_double2(0.5 * (a.y - b.y), -0.5 * (a.x - b.x));
}

// store the Hermitian extension of (Ah + i Bh) at (j,i) [and its mirror], DIT-input order.
// For the self-conjugate columns the caller passes already symmetrised values.
__device__ __forceinline__ void pack_store(double2 *Z, const Grid &g, int j, int i, double2 Ah,
                                           double2 Bh, double scale) {
    Z[g.pos[j] * g.LD + g.pos[i]] = make_double2((Ah.x - Bh.y) * scale, (Ah.y + Bh.x) * scale);
    if (i != 0 && 2 * i != g.N) {
        const int jm = neg_mod(j, g.N);
        // conj(Ah) + i conj(Bh)
        Z[g.pos[jm] * g.LD + g.pos[g.N - i]] =
            make_double2((Ah.x + Bh.y) * scale, (Bh.x - Ah.y) * scale);
    }
}

__device__ __forceinline__ double2 invert_layer(const SpecDev &d, int k, int idx, double2 q0, double2 q1) {
    const int sz = d.N * d.NK;
    const double a0 = d.a[(2 * k) * sz + idx], a1 = d.a[(2 * k + 1) * sz + idx];
    return make_double2(a0 * q0.x + a1 * q1.x, a0 * q0.y + a1 * q1.y);
}

// Build the packed spectrum of (u_k + i v_k) from qh; optionally store ph_k.
__device__ __forceinline__ void build_uv(double2 *Z, const Grid &g, const SpecDev &d, int k,
                                         const double2 *qh0, const double2 *qh1, double2 *ph_out) {
    const int N = g.N, NK = g.NK;
    for (int idx = threadIdx.x; idx < N * NK; idx += blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        const double2 ph = invert_layer(d, k, idx, qh0[idx], qh1[idx]);
        if (ph_out) ph_out[idx] = ph;
        const double kx = d.kk[i], ly = d.ll[j];
        // uh = -i l ph ; vh = i k ph
        double2 uh = make_double2(ly * ph.y, -ly * ph.x);
        double2 vh = make_double2(-kx * ph.y, kx * ph.x);
        if (i == 0 || 2 * i == N) {
            const int jm = neg_mod(j, N);
            const int idm = jm * NK + i;
            const double2 pm = invert_layer(d, k, idm, qh0[idm], qh1[idm]);
            const double lm = d.ll[jm];
            const double2 um = make_double2(lm * pm.y, -lm * pm.x);
            const double2 vm = make_double2(-kx * pm.y, kx * pm.x);
            uh = make_double2(0.5 * (uh.x + um.x), 0.5 * (uh.y - um.y));
            vh = make_double2(0.5 * (vh.x + vm.x), 0.5 * (vh.y - vm.y));
        }
        pack_store(Z, g, j, i, uh, vh, d.invN2);
    }
}

// Build the packed spectrum of (A + i B) from two half spectra in global memory.
__device__ __forceinline__ void build_pair(double2 *Z, const Grid &g, const double2 *Ah,
                                           const double2 *Bh, double scale) {
    const int N = g.N, NK = g.NK;
    for (int idx = threadIdx.x; idx < N * NK; idx += blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        double2 a = Ah[idx], b = Bh[idx];
        if (i == 0 || 2 * i == N) {
            const int idm = neg_mod(j, N) * NK + i;
            const double2 am = Ah[idm], bm = Bh[idm];
            a = make_double2(0.5 * (a.x + am.x), 0.5 * (a.y - am.y));
            b = make_double2(0.5 * (b.x + bm.x), 0.5 * (b.y - bm.y));
        }
        pack_store(Z, g, j, i, a, b, scale);
    }
}

__device__ __forceinline__ Grid make_grid(const SpecDev &d, double2 *Z, int *&pos_lds) {
    Grid g;
    g.N = d.N; g.NK = d.NK; g.LD = d.LD; g.nrad = d.nrad; g.rad = d.rad; g.tw = d.tw;
    pos_lds = reinterpret_cast<int *>(Z + d.N * d.LD);
    for (int t = threadIdx.x; t < d.N; t += blockDim.x) pos_lds[t] = d.pos[t];
    g.pos = pos_lds;
    return g;
}

extern __shared__ __attribute__((aligned(16))) char qgx_smem[];

// ------------------------------------------------------------------ one time step
__global__ void k_step_small_stamped(SpecDev d, StepArgs a, unsigned long long *stamps) {
    int sidx = 0;
#define STAMP() do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) stamps[sidx] = __builtin_amdgcn_s_memrealtime(); ++sidx; } while (0)
    STAMP();
    double2 *Z = reinterpret_cast<double2 *>(qgx_smem);
    int *pos_lds;
    Grid g = make_grid(d, Z, pos_lds);
    const int N = d.N, NK = d.NK, LD = d.LD;
    const int b = blockIdx.x;
    const size_t so = (size_t)b * 2 * N * NK, ro = (size_t)b * 2 * N * N;
    const int sz = N * NK, rz = N * N;
    const double2 *qh0 = a.qh_in + so, *qh1 = qh0 + sz;
    __syncthreads();

    // ---- subgrid forcing: Sh_k = rfft2(weight * S_k), pair packed (pyqg _do_q_subgrid_parameterization)
    if (a.has_S) {
        const double *S0 = a.S + ro, *S1 = S0 + rz;
        for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
            const int y = idx / N, x = idx - y * N;
            Z[y * LD + x] = make_double2(a.weight * S0[idx], a.weight * S1[idx]);
        }
        __syncthreads();
        STAMP();  /* S loaded */
        fft2d_fwd(Z, N, LD, g.nrad, g.rad, g.tw);
        STAMP();  /* S fft */
        for (int idx = threadIdx.x; idx < sz; idx += blockDim.x) {
            const int j = idx / NK, i = idx - j * NK;
            double2 s0, s1;
            unpack_pair(Z, g, j, i, s0, s1);
            if (a.demean && idx == 0) { s0 = make_double2(0., 0.); s1 = s0; }
            a.dqh[so + idx] = s0;
            a.dqh[so + sz + idx] = s1;
        }
        __syncthreads();
    }

    for (int k = 0; k < 2; ++k) {
        // ---- _invert: ph_k, (u_k, v_k) = irfft2(-il ph, ik ph)
        STAMP();  /* before build_uv */
        build_uv(Z, g, d, k, qh0, qh1, a.diag ? a.ph + so + k * sz : nullptr);
        STAMP();  /* build_uv */
        fft2d_inv(Z, N, LD, g.nrad, g.rad, g.tw);
        STAMP();  /* inv fft */
        // ---- _do_advection, real space: uq = (u+U) q, vq = v q
        {
            const double *qk = a.q + ro + k * rz;
            const double Uk = d.U[k];
            for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
                const int y = idx / N, x = idx - y * N;
                const double2 uv = Z[y * LD + x];
                if (a.diag) { a.u[ro + k * rz + idx] = uv.x; a.v[ro + k * rz + idx] = uv.y; }
                const double qv = qk[idx];
                Z[y * LD + x] = make_double2((uv.x + Uk) * qv, uv.y * qv);
            }
        }
        STAMP();  /* products */
        fft2d_fwd(Z, N, LD, g.nrad, g.rad, g.tw);
        STAMP();  /* fwd fft */
        // ---- spectral tendency, friction, forcing, AB3 + filter (_forward_timestep)
        for (int idx = threadIdx.x; idx < sz; idx += blockDim.x) {
            const int j = idx / NK, i = idx - j * NK;
            double2 uqh, vqh;
            unpack_pair(Z, g, j, i, uqh, vqh);
            const double2 q0 = qh0[idx], q1 = qh1[idx];
            const double2 ph = invert_layer(d, k, idx, q0, q1);
            const double kx = d.kk[i], ly = d.ll[j];
            const double kq = kx * d.Qy[k];
            // -(ik uqh + il vqh + ik Qy ph)
            double tx = (kx * uqh.y + ly * vqh.y + kq * ph.y);
            double ty = -(kx * uqh.x + ly * vqh.x + kq * ph.x);
            if (k == 1 && d.rek != 0.0) {
                const double f = d.rek * d.wv2[idx];
                tx += f * ph.x;
                ty += f * ph.y;
            }
            if (a.has_S) {
                const double2 s = a.dqh[so + k * sz + idx];
                tx += s.x;
                ty += s.y;
            }
            const size_t o = so + k * sz + idx;
            const double2 p = a.dq_p[o], pp = a.dq_pp[o];
            const double2 qk = k == 0 ? q0 : q1;
            const double f = d.filtr[idx];
            a.dq_new[o] = make_double2(tx, ty);
            a.qh_out[o] = make_double2(f * (qk.x + a.dt1 * tx + a.dt2 * p.x + a.dt3 * pp.x),
                                       f * (qk.y + a.dt1 * ty + a.dt2 * p.y + a.dt3 * pp.y));
        }
        __syncthreads();
    }
    STAMP();  /* tendency (last layer) */
    // ---- q^{n+1} = irfft2(qh^{n+1}), both layers packed
    build_pair(Z, g, a.qh_out + so, a.qh_out + so + sz, d.invN2);
    STAMP();  /* build_pair */
    fft2d_inv(Z, N, LD, g.nrad, g.rad, g.tw);
    STAMP();  /* q inv fft */
    for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
        const int y = idx / N, x = idx - y * N;
        const double2 w = Z[y * LD + x];
        a.q[ro + idx] = w.x;
        a.q[ro + rz + idx] = w.y;
    }
    STAMP();  /* q store */
#undef STAMP
}

}  // namespace qgx

// Diagnostic build ONLY (never timed as a whole): phase shares of one workgroup of the spectral step.
#include <vector>
#include <cmath>
using namespace qgx;
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1, N = 64, NK = 33, T = argc > 2 ? atoi(argv[2]) : 1024;
    qgx_config cfg = {N, B, 0, 0, 1e6, 14400., 5.787e-7, 0.25, 1.5e-11, 15000., 0.025, 0., 500., 23.6};
    qgx_model *m;
    if (qgx_create(&cfg, &m)) { printf("create failed: %s\n", qgx_last_error()); return 1; }
    std::vector<double> q((size_t)B * 2 * N * N);
    for (size_t i = 0; i < q.size(); ++i) q[i] = 1e-6 * sin(0.37 * i);
    double *qd; hipMalloc(&qd, q.size() * 8); hipMemcpy(qd, q.data(), q.size() * 8, hipMemcpyHostToDevice);
    qgx_set_q(m, qd, 0);
    qgx_step(m, 3, nullptr, 0, 0);
    unsigned long long *st; hipMalloc(&st, 64 * 8); hipMemset(st, 0, 64 * 8);
    StepArgs a; a.qh_in = m->qh[m->cur_q]; a.qh_out = m->qh[m->cur_q ^ 1]; a.q = m->q; a.S = m->S; a.dqh = m->dqh;
    a.dq_new = m->dq[m->i_pp]; a.dq_p = m->dq[m->i_new]; a.dq_pp = m->dq[m->i_p]; a.ph = m->ph; a.u = m->u; a.v = m->v;
    a.dt1 = 1.9 * 14400; a.dt2 = -1.3 * 14400; a.dt3 = 0.4 * 14400; a.weight = 1; a.has_S = 1; a.demean = 1; a.diag = 0;
    const size_t lds = (size_t)N * (N + 1) * 16 + N * 4 + 16;
    hipFuncSetAttribute((const void *)k_step_small_stamped, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_step_small_stamped, dim3(B), dim3(T), lds, 0, m->d, a, st);
        hipDeviceSynchronize();
    }
    unsigned long long h[64]; hipMemcpy(h, st, 64 * 8, hipMemcpyDeviceToHost);
    const char *names[] = {"S load", "S fwd fft", "S unpack+L0 pre", "L0 build_uv", "L0 inv fft", "L0 products", "L0 fwd fft",
                           "L0 tendency", "L1 build_uv", "L1 inv fft", "L1 products", "L1 fwd fft", "L1 tendency", "build_pair",
                           "q inv fft", "q store"};
    printf("B=%d threads=%d (100 MHz ticks -> us)\n", B, T);
    for (int i = 1; i < 17 && h[i]; ++i) printf("  %-18s %6.2f us\n", names[i - 1], (h[i] - h[i - 1]) / 100.0);
    printf("  total %.2f us\n", (h[16] - h[0]) / 100.0);
    return 0;
}
