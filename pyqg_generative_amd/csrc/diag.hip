// Time-averaged spectral diagnostics of the two-layer model, accumulated on the device.
//
// Restates pyqg 0.7.2 model.py::{_calc_diagnostics,_increment_diagnostics} and the diagnostic
// definitions of model.py / qg_model.py (KEspec, Ensspec, entspec, APEflux, KEflux, APEgenspec,
// KEfrictionspec, paramspec, paramspec_APEflux, paramspec_KEflux, Dissspec, ENSDissspec, ENSflux, ENSgenspec,
// ENSfrictionspec, ENSparamspec — the sixteen keys of comparison_tools.py:222-225), which the reference consumes in
// pyqg_generative/tools/comparison_tools.py:91,106,164-188 (paramspec_* at :174-176),222-247 and plots as
// calc_ispec(m, 0.5*ave_lev(KEspec)) (Google-Colab/online-simulations.ipynb cell 25).
// All spectra carry pyqg's 1/M^2 normalisation.  PARITY UNPINNED (pyqg is not available here):
// checked against oracle/qg_ref.py::_diag_functions only.
#include "common.hpp"
#include "diag_acc.hpp"
#include <cstdlib>

namespace qgx {
int large_q_to_qh(qgx_model *m, const double *q, double2 *qh, hipStream_t st);
int large_qh_to_q(qgx_model *m, const double2 *qh, double *q, hipStream_t st);
int large_invert(qgx_model *m, hipStream_t st);
int large_ensure_q(qgx_model *m, hipStream_t st);
bool large_diag_fused_ok(const qgx_model *m);
int large_diag_fused(qgx_model *m, const DiagConst &c, const double2 *qh, const double2 *Sh, const double2 *dq_p, const double2 *dq_pp,
                     const DiagAcc &acc, hipStream_t st);
int small_diag_increment(const SpecDev &d, const DiagConst &c, const double2 *qh, double2 *ph, double *u, double *v, double *P,
                         double *XI, double2 *S3, double2 *S4, double2 *S5, double2 *Sh, double2 *S6, double2 *S7, const double *S,
                         double weight, const double *q, const double2 *dq_p, const double2 *dq_pp, const DiagAcc &a, hipStream_t st);


bool small_diag_increment_reg_ok(const SpecDev &d);
int small_diag_increment_reg(const SpecDev &d, const DiagConst &c, const double2 *qh, double2 *ph, const double *S, double weight, const double *q,
                             const double2 *dq_p, const double2 *dq_pp, const DiagAcc &a, hipStream_t st, bool halves);

int small_diag_transforms_wide(const SpecDev &d, const DiagConst &c, const double2 *qh, double2 *ph, double *u, double *v, double *P,
                               double *XI, double2 *S3, double2 *S4, double2 *S5, double2 *Sh, double2 *S6, double2 *S7, const double *S,
                               double weight, const double *q, hipStream_t st);

// xih_k = -wv2 * ph_k
__global__ void k_diag_xih(SpecDev d, const double2 *ph, double2 *xih) {
    const int sz = d.N * d.NK, b = blockIdx.y;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 2 * sz; idx += gridDim.x * blockDim.x) {
        const size_t o = (size_t)b * 2 * sz + idx;
        const double w = -d.wv2[idx % sz];
        const double2 p = ph[o];
        xih[o] = make_double2(w * p.x, w * p.y);
    }
}

// real-space products: R3 = [ub*ptpc, vb*ptpc], R4 = [u1*xi1, v1*xi1], R5 = [u2*xi2, v2*xi2], R6 = [u1*q1, v1*q1],
// R7 = [u2*q2, v2*q2]
__global__ void k_diag_products(SpecDev d, DiagConst c, const double *u, const double *v, const double *p,
                                const double *xi, const double *q, double *R3, double *R4, double *R5, double *R6, double *R7) {
    const int rz = d.N * d.N, b = blockIdx.y;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < rz; idx += gridDim.x * blockDim.x) {
        const size_t o = (size_t)b * 2 * rz + idx;
        const double u1 = u[o], u2 = u[o + rz], v1 = v[o], v2 = v[o + rz];
        const double ptpc = p[o] - p[o + rz];
        const double ub = c.del1 * u1 + c.del2 * u2, vb = c.del1 * v1 + c.del2 * v2;
        R3[o] = ub * ptpc; R3[o + rz] = vb * ptpc;
        const double x1 = xi[o], x2 = xi[o + rz];
        R4[o] = u1 * x1; R4[o + rz] = v1 * x1;
        R5[o] = u2 * x2; R5[o + rz] = v2 * x2;
        const double q1 = q[o], q2 = q[o + rz];
        R6[o] = u1 * q1; R6[o + rz] = v1 * q1;
        R7[o] = u2 * q2; R7[o + rz] = v2 * q2;
    }
}

__global__ void k_diag_accumulate(SpecDev d, DiagConst c, const double2 *qh, const double2 *ph, const double2 *S3,
                                  const double2 *S4, const double2 *S5, const double2 *Sh, const double2 *S6, const double2 *S7,
                                  const double2 *dq_p, const double2 *dq_pp, DiagAcc a) {
    const int N = d.N, NK = d.NK, sz = N * NK, b = blockIdx.y;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < sz; idx += gridDim.x * blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        const size_t o = (size_t)b * 2 * sz + idx, o2 = (size_t)b * sz + idx;
        const double2 zero = make_double2(0., 0.);
        diag_accumulate_elem(d, c, a, idx, i, j, o, o2, sz, qh[o], qh[o + sz], ph[o], ph[o + sz], S3[o], S3[o + sz], S4[o], S4[o + sz],
                             S5[o], S5[o + sz], Sh != nullptr, Sh ? Sh[o] : zero, Sh ? Sh[o + sz] : zero, S6[o], S6[o + sz], S7[o],
                             S7[o + sz], dq_p[o], dq_p[o + sz], dq_pp[o], dq_pp[o + sz]);
    }
}

__global__ void k_diag_scale_copy(const double *src, double *dst, size_t n, double s) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i] * s;
}
__global__ void k_diag_scale_S(const double *src, double *dst, size_t n, double w) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = w * src[i];
}

static dim3 dgrid(const SpecDev &d, int n) { return dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), d.B); }

static int dalloc0(double *&p, size_t n) {
    QGX_HIP(hipMalloc((void **)&p, n * sizeof(double)));
    QGX_HIP(hipMemset(p, 0, n * sizeof(double)));
    return QGX_OK;
}

int diag_ensure_alloc(qgx_model *m) {
    const SpecDev &d = m->d;
    const size_t nr = (size_t)d.B * 2 * d.N * d.N, ns2 = (size_t)d.B * 2 * d.N * d.NK * 2, n2d = (size_t)d.B * d.N * d.NK;
    int rc;
    if (!m->dg_R[0]) {
        for (int i = 0; i < 7; ++i) if ((rc = dalloc0(m->dg_R[i], nr))) return rc;
        for (int i = 0; i < 7; ++i) if ((rc = dalloc0(m->dg_S[i], ns2))) return rc;
        for (int i = 0; i < 2; ++i) if ((rc = dalloc0(m->dg_acc[i], ns2 / 2))) return rc;
        for (int i = 2; i < N_DIAGS; ++i) if ((rc = dalloc0(m->dg_acc[i], n2d))) return rc;
    }
    return QGX_OK;
}

int diag_increment(qgx_model *m, const double *S, double weight, hipStream_t st) {
    const SpecDev &d = m->d;
    const size_t nr = (size_t)d.B * 2 * d.N * d.N;
    int rc;
    if ((rc = diag_ensure_alloc(m))) return rc;
    double2 *qh = m->qh[m->cur_q];
    double *p = m->dg_R[0], *xi = m->dg_R[1], *R3 = m->dg_R[2], *R4 = m->dg_R[3], *R5 = m->dg_R[4];
    double2 *xih = (double2 *)m->dg_S[0], *S3 = (double2 *)m->dg_S[1], *S4 = (double2 *)m->dg_S[2],
            *S5 = (double2 *)m->dg_S[3], *Sh = (double2 *)m->dg_S[4], *S6 = (double2 *)m->dg_S[5], *S7 = (double2 *)m->dg_S[6];
    DiagConst c;
    c.del1 = m->cfg.delta / (m->cfg.delta + 1.); c.del2 = 1. / (m->cfg.delta + 1.);
    c.rdm2 = pow(m->cfg.rd, -2.0); c.Udiff = m->cfg.U1 - m->cfg.U2; c.rek = m->cfg.rek;
    c.invM2 = d.invN2 * d.invN2; c.H0 = d.H[0] / d.Htot; c.H1 = d.H[1] / d.Htot;
    {   // the AB coefficients of the step this increment precedes (kernel.pyx::_forward_timestep; model.hip::model_step_once)
        const double dt = m->cfg.dt;
        if (m->ablevel == 0) { c.dt1 = dt; c.dt2 = 0.0; c.dt3 = 0.0; }
        else if (m->ablevel == 1) { c.dt1 = 1.5 * dt; c.dt2 = -0.5 * dt; c.dt3 = 0.0; }
        else { c.dt1 = 23. / 12. * dt; c.dt2 = -16. / 12. * dt; c.dt3 = 5. / 12. * dt; }
        c.invdt = 1.0 / dt;
    }
    const double2 *dq_p = m->dq[m->i_new], *dq_pp = m->dq[m->i_p];       // T_{n-1}, T_{n-2}
    DiagAcc a;
    a.KEspec = m->dg_acc[0]; a.Ensspec = m->dg_acc[1]; a.entspec = m->dg_acc[2]; a.APEflux = m->dg_acc[3];
    a.KEflux = m->dg_acc[4]; a.APEgenspec = m->dg_acc[5]; a.KEfrictionspec = m->dg_acc[6]; a.paramspec = m->dg_acc[7];
    a.paramspec_APEflux = m->dg_acc[8]; a.paramspec_KEflux = m->dg_acc[9];
    a.Dissspec = m->dg_acc[10]; a.ENSDissspec = m->dg_acc[11]; a.ENSflux = m->dg_acc[12]; a.ENSgenspec = m->dg_acc[13];
    a.ENSfrictionspec = m->dg_acc[14]; a.ENSparamspec = m->dg_acc[15];
    if (m->small && m->opts.diag_fused && (m->opts.diag_wide > 0 || (m->opts.diag_wide < 0 && 6 * d.B <= 256))) {
        // few members: the ten transforms as (member, transform) workgroups — two launches — then the accumulation kernel;
        // a single member's increment is three short kernels instead of a chain of ten transforms on one CU
        rc = small_diag_transforms_wide(d, c, qh, m->ph, m->u, m->v, p, xi, S3, S4, S5, Sh, S6, S7, S, weight, m->q, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_diag_accumulate, dgrid(d, d.N * d.NK), dim3(256), 0, st, d, c, (const double2 *)qh,
                           (const double2 *)m->ph, (const double2 *)S3, (const double2 *)S4, (const double2 *)S5,
                           S ? (const double2 *)Sh : (const double2 *)nullptr, (const double2 *)S6, (const double2 *)S7, dq_p, dq_pp, a);
        QGX_HIP(hipGetLastError());
        m->uv_stale = false;
        m->dg_count += 1;
        return QGX_OK;
    }
    if (m->small && m->opts.diag_fused && m->opts.diag_reg && small_diag_increment_reg_ok(d)) {
        // grids up to 64 x 64: the same in ONE kernel whose work fields stay in registers (k_diag_small_reg: a quarter of the
        // bytes); it stores ph but no u, v — they are marked stale and inverted on demand
        // ... as two workgroups per member (chains of 7 and 5 transforms instead of one of 10) while both fit the device at once
        const bool halves = m->opts.diag_reg == 3 || (m->opts.diag_reg == 1 && 2 * d.B <= 256);
        rc = small_diag_increment_reg(d, c, qh, m->ph, S, weight, m->q, dq_p, dq_pp, a, st, halves);
        if (rc) return rc;
        m->uv_stale = true;
        m->dg_count += 1;
        return QGX_OK;
    }
    if (m->small && m->opts.diag_fused) {
        // small grids: the whole increment (inversion, ten packed transforms, products, accumulation) in ONE kernel, a
        // workgroup per member (spectral_small.hip::k_diag_small) instead of a dozen launches
        rc = small_diag_increment(d, c, qh, m->ph, m->u, m->v, p, xi, S3, S4, S5, Sh, S6, S7, S, weight, m->q, dq_p, dq_pp, a, st);
        if (rc) return rc;
        m->uv_stale = false;
        m->dg_count += 1;
        return QGX_OK;
    }
    if (!m->small && large_diag_fused_ok(m)) {
        // large grids at the specialised sizes: the increment's four packed fields through three fused launches
        // (spectral_large.hip); ph is stored, u and v are not (uv_stale stays as it is: they are inverted on demand)
        const double2 *Shp = nullptr;
        if (S) {
            hipLaunchKernelGGL(k_diag_scale_S, dim3(1024), dim3(256), 0, st, S, R3, nr, weight);
            if ((rc = large_q_to_qh(m, R3, Sh, st))) return rc;
            Shp = Sh;
        }
        if ((rc = large_diag_fused(m, c, qh, Shp, dq_p, dq_pp, a, st))) return rc;
        m->dg_count += 1;
        return QGX_OK;
    }
    // _invert: ph, u, v of the current state
    rc = m->small ? small_invert(d, m->opts, qh, m->ph, m->u, m->v, st) : large_invert(m, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_diag_xih, dgrid(d, 2 * d.N * d.NK), dim3(256), 0, st, d, (const double2 *)m->ph, xih);
    auto inv = [&](const double2 *h, double *r) { return m->small ? small_qh_to_q(d, m->opts, h, r, st) : large_qh_to_q(m, h, r, st); };
    auto fwd = [&](const double *r, double2 *h) { return m->small ? small_q_to_qh(d, m->opts, r, h, st) : large_q_to_qh(m, r, h, st); };
    if ((rc = inv(m->ph, p)) || (rc = inv(xih, xi))) return rc;
    if (!m->small && (rc = large_ensure_q(m, st))) return rc;          // the lazy unparameterized path keeps no real-space q
    double *R6 = m->dg_R[5], *R7 = m->dg_R[6];
    hipLaunchKernelGGL(k_diag_products, dgrid(d, d.N * d.N), dim3(256), 0, st, d, c, (const double *)m->u,
                       (const double *)m->v, (const double *)p, (const double *)xi, (const double *)m->q, R3, R4, R5, R6, R7);
    if ((rc = fwd(R3, S3)) || (rc = fwd(R4, S4)) || (rc = fwd(R5, S5)) || (rc = fwd(R6, S6)) || (rc = fwd(R7, S7))) return rc;
    const double2 *Shp = nullptr;
    if (S) {
        hipLaunchKernelGGL(k_diag_scale_S, dim3(1024), dim3(256), 0, st, S, R3, nr, weight);
        if ((rc = fwd(R3, Sh))) return rc;
        Shp = Sh;
    }
    hipLaunchKernelGGL(k_diag_accumulate, dgrid(d, d.N * d.NK), dim3(256), 0, st, d, c, (const double2 *)qh,
                       (const double2 *)m->ph, (const double2 *)S3, (const double2 *)S4, (const double2 *)S5, Shp,
                       (const double2 *)S6, (const double2 *)S7, dq_p, dq_pp, a);
    QGX_HIP(hipGetLastError());
    m->dg_count += 1;
    return QGX_OK;
}

}  // namespace qgx

using namespace qgx;

extern "C" int qgx_diag_config(qgx_model *m, int64_t start_step, int every) {
    QGX_NEEDS_STATE(m, "qgx_diag_config");
    QGX_REQUIRE(m, "qgx_diag_config: null model");
    m->dg_start = start_step;
    m->dg_every = every;
    return QGX_OK;
}

extern "C" int64_t qgx_diag_count(const qgx_model *m) { return m ? m->dg_count : -1; }

extern "C" int qgx_diag_reset(qgx_model *m) {
    QGX_REQUIRE(m, "qgx_diag_reset: null model");
    m->dg_count = 0;
    if (m->dg_acc[0]) {
        const size_t ns = (size_t)m->B * 2 * m->N * m->NK, n2 = (size_t)m->B * m->N * m->NK;
        for (int i = 0; i < N_DIAGS; ++i) QGX_HIP(hipMemset(m->dg_acc[i], 0, (i < 2 ? ns : n2) * sizeof(double)));
    }
    return QGX_OK;
}

extern "C" int qgx_diag_get(qgx_model *m, int diag, double *out_dev, void *stream) {
    QGX_NEEDS_STATE(m, "qgx_diag_get");
    QGX_REQUIRE(m && out_dev && diag >= 0 && diag < N_DIAGS, "qgx_diag_get: bad argument");
    QGX_REQUIRE(m->dg_count > 0 && m->dg_acc[0], "qgx_diag_get: no diagnostics accumulated yet");
    const size_t n = (size_t)m->B * (diag < 2 ? 2 : 1) * m->N * m->NK;
    hipLaunchKernelGGL(k_diag_scale_copy, dim3(1024), dim3(256), 0, (hipStream_t)stream,
                       (const double *)m->dg_acc[diag], out_dev, n, 1.0 / (double)m->dg_count);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}
