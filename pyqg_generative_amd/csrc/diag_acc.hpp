// One spectral element of model.py::_increment_diagnostics (shared by diag.hip::k_diag_accumulate and
// spectral_small.hip::k_diag_small): the sixteen time-averaged diagnostics, pyqg's 1/M^2 normalisation.
#pragma once
#include "common.hpp"

namespace qgx {

// PARTS: which diagnostics this call updates (k_diag_small_reg accumulates the ones fed by a product transform as soon as that
// transform is in LDS: the expressions are these very ones, so the accumulators come out bit-identical): 1 = APEflux (needs the
// spectra A3, B3), 2 = KEflux (A4, B4, A5, B5), 4 = all the others; 7 = everything
// JPRE: A4 and A6 carry the layer-1 Jacobians (j1x, j1y) and (g1x, g1y) already formed by diag_jacobian() — the same two
// expressions — instead of the spectra they are formed from (B4, B6 unused): half the registers to hold across a transform
// (explicit fused multiply-adds: written as kx A + ly B it is the compiler's choice which of the two products it fuses, and it
//  chooses differently in different kernels — a last-bit difference between the variants of the increment)
__device__ __forceinline__ double2 diag_jacobian(double kx, double ly, double2 A, double2 B) {
    return make_double2(-__builtin_fma(kx, A.y, ly * B.y), __builtin_fma(kx, A.x, ly * B.x));
}
__device__ __forceinline__ double diag_dot(double2 a, double bx, double by) { return __builtin_fma(a.x, bx, a.y * by); }
template <int PARTS = 7, bool JPRE = false>
__device__ __forceinline__ void diag_accumulate_elem(const SpecDev &d, const DiagConst &c, const DiagAcc &a, int idx, int i, int j,
                                                     size_t o, size_t o2, int sz, double2 q1, double2 q2, double2 p1, double2 p2,
                                                     double2 A3, double2 B3, double2 A4, double2 B4, double2 A5, double2 B5,
                                                     bool has_S, double2 s1, double2 s2, double2 A6, double2 B6, double2 A7,
                                                     double2 B7, double2 hp1, double2 hp2, double2 hpp1, double2 hpp2) {
    const double kx = d.kk[i], ly = d.ll[j], wv2 = d.wv2[idx];
    const double dpx = p1.x - p2.x, dpy = p1.y - p2.y;
    if (PARTS & 4) {
        a.KEspec[o] += wv2 * (p1.x * p1.x + p1.y * p1.y) * c.invM2;
        a.KEspec[o + sz] += wv2 * (p2.x * p2.x + p2.y * p2.y) * c.invM2;
        a.Ensspec[o] += (q1.x * q1.x + q1.y * q1.y) * c.invM2;
        a.Ensspec[o + sz] += (q2.x * q2.x + q2.y * q2.y) * c.invM2;
        const double ex = c.del1 * q1.x + c.del2 * q2.x, ey = c.del1 * q1.y + c.del2 * q2.y;
        a.entspec[o2] += (ex * ex + ey * ey) * c.invM2;
    }
    if (PARTS & 1) {
        // Jptpc = -(ik A + il B), (A,B) = S3
        const double jx = (kx * A3.y + ly * B3.y), jy = -(kx * A3.x + ly * B3.x);
        a.APEflux[o2] += c.rdm2 * c.del1 * c.del2 * (dpx * jx + dpy * jy) * c.invM2;
    }
    if (PARTS & 2) {
        // Jpxi_k = ik A + il B
        const double2 j1 = JPRE ? A4 : diag_jacobian(kx, ly, A4, B4), j2 = diag_jacobian(kx, ly, A5, B5);
        const double j1x = j1.x, j1y = j1.y, j2x = j2.x, j2y = j2.y;
        a.KEflux[o2] += __builtin_fma(c.del1, diag_dot(p1, j1x, j1y), c.del2 * diag_dot(p2, j2x, j2y)) * c.invM2;
    }
    if (!(PARTS & 4)) return;
    // APEgenspec = U rd^-2 del1 del2 Re[ i k (del1 p1 + del2 p2) conj(p1 - p2) ]
    const double bx = c.del1 * p1.x + c.del2 * p2.x, by = c.del1 * p1.y + c.del2 * p2.y;
    // i k (bx + i by) = (-k by, k bx); Re[(.)*conj(dp)] = (-k by) dpx + (k bx) dpy
    a.APEgenspec[o2] += c.Udiff * c.rdm2 * c.del1 * c.del2 * kx * (bx * dpy - by * dpx) * c.invM2;
    a.KEfrictionspec[o2] += -c.rek * c.del2 * wv2 * (p2.x * p2.x + p2.y * p2.y) * c.invM2;
    if (has_S) {
        // -Re[ sum_k Hk/H conj(ph_k) dqh_k ]
        a.paramspec[o2] += -(c.H0 * (p1.x * s1.x + p1.y * s1.y) + c.H1 * (p2.x * s2.x + p2.y * s2.y)) * c.invM2;
        // its split into the available-potential and kinetic parts of the energy budget: with the streamfunction
        // tendency of the parameterization dph = A dqh (the model's inversion),
        //   paramspec_APEflux = rd^-2 del1 del2 Re[(p1 - p2) conj(dp1 - dp2)] / M^2
        //   paramspec_KEflux  = wv2 sum_k del_k Re[p_k conj(dp_k)] / M^2,      APEflux + KEflux == paramspec
        const double a00 = d.a[idx], a01 = d.a[sz + idx], a10 = d.a[2 * sz + idx], a11 = d.a[3 * sz + idx];
        const double d1x = a00 * s1.x + a01 * s2.x, d1y = a00 * s1.y + a01 * s2.y;
        const double d2x = a10 * s1.x + a11 * s2.x, d2y = a10 * s1.y + a11 * s2.y;
        a.paramspec_APEflux[o2] += c.rdm2 * c.del1 * c.del2 * (dpx * (d1x - d2x) + dpy * (d1y - d2y)) * c.invM2;
        a.paramspec_KEflux[o2] += wv2 * (c.del1 * (p1.x * d1x + p1.y * d1y) + c.del2 * (p2.x * d2x + p2.y * d2y)) * c.invM2;
        // ENSparamspec = Re[ sum_k Hk/H conj(qh_k) dqh_k ]
        a.ENSparamspec[o2] += (c.H0 * (q1.x * s1.x + q1.y * s1.y) + c.H1 * (q2.x * s2.x + q2.y * s2.y)) * c.invM2;
    }
    // ---- the barotropic-enstrophy budget and the filter's dissipation (pyqg model.py::_initialize_core_diagnostics,
    // qg_model.py::_initialize_model_diagnostics): every term is Re[sum_k Hk/H conj(qh_k) X_k] / M^2 for one term X_k of
    // the PV tendency, so that they close the budget of sum_k Hk/H |qh_k|^2 / 2 wavenumber by wavenumber.
    // Jq_k = ik A + il B, (A, B) = transforms of (u_k q_k, v_k q_k) with the PERTURBATION velocities (model.py::_advect)
    const double2 g1 = JPRE ? A6 : diag_jacobian(kx, ly, A6, B6), g2 = diag_jacobian(kx, ly, A7, B7);
    const double g1x = g1.x, g1y = g1.y, g2x = g2.x, g2y = g2.y;
    a.ENSflux[o2] += -__builtin_fma(c.H0, diag_dot(q1, g1x, g1y), c.H1 * diag_dot(q2, g2x, g2y)) * c.invM2;
    // -Re[conj(qh_k) ik Qy_k ph_k] = -k Qy_k (q.y p.x - q.x p.y)
    a.ENSgenspec[o2] += -kx * (c.H0 * d.Qy[0] * (q1.y * p1.x - q1.x * p1.y) + c.H1 * d.Qy[1] * (q2.y * p2.x - q2.x * p2.y)) * c.invM2;
    a.ENSfrictionspec[o2] += c.rek * c.H1 * wv2 * (q2.x * p2.x + q2.y * p2.y) * c.invM2;
    // the tendency the step is about to use: T_k = -(Jq_k + ik U_k qh_k) - ik Qy_k ph_k [+ rek wv2 ph_2] [+ dqh_k], and what
    // the exponential filter removes from the unfiltered AB update: diss_k = (filtr - 1) (qh_k + dt1 T_k + dt2 T'_k + dt3 T''_k)
    double t1x = -(g1x - kx * d.U[0] * q1.y) + kx * d.Qy[0] * p1.y, t1y = -(g1y + kx * d.U[0] * q1.x) - kx * d.Qy[0] * p1.x;
    double t2x = -(g2x - kx * d.U[1] * q2.y) + kx * d.Qy[1] * p2.y, t2y = -(g2y + kx * d.U[1] * q2.x) - kx * d.Qy[1] * p2.x;
    if (c.rek != 0.0) { t2x += c.rek * wv2 * p2.x; t2y += c.rek * wv2 * p2.y; }
    if (has_S) { t1x += s1.x; t1y += s1.y; t2x += s2.x; t2y += s2.y; }
    const double fm1 = d.filtr[idx] - 1.0;
    const double e1x = fm1 * (q1.x + c.dt1 * t1x + c.dt2 * hp1.x + c.dt3 * hpp1.x), e1y = fm1 * (q1.y + c.dt1 * t1y + c.dt2 * hp1.y + c.dt3 * hpp1.y);
    const double e2x = fm1 * (q2.x + c.dt1 * t2x + c.dt2 * hp2.x + c.dt3 * hpp2.x), e2y = fm1 * (q2.y + c.dt1 * t2y + c.dt2 * hp2.y + c.dt3 * hpp2.y);
    a.Dissspec[o2] += -(c.H0 * (p1.x * e1x + p1.y * e1y) + c.H1 * (p2.x * e2x + p2.y * e2y)) * c.invdt * c.invM2;
    a.ENSDissspec[o2] += (c.H0 * (q1.x * e1x + q1.y * e1y) + c.H1 * (q2.x * e2x + q2.y * e2y)) * c.invdt * c.invM2;
}

}  // namespace qgx
