// Host side of the C ABI for the spectral model: tables, state, stepping loop.
// Restates pyqg 0.7.2 model.py::{_initialize_grid,_initialize_filter,_step_forward},
// qg_model.py::{_initialize_background,_initialize_inversion_matrix,_calc_cfl,_calc_ke}
// (constructed by the reference at pyqg_generative/tools/simulate.py:83,121 and
// tools/stochastic_pyqg.py:78-88).
#include "common.hpp"
#include "fft_lds.hpp"
#include <cmath>
#include <new>

namespace qgx {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// spectral_large.hip
int large_prepare(const SpecDev &d);
int large_q_to_qh(qgx_model *m, const double *q, double2 *qh, hipStream_t st);
int large_qh_to_q(qgx_model *m, const double2 *qh, double *q, hipStream_t st);
int large_invert(qgx_model *m, hipStream_t st);
int large_step(qgx_model *m, const StepArgs &a, hipStream_t st);
int large_ensure_q(qgx_model *m, hipStream_t st);
int large_zpad();
int large_team_available(qgx_model *m, hipStream_t st);
int large_team_check(qgx_model *m, hipStream_t st, unsigned *flag);
int large_team_steps(qgx_model *m, int K, int ablevel0, const double coef[3][3], const double2 *qh_src, double2 *qh_dst,
                     const double2 *p_src, const double2 *pp_src, double2 *p_dst, double2 *pp_dst, hipStream_t st);

// One _step_forward: AB3 coefficient schedule of kernel.pyx::_forward_timestep, history rotation.
// the arguments of step n that do not depend on its forcing; advances the AB3 start-up level
static void step_args_begin(qgx_model *m, int diag, StepArgs &a) {
    const double dt = m->cfg.dt;
    if (m->ablevel == 0) { a.dt1 = dt; a.dt2 = 0.0; a.dt3 = 0.0; m->ablevel = 1; }
    else if (m->ablevel == 1) { a.dt1 = 1.5 * dt; a.dt2 = -0.5 * dt; a.dt3 = 0.0; m->ablevel = 2; }
    else { a.dt1 = 23. / 12. * dt; a.dt2 = -16. / 12. * dt; a.dt3 = 5. / 12. * dt; }
    // Slots after step n-1: i_new = T_{n-1}, i_p = T_{n-2}, i_pp = dead.  T_n goes to the dead slot.
    a.qh_in = m->qh[m->cur_q];
    a.qh_out = m->qh[m->cur_q ^ 1];
    a.q = m->q;
    a.dqh = m->dqh;
    a.dq_new = m->dq[m->i_pp];
    a.dq_p = m->dq[m->i_new];
    a.dq_pp = m->dq[m->i_p];
    a.ph = m->ph; a.u = m->u; a.v = m->v;
    a.diag = diag;
}

// pre: the first half of this step (k_step_small PART 1) is already in flight on the model's side stream with these
// arguments (step_core); the second half is ordered behind it here
static int model_step_once(qgx_model *m, bool has_S, const double *S, double weight, int demean, int diag,
                           hipStream_t st, const GenFuse *gf = nullptr, const StepArgs *pre = nullptr) {
    StepArgs a;
    if (pre) a = *pre; else step_args_begin(m, diag, a);
    if (gf) a.gf = *gf;
    a.S = S;
    a.weight = weight;
    a.has_S = has_S ? 1 : 0; a.demean = demean;
    int rc;
    if (pre) {
        QGX_HIP(hipStreamWaitEvent(st, m->adv_event[m->adv_slot][1], 0));
        rc = small_step(m->d, m->opts, a, st, 2);
    } else if (m->small && has_S && m->sib_flag && m->opts.siblings != 0 && small_layer_split(m->d, m->opts) &&
               (m->opts.siblings >= 1 || 4 * m->B * (m->is_half ? 2 : 1) <= 256)) {   // (two halves run side by side: both must fit)    // while all four workgroups of every member are resident at once
        // the forcing's transform on a workgroup of its own beside the inversion / advection chain (k_step_small PART 3)
        a.sib_flag = m->sib_flag;
        a.sib_epoch = ++m->sib_epoch;
        a.sib_full = m->opts.siblings == 2;
        rc = small_step(m->d, m->opts, a, st, 3);
    } else {
        rc = m->small ? small_step(m->d, m->opts, a, st) : large_step(m, a, st);
    }
    if (rc) return rc;
    const int dead = m->i_pp;
    m->i_pp = m->i_p; m->i_p = m->i_new; m->i_new = dead;
    m->cur_q ^= 1;
    m->tc += 1;
    m->uv_stale = diag == 0;
    return QGX_OK;
}

// K unparameterized steps without diagnostics output in one launch of the XCD-resident kernel (spectral_large.hip):
// the same AB3 schedule and history rotation as K calls of model_step_once.  The run is a TRANSACTION: it reads the
// live buffers (qh[cur], dq[i_new], dq[i_p]) and writes only buffers that hold nothing live (qh[cur ^ 1], dq[i_pp],
// dq[i_x]), and the bookkeeping before it is kept in team_undo until its flag word has been read back (team_settle).
static int model_step_run(qgx_model *m, int K, hipStream_t st) {
    const double dt = m->cfg.dt;
    const double coef[3][3] = {{dt, 0.0, 0.0}, {1.5 * dt, -0.5 * dt, 0.0}, {23. / 12. * dt, -16. / 12. * dt, 5. / 12. * dt}};
    qgx_model::TeamUndo &u = m->team_undo;
    u.K = K; u.cur_q = m->cur_q; u.i_new = m->i_new; u.i_p = m->i_p; u.i_pp = m->i_pp; u.i_x = m->i_x;
    u.tc = m->tc; u.ablevel = m->ablevel; u.uv_stale = m->uv_stale; u.q_stale = m->q_stale;
    // before: i_new = T_{n-1}, i_p = T_{n-2}; after: T_{n+K-1} in the old i_pp, T_{n+K-2} in the old spare slot
    int rc = large_team_steps(m, K, m->ablevel, coef, m->qh[m->cur_q], m->qh[m->cur_q ^ 1], m->dq[m->i_new],
                              m->dq[m->i_p], m->dq[m->i_pp], m->dq[m->i_x], st);
    if (rc) { u.K = 0; return rc; }
    m->i_new = u.i_pp; m->i_p = u.i_x; m->i_pp = u.i_new; m->i_x = u.i_p;
    m->cur_q ^= 1;
    m->tc += K;
    m->ablevel = m->ablevel + K > 2 ? 2 : m->ablevel + K;
    m->uv_stale = true;
    return QGX_OK;
}

// Settles the pending run of the XCD-resident kernel, if any (one stream synchronisation).  A run that raised a flag
// (its bounded waits timed out because other work held CUs, or its workgroups were not co-resident) left its inputs
// intact: the bookkeeping is restored, the kernel is switched off for this model and the K steps are replayed on the
// three-launch path — a long run degrades, it is not lost.  qgx_run_kernel_state() reports -1 afterwards.
static int team_settle(qgx_model *m, hipStream_t caller) {
    if (!m || m->small || !m->team_pending) return QGX_OK;
    unsigned flag = 0;
    hipStream_t st = m->team_stream;     // the run's own stream: the read-back and the replay are ordered behind it
    int rc = large_team_check(m, st, &flag);
    if (rc) return rc;
    const qgx_model::TeamUndo u = m->team_undo;
    m->team_undo.K = 0;
    if (flag == 0) return QGX_OK;
    m->team_state = -1;
    m->team_replays += 1;
    QGX_REQUIRE(u.K > 0, "XCD-resident step kernel raised flag %u and left no undo record: the model state is undefined", flag);
    m->cur_q = u.cur_q; m->i_new = u.i_new; m->i_p = u.i_p; m->i_pp = u.i_pp; m->i_x = u.i_x;
    m->tc = u.tc; m->ablevel = u.ablevel; m->uv_stale = u.uv_stale; m->q_stale = u.q_stale;
    for (int s = 0; s < u.K; ++s)
        if ((rc = model_step_once(m, false, nullptr, 1.0, 0, 0, st))) return rc;
    // the entry point that settles may work on another stream than the run was launched on: what it enqueues next (a copy out
    // of the state, a kernel that overwrites it) must not overtake the replayed steps.  A replay is the rare path: wait for it.
    if (caller != st) QGX_HIP(hipStreamSynchronize(st));
    return QGX_OK;
}

static bool factor_radices(int N, int *rad, int &nrad) {
    // the same greedy plan as the device's compile-time pick_radix (fft_lds.hpp)
    nrad = 0;
    int n = N;
    while (n > 1 && nrad < MAX_RADIX_PASSES) {
        const int r = pick_radix(n);
        if (n % r) return false;
        rad[nrad++] = r;
        n /= r;
    }
    return n == 1;
}

template <class T>
static int upload(T *&dst, const std::vector<T> &h) {
    QGX_HIP(hipMalloc((void **)&dst, h.size() * sizeof(T)));
    QGX_HIP(hipMemcpy(dst, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return QGX_OK;
}
template <class T>
static int dalloc(T *&dst, size_t n) {
    QGX_HIP(hipMalloc((void **)&dst, n * sizeof(T)));
    QGX_HIP(hipMemset(dst, 0, n * sizeof(T)));
    return QGX_OK;
}

// ---- status reductions: pyqg model.py::_print_status -> qg_model._calc_ke / _calc_cfl
__global__ void k_status(SpecDev d, const double2 *ph, const double *u, const double *v, double *out) {
    __shared__ double s_ke[16], s_mx[16];
    const int N = d.N, NK = d.NK, b = blockIdx.x;
    const int sz = N * NK, rz = N * N;
    const double2 *p = ph + (size_t)b * 2 * sz;
    double ke = 0.0;
    for (int idx = threadIdx.x; idx < 2 * sz; idx += blockDim.x) {
        const int k = idx / sz, r = idx - k * sz;
        const int i = r % NK;
        const double2 c = p[idx];
        double w = 2.0 * d.wv2[r] * (c.x * c.x + c.y * c.y) * d.invN2 * d.invN2;  // 2|wv ph|^2 / M^2
        if (i == 0 || i == NK - 1) w *= 0.5;
        ke += 0.5 * d.H[k] * w;
    }
    double mx = 0.0;
    const double *uu = u + (size_t)b * 2 * rz, *vv = v + (size_t)b * 2 * rz;
    for (int idx = threadIdx.x; idx < 2 * rz; idx += blockDim.x) {
        const int k = idx / rz;
        mx = fmax(mx, fmax(fabs(uu[idx] + d.U[k]), fabs(vv[idx])));
    }
    for (int o = 32; o > 0; o >>= 1) {
        ke += __shfl_down(ke, o);
        mx = fmax(mx, __shfl_down(mx, o));
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { s_ke[wave] = ke; s_mx[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tk = 0, tm = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { tk += s_ke[w]; tm = fmax(tm, s_mx[w]); }
        out[2 * b] = tk / d.Htot;
        out[2 * b + 1] = tm * d.dt / d.dx;
    }
}

}  // namespace qgx

using namespace qgx;

extern "C" const char *qgx_last_error(void) { return g_err; }
#ifndef QGX_SOURCE_HASH
#define QGX_SOURCE_HASH "unknown"
#endif
#ifdef QGX_AB
extern "C" const char *qgx_version(void) { return "qgx 0.2 (gfx950) src " QGX_SOURCE_HASH " +ab"; }   // A/B library: every kernel variant
#else
extern "C" const char *qgx_version(void) { return "qgx 0.2 (gfx950) src " QGX_SOURCE_HASH; }
#endif

extern "C" int qgx_create(const qgx_config *cfg, qgx_model **out) {
    QGX_REQUIRE(cfg && out, "qgx_create: null argument");
    const int N = cfg->nx;
    QGX_REQUIRE(N >= 8 && N <= 512 && N % 2 == 0, "qgx_create: nx=%d unsupported (even, 8..512)", N);
    QGX_REQUIRE(cfg->n_members >= 1, "qgx_create: n_members must be >= 1");
    int rad[MAX_RADIX_PASSES], nrad;
    QGX_REQUIRE(factor_radices(N, rad, nrad), "qgx_create: nx=%d is not of the form 2^a 3^b", N);
    QGX_HIP(hipSetDevice(cfg->device));

    qgx_model *m = new (std::nothrow) qgx_model();
    if (!m) { set_error("out of host memory"); return QGX_ERR_NOMEM; }
    m->cfg = *cfg;
    m->N = N; m->NK = N / 2 + 1; m->B = cfg->n_members;
    const int NK = m->NK, B = m->B;
    const double pi = 3.14159265358979323846;
    const double L = cfg->L, W = cfg->L;

    // pyqg model.py::_initialize_grid
    const double dk = 2. * pi / L, dl = 2. * pi / W;
    m->host = std::make_shared<qgx_model::HostTables>();
    m->host->kk.resize(NK); m->host->ll.resize(N);
    for (int i = 0; i < NK; ++i) m->host->kk[i] = dk * (double)i;
    for (int j = 0; j < N; ++j) m->host->ll[j] = dl * (double)(j < N / 2 ? j : j - N);
    const double dx = L / N, dy = W / N;
    m->host->wv2.resize((size_t)N * NK); m->host->filtr.resize((size_t)N * NK); m->host->a.resize((size_t)4 * N * NK);
    // pyqg qg_model.py::_initialize_background
    const double F1 = pow(cfg->rd, -2.0) / (1. + cfg->delta);
    const double F2 = cfg->delta * F1;
    const double Qy1 = cfg->beta + F1 * (cfg->U1 - cfg->U2);
    const double Qy2 = cfg->beta - F2 * (cfg->U1 - cfg->U2);
    const double cphi = 0.65 * pi;
    const size_t sz = (size_t)N * NK;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < NK; ++i) {
            const double k = m->host->kk[i], l = m->host->ll[j];
            const double wv2 = k * k + l * l;
            const size_t o = (size_t)j * NK + i;
            m->host->wv2[o] = wv2;
            // model.py::_initialize_filter
            const double wvx = sqrt((k * dx) * (k * dx) + (l * dy) * (l * dy));
            m->host->filtr[o] = wvx <= cphi ? 1.0 : exp(-cfg->filterfac * pow(wvx - cphi, 4.));
            // qg_model.py::_initialize_inversion_matrix
            const double det = wv2 * (wv2 + F1 + F2);
            const double det_inv = det != 0. ? 1.0 / det : 0.;
            m->host->a[0 * sz + o] = -(wv2 + F2) * det_inv;
            m->host->a[1 * sz + o] = -F1 * det_inv;
            m->host->a[2 * sz + o] = -F2 * det_inv;
            m->host->a[3 * sz + o] = -(wv2 + F1) * det_inv;
            if (det == 0.) for (int t = 0; t < 4; ++t) m->host->a[t * sz + o] = 0.;
        }
    // twiddles and the digit-reversal map of the DIF passes
    std::vector<double2> tw(N);
    for (int t = 0; t < N; ++t) {
        const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)t / (long double)N;
        tw[t] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    std::vector<int> pos(N);
    for (int P = 0; P < N; ++P) {
        int rem = P, kf = 0, mult = 1, n = N;
        for (int p = 0; p < nrad; ++p) {
            n /= rad[p];
            const int qd = rem / n;
            rem -= qd * n;
            kf += qd * mult;
            mult *= rad[p];
        }
        pos[kf] = P;
    }
    int rc;
    if ((rc = upload(m->t_filtr, m->host->filtr)) || (rc = upload(m->t_wv2, m->host->wv2)) ||
        (rc = upload(m->t_a, m->host->a)) || (rc = upload(m->t_kk, m->host->kk)) ||
        (rc = upload(m->t_ll, m->host->ll)) || (rc = upload(m->t_tw, tw)) || (rc = upload(m->t_pos, pos))) {
        qgx_destroy(m);
        return rc;
    }
    SpecDev &d = m->d;
    d.N = N; d.NK = NK; d.LD = N + 1; d.B = B; d.nrad = nrad;
    for (int p = 0; p < MAX_RADIX_PASSES; ++p) d.rad[p] = p < nrad ? rad[p] : 1;
    d.filtr = m->t_filtr; d.wv2 = m->t_wv2; d.a = m->t_a; d.kk = m->t_kk; d.ll = m->t_ll;
    d.tw = m->t_tw; d.pos = m->t_pos;
    d.U[0] = cfg->U1; d.U[1] = cfg->U2; d.Qy[0] = Qy1; d.Qy[1] = Qy2;
    d.rek = cfg->rek; d.invN2 = 1.0 / ((double)N * (double)N); d.dt = cfg->dt; d.dx = dx;
    d.H[0] = cfg->H1; d.H[1] = cfg->H1 / cfg->delta; d.Htot = d.H[0] + d.H[1];

    const size_t nr = (size_t)B * 2 * N * N, ns = (size_t)B * 2 * N * NK;
    m->plan_only = cfg->plan_only != 0;
    if (m->plan_only) { /* transforms only: no state */ }
    else if ((rc = dalloc(m->q, nr)) || (rc = dalloc(m->u, nr)) || (rc = dalloc(m->v, nr)) ||
        (rc = dalloc(m->S, nr)) || (rc = dalloc(m->qh[0], ns)) || (rc = dalloc(m->qh[1], ns)) ||
        (rc = dalloc(m->ph, ns)) || (rc = dalloc(m->dqh, ns)) || (rc = dalloc(m->dq[0], ns)) ||
        (rc = dalloc(m->dq[1], ns)) || (rc = dalloc(m->dq[2], ns)) || (N == 256 && (rc = dalloc(m->dq[3], ns)))) {
        qgx_destroy(m);
        return rc;
    }
    if (!m->plan_only) {   // latent noise + scratch sized for the wider (double) case
        void *p = nullptr, *p2 = nullptr;
        if (hipMalloc(&p, nr * sizeof(double)) != hipSuccess || hipMalloc(&p2, nr * sizeof(double)) != hipSuccess) {
            set_error("hipMalloc of noise buffers failed");
            qgx_destroy(m);
            return QGX_ERR_HIP;
        }
        (void)hipMemset(p, 0, nr * sizeof(double));
        (void)hipMemset(p2, 0, nr * sizeof(double));
        m->z = p; m->xi = p2;
    }
    m->small = small_path_fits(N);
    if (m->small) rc = small_prepare(d);
    else {
        rc = dalloc(m->zbuf, (size_t)B * 3 * (N + large_zpad()) * N);   // 3 complex work fields per member (spectral_large.hip ZF), padded rows
        if (!rc) rc = large_prepare(d);
    }
    if (rc) { qgx_destroy(m); return rc; }
    *out = m;
    return QGX_OK;
}

extern "C" int qgx_destroy(qgx_model *m) {
    if (!m) return QGX_OK;
    (void)hipSetDevice(m->cfg.device);
    void *ptrs[] = {m->t_filtr, m->t_wv2, m->t_a, m->t_kk, m->t_ll, m->t_tw, m->t_pos, m->q, m->u, m->v,
                    m->S, m->qh[0], m->qh[1], m->ph, m->dqh, m->dq[0], m->dq[1], m->dq[2], m->dq[3], m->dg_z, m->zbuf, m->team_ctl,
                    m->z, m->xi};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (double *p : m->dg_R) if (p) (void)hipFree(p);
    for (double *p : m->dg_S) if (p) (void)hipFree(p);
    for (double *p : m->dg_acc) if (p) (void)hipFree(p);
    if (m->sib_flag) (void)hipFree(m->sib_flag);
    for (hipStream_t sst : m->adv_stream) if (sst) (void)hipStreamDestroy(sst);
    for (auto &evs : m->adv_event) for (hipEvent_t ev : evs) if (ev) (void)hipEventDestroy(ev);
    for (hipStream_t sst : m->sub_stream) if (sst) (void)hipStreamDestroy(sst);
    for (hipEvent_t ev : m->sub_event) if (ev) (void)hipEventDestroy(ev);
    delete m;
    return QGX_OK;
}

extern "C" size_t qgx_field_bytes(const qgx_model *m, int field) {
    if (!m) return 0;
    const size_t nr = (size_t)m->B * 2 * m->N * m->N, ns = (size_t)m->B * 2 * m->N * m->NK;
    switch (field) {
        case QGX_F_Q: case QGX_F_U: case QGX_F_V: case QGX_F_S: case QGX_F_P: return nr * sizeof(double);
        case QGX_F_QH: case QGX_F_PH: case QGX_F_DQHDT: case QGX_F_DQHDT_P: case QGX_F_DQHDT_PP:
            return ns * sizeof(double2);
        case QGX_F_Z: return nr * (m->z_double ? sizeof(double) : sizeof(float));
        default: return 0;
    }
}

extern "C" int qgx_get(qgx_model *m, int field, void *out_dev, void *stream) {
    QGX_NEEDS_STATE(m, "qgx_get");
    { int trc = team_settle(m, (hipStream_t)stream); if (trc) return trc; }
    QGX_REQUIRE(m && out_dev, "qgx_get: null argument");
    if (m->uv_stale && (field == QGX_F_U || field == QGX_F_V || field == QGX_F_PH || field == QGX_F_P)) {
        // the steps since the last refresh stored no ph, u, v (refresh_diag == 0, every step of a run kernel): hand out
        // the fields of the CURRENT state rather than those of an older one (as qgx_status_ke_cfl does)
        int irc = qgx_invert(m, stream);
        if (irc) return irc;
    }
    if (field == QGX_F_P)      // pyqg's derived field p = ifft(ph) (model.py::_calc_derived_fields), straight into the caller's buffer
        return m->small ? small_qh_to_q(m->d, m->opts, m->ph, (double *)out_dev, (hipStream_t)stream)
                        : large_qh_to_q(m, m->ph, (double *)out_dev, (hipStream_t)stream);
    if (field == QGX_F_Q && !m->small) { int rc = large_ensure_q(m, (hipStream_t)stream); if (rc) return rc; }
    const void *src = nullptr;
    switch (field) {
        case QGX_F_Q: src = m->q; break;
        case QGX_F_U: src = m->u; break;
        case QGX_F_V: src = m->v; break;
        case QGX_F_S: src = m->S; break;
        case QGX_F_QH: src = m->qh[m->cur_q]; break;
        case QGX_F_PH: src = m->ph; break;
        case QGX_F_DQHDT: src = m->dq[m->i_new]; break;
        // kernel.pyx rotates by copy: after a step dqhdt_p == dqhdt and dqhdt_pp is the previous one
        case QGX_F_DQHDT_P: src = m->dq[m->i_new]; break;
        case QGX_F_DQHDT_PP: src = m->dq[m->i_p]; break;
        case QGX_F_Z: src = m->z; break;
        default: QGX_REQUIRE(false, "qgx_get: unknown field %d", field);
    }
    QGX_HIP(hipMemcpyAsync(out_dev, src, qgx_field_bytes(m, field), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return QGX_OK;
}

extern "C" int qgx_get_table(qgx_model *m, int table, double *out) {
    QGX_REQUIRE(m && out, "qgx_get_table: null argument");
    const std::vector<double> *v = nullptr;
    switch (table) {
        case QGX_T_FILTR: v = &m->host->filtr; break;
        case QGX_T_WV2: v = &m->host->wv2; break;
        case QGX_T_A: v = &m->host->a; break;
        case QGX_T_KK: v = &m->host->kk; break;
        case QGX_T_LL: v = &m->host->ll; break;
        default: QGX_REQUIRE(false, "qgx_get_table: unknown table %d", table);
    }
    memcpy(out, v->data(), v->size() * sizeof(double));
    return QGX_OK;
}

extern "C" int qgx_set_q(qgx_model *m, const double *q_dev, void *stream) {
    QGX_NEEDS_STATE(m, "qgx_set_q");
    QGX_REQUIRE(m && q_dev, "qgx_set_q: null argument");
    { int trc = team_settle(m, (hipStream_t)stream); if (trc) return trc; }
    hipStream_t st = (hipStream_t)stream;
    QGX_HIP(hipMemcpyAsync(m->q, q_dev, qgx_field_bytes(m, QGX_F_Q), hipMemcpyDeviceToDevice, st));
    m->q_stale = false;
    return m->small ? small_q_to_qh(m->d, m->opts, m->q, m->qh[m->cur_q], st) : large_q_to_qh(m, m->q, m->qh[m->cur_q], st);
}

extern "C" int qgx_set_qh(qgx_model *m, const double *qh_dev, void *stream) {
    QGX_NEEDS_STATE(m, "qgx_set_qh");
    QGX_REQUIRE(m && qh_dev, "qgx_set_qh: null argument");
    { int trc = team_settle(m, (hipStream_t)stream); if (trc) return trc; }
    hipStream_t st = (hipStream_t)stream;
    QGX_HIP(hipMemcpyAsync(m->qh[m->cur_q], qh_dev, qgx_field_bytes(m, QGX_F_QH), hipMemcpyDeviceToDevice, st));
    m->q_stale = false;
    return m->small ? small_qh_to_q(m->d, m->opts, m->qh[m->cur_q], m->q, st) : large_qh_to_q(m, m->qh[m->cur_q], m->q, st);
}

extern "C" int qgx_invert(qgx_model *m, void *stream) {
    QGX_NEEDS_STATE(m, "qgx_invert");
    { int trc = team_settle(m, (hipStream_t)stream); if (trc) return trc; }
    QGX_REQUIRE(m, "qgx_invert: null model");
    hipStream_t st = (hipStream_t)stream;
    m->uv_stale = false;
    return m->small ? small_invert(m->d, m->opts, m->qh[m->cur_q], m->ph, m->u, m->v, st) : large_invert(m, st);
}

// kernel-path switches (ModelOpts, common.hpp): cross-checks and A/B timing; every combination computes the same step
extern "C" int qgx_set_option(qgx_model *m, const char *name, int value) {
    QGX_REQUIRE(m && name, "qgx_set_option: null argument");
    ModelOpts &o = m->opts;
    if (!strcmp(name, "genfuse")) o.genfuse = value ? 1 : 0;
    else if (!strcmp(name, "diag_fused")) o.diag_fused = value ? 1 : 0;
    else if (!strcmp(name, "diag_wide")) { QGX_REQUIRE(value >= -1 && value <= 1, "diag_wide must be -1 (auto), 0 or 1"); o.diag_wide = value; }
    else if (!strcmp(name, "diag_reg")) { QGX_REQUIRE(value >= 0 && value <= 3, "diag_reg must be 0, 1 (auto), 2 (one workgroup per member) or 3 (two)"); o.diag_reg = value; }
    else if (!strcmp(name, "lsplit")) { QGX_REQUIRE(value >= -1 && value <= 1, "lsplit must be -1 (auto), 0 or 1"); o.lsplit = value; }
    else if (!strcmp(name, "siblings")) { QGX_REQUIRE(value >= -1 && value <= 2, "siblings must be -1 (auto), 0, 1 or 2"); o.siblings = value; }
    else if (!strcmp(name, "split_adv")) { QGX_REQUIRE(value == 0 || value == 1, "split_adv must be 0 or 1"); o.split_adv = value; }
    else if (!strcmp(name, "streams")) { QGX_REQUIRE(value >= 0 && value <= 2, "streams must be 0 (auto), 1 or 2"); o.streams = value; }
    else if (!strcmp(name, "spec_threads")) {
        QGX_REQUIRE(value == 0 || value == 256 || value == 512 || value == 1024, "spec_threads must be 0 (auto), 256, 512 or 1024");
        o.spec_threads = value;
    }
    else if (!strcmp(name, "team")) o.team = value ? 1 : 0;
    else if (!strcmp(name, "team_min")) { QGX_REQUIRE(value >= 1, "team_min must be >= 1"); o.team_min = value; }
    else if (!strcmp(name, "team_fault")) {
#ifdef QGX_AB
        o.team_fault = value ? 1 : 0;
#else
        QGX_REQUIRE(false, "option 'team_fault' is a test hook of the A/B library only (make ab, QGX_LIB=libqgx_ab.so)");
#endif
    }
    else if (!strcmp(name, "step_fault")) {
#ifdef QGX_AB
        QGX_REQUIRE(value >= 0 && value <= 2, "step_fault must be 0, 1 or 2");
        o.step_fault = value;
#else
        QGX_REQUIRE(false, "option 'step_fault' is a test hook of the A/B library only (make ab, QGX_LIB=libqgx_ab.so)");
#endif
    }
    else if (!strcmp(name, "large_fused")) o.large_fused = value ? 1 : 0;
    else if (!strcmp(name, "large_lazy_q")) o.large_lazy_q = value ? 1 : 0;
    else if (!strcmp(name, "large_specialised")) o.large_specialised = value ? 1 : 0;
    else QGX_REQUIRE(false, "unknown model option '%s'", name);
    return QGX_OK;
}

extern "C" int64_t qgx_step_count(const qgx_model *m) { return m ? m->tc : -1; }
extern "C" int qgx_run_kernel_state(const qgx_model *m) { return m ? m->team_state : 0; }

extern "C" int qgx_reset_time(qgx_model *m) {
    QGX_NEEDS_STATE(m, "qgx_reset_time");
    QGX_REQUIRE(m, "qgx_reset_time: null model");
    { int trc = team_settle(m, nullptr); if (trc) return trc; }
    m->tc = 0; m->ablevel = 0; m->have_noise = false; m->const_counter = 0; m->have_forcing = false;
    const size_t ns = (size_t)m->B * 2 * m->N * m->NK;
    for (int i = 0; i < 4; ++i) if (m->dq[i]) QGX_HIP(hipMemset(m->dq[i], 0, ns * sizeof(double2)));
    return QGX_OK;
}

extern "C" int qgx_status_ke_cfl(qgx_model *m, double *out_dev, void *stream) {
    QGX_NEEDS_STATE(m, "qgx_status_ke_cfl");
    { int trc = team_settle(m, (hipStream_t)stream); if (trc) return trc; }
    QGX_REQUIRE(m && out_dev, "qgx_status_ke_cfl: null argument");
    if (m->uv_stale) {
        // the steps since the last refresh stored no ph, u, v: a status of stale (or never written) fields would be
        // meaningless, so the CURRENT state is inverted first (pyqg reports the fields of its last _invert, one step back)
        int irc = qgx_invert(m, stream);
        if (irc) return irc;
    }
    hipLaunchKernelGGL(k_status, dim3(m->B), dim3(256), 0, (hipStream_t)stream, m->d, m->ph, m->u, m->v, out_dev);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

// the two-kernel step (k_step_small PART 1 / 2), small grids in layer-split form: option "split_adv" = 1 only
// (measured, bench_tools/split_adv.py: DESIGN.md section 3.1c)
static bool step_adv_applies(const qgx_model *m) {
    if (m->plan_only || !m->small || m->opts.split_adv == 0 || !small_layer_split(m->d, m->opts) || !m->adv_stream[m->adv_slot]) return false;
    return m->opts.split_adv == 1;
}
static int step_sib_ensure(qgx_model *m) {
    if (!m->small || m->opts.siblings == 0 || m->sib_flag || m->plan_only) return QGX_OK;
    // [0, 2 B): the forcing workgroups' flags; [2 B, 4 B): where the chain workgroups run
    QGX_HIP(hipMalloc((void **)&m->sib_flag, (size_t)4 * m->B * sizeof(unsigned long long)));
    QGX_HIP(hipMemset(m->sib_flag, 0, (size_t)4 * m->B * sizeof(unsigned long long)));
    return QGX_OK;
}
static int step_adv_ensure(qgx_model *m) {
    if (!m->small || m->opts.split_adv == 0) return QGX_OK;
    for (int i = 0; i < 2; ++i) {
        if (!m->adv_stream[i]) QGX_HIP(hipStreamCreateWithFlags(&m->adv_stream[i], hipStreamNonBlocking));
        for (int j = 0; j < 2; ++j)
            if (!m->adv_event[i][j]) QGX_HIP(hipEventCreateWithFlags(&m->adv_event[i][j], hipEventDisableTiming));
    }
    return QGX_OK;
}

// ---- the stepping loop: pyqg model.py::_step_forward with the plugin call of
// pyqg_generative/models/parameterization.py:23-34 and samplers of stochastic_pyqg.py:30-72
// the steps of one call on one stream, for the whole ensemble or for one half of it (qgx_step below)
static int step_core(qgx_model *m, int nsteps, const qgx_param *p, int refresh_diag, hipStream_t st) {
    const int N = m->N, B = m->B;
    if (p && p->gen) {
        QGX_REQUIRE(p->sampling == QGX_SAMPLING_AR1 || p->sampling == QGX_SAMPLING_CONSTANT,
                    "qgx_step: unknown sampling %d", p->sampling);
        QGX_REQUIRE(!(p->sampling == QGX_SAMPLING_CONSTANT && p->nsteps < 1),
                    "qgx_step: constant sampler needs nsteps >= 1");
        QGX_REQUIRE(p->nsteps != 0, "qgx_step: nsteps == 0 is not a valid decorrelation time");
        QGX_REQUIRE(!(p->z_external_dev && nsteps != 1), "qgx_step: external noise needs nsteps_to_run == 1");
        m->z_double = generator_noise_is_double(p->gen);
    }
    const bool plain = !(p && (p->gen || p->forcing_dev));
    const bool fuse_ok = m->opts.genfuse != 0;
    m->x_ready_gen = nullptr;                                            // an assembled input never outlives its call
    for (int s = 0; s < nsteps; ++s) {
        { int trc = team_settle(m, st); if (trc) return trc; }              // at most one run is ever unsettled
        if (plain && !m->small && large_team_available(m, st)) {
            // a run of steps with no diagnostics increment due inside it and no (u, v, psi) refresh asked of it
            const bool due = m->dg_every > 0 && m->tc >= 1 && m->tc >= m->dg_start && m->tc % m->dg_every == 0;
            int K = 0;
            const int last = refresh_diag ? nsteps - 1 : nsteps;          // the refreshing step takes the three-launch path
            while (s + K < last) {
                const int64_t t = m->tc + K;
                if (K > 0 && m->dg_every > 0 && t >= 1 && t >= m->dg_start && t % m->dg_every == 0) break;
                ++K;
            }
            const int kmin = m->opts.team_min;
            if (K >= kmin) {
                if (due) { int drc = diag_increment(m, nullptr, 1.0, st); if (drc) return drc; }
                int rc = model_step_run(m, K, st);
                if (rc) return rc;
                s += K - 1;
                continue;
            }
        }
        bool has_S = false;
        const double *S = nullptr;
        double weight = 1.0;
        int demean_in_kernel = 0;
        GenFuse gf;
        bool use_gf = false;
        const int diag_now = (refresh_diag && s == nsteps - 1) ? 1 : 0;
        StepArgs pre;
        bool have_pre = false;
        int ablevel_before = m->ablevel;
        if (p && p->gen) {
            weight = p->weight;
            bool compute = true;
            double a = 0.0, b = 1.0;
            bool draw = true;
            if (p->sampling == QGX_SAMPLING_AR1) {
                if (m->have_noise) {
                    if (p->nsteps > 0) {
                        a = 1.0 - 1.0 / p->nsteps;
                        b = sqrt(1.0 / p->nsteps * (2.0 - 1.0 / p->nsteps));
                    } else { a = 1.0; b = 0.0; }
                }
            } else {
                if (m->have_noise) {
                    if (m->const_counter % p->nsteps == 0) m->const_counter = 1;
                    else { m->const_counter += 1; compute = false; draw = false; }
                } else m->const_counter = 1;
            }
            NoiseUpdate nu;
            if (draw) {
                nu.xi_ext = p->z_external_dev; nu.seed = p->seed; nu.member_offset = p->member_offset;
                nu.step = m->noise_step; nu.a = a; nu.b = b;
                m->noise_step += 1;
                m->have_noise = true;
            }
            if (compute) {
                if (!m->small) { int qrc = large_ensure_q(m, st); if (qrc) return qrc; }
                // The half of the step kernel that needs nothing of the forcing (inversion, advection products, their
                // transform, the tendency without its forcing term) on the side stream, under the generator's layers: on an
                // ensemble that leaves CUs idle the step is a chain of dependent launches, and this takes two of the step
                // kernel's four transforms out of it.  Not on steps that refresh (psi, u, v) or increment the diagnostics.
                if (step_adv_applies(m) && !diag_now &&
                    !(m->dg_every > 0 && m->tc >= 1 && m->tc >= m->dg_start && m->tc % m->dg_every == 0)) {
                    const int sl = m->adv_slot;
                    ablevel_before = m->ablevel;
                    step_args_begin(m, 0, pre);
                    hipError_t e = hipEventRecord(m->adv_event[sl][0], st);
                    if (e == hipSuccess) e = hipStreamWaitEvent(m->adv_stream[sl], m->adv_event[sl][0], 0);
                    if (e != hipSuccess) { m->ablevel = ablevel_before; QGX_HIP(e); }
                    int arc = small_step(m->d, m->opts, pre, m->adv_stream[sl], 1);
                    // (whatever was enqueued on the side stream is joined below on every path)
                    e = hipEventRecord(m->adv_event[sl][1], m->adv_stream[sl]);
                    if (arc || e != hipSuccess) {
                        (void)hipStreamSynchronize(m->adv_stream[sl]);
                        m->ablevel = ablevel_before;
                        if (arc) return arc;
                        QGX_HIP(e);
                    }
                    have_pre = true;
                }
                // Small grids in layer-split form, GAN / VAE: the generator's output kernel rides in the step kernel's
                // prologue (unless this step's diagnostics need S first) and — white-in-time Philox noise, more steps to
                // come in this call — the next step's input kernel in its epilogue (GenFuse, common.hpp)
                const bool fusable = fuse_ok && m->small && small_layer_split(m->d, m->opts) && !m->z_double;
                const bool diag_due = m->dg_every > 0 && m->tc >= 1 && m->tc >= m->dg_start && m->tc % m->dg_every == 0;
                const bool input_ready = fusable && draw && m->x_ready_gen == (const void *)p->gen && m->x_ready_step == nu.step;
                m->x_ready_gen = nullptr;
                // a redraw always comes with a recompute; the sampler update rides in the input kernel
                int rc = generator_forward(p->gen, m->q, m->z, m->S, B, N, p->demean, st, draw ? &nu : nullptr,
                                           fusable && !diag_due ? &gf : nullptr, input_ready);
                if (rc) {
                    if (have_pre) {        // the first half wrote a dead tendency slot only: the state is that of step n - 1
                        (void)hipStreamWaitEvent(st, m->adv_event[m->adv_slot][1], 0);
                        m->ablevel = ablevel_before;
                    }
                    return rc;
                }
                m->have_forcing = true;
                const bool white = (p->sampling == QGX_SAMPLING_AR1 && p->nsteps == 1) ||
                                   (p->sampling == QGX_SAMPLING_CONSTANT && p->nsteps == 1);
                if (fusable && white && !p->z_external_dev && s + 1 < nsteps) {
                    if ((rc = generator_input_info(p->gen, B, N, &gf))) return rc;
                    gf.z = (float *)m->z; gf.b = 1.f;
                    gf.seed = p->seed; gf.member_offset = p->member_offset; gf.step = m->noise_step;
                    m->x_ready_gen = p->gen; m->x_ready_step = m->noise_step;
                }
                use_gf = gf.y != nullptr || gf.X != nullptr;
            }
            has_S = m->have_forcing;
            S = m->S;
        } else if (p && p->forcing_dev) {
            has_S = true;
            S = p->forcing_dev;
            weight = p->weight;
            demean_in_kernel = p->demean;
        }
        // model.py::_calc_diagnostics: t >= dt, t >= tavestart, tc % taveints == 0 (before the time step)
        if (m->dg_every > 0 && m->tc >= 1 && m->tc >= m->dg_start && m->tc % m->dg_every == 0) {
            int drc = diag_increment(m, has_S ? S : nullptr, weight, st);
            if (drc) return drc;
        }
        int rc = model_step_once(m, has_S, S, weight, demean_in_kernel, diag_now, st, use_gf ? &gf : nullptr, have_pre ? &pre : nullptr);
        if (rc) {
            if (have_pre) { (void)hipStreamSynchronize(m->adv_stream[m->adv_slot]); m->ablevel = ablevel_before; }
            return rc;
        }
    }
    return QGX_OK;
}

// Two half-ensembles on two streams.  The online step is a chain of 7 dependent launches whose ramps, tails, kernel boundaries
// and the step kernel's latency chain (one workgroup per member and layer) leave CUs idle; members are independent, so the two
// halves of a shard can advance on two internal streams and fill each other's gaps.  A half is the SAME model over a slice:
// every per-member array is member-major, so a child is a copy of the bookkeeping with B / 2 members and offset pointers; the
// generator evaluates the halves in two workspaces.  Per-member results do not depend on it (Philox streams are keyed by the
// global member id; the halves run the kernels the whole would whenever both sides of the few ensemble-size thresholds agree:
// bit-identical at 64 x 64 / 128 and 96 x 96 / 32 members).
// Measured with the bench's cadence (bench_tools/halves_cadence.py, one stream -> two): 96 x 96 with 16 / 24 / 32 / 48 / 64
// members +6.5 / +4 / +4 / +11 / +1 % (before the 96 x 96 tile shapes were chosen by quantisation: +20 / . / +11.5 / +20 / +9 %); 64 x 64 with 16 / 32 / 48 / 64 / 128 members -1 / 0 / +19 / 0 / +0.4 %; 48 x 48 with 32 /
// 64 members -25 / +27 %; 32 x 32 with 64 members -5 %: on the smaller grids the sign follows the tile-count quantisation of the
// halves against the whole, not a rule, so the automatic choice is the 96 x 96 grid (16 ... 64 members) only and option
// "streams" = 2 asks for it elsewhere.  Round 4 (bench_tools/halves_sizes.py): at 64 x 64 / 128 members the two-halves step
// takes 635 ... 639 us on every box met, the one-stream step 637 us on a fast box and 662 ... 665 us on slow ones (the boxes
// differ in the clock the f16 MFMA kernels sustain; interleaving the halves' memory-bound and MFMA-bound kernels evens that
// out): +4 % where the box is slow, nothing where it is fast.  Not the default there: layer 2 of a half runs beside other
// kernels, so the bench's live roofline of that kernel would describe the mix, not the kernel.
static bool step_in_halves(const qgx_model *m, const qgx_param *p) {
    if (m->opts.streams == 1 || !m->small || (m->B & 1) || !p || !p->gen || p->z_external_dev) return false;
    if (m->opts.streams == 2) return true;
    return m->N == 96 && m->B >= 16 && m->B <= 64;
}

extern "C" int qgx_step_streams(const qgx_model *m, const qgx_param *p) {
    return m && !m->plan_only && step_in_halves(m, p) ? 2 : 1;
}

extern "C" int qgx_step(qgx_model *m, int nsteps, const qgx_param *p, int refresh_diag, void *stream) {
    QGX_NEEDS_STATE(m, "qgx_step");
    QGX_REQUIRE(m && nsteps >= 0, "qgx_step: bad argument");
    hipStream_t st = (hipStream_t)stream;
    { int trc = team_settle(m, st); if (trc) return trc; }
    if (p && p->gen && nsteps > 0) { int arc = step_adv_ensure(m); if (arc) return arc; }
    if (p && (p->gen || p->forcing_dev) && nsteps > 0) { int src = step_sib_ensure(m); if (src) return src; }
    if (nsteps == 0 || !step_in_halves(m, p)) {
        if (p && p->gen) { int wrc = generator_select_workspace(p->gen, 0); if (wrc) return wrc; }
        return step_core(m, nsteps, p, refresh_diag, st);
    }
    const int N = m->N, NK = m->NK, Bp = m->B / 2;
    for (int i = 0; i < 2; ++i) if (!m->sub_stream[i]) QGX_HIP(hipStreamCreateWithFlags(&m->sub_stream[i], hipStreamNonBlocking));
    for (int i = 0; i < 3; ++i) if (!m->sub_event[i]) QGX_HIP(hipEventCreateWithFlags(&m->sub_event[i], hipEventDisableTiming));
    if (m->dg_every > 0) { int arc = diag_ensure_alloc(m); if (arc) return arc; }   // BEFORE the children copy the pointers
    // fork.  From here on every exit runs the join below: the caller's stream is ordered behind whatever the sub-streams
    // were given, and a step that failed half-way leaves the handle marked invalid (the halves may stand at different
    // steps, the parent's bookkeeping at neither): later calls fail loudly instead of stepping from an inconsistent state.
    QGX_HIP(hipEventRecord(m->sub_event[2], st));
    qgx_model child[2] = {*m, *m};
    qgx_param pp[2] = {*p, *p};
    const size_t sr = (size_t)2 * N * N, ss = (size_t)2 * N * NK, s2 = (size_t)N * NK;
    const size_t zbytes = generator_noise_is_double(p->gen) ? sizeof(double) : sizeof(float);
    int rc = QGX_OK;
    hipError_t herr = hipSuccess;
    const char *hwhat = "";
    for (int c = 0; c < 2; ++c) {
        if (herr == hipSuccess && (herr = hipStreamWaitEvent(m->sub_stream[c], m->sub_event[2], 0)) != hipSuccess) hwhat = "hipStreamWaitEvent (fork)";
        qgx_model &k = child[c];
        const size_t b0 = (size_t)c * Bp;
        k.B = Bp; k.d.B = Bp;
        k.q += b0 * sr; k.u += b0 * sr; k.v += b0 * sr; k.S += b0 * sr;
        k.qh[0] += b0 * ss; k.qh[1] += b0 * ss; k.ph += b0 * ss; k.dqh += b0 * ss;
        for (int i = 0; i < 3; ++i) k.dq[i] += b0 * ss;
        // latent noise and its scratch: dense (B, 2, N, N) of the element type in use (qgx_get(QGX_F_Z) reads it that way)
        k.z = (char *)k.z + b0 * sr * zbytes;
        k.xi = (char *)k.xi + b0 * sr * zbytes;
        if (k.dg_R[0]) {
            for (double *&r : k.dg_R) r += b0 * sr;
            for (double *&r : k.dg_S) r += b0 * ss * 2;
            for (int i = 0; i < qgx::N_DIAGS; ++i) k.dg_acc[i] += b0 * (i < 2 ? 2 * s2 : s2);
        }
        k.sub_stream[0] = k.sub_stream[1] = nullptr;
        k.adv_slot = c;
        k.is_half = true;
        if (k.sib_flag) k.sib_flag += b0 * 4;          // (a half owns the flag words and the placement words of its members: 4 per member)
        pp[c].member_offset = p->member_offset + b0;
    }
    // the halves take turns in chunks of steps (a chunk keeps the fused input / output kernels of consecutive steps fused and
    // both streams fed: the host enqueues a chunk in a fraction of the time the GPU needs for it)
    constexpr int CHUNK = 8;
    bool advanced = false;
    for (int s0 = 0; s0 < nsteps && !rc && herr == hipSuccess; s0 += CHUNK) {
        const int n = nsteps - s0 < CHUNK ? nsteps - s0 : CHUNK;
        const int refresh = refresh_diag && s0 + n == nsteps;
        for (int c = 0; c < 2 && !rc; ++c) {
            if ((rc = generator_select_workspace(p->gen, c))) break;
#ifdef QGX_AB
            if (m->opts.step_fault == 1 + c && s0 >= CHUNK) {        // test hook (A/B library only): this half refuses its second chunk
                qgx::set_error("qgx_step: injected failure of half %d (option step_fault)", c);
                rc = QGX_ERR_HIP;
                break;
            }
#endif
            rc = step_core(&child[c], n, &pp[c], refresh, m->sub_stream[c]);
            advanced = true;
        }
    }
    (void)generator_select_workspace(p->gen, 0);
    // join (best effort on every path: an error here is reported, but never skips the other stream)
    for (int c = 0; c < 2; ++c) {
        hipError_t e1 = hipEventRecord(m->sub_event[c], m->sub_stream[c]);
        if (e1 == hipSuccess) e1 = hipStreamWaitEvent(st, m->sub_event[c], 0);
        if (e1 != hipSuccess && herr == hipSuccess) { herr = e1; hwhat = "join of the half-ensemble streams"; }
    }
    if (herr != hipSuccess) {
        // the caller's stream could not be ordered behind the halves: wait for them here, so that nothing the caller
        // enqueues next can overtake them
        (void)hipStreamSynchronize(m->sub_stream[0]);
        (void)hipStreamSynchronize(m->sub_stream[1]);
        qgx::set_error("qgx_step: %s failed: %s", hwhat, hipGetErrorString(herr));
        if (!rc) rc = QGX_ERR_HIP;
    }
    if (rc) {
        if (advanced) {
            const std::string why = qgx_last_error();
            m->broken = "a step of two half-ensembles failed half-way (" + why + "); the halves are at steps " +
                        std::to_string((long long)child[0].tc) + " and " + std::to_string((long long)child[1].tc) +
                        "; destroy the handle";
        }
        return rc;
    }
    // the halves advanced in lockstep: the bookkeeping of either is the ensemble's
    const qgx_model &k = child[0];
    QGX_REQUIRE(k.tc == child[1].tc && k.cur_q == child[1].cur_q && k.i_new == child[1].i_new && k.noise_step == child[1].noise_step &&
                k.dg_count == child[1].dg_count, "qgx_step: the two halves of the ensemble left the lockstep");
    m->tc = k.tc; m->ablevel = k.ablevel; m->cur_q = k.cur_q; m->i_new = k.i_new; m->i_p = k.i_p; m->i_pp = k.i_pp; m->i_x = k.i_x;
    m->z_double = k.z_double; m->have_noise = k.have_noise; m->const_counter = k.const_counter; m->have_forcing = k.have_forcing;
    m->noise_step = k.noise_step; m->uv_stale = k.uv_stale; m->q_stale = k.q_stale; m->dg_count = k.dg_count;
    m->sib_epoch = k.sib_epoch > child[1].sib_epoch ? k.sib_epoch : child[1].sib_epoch;
    m->x_ready_gen = nullptr;
    return QGX_OK;
}
