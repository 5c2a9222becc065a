/*
 * qgx.h — C ABI of the MI355X (gfx950) online parameterized-QG ensemble engine.
 *
 * Drop-in boundary for ONE hot path of m2lines/pyqg_generative: the per-step work
 * that `pyqg_generative/tools/simulate.py:109-145` (run_simulation) drives through
 * `pyqg.QGModel.run_with_snapshots` and the `Parameterization.__call__` plugin
 * (`pyqg_generative/models/parameterization.py:23-34`).  Each entry point cites the
 * reference interface it replaces (paths relative to the reference repository;
 * "pyqg" = pyqg 0.7.2, the un-vendored spectral core the reference calls into).
 *
 * Conventions
 *   - every function returns 0 on success or a negative qgx_status; nothing throws
 *     across the ABI; qgx_last_error() gives a thread-local message.
 *   - all `*_dev` pointers are DEVICE pointers (e.g. torch tensor.data_ptr()),
 *     caller-owned and kept alive by the caller until `stream` is synchronised.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     No entry point on the step path synchronises the device.
 *   - ensemble-batched layouts, C order:
 *       real fields      (B, 2, N, N)      double   [member][layer][y][x]
 *       spectral fields  (B, 2, N, N/2+1)  complex double, interleaved (re,im),
 *                        l index ordered [0..N/2-1, -N/2..-1], k index [0..N/2]
 *       latent noise z   (B, 2, N, N)      float (GAN/VAE) or double (GZ)
 *   - one host thread per handle; handles are not thread-safe.
 */
#ifndef QGX_H
#define QGX_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum qgx_status {
    QGX_OK = 0,
    QGX_ERR_INVALID = -1,     /* bad argument / unsupported size            */
    QGX_ERR_HIP = -2,         /* a HIP runtime call failed                   */
    QGX_ERR_STATE = -3,       /* call not valid in the handle's state        */
    QGX_ERR_NOMEM = -4
} qgx_status;

/* ---- model ------------------------------------------------------------------
 * Replaces pyqg.QGModel(**pyqg_params) as constructed at simulate.py:83,121 and
 * stochastic_pyqg.py:78-79.  Field names and defaults are pyqg's. */
typedef struct qgx_config {
    int32_t nx;          /* N = nx = ny; even, N = 2^a 3^b, 8 <= N <= 512             */
    int32_t n_members;   /* B: ensemble members resident on this device               */
    int32_t device;      /* HIP device ordinal                                        */
    int32_t plan_only;   /* != 0: the handle is an FFT plan of its grid for qgx_rfft2 / qgx_irfft2 only: tables and
                            work space, NO model state (every state entry point returns QGX_ERR_STATE)              */
    double  L;           /* domain size [m]                     (pyqg default 1e6)     */
    double  dt;          /* time step [s]                       (7200)                */
    double  rek;         /* bottom drag [1/s]                   (5.787e-7)            */
    double  delta;       /* H1/H2                               (0.25)                */
    double  beta;        /* [1/(m s)]                           (1.5e-11)             */
    double  rd;          /* deformation radius [m]              (15000)               */
    double  U1, U2;      /* background zonal flow [m/s]         (0.025, 0)            */
    double  H1;          /* upper layer thickness [m]           (500)                 */
    double  filterfac;   /* exponential filter strength         (23.6)                */
} qgx_config;

typedef struct qgx_model qgx_model;          /* opaque */
typedef struct qgx_generator qgx_generator;  /* opaque */

enum qgx_field {               /* pyqg attribute of the same name */
    QGX_F_Q = 0,      /* m.q      real      */
    QGX_F_QH = 1,     /* m.qh     spectral  */
    QGX_F_PH = 2,     /* m.ph     spectral (from the last inversion)            */
    QGX_F_U = 3,      /* m.u      real     (from the last inversion)            */
    QGX_F_V = 4,      /* m.v      real                                          */
    QGX_F_DQHDT = 5,  /* m.dqhdt     spectral (tendency of the last step)       */
    QGX_F_DQHDT_P = 6,
    QGX_F_DQHDT_PP = 7,
    QGX_F_S = 8,      /* real: subgrid forcing used by the last step (m.PV_forcing) */
    QGX_F_Z = 9,      /* latent noise held by the sampler (float or double)     */
    QGX_F_P = 10      /* m.p      real: irfft2(ph), pyqg's derived streamfunction (to_dataset 'p') */
};

enum qgx_table {               /* grid constants, (N, N/2+1) double unless noted */
    QGX_T_FILTR = 0,  /* m.filtr */
    QGX_T_WV2 = 1,    /* m.wv2   */
    QGX_T_A = 2,      /* m.a  (2,2,N,N/2+1) */
    QGX_T_KK = 3,     /* m.kk (N/2+1)       */
    QGX_T_LL = 4      /* m.ll (N)           */
};

int qgx_create(const qgx_config *cfg, qgx_model **out);
int qgx_destroy(qgx_model *m);

/* m.q = q  (kernel.pyx property q: also refreshes qh = rfft2(q)); call sites
 * simulate.py:131, operators.py:232; set_q1q2 at simulate.py:167. */
int qgx_set_q(qgx_model *m, const double *q_dev, void *stream);
/* m.qh = qh (also refreshes q = irfft2(qh)). */
int qgx_set_qh(qgx_model *m, const double *qh_dev, void *stream);
/* copy a state field into a caller buffer of the layout given above */
int qgx_get(qgx_model *m, int field, void *out_dev, void *stream);
/* copy a grid constant (host pointer, doubles) */
int qgx_get_table(qgx_model *m, int table, double *out_host);
/* bytes of a field for this model (so callers can size buffers) */
size_t qgx_field_bytes(const qgx_model *m, int field);

/* m._invert(): ph, u, v from qh (simulate.py:132,168; operators.py:233). */
int qgx_invert(qgx_model *m, void *stream);

/* ---- stepping ----------------------------------------------------------------
 * Replaces pyqg Model._step_forward driven by run_with_snapshots
 * (simulate.py:137) including the plugin call of parameterization.py:23-34 and
 * the samplers of stochastic_pyqg.py:30-72. */
enum qgx_sampling { QGX_SAMPLING_AR1 = 0, QGX_SAMPLING_CONSTANT = 1 };

typedef struct qgx_param {
    qgx_generator *gen;      /* NULL: use `forcing_dev` as S (or no forcing if that is NULL too) */
    int32_t  sampling;       /* qgx_sampling                                             */
    int32_t  nsteps;         /* decorrelation steps (AR1: <0 freezes the noise)          */
    double   weight;         /* `model_weight * parameterization` (simulate.py:242)      */
    uint64_t seed;           /* Philox key for on-device latent noise                    */
    uint64_t member_offset;  /* global id of member 0 on this device (multi-GPU shards)  */
    const void *z_external_dev; /* if non-NULL: white noise xi for THIS step, layout of z
                                   (parity tests); only honoured for nsteps_to_run == 1  */
    const double *forcing_dev;  /* gen == NULL: externally supplied S (B,2,N,N), used as is
                                   (plain pyqg q_parameterization semantics)             */
    int32_t  demean;         /* subtract the per-layer spatial mean of S (parameterization.py:25) */
    int32_t  reserved;
} qgx_param;

/* advance `nsteps_to_run` steps; `p` may be NULL (unparameterized, simulate.py:121).
 * If `refresh_diag` != 0 the last step also stores ph,u,v (as pyqg keeps them).
 * 256 x 256 grids, p == NULL: the steps of a call that neither refresh ph,u,v nor have a diagnostics increment due
 * are executed by ONE persistent launch that occupies every CU of the device (state in registers, row/column exchange
 * in the XCDs' L2).  Such a run is a transaction: it writes only buffers that hold nothing live, and a flag raised inside
 * it (a bounded wait timed out because other work held CUs) is found by the NEXT call that touches the model, which restores
 * the bookkeeping of before the run, switches the kernel off for this model (qgx_run_kernel_state = -1) and replays the
 * steps with three launches each — the run degrades, the state is never undefined.  Issue such calls on one stream at a
 * time (a settle from another stream waits for the replay).
 * Two half-ensembles on two streams (qgx_step_streams): every exit joins both internal streams into `stream`; a call that
 * failed after one half had advanced marks the handle invalid — every later call returns QGX_ERR_STATE with the reason.
 * Option "split_adv" = 1 (default 0; small grids in the two-workgroups-per-member form, generator attached): the half of the
 * step kernel that needs nothing of the forcing runs as a kernel of its own on an internal side stream, forked from and
 * joined into `stream` inside every step; bit-identical, and measured slower on this stack. */
int qgx_step(qgx_model *m, int nsteps_to_run, const qgx_param *p, int refresh_diag, void *stream);
/* Small grids with a generator attached: an even ensemble may advance as two halves on two internal streams that fork from
 * and join `stream` inside the call (members are independent — the reference runs them as separate processes,
 * scripts/run_parameterized.py:55-63 — and the halves fill the idle phases of each other's launch chain); option "streams" of
 * qgx_set_option: 0 automatic (96 x 96 with 16 ... 64 members, where it measured +9 ... +20 %), 1 never, 2 whenever even.  Returns the number of streams (1 or 2)
 * qgx_step would use for this parameterization. */
int qgx_step_streams(const qgx_model *m, const qgx_param *p);
/* step counter / ablevel (m.tc) and reset of the AB history */
int64_t qgx_step_count(const qgx_model *m);
/* 256 x 256 grids: 1 after the census found the device fit for the single-launch runs of qgx_step, -1 if it did not
 * (or a run raised a flag), 0 before the first unparameterized step or on other grids */
int qgx_run_kernel_state(const qgx_model *m);
int qgx_reset_time(qgx_model *m);
/* Kernel-path switches of a model (no reference counterpart; cross-checks in tests/, A/B timing in bench_tools/):
 * every setting computes the same step, only the fusion / tiling differs.  "genfuse" (0|1: generator output and
 * next-input kernels folded into the small-grid step kernel), "diag_fused" (0|1: one-kernel diagnostics increment),
 * "diag_wide" (-1 auto|0|1: its transforms as (member, transform) workgroups),
 * "lsplit" (-1 auto|0|1: one workgroup per member and layer), "spec_threads" (0 auto|256|512|1024), "team" (0|1:
 * XCD-resident runs at 256 x 256), "team_min" (shortest such run), "large_fused", "large_lazy_q",
 * "large_specialised" (0|1: the large-grid kernel variants).  The library reads NO environment variable. */
int qgx_set_option(qgx_model *m, const char *name, int value);

/* status reductions of pyqg's _print_status: out_dev[2*b+0] = KE, [2*b+1] = CFL (of ph,u,v as the last step stored
 * them; after steps with refresh_diag == 0 the current state is inverted first) */
int qgx_status_ke_cfl(qgx_model *m, double *out_dev, void *stream);

/* ---- time-averaged spectral diagnostics ------------------------------------------------
 * pyqg model.py::_calc_diagnostics / _increment_diagnostics; consumed by the reference in
 * tools/comparison_tools.py:91-188.  Accumulated inside qgx_step before every step with
 * tc >= start_step and tc % every == 0 (pyqg: t >= tavestart, tc % ceil(taveint/dt) == 0). */
enum qgx_diag {              /* per member; pyqg normalisation 1/M^2 */
    QGX_D_KESPEC = 0,        /* (B,2,N,N/2+1)  wv2 |ph|^2                         */
    QGX_D_ENSSPEC = 1,       /* (B,2,N,N/2+1)  |qh|^2                             */
    QGX_D_ENTSPEC = 2,       /* (B,N,N/2+1)    |del1 qh1 + del2 qh2|^2            */
    QGX_D_APEFLUX = 3,       /* (B,N,N/2+1)                                       */
    QGX_D_KEFLUX = 4,
    QGX_D_APEGENSPEC = 5,
    QGX_D_KEFRICTIONSPEC = 6,
    QGX_D_PARAMSPEC = 7,
    QGX_D_PARAMSPEC_APEFLUX = 8,   /* the APE and KE parts of paramspec (comparison_tools.py:174-176) */
    QGX_D_PARAMSPEC_KEFLUX = 9,
    /* (B,N,N/2+1) each: the filter's dissipation of energy / barotropic enstrophy and the barotropic-enstrophy budget
     * (flux, generation, bottom friction, parameterization) — the remaining keys of comparison_tools.py:222-225,365-368 */
    QGX_D_DISSSPEC = 10,
    QGX_D_ENSDISSSPEC = 11,
    QGX_D_ENSFLUX = 12,
    QGX_D_ENSGENSPEC = 13,
    QGX_D_ENSFRICTIONSPEC = 14,
    QGX_D_ENSPARAMSPEC = 15
};
int qgx_diag_config(qgx_model *m, int64_t start_step, int every);   /* every <= 0 disables */
int qgx_diag_get(qgx_model *m, int diag, double *out_dev, void *stream);   /* time mean */
int64_t qgx_diag_count(const qgx_model *m);
int qgx_diag_reset(qgx_model *m);

/* ---- generator ---------------------------------------------------------------
 * Replaces AndrewCNN inference through apply_function (cnn_tools.py:125-176,
 * 702-735) for CGANRegression.G / CVAERegression.decoder / MeanVarModel nets. */
enum qgx_gen_kind { QGX_GEN_GAN = 0, QGX_GEN_VAE = 1, QGX_GEN_GZ = 2 };

typedef struct qgx_cnn_weights {      /* host pointers, float32, PyTorch layouts */
    int32_t n_in, n_out;              /* 4/2 and 2                                   */
    const float *conv_w[8];           /* (cout, cin, k, k)                           */
    const float *conv_b[8];           /* (cout)                                      */
    const float *bn_gamma[7], *bn_beta[7], *bn_mean[7], *bn_var[7];
    float bn_eps;                     /* 1e-5                                        */
} qgx_cnn_weights;

/* nets: GAN / VAE — the generator / decoder (n_in 4), optionally followed by the regression net `net_mean` (n_in 2) of a
 * model trained with regression != 'None' (cgan_regression.py:59-60, cvae_regression.py:49-50): n_nets 1 or 2;
 * GZ — net_mean, net_var (n_in 2): n_nets 2 (mean_var_model.py:82-100). */
int qgx_generator_create(int kind, const qgx_cnn_weights *nets, int n_nets,
                         const float x_std[2], const float y_std[2], int device,
                         qgx_generator **out);
int qgx_generator_destroy(qgx_generator *g);
/* S = y_std * G([q/x_std, z]) — with a regression net S = y_std * (G([q/x_std, z]) + net_mean(q/x_std)), summed in
 * float32 — (cgan_regression.py:157-162; cvae_regression.py:131-136; mean_var_model.py:105-109).  demean != 0 also applies parameterization.py:25.
 * z is float for GAN/VAE, double for GZ. */
int qgx_generator_forward(qgx_generator *g, const double *q_dev, const void *z_dev,
                          double *S_dev, int B, int N, int demean, void *stream);
/* raw CNN forward of net `inet`: x (B,n_in,N,N) float -> y (B,n_out,N,N) float
 * (AndrewCNN.forward in eval mode; used by predict_mean_snapshot / offline sampling) */
int qgx_cnn_forward(qgx_generator *g, int inet, const float *x_dev, float *y_dev,
                    int B, int N, void *stream);

/* ---- coarse-graining / re-gridding building blocks ----------------------------------------
 * The reference's operators (tools/operators.py:84-99,117-217,241-268) are compositions of these.
 * A model handle doubles as the FFT plan of its grid: fields are (B,2,N,N) / (B,2,N,N/2+1). */
/* numpy-compatible rfft2 / irfft2 of the model's grid; does not touch the model state
 * (pyqg m.fft / m.ifft, operators.py:244-246). */
int qgx_rfft2(qgx_model *m, const double *x_dev, double *xh_dev, void *stream);
int qgx_irfft2(qgx_model *m, const double *xh_dev, double *x_dev, void *stream);
/* dst (nfields,N,N/2+1) <- scale * filter * [resolved block of src (nfields,n,n/2+1)]: the mode
 * transfer of cut_off (operators.py:123-130) and fft_interpolate (:148-189); zero_src_2h zeroes
 * src[h,0] first, zero_dst_2h zeroes dst[h,0] and dst[:,h] (h = min(n,N)/2); filter_dev is an
 * optional real (N,N/2+1) table on the destination grid (gauss_filter :87-90, model_filter :97-99). */
int qgx_spec_regrid(const double *src_dev, double *dst_dev, int nfields, int n, int N, double scale,
                    int zero_src_2h, int zero_dst_2h, const double *filter_dev, void *stream);
/* out = ik*A + il*B on an N-grid of domain size L (divergence, operators.py:241-247); A or B may be NULL */
int qgx_spec_div(const double *ah_dev, const double *bh_dev, double *out_dev, int nfields, int N,
                 double L, void *stream);
/* out = alpha * a * b + beta * c  (b, c optional): the pointwise products of advect (:258-266) */
int qgx_real_fma(const double *a_dev, const double *b_dev, double *out_dev, size_t n, double alpha,
                 const double *c_dev, double beta, void *stream);

/* sum += y, sumsq += y*y over Monte-Carlo samples (generate_mean_var, cgan_regression.py:139-146;
 * cvae_regression.py:120-126), accumulated in float64 */
int qgx_moments_accumulate(const float *y_dev, double *sum_dev, double *sumsq_dev, size_t n, void *stream);
/* kernel-variant switches for in-process A/B measurement (bench_tools/ab_conv.py): "chunk" (16|32 input
 * channels staged per pass of k_conv), "v3" (-1 auto | 0 | 1 | 2: LDS-operand kernel k_conv3),
 * "last_valu" (0|1), "first_split" (1|2|4 output-channel slices of the first layer). */
int qgx_generator_set_option(qgx_generator *g, const char *name, int value);
/* f16x3 range guard (no reference counterpart: the reference evaluates its nets in plain float32).  The default
 * generator arithmetic carries float32 operands as f16 hi/lo pairs, which is float32-class only inside a value window;
 * qgx_generator_create calibrates each net (exact-f32 evaluation of calibration inputs) and picks the activation
 * pre-scale — or makes the exact-f32 kernels the default when the window cannot be met — and every kernel that stores
 * 16-bit activations raises a sticky flag when a value leaves the f16 range.
 * _range_read: synchronises `stream`, returns and clears the flags (bit l = layer l+1 overflowed, bit 31 = non-finite
 * forcing) and the largest |network input| seen since the last read (inputs beyond 65504, or NaN = inf, overflow too).
 * _info: what calibration decided (precision 0 | 3, log2 of the activation pre-scale, fold 0 | 1) and the calibration
 * maxima per layer (10 floats: [0..6] stored activations, [8] layer 1 before its BatchNorm). */
int qgx_generator_range_read(qgx_generator *g, unsigned *flags, float *input_absmax, void *stream);
int qgx_generator_info(const qgx_generator *g, int *precision, int *ascale_log2, int *fold, float *layer_absmax);
/* The 5x5 layer's 1-D Winograd form (conv_wino.hpp; 0.4 x the multiplications of the 25-tap form at 64 x 64): enabled for a
 * generator only if, at qgx_generator_create, its outputs on calibration inputs stayed within 1e-5 of the exact-f32 kernels'
 * (relative to the largest output).  -> whether it is in use, what calibration decided, and the error it measured. */
int qgx_generator_wino_info(const qgx_generator *g, int *enabled, int *chosen_by_calibration, float *calibration_error);
/* ... the same per grid size: the calibration is made at every size the kernels are specialised for (32, 48, 64, 96, 128 — the
 * tile shapes differ between them) and the form is admitted size by size; _wino_info reports the 64 x 64 entry.  Another N:
 * enabled = 0, error = inf.  (No reference counterpart; the arithmetic it guards is cnn_tools.py:79-98,125-176.) */
int qgx_generator_wino_info_n(const qgx_generator *g, int N, int *enabled, int *chosen_by_calibration, float *calibration_error);
/* Which kernel the 5x5 layer of net `inet` takes for B members at N x N under the options in force: 0 exact-f32 MFMA, 1 the
 * 25-tap f16x3 kernel, 2 its split-K form (tiny ensembles), 3 1-D Winograd (k_convw), 4 1-D Winograd with the input transform
 * under the MFMAs (k_convw2).  For measurement code (bench.py's roofline names the kernel it times). */
int qgx_generator_layer2_kernel(const qgx_generator *g, int inet, int B, int N, int *kernel);
/* Measurement hook (bench.py roofline leg; no reference counterpart): bracket every launch (every n-th: option "prof_every")
 * of conv layer `layer` (0..7, -1 = off) of every net with HIP events on the launch stream; _read synchronises those events,
 * returns their summed duration and the launch count, and clears the record. */
int qgx_generator_profile(qgx_generator *g, int layer);
int qgx_generator_profile_read(qgx_generator *g, double *total_ms, int64_t *launches);

/* ---- latent noise ------------------------------------------------------------
 * z <- a z + b xi with xi ~ N(0,1) from Philox4x32-10 (stochastic_pyqg.py:43-49). */
int qgx_noise_normal(void *z_dev, int is_double, int B, int n_per_member, uint64_t seed,
                     uint64_t member_offset, uint64_t step, double a, double b, void *stream);

const char *qgx_last_error(void);
const char *qgx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QGX_H */
