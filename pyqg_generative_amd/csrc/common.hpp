// Shared host-side declarations of the qgx engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <string>
#include <memory>
#include "../../include/qgx.h"

namespace qgx {

void set_error(const char *fmt, ...);

#define QGX_HIP(call)                                                              \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            qgx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                           __FILE__, __LINE__);                                    \
            return QGX_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

#define QGX_REQUIRE(cond, ...)                                                     \
    do {                                                                           \
        if (!(cond)) {                                                             \
            qgx::set_error(__VA_ARGS__);                                           \
            return QGX_ERR_INVALID;                                                \
        }                                                                          \
    } while (0)

// entry points that touch the model state refuse a plan-only handle (qgx_config::plan_only)
#define QGX_NEEDS_STATE(m, what)                                                            \
    do {                                                                                   \
        if ((m) && (m)->plan_only) {                                                       \
            qgx::set_error("%s: the handle is an FFT plan only (created with plan_only), it has no model state", what); \
            return QGX_ERR_STATE;                                                          \
        }                                                                                  \
        if ((m) && !(m)->broken.empty()) {                                                 \
            qgx::set_error("%s: the model state is invalid: %s", what, (m)->broken.c_str()); \
            return QGX_ERR_STATE;                                                          \
        }                                                                                  \
    } while (0)

constexpr int MAX_RADIX_PASSES = 12;

// Kernel-path switches of a model (qgx_set_option).  They select between equivalent kernels of the product library
// (cross-checks in tests/, A/B timing in bench_tools/); none changes what is computed beyond rounding order.
struct ModelOpts {
    int genfuse = 1;             // small grids: generator output / next-input kernels folded into the step kernel
    int diag_fused = 1;          // small grids: one-kernel diagnostics increment (0: one launch per transform, diag.hip)
    int diag_reg = 1;            // small grids up to 64 x 64: the one-kernel increment with its work fields in registers (k_diag_small_reg):
                                 //   0 off | 1 on, as two workgroups per member while 2 B <= 256 | 2 always one workgroup per member | 3 always two
    int diag_wide = -1;          // small grids: the increment's transforms as (member, transform) workgroups, 3 launches
                                 //   (-1: while 6 B <= 256 | 0 | 1)
    int lsplit = -1;             // small grids: two workgroups per member, one per layer (-1 auto | 0 | 1)
    int spec_threads = 0;        // small grids: threads of the one-workgroup-per-member kernels (0 auto | 256 | 512 | 1024)
    int siblings = -1;           // small grids in layer-split form with a forcing: FOUR workgroups per member — the forcing's transform on a
                                 //   workgroup of its own beside the inversion / advection chain of its layer, joined by a flag in memory
                                 //   (-1: while 4 B <= 256, i.e. all of them resident at once: +2 ... +9 % of the online step | 0 | 1 |
                                 //   2: as 1, always with the cross-XCD publication — the path siblings on different XCDs take)
    int split_adv = 0;           // small grids in layer-split form + generator: the half of the step kernel that needs nothing of the forcing
                                 //   (inversion, advection, its transform) as a kernel of its own on a side stream, under the generator's
                                 //   layers (0 | 1).  Bit-identical and measured SLOWER almost everywhere (the two cross-stream
                                 //   dependencies per step cost more than the two transforms taken out of the chain: DESIGN.md section 3.1c),
                                 //   so it is off unless asked for
    int streams = 0;             // small grids + generator: the two halves of the ensemble on two internal streams (0 auto: 96 x 96, 16..64 even members | 1 never | 2 whenever even)
    int team = 1;                // 256 x 256: XCD-resident runs of unparameterized steps
    int team_min = 2;            //   shortest run handed to that kernel
    int team_fault = 0;          //   A/B library only: raise the run kernel's flag at the end of the next run (test hook)
    int step_fault = 0;          //   A/B library only: half-ensemble (1 | 2) of a two-stream qgx_step refuses its second chunk (test hook)
    int large_fused = 1;         // large grids: fused row / column kernels (0: one launch per pass and pointwise phase)
    int large_lazy_q = 1;        // large grids: unparameterized steps keep no real-space q
    int large_specialised = 1;   // large grids: compile-time-N kernels at 128 / 256 / 512
};

// Tile-shape tuning aids read from the environment exist in the A/B library only (make ab); the product library
// always runs the measured defaults, so no environment variable can alter a production run.
static inline int tune_env(const char *name, int dflt) {
#ifdef QGX_AB
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// Device-visible description of one model's grid and constants (passed by value).
struct SpecDev {
    int N, NK, LD, B;
    int nrad;
    int rad[MAX_RADIX_PASSES];      // DIF pass radices, product = N
    const double *filtr, *wv2, *a;  // a: 4 tables of N*NK (a00,a01,a10,a11)
    const double *kk, *ll;
    const double2 *tw;              // tw[t] = exp(-2 pi i t / N)
    const int *pos;                 // pos[freq] -> position after the DIF passes
    double U[2], Qy[2];
    double rek, invN2, dt, dx;
    double H[2], Htot;
};

// Work the spectral step kernel does on the generator's behalf (small grids in layer-split form, GAN / VAE):
//  * y != null: the forcing is still the net's raw output (B,2,N,N) float; the kernel's prologue does what k_finish<FIN_PLAIN>
//    does — S = double(y * y_std) - mean_{y,x} — with the same arithmetic and summation order, and stores S;
//  * X != null: the kernel's epilogue assembles the NEXT step's network input from q^{n+1} and fresh white noise,
//    X = [float(q)/x_std, z], z = b * xi(Philox; seed, member, step) — what k_prep_noise does with a == 0.
struct GenFuse {
    const float *y = nullptr;
    const float *y1 = nullptr;      // regression net's output, summed with y in float32 before the scaling (k_finish<FIN_SUM>)
    float ys[2] = {1.f, 1.f};
    int demean = 0;
    unsigned *range = nullptr;      // the generator's range-guard words (conv.hip)
    float *X = nullptr;
    float *z = nullptr;
    float xs[2] = {1.f, 1.f};
    float b = 1.f;
    uint64_t seed = 0, member_offset = 0, step = 0;
};

// Arguments of one time step (spectral_small.hip / spectral_large.hip).
struct StepArgs {
    const double2 *qh_in;   // (B,2,N,NK)
    double2 *qh_out;
    double *q;              // (B,2,N,N) in: q^n, out: q^{n+1}
    const double *S;        // (B,2,N,N) or null
    double2 *dqh;           // spectral forcing of this step (pyqg m.dqh)
    double2 *dq_new;        // newest tendency (overwrites the oldest history slot)
    const double2 *dq_p, *dq_pp;
    double2 *ph;            // diagnostics (written when diag != 0)
    double *u, *v;
    double dt1, dt2, dt3;
    double weight;
    int has_S, demean, diag;
    GenFuse gf;
    // k_step_small PART 3 (forcing and advection workgroups of a (member, layer) in one launch): one word per (member, layer),
    // set to the launch's epoch by the forcing workgroup once the forcing's spectrum is in memory
    unsigned long long *sib_flag = nullptr;
    unsigned long long sib_epoch = 0;
    int sib_full = 0;           // publish across XCDs even to a sibling on the same one (option siblings = 2: exercises that path)
};

// time-averaged diagnostics (diag.hip; the small-grid increment is one kernel of spectral_small.hip)
struct DiagConst { double del1, del2, rdm2, Udiff, rek, invM2, H0, H1;
                   double dt1, dt2, dt3, invdt; };       // AB coefficients of the step about to be taken (Dissspec)
struct DiagAcc { double *KEspec, *Ensspec, *entspec, *APEflux, *KEflux, *APEgenspec, *KEfrictionspec, *paramspec,
                        *paramspec_APEflux, *paramspec_KEflux,
                        *Dissspec, *ENSDissspec, *ENSflux, *ENSgenspec, *ENSfrictionspec, *ENSparamspec; };
constexpr int N_DIAGS = 16;

}  // namespace qgx

struct qgx_generator;

struct qgx_model {
    qgx_config cfg;
    qgx::SpecDev d;
    qgx::ModelOpts opts;
    int N, NK, B;
    bool small;                     // whole member field fits one CU's LDS
    bool plan_only = false;         // FFT plan of the grid only: tables + work space, no model state (qgx_config::plan_only)
    // tables (device) + host copies for qgx_get_table
    double *t_filtr = nullptr, *t_wv2 = nullptr, *t_a = nullptr, *t_kk = nullptr, *t_ll = nullptr;
    double2 *t_tw = nullptr;
    int *t_pos = nullptr;
    // (behind one pointer: qgx_step copies the model's bookkeeping for its two half-ensembles on every call)
    struct HostTables { std::vector<double> filtr, wv2, a, kk, ll; };
    std::shared_ptr<HostTables> host;
    // a step that failed half-way (one half-ensemble advanced, the other not; a stream operation refused after the fork) leaves
    // device buffers and bookkeeping out of step: every later call on the handle fails loudly with this message
    std::string broken;
    // state (device)
    double *q = nullptr, *u = nullptr, *v = nullptr, *S = nullptr;
    double2 *qh[2] = {nullptr, nullptr};   // ping-pong; qh[cur_q] is current
    double2 *ph = nullptr, *dqh = nullptr; // dqh: spectral forcing of the last step (pyqg m.dqh)
    double2 *dq[4] = {nullptr, nullptr, nullptr, nullptr};   // AB3 tendency history; [3]: large grids, spare slot of the run kernel
    double2 *zbuf = nullptr;               // large-N path: (B,3,N,N) complex work array
    bool q_stale = false;                  // large-N path: q lags qh (unparameterized steps keep no real-space q)
    bool uv_stale = false;                 // the last step did not store ph, u, v (refresh_diag == 0): status inverts first
    void *team_ctl = nullptr;              // large-N path: census / barrier block of the XCD-resident step kernel
    int team_state = 0;                    //   0 not probed, 1 available, -1 not available
    bool team_pending = false;             //   a run was launched whose flags have not been read back yet
    struct TeamUndo {                      //   bookkeeping as it was before the pending run (a run writes ONLY to buffers
        int K = 0;                         //   that hold nothing live, so a run that raised a flag is undone by restoring
        int cur_q, i_new, i_p, i_pp, i_x;  //   these and replayed on the three-launch path)
        int64_t tc;
        int ablevel;
        bool uv_stale, q_stale;
    } team_undo;
    hipStream_t team_stream = nullptr;     //   the stream the pending run was launched on (its flag is read back there)
    int team_replays = 0;                  //   runs that raised a flag and were replayed
    int cur_q = 0;
    int i_new = 0, i_p = 1, i_pp = 2, i_x = 3;   // roles of dq[]
    void *z = nullptr;                     // latent noise (B,2,N,N) float or double
    void *xi = nullptr;                    // scratch white noise for AR1
    bool z_double = false;
    bool have_noise = false;
    int64_t const_counter = 0;
    bool have_forcing = false;
    const void *x_ready_gen = nullptr;     // the generator whose input the last step kernel assembled (GenFuse::X), or null
    uint64_t x_ready_step = 0;             //   ... for this noise step
    int64_t tc = 0;
    int ablevel = 0;
    uint64_t noise_step = 0;
    // time-averaged diagnostics (diag.hip)
    int64_t dg_start = 0, dg_count = 0;
    int dg_every = 0;
    double *dg_R[7] = {};
    double *dg_S[7] = {};
    double *dg_acc[qgx::N_DIAGS] = {};
    double2 *dg_z = nullptr;               // large grids: the four work fields of the three-launch increment (spectral_large.hip)
    // the two internal streams of a step in halves (model.hip::qgx_step) and their fork / join events
    // side stream and fork / join events of the two-kernel step (model.hip::step_core); a half-ensemble uses set adv_slot
    unsigned long long *sib_flag = nullptr;      // (B, 2) words of the four-workgroup step, and the epoch of its last launch
    unsigned long long sib_epoch = 0;
    hipStream_t adv_stream[2] = {nullptr, nullptr};
    hipEvent_t adv_event[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int adv_slot = 0;
    bool is_half = false;                        // a half-ensemble view of a model stepped as two halves on two streams (model.hip::qgx_step)
    hipStream_t sub_stream[2] = {nullptr, nullptr};
    hipEvent_t sub_event[3] = {nullptr, nullptr, nullptr};
};

namespace qgx {
// generator entry used by the stepper (conv.hip)
// optional sampler update z <- a z + b xi folded into the generator's input kernel
struct NoiseUpdate {
    const void *xi_ext;      // external draw or nullptr (Philox)
    uint64_t seed, member_offset, step;
    double a, b;
};
// defer != null (GAN / VAE): the output kernel is skipped and *defer describes what the step kernel has to do instead
// (GenFuse::y ...); input_ready: the previous step kernel already assembled the network input (GenFuse::X ...)
int generator_forward(qgx_generator *g, const double *q, const void *z, double *S, int B, int N,
                      int demean, hipStream_t st, const NoiseUpdate *nu, GenFuse *defer = nullptr, bool input_ready = false);
// the generator's input buffer, input scales and range words for the GenFuse::X part (after reserve)
int generator_input_info(qgx_generator *g, int B, int N, GenFuse *gf);
bool small_layer_split(const SpecDev &d, const ModelOpts &o);
// spectral_small.hip
bool small_path_fits(int N);
int small_prepare(const SpecDev &d);
int small_step(const SpecDev &d, const ModelOpts &o, const StepArgs &a, hipStream_t st, int part = 0);
int small_q_to_qh(const SpecDev &d, const ModelOpts &o, const double *q, double2 *qh, hipStream_t st);
int small_qh_to_q(const SpecDev &d, const ModelOpts &o, const double2 *qh, double *q, hipStream_t st);
int small_invert(const SpecDev &d, const ModelOpts &o, const double2 *qh, double2 *ph, double *u, double *v, hipStream_t st);

bool generator_noise_is_double(const qgx_generator *g);
int diag_increment(qgx_model *m, const double *S, double weight, hipStream_t st);
int diag_ensure_alloc(qgx_model *m);      // the increment's work fields and accumulators (allocated at first use)
// the generator's activation workspace for the calls that follow (1: the second half of an ensemble stepped in halves)
int generator_select_workspace(qgx_generator *g, int idx);
int noise_update(void *z, const void *xi_ext, bool is_double, int B, int n_per_member, uint64_t seed,
                 uint64_t member_offset, uint64_t step, double a, double b, hipStream_t st);
int noise_normal(void *z, bool is_double, int B, int n_per_member, uint64_t seed,
                 uint64_t member_offset, uint64_t step, double a, double b, hipStream_t st);
}  // namespace qgx
