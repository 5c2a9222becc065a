// Generator forward: the 8-layer circular-padded CNN as implicit-GEMM convolutions on the
// exact-f32 matrix cores (v_mfma_f32_32x32x2_f32), fused bias + ReLU + eval-mode BatchNorm.
//
// Replaces AndrewCNN.forward evaluated through apply_function
// (pyqg_generative/tools/cnn_tools.py:79-98 make_block = Conv2d('same', circular) -> ReLU ->
// BatchNorm2d; :125-176 channels [n_in,128,64,32,32,32,32,32,n_out], kernels [5,5,3,3,3,3,3,3];
// :702-735 eval mode, no grad) and the model wrappers
// models/cgan_regression.py:157-162, cvae_regression.py:131-136, mean_var_model.py:105-109,
// plus the per-layer de-mean of models/parameterization.py:25.
//
// Data layout: activations NHWC float32 (B, N, N, C) so that the GEMM K index
// (tap, input channel) is contiguous per pixel; the first layer reads a small planar
// (B, n_in, N, N) input and the last writes planar (B, n_out, N, N).
// One workgroup (4 waves) owns R full-width image rows (M = R*N pixels, all output
// channels); the input patch of (R+K-1) rows is staged once per input-channel chunk
// into LDS with a padded pixel stride, horizontal wrap-around is resolved when the
// A fragment address is formed, vertical wrap-around when the patch is staged.
// MFMA 32x32x2 f32: A[i = lane&31][k = lane>>5] = pixel x K, B[k][j = lane&31] = K x cout.
// Lane half h consumes K indices 8g+4h .. 8g+4h+3 of every group of 8 with one 16-byte
// read of A (LDS) and of B (packed weights, L2 resident) feeding 4 consecutive MFMAs.
#include "common.hpp"
#include <cmath>
#include <new>

namespace qgx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float *in;       // NHWC (B,N,N,CIN) or planar (B,CIN,N,N)
    float *out;            // NHWC (B,N,N,COUT) or planar (B,COUT_REAL,N,N)
    const float *w;        // packed [kgroup][COUTP][8]
    const float *bias, *scale, *shift;   // [COUTP]
    int N, R, cout_real;
};

extern __shared__ __attribute__((aligned(16))) char conv_smem[];

template <int CIN, int COUT, int KS, int CC, int MT, bool PLANAR_IN, bool FINAL>
__global__ __launch_bounds__(256) void k_conv(ConvArgs a) {
    constexpr int NT = (COUT + 31) / 32;
    constexpr int COUTP = NT * 32;
    constexpr int P = KS / 2;
    constexpr int T = KS * KS;
    constexpr int STRIDE = PLANAR_IN ? CIN : CC + 4;   // floats per patch pixel
    float *patch = reinterpret_cast<float *>(conv_smem);

    const int N = a.N, R = a.R;
    const int tiles_per_img = N / R;
    const int b = blockIdx.x / tiles_per_img;
    const int y0 = (blockIdx.x - b * tiles_per_img) * R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int ntiles = R * N / 32;
    const int PR = R + KS - 1;

    int py[MT], px[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int tile = wave + 4 * mt;
        if (tile >= ntiles) tile = wave;           // duplicate work, never stored
        const int p = tile * 32 + li;
        py[mt] = p / N;
        px[mt] = p - py[mt] * N;
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    if constexpr (PLANAR_IN) {
        // ---- stage the whole (tiny) input patch: patch[(pr*N + x)*CIN + c]
        for (int it = threadIdx.x; it < PR * CIN * N; it += 256) {
            const int x = it % N;
            const int c = (it / N) % CIN;
            const int pr = it / (N * CIN);
            int gy = y0 - P + pr;
            gy = gy < 0 ? gy + N : (gy >= N ? gy - N : gy);
            patch[(pr * N + x) * CIN + c] = a.in[(((size_t)b * CIN + c) * N + gy) * N + x];
        }
        __syncthreads();
        constexpr int NG = (T * CIN + 7) / 8;
        const float4 *wp = reinterpret_cast<const float4 *>(a.w) + (size_t)li * 2 + h;
        for (int g = 0; g < NG; ++g) {
            float4 A[MT], Bf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Bf[nt] = wp[((size_t)g * COUTP + nt * 32) * 2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (CIN == 4) {
                    int tap = 2 * g + h;
                    if (tap >= T) tap = 0;             // zero weights there
                    const int ky = tap / KS, kx = tap - ky * KS;
                    int col = px[mt] + kx - P;
                    col = col < 0 ? col + N : (col >= N ? col - N : col);
                    A[mt] = *reinterpret_cast<const float4 *>(&patch[((py[mt] + ky) * N + col) * 4]);
                } else {   // CIN == 2: two taps per lane half
                    float2 v[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        int tap = 4 * g + 2 * h + e;
                        if (tap >= T) tap = 0;
                        const int ky = tap / KS, kx = tap - ky * KS;
                        int col = px[mt] + kx - P;
                        col = col < 0 ? col + N : (col >= N ? col - N : col);
                        v[e] = *reinterpret_cast<const float2 *>(&patch[((py[mt] + ky) * N + col) * 2]);
                    }
                    A[mt] = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            (&A[mt].x)[e], (&Bf[nt].x)[e], acc[mt][nt], 0, 0, 0);
        }
    } else {
        constexpr int G8 = CC / 8;
        constexpr int C4 = CC / 4;
        const float4 *wp = reinterpret_cast<const float4 *>(a.w) + (size_t)li * 2 + h;
        for (int c0 = 0; c0 < CIN; c0 += CC) {
            __syncthreads();
            // ---- stage patch chunk: (PR rows) x N x CC channels, pixel stride CC+4 floats
            for (int it = threadIdx.x; it < PR * N * C4; it += 256) {
                const int c4 = it % C4;
                const int x = (it / C4) % N;
                const int pr = it / (C4 * N);
                int gy = y0 - P + pr;
                gy = gy < 0 ? gy + N : (gy >= N ? gy - N : gy);
                const float4 vv = *reinterpret_cast<const float4 *>(
                    &a.in[(((size_t)b * N + gy) * N + x) * CIN + c0 + c4 * 4]);
                *reinterpret_cast<float4 *>(&patch[(pr * N + x) * STRIDE + c4 * 4]) = vv;
            }
            __syncthreads();
            for (int ky = 0; ky < KS; ++ky)
                for (int kx = 0; kx < KS; ++kx) {
                    int aoff[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        int col = px[mt] + kx - P;
                        col = col < 0 ? col + N : (col >= N ? col - N : col);
                        aoff[mt] = ((py[mt] + ky) * N + col) * STRIDE + 4 * h;
                    }
#pragma unroll
                    for (int g8 = 0; g8 < G8; ++g8) {
                        float4 A[MT], Bf[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) Bf[nt] = wp[((size_t)g8 * COUTP + nt * 32) * 2];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            A[mt] = *reinterpret_cast<const float4 *>(&patch[aoff[mt] + g8 * 8]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                        (&A[mt].x)[e], (&Bf[nt].x)[e], acc[mt][nt], 0, 0, 0);
                    }
                    wp += (size_t)G8 * COUTP * 2;
                }
        }
    }

    // ---- epilogue: bias (+ ReLU + BatchNorm affine), store
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int tile = wave + 4 * mt;
        if (tile >= ntiles) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 32 + li;
            const float bias = a.bias[co];
            if constexpr (FINAL) {
                if (co < a.cout_real) {
                    float *o = a.out + ((size_t)b * a.cout_real + co) * N * N + (size_t)y0 * N;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        o[p] = acc[mt][nt][r] + bias;
                    }
                }
            } else {
                const float sc = a.scale[co], sh = a.shift[co];
                float *o = a.out + ((size_t)b * N * N + (size_t)y0 * N) * COUT + co;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float vv = acc[mt][nt][r] + bias;
                    vv = fmaxf(vv, 0.f);
                    o[(size_t)p * COUT] = vv * sc + sh;
                }
            }
        }
    }
}

// ---- small pointwise kernels around the CNN ---------------------------------------------
// X = [float(q)/x_std, z]  (cgan_regression.py:158 + generate :133-137)
__global__ void k_prep_input(const double *q, const float *z, float *X, int n_in, int npix, float xs0, float xs1) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const size_t qo = (size_t)b * 2 * npix + i;
        float *x = X + (size_t)b * n_in * npix + i;
        x[0] = (float)q[qo] / xs0;
        x[npix] = (float)q[qo + npix] / xs1;
        if (n_in == 4) {
            x[2 * (size_t)npix] = z[qo];
            x[3 * (size_t)npix] = z[qo + npix];
        }
    }
}

// S = double(y * y_std)   (cgan_regression.py:162)
__global__ void k_finish_gan(const float *y, double *S, int npix, float ys0, float ys1) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * npix; i += gridDim.x * blockDim.x) {
        const size_t o = (size_t)b * 2 * npix + i;
        S[o] = (double)(y[o] * (i < npix ? ys0 : ys1));
    }
}

// S = (mean + z * sqrt(softplus(var))) * y_std   (mean_var_model.py:14-17,105-109), z double
__global__ void k_finish_gz(const float *ymean, const float *yvar, const double *z, double *S, int npix,
                            float ys0, float ys1) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * npix; i += gridDim.x * blockDim.x) {
        const size_t o = (size_t)b * 2 * npix + i;
        const float vr = yvar[o];
        const float sp = vr > 20.f ? vr : log1pf(expf(vr));
        const double val = (double)ymean[o] + z[o] * (double)sqrtf(sp);
        S[o] = val * (double)(i < npix ? ys0 : ys1);
    }
}

// S -= mean_{y,x}(S) per (member, layer)   (parameterization.py:25)
__global__ void k_demean(double *S, int npix) {
    __shared__ double sm[16];
    __shared__ double mean_s;
    double *s = S + (size_t)blockIdx.x * npix;
    double acc = 0.0;
    for (int i = threadIdx.x; i < npix; i += blockDim.x) acc += s[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
        mean_s = t / (double)npix;
    }
    __syncthreads();
    const double mu = mean_s;
    for (int i = threadIdx.x; i < npix; i += blockDim.x) s[i] -= mu;
}

// ---- host side ---------------------------------------------------------------------------
struct LayerHost {
    int cin, cout, ks, coutp, cc, ngroups;
    float *w = nullptr, *bias = nullptr, *scale = nullptr, *shift = nullptr;
};
struct NetHost {
    int n_in, n_out;
    LayerHost L[8];
};

}  // namespace qgx

struct qgx_generator {
    int kind, device, n_nets;
    qgx::NetHost nets[2];
    float x_std[2], y_std[2];
    // workspace (grown on demand, outside any captured region)
    size_t cap_elems = 0;          // capacity in units of B*N*N pixels
    float *actA = nullptr, *actB = nullptr, *X = nullptr, *Y0 = nullptr, *Y1 = nullptr;
    // optional per-layer timing with HIP events on the launch stream (bench.py roofline leg)
    int prof_layer = -1;
    std::vector<hipEvent_t> prof_ev;    // pairs (start, stop)
    size_t prof_used = 0;
};

namespace qgx {

static const int KSZ[8] = {5, 5, 3, 3, 3, 3, 3, 3};
static const int HID[7] = {128, 64, 32, 32, 32, 32, 32};

static int pack_layer(LayerHost &L, int li, const qgx_cnn_weights *w, bool planar_in) {
    const int cin = L.cin, cout = L.cout, ks = L.ks, T = ks * ks;
    L.coutp = ((cout + 31) / 32) * 32;
    L.cc = planar_in ? cin : 32;
    L.ngroups = planar_in ? (T * cin + 7) / 8 : (cin / L.cc) * T * (L.cc / 8);
    std::vector<float> pw((size_t)L.ngroups * L.coutp * 8, 0.f);
    const float *W = w->conv_w[li];
    for (int g = 0; g < L.ngroups; ++g)
        for (int co = 0; co < cout; ++co)
            for (int e = 0; e < 8; ++e) {
                int tap, c;
                if (planar_in) {
                    const int kidx = 8 * g + e;
                    if (kidx >= T * cin) continue;
                    tap = kidx / cin; c = kidx % cin;
                } else {
                    const int g8n = L.cc / 8;
                    const int g8 = g % g8n, tt = (g / g8n) % T, chunk = g / (g8n * T);
                    tap = tt; c = chunk * L.cc + g8 * 8 + e;
                }
                const int ky = tap / ks, kx = tap % ks;
                pw[((size_t)g * L.coutp + co) * 8 + e] = W[(((size_t)co * cin + c) * ks + ky) * ks + kx];
            }
    std::vector<float> bias(L.coutp, 0.f), sc(L.coutp, 1.f), sh(L.coutp, 0.f);
    for (int co = 0; co < cout; ++co) {
        bias[co] = w->conv_b[li][co];
        if (li < 7) {
            // eval-mode BatchNorm as PyTorch evaluates it: alpha = gamma * invstd, y = x*alpha + (beta - mean*alpha)
            const float invstd = 1.0f / sqrtf(w->bn_var[li][co] + w->bn_eps);
            const float alpha = w->bn_gamma[li][co] * invstd;
            sc[co] = alpha;
            sh[co] = w->bn_beta[li][co] - w->bn_mean[li][co] * alpha;
        }
    }
    auto up = [](float *&dst, const std::vector<float> &h) -> int {
        QGX_HIP(hipMalloc((void **)&dst, h.size() * sizeof(float)));
        QGX_HIP(hipMemcpy(dst, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        return QGX_OK;
    };
    int rc;
    if ((rc = up(L.w, pw)) || (rc = up(L.bias, bias)) || (rc = up(L.scale, sc)) || (rc = up(L.shift, sh))) return rc;
    return QGX_OK;
}

static int choose_rows(int N) {
    if (N <= 256 && 256 % N == 0) return 256 / N;     // 8 M-tiles
    if (N <= 384 && 384 % N == 0) return 384 / N;     // 12 M-tiles
    return 0;
}

static int prof_begin(qgx_generator *g, int layer, hipStream_t st, hipEvent_t &stop) {
    stop = nullptr;
    if (g->prof_layer != layer) return QGX_OK;
    if (g->prof_used + 2 > g->prof_ev.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            QGX_HIP(hipEventCreate(&e));
            g->prof_ev.push_back(e);
        }
    }
    QGX_HIP(hipEventRecord(g->prof_ev[g->prof_used], st));
    stop = g->prof_ev[g->prof_used + 1];
    g->prof_used += 2;
    return QGX_OK;
}

template <int CIN, int COUT, int KS, int CC, bool PLANAR_IN, bool FINAL>
static int launch_conv(qgx_generator *g, int layer, const LayerHost &L, const float *in, float *out, int B,
                       int N, int cout_real, hipStream_t st) {
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    const int R = choose_rows(N);
    QGX_REQUIRE(R > 0 && N % R == 0, "generator: unsupported grid size N=%d", N);
    const int ntiles = R * N / 32;
    ConvArgs a;
    a.in = in; a.out = out; a.w = L.w; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.N = N; a.R = R; a.cout_real = cout_real;
    constexpr int STRIDE = PLANAR_IN ? CIN : CC + 4;
    const size_t lds = (size_t)(R + KS - 1) * N * STRIDE * sizeof(float);
    QGX_REQUIRE(lds <= 160 * 1024, "generator: LDS patch %zu B too large for N=%d", lds, N);
    dim3 grid(B * (N / R)), block(256);
    if (ntiles <= 8) {
        auto kern = k_conv<CIN, COUT, KS, CC, 2, PLANAR_IN, FINAL>;
        QGX_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    } else {
        auto kern = k_conv<CIN, COUT, KS, CC, 3, PLANAR_IN, FINAL>;
        QGX_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    }
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}

static int reserve(qgx_generator *g, int B, int N) {
    const size_t need = (size_t)B * N * N;
    if (need <= g->cap_elems) return QGX_OK;
    float **bufs[] = {&g->actA, &g->actB, &g->X, &g->Y0, &g->Y1};
    for (auto p : bufs) if (*p) { (void)hipFree(*p); *p = nullptr; }
    g->cap_elems = 0;
    QGX_HIP(hipMalloc((void **)&g->actA, need * 128 * sizeof(float)));
    QGX_HIP(hipMalloc((void **)&g->actB, need * 64 * sizeof(float)));
    QGX_HIP(hipMalloc((void **)&g->X, need * 4 * sizeof(float)));
    QGX_HIP(hipMalloc((void **)&g->Y0, need * 2 * sizeof(float)));
    QGX_HIP(hipMalloc((void **)&g->Y1, need * 2 * sizeof(float)));
    g->cap_elems = need;
    return QGX_OK;
}

// AndrewCNN.forward: x planar (B,n_in,N,N) -> y planar (B,n_out,N,N)
static int cnn_forward(qgx_generator *g, const NetHost &net, const float *x, float *y, int B, int N,
                       hipStream_t st) {
    int rc;
    float *A = g->actA, *Bb = g->actB;
    if (net.n_in == 4) rc = launch_conv<4, 128, 5, 4, true, false>(g, 0, net.L[0], x, A, B, N, 128, st);
    else rc = launch_conv<2, 128, 5, 2, true, false>(g, 0, net.L[0], x, A, B, N, 128, st);
    if (rc) return rc;
    if ((rc = launch_conv<128, 64, 5, 32, false, false>(g, 1, net.L[1], A, Bb, B, N, 64, st))) return rc;
    if ((rc = launch_conv<64, 32, 3, 32, false, false>(g, 2, net.L[2], Bb, A, B, N, 32, st))) return rc;
    if ((rc = launch_conv<32, 32, 3, 32, false, false>(g, 3, net.L[3], A, Bb, B, N, 32, st))) return rc;
    if ((rc = launch_conv<32, 32, 3, 32, false, false>(g, 4, net.L[4], Bb, A, B, N, 32, st))) return rc;
    if ((rc = launch_conv<32, 32, 3, 32, false, false>(g, 5, net.L[5], A, Bb, B, N, 32, st))) return rc;
    if ((rc = launch_conv<32, 32, 3, 32, false, false>(g, 6, net.L[6], Bb, A, B, N, 32, st))) return rc;
    if ((rc = launch_conv<32, 2, 3, 32, false, true>(g, 7, net.L[7], A, y, B, N, net.n_out, st))) return rc;
    return QGX_OK;
}

bool generator_noise_is_double(const qgx_generator *g) { return g->kind == QGX_GEN_GZ; }

int generator_forward(qgx_generator *g, const double *q, const void *z, double *S, int B, int N,
                      int demean, hipStream_t st) {
    QGX_REQUIRE(g && q && z && S && B > 0, "generator_forward: bad argument");
    int rc = reserve(g, B, N);
    if (rc) return rc;
    const int npix = N * N;
    dim3 pg((npix + 255) / 256, B), pb(256);
    if (g->kind == QGX_GEN_GZ) {
        hipLaunchKernelGGL(k_prep_input, pg, pb, 0, st, q, (const float *)nullptr, g->X, 2, npix, g->x_std[0], g->x_std[1]);
        if ((rc = cnn_forward(g, g->nets[0], g->X, g->Y0, B, N, st))) return rc;
        if ((rc = cnn_forward(g, g->nets[1], g->X, g->Y1, B, N, st))) return rc;
        dim3 fg((2 * npix + 255) / 256, B);
        hipLaunchKernelGGL(k_finish_gz, fg, pb, 0, st, g->Y0, g->Y1, (const double *)z, S, npix, g->y_std[0], g->y_std[1]);
    } else {
        hipLaunchKernelGGL(k_prep_input, pg, pb, 0, st, q, (const float *)z, g->X, 4, npix, g->x_std[0], g->x_std[1]);
        if ((rc = cnn_forward(g, g->nets[0], g->X, g->Y0, B, N, st))) return rc;
        dim3 fg((2 * npix + 255) / 256, B);
        hipLaunchKernelGGL(k_finish_gan, fg, pb, 0, st, g->Y0, S, npix, g->y_std[0], g->y_std[1]);
    }
    if (demean) hipLaunchKernelGGL(k_demean, dim3(2 * B), dim3(256), 0, st, S, npix);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

}  // namespace qgx

using namespace qgx;

extern "C" int qgx_generator_create(int kind, const qgx_cnn_weights *nets, int n_nets, const float x_std[2],
                                    const float y_std[2], int device, qgx_generator **out) {
    QGX_REQUIRE(nets && out && x_std && y_std, "qgx_generator_create: null argument");
    QGX_REQUIRE(kind == QGX_GEN_GAN || kind == QGX_GEN_VAE || kind == QGX_GEN_GZ, "unknown generator kind %d", kind);
    QGX_REQUIRE(n_nets == (kind == QGX_GEN_GZ ? 2 : 1), "generator kind %d needs %d nets", kind, kind == QGX_GEN_GZ ? 2 : 1);
    QGX_HIP(hipSetDevice(device));
    qgx_generator *g = new (std::nothrow) qgx_generator();
    if (!g) { set_error("out of host memory"); return QGX_ERR_NOMEM; }
    g->kind = kind; g->device = device; g->n_nets = n_nets;
    for (int i = 0; i < 2; ++i) { g->x_std[i] = x_std[i]; g->y_std[i] = y_std[i]; }
    for (int n = 0; n < n_nets; ++n) {
        const qgx_cnn_weights *w = &nets[n];
        const int want_in = kind == QGX_GEN_GZ ? 2 : 4;
        if (w->n_in != want_in || w->n_out != 2) {
            set_error("net %d: n_in=%d n_out=%d, expected %d and 2", n, w->n_in, w->n_out, want_in);
            qgx_generator_destroy(g);
            return QGX_ERR_INVALID;
        }
        NetHost &net = g->nets[n];
        net.n_in = w->n_in; net.n_out = w->n_out;
        for (int li = 0; li < 8; ++li) {
            LayerHost &L = net.L[li];
            L.cin = li == 0 ? w->n_in : HID[li - 1];
            L.cout = li == 7 ? w->n_out : HID[li];
            L.ks = KSZ[li];
            int rc = pack_layer(L, li, w, li == 0);
            if (rc) { qgx_generator_destroy(g); return rc; }
        }
    }
    *out = g;
    return QGX_OK;
}

extern "C" int qgx_generator_destroy(qgx_generator *g) {
    if (!g) return QGX_OK;
    (void)hipSetDevice(g->device);
    for (int n = 0; n < 2; ++n)
        for (int li = 0; li < 8; ++li) {
            LayerHost &L = g->nets[n].L[li];
            float *ptrs[] = {L.w, L.bias, L.scale, L.shift};
            for (float *p : ptrs) if (p) (void)hipFree(p);
        }
    float *bufs[] = {g->actA, g->actB, g->X, g->Y0, g->Y1};
    for (float *p : bufs) if (p) (void)hipFree(p);
    for (hipEvent_t e : g->prof_ev) (void)hipEventDestroy(e);
    delete g;
    return QGX_OK;
}

extern "C" int qgx_generator_forward(qgx_generator *g, const double *q_dev, const void *z_dev, double *S_dev,
                                     int B, int N, int demean, void *stream) {
    return generator_forward(g, q_dev, z_dev, S_dev, B, N, demean, (hipStream_t)stream);
}

extern "C" int qgx_cnn_forward(qgx_generator *g, int inet, const float *x_dev, float *y_dev, int B, int N,
                               void *stream) {
    QGX_REQUIRE(g && x_dev && y_dev && inet >= 0 && inet < g->n_nets && B > 0, "qgx_cnn_forward: bad argument");
    int rc = reserve(g, B, N);
    if (rc) return rc;
    return cnn_forward(g, g->nets[inet], x_dev, y_dev, B, N, (hipStream_t)stream);
}

extern "C" int qgx_generator_profile(qgx_generator *g, int layer) {
    QGX_REQUIRE(g && layer >= -1 && layer < 8, "qgx_generator_profile: bad argument");
    g->prof_layer = layer;
    g->prof_used = 0;
    return QGX_OK;
}

extern "C" int qgx_generator_profile_read(qgx_generator *g, double *total_ms, int64_t *launches) {
    QGX_REQUIRE(g && total_ms && launches, "qgx_generator_profile_read: null argument");
    double tot = 0.0;
    for (size_t i = 0; i + 1 < g->prof_used; i += 2) {
        QGX_HIP(hipEventSynchronize(g->prof_ev[i + 1]));
        float ms = 0.f;
        QGX_HIP(hipEventElapsedTime(&ms, g->prof_ev[i], g->prof_ev[i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int64_t)(g->prof_used / 2);
    g->prof_used = 0;
    return QGX_OK;
}
