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
#include <unordered_map>
#include "philox.hpp"
#include <cmath>
#include <utility>
#include <new>

namespace qgx {

#include "conv_types.hpp"

struct ConvArgs {
    const float *in;       // NHWC (B,N,N,CIN) or planar (B,CIN,N,N)
    float *out;            // NHWC (B,N,N,COUT) or planar (B,COUT_REAL,N,N)
    const float *w;        // packed [kgroup][COUTP][8]
    const float *bias, *scale, *shift;   // [COUTP]
    int N, R, cout_real;
    size_t npix_total;     // B*N*N (stride of one split-K partial plane)
    float ascale;          // OUTH: power-of-two pre-scale of the stored 16-bit activations
    unsigned *range;       // OUTH: f16x3 range guard flag word (conv_half.hpp::range_guard)
};

#include "conv_half.hpp"
#include "conv_pair.hpp"
#include "conv_wino.hpp"
// conv_wino2.hip (a translation unit of its own: built without the SLP vectoriser): the same layer with the input transform
// under the MFMAs, bit-identical to k_convw, for the tile shapes where it is faster; *done = false: take k_convw
int launch_convw2(int N, int TW, int R, const ConvWArgs &a, int total_tiles, hipStream_t st, bool *done);
// Kernel variants that were measured slower than the ones above (k_convh generic / plain f16, k_convh_res,
// k_convh3 on 16x16x32 MFMAs, k_convh4 with full-line chunks) are compiled only into the A/B library
// (`make ab` -> libqgx_ab.so, bench_tools/ab_conv.py): the product library carries one path per layer and size.
#ifdef QGX_AB
#include "conv_h16.hpp"
#include "conv_h4.hpp"
#endif

// OUTH = 0: f32 NHWC output.  OUTH = 1 / 2 (first layer only): the MFMA roles are swapped (lane = pixel)
// and the epilogue writes the packed f16 / f16 hi-lo activation layout of conv_half.hpp.
template <int CIN, int COUT, int KS, int CC, int MT, bool PLANAR_IN, bool FINAL, int CSPLIT = 1, bool PARTIAL = false, int OUTH = 0>
__global__ __launch_bounds__(256) void k_conv(ConvArgs a) {
    static_assert(OUTH == 0 || (PLANAR_IN && !FINAL && !PARTIAL), "16-bit output: first layer only");
    constexpr int NTF = (COUT + 31) / 32;        // all output-channel tiles of the layer
    constexpr int NT = NTF / CSPLIT;             // tiles owned by this workgroup (blockIdx.y picks the slice)
    constexpr int COUTP = NTF * 32;
    const int nt0 = CSPLIT > 1 ? blockIdx.y * NT : 0;
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
        if (tile >= ntiles) tile = 0;              // duplicate work on a valid tile, never stored
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
            for (int nt = 0; nt < NT; ++nt) Bf[nt] = wp[((size_t)g * COUTP + (nt0 + nt) * 32) * 2];
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
                        acc[mt][nt] = OUTH ? __builtin_amdgcn_mfma_f32_32x32x2f32(
                                                 (&Bf[nt].x)[e], (&A[mt].x)[e], acc[mt][nt], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_32x32x2f32(
                                                 (&A[mt].x)[e], (&Bf[nt].x)[e], acc[mt][nt], 0, 0, 0);
        }
    } else {
        constexpr int G8 = CC / 8;
        constexpr int C4 = CC / 4;
        const float4 *wp = reinterpret_cast<const float4 *>(a.w) + (size_t)li * 2 + h;
        // split-K (small ensembles): blockIdx.y owns a contiguous range of input-channel chunks and
        // stores raw partial sums; k_conv_reduce adds them in a fixed order and applies the epilogue
        int cbeg = 0, cend = CIN;
        if constexpr (PARTIAL) {
            const int per = CIN / (int)gridDim.y;
            cbeg = blockIdx.y * per;
            cend = cbeg + per;
            wp += (size_t)(cbeg / CC) * T * G8 * COUTP * 2;
        }
        for (int c0 = cbeg; c0 < cend; c0 += CC) {
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
            // ---- K loop over (tap, 8-channel group), software pipelined: the A fragment (LDS) and the
            // B fragment (packed weights, L2) of step i+1 are requested before the MFMAs of step i.
            int ky = 0, kx = 0;
            int aoff[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int col = px[mt] - P;
                col = col < 0 ? col + N : col;
                aoff[mt] = (py[mt] * N + col) * STRIDE + 4 * h;
            }
            float4 An[MT], Bn[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Bn[nt] = wp[(size_t)(nt0 + nt) * 64];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) An[mt] = *reinterpret_cast<const float4 *>(&patch[aoff[mt]]);
            for (int tap = 0; tap < T; ++tap) {
                const bool last_tap = tap == T - 1;
                int nkx = kx + 1, nky = ky;
                if (nkx == KS) { nkx = 0; ++nky; }
                int aoff_n[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int col = px[mt] + nkx - P;
                    col = col < 0 ? col + N : (col >= N ? col - N : col);
                    aoff_n[mt] = last_tap ? aoff[mt] : ((py[mt] + nky) * N + col) * STRIDE + 4 * h;
                }
                const float4 *wp_n = last_tap ? wp : wp + (size_t)G8 * COUTP * 2;
#pragma unroll
                for (int g8 = 0; g8 < G8; ++g8) {
                    float4 A[MT], Bf[NT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) A[mt] = An[mt];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) Bf[nt] = Bn[nt];
                    if (g8 + 1 < G8) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) Bn[nt] = wp[((size_t)(g8 + 1) * COUTP + (nt0 + nt) * 32) * 2];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            An[mt] = *reinterpret_cast<const float4 *>(&patch[aoff[mt] + (g8 + 1) * 8]);
                    } else {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) Bn[nt] = wp_n[(size_t)(nt0 + nt) * 64];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            An[mt] = *reinterpret_cast<const float4 *>(&patch[aoff_n[mt]]);
                    }
                    // keep the prefetch ABOVE this step's MFMAs (hipcc otherwise sinks the loads to their use)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                    (&A[mt].x)[e], (&Bf[nt].x)[e], acc[mt][nt], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) aoff[mt] = aoff_n[mt];
                wp = wp_n;
                kx = nkx; ky = nky;
            }
            wp += (size_t)G8 * COUTP * 2;      // the last tap did not advance
        }
    }

    // ---- epilogue: bias (+ ReLU + BatchNorm affine), store
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int tile = wave + 4 * mt;
        if (tile >= ntiles) continue;
        if constexpr (OUTH != 0) {
            char *pix = reinterpret_cast<char *>(a.out) +
                        ((size_t)b * N * N + (size_t)y0 * N + tile * 32 + li) * (COUT * 2 * OUTH);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                store_tile_t<OUTH, false>(acc[mt][nt], (nt0 + nt) * 32, h, pix, a.bias, a.scale, a.shift, 1.0f, a.ascale, a.range, 1u);
            continue;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = (nt0 + nt) * 32 + li;
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
            } else if constexpr (PARTIAL) {
                float *o = a.out + ((size_t)blockIdx.y * a.npix_total + (size_t)b * N * N + (size_t)y0 * N) * COUT + co;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    o[(size_t)p * COUT] = acc[mt][nt][r];
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

// ---- hidden layers, LDS-only operands with register-prefetched staging ---------------------------
// Both MFMA operands come from LDS: the input patch (double buffered per 16-channel chunk) and the
// weight slice of TPS taps (double buffered).  The global loads that fetch the NEXT weight slice and
// the next chunk's patch are issued at the start of a stage into registers and written to the idle
// LDS buffers at its end, so the K loop contains no vector-memory instruction and nothing ever waits
// on HBM/L2 latency except the stage boundary.  The workgroup is persistent over its tiles, so the
// prefetch also runs across tile boundaries; one barrier per stage.
template <int CIN, int COUT, int KS, int CC, int MT, int TPS, int PPT, int NW = 4>
__global__ __launch_bounds__(NW * 64) void k_conv3(ConvArgs a, int total_tiles) {
    constexpr int NTHR = NW * 64;
    constexpr int NT = (COUT + 31) / 32;
    constexpr int COUTP = NT * 32;
    constexpr int P = KS / 2;
    constexpr int T = KS * KS;
    constexpr int G8 = CC / 8;
    constexpr int C4 = CC / 4;
    constexpr int NCH = CIN / CC;
    constexpr int STRIDE = CC + 4;
    constexpr int NSL = T / TPS;                        // weight slices per chunk
    constexpr int WSL = TPS * G8 * 2 * COUTP * 4;       // floats per slice, layout [tap][g8][h][cout][4]
    constexpr int WF4 = WSL / 4;
    constexpr int WPT = (WF4 + NTHR - 1) / NTHR;
    static_assert(T % TPS == 0, "taps per slice must divide the tap count");
    const int N = a.N, R = a.R;
    const int PR = R + KS - 1;
    const int patch_floats = PR * N * STRIDE;
    // LDS addresses are always formed arithmetically from the shared symbol: selecting between pointers
    // kept in an array makes hipcc lose the address space and emit flat_load (vmcnt+lgkmcnt, full drains)
    float *const lds0 = reinterpret_cast<float *>(conv_smem);
    float *const wlds0 = lds0 + 2 * patch_floats;
    const int tiles_per_img = N / R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int n_my = blockIdx.x < (unsigned)total_tiles ? (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int ntiles = R * N / 32;
    const int PF4 = PR * N * C4;                        // float4 per patch chunk
    const int PSH = (PF4 + NSL - 1) / NSL;              // patch float4 fetched per stage

    // The staging code below is written inline (no helper lambdas taking array references): hipcc
    // keeps the prefetch registers in VGPRs only when every index is a compile-time constant in the
    // kernel body; through a by-reference helper they went to scratch and each load was drained.
#define QGX_PATCH_LOAD(TI, CH, LO, HI, V)                                                                   \
    {                                                                                                       \
        const int tile_ = blockIdx.x + (TI) * gridDim.x;                                                    \
        const int b_ = tile_ / tiles_per_img;                                                               \
        const int y0_ = (tile_ - b_ * tiles_per_img) * R;                                                   \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            int it_ = (LO) + u * NTHR + threadIdx.x;                                                         \
            it_ = it_ < (HI) ? it_ : (HI) - 1; /* clamped: branch-free, the store is predicated instead */  \
            const int c4_ = it_ % C4, pl_ = it_ / C4;                                                       \
            const int pr_ = pl_ / N, x_ = pl_ - pr_ * N;                                                    \
            int gy_ = y0_ - P + pr_;                                                                        \
            gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                           \
            V[u] = *reinterpret_cast<const f32x4 *>(                                                        \
                &a.in[(((size_t)b_ * N + gy_) * N + x_) * CIN + (CH) * CC + c4_ * 4]);                      \
        }                                                                                                   \
    }
#define QGX_PATCH_STORE(BUF, LO, HI, V)                                                                     \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            const int it_ = (LO) + u * NTHR + threadIdx.x;                                                   \
            if (it_ < (HI)) {                                                                               \
                const int c4_ = it_ % C4, pl_ = it_ / C4;                                                   \
                *reinterpret_cast<f32x4 *>(&(BUF)[pl_ * STRIDE + c4_ * 4]) = V[u];                          \
            }                                                                                               \
        }                                                                                                   \
    }
#define QGX_W_LOAD(CH, SL, V)                                                                               \
    {                                                                                                       \
        const f32x4 *src_ = reinterpret_cast<const f32x4 *>(a.w) + ((size_t)(CH) * NSL + (SL)) * WF4;       \
        _Pragma("unroll") for (int u = 0; u < WPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            V[u] = src_[it_ < WF4 ? it_ : WF4 - 1];                                                         \
        }                                                                                                   \
    }
#define QGX_W_STORE(BUF, V)                                                                                 \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < WPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            if (it_ < WF4) reinterpret_cast<f32x4 *>(BUF)[it_] = V[u];                                      \
        }                                                                                                   \
    }

    if (n_my == 0) return;
    // ---- prologue: first chunk's patch and first weight slice, synchronously
    {
        f32x4 pv[PPT];
        for (int lo = 0; lo < PF4; lo += PPT * NTHR) {
            const int hi = lo + PPT * NTHR < PF4 ? lo + PPT * NTHR : PF4;
            QGX_PATCH_LOAD(0, 0, lo, hi, pv)
            QGX_PATCH_STORE(lds0, lo, hi, pv)
        }
        f32x4 wv[WPT];
        QGX_W_LOAD(0, 0, wv)
        QGX_W_STORE(wlds0, wv)
    }
    __syncthreads();

    int py[MT], px[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int tile = wave + NW * mt;
        if (tile >= ntiles) tile = wave % ntiles;
        const int p = tile * 32 + li;
        py[mt] = p / N;
        px[mt] = p - py[mt] * N;
    }
    f32x16 acc[MT][NT];
    int cur_p = 0, cur_w = 0;
    for (int ti = 0; ti < n_my; ++ti) {
        for (int ch = 0; ch < NCH; ++ch) {
            if (ch == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
            }
            // the chunk after this one (possibly the first chunk of the next tile)
            const int nch = ch + 1 < NCH ? ch + 1 : 0;
            const int nti = ch + 1 < NCH ? ti : ti + 1;
            const bool have_next_chunk = nti < n_my;
            for (int sl = 0; sl < NSL; ++sl) {
                // ---- prefetch into registers (consumed after the K loop of this stage)
                f32x4 wv[WPT], pv[PPT];
                const bool last_stage = !have_next_chunk && sl == NSL - 1;
                // (always issued, from a valid dummy source on the very last stage: no branch around loads)
                {
                    const int wch = sl + 1 < NSL ? ch : (have_next_chunk ? nch : ch);
                    const int wsl = sl + 1 < NSL ? sl + 1 : (have_next_chunk ? 0 : sl);
                    QGX_W_LOAD(wch, wsl, wv)
                }
                const int plo = sl * PSH, phi = (sl + 1) * PSH < PF4 ? (sl + 1) * PSH : PF4;
                QGX_PATCH_LOAD(have_next_chunk ? nti : ti, have_next_chunk ? nch : ch, plo, phi, pv)

                // ---- K loop over the TPS taps of this slice; operands from LDS, pipelined one step ahead
                const float *patch = lds0 + cur_p * patch_floats;
                const float *wl = wlds0 + cur_w * WSL + h * COUTP * 4 + li * 4;
                const int tap0 = sl * TPS;
                int ky = tap0 / KS, kx = tap0 - ky * KS;
                int aoff[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int col = px[mt] + kx - P;
                    col = col < 0 ? col + N : (col >= N ? col - N : col);
                    aoff[mt] = ((py[mt] + ky) * N + col) * STRIDE + 4 * h;
                }
                float4 An[MT], Bn[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) Bn[nt] = *reinterpret_cast<const float4 *>(wl + nt * 128);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) An[mt] = *reinterpret_cast<const float4 *>(&patch[aoff[mt]]);
                for (int tl = 0; tl < TPS; ++tl) {
                    const bool last_tap = tl == TPS - 1;
                    int nkx = kx + 1, nky = ky;
                    if (nkx == KS) { nkx = 0; ++nky; }
                    int aoff_n[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        int col = px[mt] + nkx - P;
                        col = col < 0 ? col + N : (col >= N ? col - N : col);
                        aoff_n[mt] = last_tap ? aoff[mt] : ((py[mt] + nky) * N + col) * STRIDE + 4 * h;
                    }
                    const float *wl_t = wl + (size_t)tl * G8 * 2 * COUTP * 4;
                    const float *wl_n = last_tap ? wl_t : wl_t + (size_t)G8 * 2 * COUTP * 4;
#pragma unroll
                    for (int g8 = 0; g8 < G8; ++g8) {
                        float4 A[MT], Bf[NT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) A[mt] = An[mt];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) Bf[nt] = Bn[nt];
                        if (g8 + 1 < G8) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                Bn[nt] = *reinterpret_cast<const float4 *>(wl_t + (size_t)(g8 + 1) * 2 * COUTP * 4 + nt * 128);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                An[mt] = *reinterpret_cast<const float4 *>(&patch[aoff[mt] + (g8 + 1) * 8]);
                        } else {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) Bn[nt] = *reinterpret_cast<const float4 *>(wl_n + nt * 128);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                An[mt] = *reinterpret_cast<const float4 *>(&patch[aoff_n[mt]]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                        (&A[mt].x)[e], (&Bf[nt].x)[e], acc[mt][nt], 0, 0, 0);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) aoff[mt] = aoff_n[mt];
                    kx = nkx; ky = nky;
                }

                // ---- retire the prefetch into the idle LDS buffers
                if (!last_stage) QGX_W_STORE(wlds0 + (cur_w ^ 1) * WSL, wv)
                if (have_next_chunk) QGX_PATCH_STORE(lds0 + (cur_p ^ 1) * patch_floats, plo, phi, pv)

                if (sl == NSL - 1 && ch == NCH - 1) {
                    // ---- epilogue of this tile: bias + ReLU + BatchNorm affine, NHWC store
                    const int tile_g = blockIdx.x + ti * gridDim.x;
                    const int b = tile_g / tiles_per_img;
                    const int y0 = (tile_g - b * tiles_per_img) * R;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int tile = wave + NW * mt;
                        if (tile >= ntiles) continue;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const int co = nt * 32 + li;
                            const float bias = a.bias[co], sc = a.scale[co], sh = a.shift[co];
                            float *o = a.out + ((size_t)b * N * N + (size_t)y0 * N) * COUT + co;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int p = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                                float vv = fmaxf(acc[mt][nt][r] + bias, 0.f);
                                o[(size_t)p * COUT] = vv * sc + sh;
                            }
                        }
                    }
                }
                __syncthreads();
                cur_w ^= 1;
            }
            cur_p ^= 1;
        }
    }
}
#undef QGX_PATCH_LOAD
#undef QGX_PATCH_STORE
#undef QGX_W_LOAD
#undef QGX_W_STORE

// out[p][c] = BN(ReLU(sum_s partial[s][p][c] + bias[c])): deterministic split-K combine (fixed order)
template <int COUT>
__global__ void k_conv_reduce(const float *partial, int nsplit, size_t npix_total, const float *bias,
                              const float *scale, const float *shift, float *out) {
    const size_t n4 = npix_total * COUT / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)((i * 4) % COUT);
        float4 v = reinterpret_cast<const float4 *>(partial)[i];
        for (int s2 = 1; s2 < nsplit; ++s2) {
            const float4 w = reinterpret_cast<const float4 *>(partial)[(size_t)s2 * n4 + i];
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        float4 o;
        o.x = fmaxf(v.x + bias[c + 0], 0.f) * scale[c + 0] + shift[c + 0];
        o.y = fmaxf(v.y + bias[c + 1], 0.f) * scale[c + 1] + shift[c + 1];
        o.z = fmaxf(v.z + bias[c + 2], 0.f) * scale[c + 2] + shift[c + 2];
        o.w = fmaxf(v.w + bias[c + 3], 0.f) * scale[c + 3] + shift[c + 3];
        reinterpret_cast<float4 *>(out)[i] = o;
    }
}

// ---- last layer (32 -> n_out <= 2, 3x3): VALU kernel ---------------------------------------------
// Two output channels would fill 2 of 32 MFMA columns; on the vector ALUs the 288x2 dot products per
// pixel run at full useful rate: one thread per pixel, weights broadcast from scalar registers.
struct LastWeights { float w[3 * 3 * 32 * 2]; };   // [tap][c][2], passed BY VALUE: kernarg -> scalar loads

// PARTS = 4 (single-row tiles of 64 pixels, i.e. one or two members at 64 x 64): the four waves split the input
// channels of the same 64 pixels and combine through LDS — a 4x shorter dependent FMA chain per thread
template <int CIN, int KS, int PARTS = 1>
__global__ __launch_bounds__(256) void k_conv_last(ConvArgs a, LastWeights lw) {
    constexpr int P = KS / 2, STRIDE = CIN + 4, C4 = CIN / 4;
    float *patch = reinterpret_cast<float *>(conv_smem);
    const int N = a.N, R = a.R;
    const int tiles_per_img = N / R;
    const int b = blockIdx.x / tiles_per_img;
    const int y0 = (blockIdx.x - b * tiles_per_img) * R;
    const int PR = R + KS - 1;
    for (int it = threadIdx.x; it < PR * N * C4; it += 256) {
        const int c4 = it % C4, pl = it / C4;
        const int pr = pl / N, x = pl - pr * N;
        int gy = y0 - P + pr;
        gy = gy < 0 ? gy + N : (gy >= N ? gy - N : gy);
        *reinterpret_cast<float4 *>(&patch[pl * STRIDE + c4 * 4]) =
            *reinterpret_cast<const float4 *>(&a.in[(((size_t)b * N + gy) * N + x) * CIN + c4 * 4]);
    }
    __syncthreads();
    const float *w = lw.w;
    const int part = PARTS > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;   // wave-uniform
    constexpr int CPP = C4 / PARTS;                    // float4 channel groups per part
    float *red = patch + PR * N * STRIDE;              // PARTS > 1: [part][pixel][2]
    for (int p = PARTS > 1 ? (threadIdx.x & 63) : threadIdx.x; p < R * N; p += PARTS > 1 ? 64 : 256) {
        const int py = p / N, px = p - py * N;
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 1
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                int col = px + kx - P;
                col = col < 0 ? col + N : (col >= N ? col - N : col);
                const float *src = &patch[((py + ky) * N + col) * STRIDE + part * CPP * 4];
                const float *wt = w + ((ky * KS + kx) * CIN + part * CPP * 4) * 2;
#pragma unroll
                for (int c4 = 0; c4 < CPP; ++c4) {
                    const float4 v = *reinterpret_cast<const float4 *>(src + c4 * 4);
                    acc0 = fmaf(v.x, wt[(c4 * 4 + 0) * 2], acc0); acc1 = fmaf(v.x, wt[(c4 * 4 + 0) * 2 + 1], acc1);
                    acc0 = fmaf(v.y, wt[(c4 * 4 + 1) * 2], acc0); acc1 = fmaf(v.y, wt[(c4 * 4 + 1) * 2 + 1], acc1);
                    acc0 = fmaf(v.z, wt[(c4 * 4 + 2) * 2], acc0); acc1 = fmaf(v.z, wt[(c4 * 4 + 2) * 2 + 1], acc1);
                    acc0 = fmaf(v.w, wt[(c4 * 4 + 3) * 2], acc0); acc1 = fmaf(v.w, wt[(c4 * 4 + 3) * 2 + 1], acc1);
                }
            }
        if constexpr (PARTS > 1) {
            red[(part * 64 + p) * 2] = acc0;
            red[(part * 64 + p) * 2 + 1] = acc1;
        } else {
            float *o = a.out + (size_t)b * a.cout_real * N * N + (size_t)y0 * N + p;
            o[0] = acc0 + a.bias[0];
            if (a.cout_real > 1) o[(size_t)N * N] = acc1 + a.bias[1];
        }
    }
    if constexpr (PARTS > 1) {
        __syncthreads();
        const int p = threadIdx.x >> 1, c = threadIdx.x & 1;
        if (p < R * N && c < a.cout_real) {
            float v = a.bias[c];
#pragma unroll
            for (int q = 0; q < PARTS; ++q) v += red[(q * 64 + p) * 2 + c];   // fixed order
            a.out[((size_t)b * a.cout_real + c) * N * N + (size_t)y0 * N + p] = v;
        }
    }
}

// ---- small pointwise kernels around the CNN ---------------------------------------------
// X = [float(q)/x_std, z]  (cgan_regression.py:158 + generate :133-137)
// largest |x| of the network input, for the f16x3 range guard: a NaN counts as infinity; non-negative floats
// order like their bit patterns, so one atomicMax on the bits per wave keeps the running maximum
__device__ __forceinline__ void input_absmax(float m, unsigned *range) {
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o));
    // the running maximum settles after the first launches: read it, and only a new record costs an atomic
    if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __builtin_nontemporal_load(range + 1)) atomicMax(range + 1, __float_as_uint(m));
}
__device__ __forceinline__ float abs_or_inf(float x) { return x != x ? __uint_as_float(0x7f800000u) : fabsf(x); }

__global__ void k_prep_input(const double *q, const float *z, float *X, int n_in, int npix, float xs0, float xs1,
                             unsigned *range) {
    const int b = blockIdx.y;
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const size_t qo = (size_t)b * 2 * npix + i;
        float *x = X + (size_t)b * n_in * npix + i;
        const float x0 = (float)q[qo] / xs0, x1 = (float)q[qo + npix] / xs1;
        x[0] = x0;
        x[npix] = x1;
        m = fmaxf(m, fmaxf(abs_or_inf(x0), abs_or_inf(x1)));
        if (n_in == 4) {
            const float z0 = z[qo], z1 = z[qo + npix];
            x[2 * (size_t)npix] = z0;
            x[3 * (size_t)npix] = z1;
            m = fmaxf(m, fmaxf(abs_or_inf(z0), abs_or_inf(z1)));
        }
    }
    input_absmax(m, range);
}

// the normalised PV of every member (channels 0, 1 of the generator's 4-channel input) as the 2-channel input of the
// regression net (cgan_regression.py:159-161: apply_function(self.net_mean, X) on the same X)
__global__ void k_take2(const float *X, float *X2, int npix2) {      // npix2 = 2 npix, a multiple of 4
    const int b = blockIdx.y;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(X + (size_t)b * 2 * npix2);
    f32x4 *dst = reinterpret_cast<f32x4 *>(X2 + (size_t)b * npix2);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix2 / 4; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ void k_absmax(const float *x, size_t n, unsigned *range) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fmaxf(m, abs_or_inf(x[i]));
    input_absmax(m, range);
}

// out[0] = max |a - b|, out[1] = max |b| (float bits; calibration of the Winograd layer against the exact-f32 path)
__global__ void k_absdiff_max(const float *a, const float *b, size_t n, unsigned *out) {
    float d = 0.f, m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        d = fmaxf(d, abs_or_inf(a[i] - b[i]));
        m = fmaxf(m, abs_or_inf(b[i]));
    }
    for (int o = 32; o > 0; o >>= 1) { d = fmaxf(d, __shfl_down(d, o)); m = fmaxf(m, __shfl_down(m, o)); }
    if ((threadIdx.x & 63) == 0) { atomicMax(out, __float_as_uint(d)); atomicMax(out + 1, __float_as_uint(m)); }
}

// Fused sampler update + input assembly of one online step (GAN / VAE):
//   z <- a z + b xi  (float; xi from Philox or the external draw), X = [float(q)/x_std, z]
// One thread per quad of 4 consecutive elements of the (2,N,N) member field.
__global__ void k_prep_noise(const double *q, float *z, const float *xi_ext, float *X, int npix, float xs0,
                             float xs1, uint64_t seed, uint64_t member_offset, uint64_t step, float a, float b,
                             unsigned *range) {
    const int member = blockIdx.y;
    const int quads = 2 * npix / 4;
    const int quad = blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.f;
    if (quad < quads) {
        const size_t o = (size_t)member * 2 * npix + 4 * (size_t)quad;
        float x[4];
        if (xi_ext) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = xi_ext[o + e];
        } else {
            philox_normal4(seed, member_offset + member, step, (uint32_t)quad, x);
        }
        float *Xm = X + (size_t)member * 4 * npix;
        const int i = 4 * quad;                          // flat index in (2, npix); npix % 4 == 0
        const float xs = i < npix ? xs0 : xs1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float zn = a == 0.f ? b * x[e] : a * z[o + e] + b * x[e];
            z[o + e] = zn;
            Xm[2 * (size_t)npix + i + e] = zn;
            const float xq = (float)q[o + e] / xs;
            Xm[i + e] = xq;
            m = fmaxf(m, fmaxf(abs_or_inf(zn), abs_or_inf(xq)));
        }
    }
    input_absmax(m, range);
}

// Fused output scaling + per-layer de-mean: one workgroup per (member, layer).
//   GAN/VAE: S = double(y * y_std)                         (cgan_regression.py:162)
//   GZ:      S = (mean + z sqrt(softplus(var))) * y_std    (mean_var_model.py:14-17,105-109)
//   GAN/VAE with regression != 'None': S = double((y + net_mean(x)) * y_std), the sum in float32
//                                                          (cgan_regression.py:159-162, cvae_regression.py:133-136)
//   then S -= mean_{y,x} S                                 (parameterization.py:25)
enum { FIN_PLAIN = 0, FIN_GZ = 1, FIN_SUM = 2 };
template <int MODE>
__global__ __launch_bounds__(1024) void k_finish(const float *y0, const float *y1, const double *z, double *S, int npix, float ys0,
                                                 float ys1, int demean, unsigned *range) {
    __shared__ double sm[16];
    __shared__ double mean_s;
    const size_t o = (size_t)blockIdx.x * npix;
    const float ys = (blockIdx.x & 1) ? ys1 : ys0;
    auto value = [&](int i) -> double {
        if constexpr (MODE == FIN_GZ) {
            const float vr = y1[o + i];
            const float sp = vr > 20.f ? vr : log1pf(expf(vr));
            return ((double)y0[o + i] + z[o + i] * (double)sqrtf(sp)) * (double)ys;
        } else if constexpr (MODE == FIN_SUM) {
            return (double)((y0[o + i] + y1[o + i]) * ys);
        } else {
            return (double)(y0[o + i] * ys);
        }
    };
    // one workgroup per (member, layer) is a short latency chain: 1024 threads, and the values are read ONCE
    // (kept in registers between the mean and the store for grids up to 128 x 128)
    constexpr int KEEP = 16;
    double keep[KEEP];
    const bool cached = npix <= KEEP * (int)blockDim.x;
    double acc = 0.0;
    if (cached) {
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            const int i = u * blockDim.x + threadIdx.x;
            keep[u] = i < npix ? value(i) : 0.0;
            acc += keep[u];
        }
    } else if (demean) {
        for (int i = threadIdx.x; i < npix; i += blockDim.x) acc += value(i);
    }
    double mu = 0.0;
    if (demean) {
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
            mean_s = t / (double)npix;
        }
        __syncthreads();
        mu = mean_s;
    }
    if (cached) {
        bool bad = false;
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            const int i = u * blockDim.x + threadIdx.x;
            if (i < npix) S[o + i] = keep[u] - mu;
            bad |= !(fabs(keep[u]) <= 1.79e308);
        }
        if (bad) atomicOr(range, 0x80000000u);      // a non-finite forcing never reaches the model unnoticed
    } else {
        bool bad = false;
        for (int i = threadIdx.x; i < npix; i += blockDim.x) {
            const double val = value(i);
            S[o + i] = val - mu;
            bad |= !(fabs(val) <= 1.79e308);
        }
        if (bad) atomicOr(range, 0x80000000u);
    }
}

// running first and second moments over Monte-Carlo samples (generate_mean_var, cgan_regression.py:139-146)
__global__ void k_moments(const float *y, double *sum, double *sumsq, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double v = (double)y[i];
        sum[i] += v;
        sumsq[i] += v * v;
    }
}

// ---- host side ---------------------------------------------------------------------------
struct LayerHost {
    int cin, cout, ks, coutp, cc, ngroups;
    LastWeights wv_host;   // last layer, VALU kernel layout (kernel argument)
    float *wl16 = nullptr, *wl8 = nullptr;   // k_conv3 layout [chunk][tap][g8][h][coutp][4], 16- / 8-channel chunks
    float *w = nullptr, *w32 = nullptr, *bias = nullptr, *scale = nullptr, *shift = nullptr;   // w: 16-ch chunks (or planar), w32: 32-ch chunks
    void *wh[2] = {nullptr, nullptr};        // conv_half.hpp layouts: [0] f16 (NS = 1), [1] f16 hi/lo (NS = 2)
    float wh_unscale[2] = {1.f, 1.f};        // 2^-s of the power-of-two weight pre-scale
    // layer 2 with layer 1's BatchNorm folded in (W' = W alpha[c_in], b' = b + sum W beta'[c_in]; exact under circular
    // padding): layer 1 then stores ReLU output, half of which is exactly zero -> a sparser MFMA operand
    void *whF = nullptr, *wh16F = nullptr; float whF_unscale = 1.f; float *biasF = nullptr;
    // layer 2 as a 1-D Winograd convolution F(4, 5) along x (conv_wino.hpp): transformed weights, [0] plain, [1] with layer
    // 1's BatchNorm folded in; per-position 2^-s of the power-of-two pre-scale
    void *ww[2] = {nullptr, nullptr};
    float ww_unscale[2][8] = {{1, 1, 1, 1, 1, 1, 1, 1}, {1, 1, 1, 1, 1, 1, 1, 1}};
    float *ones = nullptr, *zeros = nullptr; // layer 1: identity BatchNorm for the folded variant
    void *wh16 = nullptr;                    // k_convh3 (16x16x32 MFMA): [chunk32][tap][part][octet][cout][8] f16
    void *whf = nullptr;                     // first layer, f16x3: [step][part][h][128][8] f16
    float whf_unscale = 1.f;
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
    float *part = nullptr;         // split-K partial sums of the small-ensemble path
    size_t part_elems = 0;
    // second workspace: the other half of an ensemble stepped in halves on two streams (model.hip::qgx_step); the members
    // above are the ACTIVE set, generator_select_workspace swaps the two
    struct Workspace { size_t cap_elems = 0, part_elems = 0; float *actA = nullptr, *actB = nullptr, *X = nullptr, *Y0 = nullptr, *Y1 = nullptr, *part = nullptr; } ws_other;
    int ws_active = 0;
    // optional per-layer timing with HIP events on the launch stream (bench.py roofline leg)
    // kernel variant selection (qgx_generator_set_option; defaults = fastest measured)
    int opt_cc = 32, opt_last_valu = 1, opt_first_split = 2, opt_v3 = -1, opt_small = 1;
    unsigned long long *stamps = nullptr;   // diagnostic builds only
    int stamp_layer = -1;
    int opt_h2_tw32 = 0;           // 5x5 layer at 64 x 64: 16-row x 32-column tiles instead of 8 full rows
    int opt_h2_x96 = 1;            // 3x3 layers at 96 x 96 as 8-wave workgroups on 16-row x 32-column tiles (-1.4 % of the step at 32 members, -3.6 % at 64)
    int opt_h2_w8_min96 = 1024;    // 5x5 layer at 96 x 96: minimum tile count for the 8-wave x-tiled kernel
    int opt_h2_w8 = 3;             // k_convh2 as one 8-wave workgroup per CU: bit 0 the 5x5 layer, bit 1 the 3x3 layers (64 x 64)
    int opt_prio_alt = 1;          // k_convh2 with two workgroups per CU: alternate their wave priority per tile
    int opt_h4 = 0;                // 5x5 layer: k_convh4 (full-line patch chunks, 8 waves, R = 8)
    int opt_h2_grid = 0;           // k_convh2: persistent workgroups per launch (0 = one or two per CU by LDS size)
    int opt_wino = 2;              // f16x3: the 5x5 layer as a 1-D Winograd convolution F(4, 5) along x (k_convw): 0 never, 1 on every
                                   //   specialised grid, 2 = per grid size, where calibrate_wino() admitted it
    int auto_wino_n[5] = {0, 0, 0, 0, 0};            //   ... what calibrate_wino() decided for N = 32, 48, 64, 96, 128 and the errors it measured
    float wino_err_n[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int opt_stop_layer = 0;        //   A/B library, debugging: return after this many layers (the activation buffers keep their outputs)
    int opt_wino2 = 1;             //   ... as k_convw2 (conv_wino2.hpp: transform under the MFMAs, bit-identical) where that kernel exists; 0 = k_convw
    int opt_wino_pl = 0;           //   A/B library: 1 = channel-planar layer-1 output and the MFMA input transform (measured: see wino_planar)
    int opt_wino_exp = 0;          //   A/B library: timing experiments (conv_wino.hpp EXP)
    int opt_h2_rows96 = 0;         // 3x3 layers at 96 x 96: tile rows, 0 = by tile-count quantisation, 12 (6 waves), 16 (8 waves)
    int opt_wino_rows64 = 0;       //   ... its tile shape at 64 x 64, 128 x 128 and 32 x 32: 0 = by tile-count quantisation, 4 = the half-height
                                   //   shape (4 x 64 tiles; 8 x 32 at 32 x 32), 8 = the full one
    int opt_wino_rows96 = 0;       //   ... its tile rows at 96 x 96: 0 = by tile-count quantisation (launch_convw), 12, 16
    int opt_wino_min_tiles = 48;   //   ... from this many full-height tiles on (measured crossovers, bench_tools/ab_conv.py: with the half-height
                                   //   shapes the Winograd kernel is ahead of the 25-tap kernels from 6 members at 64 x 64, 4 at 96 x 96, 24 at 32 x 32)
    int opt_fold = 1;              // f16x3: layer 1 stores ReLU output, its BatchNorm is folded into layer 2's weights
    int opt_part_max_tiles = 96;   // f16x3: split K on the wide layers below this many quarter-height tiles (crossover against the Winograd
                                   //   layer's half-height shape: 6 members at 64 x 64 — forward 166.8 -> 158.0 us —, 4 at 96 x 96: 217.5 -> 184.3)
    int opt_last_rows = 0;         // VALU last layer: rows per workgroup (0 = automatic)
    int opt_h3 = 0;                // 5x5 layer on 16x16x32 MFMAs (k_convh3): measured no faster in the full kernel
    int opt_half_min_tiles = 1;
    int opt_tiny_pairs = 7;        // tiny ensembles at 64 x 64 (split-K path): bit 0 layers (7, 8), bit 1 layers (5, 6), bit 2 layers (3, 4) as ONE fused launch on 2-row strips
    int opt_pair_lp = 1;           // A/B library only: 0 = the pair kernels fetch the two halves of a line in different chunk iterations
    int opt_small_tiles = 1;       // 64 x 64, at most 4 members: half-height tiles (small_tiles())
    int opt_fuse96 = 2;            // ... at 96 x 96 (4-row strips): bit 0 (5,6), 1 (7,8)
    int opt_fuse = 3;              // f16x3, 64x64: 3x3 layers fused pairwise (k_convh_pair): bit 0 (5,6), 1 (7,8), 2 (3,4)
    int opt_pair = 1;              // 3x3 k_convh2: fetch both 64-byte halves of a pixel's 128-byte line together
    int opt_h2 = 3;                // bit 1: k_convh2 for the 5x5 layer, bit 0: for the 3x3 layers (64x64 grids)
    int opt_res = 1;               // f16x3 3x3 layers: resident-weight kernel where its tile fits in LDS
    int opt_member_chunk = 0;      // 16-bit path: members per sub-batch (0 = whole ensemble)
    int opt_half_nw = 8;           // 16-bit hidden layers: 4 waves x 2 workgroups per CU, or 8 x 1
    int opt_first_h = 1;           // f16x3 path: first layer on the 16-bit cores too (0: exact-f32 MFMA first layer)
    int opt_precision = 3;         // 0 = exact f32 MFMA, 1 = f16 MFMA, 3 = f16x3 split (f32-class accuracy; default
                                   // wherever the ensemble fills the 8-wave tiles, see half_path_ok)
    float opt_ascale = 1.f;        // power-of-two pre-scale of stored 16-bit activations (chosen by calibrate())
    // f16x3 range guard (conv_half.hpp::range_guard): [0] sticky flags — bit l: layer l stored a value beyond the f16
    // range, bit 31: non-finite forcing; [1] bits of the largest |network input| seen
    unsigned *range_dev = nullptr;
    unsigned *calib_dev = nullptr; // calibration only: per-layer max |activation| of the exact-f32 evaluation
    float calib_max[10] = {0};     // [0..6] stored (post-BatchNorm) activations, [8] layer 1 before its BatchNorm
    int auto_precision = 3, auto_fold = 1, auto_ascale_log2 = 0;   // what calibrate() decided
    int prof_layer = -1;
    int prof_every = 1;                 // bracket every n-th launch of the profiled layer ("prof_every" option)
    long prof_seen = 0;
    std::vector<hipEvent_t> prof_ev;    // pairs (start, stop)
    size_t prof_used = 0;
};

namespace qgx {

// hipFuncSetAttribute takes microseconds of host time; the single-member step is a chain of 5-15 us kernels whose
// launches the host has to keep ahead of: the dynamic-LDS cap of a kernel is raised once per kernel, device and host
// thread, not once per launch (rocprofv3: 14 launches / 116 us of kernels per step were taking 128 us of wall time)
static int ensure_dynamic_lds(const void *kern, int bytes) {
    static thread_local std::unordered_map<const void *, int> seen[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto &m = seen[dev & 15];
    const auto it = m.find(kern);
    if (it != m.end() && it->second >= bytes) return QGX_OK;
    QGX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    m[kern] = bytes;
    return QGX_OK;
}

static const int KSZ[8] = {5, 5, 3, 3, 3, 3, 3, 3};
static const int HID[7] = {128, 64, 32, 32, 32, 32, 32};

static int upf(float *&dst, const std::vector<float> &h) {
    QGX_HIP(hipMalloc((void **)&dst, h.size() * sizeof(float)));
    QGX_HIP(hipMemcpy(dst, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return QGX_OK;
}

// packed weights [k-group][coutp][8] for channel-chunk size cc (planar first layer: cc = cin)
static int pack_weights(const LayerHost &L, int li, const qgx_cnn_weights *w, bool planar_in, int cc, float *&dst) {
    const int cin = L.cin, cout = L.cout, ks = L.ks, T = ks * ks;
    const int ngroups = planar_in ? (T * cin + 7) / 8 : (cin / cc) * T * (cc / 8);
    std::vector<float> pw((size_t)ngroups * L.coutp * 8, 0.f);
    const float *W = w->conv_w[li];
    for (int g = 0; g < ngroups; ++g)
        for (int co = 0; co < cout; ++co)
            for (int e = 0; e < 8; ++e) {
                int tap, c;
                if (planar_in) {
                    const int kidx = 8 * g + e;
                    if (kidx >= T * cin) continue;
                    tap = kidx / cin; c = kidx % cin;
                } else {
                    const int g8n = cc / 8;
                    const int g8 = g % g8n, tt = (g / g8n) % T, chunk = g / (g8n * T);
                    tap = tt; c = chunk * cc + g8 * 8 + e;
                }
                const int ky = tap / ks, kx = tap % ks;
                pw[((size_t)g * L.coutp + co) * 8 + e] = W[(((size_t)co * cin + c) * ks + ky) * ks + kx];
            }
    return upf(dst, pw);
}

// conv_half.hpp weight layout [chunk][tap][j][h][cout][8] f16, pre-scaled by 2^s with max|w| 2^s in [2^13, 2^14)
static int pack_half(LayerHost &L, int li, const qgx_cnn_weights *w, int NS, const float *cin_scale = nullptr) {
    const int cin = L.cin, cout = L.cout, T = L.ks * L.ks;
    const int coutp = ((cout + 31) / 32) * 32;         // the 32 -> 2 last layer is padded with zero rows
    const int CC = NS == 1 ? 32 : 16, nch = cin / CC;
    const float *W = w->conv_w[li];
    float mx = 0.f;
    for (int co = 0; co < cout; ++co)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < T; ++t)
                mx = fmaxf(mx, fabsf(W[((size_t)co * cin + c) * T + t] * (cin_scale ? cin_scale[c] : 1.f)));
    int e = 0;
    if (mx > 0.f) { (void)frexpf(mx, &e); }             // mx = m 2^e, m in [0.5, 1)
    int sexp = 14 - e;
    sexp = sexp < -20 ? -20 : (sexp > 40 ? 40 : sexp);
    const float sc = ldexpf(1.f, sexp);
    std::vector<_Float16> pw((size_t)nch * T * 4 * coutp * 8, (_Float16)0.f);
    for (int ch = 0; ch < nch; ++ch)
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < 2; ++j)
                for (int hh = 0; hh < 2; ++hh)
                    for (int co = 0; co < cout; ++co)
                        for (int e8 = 0; e8 < 8; ++e8) {
                            const int c = NS == 1 ? ch * 32 + j * 16 + hh * 8 + e8 : ch * 16 + hh * 8 + e8;
                            const float x = W[((size_t)co * cin + c) * T + t] * (cin_scale ? cin_scale[c] : 1.f) * sc;
                            const _Float16 xh = (_Float16)x;
                            const _Float16 v = (NS == 1 || j == 0) ? xh : (_Float16)(x - (float)xh);
                            pw[(((((size_t)ch * T + t) * 2 + j) * 2 + hh) * coutp + co) * 8 + e8] = v;
                        }
    void *&dst = cin_scale ? L.whF : L.wh[NS - 1];
    QGX_HIP(hipMalloc(&dst, pw.size() * sizeof(_Float16)));
    QGX_HIP(hipMemcpy(dst, pw.data(), pw.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    (cin_scale ? L.whF_unscale : L.wh_unscale[NS - 1]) = ldexpf(1.f, -sexp);
    return QGX_OK;
}

// conv_wino.hpp weight layout [chunk][ky][p][part][h][cout][8] f16: U_p,ky(o, c) = sum_kx G[p][kx] w(o, c, ky, kx) of the
// Toom-Cook algorithm F(4, 5) with the points 0, +-1, +-2, +-1/2, infinity, evaluated in float64, pre-scaled per position by
// 2^s_p with max|U_p| 2^s_p in [2^13, 2^14), split into f16 hi / lo
static const double WINO_G[8][5] = {{-1, 0, 0, 0, 0},
                                    {-2. / 9, -2. / 9, -2. / 9, -2. / 9, -2. / 9},
                                    {-2. / 9, 2. / 9, -2. / 9, 2. / 9, -2. / 9},
                                    {1. / 90, 2. / 90, 4. / 90, 8. / 90, 16. / 90},
                                    {1. / 90, -2. / 90, 4. / 90, -8. / 90, 16. / 90},
                                    {32. / 45, 16. / 45, 8. / 45, 4. / 45, 2. / 45},
                                    {32. / 45, -16. / 45, 8. / 45, -4. / 45, 2. / 45},
                                    {0, 0, 0, 0, 1}};
[[maybe_unused]] static const float WINO_AT[4][8] = {{1, 1, 1, 1, 1, 1, 1, 0},
                                    {0, 1, -1, 2, -2, .5f, -.5f, 0},
                                    {0, 1, 1, 4, 4, .25f, .25f, 0},
                                    {0, 1, -1, 8, -8, .125f, -.125f, 1}};
static int pack_wino(LayerHost &L, int li, const qgx_cnn_weights *w, const float *cin_scale, int which) {
    const int cin = L.cin, cout = L.cout;
    if (cin != 128 || cout != 64 || L.ks != 5) return QGX_OK;
    const int nch = cin / 16;
    const float *W = w->conv_w[li];
    std::vector<double> U((size_t)8 * 5 * cout * cin);
    double mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = 0; p < 8; ++p)
        for (int ky = 0; ky < 5; ++ky)
            for (int co = 0; co < cout; ++co)
                for (int c = 0; c < cin; ++c) {
                    double u = 0.0;
                    for (int kx = 0; kx < 5; ++kx)
                        u += WINO_G[p][kx] * (double)W[((size_t)co * cin + c) * 25 + ky * 5 + kx];
                    u *= cin_scale ? (double)cin_scale[c] : 1.0;
                    U[(((size_t)p * 5 + ky) * cout + co) * cin + c] = u;
                    mx[p] = fmax(mx[p], fabs(u));
                }
    std::vector<_Float16> pw((size_t)nch * 5 * 8 * 4 * cout * 8, (_Float16)0.f);
    for (int p = 0; p < 8; ++p) {
        int e = 0;
        if (mx[p] > 0.) (void)frexp(mx[p], &e);
        int sexp = 14 - e;
        sexp = sexp < -20 ? -20 : (sexp > 40 ? 40 : sexp);
        const double sc = ldexp(1.0, sexp);
        L.ww_unscale[which][p] = ldexpf(1.f, -sexp);
        for (int ch = 0; ch < nch; ++ch)
            for (int ky = 0; ky < 5; ++ky)
                for (int j = 0; j < 2; ++j)
                    for (int hh = 0; hh < 2; ++hh)
                        for (int co = 0; co < cout; ++co)
                            for (int e8 = 0; e8 < 8; ++e8) {
                                const int c = ch * 16 + hh * 8 + e8;
                                const float x = (float)(U[(((size_t)p * 5 + ky) * cout + co) * cin + c] * sc);
                                const _Float16 xh = (_Float16)x;
                                pw[((((((size_t)ch * 5 + ky) * 8 + p) * 2 + j) * 2 + hh) * cout + co) * 8 + e8] =
                                    j == 0 ? xh : (_Float16)(x - (float)xh);
                            }
    }
    QGX_HIP(hipMalloc(&L.ww[which], pw.size() * sizeof(_Float16)));
    QGX_HIP(hipMemcpy(L.ww[which], pw.data(), pw.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    return QGX_OK;
}

// k_convh3 layout: 32-channel chunks, [chunk][tap][part][octet][cout][8]; same power-of-two pre-scale as wh[1]
static int pack_half16(LayerHost &L, int li, const qgx_cnn_weights *w, const float *cin_scale = nullptr) {
    const int cin = L.cin, cout = L.cout, T = L.ks * L.ks, nch = cin / 32;
    const float *W = w->conv_w[li];
    const float sc = 1.0f / (cin_scale ? L.whF_unscale : L.wh_unscale[1]);
    std::vector<_Float16> pw((size_t)nch * T * 2 * 4 * cout * 8, (_Float16)0.f);
    for (int ch = 0; ch < nch; ++ch)
        for (int t = 0; t < T; ++t)
            for (int part = 0; part < 2; ++part)
                for (int o = 0; o < 4; ++o)
                    for (int co = 0; co < cout; ++co)
                        for (int e8 = 0; e8 < 8; ++e8) {
                            const int c = ch * 32 + o * 8 + e8;
                            const float x = W[((size_t)co * cin + c) * T + t] * (cin_scale ? cin_scale[c] : 1.f) * sc;
                            const _Float16 xh = (_Float16)x;
                            pw[(((((size_t)ch * T + t) * 2 + part) * 4 + o) * cout + co) * 8 + e8] =
                                part == 0 ? xh : (_Float16)(x - (float)xh);
                        }
    void *&dst16 = cin_scale ? L.wh16F : L.wh16;
    QGX_HIP(hipMalloc(&dst16, pw.size() * sizeof(_Float16)));
    QGX_HIP(hipMemcpy(dst16, pw.data(), pw.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    return QGX_OK;
}

// first layer, f16x3 (k_convh_first): [step][part][h][cout][8], element j of lane half h in step s is
// (tap, channel) = (TPS s + TPF h + j / n_in, j % n_in); tap slots >= 25 hold zeros
static int pack_half_first(LayerHost &L, const qgx_cnn_weights *w) {
    const int nin = L.cin, cout = L.cout, T = 25;
    const int TPF = 8 / nin, TPS = 2 * TPF, nstep = (T + TPS - 1) / TPS;
    const float *W = w->conv_w[0];
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)cout * nin * T; ++i) mx = fmaxf(mx, fabsf(W[i]));
    int e = 0;
    if (mx > 0.f) (void)frexpf(mx, &e);
    int sexp = 14 - e;
    sexp = sexp < -20 ? -20 : (sexp > 40 ? 40 : sexp);
    const float sc = ldexpf(1.f, sexp);
    std::vector<_Float16> pw((size_t)nstep * 4 * cout * 8, (_Float16)0.f);
    for (int s = 0; s < nstep; ++s)
        for (int part = 0; part < 2; ++part)
            for (int hh = 0; hh < 2; ++hh)
                for (int co = 0; co < cout; ++co)
                    for (int j = 0; j < 8; ++j) {
                        const int tap = TPS * s + TPF * hh + j / nin, c = j % nin;
                        if (tap >= T) continue;
                        const float x = W[((size_t)co * nin + c) * T + tap] * sc;
                        const _Float16 xh = (_Float16)x;
                        pw[((((size_t)s * 2 + part) * 2 + hh) * cout + co) * 8 + j] = part == 0 ? xh : (_Float16)(x - (float)xh);
                    }
    QGX_HIP(hipMalloc(&L.whf, pw.size() * sizeof(_Float16)));
    QGX_HIP(hipMemcpy(L.whf, pw.data(), pw.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    L.whf_unscale = ldexpf(1.f, -sexp);
    return QGX_OK;
}

static int pack_layer(LayerHost &L, int li, const qgx_cnn_weights *w, bool planar_in) {
    const int cout = L.cout;
    L.coutp = ((cout + 31) / 32) * 32;
    int rc;
    if (planar_in) {
        if ((rc = pack_weights(L, li, w, true, L.cin, L.w))) return rc;
        if ((rc = pack_half_first(L, w))) return rc;
        {
            std::vector<float> o1(L.coutp, 1.f), z0(L.coutp, 0.f);
            if ((rc = upf(L.ones, o1)) || (rc = upf(L.zeros, z0))) return rc;
        }
    } else {
        if ((rc = pack_weights(L, li, w, false, 16, L.w))) return rc;
        if ((rc = pack_weights(L, li, w, false, 32, L.w32))) return rc;
        for (int cc : {16}) {      // LDS-operand layout for k_conv3
            const int cin = L.cin, ks = L.ks, T = ks * ks, g8n = cc / 8, nch = cin / cc;
            std::vector<float> pw((size_t)nch * T * g8n * 2 * L.coutp * 4, 0.f);
            for (int ch = 0; ch < nch; ++ch)
                for (int t = 0; t < T; ++t)
                    for (int g8 = 0; g8 < g8n; ++g8)
                        for (int hh = 0; hh < 2; ++hh)
                            for (int co = 0; co < cout; ++co)
                                for (int e = 0; e < 4; ++e) {
                                    const int c = ch * cc + g8 * 8 + hh * 4 + e;
                                    pw[(((((size_t)ch * T + t) * g8n + g8) * 2 + hh) * L.coutp + co) * 4 + e] =
                                        w->conv_w[li][((size_t)co * cin + c) * T + t];
                                }
            if ((rc = upf(cc == 16 ? L.wl16 : L.wl8, pw))) return rc;
        }
        if (li < 7) {
            if ((rc = pack_half(L, li, w, 2))) return rc;
#ifdef QGX_AB
            if ((rc = pack_half(L, li, w, 1))) return rc;
            if (li == 1 && (rc = pack_half16(L, li, w))) return rc;
#endif
            if (li == 1) {
                // fold layer 1's BatchNorm (alpha, beta' of its 128 output channels) into this layer
                const int cin = L.cin, T = L.ks * L.ks;
                std::vector<float> al(cin), be(cin), bf(L.coutp, 0.f);
                for (int c = 0; c < cin; ++c) {
                    const float invstd = 1.0f / sqrtf(w->bn_var[0][c] + w->bn_eps);
                    al[c] = w->bn_gamma[0][c] * invstd;
                    be[c] = w->bn_beta[0][c] - w->bn_mean[0][c] * al[c];
                }
                for (int co = 0; co < cout; ++co) {
                    double acc = w->conv_b[li][co];
                    for (int c = 0; c < cin; ++c)
                        for (int t = 0; t < T; ++t) acc += (double)w->conv_w[li][((size_t)co * cin + c) * T + t] * be[c];
                    bf[co] = (float)acc;
                }
                if ((rc = pack_half(L, li, w, 2, al.data())) || (rc = upf(L.biasF, bf))) return rc;
                if ((rc = pack_wino(L, li, w, nullptr, 0)) || (rc = pack_wino(L, li, w, al.data(), 1))) return rc;
#ifdef QGX_AB
                if ((rc = pack_half16(L, li, w, al.data()))) return rc;
#endif
            }
        } else if ((rc = pack_half(L, li, w, 2))) return rc;      // fused (layer 7, layer 8) pair
    }
    if (li == 7) {      // [tap][c][2] for the VALU last-layer kernel
        const int cin = L.cin, ks = L.ks;
        memset(&L.wv_host, 0, sizeof(L.wv_host));
        for (int co = 0; co < cout && co < 2; ++co)
            for (int c = 0; c < cin; ++c)
                for (int t = 0; t < ks * ks; ++t)
                    L.wv_host.w[((size_t)t * cin + c) * 2 + co] = w->conv_w[li][((size_t)co * cin + c) * ks * ks + t];
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
    if ((rc = upf(L.bias, bias)) || (rc = upf(L.scale, sc)) || (rc = upf(L.shift, sh))) return rc;
    return QGX_OK;
}

// Very small ensembles at 64 x 64 (a single member above all: BASELINE configs[1]) are chains of launch latencies; with 4-row
// tiles a member is 16 workgroups, with 2-row tiles 32 with half the K loop each (bit-identical: the tile shape does not enter
// the summation order).  Measured at one member: the 5x5 layer's split-K kernel + combine 23.6 -> 19.9 us, layer 3 12.1 -> 10.9.
static bool small_tiles(const qgx_generator *g, int B, int N) { return g->opt_small_tiles && N == 64 && B * 16 <= 64; }

static int choose_rows(int N) {
    if (N <= 256 && 256 % N == 0) return 256 / N;     // 8 M-tiles
    if (N <= 384 && 384 % N == 0) return 384 / N;     // 12 M-tiles
    return 0;
}

static int prof_begin(qgx_generator *g, int layer, hipStream_t st, hipEvent_t &stop) {
    stop = nullptr;
    if (g->prof_layer != layer) return QGX_OK;
    // an event pair costs ~6 us of idle GPU on each side of the kernel: bracket every prof_every-th launch only
    if (g->prof_every > 1 && (g->prof_seen++ % g->prof_every) != 0) return QGX_OK;
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

template <int CIN, int COUT, int KS, int CC, bool PLANAR_IN, bool FINAL, int CSPLIT = 1, int OUTH = 0>
static int launch_conv(qgx_generator *g, int layer, const LayerHost &L, const float *in, float *out, int B,
                       int N, int cout_real, hipStream_t st) {
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    const int R = choose_rows(N);
    QGX_REQUIRE(R > 0 && N % R == 0, "generator: unsupported grid size N=%d", N);
    const int ntiles = R * N / 32;
    ConvArgs a = {};
    a.in = in; a.out = out; a.w = (!PLANAR_IN && CC == 32) ? L.w32 : L.w; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.N = N; a.R = R; a.cout_real = cout_real; a.npix_total = (size_t)B * N * N; a.ascale = g->opt_ascale;
    a.range = g->range_dev;
    constexpr int STRIDE = PLANAR_IN ? CIN : CC + 4;
    const size_t lds = (size_t)(R + KS - 1) * N * STRIDE * sizeof(float);
    QGX_REQUIRE(lds <= 160 * 1024, "generator: LDS patch %zu B too large for N=%d", lds, N);
    dim3 grid(B * (N / R), CSPLIT), block(256);
    if (ntiles <= 8) {
        auto kern = k_conv<CIN, COUT, KS, CC, 2, PLANAR_IN, FINAL, CSPLIT, false, OUTH>;
        { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    } else {
        auto kern = k_conv<CIN, COUT, KS, CC, 3, PLANAR_IN, FINAL, CSPLIT, false, OUTH>;
        { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    }
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}

// ---- small ensembles (few tiles): 4 M-tiles per workgroup (one per wave) and split-K over the
// 32-channel chunks, so that a single member still spreads over >= 128 workgroups -----------------
// rows per workgroup of the small-ensemble path: at most 4 M-tiles of 32 pixels (one per wave), as many as fit
static int rows_small(int N) {
    int best = 0;
    for (int R = 1; R <= N; ++R) {
        if (N % R || (R * N) % 32) continue;
        if (R * N / 32 > 4) break;
        best = R;
    }
    return best;
}
static bool small_ensemble(int B, int N) {
    const int R = choose_rows(N);
    return R > 0 && B * (N / R) <= 192 && rows_small(N) > 0 && rows_small(N) < R;   // crossover ~ B=13 at 64x64
}

template <int CIN, int COUT, int KS, int CC = 32>
static int launch_conv_small(qgx_generator *g, int layer, const LayerHost &L, const float *in, float *out, int B,
                             int N, hipStream_t st) {
    const int R = rows_small(N);                   // <= 4 M-tiles of 32 pixels
    const int nsplit = CIN >= 64 ? CIN / CC : 1;
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    const size_t npix = (size_t)B * N * N;
    if (nsplit > 1 && g->part_elems < npix * COUT * nsplit) {
        if (g->part) (void)hipFree(g->part);
        g->part = nullptr; g->part_elems = 0;
        QGX_HIP(hipMalloc((void **)&g->part, npix * COUT * nsplit * sizeof(float)));
        g->part_elems = npix * COUT * nsplit;
    }
    ConvArgs a = {};
    a.in = in; a.out = nsplit > 1 ? g->part : out; a.w = CC == 32 ? L.w32 : L.w; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.N = N; a.R = R; a.cout_real = COUT; a.npix_total = npix;
    const size_t lds = (size_t)(R + KS - 1) * N * (CC + 4) * sizeof(float);
    dim3 grid(B * (N / R), nsplit), block(256);
    if (nsplit > 1) {
        auto kern = k_conv<CIN, COUT, KS, CC, 1, false, false, 1, true>;
        { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);
        const size_t n4 = npix * COUT / 4;
        hipLaunchKernelGGL(k_conv_reduce<COUT>, dim3((unsigned)((n4 + 255) / 256 > 1024 ? 1024 : (n4 + 255) / 256)), dim3(256),
                           0, st, (const float *)g->part, nsplit, npix, (const float *)L.bias, (const float *)L.scale,
                           (const float *)L.shift, out);
    } else {
        auto kern = k_conv<CIN, COUT, KS, CC, 1, false, false, 1, false>;
        { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    }
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}

// ---- k_conv3 launcher -------------------------------------------------------------------------
template <int CIN, int COUT, int KS, int TPS, int CC = 16, int NW = 4>
static int launch_conv3(qgx_generator *g, int layer, const LayerHost &L, const float *in, float *out, int B,
                        int N, hipStream_t st, bool &done) {
    done = false;
    constexpr int T = KS * KS, NSL = T / TPS, NTc = (COUT + 31) / 32;
    constexpr size_t WSLB = (size_t)TPS * (CC / 8) * 2 * NTc * 32 * 4 * sizeof(float);
    // rows per tile: as k_conv (8 or 12 M-tiles; twice that with 8 waves), LDS = 2 patches + 2 weight slices
    int R = choose_rows(N);
    if (R == 0 || N % R) return QGX_OK;
    if (NW == 8) { if (N % (2 * R)) return QGX_OK; R *= 2; }
    const int ntiles = R * N / 32;
    const int PR = R + KS - 1;
    const size_t lds = (size_t)2 * PR * N * (CC + 4) * sizeof(float) + 2 * WSLB;
    if (lds > 160 * 1024 - 256) return QGX_OK;
    const int PF4 = PR * N * (CC / 4);
    const int ppt = (((PF4 + NSL - 1) / NSL) + NW * 64 - 1) / (NW * 64);
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    ConvArgs a = {};
    a.in = in; a.out = out; a.w = CC == 16 ? L.wl16 : L.wl8; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.N = N; a.R = R; a.cout_real = COUT;
    const int total_tiles = B * (N / R);
    const int wgs = lds * 2 <= 160 * 1024 ? 2 : 1;
    int grid = 256 * wgs;
    if (grid > total_tiles) grid = total_tiles;
#define QGX_L3(MTV, PPTV)                                                                                     \
    {                                                                                                         \
        auto kern = k_conv3<CIN, COUT, KS, CC, MTV, TPS, PPTV, NW>;                                           \
        { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; } \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a, total_tiles);                         \
    }
    const int mtv = (ntiles + NW - 1) / NW <= 1 ? 1 : ((ntiles + NW - 1) / NW == 2 ? 2 : 3);
    if (mtv == 1) {
        if (ppt <= 2) QGX_L3(1, 2) else if (ppt <= 4) QGX_L3(1, 4) else return QGX_OK;
    } else if (ppt <= 2) { if (mtv == 2) QGX_L3(2, 2) else QGX_L3(3, 2) }
    else if (ppt <= 4) { if (mtv == 2) QGX_L3(2, 4) else QGX_L3(3, 4) }
    else if (ppt <= 8) { if (mtv == 2) QGX_L3(2, 8) else QGX_L3(3, 8) }
    else return QGX_OK;
#undef QGX_L3
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    done = true;
    return QGX_OK;
}

// hidden layers: k_conv3 (LDS-only operands, prefetched staging) where it measured faster (CIN <= 64 at
// 64x64: -8..-13 %; the 128->64 5x5 layer ties), else the one-shot k_conv.  "v3" option: -1 auto, 0 off,
// 1 = 4 waves, weight slice per tap row; 2 = 4 waves, slice per chunk; 6 / 7 = the same with 8 waves and
// double-height tiles.  Auto: 8 waves (slice per tap row for CIN = 32, per chunk for CIN = 64), falling
// back to 4 waves and then to k_conv when the LDS budget does not fit.
template <int CIN, int COUT, int KS>
static int conv_hidden(qgx_generator *g, int layer, const LayerHost &L, const float *in, float *out, int B,
                       int N, hipStream_t st) {
    if (g->opt_small && small_ensemble(B, N)) {
        // wide layers: split K per 16-channel chunk (8 / 4 partial sums) so that a single member still spreads
        // over >= 128 workgroups; the same split for every small ensemble, so that results do not depend
        // on the member count within this kernel family
        return launch_conv_small<CIN, COUT, KS, CIN >= 64 ? 16 : 32>(g, layer, L, in, out, B, N, st);
    }
    bool done = false;
    int rc = QGX_OK;
    const int v3 = g->opt_v3;
    if (v3 < 0) {
        if (CIN <= 64) {
            rc = CIN == 64 ? launch_conv3<CIN, COUT, KS, KS * KS, 16, 8>(g, layer, L, in, out, B, N, st, done)
                           : launch_conv3<CIN, COUT, KS, KS, 16, 8>(g, layer, L, in, out, B, N, st, done);
            if (!rc && !done) rc = launch_conv3<CIN, COUT, KS, KS>(g, layer, L, in, out, B, N, st, done);
        }
    } else if (v3 == 1) rc = launch_conv3<CIN, COUT, KS, KS>(g, layer, L, in, out, B, N, st, done);
    else if (v3 == 2) rc = launch_conv3<CIN, COUT, KS, KS * KS>(g, layer, L, in, out, B, N, st, done);
    else if (v3 == 6) rc = launch_conv3<CIN, COUT, KS, KS, 16, 8>(g, layer, L, in, out, B, N, st, done);
    else if (v3 == 7) rc = launch_conv3<CIN, COUT, KS, KS * KS, 16, 8>(g, layer, L, in, out, B, N, st, done);
    if (rc || done) return rc;
    if (g->opt_cc == 32) return launch_conv<CIN, COUT, KS, 32, false, false>(g, layer, L, in, out, B, N, COUT, st);
    return launch_conv<CIN, COUT, KS, 16, false, false>(g, layer, L, in, out, B, N, COUT, st);
}

static int launch_conv_last(qgx_generator *g, const LayerHost &L, const float *in, float *out, int B, int N,
                            int n_out, hipStream_t st) {
    int R = choose_rows(N);
    if (g->opt_small && R > 0 && B * (N / R) <= 192) R = 1;      // small ensembles: one row per workgroup
    else if (R > 2 && N % 2 == 0) R = 2;                         // 37-55 KB of LDS: 2-4 workgroups per CU hide the
                                                                 // load -> barrier -> compute -> store chain (-10..-20 %)
    if (g->opt_last_rows > 0 && N % g->opt_last_rows == 0) R = g->opt_last_rows;
    QGX_REQUIRE(R > 0 && N % R == 0, "generator: unsupported grid size N=%d", N);
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, 7, st, prof_stop); if (prc) return prc; }
    ConvArgs a = {};
    a.in = in; a.out = out; a.w = nullptr; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.N = N; a.R = R; a.cout_real = n_out;
    const bool split = R * N == 64;                        // one wave of pixels: the waves split the channels
    const size_t lds = (size_t)(R + 2) * N * 36 * sizeof(float) + (split ? 4 * 64 * 2 * sizeof(float) : 0);
    QGX_REQUIRE(lds <= 160 * 1024, "generator: LDS patch %zu B too large for N=%d", lds, N);
    if (split) {
        auto kern = k_conv_last<32, 3, 4>;
        { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
        hipLaunchKernelGGL(kern, dim3(B * (N / R)), dim3(256), lds, st, a, L.wv_host);
    } else {
        auto kern = k_conv_last<32, 3>;
        { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
        hipLaunchKernelGGL(kern, dim3(B * (N / R)), dim3(256), lds, st, a, L.wv_host);
    }
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}

// ---- 16-bit matrix-core path (conv_half.hpp) ------------------------------------------------------
// rows per 8-wave workgroup: 16 or 24 M-tiles of 32 pixels
static int rows_half(int N) {
    int R = 0;
    if (N <= 512 && 512 % N == 0) R = 512 / N;
    else if (N <= 768 && 768 % N == 0) R = 768 / N;
    if (R == 0 || N % R) return 0;
    if ((size_t)(R + 4) * N * 80 + 2 * 5 * 4 * 64 * 16 > 160 * 1024 - 256) return 0;
    return R;
}
// rows per tile of k_convh2 (4 waves x MT M-tiles of 32 pixels), 0 when the grid size has no specialisation
static int rows_h2(int N) {
    switch (N) {
        case 32: return 8;
        case 48: return 8;
        case 64: return 4;
        case 96: return 4;
        case 128: return 2;
        default: return 0;
    }
}
// with the split-K variant for single members the 16-bit path wins at every ensemble size on the grids
// k_convh2 is specialised for, so it is always taken there ("half_min_tiles" = 1; raise it to send small
// ensembles to the exact-f32 split-K kernels)
static bool half_path_ok(const qgx_generator *g, int B, int N) {
#ifndef QGX_AB
    // product library: f16x3 on the grids k_convh2 is specialised for, exact f32 everywhere else
    if (g->opt_precision != 3 || g->opt_h2 != 3 || rows_h2(N) <= 0) return false;
#endif
    if (N > 128 || choose_rows(N) <= 0) return false;
    int R = (g->opt_h2 == 3 && g->opt_precision == 3) ? rows_h2(N) : 0;
    if (R == 0) R = g->opt_half_nw == 4 ? choose_rows(N) : rows_half(N);
    if (R <= 0 || N % R) return false;
    return B * (N / R) >= g->opt_half_min_tiles;
}

#ifdef QGX_AB
template <int CIN, int COUT, int KS, int NS, bool OUTF32>
static int launch_convh(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B, int N,
                        hipStream_t st) {
    constexpr int TPS = KS == 5 ? 5 : 9;
    // two shapes: 4 waves x 2 workgroups per CU (swizzled 64-byte patch pixels; the two workgroups run out
    // of phase, so one's staging overlaps the other's MFMAs) or 8 waves x 1 workgroup (double-height tiles)
    const bool four = g->opt_half_nw == 4;
    const int R = four ? choose_rows(N) : rows_half(N);
    QGX_REQUIRE(R > 0 && N % R == 0, "generator: 16-bit path does not support N=%d", N);
    const int nw = four ? 4 : 8;
    const int PR = R + KS - 1;
    const int ntiles = R * N / 32;
    const int mtv = ntiles / nw;
    const int ppt = (PR * N * 4 + nw * 64 - 1) / (nw * 64);
    const size_t lds = (size_t)PR * N * (four ? 64 : 80) + (size_t)2 * TPS * 4 * COUT * 16 + 3 * COUT * sizeof(float);
    QGX_REQUIRE(lds <= 160 * 1024 - 256 && (mtv == 2 || mtv == 3) && ntiles % nw == 0 && ppt <= 12,
                "generator: 16-bit path tile shape unsupported for N=%d", N);
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    ConvHArgs a = {};
    a.in = in; a.out = out; a.w = L.wh[NS - 1]; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.unscale = L.wh_unscale[NS - 1] / g->opt_ascale; a.ascale = OUTF32 ? 1.f : g->opt_ascale;
    a.range = g->range_dev; a.range_bit = 1u << layer;
    a.N = N; a.R = R;
    a.stamps = layer == g->stamp_layer ? g->stamps : nullptr;
    const int total_tiles = B * (N / R);
    int grid = 256 * (lds * 2 <= 160 * 1024 ? 2 : 1);
    if (grid > total_tiles) grid = total_tiles;
#define QGX_LH(MTV, PPTV, NWV, SWZV)                                                                          \
    {                                                                                                         \
        auto kern = k_convh<CIN, COUT, KS, NS, MTV, TPS, PPTV, OUTF32, NWV, SWZV>;                            \
        { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; } \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NWV * 64), lds, st, a, total_tiles);                        \
    }
    if (four) {
        if (mtv == 2) { if (ppt <= 6) QGX_LH(2, 6, 4, true) else if (ppt <= 8) QGX_LH(2, 8, 4, true) else QGX_LH(2, 12, 4, true) }
        else QGX_LH(3, 12, 4, true)
    } else {
        QGX_REQUIRE(ppt <= 10, "generator: 16-bit path tile shape unsupported for N=%d", N);
        if (mtv == 2) { if (ppt <= 6) QGX_LH(2, 6, 8, false) else QGX_LH(2, 10, 8, false) }
        else { if (ppt <= 6) QGX_LH(3, 6, 8, false) else QGX_LH(3, 10, 8, false) }
    }
#undef QGX_LH
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}

#endif  // QGX_AB

static int launch_conv_last(qgx_generator *g, const LayerHost &L, const float *in, float *out, int B, int N,
                            int n_out, hipStream_t st);

// f16x3 hidden layers at the grid sizes with a compile-time specialisation (k_convh2; 2 workgroups per CU
// where the LDS budget allows)
template <int CIN, int COUT, int KS, bool OUTF32, int NN, int MT>
static int launch_convh2_n(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B,
                           hipStream_t st) {
    constexpr int TPS = KS == 5 ? 5 : 9;
    constexpr int R = 4 * MT * 32 / NN, PR = R + KS - 1, PW = NN + 2 * (KS / 2);
    constexpr bool WDB = KS == 3;                                // double-buffered weight slice (3x3 layers)
    constexpr size_t lds = (size_t)PR * PW * 80 + (size_t)(WDB ? 2 : 1) * TPS * 4 * COUT * 16 + 3 * COUT * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    ConvHArgs a = {};
    a.in = in; a.out = out; a.w = L.wh[1]; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.unscale = L.wh_unscale[1] / g->opt_ascale; a.ascale = OUTF32 ? 1.f : g->opt_ascale;
    a.range = g->range_dev; a.range_bit = 1u << layer;
    a.N = NN; a.R = R;
    a.prio_alt = KS == 5 ? g->opt_prio_alt : (g->opt_prio_alt == 3 ? 3 : 0);      // measured: -2 % on the 5x5 layer, nothing on the 3x3 layers (3: by phase)
    a.stamps = layer == g->stamp_layer ? g->stamps : nullptr;
    const int total_tiles = B * (NN / R);
    int grid = lds * 2 <= 160 * 1024 ? 512 : 256;
    if (g->opt_h2_grid > 0) grid = g->opt_h2_grid;
    if (grid > total_tiles) grid = total_tiles;
    constexpr bool TWO = lds * 2 <= 160 * 1024 && MT == 2;        // two workgroups per CU: <= 256 registers
    auto kern = g->opt_pair && KS == 3 ? k_convh2<CIN, COUT, KS, NN, MT, TPS, OUTF32, WDB, TWO, KS == 3>
                                        : k_convh2<CIN, COUT, KS, NN, MT, TPS, OUTF32, WDB, TWO, false>;
    { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, total_tiles);
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}

// single members: split K over the 16-channel chunks (k_convh2<PART> + k_convh_reduce), so that the wide
// layers spread over >= 64 workgroups
template <int CIN, int COUT, int KS, bool OUTF32, int NN, int MT>
static int launch_convh2_part_n(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B,
                                hipStream_t st) {
    constexpr int TPS = KS == 5 ? 5 : 9, NCH = CIN / 16;
    constexpr int R = 4 * MT * 32 / NN, PR = R + KS - 1, PW = NN + 2 * (KS / 2);
    constexpr bool WDB = KS == 3;
    constexpr size_t lds = (size_t)PR * PW * 80 + (size_t)(WDB ? 2 : 1) * TPS * 4 * COUT * 16 + 3 * COUT * sizeof(float);
    constexpr int nsplit = NCH >= 8 ? 8 : NCH;
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    const size_t npix = (size_t)B * NN * NN;
    if (g->part_elems < npix * COUT * nsplit) {
        if (g->part) (void)hipFree(g->part);
        g->part = nullptr; g->part_elems = 0;
        QGX_HIP(hipMalloc((void **)&g->part, npix * COUT * nsplit * sizeof(float)));
        g->part_elems = npix * COUT * nsplit;
    }
    ConvHArgs a = {};
    a.in = in; a.out = g->part; a.w = L.wh[1]; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.unscale = L.wh_unscale[1] / g->opt_ascale; a.ascale = OUTF32 ? 1.f : g->opt_ascale;
    a.range = g->range_dev; a.range_bit = 1u << layer;
    a.N = NN; a.R = R; a.npix_total = npix;
    const int total_tiles = B * (NN / R);
    constexpr bool TWO = lds * 2 <= 160 * 1024 && MT == 2;
    auto kern = k_convh2<CIN, COUT, KS, NN, MT, TPS, OUTF32, WDB, TWO, false, true>;
    { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
    hipLaunchKernelGGL(kern, dim3(total_tiles, nsplit), dim3(256), lds, st, a, total_tiles);
    const size_t n = npix * (COUT / 8);
    hipLaunchKernelGGL((k_convh_reduce<COUT, OUTF32>), dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, st,
                       (const float *)g->part, nsplit, npix, (const float *)L.bias, (const float *)L.scale, (const float *)L.shift,
                       a.unscale, a.ascale, out, a.range, a.range_bit);
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}

template <int CIN, int COUT, int KS, bool OUTF32>
static int launch_convh2_part(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B, int N,
                              hipStream_t st, bool &done) {
    done = true;
    switch (N) {
        case 32: return launch_convh2_part_n<CIN, COUT, KS, OUTF32, 32, 2>(g, layer, L, in, out, B, st);
        case 48: return launch_convh2_part_n<CIN, COUT, KS, OUTF32, 48, 3>(g, layer, L, in, out, B, st);
        // (half-height tiles for the smallest ensembles: 32 instead of 16 workgroups per member and split — small_tiles())
        case 64: return small_tiles(g, B, 64) && B == 1      // (at 4 members the 5x5 layer's 8 splits x 128 tiles are slower)
                            ? launch_convh2_part_n<CIN, COUT, KS, OUTF32, 64, 1>(g, layer, L, in, out, B, st)
                                              : launch_convh2_part_n<CIN, COUT, KS, OUTF32, 64, 2>(g, layer, L, in, out, B, st);
        case 96: return launch_convh2_part_n<CIN, COUT, KS, OUTF32, 96, 3>(g, layer, L, in, out, B, st);
        case 128: return launch_convh2_part_n<CIN, COUT, KS, OUTF32, 128, 2>(g, layer, L, in, out, B, st);
        default: done = false; return QGX_OK;
    }
}

// the 5x5 layer as ONE 8-wave workgroup per CU (R = 8 rows at 64 x 64: row halo 1.5x instead of 2x, half the
// weight traffic, double-buffered weight slice, 7 + 3 prefetch registers per thread)
template <int NN, int MT, int TW = NN>
static int launch_convh2_w8(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B, hipStream_t st) {
    constexpr int CIN = 128, COUT = 64, KS = 5, TPS = 5, NW = 8;
    constexpr int R = NW * MT * 32 / TW, PR = R + KS - 1, PW = TW + 4;
    constexpr size_t lds = (size_t)PR * PW * 80 + (size_t)2 * TPS * 4 * COUT * 16 + 3 * COUT * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    ConvHArgs a = {};
    a.in = in; a.out = out; a.w = L.wh[1]; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.unscale = L.wh_unscale[1] / g->opt_ascale; a.ascale = g->opt_ascale;
    a.range = g->range_dev; a.range_bit = 1u << layer;
    a.N = NN; a.R = R;
    a.stamps = layer == g->stamp_layer ? g->stamps : nullptr;
    const int total_tiles = B * (NN / R) * (NN / TW);
    int grid = 256;
    if (grid > total_tiles) grid = total_tiles;
    auto kern = k_convh2<CIN, COUT, KS, NN, MT, TPS, false, true, false, false, false, NW, TW>;
    { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a, total_tiles);
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}

// the 3x3 layers in the same 8-wave shape (64 x 64)
template <int CIN, bool OUTF32, int NN = 64, int TW = NN, int NW = 8>
static int launch_convh2_w8_3x3(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B, hipStream_t st) {
    constexpr int COUT = 32, KS = 3, MT = 2, TPS = 9;
    constexpr int R = NW * MT * 32 / TW, PR = R + KS - 1, PW = TW + 2;
    constexpr size_t lds = (size_t)PR * PW * 80 + (size_t)2 * TPS * 4 * COUT * 16 + 3 * COUT * sizeof(float);
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    ConvHArgs a = {};
    a.in = in; a.out = out; a.w = L.wh[1]; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.unscale = L.wh_unscale[1] / g->opt_ascale; a.ascale = OUTF32 ? 1.f : g->opt_ascale;
    a.range = g->range_dev; a.range_bit = 1u << layer;
    a.N = NN; a.R = R;
    a.stamps = layer == g->stamp_layer ? g->stamps : nullptr;
    const int total_tiles = B * (NN / R) * (NN / TW);
    int grid = 256;
    if (grid > total_tiles) grid = total_tiles;
    auto kern = k_convh2<CIN, COUT, KS, NN, MT, TPS, OUTF32, true, false, true, false, NW, TW>;
    { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a, total_tiles);
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}

template <int CIN, int COUT, int KS, bool OUTF32>
static int launch_convh2(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B, int N,
                         hipStream_t st, bool &done) {
    done = true;
    if constexpr (KS == 3 && COUT == 32) {
        if ((g->opt_h2_w8 & 2) && g->opt_pair && N == 64 && B * 8 >= 256) return launch_convh2_w8_3x3<CIN, OUTF32>(g, layer, L, in, out, B, st);
        // 96 = 3 x 32: tiles of 16 rows x 32 columns give the 3x3 layers the two-M-tiles-per-wave 8-wave shape too
        if ((g->opt_h2_w8 & 2) && g->opt_pair && g->opt_h2_x96 && N == 96 && B * 18 >= 256) {
            // 16 x 32 tiles on 8 waves (18 per member) or 12 x 32 tiles on 6 waves (24 per member): by rounds x cost per tile, as the
            // Winograd layer's shapes (launch_convw); a 6-wave tile costs 0.85 of an 8-wave one (measured: 32 members 32.4 -> 30.4 us
            // per 32 -> 32 layer, 16 members 23.0 -> 21.1, 64 members 52.1 / 52.2)
            const int r16 = (B * 18 + 255) / 256, r12 = (B * 24 + 255) / 256;
            const bool rows12 = g->opt_h2_rows96 == 12 || (g->opt_h2_rows96 == 0 && 0.85 * r12 < 1.0 * r16);
            return rows12 ? launch_convh2_w8_3x3<CIN, OUTF32, 96, 32, 6>(g, layer, L, in, out, B, st)
                          : launch_convh2_w8_3x3<CIN, OUTF32, 96, 32>(g, layer, L, in, out, B, st);
        }
    }
    if constexpr (KS == 5 && CIN == 128) {
        // one 8-wave workgroup per CU once its double-height tiles fill the CUs: -5.5 % at 64 x 64, -3.5 % at
        // 32 x 32; the three-tile-per-wave shapes of 96 / 48 spill at 256 registers (+40 %), 128 ties
        if (g->opt_h2_w8) {
            if (N == 64 && B * 8 >= 256)
                return g->opt_h2_tw32 ? launch_convh2_w8<64, 2, 32>(g, layer, L, in, out, B, st) : launch_convh2_w8<64, 2>(g, layer, L, in, out, B, st);
            if (N == 32 && B * 2 >= 256) return launch_convh2_w8<32, 2>(g, layer, L, in, out, B, st);
            // 96 = 3 x 32: tiles of 16 rows x 32 columns keep the two-M-tiles-per-wave shape
            if (N == 96 && B * 18 >= g->opt_h2_w8_min96) return launch_convh2_w8<96, 2, 32>(g, layer, L, in, out, B, st);
        }
    }
    switch (N) {
        case 32: return launch_convh2_n<CIN, COUT, KS, OUTF32, 32, 2>(g, layer, L, in, out, B, st);
        case 48: return launch_convh2_n<CIN, COUT, KS, OUTF32, 48, 3>(g, layer, L, in, out, B, st);
        case 64: return small_tiles(g, B, 64) ? launch_convh2_n<CIN, COUT, KS, OUTF32, 64, 1>(g, layer, L, in, out, B, st)
                                              : launch_convh2_n<CIN, COUT, KS, OUTF32, 64, 2>(g, layer, L, in, out, B, st);
        case 96: return launch_convh2_n<CIN, COUT, KS, OUTF32, 96, 3>(g, layer, L, in, out, B, st);
        case 128: return launch_convh2_n<CIN, COUT, KS, OUTF32, 128, 2>(g, layer, L, in, out, B, st);
        default: done = false; return QGX_OK;
    }
}

#ifdef QGX_AB
// 3x3 layers, f16x3, resident weights (k_convh_res); done = false when the tile does not fit in LDS
template <int CIN, int COUT, bool OUTF32>
static int launch_convh_res(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B, int N,
                            hipStream_t st, bool &done) {
    done = false;
    const int R = rows_half(N);
    if (R <= 0 || N % R) return QGX_OK;
    const int PR = R + 2, ntiles = R * N / 32, mtv = ntiles / 8;
    const size_t lds = (size_t)(CIN / 16) * 9 * 4 * COUT * 16 + (size_t)PR * N * 128 + 3 * COUT * sizeof(float);
    const int ppt = (PR * N * 8 + 511) / 512;
    if (lds > 160 * 1024 || ntiles % 8 || (mtv != 2 && mtv != 3) || ppt > 14) return QGX_OK;
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    ConvHArgs a = {};
    a.in = in; a.out = out; a.w = L.wh[1]; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.unscale = L.wh_unscale[1] / g->opt_ascale; a.ascale = OUTF32 ? 1.f : g->opt_ascale;
    a.range = g->range_dev; a.range_bit = 1u << layer;
    a.N = N; a.R = R;
    a.stamps = layer == g->stamp_layer ? g->stamps : nullptr;
    const int total_tiles = B * (N / R);
    int grid = 256;
    if (grid > total_tiles) grid = total_tiles;
#define QGX_LR(MTV, PPTV)                                                                                     \
    {                                                                                                         \
        auto kern = k_convh_res<CIN, COUT, MTV, PPTV, OUTF32>;                                                \
        { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; } \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a, total_tiles);                             \
    }
    if (mtv == 2) { if (ppt <= 10) QGX_LR(2, 10) else QGX_LR(2, 14) }
    else { if (ppt <= 10) QGX_LR(3, 10) else QGX_LR(3, 14) }
#undef QGX_LR
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    done = true;
    return QGX_OK;
}

#endif  // QGX_AB

template <int CIN, int COUT, int NS, bool OUTF32>
static int conv3x3_half(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B, int N,
                        hipStream_t st) {
    if (NS == 2 && g->opt_h2) {
        bool done = false;
        int rc = launch_convh2<CIN, COUT, 3, OUTF32>(g, layer, L, in, out, B, N, st, done);
        if (rc || done) return rc;
    }
#ifdef QGX_AB
    if (NS == 2 && g->opt_res) {
        bool done = false;
        int rc = launch_convh_res<CIN, COUT, OUTF32>(g, layer, L, in, out, B, N, st, done);
        if (rc || done) return rc;
    }
    return launch_convh<CIN, COUT, 3, NS, OUTF32>(g, layer, L, in, out, B, N, st);
#else
    QGX_REQUIRE(false, "generator: no f16x3 kernel for N=%d (half_path_ok admits only the specialised grids)", N);
#endif
}

#ifdef QGX_AB
// the 128 -> 64, 5x5 layer on 16x16x32 MFMAs (k_convh3); grids whose rows tile 256 pixels
static int launch_convh3(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B, int N,
                         hipStream_t st, bool &done) {
    done = false;
    if (N != 64 || !L.wh16) return QGX_OK;
    constexpr int NN = 64, R = 256 / NN, PR = R + 4, PW = NN + 4;
    constexpr size_t lds = (size_t)4 * PR * PW * 48 + (size_t)5 * 2 * 4 * 64 * 16 + 3 * 64 * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    ConvHArgs a = {};
    a.in = in; a.out = out; a.w = L.wh16; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.unscale = L.wh_unscale[1] / g->opt_ascale; a.ascale = g->opt_ascale;
    a.range = g->range_dev; a.range_bit = 1u << layer;
    a.N = N; a.R = R;
    const int total_tiles = B * (N / R);
    int grid = 256;
    if (grid > total_tiles) grid = total_tiles;
    auto kern = k_convh3<NN>;
    { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, total_tiles);
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    done = true;
    return QGX_OK;
}

// the 128 -> 64, 5x5 layer with full-line patch chunks (k_convh4: 8 waves, R = 8)
template <int NN, int MT, int NW>
static int launch_convh4_n(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B, hipStream_t st) {
    constexpr int R = NW * MT * 32 / NN, PR = R + 4, PW = NN + 4;
    constexpr size_t lds = (size_t)PR * PW * 144 + (size_t)2 * 5 * 4 * 64 * 16 + 3 * 64 * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    ConvHArgs a = {};
    a.in = in; a.out = out; a.w = L.wh[1]; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.unscale = L.wh_unscale[1] / g->opt_ascale; a.ascale = g->opt_ascale;
    a.range = g->range_dev; a.range_bit = 1u << layer;
    a.N = NN; a.R = R;
    const int total_tiles = B * (NN / R);
    int grid = 256;
    if (grid > total_tiles) grid = total_tiles;
    auto kern = k_convh4<NN, MT, NW>;
    { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a, total_tiles);
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}
static int launch_convh4(qgx_generator *g, int layer, const LayerHost &L, const void *in, void *out, int B, int N,
                         hipStream_t st, bool &done) {
    done = true;
    if (N == 64 && B * 8 >= 256)
        return g->opt_h4 == 2 ? launch_convh4_n<64, 4, 4>(g, layer, L, in, out, B, st) : launch_convh4_n<64, 2, 8>(g, layer, L, in, out, B, st);
    done = false;
    return QGX_OK;
}

// two fused 3x3 layers (k_convh_pair); 64 x 64 grids
#endif  // QGX_AB

template <int CINA, bool LAST, bool BOUTF32, int NN = 64, int R = 8>
static int launch_convh_pair(qgx_generator *g, int layerA, const LayerHost &LA, const LayerHost &LB, const void *in,
                             void *out, int B, int N, int n_out, hipStream_t st) {
    constexpr int PW = NN + 2;
    QGX_REQUIRE(N == NN, "generator: fused layer pairs compiled for N=%d", NN);
    constexpr size_t reg0 = (size_t)(R + 2) * PW * 144 > (size_t)(R + 4) * PW * 80 ? (size_t)(R + 2) * PW * 144 : (size_t)(R + 4) * PW * 80;
    constexpr size_t lds = reg0 + 3 * (9 * 4 * 32 * 16) + 2 * 96 * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layerA, st, prof_stop); if (prc) return prc; }
    ConvPairArgs a = {};
    a.in = in; a.out = out; a.wA = LA.wh[1]; a.wB = LB.wh[1];
    a.biasA = LA.bias; a.scaleA = LA.scale; a.shiftA = LA.shift;
    a.biasB = LB.bias; a.scaleB = LB.scale; a.shiftB = LB.shift;
    a.unscaleA = LA.wh_unscale[1] / g->opt_ascale; a.unscaleB = LB.wh_unscale[1] / g->opt_ascale; a.ascale = g->opt_ascale;
    a.range = g->range_dev; a.range_bit = 1u << layerA;
    a.n_out = n_out;
    a.stamps = layerA == g->stamp_layer ? g->stamps : nullptr;
    const int total_tiles = B * (N / R);
    int grid = 256;
    if (grid > total_tiles) grid = total_tiles;
    auto kern = k_convh_pair<CINA, NN, LAST, BOUTF32, CINA == 32, R>;
#ifdef QGX_AB
    if (!g->opt_pair_lp) kern = k_convh_pair<CINA, NN, LAST, BOUTF32, false, R>;
#endif
    { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a, total_tiles);
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}

template <int NIN>
static int launch_convh_first(qgx_generator *g, const LayerHost &L, const float *in, void *out, int B, int N,
                              hipStream_t st, bool wino_next = false, bool planar = false) {
    int R = choose_rows(N);
    QGX_REQUIRE(R > 0 && N % R == 0 && N % 4 == 0, "generator: unsupported grid size N=%d", N);
    // one 8-wave workgroup per CU with double-height tiles where two M-tiles per wave result (64 x 64, 32 x 32) and
    // the ensemble fills the CUs: of two co-resident 4-wave workgroups the older wins every issue arbitration and
    // the younger finishes 30 % later (global-clock stamps), and the 57 KB weight set is staged once per CU
    const bool w8 = g->opt_h2_w8 && R * N / 32 == 8 && N % (2 * R) == 0 && B * (N / (2 * R)) >= 256;
    if (w8) R *= 2;
    const bool half_rows = !w8 && small_tiles(g, B, N) && R % 2 == 0;      // one M-tile per wave (small_tiles())
    if (half_rows) R /= 2;
    const int nw = w8 ? 8 : 4;
    const int PR = R + 4, ntiles = R * N / 32;
    constexpr int nstep = NIN == 4 ? 7 : 4;
    const size_t lds = (size_t)nstep * 4 * 128 * 16 + (size_t)2 * PR * N * NIN * 4 + 3 * 128 * sizeof(float);
    const int ppt = (PR * NIN * (N / 4) + nw * 64 - 1) / (nw * 64);
    QGX_REQUIRE(lds <= 160 * 1024 - 256 && ppt <= 3 && (ntiles == 2 * nw || ntiles == 3 * nw || (half_rows && ntiles == nw)),
                "generator: 16-bit first layer unsupported for N=%d", N);
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, 0, st, prof_stop); if (prc) return prc; }
    ConvHFirstArgs a = {};
    a.stamps = g->stamp_layer == 0 ? g->stamps : nullptr;
    a.in = in; a.out = out; a.w = L.whf; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    a.unscale = L.whf_unscale; a.ascale = g->opt_ascale; a.N = N; a.R = R;
    a.guard_mul = wino_next ? 16.f : 1.f;
    a.range = g->range_dev; a.range_bit = 1u;
    const int total_tiles = B * (N / R);
    const int wgs = lds * 2 <= 160 * 1024 ? 2 : 1;
    int grid = 256 * wgs;
    if (grid > total_tiles) grid = total_tiles;
#ifdef QGX_AB
#define QGX_LF_KERN(MTV, PPTV, NWV) (planar ? k_convh_first<NIN, MTV, PPTV, NWV, true> : k_convh_first<NIN, MTV, PPTV, NWV, false>)
#else
#define QGX_LF_KERN(MTV, PPTV, NWV) k_convh_first<NIN, MTV, PPTV, NWV, false>
#endif
#define QGX_LF(MTV, PPTV, NWV)                                                                                \
    {                                                                                                         \
        auto kern = QGX_LF_KERN(MTV, PPTV, NWV);                                                              \
        { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; } \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NWV * 64), lds, st, a, total_tiles);                        \
    }
    if (half_rows) { if (ppt <= 2) QGX_LF(1, 2, 4) else QGX_LF(1, 3, 4) }
    else if (w8) { if (ppt <= 2) QGX_LF(2, 2, 8) else QGX_LF(2, 3, 8) }
    else if (ntiles == 8) { if (ppt <= 2) QGX_LF(2, 2, 4) else QGX_LF(2, 3, 4) }
    else { if (ppt <= 2) QGX_LF(3, 2, 4) else QGX_LF(3, 3, 4) }
#undef QGX_LF
#undef QGX_LF_KERN
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}

// layer 2 as a 1-D Winograd convolution (conv_wino.hpp); done = false: no specialisation for this grid / ensemble size
template <int NN, int TW, int R, bool PL = false>
static int launch_convw_n(qgx_generator *g, int layer, const LayerHost &L, int which, const void *in, void *out, int B,
                          hipStream_t st) {
    hipEvent_t prof_stop;
    { int prc = prof_begin(g, layer, st, prof_stop); if (prc) return prc; }
    ConvWArgs a = {};
    a.in = in; a.out = out; a.w = L.ww[which]; a.bias = L.bias; a.scale = L.scale; a.shift = L.shift;
    for (int p = 0; p < 8; ++p) a.pscale[p] = L.ww_unscale[which][p] / g->opt_ascale;
    a.ascale = g->opt_ascale;
    a.range = g->range_dev; a.range_bit = 1u << layer;
    const int total_tiles = B * (NN / R) * (NN / TW);
    constexpr size_t lds = convw_lds_bytes(NN, TW, R, PL);
    static_assert(lds <= 160 * 1024 - 256, "k_convw: LDS");
    if (!PL && g->opt_wino2 && g->opt_wino_exp == 0) {
        bool done2 = false;
        const int rc2 = launch_convw2(NN, TW, R, a, total_tiles, st, &done2);
        if (rc2) return rc2;
        if (done2) {
            if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
            return QGX_OK;
        }
    }
    const int grid = total_tiles < 256 ? total_tiles : 256;
    void (*kern)(ConvWArgs, int) = k_convw<NN, TW, R, 0, PL>;
#ifdef QGX_AB       // timing experiments: parts of the kernel switched off (wrong results)
    if (g->opt_wino_exp == 1) kern = k_convw<NN, TW, R, 1, PL>;
    else if (g->opt_wino_exp == 2) kern = k_convw<NN, TW, R, 2, PL>;
    else if (g->opt_wino_exp == 4) kern = k_convw<NN, TW, R, 4, PL>;
    else if (g->opt_wino_exp == 5) kern = k_convw<NN, TW, R, 5, PL>;
    else if (g->opt_wino_exp == 6) kern = k_convw<NN, TW, R, 6, PL>;
    else if (g->opt_wino_exp == 7) kern = k_convw<NN, TW, R, 7, PL>;
    else if (g->opt_wino_exp == 8) kern = k_convw<NN, TW, R, 8, PL>;
#endif
    { const int lrc_ = ensure_dynamic_lds((const void *)kern, (int)lds); if (lrc_) return lrc_; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a, total_tiles);
    QGX_HIP(hipGetLastError());
    if (prof_stop) QGX_HIP(hipEventRecord(prof_stop, st));
    return QGX_OK;
}
// tiles of 512 pixels: 8 x 64 (64 x 64, 128 x 128), 16 x 32 (96 x 96, 32 x 32); 16 x 16 at 48 x 48
static int wino_tiles(int B, int N) {
    switch (N) {
        case 32: return B * 2;
        case 48: return B * 9;
        case 64: return B * 8;
        case 96: return B * 18;
        case 128: return B * 32;
        default: return 0;
    }
}
// does the 5x5 layer run as the Winograd form for this ensemble?  (decided BEFORE layer 1: its output layout — channel-planar
// for the MFMA input transform — and its range guard depend on it)
static int wino_size_index(int N) {
    switch (N) { case 32: return 0; case 48: return 1; case 64: return 2; case 96: return 3; case 128: return 4; default: return -1; }
}
static bool wino_applies(const qgx_generator *g, const LayerHost &L, int which, int B, int N) {
    const int tiles = wino_tiles(B, N), si = wino_size_index(N);
    const bool on = g->opt_wino == 1 || (g->opt_wino == 2 && si >= 0 && g->auto_wino_n[si]);
    return on && L.ww[which] && tiles > 0 && tiles >= g->opt_wino_min_tiles;
}
// A/B library only: layer 1 stores channel-planar rows and the Winograd layer's input transform runs on the matrix cores
// (conv_wino.hpp PL).  Measured against the product's pixel-major form (bench_tools/ab_conv.py, wino_pl): 64 x 64 / 128 members
// -3 us on layer 2 and -2...-4 us on layer 1 of 370; 96 x 96 302 vs 206 us, 128 x 128 230 vs 153 us (8-pixel octets make the x halo of
// the column-tiled shapes 1.33x, and the kernels spill) — not in the product.
static bool wino_planar(const qgx_generator *g) {
#ifdef QGX_AB
    return g->opt_wino_pl != 0;
#else
    (void)g;
    return false;
#endif
}
static int launch_convw(qgx_generator *g, int layer, const LayerHost &L, int which, const void *in, void *out, int B, int N,
                        hipStream_t st, bool &done) {
    done = false;
    if (!wino_applies(g, L, which, B, N)) return QGX_OK;
    int rc;
#ifdef QGX_AB
    if (wino_planar(g)) {
        switch (N) {
            case 32: rc = launch_convw_n<32, 32, 16, true>(g, layer, L, which, in, out, B, st); break;
            case 48: rc = launch_convw_n<48, 16, 16, true>(g, layer, L, which, in, out, B, st); break;
            case 64: rc = launch_convw_n<64, 64, 8, true>(g, layer, L, which, in, out, B, st); break;
            case 96: rc = launch_convw_n<96, 32, 16, true>(g, layer, L, which, in, out, B, st); break;
            default: rc = launch_convw_n<128, 32, 16, true>(g, layer, L, which, in, out, B, st); break;   // (64-column tiles: raw + transformed patch > 160 KB)
        }
        if (!rc) done = true;
        return rc;
    }
#endif
    switch (N) {
        case 32: {       // 16 x 32 tiles (2 per member) or 8 x 32 (4 per member): 32 / 64 members 63 -> 42 / 68 -> 49 us, 96 / 128 members stay (73 / 85, 83 / 95)
            const int r16 = (B * 2 + 255) / 256, r8 = (B * 4 + 255) / 256;
            const bool rows8 = g->opt_wino_rows64 == 4 || (g->opt_wino_rows64 == 0 && 0.66 * r8 < 1.0 * r16);
            rc = rows8 ? launch_convw_n<32, 32, 8>(g, layer, L, which, in, out, B, st)
                       : launch_convw_n<32, 32, 16>(g, layer, L, which, in, out, B, st);
            break;
        }
        case 48: rc = launch_convw_n<48, 16, 16>(g, layer, L, which, in, out, B, st); break;
        case 64: {
            // 8-row tiles (8 per member) or 4-row tiles (16 per member, half the work each at a 2 x instead of 1.5 x row halo): by
            // rounds x cost per tile as at 96 x 96 — the small shape pays while the large one leaves CUs idle (measured, layer time in
            // us, 8 / 4 rows: 8 members 65 / 44, 16: 69 / 50, 24: 74 / 87, 31: 80 / 92, 48: 139 / 134 -> 0.66 per tile)
            const int r8 = (B * 8 + 255) / 256, r4 = (B * 16 + 255) / 256;
            const bool rows4 = g->opt_wino_rows64 == 4 || (g->opt_wino_rows64 == 0 && 0.66 * r4 < 1.0 * r8);
            rc = rows4 ? launch_convw_n<64, 64, 4>(g, layer, L, which, in, out, B, st)
                       : launch_convw_n<64, 64, 8>(g, layer, L, which, in, out, B, st);
            break;
        }
        case 96: {
            // 16 x 32 tiles (18 per member) or 12 x 32 (24 per member, 0.75 x the work each at a 1.33 x instead of 1.25 x row halo):
            // 256 persistent workgroups take ceil(tiles / 256) rounds, so the shape is chosen by rounds x cost per tile — at 32 members
            // 3 x 1.0 against 3 x 0.79 (measured 215 -> 171 us), at 24 members 2 x 1.0 against 3 x 0.79 (149 / 160 us), at 128 9 x 1.0
            // against 12 x 0.79 (690 / 749 us)
            const int r16 = (B * 18 + 255) / 256, r12 = (B * 24 + 255) / 256;
            const bool rows12 = g->opt_wino_rows96 == 12 || (g->opt_wino_rows96 == 0 && 0.79 * r12 < 1.0 * r16);
            rc = rows12 ? launch_convw_n<96, 32, 12>(g, layer, L, which, in, out, B, st)
                        : launch_convw_n<96, 32, 16>(g, layer, L, which, in, out, B, st);
            break;
        }
        default: {       // 128 x 128: 8 x 64 tiles (32 per member) or 4 x 64 (64 per member), as at 64 x 64: 2 / 4 members 67 -> 45 / 72 -> 53 us,
                         // 6 / 8 members stay (78 / 89, 86 / 97), 12 members 155 -> 143
            const int r8 = (B * 32 + 255) / 256, r4 = (B * 64 + 255) / 256;
            const bool rows4 = g->opt_wino_rows64 == 4 || (g->opt_wino_rows64 == 0 && 0.66 * r4 < 1.0 * r8);
            rc = rows4 ? launch_convw_n<128, 64, 4>(g, layer, L, which, in, out, B, st)
                       : launch_convw_n<128, 64, 8>(g, layer, L, which, in, out, B, st);
            break;
        }
    }
    if (!rc) done = true;
    return rc;
}

template <int NS>
static int cnn_forward_half(qgx_generator *g, const NetHost &net, const float *x, float *y, int B, int N,
                            hipStream_t st) {
    int rc;
    float *A = g->actA, *Bb = g->actB;
    // optional member sub-batches ("member_chunk"): all layers of one sub-batch before the next, so that the
    // inter-layer activations of a sub-batch can stay in the 256 MB Infinity Cache
    const int mc = g->opt_member_chunk > 0 && g->opt_member_chunk < B ? g->opt_member_chunk : B;
    for (int b0 = 0; b0 < B; b0 += mc) {
        const int Bc = B - b0 < mc ? B - b0 : mc;
        const float *xc = x + (size_t)b0 * net.n_in * N * N;
        float *yc = y + (size_t)b0 * net.n_out * N * N;
        // "fold": layer 1 stores its ReLU output (identity BatchNorm in the epilogue; 50 % exact zeros with the
        // shipped weights) and layer 2 uses the weights / bias with that BatchNorm folded in
        const bool fold = NS == 2 && g->opt_first_h && g->opt_fold && net.L[1].whF;
        LayerHost L0 = net.L[0], L1 = net.L[1];
        if (fold) {
            L0.scale = L0.ones; L0.shift = L0.zeros;
            L1.wh[1] = L1.whF; L1.wh_unscale[1] = L1.whF_unscale; L1.bias = L1.biasF; L1.wh16 = L1.wh16F;
        }
        bool wino2 = false;
        if (NS == 2 && g->opt_first_h) {
            // (will the 5x5 layer run as the Winograd form?  then layer 1's range guard covers its input transform too)
            const int r2w = rows_h2(N);
            wino2 = NS == 2 && wino_applies(g, L1, fold ? 1 : 0, Bc, N) &&
                    !(g->opt_h2 == 3 && r2w > 0 && Bc * (N / r2w) < g->opt_part_max_tiles);
            rc = net.n_in == 4 ? launch_convh_first<4>(g, L0, xc, A, Bc, N, st, wino2, wino2 && wino_planar(g))
                               : launch_convh_first<2>(g, L0, xc, A, Bc, N, st, wino2, wino2 && wino_planar(g));
        } else if (net.n_in == 4) rc = launch_conv<4, 128, 5, 4, true, false, 2, NS>(g, 0, net.L[0], xc, A, Bc, N, 128, st);
        else rc = launch_conv<2, 128, 5, 2, true, false, 2, NS>(g, 0, net.L[0], xc, A, Bc, N, 128, st);
        if (rc) return rc;
#ifdef QGX_AB
        if (g->opt_stop_layer == 1) return QGX_OK;
#endif
        // single members / tiny ensembles (fewer than "part_max_tiles" tiles): split K on the two wide layers and
        // do not fuse (7, 8) (8 tiles of 512 pixels would leave 248 CUs idle)
        const int r2 = rows_h2(N);
        const bool tiny = NS == 2 && g->opt_h2 == 3 && r2 > 0 && Bc * (N / r2) < g->opt_part_max_tiles;
        bool done1 = false;
        if (tiny && (rc = launch_convh2_part<128, 64, 5, false>(g, 1, L1, A, Bb, Bc, N, st, done1))) return rc;
        if (!done1 && wino2) {
            if ((rc = launch_convw(g, 1, L1, fold ? 1 : 0, A, Bb, Bc, N, st, done1))) return rc;
            QGX_REQUIRE(done1, "generator: layer 1 was stored for the Winograd layer, which did not run (N=%d)", N);
        }
#ifdef QGX_AB
        if (!done1 && NS == 2 && g->opt_h4 && (rc = launch_convh4(g, 1, L1, A, Bb, Bc, N, st, done1))) return rc;
        if (!done1 && NS == 2 && g->opt_h3 && (rc = launch_convh3(g, 1, L1, A, Bb, Bc, N, st, done1))) return rc;
#endif
        if (!done1 && NS == 2 && (g->opt_h2 & 2) && (rc = launch_convh2<128, 64, 5, false>(g, 1, L1, A, Bb, Bc, N, st, done1))) return rc;
#ifdef QGX_AB
        if (!done1 && (rc = launch_convh<128, 64, 5, NS, false>(g, 1, L1, A, Bb, Bc, N, st))) return rc;
#else
        QGX_REQUIRE(done1, "generator: no f16x3 kernel for N=%d", N);
#endif
#ifdef QGX_AB
        if (g->opt_stop_layer == 2) return QGX_OK;
#endif
        if (tiny) {
            // (local names: the 64-channel activation of layer 2 is in Bb; `cur` holds a layer's input, `oth` takes its output)
            float *cur = Bb, *oth = A;
            // Layers (3, 4), (5, 6) and (7, 8) each as ONE launch on 2-row strips ("tiny_pairs" bits 2, 1, 0): a single member is
            // a chain of launch latencies, and a strip's fixed costs — layer B's 36 KB of MFMA weights above all — are cheaper
            // than a kernel boundary there; for (3, 4) the one launch also replaces split-K and its combine kernel
            const bool strips = NS == 2 && N == 64 && net.n_out <= 2;
            if (strips && (g->opt_tiny_pairs & 4)) {
                if ((rc = launch_convh_pair<64, false, false, 64, 2>(g, 2, net.L[2], net.L[3], cur, oth, Bc, N, 0, st))) return rc;
                std::swap(cur, oth);
            } else {
                bool done2 = false;
                if ((rc = launch_convh2_part<64, 32, 3, false>(g, 2, net.L[2], cur, oth, Bc, N, st, done2))) return rc;
                if (!done2 && (rc = conv3x3_half<64, 32, NS, false>(g, 2, net.L[2], cur, oth, Bc, N, st))) return rc;
                if ((rc = conv3x3_half<32, 32, NS, false>(g, 3, net.L[3], oth, cur, Bc, N, st))) return rc;
            }
            if (strips && (g->opt_tiny_pairs & 2)) {
                if ((rc = launch_convh_pair<32, false, false, 64, 2>(g, 4, net.L[4], net.L[5], cur, oth, Bc, N, 0, st))) return rc;
                std::swap(cur, oth);
            } else {
                if ((rc = conv3x3_half<32, 32, NS, false>(g, 4, net.L[4], cur, oth, Bc, N, st))) return rc;
                if ((rc = conv3x3_half<32, 32, NS, false>(g, 5, net.L[5], oth, cur, Bc, N, st))) return rc;
            }
            if (strips && (g->opt_tiny_pairs & 1)) {
                if ((rc = launch_convh_pair<32, true, false, 64, 2>(g, 6, net.L[6], net.L[7], cur, yc, Bc, N, net.n_out, st))) return rc;
                continue;
            }
            if ((rc = conv3x3_half<32, 32, NS, true>(g, 6, net.L[6], cur, oth, Bc, N, st))) return rc;
            if ((rc = launch_conv_last(g, net.L[7], oth, yc, Bc, N, net.n_out, st))) return rc;
            continue;
        }
        if (NS == 2 && g->opt_fuse && N == 64 && net.n_out <= 2) {
            // 3x3 layers fused pairwise (the intermediate activation stays in LDS); "fuse" bits: 1 = layers
            // (5, 6), 2 = layers (7, 8), 4 = layers (3, 4) — the 64-channel pair measured slower fused
            if (g->opt_fuse & 4) {
                if ((rc = launch_convh_pair<64, false, false>(g, 2, net.L[2], net.L[3], Bb, A, Bc, N, 0, st))) return rc;
            } else {
                if ((rc = conv3x3_half<64, 32, NS, false>(g, 2, net.L[2], Bb, A, Bc, N, st))) return rc;
                if ((rc = conv3x3_half<32, 32, NS, false>(g, 3, net.L[3], A, Bb, Bc, N, st))) return rc;
            }
            // here the activation is in A (fused pair) or Bb (two kernels)
            float *cur = (g->opt_fuse & 4) ? A : Bb, *oth = (g->opt_fuse & 4) ? Bb : A;
            // (4-row strips for the pairs, as the Winograd layer's small shape, measured slower at 8 ... 48 members: 25.4 -> 30.8 us at
            // 16, 26.3 -> 47.4 at 24 — a strip's fixed costs, layer B's 36 KB of weights above all, do not halve with its rows)
            if (g->opt_fuse & 1) {
                if ((rc = launch_convh_pair<32, false, false>(g, 4, net.L[4], net.L[5], cur, oth, Bc, N, 0, st))) return rc;
                std::swap(cur, oth);
            } else {
                if ((rc = conv3x3_half<32, 32, NS, false>(g, 4, net.L[4], cur, oth, Bc, N, st))) return rc;
                if ((rc = conv3x3_half<32, 32, NS, false>(g, 5, net.L[5], oth, cur, Bc, N, st))) return rc;
            }
            if (g->opt_fuse & 2) {
                if ((rc = launch_convh_pair<32, true, false>(g, 6, net.L[6], net.L[7], cur, yc, Bc, N, net.n_out, st))) return rc;
            } else {
                if ((rc = conv3x3_half<32, 32, NS, true>(g, 6, net.L[6], cur, oth, Bc, N, st))) return rc;
                if ((rc = launch_conv_last(g, net.L[7], oth, yc, Bc, N, net.n_out, st))) return rc;
            }
            continue;
        }
        if ((rc = conv3x3_half<64, 32, NS, false>(g, 2, net.L[2], Bb, A, Bc, N, st))) return rc;
        if ((rc = conv3x3_half<32, 32, NS, false>(g, 3, net.L[3], A, Bb, Bc, N, st))) return rc;
        if (NS == 2 && g->opt_fuse96 && N == 96 && net.n_out <= 2 && Bc * 24 >= 256) {
            // 96 x 96: the pair kernel on 4-row strips (a 6-row intermediate patch of 98 columns is 85 KB); "fuse96" bits as "fuse"
            float *cur = Bb, *oth = A;
            if (g->opt_fuse96 & 1) {
                if ((rc = launch_convh_pair<32, false, false, 96, 4>(g, 4, net.L[4], net.L[5], cur, oth, Bc, N, 0, st))) return rc;
                std::swap(cur, oth);
            } else {
                if ((rc = conv3x3_half<32, 32, NS, false>(g, 4, net.L[4], cur, oth, Bc, N, st))) return rc;
                if ((rc = conv3x3_half<32, 32, NS, false>(g, 5, net.L[5], oth, cur, Bc, N, st))) return rc;
            }
            if (g->opt_fuse96 & 2) {
                if ((rc = launch_convh_pair<32, true, false, 96, 4>(g, 6, net.L[6], net.L[7], cur, yc, Bc, N, net.n_out, st))) return rc;
            } else {
                if ((rc = conv3x3_half<32, 32, NS, true>(g, 6, net.L[6], cur, oth, Bc, N, st))) return rc;
                if ((rc = launch_conv_last(g, net.L[7], oth, yc, Bc, N, net.n_out, st))) return rc;
            }
            continue;
        }
        if ((rc = conv3x3_half<32, 32, NS, false>(g, 4, net.L[4], Bb, A, Bc, N, st))) return rc;
        if ((rc = conv3x3_half<32, 32, NS, false>(g, 5, net.L[5], A, Bb, Bc, N, st))) return rc;
        if ((rc = conv3x3_half<32, 32, NS, true>(g, 6, net.L[6], Bb, A, Bc, N, st))) return rc;
        if ((rc = launch_conv_last(g, net.L[7], A, yc, Bc, N, net.n_out, st))) return rc;
    }
    return QGX_OK;
}

int generator_select_workspace(qgx_generator *g, int idx) {
    QGX_REQUIRE(g && (idx == 0 || idx == 1), "generator_select_workspace: bad argument");
    if (idx == g->ws_active) return QGX_OK;
    std::swap(g->cap_elems, g->ws_other.cap_elems);
    std::swap(g->actA, g->ws_other.actA); std::swap(g->actB, g->ws_other.actB);
    std::swap(g->X, g->ws_other.X); std::swap(g->Y0, g->ws_other.Y0); std::swap(g->Y1, g->ws_other.Y1);
    std::swap(g->part, g->ws_other.part); std::swap(g->part_elems, g->ws_other.part_elems);
    g->ws_active = idx;
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
    QGX_HIP(hipMalloc((void **)&g->X, need * 6 * sizeof(float)));    // (B, 4, N, N), and behind it (B, 2, N, N) for a regression net
    QGX_HIP(hipMalloc((void **)&g->Y0, need * 2 * sizeof(float)));
    QGX_HIP(hipMalloc((void **)&g->Y1, need * 2 * sizeof(float)));
    g->cap_elems = need;
    return QGX_OK;
}

// AndrewCNN.forward: x planar (B,n_in,N,N) -> y planar (B,n_out,N,N)
static int cnn_forward(qgx_generator *g, const NetHost &net, const float *x, float *y, int B, int N,
                       hipStream_t st) {
    int rc;
    if (g->opt_precision && half_path_ok(g, B, N))
#ifdef QGX_AB
        return g->opt_precision == 1 ? cnn_forward_half<1>(g, net, x, y, B, N, st) : cnn_forward_half<2>(g, net, x, y, B, N, st);
#else
        return cnn_forward_half<2>(g, net, x, y, B, N, st);
#endif
    float *A = g->actA, *Bb = g->actB;
    if (net.n_in == 4) rc = g->opt_first_split == 2 ? launch_conv<4, 128, 5, 4, true, false, 2>(g, 0, net.L[0], x, A, B, N, 128, st)
                       : g->opt_first_split == 4 ? launch_conv<4, 128, 5, 4, true, false, 4>(g, 0, net.L[0], x, A, B, N, 128, st)
                                                 : launch_conv<4, 128, 5, 4, true, false, 1>(g, 0, net.L[0], x, A, B, N, 128, st);
    else rc = g->opt_first_split == 2 ? launch_conv<2, 128, 5, 2, true, false, 2>(g, 0, net.L[0], x, A, B, N, 128, st)
            : g->opt_first_split == 4 ? launch_conv<2, 128, 5, 2, true, false, 4>(g, 0, net.L[0], x, A, B, N, 128, st)
                                      : launch_conv<2, 128, 5, 2, true, false, 1>(g, 0, net.L[0], x, A, B, N, 128, st);
    if (rc) return rc;
    // calibration runs record the largest stored activation of every layer (k_absmax keeps a running maximum
    // in calib_dev[2 * layer + 1]: it updates word [1] of the pair it is given)
    auto rec = [&](int layer, const float *buf, int ch) {
        if (g->calib_dev)
            hipLaunchKernelGGL(k_absmax, dim3(256), dim3(256), 0, st, buf, (size_t)B * N * N * ch, g->calib_dev + 2 * layer);
    };
    rec(0, A, 128);
    if ((rc = conv_hidden<128, 64, 5>(g, 1, net.L[1], A, Bb, B, N, st))) return rc;
    rec(1, Bb, 64);
    if ((rc = conv_hidden<64, 32, 3>(g, 2, net.L[2], Bb, A, B, N, st))) return rc;
    rec(2, A, 32);
    if ((rc = conv_hidden<32, 32, 3>(g, 3, net.L[3], A, Bb, B, N, st))) return rc;
    rec(3, Bb, 32);
    if ((rc = conv_hidden<32, 32, 3>(g, 4, net.L[4], Bb, A, B, N, st))) return rc;
    rec(4, A, 32);
    if ((rc = conv_hidden<32, 32, 3>(g, 5, net.L[5], A, Bb, B, N, st))) return rc;
    rec(5, Bb, 32);
    if ((rc = conv_hidden<32, 32, 3>(g, 6, net.L[6], Bb, A, B, N, st))) return rc;
    rec(6, A, 32);
    if (g->opt_last_valu) rc = launch_conv_last(g, net.L[7], A, y, B, N, net.n_out, st);
    else rc = launch_conv<32, 2, 3, 16, false, true>(g, 7, net.L[7], A, y, B, N, net.n_out, st);
    if (rc) return rc;
    return QGX_OK;
}


// ---- f16x3 range calibration (run once by qgx_generator_create) ------------------------------------------------
// The f16x3 arithmetic stores every activation x as hi = f16(s x), lo = f16(s x - hi) with ONE power-of-two scale s
// for the whole net.  It is float32-class only while the stored values stay inside a window: above 65504 the hi part
// overflows; below 2^-2 (relative to a layer's largest value) the lo part is subnormal and the pair no longer carries
// 22 bits against that layer's scale.  Nothing about a user-trained net guarantees that, so the nets are evaluated
// here once in EXACT f32 on calibration inputs (unit-variance white noise, smooth large-scale fields of amplitude 2-3,
// constants: what ChannelwiseScaler-normalised PV and N(0,1) latent noise look like) and the largest stored
// activation of every layer decides:
//   * s = the power of two that puts the largest layer maximum at <= 2^10 (64x headroom below the f16 overflow for
//     inputs hotter than the calibration set; the run-time guard range_guard() catches what still escapes),
//     s = 1 whenever that already holds (the shipped nets);
//   * "fold" (layer 1 stores its pre-BatchNorm ReLU output) only if that tensor fits the same window;
//   * if the layer maxima span more than the window (2^12), f16x3 cannot be float32-class for this net:
//     precision 0 (the exact-f32 MFMA kernels) becomes the default.
// The 1-D Winograd form of the 5x5 layer (conv_wino.hpp) multiplies 0.4 x as much, but its float32 transforms and
// accumulators carry the condition of the Toom-Cook matrices: for the layer, 3-4 x the rounding error of the 25-tap form
// (float32 evaluation of both: tests/test_conv_transform_numerics_cpu.py), and how much of that reaches the net's output
// depends on the weights behind it (shipped nets: 3e-6 ... 9e-6 of max|y|; random-weight nets up to 5e-5).  So it is
// MEASURED per generator AND per grid size it is specialised for (32, 48, 64, 96, 128: the tile shapes, and with them the
// rows and quads a workgroup transforms together, differ between them): every net is evaluated on calibration inputs
// (white noise, one member hotter; smooth fields; constants) with the Winograd layer and with the exact-f32 kernels, and
// the Winograd form becomes the default AT THAT SIZE only if the largest difference stays below WINO_MAX_ERR of the
// largest output — half of the tolerance the golden vectors are held to.  (qgx_generator_wino_info_n reports the decision
// per size; option "wino" = 0 / 1 overrides it, 2 restores it.)  The two tile shapes a size may run in (full / half
// height) give bit-identical results, as do k_convw and k_convw2: one evaluation per size covers them.
static constexpr float WINO_MAX_ERR = 1e-5f;
static int calibrate_wino(qgx_generator *g) {
    for (int i = 0; i < 5; ++i) { g->auto_wino_n[i] = 0; g->wino_err_n[i] = 0.f; }
    g->opt_wino = 0;
    if (g->opt_precision != 3) return QGX_OK;
    static const int SIZES[5] = {32, 48, 64, 96, 128};
    unsigned *cd = nullptr;
    QGX_HIP(hipMalloc((void **)&cd, 2 * sizeof(unsigned)));
    const int saved_part = g->opt_part_max_tiles, saved_min = g->opt_wino_min_tiles;
    int rc = QGX_OK;
    for (int si = 0; si < 5 && !rc; ++si) {
        const int N = SIZES[si], npix = N * N;
        // as many members as keep the workspace of the calibration at what eight 64 x 64 members need (2 at 128 x 128)
        const int B = N <= 64 ? 8 : (N == 96 ? 4 : 2);
        // member patterns: 0-2 white noise, 3 white x 2, 4 white x 1.5, 5-6 smooth large-scale fields + noise, 7 constants;
        // with fewer than eight members the hot, the smooth and the constant ones come first
        static const int PAT8[8] = {0, 1, 2, 3, 4, 5, 6, 7}, PAT4[4] = {3, 5, 7, 0}, PAT2[2] = {3, 5};
        const int *pat = B == 8 ? PAT8 : (B == 4 ? PAT4 : PAT2);
        rc = reserve(g, B, N);
        if (rc) break;
        float worst = 0.f;
        for (int n = 0; n < g->n_nets && !rc; ++n) {
            const NetHost &net = g->nets[n];
            if (!net.L[1].ww[0]) { worst = INFINITY; break; }
            std::vector<float> x((size_t)B * net.n_in * npix);
            uint32_t lcg = 54321u + 977u * n;
            auto uni = [&]() { lcg = lcg * 1664525u + 1013904223u; return ((lcg >> 8) + 0.5f) * (1.0f / 16777216.0f); };
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < net.n_in; ++c)
                    for (int y = 0; y < N; ++y)
                        for (int xx = 0; xx < N; ++xx) {
                            float v;
                            const int pb = pat[b];
                            const float white = sqrtf(-2.f * logf(uni())) * cosf(6.2831853f * uni());
                            if (pb < 5) v = white * (pb == 3 ? 2.f : (pb == 4 ? 1.5f : 1.f));
                            else if (pb < 7) v = 2.f * sinf(6.2831853f * (y * (c + 1) + xx * (pb - 4)) / N)
                                                 + cosf(6.2831853f * 2 * xx / N) + 0.3f * white;
                            else v = 3.f * ((c & 1) ? -1.f : 1.f);
                            x[(((size_t)b * net.n_in + c) * N + y) * N + xx] = v;
                        }
            QGX_HIP(hipMemcpy(g->X, x.data(), x.size() * sizeof(float), hipMemcpyHostToDevice));
            g->opt_precision = 0;
            rc = cnn_forward(g, net, g->X, g->Y0, B, N, nullptr);
            g->opt_precision = 3; g->opt_wino = 1; g->opt_wino_min_tiles = 1; g->opt_part_max_tiles = 0;
            if (!rc) rc = cnn_forward(g, net, g->X, g->Y1, B, N, nullptr);
            g->opt_wino = 0; g->opt_wino_min_tiles = saved_min; g->opt_part_max_tiles = saved_part;
            if (rc) break;
            QGX_HIP(hipMemset(cd, 0, 2 * sizeof(unsigned)));
            hipLaunchKernelGGL(k_absdiff_max, dim3(64), dim3(256), 0, nullptr, (const float *)g->Y1, (const float *)g->Y0,
                               (size_t)B * net.n_out * npix, cd);
            float h[2];
            QGX_HIP(hipMemcpy(h, cd, sizeof(h), hipMemcpyDeviceToHost));
            const float err = h[1] > 0.f ? h[0] / h[1] : INFINITY;
            worst = fmaxf(worst, std::isfinite(err) ? err : INFINITY);
        }
        g->wino_err_n[si] = worst;
        g->auto_wino_n[si] = !rc && worst <= WINO_MAX_ERR ? 1 : 0;
    }
    (void)hipFree(cd);
    QGX_HIP(hipMemset(g->range_dev, 0, 2 * sizeof(unsigned)));     // the calibration runs are not the caller's
    if (rc) return rc;
    g->opt_wino = 2;
    return QGX_OK;
}

static int calibrate(qgx_generator *g) {
    QGX_HIP(hipMalloc((void **)&g->range_dev, 2 * sizeof(unsigned)));
    QGX_HIP(hipMemset(g->range_dev, 0, 2 * sizeof(unsigned)));
    const int N = 32, B = 8, npix = N * N;
    int rc = reserve(g, B, N);
    if (rc) return rc;
    unsigned *cd = nullptr;
    QGX_HIP(hipMalloc((void **)&cd, 20 * sizeof(unsigned)));
    QGX_HIP(hipMemset(cd, 0, 20 * sizeof(unsigned)));
    const int saved_precision = g->opt_precision;
    g->opt_precision = 0;
    for (int n = 0; n < g->n_nets && !rc; ++n) {
        const NetHost &net = g->nets[n];
        std::vector<float> x((size_t)B * net.n_in * npix);
        uint32_t lcg = 12345u + 977u * n;
        auto uni = [&]() { lcg = lcg * 1664525u + 1013904223u; return ((lcg >> 8) + 0.5f) * (1.0f / 16777216.0f); };
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < net.n_in; ++c)
                for (int y = 0; y < N; ++y)
                    for (int xx = 0; xx < N; ++xx) {
                        float v;
                        const float white = sqrtf(-2.f * logf(uni())) * cosf(6.2831853f * uni());
                        if (b < 4) v = white * (b == 3 ? 2.f : 1.f);                      // white noise, one member hotter
                        else if (b < 6) v = 2.f * sinf(6.2831853f * (y * (c + 1) + xx * (b - 3)) / N)
                                            + cosf(6.2831853f * 2 * xx / N) + 0.3f * white;   // smooth, amplitude ~3
                        else v = (b == 6 ? 3.f : -3.f) * ((c & 1) ? -1.f : 1.f);         // constants
                        x[(((size_t)b * net.n_in + c) * N + y) * N + xx] = v;
                    }
        QGX_HIP(hipMemcpy(g->X, x.data(), x.size() * sizeof(float), hipMemcpyHostToDevice));
        g->calib_dev = cd;
        rc = cnn_forward(g, net, g->X, g->Y0, B, N, nullptr);
        g->calib_dev = nullptr;
        if (rc) break;
        // layer 1 with an identity BatchNorm = what the folded f16x3 variant stores
        LayerHost L0 = net.L[0];
        L0.scale = L0.ones; L0.shift = L0.zeros;
        rc = net.n_in == 4 ? launch_conv<4, 128, 5, 4, true, false, 2>(g, 0, L0, g->X, g->actA, B, N, 128, nullptr)
                           : launch_conv<2, 128, 5, 2, true, false, 2>(g, 0, L0, g->X, g->actA, B, N, 128, nullptr);
        if (rc) break;
        hipLaunchKernelGGL(k_absmax, dim3(256), dim3(256), 0, nullptr, (const float *)g->actA, (size_t)B * npix * 128, cd + 16);
    }
    g->opt_precision = saved_precision;
    unsigned h[20];
    if (!rc) {
        if (hipMemcpy(h, cd, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) rc = QGX_ERR_HIP;
    }
    (void)hipFree(cd);
    if (rc) return rc;
    for (int l = 0; l < 10; ++l) memcpy(&g->calib_max[l], &h[2 * l + 1], sizeof(float));
    auto window = [&](bool with_fold, float &lo, float &hi) {
        lo = INFINITY; hi = 0.f;
        for (int l = with_fold ? 1 : 0; l < 7; ++l)
            if (g->calib_max[l] > 0.f) { lo = fminf(lo, g->calib_max[l]); hi = fmaxf(hi, g->calib_max[l]); }
        if (with_fold && g->calib_max[8] > 0.f) { lo = fminf(lo, g->calib_max[8]); hi = fmaxf(hi, g->calib_max[8]); }
    };
    const float TOP = 1024.f, BOT = 0.25f;                 // window of s * (layer maximum)
    float lo, hi;
    int fold = 1;
    window(true, lo, hi);
    if (!(hi > 0.f) || !std::isfinite(hi) || hi / lo > TOP / BOT) { fold = 0; window(false, lo, hi); }
    int precision = 3, e = 0;
    if (!(hi > 0.f) || !std::isfinite(hi) || hi / lo > TOP / BOT) precision = 0;
    else if (hi > TOP || lo < BOT) {
        int eh = 0;
        (void)frexpf(hi, &eh);                             // hi = m 2^eh, m in [0.5, 1): hi * 2^(10 - eh) in [512, 1024)
        e = 10 - eh;
    }
    g->auto_precision = precision; g->auto_fold = fold; g->auto_ascale_log2 = e;
    g->opt_precision = precision; g->opt_fold = fold; g->opt_ascale = ldexpf(1.f, e);
    return QGX_OK;
}

bool generator_noise_is_double(const qgx_generator *g) { return g->kind == QGX_GEN_GZ; }

int generator_input_info(qgx_generator *g, int B, int N, GenFuse *gf) {
    QGX_REQUIRE(g && gf && g->kind != QGX_GEN_GZ, "generator_input_info: bad argument");
    int rc = reserve(g, B, N);
    if (rc) return rc;
    gf->X = g->X; gf->xs[0] = g->x_std[0]; gf->xs[1] = g->x_std[1]; gf->range = g->range_dev;
    return QGX_OK;
}

int generator_forward(qgx_generator *g, const double *q, const void *z, double *S, int B, int N,
                      int demean, hipStream_t st, const NoiseUpdate *nu, GenFuse *defer, bool input_ready) {
    QGX_REQUIRE(g && q && z && S && B > 0, "generator_forward: bad argument");
    int rc = reserve(g, B, N);
    if (rc) return rc;
    const int npix = N * N;
    QGX_REQUIRE(npix % 4 == 0, "generator_forward: N*N must be a multiple of 4");
    dim3 pg((npix + 255) / 256, B), pb(256);
    if (g->kind == QGX_GEN_GZ) {
        if (nu && (rc = noise_update(const_cast<void *>(z), nu->xi_ext, true, B, 2 * npix, nu->seed, nu->member_offset,
                                     nu->step, nu->a, nu->b, st))) return rc;
        hipLaunchKernelGGL(k_prep_input, pg, pb, 0, st, q, (const float *)nullptr, g->X, 2, npix, g->x_std[0], g->x_std[1], g->range_dev);
        if ((rc = cnn_forward(g, g->nets[0], g->X, g->Y0, B, N, st))) return rc;
        if ((rc = cnn_forward(g, g->nets[1], g->X, g->Y1, B, N, st))) return rc;
        hipLaunchKernelGGL(k_finish<FIN_GZ>, dim3(2 * B), dim3(1024), 0, st, (const float *)g->Y0, (const float *)g->Y1,
                           (const double *)z, S, npix, g->y_std[0], g->y_std[1], demean, g->range_dev);
    } else {
        if (input_ready) {
            // the previous step kernel wrote X and z (GenFuse::X)
        } else if (nu) {
            dim3 qg((2 * npix / 4 + 255) / 256, B);
            hipLaunchKernelGGL(k_prep_noise, qg, pb, 0, st, q, (float *)const_cast<void *>(z), (const float *)nu->xi_ext,
                               g->X, npix, g->x_std[0], g->x_std[1], nu->seed, nu->member_offset, nu->step,
                               (float)nu->a, (float)nu->b, g->range_dev);
        } else {
            hipLaunchKernelGGL(k_prep_input, pg, pb, 0, st, q, (const float *)z, g->X, 4, npix, g->x_std[0], g->x_std[1], g->range_dev);
        }
        const bool regression = g->n_nets == 2;     // regression != 'None': Y += net_mean(X) on the normalised PV alone
        if (regression) {
            float *X2 = g->X + (size_t)B * 4 * npix;
            hipLaunchKernelGGL(k_take2, dim3((2 * npix / 4 + 255) / 256, B), pb, 0, st, (const float *)g->X, X2, 2 * npix);
            if ((rc = cnn_forward(g, g->nets[1], X2, g->Y1, B, N, st))) return rc;
        }
        if ((rc = cnn_forward(g, g->nets[0], g->X, g->Y0, B, N, st))) return rc;
        if (defer) {
            defer->y = g->Y0; defer->y1 = regression ? g->Y1 : nullptr;
            defer->ys[0] = g->y_std[0]; defer->ys[1] = g->y_std[1]; defer->demean = demean;
            defer->range = g->range_dev;
        } else if (regression) {
            hipLaunchKernelGGL(k_finish<FIN_SUM>, dim3(2 * B), dim3(1024), 0, st, (const float *)g->Y0, (const float *)g->Y1,
                               (const double *)nullptr, S, npix, g->y_std[0], g->y_std[1], demean, g->range_dev);
        } else {
            hipLaunchKernelGGL(k_finish<FIN_PLAIN>, dim3(2 * B), dim3(1024), 0, st, (const float *)g->Y0, (const float *)nullptr,
                               (const double *)nullptr, S, npix, g->y_std[0], g->y_std[1], demean, g->range_dev);
        }
    }
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

}  // namespace qgx

using namespace qgx;

extern "C" int qgx_generator_create(int kind, const qgx_cnn_weights *nets, int n_nets, const float x_std[2],
                                    const float y_std[2], int device, qgx_generator **out) {
    QGX_REQUIRE(nets && out && x_std && y_std, "qgx_generator_create: null argument");
    QGX_REQUIRE(kind == QGX_GEN_GAN || kind == QGX_GEN_VAE || kind == QGX_GEN_GZ, "unknown generator kind %d", kind);
    // GAN / VAE: the generator or decoder, and optionally (regression != 'None', cgan_regression.py:59-60) the 2-channel net_mean
    QGX_REQUIRE(kind == QGX_GEN_GZ ? n_nets == 2 : (n_nets == 1 || n_nets == 2), "generator kind %d needs %s nets, not %d", kind,
                kind == QGX_GEN_GZ ? "2" : "1 or 2", n_nets);
    QGX_HIP(hipSetDevice(device));
    qgx_generator *g = new (std::nothrow) qgx_generator();
    if (!g) { set_error("out of host memory"); return QGX_ERR_NOMEM; }
    g->kind = kind; g->device = device; g->n_nets = n_nets;
    for (int i = 0; i < 2; ++i) { g->x_std[i] = x_std[i]; g->y_std[i] = y_std[i]; }
    for (int n = 0; n < n_nets; ++n) {
        const qgx_cnn_weights *w = &nets[n];
        const int want_in = kind == QGX_GEN_GZ || n == 1 ? 2 : 4;
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
    {
        int rc = calibrate(g);
        if (!rc) rc = calibrate_wino(g);
        if (rc) { qgx_generator_destroy(g); return rc; }
    }
    *out = g;
    return QGX_OK;
}

extern "C" int qgx_generator_range_read(qgx_generator *g, unsigned *flags, float *input_absmax, void *stream) {
    QGX_REQUIRE(g && flags && input_absmax, "qgx_generator_range_read: null argument");
    unsigned h[2] = {0, 0};
    QGX_HIP(hipMemcpyAsync(h, g->range_dev, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    QGX_HIP(hipMemsetAsync(g->range_dev, 0, sizeof(h), (hipStream_t)stream));
    QGX_HIP(hipStreamSynchronize((hipStream_t)stream));
    *flags = h[0];
    memcpy(input_absmax, &h[1], sizeof(float));
    return QGX_OK;
}

extern "C" int qgx_generator_wino_info_n(const qgx_generator *g, int N, int *enabled, int *chosen_by_calibration, float *calibration_error) {
    QGX_REQUIRE(g, "qgx_generator_wino_info_n: null generator");
    const int si = wino_size_index(N);
    if (enabled) *enabled = si >= 0 && (g->opt_wino == 1 || (g->opt_wino == 2 && g->auto_wino_n[si]));
    if (chosen_by_calibration) *chosen_by_calibration = si >= 0 ? g->auto_wino_n[si] : 0;
    if (calibration_error) *calibration_error = si >= 0 ? g->wino_err_n[si] : INFINITY;
    return QGX_OK;
}

// which kernel the 5x5 layer (layer 2) of net `inet` takes for an ensemble of B members at N x N with the options in
// force: 0 exact-f32 MFMA, 1 the 25-tap f16x3 kernel, 2 its split-K form (tiny ensembles), 3 1-D Winograd (k_convw),
// 4 1-D Winograd with the transform under the MFMAs (k_convw2).  Mirrors cnn_forward_half / launch_convw.
extern "C" int qgx_generator_layer2_kernel(const qgx_generator *g, int inet, int B, int N, int *kernel) {
    QGX_REQUIRE(g && kernel, "qgx_generator_layer2_kernel: null argument");
    QGX_REQUIRE(inet >= 0 && inet < g->n_nets, "qgx_generator_layer2_kernel: net %d of %d", inet, g->n_nets);
    const NetHost &net = g->nets[inet];
    *kernel = 0;
    if (g->opt_precision != 3 || rows_h2(N) <= 0) return QGX_OK;
    const bool fold = g->opt_first_h && g->opt_fold && net.L[1].whF;
    const int r2 = rows_h2(N);
    const bool tiny = g->opt_h2 == 3 && r2 > 0 && B * (N / r2) < g->opt_part_max_tiles;
    if (tiny) { *kernel = 2; return QGX_OK; }
    *kernel = 1;
    if (g->opt_first_h && wino_applies(g, net.L[1], fold ? 1 : 0, B, N)) {
        *kernel = 3;
        if (g->opt_wino2 && !wino_planar(g) && g->opt_wino_exp == 0) {
            bool rows_full = true;       // the full-height shape of launch_convw (k_convw2 exists for it at 64 x 64 and as 12 rows at 96 x 96)
            if (N == 64) { const int r8 = (B * 8 + 255) / 256, r4 = (B * 16 + 255) / 256; rows_full = !(g->opt_wino_rows64 == 4 || (g->opt_wino_rows64 == 0 && 0.66 * r4 < 1.0 * r8)); if (rows_full) *kernel = 4; }
            else if (N == 96) { const int r16 = (B * 18 + 255) / 256, r12 = (B * 24 + 255) / 256; if (g->opt_wino_rows96 == 12 || (g->opt_wino_rows96 == 0 && 0.79 * r12 < 1.0 * r16)) *kernel = 4; }
        }
    }
    return QGX_OK;
}

extern "C" int qgx_generator_wino_info(const qgx_generator *g, int *enabled, int *chosen_by_calibration, float *calibration_error) {
    QGX_REQUIRE(g, "qgx_generator_wino_info: null generator");
    // the 64 x 64 grid (the headline workload); qgx_generator_wino_info_n reports every size
    if (enabled) *enabled = g->opt_wino == 1 || (g->opt_wino == 2 && g->auto_wino_n[2]);
    if (chosen_by_calibration) *chosen_by_calibration = g->auto_wino_n[2];
    if (calibration_error) *calibration_error = g->wino_err_n[2];
    return QGX_OK;
}

extern "C" int qgx_generator_info(const qgx_generator *g, int *precision, int *ascale_log2, int *fold, float *layer_absmax) {
    QGX_REQUIRE(g, "qgx_generator_info: null generator");
    if (precision) *precision = g->opt_precision;
    if (ascale_log2) { int e = 0; (void)frexpf(g->opt_ascale, &e); *ascale_log2 = e - 1; }
    if (fold) *fold = g->opt_fold;
    if (layer_absmax) memcpy(layer_absmax, g->calib_max, sizeof(g->calib_max));
    return QGX_OK;
}

extern "C" int qgx_generator_destroy(qgx_generator *g) {
    if (!g) return QGX_OK;
    (void)hipSetDevice(g->device);
    for (int n = 0; n < 2; ++n)
        for (int li = 0; li < 8; ++li) {
            LayerHost &L = g->nets[n].L[li];
            float *ptrs[] = {L.w, L.w32, L.wl16, L.wl8, L.bias, L.scale, L.shift};
            for (float *p : ptrs) if (p) (void)hipFree(p);
            for (void *p : L.wh) if (p) (void)hipFree(p);
            if (L.whf) (void)hipFree(L.whf);
            if (L.wh16) (void)hipFree(L.wh16);
            if (L.whF) (void)hipFree(L.whF);
            for (void *pw_ : L.ww) if (pw_) (void)hipFree(pw_);
            if (L.wh16F) (void)hipFree(L.wh16F);
            for (float *p : {L.biasF, L.ones, L.zeros}) if (p) (void)hipFree(p);
        }
    float *bufs[] = {g->actA, g->actB, g->X, g->Y0, g->Y1, g->part,
                     g->ws_other.actA, g->ws_other.actB, g->ws_other.X, g->ws_other.Y0, g->ws_other.Y1, g->ws_other.part};
    for (float *p : bufs) if (p) (void)hipFree(p);
    if (g->range_dev) (void)hipFree(g->range_dev);
    for (hipEvent_t e : g->prof_ev) (void)hipEventDestroy(e);
    delete g;
    return QGX_OK;
}

extern "C" int qgx_generator_forward(qgx_generator *g, const double *q_dev, const void *z_dev, double *S_dev,
                                     int B, int N, int demean, void *stream) {
    return generator_forward(g, q_dev, z_dev, S_dev, B, N, demean, (hipStream_t)stream, nullptr);
}

extern "C" int qgx_cnn_forward(qgx_generator *g, int inet, const float *x_dev, float *y_dev, int B, int N,
                               void *stream) {
    QGX_REQUIRE(g && x_dev && y_dev && inet >= 0 && inet < g->n_nets && B > 0, "qgx_cnn_forward: bad argument");
    int rc = reserve(g, B, N);
    if (rc) return rc;
    hipLaunchKernelGGL(k_absmax, dim3(256), dim3(256), 0, (hipStream_t)stream, x_dev,
                       (size_t)B * g->nets[inet].n_in * N * N, g->range_dev);
    return cnn_forward(g, g->nets[inet], x_dev, y_dev, B, N, (hipStream_t)stream);
}

extern "C" int qgx_generator_profile(qgx_generator *g, int layer) {
    QGX_REQUIRE(g && layer >= -1 && layer < 8, "qgx_generator_profile: bad argument");
    g->prof_layer = layer;
    g->prof_used = 0;
    g->prof_seen = 0;
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

extern "C" int qgx_generator_set_option(qgx_generator *g, const char *name, int value) {
    QGX_REQUIRE(g && name, "qgx_generator_set_option: null argument");
    if (!strcmp(name, "chunk")) { QGX_REQUIRE(value == 16 || value == 32, "chunk must be 16 or 32"); g->opt_cc = value; }
    else if (!strcmp(name, "last_valu")) g->opt_last_valu = value ? 1 : 0;
    else if (!strcmp(name, "small")) g->opt_small = value ? 1 : 0;
    else if (!strcmp(name, "v3")) g->opt_v3 = value;   // -1 auto, 0 off, 1 = slice per tap row, 2 = per chunk
#ifndef QGX_AB
    else if (!strcmp(name, "res") || !strcmp(name, "h3") || !strcmp(name, "h4") || !strcmp(name, "half_nw") ||
             !strcmp(name, "h2") || (!strcmp(name, "precision") && value == 1))
        QGX_REQUIRE(false, "generator option '%s'=%d selects a kernel of the A/B library only (make ab, QGX_LIB=libqgx_ab.so)", name, value);
#endif
    else if (!strcmp(name, "precision")) { QGX_REQUIRE(value == 0 || value == 1 || value == 3, "precision must be 0 (f32), 1 (f16) or 3 (f16x3)"); g->opt_precision = value; }
    else if (!strcmp(name, "member_chunk")) { QGX_REQUIRE(value >= 0, "member_chunk must be >= 0"); g->opt_member_chunk = value; }
    else if (!strcmp(name, "res")) g->opt_res = value ? 1 : 0;
    else if (!strcmp(name, "h2")) g->opt_h2 = value & 3;
    else if (!strcmp(name, "pair")) g->opt_pair = value ? 1 : 0;
    else if (!strcmp(name, "fuse")) g->opt_fuse = value & 7;
    else if (!strcmp(name, "fuse96")) g->opt_fuse96 = value & 3;
    else if (!strcmp(name, "small_tiles")) g->opt_small_tiles = value ? 1 : 0;
    else if (!strcmp(name, "tiny_pairs")) { QGX_REQUIRE(value >= 0 && value <= 7, "tiny_pairs: bits 0 (layers 7, 8), 1 (layers 5, 6), 2 (layers 3, 4)"); g->opt_tiny_pairs = value; }
#ifdef QGX_AB
    else if (!strcmp(name, "pair_lp")) g->opt_pair_lp = value;
#endif
    else if (!strcmp(name, "h3")) g->opt_h3 = value ? 1 : 0;
    else if (!strcmp(name, "last_rows")) g->opt_last_rows = value;
    else if (!strcmp(name, "part_max_tiles")) g->opt_part_max_tiles = value;
    else if (!strcmp(name, "fold")) g->opt_fold = value ? 1 : 0;
    else if (!strcmp(name, "wino")) { QGX_REQUIRE(value >= 0 && value <= 2, "wino must be 0 (never), 1 (every specialised grid) or 2 (per grid size, as calibrated)"); g->opt_wino = value; }
    else if (!strcmp(name, "wino2")) g->opt_wino2 = value ? 1 : 0;
#ifdef QGX_AB
    else if (!strcmp(name, "wino_exp")) g->opt_wino_exp = value;
    else if (!strcmp(name, "wino_pl")) g->opt_wino_pl = value;
    else if (!strcmp(name, "stop_layer")) g->opt_stop_layer = value;
#endif
    else if (!strcmp(name, "h2_rows96")) { QGX_REQUIRE(value == 0 || value == 12 || value == 16, "h2_rows96 must be 0 (auto), 12 or 16"); g->opt_h2_rows96 = value; }
    else if (!strcmp(name, "wino_rows64")) { QGX_REQUIRE(value == 0 || value == 4 || value == 8, "wino_rows64 must be 0 (auto), 4 or 8"); g->opt_wino_rows64 = value; }
    else if (!strcmp(name, "wino_rows96")) { QGX_REQUIRE(value == 0 || value == 12 || value == 16, "wino_rows96 must be 0 (auto), 12 or 16"); g->opt_wino_rows96 = value; }
    else if (!strcmp(name, "wino_min_tiles")) { QGX_REQUIRE(value >= 1, "wino_min_tiles must be >= 1"); g->opt_wino_min_tiles = value; }
    else if (!strcmp(name, "h2_grid")) g->opt_h2_grid = value;
    else if (!strcmp(name, "h4")) g->opt_h4 = value;
    else if (!strcmp(name, "prio_alt")) g->opt_prio_alt = value;
    else if (!strcmp(name, "h2_w8")) g->opt_h2_w8 = value & 3;
    else if (!strcmp(name, "h2_tw32")) g->opt_h2_tw32 = value ? 1 : 0;
    else if (!strcmp(name, "h2_x96")) g->opt_h2_x96 = value ? 1 : 0;
    else if (!strcmp(name, "h2_w8_min96")) g->opt_h2_w8_min96 = value;
    else if (!strcmp(name, "half_min_tiles")) { QGX_REQUIRE(value >= 1, "half_min_tiles must be >= 1"); g->opt_half_min_tiles = value; }
    else if (!strcmp(name, "first_h")) g->opt_first_h = value ? 1 : 0;
    else if (!strcmp(name, "half_nw")) { QGX_REQUIRE(value == 4 || value == 8, "half_nw must be 4 or 8"); g->opt_half_nw = value; }
    else if (!strcmp(name, "ascale_log2")) { QGX_REQUIRE(value >= -24 && value <= 24, "ascale_log2 must be in -24..24"); g->opt_ascale = ldexpf(1.f, value); }
    else if (!strcmp(name, "prof_every")) { QGX_REQUIRE(value >= 1, "prof_every must be >= 1"); g->prof_every = value; }
    else if (!strcmp(name, "auto")) { g->opt_precision = g->auto_precision; g->opt_fold = g->auto_fold; g->opt_ascale = ldexpf(1.f, g->auto_ascale_log2); g->opt_wino = 2; }
    else if (!strcmp(name, "first_split")) { QGX_REQUIRE(value == 1 || value == 2 || value == 4, "first_split must be 1, 2 or 4"); g->opt_first_split = value; }
    else QGX_REQUIRE(false, "unknown generator option '%s'", name);
    return QGX_OK;
}

#ifdef QGX_AB
// A/B library, debugging: copy the first nbytes of activation buffer `which` (0: the odd layers' outputs, 1: the even layers')
extern "C" int qgx_debug_read_act(qgx_generator *g, int which, void *dst_dev, size_t nbytes, void *stream) {
    QGX_REQUIRE(g && dst_dev, "qgx_debug_read_act: null argument");
    QGX_HIP(hipMemcpyAsync(dst_dev, which ? (const void *)g->actB : (const void *)g->actA, nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return QGX_OK;
}
#endif
#ifdef QGX_STAMPS
// diagnostic builds only (bench_tools/conv_stamps.py): where the s_memtime trace of k_convh_res goes
extern "C" int qgx_debug_set_stamps(qgx_generator *g, void *buf, int layer) {
    g->stamps = (unsigned long long *)buf;
    g->stamp_layer = layer;
    return QGX_OK;
}
#endif

extern "C" int qgx_moments_accumulate(const float *y_dev, double *sum_dev, double *sumsq_dev, size_t n, void *stream) {
    QGX_REQUIRE(y_dev && sum_dev && sumsq_dev && n > 0, "qgx_moments_accumulate: bad argument");
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_moments, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       y_dev, sum_dev, sumsq_dev, n);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}
