// The 128 -> 64, 5x5 layer (75 % of the generator's FLOPs) on v_mfma_f32_16x16x32_f16, f16x3 arithmetic.
// Included by conv.hip after conv_half.hpp.
//
// Why another shape: the layer is bound by the matrix cores, and on this chip a dense MFMA stream is
// clock-limited by power management, not by issue slots.  With identical operand traffic a bare loop of
// 16x16x32 MFMAs sustains 1.15-1.17x the FLOP/s of the 32x32x16 loop (bench_tools/mfma_peak_f16.hip:
// 1.61 vs 1.37 PFLOP/s on random data, shader clock 1.91 vs 1.60 GHz).
//
// K = 32 of one MFMA = the four channel octets of a 32-channel chunk at ONE tap: lane quarter q = lane/16
// supplies octet q, so the chunk is a pixel's full 128-byte line (one HBM fetch per line, where the
// 16-channel chunks of k_convh2 fetched its halves in different iterations).
//   LDS patch: [octet plane][pixel][hi | lo | pad] with 48-byte pixels and a plane size that is a multiple
//              of 256 B -> every ds_read_b128 phase is conflict-free; wrapped x-halo -> taps are
//              compile-time byte offsets from one base address per pixel group
//   weights:   [tap][part][octet][cout][8 f16], slice = one tap row (5 taps, 40 KB), single buffer,
//              staged through registers one stage ahead
//   wave tile: 64 pixels x 64 output channels = 4 x 4 accumulator tiles of 16 x 16
// 4 waves, one workgroup per CU, one wave per SIMD (the 512-register budget pays for full double
// buffering of both operand streams and for the 17 + 10 prefetch registers).
//
// RESULT (in-process A/B, bench_tools/ab_conv.py, option "h3"): 576-643 us against 553-584 us for the
// 32x32x16 kernel k_convh2 on the same device (537 against 510 us after both got the spread prefetch loads
// and the BatchNorm fold) — the microbenchmark's advantage does not survive in the
// full kernel (both sit at the same power-limited rate), so this kernel is OFF by default and kept as
// the record of that experiment.
#pragma once

typedef float f32x4a __attribute__((ext_vector_type(4)));

template <int NN>
__global__ __launch_bounds__(256) void k_convh3(ConvHArgs a, int total_tiles) {
    constexpr int CIN = 128, COUT = 64, KS = 5, P = 2, T = 25, TPS = 5, NSL = 5;
    constexpr int NW = 4, NTHR = 256;
    constexpr int NCH = CIN / 32;
    constexpr int PIXB = CIN * 4, OPIXB = COUT * 4;
    constexpr int N = NN, R = NW * 64 / NN, PR = R + KS - 1, PW = NN + 2 * P;
    constexpr int PSTR = 48, PLANE = PR * PW * PSTR;
    constexpr int TAPB = 2 * 4 * COUT * 16, WSB = TPS * TAPB;
    constexpr int PU = PR * PW * 8, PPT = (PU + NTHR - 1) / NTHR;
    constexpr int WU = WSB / 16, WPT = (WU + NTHR - 1) / NTHR;
    static_assert((NW * 64) % NN == 0 && NN % R == 0 && NN % 16 == 0 && PLANE % 256 == 0, "shape");
    char *const lds0 = conv_smem;
    char *const wlds0 = lds0 + 4 * PLANE;
    float *const ep = reinterpret_cast<float *>(wlds0 + WSB);
    const char *const inb = reinterpret_cast<const char *>(a.in);
    const char *const wb = reinterpret_cast<const char *>(a.w);
    constexpr int tiles_per_img = N / R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lc = lane & 15, q = lane >> 4;
    const int n_my = blockIdx.x < (unsigned)total_tiles ? (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (n_my == 0) return;

#define QGX_H3P_LOAD(TI, CH, V)                                                                             \
    {                                                                                                       \
        const int tile_ = blockIdx.x + (TI) * gridDim.x;                                                    \
        const int b_ = tile_ / tiles_per_img;                                                               \
        const int y0_ = (tile_ - b_ * tiles_per_img) * R;                                                   \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            int it_ = u * NTHR + threadIdx.x;                                                                \
            it_ = it_ < PU ? it_ : PU - 1;                                                                  \
            const int un_ = it_ & 7, pl_ = it_ >> 3;                                                        \
            const int pr_ = pl_ / PW, xx_ = pl_ - pr_ * PW;                                                 \
            int gy_ = y0_ - P + pr_, gx_ = xx_ - P;                                                         \
            gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                           \
            gx_ = gx_ < 0 ? gx_ + N : (gx_ >= N ? gx_ - N : gx_);                                           \
            V[u] = *reinterpret_cast<const f32x4 *>(                                                        \
                inb + (((size_t)b_ * N + gy_) * N + gx_) * PIXB + (CH) * 128 + un_ * 16);                   \
        }                                                                                                   \
    }
#define QGX_H3P_LOAD1(TI, CH, V, U)                                                                         \
    {                                                                                                       \
        const int tile_ = blockIdx.x + (TI) * gridDim.x;                                                    \
        const int b_ = tile_ / tiles_per_img;                                                               \
        const int y0_ = (tile_ - b_ * tiles_per_img) * R;                                                   \
        int it_ = (U) * NTHR + threadIdx.x;                                                                  \
        it_ = it_ < PU ? it_ : PU - 1;                                                                      \
        const int un_ = it_ & 7, pl_ = it_ >> 3;                                                            \
        const int pr_ = pl_ / PW, xx_ = pl_ - pr_ * PW;                                                     \
        int gy_ = y0_ - P + pr_, gx_ = xx_ - P;                                                             \
        gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                               \
        gx_ = gx_ < 0 ? gx_ + N : (gx_ >= N ? gx_ - N : gx_);                                               \
        V[U] = *reinterpret_cast<const f32x4 *>(                                                            \
            inb + (((size_t)b_ * N + gy_) * N + gx_) * PIXB + (CH) * 128 + un_ * 16);                       \
    }
#define QGX_H3W_LOAD1(CH, SL, V, U)                                                                         \
    {                                                                                                       \
        const f32x4 *src_ = reinterpret_cast<const f32x4 *>(wb + ((size_t)(CH) * NSL + (SL)) * WSB);        \
        const int it_ = (U) * NTHR + threadIdx.x;                                                            \
        V[U] = src_[it_ < WU ? it_ : WU - 1];                                                               \
    }
#define QGX_H3P_STORE(V)                                                                                    \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            if (it_ < PU)                                                                                   \
                *reinterpret_cast<f32x4 *>(lds0 + ((it_ & 7) >> 1) * PLANE + (it_ >> 3) * PSTR + (it_ & 1) * 16) = V[u]; \
        }                                                                                                   \
    }
#define QGX_H3W_LOAD(CH, SL, V)                                                                             \
    {                                                                                                       \
        const f32x4 *src_ = reinterpret_cast<const f32x4 *>(wb + ((size_t)(CH) * NSL + (SL)) * WSB);        \
        _Pragma("unroll") for (int u = 0; u < WPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            V[u] = src_[it_ < WU ? it_ : WU - 1];                                                           \
        }                                                                                                   \
    }
#define QGX_H3W_STORE(V)                                                                                    \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < WPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            if (it_ < WU) *reinterpret_cast<f32x4 *>(wlds0 + it_ * 16) = V[u];                              \
        }                                                                                                   \
    }

    for (int i = threadIdx.x; i < 3 * COUT; i += NTHR)
        ep[i] = i < COUT ? a.bias[i] : (i < 2 * COUT ? a.scale[i - COUT] : a.shift[i - 2 * COUT]);
    {
        f32x4 pv[PPT], wv[WPT];
        QGX_H3P_LOAD(0, 0, pv)
        QGX_H3W_LOAD(0, 0, wv)
        QGX_H3P_STORE(pv)
        QGX_H3W_STORE(wv)
    }
    __syncthreads();

    // one base address per group of 16 pixels (octet plane q), one for the weights (octet q, channel lc)
    int pbase[4];
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
        const int p = wave * 64 + pg * 16 + lc;
        const int py = p / NN, px = p - py * NN;
        pbase[pg] = q * PLANE + (py * PW + px) * PSTR;
    }
    const char *const wl = wlds0 + q * (COUT * 16) + lc * 16;
    f32x4a acc[4][4];                                     // [cout group][pixel group]
    for (int ti = 0; ti < n_my; ++ti) {
        for (int ch = 0; ch < NCH; ++ch) {
            if (ch == 0) {
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                    for (int pg = 0; pg < 4; ++pg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[cg][pg][r] = 0.f;
            }
            const int nch = ch + 1 < NCH ? ch + 1 : 0;
            const int nti = ch + 1 < NCH ? ti : ti + 1;
            const bool have_next_chunk = nti < n_my;
            f32x4 pv[PPT];
            const int p_ti = have_next_chunk ? nti : ti, p_ch = have_next_chunk ? nch : ch;
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) {
                f32x4 wv[WPT];
                const bool last_stage = !have_next_chunk && sl == NSL - 1;
                // global prefetch loads spread over the first taps of the stage (see k_convh2)
                const int wch = sl + 1 < NSL ? ch : (have_next_chunk ? nch : ch);
                const int wsl = sl + 1 < NSL ? sl + 1 : (have_next_chunk ? 0 : sl);
                const int n_ld = WPT + (sl == 0 ? PPT : 0);
                h8 Pn[4][2], Wn[4][2];
#define QGX_H3_FRAGS(TL)                                                                                    \
                {                                                                                           \
                    const int tap_ = sl * TPS + (TL), ky_ = tap_ / KS, kx_ = tap_ - ky_ * KS;               \
                    _Pragma("unroll") for (int cg = 0; cg < 4; ++cg)                                        \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                       \
                            Wn[cg][j] = *reinterpret_cast<const h8 *>(wl + (TL) * TAPB + j * (4 * COUT * 16) + cg * 256); \
                    _Pragma("unroll") for (int pg = 0; pg < 4; ++pg)                                        \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                       \
                            Pn[pg][j] = *reinterpret_cast<const h8 *>(lds0 + pbase[pg] + (ky_ * PW + kx_) * PSTR + j * 16); \
                }
                QGX_H3_FRAGS(0)
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl) {
                    h8 Pc[4][2], Wc[4][2];
#pragma unroll
                    for (int g = 0; g < 4; ++g) { Pc[g][0] = Pn[g][0]; Pc[g][1] = Pn[g][1]; Wc[g][0] = Wn[g][0]; Wc[g][1] = Wn[g][1]; }
                    // the 16 fragment reads of the next tap go out in two groups of 8 with half of this tap's
                    // MFMAs between them: the LDS counter only counts to 15, so 16 reads in flight would turn
                    // every wait for the previous tap's data into a wait for the reads just issued
                    const int tln = tl + 1 < TPS ? tl + 1 : tl;
                    const int tap_ = sl * TPS + tln, ky_ = tap_ / KS, kx_ = tap_ - ky_ * KS;
                    if (tl + 1 < TPS) {
#pragma unroll
                        for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                Wn[cg][j] = *reinterpret_cast<const h8 *>(wl + tln * TAPB + j * (4 * COUT * 16) + cg * 256);
                    }
#pragma unroll
                    for (int i = 0; i < WPT + PPT; ++i) {
                        if (i < n_ld && (i * (TPS - 1)) / n_ld == tl) {
                            if (i < WPT) { QGX_H3W_LOAD1(wch, wsl, wv, i) }
                            else { QGX_H3P_LOAD1(p_ti, p_ch, pv, i - WPT) }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                        for (int pg = 0; pg < 4; ++pg) {
                            acc[cg][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc[cg][1], Pc[pg][0], acc[cg][pg], 0, 0, 0);
                            acc[cg][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc[cg][0], Pc[pg][1], acc[cg][pg], 0, 0, 0);
                            acc[cg][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc[cg][0], Pc[pg][0], acc[cg][pg], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    if (tl + 1 < TPS) {
#pragma unroll
                        for (int pg = 0; pg < 4; ++pg)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                Pn[pg][j] = *reinterpret_cast<const h8 *>(lds0 + pbase[pg] + (ky_ * PW + kx_) * PSTR + j * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int cg = 2; cg < 4; ++cg)
#pragma unroll
                        for (int pg = 0; pg < 4; ++pg) {
                            acc[cg][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc[cg][1], Pc[pg][0], acc[cg][pg], 0, 0, 0);
                            acc[cg][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc[cg][0], Pc[pg][1], acc[cg][pg], 0, 0, 0);
                            acc[cg][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wc[cg][0], Pc[pg][0], acc[cg][pg], 0, 0, 0);
                        }
                }
#undef QGX_H3_FRAGS
                // ---- retire the prefetches (single weight buffer: two barriers per stage)
                __syncthreads();
                if (!last_stage) QGX_H3W_STORE(wv)
                if (sl == NSL - 1 && have_next_chunk) QGX_H3P_STORE(pv)
                __syncthreads();
                if (sl == NSL - 1 && ch == NCH - 1) {
                    // ---- epilogue: lane = pixel, register r of acc[cg] = channel 16 cg + 4 q + r.  Lanes 16 apart
                    // hold the two halves of an octet: one v_permlane16_swap per register pair gives every lane
                    // a whole octet of cout group A (even quarters) or B (odd quarters)
                    const int tile_g = blockIdx.x + ti * gridDim.x;
                    const int b = tile_g / tiles_per_img;
                    const int y0 = (tile_g - b * tiles_per_img) * R;
                    char *ob = reinterpret_cast<char *>(a.out) + ((size_t)b * N * N + (size_t)y0 * N) * OPIXB;
#pragma unroll
                    for (int pg = 0; pg < 4; ++pg) {
                        char *pix = ob + (size_t)(wave * 64 + pg * 16 + lc) * OPIXB;
#pragma unroll
                        for (int cp = 0; cp < 2; ++cp) {
                            unsigned hw[2][2], lw[2][2];       // [cout group A/B][register pair]
#pragma unroll
                            for (int ab = 0; ab < 2; ++ab) {
                                const int cg = 2 * cp + ab;
                                const int c0 = cg * 16 + 4 * q;
                                const f32x4 bi = *reinterpret_cast<const f32x4 *>(ep + c0);
                                const f32x4 sc = *reinterpret_cast<const f32x4 *>(ep + COUT + c0);
                                const f32x4 sh = *reinterpret_cast<const f32x4 *>(ep + 2 * COUT + c0);
                                float hi[4], lo[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float x = (fmaxf(acc[cg][pg][e] * a.unscale + bi[e], 0.f) * sc[e] + sh[e]) * a.ascale;
                                    const _Float16 xh = (_Float16)x;
                                    hi[e] = (float)xh;
                                    lo[e] = x - hi[e];
                                    range_guard(fabsf(x), a.range, a.range_bit);
                                }
                                hw[ab][0] = pack_h2(hi[0], hi[1]); hw[ab][1] = pack_h2(hi[2], hi[3]);
                                lw[ab][0] = pack_h2(lo[0], lo[1]); lw[ab][1] = pack_h2(lo[2], lo[3]);
                            }
                            const auto h0 = __builtin_amdgcn_permlane16_swap(hw[0][0], hw[1][0], false, false);
                            const auto h1 = __builtin_amdgcn_permlane16_swap(hw[0][1], hw[1][1], false, false);
                            const auto l0 = __builtin_amdgcn_permlane16_swap(lw[0][0], lw[1][0], false, false);
                            const auto l1 = __builtin_amdgcn_permlane16_swap(lw[0][1], lw[1][1], false, false);
                            const int g = 2 * (2 * cp + (q & 1)) + (q >> 1);
                            const u32x4 oh = {h0[0], h1[0], h0[1], h1[1]};
                            const u32x4 ol = {l0[0], l1[0], l0[1], l1[1]};
                            *reinterpret_cast<u32x4 *>(pix + g * 32) = oh;
                            *reinterpret_cast<u32x4 *>(pix + g * 32 + 16) = ol;
                        }
                    }
                }
            }
        }
    }
#undef QGX_H3P_LOAD
#undef QGX_H3P_LOAD1
#undef QGX_H3W_LOAD1
#undef QGX_H3P_STORE
#undef QGX_H3W_LOAD
#undef QGX_H3W_STORE
}
