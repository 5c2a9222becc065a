// The 128 -> 64, 5x5 layer (f16x3) with FULL-LINE patch chunks: 8 waves, R = 8 rows per workgroup.
// Included by conv.hip after conv_half.hpp.
//
// k_convh2 stages the input patch per 16-channel chunk = 64 bytes per pixel = HALF a 128-byte line, and
// the other half is fetched five stages later: PMC FETCH_SIZE showed every line of this layer's input
// coming from HBM twice (1.2 GB per launch where the input, read once with the halo, is 0.54 GB), on top
// of the 2x row halo of its R = 4 tiles.  Here the patch chunk is 32 channels = one whole line per pixel
// (144-byte LDS pixels: conflict-free ds_read_b128, wrapped x-halo -> compile-time tap offsets) and a
// workgroup owns R = 8 rows (halo 1.5x): the weight slices stay per 16-channel chunk and tap row
// (double buffered), two sub-chunks x five slices = ten stages per patch chunk, and the next patch chunk
// is fetched into registers spread over the first eight of them.
//
// RESULT (in-process A/B, option "h4"): bit-identical output, HBM traffic of the layer halved, but 513-553 us
// against 488-517 us for k_convh2: the 13 (8 waves) / 26 (4 waves, MT = 4) prefetch registers of a 117 KB patch
// chunk push the ten-stage body into scratch.  OFF by default; the layer is MFMA-bound, so the extra traffic of
// k_convh2 (2.3 TB/s) costs nothing measurable.
#pragma once

template <int NN, int MT, int NW = 8>
__global__ __launch_bounds__(NW * 64) void k_convh4(ConvHArgs a, int total_tiles) {
    constexpr int CIN = 128, COUT = 64, KS = 5, P = 2, TPS = 5, NSL = 5;
    constexpr int NTHR = NW * 64, NT = COUT / 32;
    constexpr int NPC = CIN / 32;
    constexpr int PIXB = CIN * 4, OPIXB = COUT * 4;
    constexpr int N = NN, R = NW * MT * 32 / NN, PR = R + KS - 1, PW = NN + 2 * P;
    constexpr int PSTR = 144;
    constexpr int patch_bytes = PR * PW * PSTR;
    constexpr int TAPB = 4 * COUT * 16, WSB = TPS * TAPB;
    constexpr int PU = PR * PW * 8, PPT = (PU + NTHR - 1) / NTHR;
    constexpr int WU = WSB / 16, WPT = (WU + NTHR - 1) / NTHR;
    constexpr int NST = 2 * NSL;                         // stages per patch chunk
    constexpr int PQ = (PPT + NST - 3) / (NST - 2);      // patch loads per stage (none in the last two)
    static_assert((NW * MT * 32) % NN == 0 && NN % R == 0, "shape");
    char *const lds0 = conv_smem;
    char *const wlds0 = lds0 + patch_bytes;
    float *const ep = reinterpret_cast<float *>(wlds0 + 2 * WSB);
    const char *const inb = reinterpret_cast<const char *>(a.in);
    const char *const wb = reinterpret_cast<const char *>(a.w);
    constexpr int tiles_per_img = N / R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int n_my = blockIdx.x < (unsigned)total_tiles ? (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (n_my == 0) return;

#define QGX_H4P_LOAD1(TI, PC, V, U)                                                                         \
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
            inb + (((size_t)b_ * N + gy_) * N + gx_) * PIXB + (PC) * 128 + un_ * 16);                       \
    }
#define QGX_H4P_STORE(V)                                                                                    \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            if (it_ < PU) *reinterpret_cast<f32x4 *>(lds0 + (it_ >> 3) * PSTR + (it_ & 7) * 16) = V[u];     \
        }                                                                                                   \
    }
#define QGX_H4W_LOAD1(C16, SL, V, U)                                                                        \
    {                                                                                                       \
        const f32x4 *src_ = reinterpret_cast<const f32x4 *>(wb + ((size_t)(C16) * NSL + (SL)) * WSB);       \
        const int it_ = (U) * NTHR + threadIdx.x;                                                            \
        V[U] = src_[it_ < WU ? it_ : WU - 1];                                                               \
    }
#define QGX_H4W_STORE(BUF, V)                                                                               \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < WPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            if (it_ < WU) *reinterpret_cast<f32x4 *>((BUF) + it_ * 16) = V[u];                              \
        }                                                                                                   \
    }

    for (int i = threadIdx.x; i < 3 * COUT; i += NTHR)
        ep[i] = i < COUT ? a.bias[i] : (i < 2 * COUT ? a.scale[i - COUT] : a.shift[i - 2 * COUT]);
    {
        f32x4 pv[PPT], wv[WPT];
#pragma unroll
        for (int u = 0; u < PPT; ++u) QGX_H4P_LOAD1(0, 0, pv, u)
#pragma unroll
        for (int u = 0; u < WPT; ++u) QGX_H4W_LOAD1(0, 0, wv, u)
        QGX_H4P_STORE(pv)
        QGX_H4W_STORE(wlds0, wv)
    }
    __syncthreads();

    int pbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = (wave + NW * mt) * 32 + li;
        const int py = p / NN, px = p - py * NN;
        pbase[mt] = (py * PW + px) * PSTR + h * 32;
    }
    const int wofs = (h * COUT + li) * 16;
    f32x16 acc[MT][NT];
    int cur_w = 0;
    for (int ti = 0; ti < n_my; ++ti) {
        for (int pc = 0; pc < NPC; ++pc) {
            if (pc == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
            }
            const int npc = pc + 1 < NPC ? pc + 1 : 0;
            const int nti = pc + 1 < NPC ? ti : ti + 1;
            const bool have_next = nti < n_my;
            const int p_ti = have_next ? nti : ti, p_pc = have_next ? npc : pc;
            f32x4 pv[PPT];
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int s = st / NSL, sl = st - s * NSL;          // sub-chunk (16 channels), tap row
                const int c16 = 2 * pc + s;
                f32x4 wv[WPT];
                const bool last_stage = !have_next && st == NST - 1;
                // weight slice of the next stage
                const int wc = sl + 1 < NSL ? c16 : (s == 0 ? c16 + 1 : (have_next ? 2 * npc : c16));
                const int ws = sl + 1 < NSL ? sl + 1 : ((s == 0 || have_next) ? 0 : sl);
                const int pl0 = st * PQ, pl1 = (st + 1) * PQ < PPT ? (st + 1) * PQ : PPT;   // patch loads of this stage
                const int n_pl = st < NST - 2 && pl1 > pl0 ? pl1 - pl0 : 0;
                const int n_ld = WPT + n_pl;
                const char *wl = wlds0 + cur_w * WSB + wofs;
                h8 Pn[MT][2], Wn[NT][2];
#define QGX_H4_FRAGS(TL)                                                                                    \
                {                                                                                           \
                    const int tap_ = sl * TPS + (TL), ky_ = tap_ / KS, kx_ = tap_ - ky_ * KS;               \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                       \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                       \
                            Wn[nt][j] = *reinterpret_cast<const h8 *>(wl + (TL) * TAPB + (j * 2 * COUT + nt * 32) * 16); \
                    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                       \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                       \
                            Pn[mt][j] = *reinterpret_cast<const h8 *>(lds0 + pbase[mt] + (ky_ * PW + kx_) * PSTR + s * 64 + j * 16); \
                }
                QGX_H4_FRAGS(0)
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl) {
                    h8 Pc[MT][2], Wc[NT][2];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) { Pc[mt][0] = Pn[mt][0]; Pc[mt][1] = Pn[mt][1]; }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { Wc[nt][0] = Wn[nt][0]; Wc[nt][1] = Wn[nt][1]; }
                    if (tl + 1 < TPS) {
                        const int tln = tl + 1;
                        const int tap_ = sl * TPS + tln, ky_ = tap_ / KS, kx_ = tap_ - ky_ * KS;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                Wn[nt][j] = *reinterpret_cast<const h8 *>(wl + tln * TAPB + (j * 2 * COUT + nt * 32) * 16);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                Pn[mt][j] = *reinterpret_cast<const h8 *>(lds0 + pbase[mt] + (ky_ * PW + kx_) * PSTR + s * 64 + j * 16);
                    }
                    // this stage's share of the global prefetch loads, spread over the first taps
#pragma unroll
                    for (int i = 0; i < WPT + PQ; ++i) {
                        if (i < n_ld && (i * (TPS - 1)) / n_ld == tl) {
                            if (i < WPT) { QGX_H4W_LOAD1(wc, ws, wv, i) }
                            else { QGX_H4P_LOAD1(p_ti, p_pc, pv, pl0 + i - WPT) }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][1], Pc[mt][0], acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][0], Pc[mt][1], acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][0], Pc[mt][0], acc[mt][nt], 0, 0, 0);
                        }
                }
#undef QGX_H4_FRAGS
                // ---- retire the prefetches: weights into the idle buffer; the patch after the last stage of the chunk
                if (st == NST - 1) {
                    __syncthreads();                     // every wave is done with this patch chunk
                    if (have_next) QGX_H4P_STORE(pv)
                }
                if (!last_stage) QGX_H4W_STORE(wlds0 + (cur_w ^ 1) * WSB, wv)
                __syncthreads();
                cur_w ^= 1;
                if (st == NST - 1 && pc == NPC - 1) {
                    const int tile_g = blockIdx.x + ti * gridDim.x;
                    const int b = tile_g / tiles_per_img;
                    const int y0 = (tile_g - b * tiles_per_img) * R;
                    char *ob = reinterpret_cast<char *>(a.out) + ((size_t)b * N * N + (size_t)y0 * N) * OPIXB;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        char *pix = ob + (size_t)((wave + NW * mt) * 32 + li) * OPIXB;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            store_tile_t<2, false>(acc[mt][nt], nt * 32, h, pix, ep, ep + COUT, ep + 2 * COUT, a.unscale, a.ascale, a.range, a.range_bit);
                    }
                }
            }
        }
    }
#undef QGX_H4P_LOAD1
#undef QGX_H4P_STORE
#undef QGX_H4W_LOAD1
#undef QGX_H4W_STORE
}
