// Hidden conv layers on the 16-bit matrix cores (v_mfma_f32_32x32x16_f16, f32 accumulate).
// Included by conv.hip (inside namespace qgx, after ConvArgs / conv_smem).
//
// Two arithmetic modes share one kernel:
//   NS = 2  "f16x3": every f32 operand x is carried as the pair hi = f16(x), lo = f16(x - hi)
//           (22 significant bits) and a product a*b is evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi
//           with three MFMAs into one f32 accumulator; the dropped a_lo*b_lo term is 2^-22 relative,
//           i.e. the result carries f32-class accuracy at 16/3 of the f32 MFMA rate.
//   NS = 1  "f16": plain f16 operands (the precision class of TF32, which is what the reference's
//           PyTorch convolutions use by default on its own GPUs), one MFMA per product.
// Weights are pre-scaled by a per-layer power of two so that neither part is subnormal; the
// epilogue removes the scale exactly.
//
// Activation layout in HBM: [B][N][N][C/8][NS][8] f16 — per pixel, per group of 8 channels, the hi
// octet followed (NS = 2) by the lo octet.  A channel chunk (32 channels for NS = 1, 16 for NS = 2)
// is 64 contiguous bytes of a pixel's record = four 16-byte units = exactly the K = 16 operand
// fragments of the MFMA: lane half h reads unit (2j + h) [NS = 1, j = K step] or (2h + j) [NS = 2,
// j = part].
//
// The MFMA roles are swapped with respect to the f32 kernels: A = weights (rows = output channels),
// B = pixels (columns), so that an accumulator lane owns ONE pixel and 4 consecutive output channels
// per register quad; one v_permlane32_swap per register pair assembles whole 8-channel octets and the
// epilogue (bias + ReLU + BatchNorm affine + hi/lo split) stores 16 bytes per lane.
//
// One workgroup = 8 waves owns R full-width rows (16 or 24 M-tiles of 32 pixels), is persistent over
// its tiles, and takes BOTH operands from LDS: the (R+K-1)-row patch of one channel chunk (pixel
// stride 80 B: conflict-free ds_read_b128) and the weight slice of TPS taps (double buffered).  The
// next chunk's patch and the next weight slice are fetched into registers while the current one is
// consumed and written to LDS at the stage boundary.
#pragma once

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ConvHArgs {
    const void *in;        // [B][N][N][CIN/8][NS][8] f16
    void *out;             // same with COUT, or NHWC f32 (OUTF32)
    const void *w;         // [chunk][tap][j][h][COUT][8] f16 (slices of TPS taps are contiguous)
    const float *bias, *scale, *shift;
    float unscale;         // 1 / (weight pre-scale * input activation pre-scale), a power of two
    float ascale;          // pre-scale of the stored output activations, a power of two
    int N, R;
};

__device__ __forceinline__ unsigned pack_h2(float a, float b) {
    h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}

// Epilogue of one 32(out channels) x 32(pixels) accumulator tile in the swapped-role layout:
// lane (li, h) owns pixel li; register r holds output channel cb + (r & 3) + 8 (r >> 2) + 4 h.
template <int NS, bool OUTF32>
__device__ __forceinline__ void store_tile_t(const f32x16 &acc, int cb, int h, char *pix, const float *bias,
                                             const float *scale, const float *shift, float unscale, float ascale) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c0 = cb + 8 * q + 4 * h;
        const f32x4 bi = *reinterpret_cast<const f32x4 *>(bias + c0);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = fmaxf(acc[4 * q + e] * unscale + bi[e], 0.f) * sc[e] + sh[e];
    }
    if constexpr (OUTF32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            *reinterpret_cast<f32x4 *>(pix + (size_t)(cb + 8 * q + 4 * h) * 4) = o;
        }
    } else {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = v[8 * m + e] * ascale;
                const _Float16 xh = (_Float16)x;
                hi[e] = (float)xh;
                lo[e] = x - hi[e];
            }
            // this lane: channels 4h..4h+3 of octet ga (e = 0..3) and of octet gb = ga + 1 (e = 4..7)
            const int g = (cb >> 3) + 2 * m + h;
            unsigned a0 = pack_h2(hi[0], hi[1]), a1 = pack_h2(hi[2], hi[3]);
            unsigned b0 = pack_h2(hi[4], hi[5]), b1 = pack_h2(hi[6], hi[7]);
            auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            *reinterpret_cast<u32x4 *>(pix + (size_t)g * 16 * NS) = o;
            if constexpr (NS == 2) {
                a0 = pack_h2(lo[0], lo[1]); a1 = pack_h2(lo[2], lo[3]);
                b0 = pack_h2(lo[4], lo[5]); b1 = pack_h2(lo[6], lo[7]);
                s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                u32x4 ol = {s0[0], s1[0], s0[1], s1[1]};
                *reinterpret_cast<u32x4 *>(pix + (size_t)g * 32 + 16) = ol;
            }
        }
    }
}

template <int CIN, int COUT, int KS, int NS, int MT, int TPS, int PPT, bool OUTF32>
__global__ __launch_bounds__(512) void k_convh(ConvHArgs a, int total_tiles) {
    constexpr int NW = 8, NTHR = 512;
    constexpr int NT = COUT / 32;
    constexpr int P = KS / 2, T = KS * KS;
    constexpr int CC = NS == 1 ? 32 : 16;
    constexpr int NCH = CIN / CC;
    constexpr int PIXB = CIN * 2 * NS;                  // bytes of one input pixel record
    constexpr int OPIXB = OUTF32 ? COUT * 4 : COUT * 2 * NS;
    constexpr int PSTR = 80;                            // LDS bytes per patch pixel (64 payload + 16 pad)
    constexpr int TAPB = 4 * COUT * 16;                 // weight bytes per tap: [j][h][cout][8 f16]
    constexpr int WSB = TPS * TAPB;
    constexpr int NSL = T / TPS;
    constexpr int WU = WSB / 16;
    constexpr int WPT = (WU + NTHR - 1) / NTHR;
    static_assert(T % TPS == 0 && COUT % 32 == 0 && CIN % CC == 0, "shape");
    const int N = a.N, R = a.R;
    const int PR = R + KS - 1;
    const int patch_bytes = PR * N * PSTR;
    const int PU = PR * N * 4;                          // 16-byte units of the patch payload
    char *const lds0 = conv_smem;
    char *const wlds0 = lds0 + patch_bytes;
    const char *const inb = reinterpret_cast<const char *>(a.in);
    const char *const wb = reinterpret_cast<const char *>(a.w);
    const int tiles_per_img = N / R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int n_my = blockIdx.x < (unsigned)total_tiles ? (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int ntiles = R * N / 32;

#define QGX_HP_LOAD(TI, CH, V)                                                                              \
    {                                                                                                       \
        const int tile_ = blockIdx.x + (TI) * gridDim.x;                                                    \
        const int b_ = tile_ / tiles_per_img;                                                               \
        const int y0_ = (tile_ - b_ * tiles_per_img) * R;                                                   \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            int it_ = u * NTHR + threadIdx.x;                                                                \
            it_ = it_ < PU ? it_ : PU - 1; /* clamped: branch-free, the store is predicated instead */      \
            const int un_ = it_ & 3, pl_ = it_ >> 2;                                                        \
            const int pr_ = pl_ / N, x_ = pl_ - pr_ * N;                                                    \
            int gy_ = y0_ - P + pr_;                                                                        \
            gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                           \
            V[u] = *reinterpret_cast<const f32x4 *>(                                                        \
                inb + (((size_t)b_ * N + gy_) * N + x_) * PIXB + (CH) * 64 + un_ * 16);                     \
        }                                                                                                   \
    }
#define QGX_HP_STORE(V)                                                                                     \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            if (it_ < PU) *reinterpret_cast<f32x4 *>(lds0 + (it_ >> 2) * PSTR + (it_ & 3) * 16) = V[u];     \
        }                                                                                                   \
    }
#define QGX_HW_LOAD(CH, SL, V)                                                                              \
    {                                                                                                       \
        const f32x4 *src_ = reinterpret_cast<const f32x4 *>(wb + ((size_t)(CH) * NSL + (SL)) * WSB);        \
        _Pragma("unroll") for (int u = 0; u < WPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            V[u] = src_[it_ < WU ? it_ : WU - 1];                                                           \
        }                                                                                                   \
    }
#define QGX_HW_STORE(BUF, V)                                                                                \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < WPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            if (it_ < WU) *reinterpret_cast<f32x4 *>((BUF) + it_ * 16) = V[u];                              \
        }                                                                                                   \
    }

    if (n_my == 0) return;
    // ---- prologue: first chunk's patch and first weight slice, synchronously
    {
        f32x4 pv[PPT];
        QGX_HP_LOAD(0, 0, pv)
        QGX_HP_STORE(pv)
        f32x4 wv[WPT];
        QGX_HW_LOAD(0, 0, wv)
        QGX_HW_STORE(wlds0, wv)
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
    // lane-dependent parts of the fragment addresses
    const int pofs = NS == 1 ? h * 16 : h * 32;         // + j*32 (NS=1) / + j*16 (NS=2)
    constexpr int PJ = NS == 1 ? 32 : 16;
    const int wofs = (h * COUT + li) * 16;              // + (j*2*COUT + nt*32)*16
    f32x16 acc[MT][NT];
    int cur_w = 0;
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
            const int nch = ch + 1 < NCH ? ch + 1 : 0;
            const int nti = ch + 1 < NCH ? ti : ti + 1;
            const bool have_next_chunk = nti < n_my;
            // ---- next chunk's patch: into registers now, into LDS when this chunk is consumed
            f32x4 pv[PPT];
            QGX_HP_LOAD(have_next_chunk ? nti : ti, have_next_chunk ? nch : ch, pv)
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) {
                f32x4 wv[WPT];
                const bool last_stage = !have_next_chunk && sl == NSL - 1;
                {
                    const int wch = sl + 1 < NSL ? ch : (have_next_chunk ? nch : ch);
                    const int wsl = sl + 1 < NSL ? sl + 1 : (have_next_chunk ? 0 : sl);
                    QGX_HW_LOAD(wch, wsl, wv)
                }
                // ---- K loop over the TPS taps of this slice, fragments requested one tap ahead
                const char *wl = wlds0 + cur_w * WSB + wofs;
                const int tap0 = sl * TPS;
                int ky = tap0 / KS, kx = tap0 - ky * KS;
                int aoff[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int col = px[mt] + kx - P;
                    col = col < 0 ? col + N : (col >= N ? col - N : col);
                    aoff[mt] = ((py[mt] + ky) * N + col) * PSTR + pofs;
                }
                h8 Pn[MT][2], Wn[NT][2];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        Wn[nt][j] = *reinterpret_cast<const h8 *>(wl + (j * 2 * COUT + nt * 32) * 16);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Pn[mt][j] = *reinterpret_cast<const h8 *>(lds0 + aoff[mt] + j * PJ);
                for (int tl = 0; tl < TPS; ++tl) {
                    const bool last_tap = tl == TPS - 1;
                    int nkx = kx + 1, nky = ky;
                    if (nkx == KS) { nkx = 0; ++nky; }
                    int aoff_n[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        int col = px[mt] + nkx - P;
                        col = col < 0 ? col + N : (col >= N ? col - N : col);
                        aoff_n[mt] = last_tap ? aoff[mt] : ((py[mt] + nky) * N + col) * PSTR + pofs;
                    }
                    const char *wl_n = last_tap ? wl : wl + TAPB;
                    h8 Pc[MT][2], Wc[NT][2];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) { Pc[mt][0] = Pn[mt][0]; Pc[mt][1] = Pn[mt][1]; }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { Wc[nt][0] = Wn[nt][0]; Wc[nt][1] = Wn[nt][1]; }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            Wn[nt][j] = *reinterpret_cast<const h8 *>(wl_n + (j * 2 * COUT + nt * 32) * 16);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int j = 0; j < 2; ++j) Pn[mt][j] = *reinterpret_cast<const h8 *>(lds0 + aoff_n[mt] + j * PJ);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            if constexpr (NS == 1) {
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][0], Pc[mt][0], acc[mt][nt], 0, 0, 0);
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][1], Pc[mt][1], acc[mt][nt], 0, 0, 0);
                            } else {
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][1], Pc[mt][0], acc[mt][nt], 0, 0, 0);
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][0], Pc[mt][1], acc[mt][nt], 0, 0, 0);
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][0], Pc[mt][0], acc[mt][nt], 0, 0, 0);
                            }
                        }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) aoff[mt] = aoff_n[mt];
                    wl = wl_n;
                    kx = nkx; ky = nky;
                }

                if (sl == NSL - 1 && ch == NCH - 1) {
                    // ---- epilogue of this tile
                    const int tile_g = blockIdx.x + ti * gridDim.x;
                    const int b = tile_g / tiles_per_img;
                    const int y0 = (tile_g - b * tiles_per_img) * R;
                    char *ob = reinterpret_cast<char *>(a.out) + ((size_t)b * N * N + (size_t)y0 * N) * OPIXB;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int tile = wave + NW * mt;
                        if (tile >= ntiles) continue;
                        char *pix = ob + (size_t)(tile * 32 + li) * OPIXB;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            store_tile_t<NS, OUTF32>(acc[mt][nt], nt * 32, h, pix, a.bias, a.scale, a.shift, a.unscale, a.ascale);
                    }
                }
                // ---- retire the prefetches
                if (sl == NSL - 1) {
                    __syncthreads();                     // every wave is done with this chunk's patch
                    if (have_next_chunk) QGX_HP_STORE(pv)
                }
                if (!last_stage) QGX_HW_STORE(wlds0 + (cur_w ^ 1) * WSB, wv)
                __syncthreads();
                cur_w ^= 1;
            }
        }
    }
#undef QGX_HP_LOAD
#undef QGX_HP_STORE
#undef QGX_HW_LOAD
#undef QGX_HW_STORE
}
