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

// (h8, h2, u32x4, pack_h2, range_guard: conv_types.hpp)

struct ConvHArgs {
    const void *in;        // [B][N][N][CIN/8][NS][8] f16
    void *out;             // same with COUT, or NHWC f32 (OUTF32)
    const void *w;         // [chunk][tap][j][h][COUT][8] f16 (slices of TPS taps are contiguous)
    const float *bias, *scale, *shift;
    float unscale;         // 1 / (weight pre-scale * input activation pre-scale), a power of two
    float ascale;          // pre-scale of the stored output activations, a power of two
    int N, R;
    size_t npix_total;     // split-K: B*N*N, the stride of one partial-sum plane
    int prio_alt;          // k_convh2: alternate the wave priority of the two co-resident workgroups per tile
    unsigned *range;       // f16x3 range guard: sticky flag word of the generator (bit = layer), see range_guard()
    unsigned range_bit;
    unsigned long long *stamps;   // diagnostic builds (-DQGX_STAMPS) only: s_memtime trace, 64 slots per workgroup
};

#ifdef QGX_STAMPS
#ifdef QGX_STAMPS_REALTIME      // 100 MHz, the same counter on every XCD: workgroup start / end skew across the chip
#define QGX_STAMP_CLOCK __builtin_amdgcn_s_memrealtime
#else                           // shader-clock cycles, comparable only within one CU
#define QGX_STAMP_CLOCK __builtin_amdgcn_s_memtime
#endif
#define QGX_STAMP()                                                                          \
    if (a.stamps && threadIdx.x == 0) {                                                      \
        a.stamps[blockIdx.x * 64 + (stamp_i < 63 ? stamp_i : 63)] = QGX_STAMP_CLOCK();       \
        if (stamp_i < 63) ++stamp_i;                                                         \
    }
#else
#define QGX_STAMP()
#endif

// prologue copies global -> LDS through registers: ALL loads are issued before the first store, so a
// prologue pays one memory latency instead of one per loop iteration
#define QGX_BULK_LOAD(V, SRC, NUNITS, NTHR_)                                                                \
    _Pragma("unroll") for (int u = 0; u < ((NUNITS) + (NTHR_) - 1) / (NTHR_); ++u) {                        \
        const int it_ = u * (NTHR_) + threadIdx.x;                                                           \
        V[u] = reinterpret_cast<const f32x4 *>(SRC)[it_ < (NUNITS) ? it_ : (NUNITS) - 1];                   \
    }
#define QGX_BULK_STORE(V, DST, NUNITS, NTHR_)                                                               \
    _Pragma("unroll") for (int u = 0; u < ((NUNITS) + (NTHR_) - 1) / (NTHR_); ++u) {                        \
        const int it_ = u * (NTHR_) + threadIdx.x;                                                           \
        if (it_ < (NUNITS)) reinterpret_cast<f32x4 *>(DST)[it_] = V[u];                                     \
    }

// Epilogue of one 32(out channels) x 32(pixels) accumulator tile in the swapped-role layout:
// lane (li, h) owns pixel li; register r holds output channel cb + (r & 3) + 8 (r >> 2) + 4 h.
// guard_mul: the consumer's amplification of the stored values before ITS 16-bit split (the Winograd layer's input
// transform: up to 16) — the guard then fires at 65504 / guard_mul
template <int NS, bool OUTF32>
__device__ __forceinline__ void store_tile_t(const f32x16 &acc, int cb, int h, char *pix, const float *bias,
                                             const float *scale, const float *shift, float unscale, float ascale,
                                             unsigned *range, unsigned range_bit, float guard_mul = 1.f) {
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
        float mx = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(v[e]));
        range_guard(mx * ascale * guard_mul, range, range_bit);
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

// Epilogue of one 32(pixels) x 32(out channels) accumulator tile in the UN-swapped layout (mfma(patch, weights)):
// lane (li, h) owns output channel cb + li; register r holds pixel (r & 3) + 8 (r >> 2) + 4 h of the tile's 32
// consecutive strip pixels.  Stores the channel-planar form the Winograd layer's MFMA input transform reads
// (conv_wino.hpp): [b][y][c][hi | lo][x] f16 — after one v_permlane32_swap per register pair a lane holds 8 consecutive
// pixels of its channel: lane half h stores the octets 2 m + h (m = 0, 1), 16 bytes of hi and 16 of lo each.
// rowbase: the channel-0 hi row of the tile's first pixel's image row; p0: index of the tile's first pixel in the strip.
__device__ __forceinline__ void store_tile_planar(const f32x16 &acc, int c, int h, char *strip, int p0, int N, int COUT,
                                                  const float *bias, const float *scale, const float *shift, float unscale,
                                                  float ascale, unsigned *range, unsigned range_bit, float guard_mul) {
    const float bi = bias[c], sc = scale[c] * ascale, sh = shift[c] * ascale;     // ascale: a power of two, exact
    float v[16];
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        v[r] = fmaxf(acc[r] * unscale + bi, 0.f) * sc + sh;
        mx = fmaxf(mx, fabsf(v[r]));
    }
    range_guard(mx * guard_mul, range, range_bit);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        // registers 8 m .. 8 m + 3: pixels 16 m + 4 h + (0..3) ("a"), 8 m + 4 .. 8 m + 7: pixels 16 m + 8 + 4 h + (0..3) ("b")
        unsigned ah[2], bh[2], al[2], bl[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const float a0 = v[8 * m + 2 * e2], a1 = v[8 * m + 2 * e2 + 1], b0 = v[8 * m + 4 + 2 * e2], b1 = v[8 * m + 5 + 2 * e2];
            const _Float16 a0h = (_Float16)a0, a1h = (_Float16)a1, b0h = (_Float16)b0, b1h = (_Float16)b1;
            ah[e2] = pack_h2((float)a0h, (float)a1h); bh[e2] = pack_h2((float)b0h, (float)b1h);
            al[e2] = pack_h2(a0 - (float)a0h, a1 - (float)a1h); bl[e2] = pack_h2(b0 - (float)b0h, b1 - (float)b1h);
        }
        auto s0 = __builtin_amdgcn_permlane32_swap(ah[0], bh[0], false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(ah[1], bh[1], false, false);
        const u32x4 oh = {s0[0], s1[0], s0[1], s1[1]};
        s0 = __builtin_amdgcn_permlane32_swap(al[0], bl[0], false, false);
        s1 = __builtin_amdgcn_permlane32_swap(al[1], bl[1], false, false);
        const u32x4 ol = {s0[0], s1[0], s0[1], s1[1]};
        const int px = p0 + 8 * (2 * m + h);                  // first of this lane's 8 consecutive strip pixels
        const int y = px / N, x = px - y * N;
        char *o = strip + (((size_t)y * COUT + c) * 2 * N + x) * 2;
        *reinterpret_cast<u32x4 *>(o) = oh;
        *reinterpret_cast<u32x4 *>(o + 2 * N) = ol;
    }
}

#ifdef QGX_AB   // generic run-time-N kernel (and the plain-f16 mode): A/B builds only, see conv.hip
template <int CIN, int COUT, int KS, int NS, int MT, int TPS, int PPT, bool OUTF32, int NW = 8, bool SWZ = false>
__global__ __launch_bounds__(NW * 64) void k_convh(ConvHArgs a, int total_tiles) {
    constexpr int NTHR = NW * 64;
    constexpr int NT = COUT / 32;
    constexpr int P = KS / 2, T = KS * KS;
    constexpr int CC = NS == 1 ? 32 : 16;
    constexpr int NCH = CIN / CC;
    constexpr int PIXB = CIN * 2 * NS;                  // bytes of one input pixel record
    constexpr int OPIXB = OUTF32 ? COUT * 4 : COUT * 2 * NS;
    // LDS bytes per patch pixel: 64 payload + 16 pad, or (SWZ) 64 with the unit index XOR-swizzled by
    // bits 2..3 of the pixel index — both make every 16-lane ds_read_b128 group conflict-free
    constexpr int PSTR = SWZ ? 64 : 80;
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
    // epilogue parameters (bias | BN scale | BN shift) live in LDS: a global load in the epilogue would put a
    // full memory latency, and a wait on the tile's own output stores, on the critical path of every tile
    float *const ep = reinterpret_cast<float *>(wlds0 + 2 * WSB);
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
            if (it_ < PU)                                                                                   \
                *reinterpret_cast<f32x4 *>(lds0 + (it_ >> 2) * PSTR +                                       \
                                           (SWZ ? ((it_ & 3) ^ ((it_ >> 4) & 3)) : (it_ & 3)) * 16) = V[u];  \
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
    int stamp_i = 0;
    (void)stamp_i;
    QGX_STAMP()
    for (int i = threadIdx.x; i < 3 * COUT; i += NTHR)
        ep[i] = i < COUT ? a.bias[i] : (i < 2 * COUT ? a.scale[i - COUT] : a.shift[i - 2 * COUT]);
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
    // lane-dependent parts of the fragment addresses: unit u0 (+ j*PJ bytes for fragment j)
    const int u0 = NS == 1 ? h : 2 * h;
    constexpr int PJ = NS == 1 ? 32 : 16;
#define QGX_HP_ADDR(PL) (SWZ ? (PL) * 64 + ((u0 ^ (((PL) >> 2) & 3)) * 16) : (PL) * 80 + u0 * 16)
#define QGX_HP_FRAG(AOFF, J) (SWZ ? ((AOFF) ^ ((J) * PJ)) : ((AOFF) + (J) * PJ))
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
                QGX_STAMP()
                // ---- K loop over the TPS taps of this slice, fragments requested one tap ahead
                const char *wl = wlds0 + cur_w * WSB + wofs;
                const int tap0 = sl * TPS;
                int ky = tap0 / KS, kx = tap0 - ky * KS;
                int aoff[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int col = px[mt] + kx - P;
                    col = col < 0 ? col + N : (col >= N ? col - N : col);
                    aoff[mt] = QGX_HP_ADDR((py[mt] + ky) * N + col);
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
                    for (int j = 0; j < 2; ++j) Pn[mt][j] = *reinterpret_cast<const h8 *>(lds0 + QGX_HP_FRAG(aoff[mt], j));
                for (int tl = 0; tl < TPS; ++tl) {
                    const bool last_tap = tl == TPS - 1;
                    int nkx = kx + 1, nky = ky;
                    if (nkx == KS) { nkx = 0; ++nky; }
                    int aoff_n[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        int col = px[mt] + nkx - P;
                        col = col < 0 ? col + N : (col >= N ? col - N : col);
                        aoff_n[mt] = last_tap ? aoff[mt] : QGX_HP_ADDR((py[mt] + nky) * N + col);
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
                        for (int j = 0; j < 2; ++j) Pn[mt][j] = *reinterpret_cast<const h8 *>(lds0 + QGX_HP_FRAG(aoff_n[mt], j));
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

                QGX_STAMP()
                // ---- retire the prefetches
                if (sl == NSL - 1) {
                    __syncthreads();                     // every wave is done with this chunk's patch
                    if (have_next_chunk) QGX_HP_STORE(pv)
                }
                if (!last_stage) QGX_HW_STORE(wlds0 + (cur_w ^ 1) * WSB, wv)
                __syncthreads();
                QGX_STAMP()
                cur_w ^= 1;
                if (sl == NSL - 1 && ch == NCH - 1) {
                    // ---- epilogue of this tile, AFTER the prefetches were retired: its stores then drain during the
                    // next stage instead of being waited for (vmcnt counts loads and stores together, in order)
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
                            store_tile_t<NS, OUTF32>(acc[mt][nt], nt * 32, h, pix, ep, ep + COUT, ep + 2 * COUT, a.unscale, a.ascale, a.range, a.range_bit);
                    }
                }
            }
        }
    }
#undef QGX_HP_LOAD
#undef QGX_HP_STORE
#undef QGX_HW_LOAD
#undef QGX_HW_STORE
#undef QGX_HP_ADDR
#undef QGX_HP_FRAG
}

#endif  // QGX_AB

// ---- f16x3 hidden layers, compile-time grid size: 4 waves x 2 workgroups per CU ---------------------------
// Same arithmetic and data layouts as k_convh<NS = 2>, built so that TWO workgroups share a CU and run
// out of phase (one's barriers, LDS staging and operand latency under the other's MFMAs): that needs
// <= 256 registers per wave, which the generic kernel misses because hipcc hoists its 25 x MT
// loop-invariant per-lane tap addresses into registers.  Here the LDS patch carries a wrapped halo of
// KS/2 columns on both sides, so a tap is a compile-time byte offset from ONE per-lane base address
// (ds_read offset immediates, no address arithmetic, no address registers).  The weight slice is single
// buffered where two buffers would not leave room for two workgroups (the 5x5 layer).
// TW: tile width (default: full rows).  TW < NN tiles the image in x as well (96 = 3 x 32: sixteen 32-pixel rows
// give the two-M-tiles-per-wave shape that does not spill, where full 96-pixel rows need three).
template <int CIN, int COUT, int KS, int NN, int MT, int TPS, bool OUTF32, bool WDB, bool TWO, bool PAIR = false, bool PART = false,
          int NW = 4, int TW = NN>
__global__ __launch_bounds__(NW * 64, TWO ? 2 : 1) void k_convh2(ConvHArgs a, int total_tiles) {
    constexpr int NTHR = NW * 64;
    constexpr int NT = COUT / 32;
    constexpr int P = KS / 2, T = KS * KS;
    constexpr int NCH = CIN / 16;
    constexpr int PIXB = CIN * 4, OPIXB = COUT * 4;
    constexpr int N = NN, R = NW * MT * 32 / TW, PR = R + KS - 1, PW = TW + 2 * P, XT = NN / TW;
    constexpr int PSTR = 80;
    constexpr int patch_bytes = PR * PW * PSTR;
    constexpr int TAPB = 4 * COUT * 16, WSB = TPS * TAPB, NSL = T / TPS;
    constexpr int PU = PR * PW * 4, PPT = (PU + NTHR - 1) / NTHR;
    constexpr int WU = WSB / 16, WPT = (WU + NTHR - 1) / NTHR;
    static_assert(T % TPS == 0 && (NW * MT * 32) % TW == 0 && NN % R == 0 && NN % TW == 0 && (TW == NN || TW % 32 == 0), "shape");
    static_assert(TW == NN || !PART, "split-K uses full-row tiles");
    static_assert(!PAIR || (NCH % 2 == 0 && NSL == 1), "line-pair prefetch: even chunk count, one slice per chunk");
    static_assert(!(PAIR && PART), "split-K runs without the line-pair prefetch");
    // PART (single members): blockIdx.y owns the chunks [cbeg, cend) of every tile and stores raw f32 partial
    // sums [split][pixel][cout]; k_convh_reduce adds them in a fixed order and applies the epilogue
    const int cbeg = PART ? (int)blockIdx.y * (NCH / (int)gridDim.y) : 0;
    const int cend = PART ? cbeg + NCH / (int)gridDim.y : NCH;
    char *const lds0 = conv_smem;
    char *const wlds0 = lds0 + patch_bytes;
    float *const ep = reinterpret_cast<float *>(wlds0 + (WDB ? 2 : 1) * WSB);
    const char *const inb = reinterpret_cast<const char *>(a.in);
    const char *const wb = reinterpret_cast<const char *>(a.w);
    constexpr int tiles_per_img = (N / R) * XT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int n_my = blockIdx.x < (unsigned)total_tiles ? (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (n_my == 0) return;
    int stamp_i = 0;
    (void)stamp_i;
    QGX_STAMP()

#define QGX_H2P_LOAD(TI, CH, V)                                                                             \
    {                                                                                                       \
        const int tile_ = blockIdx.x + (TI) * gridDim.x;                                                    \
        const int b_ = tile_ / tiles_per_img;                                                               \
        const int tr_ = tile_ - b_ * tiles_per_img;                                                         \
        const int y0_ = (tr_ / XT) * R, x0_ = (tr_ % XT) * TW;                                              \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            int it_ = u * NTHR + threadIdx.x;                                                                \
            it_ = it_ < PU ? it_ : PU - 1;                                                                  \
            const int un_ = it_ & 3, pl_ = it_ >> 2;                                                        \
            const int pr_ = pl_ / PW, xx_ = pl_ - pr_ * PW;                                                 \
            int gy_ = y0_ - P + pr_, gx_ = x0_ + xx_ - P;                                                   \
            gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                           \
            gx_ = gx_ < 0 ? gx_ + N : (gx_ >= N ? gx_ - N : gx_);                                           \
            V[u] = *reinterpret_cast<const f32x4 *>(                                                        \
                inb + (((size_t)b_ * N + gy_) * N + gx_) * PIXB + (CH) * 64 + un_ * 16);                    \
        }                                                                                                   \
    }
#define QGX_H2P_LOAD1(TI, CH, V, U)                                                                         \
    {                                                                                                       \
        const int tile_ = blockIdx.x + (TI) * gridDim.x;                                                    \
        const int b_ = tile_ / tiles_per_img;                                                               \
        const int tr_ = tile_ - b_ * tiles_per_img;                                                         \
        const int y0_ = (tr_ / XT) * R, x0_ = (tr_ % XT) * TW;                                              \
        int it_ = (U) * NTHR + threadIdx.x;                                                                  \
        it_ = it_ < PU ? it_ : PU - 1;                                                                      \
        const int un_ = it_ & 3, pl_ = it_ >> 2;                                                            \
        const int pr_ = pl_ / PW, xx_ = pl_ - pr_ * PW;                                                     \
        int gy_ = y0_ - P + pr_, gx_ = x0_ + xx_ - P;                                                       \
        gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                               \
        gx_ = gx_ < 0 ? gx_ + N : (gx_ >= N ? gx_ - N : gx_);                                               \
        V[U] = *reinterpret_cast<const f32x4 *>(                                                            \
            inb + (((size_t)b_ * N + gy_) * N + gx_) * PIXB + (CH) * 64 + un_ * 16);                        \
    }
#define QGX_H2W_LOAD1(CH, SL, V, U)                                                                         \
    {                                                                                                       \
        const f32x4 *src_ = reinterpret_cast<const f32x4 *>(wb + ((size_t)(CH) * NSL + (SL)) * WSB);        \
        const int it_ = (U) * NTHR + threadIdx.x;                                                            \
        V[U] = src_[it_ < WU ? it_ : WU - 1];                                                               \
    }
#define QGX_H2P_STORE(V)                                                                                    \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            if (it_ < PU) *reinterpret_cast<f32x4 *>(lds0 + (it_ >> 2) * PSTR + (it_ & 3) * 16) = V[u];     \
        }                                                                                                   \
    }
#define QGX_H2W_LOAD(CH, SL, V)                                                                             \
    {                                                                                                       \
        const f32x4 *src_ = reinterpret_cast<const f32x4 *>(wb + ((size_t)(CH) * NSL + (SL)) * WSB);        \
        _Pragma("unroll") for (int u = 0; u < WPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            V[u] = src_[it_ < WU ? it_ : WU - 1];                                                           \
        }                                                                                                   \
    }
#define QGX_H2W_STORE(BUF, V)                                                                               \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < WPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            if (it_ < WU) *reinterpret_cast<f32x4 *>((BUF) + it_ * 16) = V[u];                              \
        }                                                                                                   \
    }

    // PAIR: a pixel's two consecutive 16-channel chunks are the two halves of one 128-byte line; fetching them
    // in different chunk iterations re-fetched the line from HBM (FETCH_SIZE 1.3-1.8x the algorithmic
    // bytes on these bandwidth-bound layers), so both halves are loaded together, one chunk PAIR ahead, and
    // the odd half waits in registers for its turn
    f32x4 pvA[PAIR ? PPT : 1], pvB[PAIR ? PPT : 1];
    {
        // prologue: EVERY global load (first patch, first weight slice, epilogue parameters) is issued before the
        // first wait, so that a workgroup pays one memory latency here instead of three in a row (stamped: 11.3 k
        // cycles = 14 % of a 3x3 layer's kernel)
        f32x4 pv[PAIR ? 1 : PPT];
        f32x4 wv[WPT];
        if constexpr (PAIR) {
            QGX_H2P_LOAD(0, 0, pvA)
            QGX_H2P_LOAD(0, 1, pvB)
        } else {
            QGX_H2P_LOAD(0, cbeg, pv)
        }
        QGX_H2W_LOAD(cbeg, 0, wv)
        for (int i = threadIdx.x; i < 3 * COUT; i += NTHR)
            ep[i] = i < COUT ? a.bias[i] : (i < 2 * COUT ? a.scale[i - COUT] : a.shift[i - 2 * COUT]);
        if constexpr (PAIR) {
            QGX_H2P_STORE(pvA)
        } else {
            QGX_H2P_STORE(pv)
        }
        QGX_H2W_STORE(wlds0, wv)
    }
    __syncthreads();

    // ONE base address per M-tile: pixel (py, px) of the tile sits at patch row py, column px (the halo
    // shifts the origin by -P, -P); tap (ky, kx) adds the compile-time (ky * PW + kx) * PSTR
    int pbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = (wave + NW * mt) * 32 + li;
        const int py = p / TW, px = p - py * TW;
        pbase[mt] = (py * PW + px) * PSTR + h * 32;
    }
    const int wofs = (h * COUT + li) * 16;
    f32x16 acc[MT][NT];
    int cur_w = 0;
    for (int ti = 0; ti < n_my; ++ti) {
        // two workgroups share a CU and the older one wins every issue arbitration: it finished its tiles 12-19 %
        // earlier and left the younger one to run alone (without a partner to hide its latencies) at the end.
        // Alternating the wave priority per tile between the two keeps them in step.
        if (TWO && a.prio_alt == 1) {
            if ((ti ^ (blockIdx.x >= (gridDim.x >> 1) ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
        // PAIR: a run-time loop over chunk pairs with the two chunks of a pair unrolled (the parity decides
        // which prefetch set is loaded / stored, so it has to be a compile-time constant)
        for (int cp = cbeg; cp < cend; cp += (PAIR ? 2 : 1))
#pragma unroll
        for (int ci = 0; ci < (PAIR ? 2 : 1); ++ci) {
            const int ch = cp + ci;
            const bool odd = ci == 1;
            if (TWO && a.prio_alt == 2) {
                if ((ch ^ ti ^ (blockIdx.x >= (gridDim.x >> 1) ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(0);
            }
            if (ch == cbeg) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
            }
            const int nch = ch + 1 < cend ? ch + 1 : cbeg;
            const int nti = ch + 1 < cend ? ti : ti + 1;
            const bool have_next_chunk = nti < n_my;
            f32x4 pv[PAIR ? 1 : PPT];
            // the global prefetch loads of a stage (next weight slice; at the first slice of a chunk the next
            // chunk's patch — PAIR: at odd chunks the next pair, nch is even then) are NOT issued as a burst
            // in front of the K loop: the CU's vector-memory path takes 64 B per clock, so a burst of 14-19
            // 1-KB wave loads from every wave kept the in-order waves away from their MFMAs for 1-3 k cycles
            // per stage (stamped timeline); they are spread over the taps instead, weights first
            const int p_ti = have_next_chunk ? nti : ti;
            const int p_ch = PAIR ? (have_next_chunk ? nch : ch - 1) : (have_next_chunk ? nch : ch);
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) {
                f32x4 wv[WPT];
                const bool last_stage = !have_next_chunk && sl == NSL - 1;
                const int wch = sl + 1 < NSL ? ch : (have_next_chunk ? nch : ch);
                const int wsl = sl + 1 < NSL ? sl + 1 : (have_next_chunk ? 0 : sl);
                const int n_pl = PAIR ? (odd ? 2 * PPT : 0) : (sl == 0 ? PPT : 0);   // patch loads of this stage
                const int n_ld = WPT + n_pl;
                constexpr int TSPREAD = TPS > 2 ? TPS - 1 : TPS;                       // the last tap carries none
                const char *wl = wlds0 + (WDB ? cur_w * WSB : 0) + wofs;
                h8 Pn[MT][2], Wn[NT][2];
#define QGX_H2_FRAGS(TL)                                                                                    \
                {                                                                                           \
                    const int tap_ = sl * TPS + (TL), ky_ = tap_ / KS, kx_ = tap_ - ky_ * KS;               \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                       \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                       \
                            Wn[nt][j] = *reinterpret_cast<const h8 *>(wl + (TL) * TAPB + (j * 2 * COUT + nt * 32) * 16); \
                    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                       \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                       \
                            Pn[mt][j] = *reinterpret_cast<const h8 *>(lds0 + pbase[mt] + (ky_ * PW + kx_) * PSTR + j * 16); \
                }
                QGX_STAMP()
                // prio_alt == 3 (two workgroups per CU): the MFMA loop at priority 0, everything else (prefetch stores, barriers,
                // epilogue) at priority 3 — beside a partner wave whose next MFMA is always pending, vector instructions of equal
                // priority issue every ~17 cycles instead of every ~5 (bench_tools/coissue.hip); the MFMA stream does not slow down
                if (TWO && a.prio_alt == 3) __builtin_amdgcn_s_setprio(0);
                QGX_H2_FRAGS(0)
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
                                Pn[mt][j] = *reinterpret_cast<const h8 *>(lds0 + pbase[mt] + (ky_ * PW + kx_) * PSTR + j * 16);
                    }
#pragma unroll
                    for (int i = 0; i < WPT + 2 * PPT; ++i) {
                        if (i < n_ld && (i * TSPREAD) / n_ld == tl) {
                            if (i < WPT) {
                                QGX_H2W_LOAD1(wch, wsl, wv, i)
                            } else if constexpr (PAIR) {
                                if (i - WPT < PPT) { QGX_H2P_LOAD1(p_ti, p_ch, pvA, i - WPT) }
                                else { QGX_H2P_LOAD1(p_ti, p_ch + 1, pvB, i - WPT - PPT) }
                            } else {
                                if (i - WPT < PPT) QGX_H2P_LOAD1(p_ti, p_ch, pv, i - WPT)
                            }
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
#undef QGX_H2_FRAGS
                QGX_STAMP()
                if (TWO && a.prio_alt == 3) __builtin_amdgcn_s_setprio(3);
                // ---- retire the prefetches
                if constexpr (WDB) {
                    if (sl == NSL - 1) {
                        __syncthreads();
                        QGX_STAMP()
                        if constexpr (PAIR) {
                            if (odd) { if (have_next_chunk) QGX_H2P_STORE(pvA) } else QGX_H2P_STORE(pvB)
                        } else {
                            if (have_next_chunk) QGX_H2P_STORE(pv)
                        }
                    }
                    if (!last_stage) QGX_H2W_STORE(wlds0 + (cur_w ^ 1) * WSB, wv)
                    QGX_STAMP()
                    __syncthreads();
                    QGX_STAMP()
                    cur_w ^= 1;
                } else {
                    __syncthreads();                     // every wave is done with this slice (and chunk)
                    if (!last_stage) QGX_H2W_STORE(wlds0, wv)
                    if constexpr (!PAIR) { if (sl == NSL - 1 && have_next_chunk) QGX_H2P_STORE(pv) }
                    __syncthreads();
                }
                if (sl == NSL - 1 && ch == cend - 1) {
                    const int tile_g = blockIdx.x + ti * gridDim.x;
                    const int b = tile_g / tiles_per_img;
                    const int tr = tile_g - b * tiles_per_img;
                    const int y0 = (tr / XT) * R, x0 = (tr % XT) * TW;
                    if constexpr (PART) {
                        // raw partial sums, f32 [split][pixel][cout]: register quad q of a lane = 4 consecutive channels
                        float *pb = reinterpret_cast<float *>(a.out) +
                                    ((size_t)blockIdx.y * a.npix_total + (size_t)b * N * N + (size_t)y0 * N) * COUT;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) {
                                    const f32x4 v = {acc[mt][nt][4 * q4], acc[mt][nt][4 * q4 + 1], acc[mt][nt][4 * q4 + 2], acc[mt][nt][4 * q4 + 3]};
                                    *reinterpret_cast<f32x4 *>(pb + (size_t)((wave + NW * mt) * 32 + li) * COUT + nt * 32 + 8 * q4 + 4 * h) = v;
                                }
                    } else {
                        char *ob = reinterpret_cast<char *>(a.out) + ((size_t)b * N * N + (size_t)y0 * N + x0) * OPIXB;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const int p = (wave + NW * mt) * 32 + li;
                            char *pix = ob + (size_t)((p / TW) * N + p % TW) * OPIXB;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                store_tile_t<2, OUTF32>(acc[mt][nt], nt * 32, h, pix, ep, ep + COUT, ep + 2 * COUT, a.unscale, a.ascale, a.range, a.range_bit);
                        }
                    }
                }
            }
        }
    }
#undef QGX_H2P_LOAD
#undef QGX_H2P_LOAD1
#undef QGX_H2W_LOAD1
#undef QGX_H2P_STORE
#undef QGX_H2W_LOAD
#undef QGX_H2W_STORE
}

// split-K combine of k_convh2<PART>: out = epilogue(sum_s partial[s]) in a fixed order; one thread per pixel
// and octet of channels, output in the packed hi/lo layout (or NHWC f32)
template <int COUT, bool OUTF32>
__global__ void k_convh_reduce(const float *partial, int nsplit, size_t npix_total, const float *bias, const float *scale,
                               const float *shift, float unscale, float ascale, void *out, unsigned *range,
                               unsigned range_bit) {
    const size_t n = npix_total * (COUT / 8);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % (COUT / 8));
        const size_t pix = i / (COUT / 8);
        const float *p0 = partial + pix * COUT + g * 8;
        f32x4 v0 = *reinterpret_cast<const f32x4 *>(p0), v1 = *reinterpret_cast<const f32x4 *>(p0 + 4);
        for (int s2 = 1; s2 < nsplit; ++s2) {
            v0 += *reinterpret_cast<const f32x4 *>(p0 + (size_t)s2 * npix_total * COUT);
            v1 += *reinterpret_cast<const f32x4 *>(p0 + (size_t)s2 * npix_total * COUT + 4);
        }
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = g * 8 + e;
            const float r = (e < 4 ? v0[e] : v1[e - 4]) * unscale + bias[c];
            x[e] = fmaxf(r, 0.f) * scale[c] + shift[c];
        }
        if constexpr (OUTF32) {
            float *o = reinterpret_cast<float *>(out) + pix * COUT + g * 8;
            const f32x4 o0 = {x[0], x[1], x[2], x[3]}, o1 = {x[4], x[5], x[6], x[7]};
            *reinterpret_cast<f32x4 *>(o) = o0;
            *reinterpret_cast<f32x4 *>(o + 4) = o1;
        } else {
            unsigned hw[4], lw[4];
            float mx = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(x[e]));
            range_guard(mx * ascale, range, range_bit);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = x[2 * e] * ascale, a1 = x[2 * e + 1] * ascale;
                const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
                hw[e] = pack_h2((float)h0, (float)h1);
                lw[e] = pack_h2(a0 - (float)h0, a1 - (float)h1);
            }
            char *o = reinterpret_cast<char *>(out) + pix * (COUT * 4) + g * 32;
            const u32x4 oh = {hw[0], hw[1], hw[2], hw[3]}, ol = {lw[0], lw[1], lw[2], lw[3]};
            *reinterpret_cast<u32x4 *>(o) = oh;
            *reinterpret_cast<u32x4 *>(o + 16) = ol;
        }
    }
}

// ---- first layer (n_in = 4 or 2 planar f32 channels -> 128, 5x5) in the f16x3 arithmetic ---------------
// K = 25 taps x n_in channels is tiny, so the whole weight set lives in LDS (57 KB) for the lifetime of
// the persistent workgroup; the input patch is split into hi/lo f16 while it is staged
// ([pixel][n_in hi][n_in lo], 16 or 8 bytes per pixel).  One K = 16 MFMA step covers 4 (n_in = 4) or
// 8 (n_in = 2) taps: lane half h takes taps TPS*s + TPF*h .. + TPF-1; tap slots beyond 25 carry zero
// weights and re-read tap 24.  Each wave computes its MT pixel tiles against the 128 output channels
// in two halves of 64 (accumulators 2 x MT x 16 registers).
struct ConvHFirstArgs {
    const float *in;       // planar (B, NIN, N, N) f32
    void *out;             // [B][N][N][16][2][8] f16
    const void *w;         // [step][part][h][128][8] f16
    const float *bias, *scale, *shift;
    float unscale, ascale;
    float guard_mul;       // 16 when the 5x5 layer behind it is the Winograd form (its input transform amplifies by <= 15)
    int N, R;
    unsigned long long *stamps;   // diagnostic builds only
    unsigned *range;       // range guard flag word, bit of this layer (see range_guard)
    unsigned range_bit;
};

template <int NIN, int MT, int PPT, int NW = 4, bool PLANAR = false>
__global__ __launch_bounds__(NW * 64) void k_convh_first(ConvHFirstArgs a, int total_tiles) {
    constexpr int NTHR = NW * 64;
    constexpr int KS = 5, P = 2, T = 25, COUT = 128;
    constexpr int TPF = 8 / NIN;                 // taps per 8-element fragment
    constexpr int TPS = 2 * TPF;                 // taps per K = 16 step
    constexpr int NSTEP = (T + TPS - 1) / TPS;
    constexpr int PXB = NIN * 4;                 // LDS bytes per patch pixel
    constexpr int WBYTES = NSTEP * 4 * COUT * 16;
    const int N = a.N, R = a.R;
    const int PR = R + KS - 1;
    const int patch_bytes = PR * N * PXB;
    float *const ep = reinterpret_cast<float *>(conv_smem + WBYTES + 2 * patch_bytes);   // bias | scale | shift
    char *const wl0 = conv_smem;
    char *const pl0 = conv_smem + WBYTES;
    const int tiles_per_img = N / R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int n_my = blockIdx.x < (unsigned)total_tiles ? (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int ntiles = R * N / 32;
    const int NX4 = N / 4;
    const int PI = PR * NIN * NX4;               // float4 items of one patch
    if (n_my == 0) return;
    int stamp_i = 0;
    (void)stamp_i;
    QGX_STAMP()

#define QGX_F_LOAD(TI, V)                                                                                   \
    {                                                                                                       \
        const int tile_ = blockIdx.x + (TI) * gridDim.x;                                                    \
        const int b_ = tile_ / tiles_per_img;                                                               \
        const int y0_ = (tile_ - b_ * tiles_per_img) * R;                                                   \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            int it_ = u * NTHR + threadIdx.x;                                                                 \
            it_ = it_ < PI ? it_ : PI - 1;                                                                  \
            const int x4_ = it_ % NX4, c_ = (it_ / NX4) % NIN, pr_ = it_ / (NX4 * NIN);                     \
            int gy_ = y0_ - P + pr_;                                                                        \
            gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                           \
            V[u] = *reinterpret_cast<const f32x4 *>(&a.in[(((size_t)b_ * NIN + c_) * N + gy_) * N + x4_ * 4]); \
        }                                                                                                   \
    }
#define QGX_F_STORE(BUF, V)                                                                                 \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                           \
            if (it_ < PI) {                                                                                 \
                const int x4_ = it_ % NX4, c_ = (it_ / NX4) % NIN, pr_ = it_ / (NX4 * NIN);                 \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                             \
                    const float x_ = V[u][e];                                                               \
                    const _Float16 xh_ = (_Float16)x_;                                                      \
                    _Float16 *px_ = reinterpret_cast<_Float16 *>((BUF) + (pr_ * N + x4_ * 4 + e) * PXB);    \
                    px_[c_] = xh_;                                                                          \
                    px_[NIN + c_] = (_Float16)(x_ - (float)xh_);                                            \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
    }

    for (int i = threadIdx.x; i < 3 * COUT; i += NTHR)
        ep[i] = i < COUT ? a.bias[i] : (i < 2 * COUT ? a.scale[i - COUT] : a.shift[i - 2 * COUT]);
    // ---- prologue: weights (once) and the first patch
    {
        f32x4 pv[PPT], wtmp[(WBYTES / 16 + NTHR - 1) / NTHR];
        QGX_F_LOAD(0, pv)
        QGX_BULK_LOAD(wtmp, a.w, WBYTES / 16, NTHR)
        QGX_F_STORE(pl0, pv)
        QGX_BULK_STORE(wtmp, wl0, WBYTES / 16, NTHR)
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
    int cur = 0;
    QGX_STAMP()
    // The epilogue of one (tile, half) unit — bias / ReLU / hi-lo split on the VALU and 16 output stores per lane — is
    // issued INSIDE the MFMA loop of the next unit, a tile of it after each K step: done back to back, all eight waves
    // of the CU converted and stored at the same time and then all multiplied at the same time, so the matrix pipe, the
    // VALU and the memory pipe (10 B/clk of HBM write rate per CU: 26 k cycles per tile, the kernel's floor) took
    // turns instead of overlapping.  Two accumulator sets (one per half) ping-pong.
    f32x16 acc[2][MT][2];
    char *ob_prev = nullptr;
    auto epilogue_tile = [&](const f32x16 (&ac)[MT][2], int e, char *obase, int half_of) {
        const int mt = e >> 1, nt = e & 1;
        const int tile = wave + NW * mt;
        if (tile >= ntiles) return;
        if constexpr (PLANAR) {
            store_tile_planar(ac[mt][nt], half_of * 64 + nt * 32 + li, h, obase, tile * 32, N, COUT, ep, ep + COUT, ep + 2 * COUT,
                              a.unscale, a.ascale, a.range, a.range_bit, a.guard_mul);
            return;
        }
        char *pix = obase + (size_t)(tile * 32 + li) * (COUT * 4);
        store_tile_t<2, false>(ac[mt][nt], half_of * 64 + nt * 32, h, pix, ep, ep + COUT, ep + 2 * COUT, a.unscale, a.ascale, a.range, a.range_bit, a.guard_mul);
    };
    for (int ti = 0; ti < n_my; ++ti) {
        const bool have_next = ti + 1 < n_my;
        f32x4 pv[PPT];
        QGX_F_LOAD(have_next ? ti + 1 : ti, pv)
        const char *pl = pl0 + cur * patch_bytes;
        const int tile_g = blockIdx.x + ti * gridDim.x;
        const int b = tile_g / tiles_per_img;
        const int y0 = (tile_g - b * tiles_per_img) * R;
        char *ob = reinterpret_cast<char *>(a.out) + ((size_t)b * N * N + (size_t)y0 * N) * (COUT * 4);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // the unit whose results are converted and stored under this unit's MFMAs
            const bool have_prev = half == 1 || ti > 0;
            char *const obp = half == 1 ? ob : ob_prev;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[half][mt][nt][r] = 0.f;
            const char *wl = wl0 + (h * COUT + half * 64 + li) * 16;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                h8 W[2][2];     // [nt][part]
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int part = 0; part < 2; ++part)
                        W[nt][part] = *reinterpret_cast<const h8 *>(wl + ((s * 2 + part) * 2 * COUT + nt * 32) * 16);
                h8 Ph[MT], Pl[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    unsigned hw[4], lw[4];
#pragma unroll
                    for (int f = 0; f < TPF; ++f) {
                        int tap = TPS * s + TPF * h + f;
                        tap = tap < T ? tap : T - 1;
                        const int ky = tap / KS, kx = tap - ky * KS;
                        int col = px[mt] + kx - P;
                        col = col < 0 ? col + N : (col >= N ? col - N : col);
                        const char *src = pl + ((py[mt] + ky) * N + col) * PXB;
                        if constexpr (NIN == 4) {
                            const u32x4 v = *reinterpret_cast<const u32x4 *>(src);
                            hw[2 * f] = v[0]; hw[2 * f + 1] = v[1]; lw[2 * f] = v[2]; lw[2 * f + 1] = v[3];
                        } else {
                            const uint2 v = *reinterpret_cast<const uint2 *>(src);
                            hw[f] = v.x; lw[f] = v.y;
                        }
                    }
                    const u32x4 hv = {hw[0], hw[1], hw[2], hw[3]}, lv = {lw[0], lw[1], lw[2], lw[3]};
                    Ph[mt] = __builtin_bit_cast(h8, hv);
                    Pl[mt] = __builtin_bit_cast(h8, lv);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if constexpr (PLANAR) {     // operands exchanged: the tile comes out transposed, lane = output channel
                            acc[half][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ph[mt], W[nt][1], acc[half][mt][nt], 0, 0, 0);
                            acc[half][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Pl[mt], W[nt][0], acc[half][mt][nt], 0, 0, 0);
                            acc[half][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ph[mt], W[nt][0], acc[half][mt][nt], 0, 0, 0);
                        } else {
                            acc[half][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[nt][1], Ph[mt], acc[half][mt][nt], 0, 0, 0);
                            acc[half][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[nt][0], Pl[mt], acc[half][mt][nt], 0, 0, 0);
                            acc[half][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[nt][0], Ph[mt], acc[half][mt][nt], 0, 0, 0);
                        }
                    }
                if (have_prev) {
#pragma unroll
                    for (int e = 0; e < 2 * MT; ++e)
                        if ((e * NSTEP) / (2 * MT) == s) epilogue_tile(acc[half ^ 1], e, obp, half ^ 1);
                }
            }
            QGX_STAMP()
            // the next patch goes to LDS once its loads have landed.  (gfx9 counts loads and stores in one counter and the
            // compiler must assume they retire out of order, so this wait also drains the stores issued above; moving the
            // prefetch so that no store is in flight at the wait measured the same time: the drain is not what bounds it.)
            if (half == 0 && have_next) QGX_F_STORE(pl0 + (cur ^ 1) * patch_bytes, pv)
        }
        ob_prev = ob;
        __syncthreads();
        QGX_STAMP()
        cur ^= 1;
    }
    // the last unit's results
#pragma unroll
    for (int e = 0; e < 2 * MT; ++e) epilogue_tile(acc[1], e, ob_prev, 1);
#undef QGX_F_LOAD
#undef QGX_F_STORE
}

#ifdef QGX_AB   // resident-weight 3x3 kernel: measured no faster than k_convh2, A/B builds only
// ---- 3x3 hidden layers with RESIDENT weights (f16x3) ------------------------------------------------------
// The 3x3 layers are small (<= 73.7 KB of hi/lo weights), so a persistent 8-wave workgroup keeps the
// whole weight set in LDS and stages only the input patch: 32 channels (128 bytes: 4 octets x hi/lo) per
// pixel and chunk, unit index XOR-swizzled by bits 1..3 of the pixel index (conflict-free
// ds_read_b128 at a 128-byte pixel stride).  The next chunk's / tile's patch is fetched into registers
// at the start of a chunk, i.e. a whole 32-channel chunk (~3.5 us) ahead of its use, which is what the
// 16-channel kernel above lacked on these memory-latency-bound layers; two barriers per chunk.
template <int CIN, int COUT, int MT, int PPT, bool OUTF32>
__global__ __launch_bounds__(512) void k_convh_res(ConvHArgs a, int total_tiles) {
    constexpr int NW = 8, NTHR = 512, KS = 3, P = 1, T = 9;
    constexpr int NT = COUT / 32;
    constexpr int NCH = CIN / 32;
    constexpr int PIXB = CIN * 4;
    constexpr int OPIXB = OUTF32 ? COUT * 4 : COUT * 4;
    constexpr int WBYTES = (CIN / 16) * T * 4 * COUT * 16;
    const int N = a.N, R = a.R;
    const int PR = R + KS - 1;
    const int PU = PR * N * 8;                          // 16-byte units of one patch chunk
    char *const wl0 = conv_smem;
    char *const lds0 = conv_smem + WBYTES;
    float *const ep = reinterpret_cast<float *>(lds0 + PR * N * 128);   // bias | scale | shift
    const char *const inb = reinterpret_cast<const char *>(a.in);
    const int tiles_per_img = N / R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int n_my = blockIdx.x < (unsigned)total_tiles ? (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int ntiles = R * N / 32;
    if (n_my == 0) return;
    int stamp_i = 0;
    (void)stamp_i;
    QGX_STAMP()

#define QGX_RP_LOAD(TI, CH, V)                                                                              \
    {                                                                                                       \
        const int tile_ = blockIdx.x + (TI) * gridDim.x;                                                    \
        const int b_ = tile_ / tiles_per_img;                                                               \
        const int y0_ = (tile_ - b_ * tiles_per_img) * R;                                                   \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            int it_ = u * NTHR + threadIdx.x;                                                                \
            it_ = it_ < PU ? it_ : PU - 1;                                                                  \
            const int un_ = it_ & 7, pl_ = it_ >> 3;                                                        \
            const int pr_ = pl_ / N, x_ = pl_ - pr_ * N;                                                    \
            int gy_ = y0_ - P + pr_;                                                                        \
            gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                           \
            V[u] = *reinterpret_cast<const f32x4 *>(                                                        \
                inb + (((size_t)b_ * N + gy_) * N + x_) * PIXB + (CH) * 128 + un_ * 16);                    \
        }                                                                                                   \
    }
#define QGX_RP_STORE(V)                                                                                     \
    {                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u) {                                                   \
            const int it_ = u * NTHR + threadIdx.x;                                                          \
            if (it_ < PU)                                                                                   \
                *reinterpret_cast<f32x4 *>(lds0 + (it_ >> 3) * 128 + (((it_ & 7) ^ ((it_ >> 4) & 7)) * 16)) = V[u]; \
        }                                                                                                   \
    }

    for (int i = threadIdx.x; i < 3 * COUT; i += NTHR)
        ep[i] = i < COUT ? a.bias[i] : (i < 2 * COUT ? a.scale[i - COUT] : a.shift[i - 2 * COUT]);
    // ---- prologue: the whole weight set (once) and the first patch chunk
    {
        f32x4 pv[PPT], wtmp[(WBYTES / 16 + NTHR - 1) / NTHR];
        QGX_RP_LOAD(0, 0, pv)
        QGX_BULK_LOAD(wtmp, a.w, WBYTES / 16, NTHR)
        QGX_RP_STORE(pv)
        QGX_BULK_STORE(wtmp, wl0, WBYTES / 16, NTHR)
    }
    __syncthreads();
    QGX_STAMP()

    int py[MT], px[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int tile = wave + NW * mt;
        if (tile >= ntiles) tile = wave % ntiles;
        const int p = tile * 32 + li;
        py[mt] = p / N;
        px[mt] = p - py[mt] * N;
    }
    const char *const wlane = wl0 + (h * COUT + li) * 16;
    f32x16 acc[MT][NT];
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
            const bool have_next = nti < n_my;
            f32x4 pv[PPT];
            QGX_RP_LOAD(have_next ? nti : ti, have_next ? nch : ch, pv)

            // ---- K loop: 9 taps x 2 K=16 steps (octet pairs 0-1 / 2-3 of the chunk), one step ahead
            // fragment address of step s = 2 tap + t: hi at base[tap] ^ (t * 64), lo at hi ^ 16
            h8 Pn[MT][2], Wn[NT][2];
#define QGX_RP_FRAGS(S)                                                                                     \
            {                                                                                               \
                constexpr int tap_ = (S) >> 1, t_ = (S) & 1, ky_ = tap_ / 3, kx_ = tap_ - 3 * ky_;          \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                         \
                    int col = px[mt] + kx_ - P;                                                             \
                    col = col < 0 ? col + N : (col >= N ? col - N : col);                                   \
                    const int pl_ = (py[mt] + ky_) * N + col;                                               \
                    const int hi_ = pl_ * 128 + ((((2 * h) ^ ((pl_ >> 1) & 7)) * 16) ^ (t_ * 64));          \
                    Pn[mt][0] = *reinterpret_cast<const h8 *>(lds0 + hi_);                                  \
                    Pn[mt][1] = *reinterpret_cast<const h8 *>(lds0 + (hi_ ^ 16));                           \
                }                                                                                           \
                _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                           \
                    _Pragma("unroll") for (int part = 0; part < 2; ++part)                                  \
                        Wn[nt][part] = *reinterpret_cast<const h8 *>(                                       \
                            wlane + ((size_t)(((ch * 2 + t_) * T + tap_) * 4 + part * 2) * COUT + nt * 32) * 16); \
            }
            QGX_STAMP()
            QGX_RP_FRAGS(0)
#define QGX_RP_STEP(S)                                                                                      \
            {                                                                                               \
                h8 Pc[MT][2], Wc[NT][2];                                                                    \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) { Pc[mt][0] = Pn[mt][0]; Pc[mt][1] = Pn[mt][1]; } \
                _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { Wc[nt][0] = Wn[nt][0]; Wc[nt][1] = Wn[nt][1]; } \
                if constexpr ((S) + 1 < 2 * T) QGX_RP_FRAGS((S) + 1)                                        \
                __builtin_amdgcn_sched_barrier(0);                                                          \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                           \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                     \
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][1], Pc[mt][0], acc[mt][nt], 0, 0, 0); \
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][0], Pc[mt][1], acc[mt][nt], 0, 0, 0); \
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][0], Pc[mt][0], acc[mt][nt], 0, 0, 0); \
                    }                                                                                       \
            }
            QGX_RP_STEP(0) QGX_RP_STEP(1) QGX_RP_STEP(2) QGX_RP_STEP(3) QGX_RP_STEP(4) QGX_RP_STEP(5)
            QGX_RP_STEP(6) QGX_RP_STEP(7) QGX_RP_STEP(8) QGX_RP_STEP(9) QGX_RP_STEP(10) QGX_RP_STEP(11)
            QGX_RP_STEP(12) QGX_RP_STEP(13) QGX_RP_STEP(14) QGX_RP_STEP(15) QGX_RP_STEP(16) QGX_RP_STEP(17)
#undef QGX_RP_STEP
#undef QGX_RP_FRAGS
            QGX_STAMP()

            QGX_STAMP()
            __syncthreads();                             // every wave is done with this chunk's patch
            QGX_STAMP()
            if (have_next) QGX_RP_STORE(pv)
            QGX_STAMP()
            __syncthreads();
            // epilogue AFTER the prefetch was retired: its stores drain during the next chunk's K loop
            if (ch == NCH - 1) {
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
                        store_tile_t<2, OUTF32>(acc[mt][nt], nt * 32, h, pix, ep, ep + COUT, ep + 2 * COUT, a.unscale, a.ascale, a.range, a.range_bit);
                }
            }
            QGX_STAMP()
        }
    }
#undef QGX_RP_LOAD
#undef QGX_RP_STORE
}
#endif  // QGX_AB
