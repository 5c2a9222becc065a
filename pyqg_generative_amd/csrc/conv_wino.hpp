// Generator layer 2 (128 -> 64 channels, 5x5, circular padding; 75 % of the step's multiply-adds) as a ONE-dimensional
// Toom-Cook / Winograd convolution F(4, 5) along x, direct along y, in the f16x3 arithmetic of conv_half.hpp.
// Included by conv.hip (inside namespace qgx, after conv_half.hpp).
//
// Replaces, for this one layer, the 25-tap implicit GEMM of k_convh2 (AndrewCNN.forward, cnn_tools.py:125-176; the
// arithmetic the reference evaluates is torch's float32 conv2d with padding_mode='circular', cnn_tools.py:79-98).
//
//   y(r, 4t + j, o) = sum_p AT[j][p] M_p(r, t, o),      M_p(r, t, o) = sum_ky sum_c U_p,ky(o, c) V_p(r + ky - 2, t, c)
//   V_p(row, t, c)  = sum_k BT[p][k] x(row, 4t - 2 + k, c)        (8 positions p per quad t of 4 output columns)
//   U_p,ky(o, c)    = sum_kx G[p][kx] w(o, c, ky, kx)            (float64 on the host, qgx_generator_create)
//
// 8 positions x 5 rows = 40 multiplications per 4 outputs instead of 100: 0.4 x the MFMAs of the direct form, and only
// a 2 x expansion of the transformed operand (the nested 2-D form F(2x2, 5x5) needs 0.36 x but a 9 x expansion: its 36
// accumulators per output quad and 1.18 MB of transformed weights per 128 pixels exceed what a CU's register file and the
// L2 -> CU path can hold / stream).  Evaluation points 0, +-1, +-2, +-1/2, infinity.  Measured
// against a float64 evaluation of the same float32 parameters (tests/test_conv_transform_numerics_cpu.py): 6.8e-7 of
// max|y| for this layer with the f16x3 split applied AFTER the input transform — the float32 error class (direct float32:
// 2.2e-7; the 25-tap f16x3 kernel: same class).
//
// One workgroup = 8 waves owns a tile of R rows x TW columns (512 pixels: 8 x 64 at 64 x 64, 16 x 32 at 96 x 96 and
// 32 x 32, 8 x 64 at 128 x 128; 16 x 16 at 48 x 48); wave p owns POSITION p: its A operand U_p,ky comes straight
// from global memory / L2 in fragment layout (no other wave uses it, so staging it through LDS would only add barriers),
// its B operand V_p from the transformed patch in LDS ([p][row][quad] records of 64 payload + 16 pad bytes), its
// accumulators are M_p of the whole tile (4 M-tiles of 32 (row, quad) pairs x 64 output channels = 128 registers).
// Per 16-channel chunk: the input transform (float32: x = hi + lo, BT x, split to hi / lo) of the (R + 4)-row patch by 384
// of the 512 threads, straight from global memory, then 5 row offsets x 24 MFMAs per wave.  After the last chunk the
// output transform AT (positions live in different waves) goes through LDS one M-tile at a time, fused with the layer's
// epilogue (bias, ReLU, BatchNorm affine, hi / lo split, range guard).
#pragma once

struct ConvWArgs {
    const void *in;        // [B][N][N][128/8][hi|lo][8] f16 (layer 1's output)
    void *out;             // [B][N][N][64/8][hi|lo][8] f16
    const void *w;         // [chunk 8][ky 5][p 8][part 2][h 2][cout 64][8] f16: U_p,ky pre-scaled per position, hi / lo
    const float *bias, *scale, *shift;
    float pscale[8];       // u_p = 1 / (weight pre-scale of position p x input activation pre-scale), powers of two
    float ascale;          // pre-scale of the stored output activations
    unsigned *range;       // f16x3 range guard
    unsigned range_bit;
};

// 1-D input transform, one channel: x[k] (8 pixels, float32) -> V[p] (8 positions), BT of F(4,5)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// float(hi[SEL]) + float(lo[SEL]) of two packed f16 pairs in ONE instruction (v_fma_mix_f32: f16 sources, f32 result;
// the compiler emits two conversions and an add), and v - float(hp[SEL]) likewise
template <int SEL>
__device__ __forceinline__ float mix_sum(unsigned hi, unsigned lo) {
    float r;
    if constexpr (SEL == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi), "v"(lo));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi), "v"(lo));
    return r;
}
template <int SEL>
__device__ __forceinline__ float mix_rest(unsigned hp, float v) {
    float r;
    if constexpr (SEL == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
    return r;
}

__device__ __forceinline__ void wino_bt8(const float (&x)[8][4], int e, float (&v)[8]) {
    const float x0 = x[0][e], x1 = x[1][e], x2 = x[2][e], x3 = x[3][e], x4 = x[4][e], x5 = x[5][e], x6 = x[6][e], x7 = x[7][e];
    v[0] = fmaf(5.25f, x2 - x4, x6) - x0;
    v[7] = fmaf(5.25f, x3 - x5, x7) - x1;
    const float a1 = fmaf(-4.25f, x4, x2 + x6), b1 = fmaf(-4.25f, x3, x1 + x5);
    v[1] = a1 + b1; v[2] = a1 - b1;
    const float a3 = fmaf(-1.25f, x4, fmaf(0.25f, x2, x6)), b3 = fmaf(2.f, x5, fmaf(-2.5f, x3, 0.5f * x1));
    v[3] = a3 + b3; v[4] = a3 - b3;
    const float a5 = fmaf(-5.f, x4, fmaf(4.f, x2, x6)), b5 = fmaf(0.5f, x5, fmaf(-2.5f, x3, 2.f * x1));
    v[5] = a5 + b5; v[6] = a5 - b5;
}

// EXP (A/B library, timing experiments only — the results are wrong; 6 no output stores, 7 = 1 + 2, 8 = 2 + 6): 1 no input transform, 2 no MFMAs, 4 weights loaded
// once, 5 no raw-patch copy
template <int NN, int TW, int R, int EXP = 0, bool PL = false>
__global__ __launch_bounds__(512) void k_convw(ConvWArgs a, int total_tiles) {
    constexpr int N = NN, CIN = 128, COUT = 64, NCH = CIN / 16, KY = 5;
    constexpr int NQT = TW / 4;                     // quads per tile row
    constexpr int RM = 32 / NQT;                    // rows per M-tile of 32 (row, quad) pairs
    constexpr int MT = R / RM, PR = R + 4;          // M-tiles, patch rows of a tile
    constexpr int XT = N / TW;
    constexpr bool FULLW = TW == N;                 // full-width tiles: the x halo is the row itself, wrapped
    constexpr int PW = FULLW ? N : TW + 4;          // patch columns held in LDS: x0 - 2 ... x0 + TW + 1 (wrapped)
    constexpr int SQ = PW / 4;                      // quads of the raw patch: column slot = SQ (xl & 3) + (xl >> 2)
    constexpr int PIXB = CIN * 4, OPIXB = COUT * 4;
    // LDS: the transformed patch VT[p][row][quad] in 64-byte records whose four 16-byte units (octet 0 hi | lo, octet 1
    // hi | lo) are XOR-swizzled by bits 2..3 of the record's pair index row * NQT + quad (conflict-free ds_read_b128 of 16
    // consecutive pairs without padding); the RAW patch of the NEXT chunk, transposed to [row][unit][column slot] with
    // slot = SQ (xl & 3) + (xl >> 2) for the local column xl = x - (x0 - 2) (full-width tiles: xl = x), so that the
    // transform's reads of consecutive quads are consecutive 16-byte slots
    // PL (A/B library only — measured, not faster overall: conv.hip::wino_planar; planar input, conv_half.hpp::store_tile_planar: [b][y][c][hi | lo][x] f16): the input transform runs on the
    // matrix cores.  BT contracts over PIXELS, so with pixel-contiguous rows the raw patch feeds an MFMA directly:
    // A = activations, 32 rows = (2 patch rows) x (16 channels), K = 32 pixels of the window 16 Q - 8 ... 16 Q + 23 around the
    // quad group Q (two K = 16 steps, hi and lo: 4 MFMAs), B = the constant banded matrix BT shifted per quad
    // (32 columns = 4 quads x 8 positions; every entry exact in f16).  The accumulator tile comes out with lane =
    // (position, quad), register = channel: one 64-byte record of the transformed patch per lane after the hi / lo split.
    // Raw patch in LDS: [row][channel][hi | lo][XO octets of 8 pixels] (window x0 - 8 ... x0 + TW + 7, or the whole wrapped
    // row), RS bytes per (row, channel) and RYS per row chosen so that the A-fragment reads are bank-conflict-free.
    constexpr int XO = FULLW ? N / 8 : (TW + 16) / 8;
    constexpr int RS = 2 * XO * 16 + 16, RYS = 16 * RS + 128;
    static_assert(!PL || !FULLW || (XO & (XO - 1)) == 0, "wrapped octet index");
    constexpr int REC = 64;
    constexpr int VPS = PR * NQT * REC;             // bytes between positions (PL: a record's unit index is XORed with p & 3 as well,
                                                    // so that the transform's record writes of four positions spread over the banks)
    constexpr int VT_BYTES = 8 * VPS;
    constexpr int MREC = COUT * 4 + 16;             // output staging: bytes per (p, pair) record
    constexpr int A_BYTES = VT_BYTES > 8 * 32 * MREC ? VT_BYTES : 8 * 32 * MREC;
    constexpr int RAW_BYTES = PL ? PR * RYS : PR * 4 * PW * 16;
    constexpr int NITEM = PR * NQT * 2;             // transform items (row, quad, octet) per chunk
    constexpr int NUNIT = PL ? PR * 32 * XO : PR * PW * 4;   // 16-byte units of a chunk's raw patch
    constexpr int UPT = (NUNIT + 511) / 512;        // ... per thread
    static_assert(32 % NQT == 0 && R % RM == 0 && N % TW == 0 && N % R == 0 && NITEM <= 512 && UPT <= 8, "tile shape");
    static_assert(!PL || (PR % 2 == 0 && NQT % 4 == 0 && (PR / 2) * (NQT / 4) >= 8), "MFMA input transform: row pairs x quad groups");
    char *const vt = conv_smem;
    char *const rawb = conv_smem + A_BYTES;
    float *const ep = reinterpret_cast<float *>(conv_smem + A_BYTES + RAW_BYTES);         // bias | scale | shift | AT'[4][8]
    const char *const inb = reinterpret_cast<const char *>(a.in);
    const char *const wb = reinterpret_cast<const char *>(a.w);
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;                 // wave p owns position p
    const int li = lane & 31, h = lane >> 5;
    constexpr int tiles_per_img = (N / R) * XT;
    const int n_my = blockIdx.x < (unsigned)total_tiles ? (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (n_my == 0) return;
    for (int i = threadIdx.x; i < 3 * COUT + 32; i += 512)
        ep[i] = i < COUT ? a.bias[i] : (i < 2 * COUT ? a.scale[i - COUT] : (i < 3 * COUT ? a.shift[i - 2 * COUT] : a.pscale[(i - 3 * COUT) & 7]));

    // transform work items: (row, quad, octet, half of the octet's 8 channels); lanes run over the quads first (consecutive
    // slots of the raw patch).  NHALF = 2 NITEM half-items over 512 threads in two rounds: at 64 x 64 waves 0..3 take two,
    // waves 4..7 one, i.e. 1.5 item-times per SIMD (whole items on waves 0..5 made it 2 on SIMDs 0 and 1)
    constexpr int NHALF = 2 * NITEM;
    // B fragment of this lane: pair li = (row li / NQT, quad li % NQT) of an M-tile, octet h (units 2h | 2h + 1, swizzled)
    const int fq = li % NQT, fr = li / NQT;
    const int psw = PL ? (p & 3) : 0;
    const int fbase = p * VPS + (fr * NQT + fq) * REC + (((2 * h) ^ psw ^ (((fr * NQT + fq) >> 2) & 3)) * 16);
    auto frag = [&](int row_off) -> int {        // byte offset of the hi unit of patch row fr + row_off; the lo unit is ^ 16
        if constexpr (NQT == 16) return fbase + row_off * NQT * REC;                         // swizzle: the quad alone
        else if constexpr (NQT == 8) return (fbase + row_off * NQT * REC) ^ ((row_off & 1) * 32);   // ... and the row's parity
        else {
            const int prow = fr + row_off;
            return p * VPS + (prow * NQT + fq) * REC + (((2 * h) ^ psw ^ (((prow * NQT + fq) >> 2) & 3)) * 16);
        }
    };
    // A-fragment base: [p][part][h][cout][8] within a (chunk, ky) slice of 8 * 4 * 64 * 16 bytes
    const int wofs = (p * 4 + h) * COUT * 16 + li * 16;
    constexpr int WSLICE = 8 * 4 * COUT * 16;

    // ---- raw patch of (tile, chunk): unit j of this thread, global -> register -> LDS ----
    // unit u = j * 512 + tid = ((row * PW + xl) * 4 + unit-in-pixel): consecutive lanes read the 64 contiguous bytes of a
    // pixel's chunk and consecutive pixels.  (LDS-DMA would need no registers, but while one is in flight hipcc drains
    // vmcnt(0) at every use of an ordinary load — here the weight fragments of every block.)
#define QGX_RAW_LOAD(J, TILE, CH, DST)                                                                          \
    {                                                                                                           \
        int u_ = (J) * 512 + (int)threadIdx.x;                                                                  \
        u_ = u_ < NUNIT ? u_ : NUNIT - 1;                                                                       \
        const int b_ = (TILE) / tiles_per_img, tr_ = (TILE) - b_ * tiles_per_img;                               \
        if constexpr (PL) {                                                                                     \
            /* unit = (row, channel, hi | lo, octet): consecutive lanes read consecutive 16-byte octets of a 2 N-byte plane row */ \
            const int oc_ = u_ % XO, hl_ = (u_ / XO) & 1, c_ = (u_ / (2 * XO)) & 15, r_ = u_ / (32 * XO);         \
            int gy_ = (tr_ / XT) * R - 2 + r_, go_ = FULLW ? oc_ : (tr_ % XT) * (TW / 8) - 1 + oc_;             \
            gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                               \
            go_ = go_ < 0 ? go_ + N / 8 : (go_ >= N / 8 ? go_ - N / 8 : go_);                                   \
            DST = *reinterpret_cast<const f32x4 *>(inb + (((((size_t)b_ * N + gy_) * CIN + (CH) * 16 + c_) * 2 + hl_) * N + go_ * 8) * 2); \
        } else {                                                                                                \
            const int un_ = u_ & 3, xl_ = (u_ >> 2) % PW, r_ = u_ / (4 * PW);                                   \
            int gy_ = (tr_ / XT) * R - 2 + r_, gx_ = FULLW ? xl_ : (tr_ % XT) * TW - 2 + xl_;                   \
            gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                               \
            gx_ = gx_ < 0 ? gx_ + N : (gx_ >= N ? gx_ - N : gx_);                                               \
            DST = *reinterpret_cast<const f32x4 *>(inb + (((size_t)b_ * N + gy_) * N + gx_) * PIXB + (CH) * 64 + un_ * 16); \
        }                                                                                                       \
    }
#define QGX_RAW_STORE(J, SRC)                                                                                   \
    {                                                                                                           \
        const int u_ = (J) * 512 + (int)threadIdx.x;                                                            \
        if (u_ < NUNIT) {                                                                                       \
            if constexpr (PL) {                                                                                 \
                const int oc_ = u_ % XO, hl_ = (u_ / XO) & 1, c_ = (u_ / (2 * XO)) & 15, r_ = u_ / (32 * XO);     \
                *reinterpret_cast<f32x4 *>(rawb + r_ * RYS + c_ * RS + (hl_ * XO + oc_) * 16) = SRC;            \
            } else {                                                                                            \
                const int un_ = u_ & 3, xl_ = (u_ >> 2) % PW, r_ = u_ / (4 * PW);                               \
                *reinterpret_cast<f32x4 *>(rawb + (((r_ * 4 + un_) * PW) + (xl_ & 3) * SQ + (xl_ >> 2)) * 16) = SRC;   \
            }                                                                                                   \
        }                                                                                                       \
    }

    f32x16 acc[MT][2];
    h8 Wn[2][2];                                    // [nt][part] of the NEXT (chunk, ky) step, prefetched from L2
#define QGX_W_LOAD(S)                                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
            Wn[nt][j] = *reinterpret_cast<const h8 *>(wb + (size_t)(S) * WSLICE + wofs + j * 2 * COUT * 16 + nt * 32 * 16);
    QGX_W_LOAD(0)
    // PL: the B operand of the transform, column n = li = (position li >> 2, quad-in-group li & 3), K index k = 16 ks + 8 h + e
    // = pixel 16 Q - 8 + k; quad j reads the pixels 16 Q + 4 j - 2 ... + 5, i.e. d = k - 6 - 4 j in 0..7
    h8 Bt[2];
    if constexpr (PL) {
        constexpr float BT8[8][8] = {{-1.f, 0.f, 5.25f, 0.f, -5.25f, 0.f, 1.f, 0.f},   {0.f, 1.f, 1.f, -4.25f, -4.25f, 1.f, 1.f, 0.f},
                                     {0.f, -1.f, 1.f, 4.25f, -4.25f, -1.f, 1.f, 0.f},  {0.f, .5f, .25f, -2.5f, -1.25f, 2.f, 1.f, 0.f},
                                     {0.f, -.5f, .25f, 2.5f, -1.25f, -2.f, 1.f, 0.f},  {0.f, 2.f, 4.f, -2.5f, -5.f, .5f, 1.f, 0.f},
                                     {0.f, -2.f, 4.f, 2.5f, -5.f, -.5f, 1.f, 0.f},     {0.f, -1.f, 0.f, 5.25f, 0.f, -5.25f, 0.f, 1.f}};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = 16 * ks + 8 * h + e - 6 - 4 * (li & 3);
                const float bv = d >= 0 && d < 8 ? BT8[li >> 2][d & 7] : 0.f;
                Bt[ks][e] = (_Float16)bv;
            }
    }
    {   // prologue: the first tile's first chunk, synchronously
        f32x4 r0[UPT];
#pragma unroll
        for (int j = 0; j < UPT; ++j) QGX_RAW_LOAD(j, (int)blockIdx.x, 0, r0[j])
#pragma unroll
        for (int j = 0; j < UPT; ++j) QGX_RAW_STORE(j, r0[j])
    }

    for (int ti = 0; ti < n_my; ++ti) {
        const int tile_g = blockIdx.x + ti * gridDim.x;
        const int b = tile_g / tiles_per_img, tr = tile_g - b * tiles_per_img;
        const int y0 = (tr / XT) * R, x0 = (tr % XT) * TW;
        for (int ch = 0; ch < NCH; ++ch) {
            // ---- input transform of this chunk: raw patch (LDS) -> float32 BT -> hi / lo -> transformed patch (LDS) ----
            __syncthreads();        // the raw patch has landed; every wave is done reading the previous transformed patch
            if constexpr (PL) {
                constexpr int NQG = NQT / 4, NTT = (PR / 2) * NQG;                      // (row pair, quad group) tiles of 32 x 32
                if (EXP != 1 && EXP != 7) {
                    const int ayl = (li >> 2) & 1, ac = (li & 3) + 4 * (li >> 3);         // A row li = (patch row parity, channel)
                    // one tile at a time (the 128 accumulator registers of the tile stay live through this phase), the next
                    // tile's four fragments in flight under this tile's MFMAs and hi / lo split.  (Software-pipelining the
                    // chains of two tiles needs 16 more registers: 324 bytes of scratch at 64 x 64.)
                    h8 An[4];
#define QGX_TA_LOAD(TT)                                                                                         \
                    {                                                                                           \
                        const int rp_ = (TT) / NQG, Q_ = (TT) - rp_ * NQG;                                      \
                        const char *rowp_ = rawb + (2 * rp_ + ayl) * RYS + ac * RS;                             \
                        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                      \
                            const int oc_ = FULLW ? ((2 * Q_ - 1 + 2 * ks + h) & (XO - 1)) : 2 * Q_ + 2 * ks + h; \
                            An[ks] = *reinterpret_cast<const h8 *>(rowp_ + oc_ * 16);                           \
                            An[2 + ks] = *reinterpret_cast<const h8 *>(rowp_ + (XO + oc_) * 16);                \
                        }                                                                                       \
                    }
                    QGX_TA_LOAD(p)
#pragma nounroll
                    for (int tt = p; tt < NTT; tt += 8) {                                  // wave-uniform
                        const h8 A0 = An[0], A1 = An[1], A2 = An[2], A3 = An[3];
                        if (tt + 8 < NTT) QGX_TA_LOAD(tt + 8)
                        // two independent chains (hi, lo) of two MFMAs instead of one of four
                        f32x16 tv, tl;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { tv[r] = 0.f; tl[r] = 0.f; }
                        tv = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, Bt[0], tv, 0, 0, 0);
                        tl = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, Bt[0], tl, 0, 0, 0);
                        tv = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, Bt[1], tv, 0, 0, 0);
                        tl = __builtin_amdgcn_mfma_f32_32x32x16_f16(A3, Bt[1], tl, 0, 0, 0);
                        // lane (li, h): position li >> 2, quad 4 Q + (li & 3), patch row 2 rp + h; register r = channel r
                        const int rp = tt / NQG, Q = tt - rp * NQG;
                        const int pair = (2 * rp + h) * NQT + 4 * Q + (li & 3);
                        char *dst = vt + (li >> 2) * VPS + pair * REC;
                        const int sw = ((pair >> 2) ^ (li >> 2)) & 3;
#pragma unroll
                        for (int o = 0; o < 2; ++o) {
                            unsigned hw[4], lw[4];
#pragma unroll
                            for (int e2 = 0; e2 < 4; ++e2) {
                                const float v0 = tv[8 * o + 2 * e2] + tl[8 * o + 2 * e2], v1 = tv[8 * o + 2 * e2 + 1] + tl[8 * o + 2 * e2 + 1];
                                hw[e2] = pack_h2(v0, v1);
                                lw[e2] = pack_h2(mix_rest<0>(hw[e2], v0), mix_rest<1>(hw[e2], v1));
                            }
                            const int uh = ((2 * o) ^ sw) * 16;
                            const u32x4 oh = {hw[0], hw[1], hw[2], hw[3]}, ol = {lw[0], lw[1], lw[2], lw[3]};
                            *reinterpret_cast<u32x4 *>(dst + uh) = oh;
                            *reinterpret_cast<u32x4 *>(dst + (uh ^ 16)) = ol;
                        }
                    }
#undef QGX_TA_LOAD
                }
            } else {
#pragma nounroll
            for (int rep = 0; rep < (NHALF + 511) / 512; ++rep) {
                const int hi_ = rep * 512 + (int)threadIdx.x;
                if (hi_ < NHALF && EXP != 1 && EXP != 7) {
                    const int it_t = hi_ % NQT, it_o = (hi_ / NQT) & 1, hf = (hi_ / (2 * NQT)) & 1, it_r = hi_ / (4 * NQT);
                    const int it_sw = ((it_r * NQT + it_t) >> 2) & 3;
                    const char *src = rawb + (size_t)(it_r * 4 + it_o * 2) * PW * 16;
                    char *dst = vt + ((size_t)it_r * NQT + it_t) * REC;
                    const int uh = ((2 * it_o) ^ it_sw) * 16, ul = uh ^ 16;
                    u32x2 raw[8][2];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        // x-tiled: local column xl = 4 t + k -> slot SQ (k & 3) + t + (k >> 2); full width: x = 4 t - 2 + k
                        // wrapped -> slot SQ ((k + 2) & 3) + (t - 1 | t | t + 1 mod NQT)
                        const int sl = FULLW ? ((k + 2) & 3) * SQ + ((it_t + (k < 2 ? NQT - 1 : (k < 6 ? 0 : 1))) & (NQT - 1))
                                             : (k & 3) * SQ + it_t + (k >> 2);
                        raw[k][0] = *reinterpret_cast<const u32x2 *>(src + sl * 16 + hf * 8);
                        raw[k][1] = *reinterpret_cast<const u32x2 *>(src + (PW + sl) * 16 + hf * 8);
                    }
                    float x[8][4];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        x[k][0] = mix_sum<0>(raw[k][0][0], raw[k][1][0]); x[k][1] = mix_sum<1>(raw[k][0][0], raw[k][1][0]);
                        x[k][2] = mix_sum<0>(raw[k][0][1], raw[k][1][1]); x[k][3] = mix_sum<1>(raw[k][0][1], raw[k][1][1]);
                    }
                    float v[4][8];                             // [channel e][position]
#pragma unroll
                    for (int e = 0; e < 4; ++e) wino_bt8(x, e, v[e]);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        unsigned hw[2], lw[2];
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const float v0 = v[2 * e2][q], v1 = v[2 * e2 + 1][q];
                            hw[e2] = pack_h2(v0, v1);
                            lw[e2] = pack_h2(mix_rest<0>(hw[e2], v0), mix_rest<1>(hw[e2], v1));
                        }
                        const u32x2 oh = {hw[0], hw[1]}, ol = {lw[0], lw[1]};
                        *reinterpret_cast<u32x2 *>(dst + (size_t)q * PR * NQT * REC + uh + hf * 8) = oh;
                        *reinterpret_cast<u32x2 *>(dst + (size_t)q * PR * NQT * REC + ul + hf * 8) = ol;
                    }
                }
            }
            }
            __syncthreads();
            if (ch == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
            }
            // the raw patch of the NEXT chunk (or of the next tile's first chunk) rides through this chunk's MFMA phase
            const bool more = ch + 1 < NCH || ti + 1 < n_my;
            const int n_tile = ch + 1 < NCH ? tile_g : tile_g + (int)gridDim.x, n_ch = ch + 1 < NCH ? ch + 1 : 0;
            f32x4 rw0, rw1;
            // ---- 5 row offsets x MT M-tiles x 2 output-channel tiles x 3 MFMAs ----
#pragma unroll
            for (int ky = 0; ky < (EXP == 2 || EXP == 7 || EXP == 8 ? 0 : KY); ++ky) {
                h8 Wc[2][2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) { Wc[nt][0] = Wn[nt][0]; Wc[nt][1] = Wn[nt][1]; }
                {
                    int s = ch * KY + ky + 1;
                    if (s == NCH * KY) s = 0;                  // the next tile starts over
                    if (EXP != 4) QGX_W_LOAD(s)
                }
                // raw copy, two units per thread at a time: loaded in blocks 0, 2 and 4, stored at the END of blocks 1, 3 and 4 —
                // two blocks (~1.6 us) of flight for four of the six units instead of one for all of them (a store at the
                // end of the block that issued the load made the wave wait out the HBM latency there); the same 8 registers
                if (more && EXP != 5) {                        // AFTER the weight loads: vmcnt retires in order
                    if (ky == 0) { QGX_RAW_LOAD(0, n_tile, n_ch, rw0) if (UPT > 1) QGX_RAW_LOAD(1, n_tile, n_ch, rw1) }
                    if (ky == 2 && UPT > 2) { QGX_RAW_LOAD(2, n_tile, n_ch, rw0) if (UPT > 3) QGX_RAW_LOAD(3, n_tile, n_ch, rw1) }
                    if (ky == 4 && UPT > 4) { QGX_RAW_LOAD(4, n_tile, n_ch, rw0) if (UPT > 5) QGX_RAW_LOAD(5, n_tile, n_ch, rw1) }
                }
                __builtin_amdgcn_sched_barrier(0);             // hipcc otherwise sinks the prefetches to their use
                h8 Pn[2];
                {
                    const int f0 = frag(ky);
                    Pn[0] = *reinterpret_cast<const h8 *>(vt + f0);
                    Pn[1] = *reinterpret_cast<const h8 *>(vt + (f0 ^ 16));
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const h8 Ph = Pn[0], Pl = Pn[1];
                    if (mt + 1 < MT) {
                        const int f1 = frag((mt + 1) * RM + ky);
                        Pn[0] = *reinterpret_cast<const h8 *>(vt + f1);
                        Pn[1] = *reinterpret_cast<const h8 *>(vt + (f1 ^ 16));
                    }
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][1], Ph, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][0], Pl, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wc[nt][0], Ph, acc[mt][nt], 0, 0, 0);
                    }
                }
                if (more && EXP != 5) {
                    if (ky == 1) { QGX_RAW_STORE(0, rw0) if (UPT > 1) QGX_RAW_STORE(1, rw1) }
                    if (ky == 3 && UPT > 2) { QGX_RAW_STORE(2, rw0) if (UPT > 3) QGX_RAW_STORE(3, rw1) }
                    if (ky == 4 && UPT > 4) { QGX_RAW_STORE(4, rw0) if (UPT > 5) QGX_RAW_STORE(5, rw1) }
                    if (ky == 4 && UPT > 6) {
                        f32x4 rw2, rw3;
                        QGX_RAW_LOAD(6, n_tile, n_ch, rw2)
                        if (UPT > 7) QGX_RAW_LOAD(7, n_tile, n_ch, rw3)
                        QGX_RAW_STORE(6, rw2)
                        if (UPT > 7) QGX_RAW_STORE(7, rw3)
                    }
                }
            }
        }
        // ---- output transform + epilogue, one M-tile at a time through the (now free) patch region ----
        char *const ob = reinterpret_cast<char *>(a.out) + (((size_t)b * N + y0) * N + x0) * OPIXB;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            __syncthreads();                                   // patch reads / previous staging reads are over
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 vv = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
                    *reinterpret_cast<f32x4 *>(vt + (size_t)(p * 32 + li) * MREC + (nt * 32 + 8 * q + 4 * h) * 4) = vv;
                }
            __syncthreads();
            {
                // item (pair pl, octet g, half hf): the FOUR outputs of the quad for four output channels, every M_p read
                // once.  AT by its structure (points 0, +-1, +-2, +-1/2, inf) on the pre-scaled products u_p M_p
                // (u_p: powers of two, exact): 19 operations per 4 outputs instead of a dense 4 x 8 product's 32
                const int hf = threadIdx.x & 1, g = (threadIdx.x >> 1) & 7, pl = threadIdx.x >> 4;
                const float *const u = ep + 3 * COUT;
                f32x4 m[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    m[q] = *reinterpret_cast<const f32x4 *>(vt + (size_t)(q * 32 + pl) * MREC + g * 32 + hf * 16);
                float y[4][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t1 = u[1] * m[1][e], t2 = u[3] * m[3][e], t3 = u[5] * m[5][e];
                    const float s1 = fmaf(u[2], m[2][e], t1), d1 = fmaf(-u[2], m[2][e], t1);
                    const float s2 = fmaf(u[4], m[4][e], t2), d2 = fmaf(-u[4], m[4][e], t2);
                    const float s3 = fmaf(u[6], m[6][e], t3), d3 = fmaf(-u[6], m[6][e], t3);
                    y[0][e] = fmaf(u[0], m[0][e], (s1 + s2) + s3);
                    y[1][e] = fmaf(.5f, d3, fmaf(2.f, d2, d1));
                    y[2][e] = fmaf(.25f, s3, fmaf(4.f, s2, s1));
                    y[3][e] = fmaf(u[7], m[7][e], fmaf(.125f, d3, fmaf(8.f, d2, d1)));
                }
                const int c0 = g * 8 + hf * 4;
                const f32x4 bi = *reinterpret_cast<const f32x4 *>(ep + c0);
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(ep + COUT + c0);
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(ep + 2 * COUT + c0);
                float mx = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        y[j][e] = fmaxf(y[j][e] + bi[e], 0.f) * sc[e] + sh[e];
                        mx = fmaxf(mx, fabsf(y[j][e]));
                    }
                range_guard(mx * a.ascale, a.range, a.range_bit);
                const int row = mt * RM + pl / NQT, col = 4 * (pl % NQT);
                char *o = ob + ((size_t)row * N + col) * OPIXB + g * 32 + hf * 8;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned hw[2], lw[2];
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const float v0 = y[j][2 * e2] * a.ascale, v1 = y[j][2 * e2 + 1] * a.ascale;
                        hw[e2] = pack_h2(v0, v1);
                        lw[e2] = pack_h2(mix_rest<0>(hw[e2], v0), mix_rest<1>(hw[e2], v1));
                    }
                    const u32x2 oh = {hw[0], hw[1]}, ol = {lw[0], lw[1]};
                    if ((EXP == 6 || EXP == 8) && mx >= 0.f) continue;       // experiment: no output stores
                    *reinterpret_cast<u32x2 *>(o + (size_t)j * OPIXB) = oh;
                    *reinterpret_cast<u32x2 *>(o + (size_t)j * OPIXB + 16) = ol;
                }
            }
        }
    }
#undef QGX_W_LOAD
#undef QGX_RAW_LOAD
#undef QGX_RAW_STORE
}

// LDS bytes of k_convw<NN, TW, R, ., PL>
constexpr size_t convw_lds_bytes(int NN, int TW, int R, bool PL = false) {
    const size_t xo = TW == NN ? NN / 8 : (TW + 16) / 8;
    const size_t vtb = (size_t)8 * (R + 4) * (TW / 4) * 64, st = (size_t)8 * 32 * (64 * 4 + 16);
    const size_t raw = PL ? (size_t)(R + 4) * (16 * (2 * xo * 16 + 16) + 128) : (size_t)(R + 4) * 4 * (TW == NN ? NN : TW + 4) * 16;
    return (vtb > st ? vtb : st) + raw + (3 * 64 + 32) * sizeof(float);
}
