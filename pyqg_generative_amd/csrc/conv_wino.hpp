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
    float atp[4][8];       // AT[j][p] / (weight pre-scale of position p x input activation pre-scale)
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

// EXP (A/B library, timing experiments only — the results are wrong): 1 no input transform, 2 no MFMAs, 4 weights loaded
// once, 5 no raw-patch copy
template <int NN, int TW, int R, int EXP = 0>
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
    constexpr int REC = 64;
    constexpr int VT_BYTES = 8 * PR * NQT * REC;
    constexpr int MREC = COUT * 4 + 16;             // output staging: bytes per (p, pair) record
    constexpr int A_BYTES = VT_BYTES > 8 * 32 * MREC ? VT_BYTES : 8 * 32 * MREC;
    constexpr int RAW_BYTES = PR * 4 * PW * 16;
    constexpr int NITEM = PR * NQT * 2;             // transform items (row, quad, octet) per chunk
    constexpr int NUNIT = PR * PW * 4;              // 16-byte units of a chunk's raw patch
    constexpr int UPT = (NUNIT + 511) / 512;        // ... per thread
    static_assert(32 % NQT == 0 && R % RM == 0 && N % TW == 0 && N % R == 0 && NITEM <= 512 && UPT <= 7, "tile shape");
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
        ep[i] = i < COUT ? a.bias[i] : (i < 2 * COUT ? a.scale[i - COUT] : (i < 3 * COUT ? a.shift[i - 2 * COUT] : a.atp[(i - 3 * COUT) >> 3][(i - 3 * COUT) & 7]));

    // transform work items: (row, quad, octet, half of the octet's 8 channels); lanes run over the quads first (consecutive
    // slots of the raw patch).  NHALF = 2 NITEM half-items over 512 threads in two rounds: at 64 x 64 waves 0..3 take two,
    // waves 4..7 one, i.e. 1.5 item-times per SIMD (whole items on waves 0..5 made it 2 on SIMDs 0 and 1)
    constexpr int NHALF = 2 * NITEM;
    // B fragment of this lane: pair li = (row li / NQT, quad li % NQT) of an M-tile, octet h (units 2h | 2h + 1, swizzled)
    const int fq = li % NQT, fr = li / NQT;
    const int fbase = ((p * PR + fr) * NQT + fq) * REC + (((2 * h) ^ (((fr * NQT + fq) >> 2) & 3)) * 16);
    auto frag = [&](int row_off) -> int {        // byte offset of the hi unit of patch row fr + row_off; the lo unit is ^ 16
        if constexpr (NQT == 16) return fbase + row_off * NQT * REC;                         // swizzle: the quad alone
        else if constexpr (NQT == 8) return (fbase + row_off * NQT * REC) ^ ((row_off & 1) * 32);   // ... and the row's parity
        else {
            const int prow = fr + row_off;
            return ((p * PR + prow) * NQT + fq) * REC + (((2 * h) ^ (((prow * NQT + fq) >> 2) & 3)) * 16);
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
        const int un_ = u_ & 3, xl_ = (u_ >> 2) % PW, r_ = u_ / (4 * PW);                                       \
        const int b_ = (TILE) / tiles_per_img, tr_ = (TILE) - b_ * tiles_per_img;                               \
        int gy_ = (tr_ / XT) * R - 2 + r_, gx_ = FULLW ? xl_ : (tr_ % XT) * TW - 2 + xl_;                       \
        gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                                   \
        gx_ = gx_ < 0 ? gx_ + N : (gx_ >= N ? gx_ - N : gx_);                                                   \
        DST = *reinterpret_cast<const f32x4 *>(inb + (((size_t)b_ * N + gy_) * N + gx_) * PIXB + (CH) * 64 + un_ * 16); \
    }
#define QGX_RAW_STORE(J, SRC)                                                                                   \
    {                                                                                                           \
        const int u_ = (J) * 512 + (int)threadIdx.x;                                                            \
        if (u_ < NUNIT) {                                                                                       \
            const int un_ = u_ & 3, xl_ = (u_ >> 2) % PW, r_ = u_ / (4 * PW);                                   \
            *reinterpret_cast<f32x4 *>(rawb + (((r_ * 4 + un_) * PW) + (xl_ & 3) * SQ + (xl_ >> 2)) * 16) = SRC;       \
        }                                                                                                       \
    }

    f32x16 acc[MT][2];
    h8 Wn[2][2];                                    // [nt][part] of the NEXT (chunk, ky) step, prefetched from L2
#define QGX_W_LOAD(S)                                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
            Wn[nt][j] = *reinterpret_cast<const h8 *>(wb + (size_t)(S) * WSLICE + wofs + j * 2 * COUT * 16 + nt * 32 * 16);
    QGX_W_LOAD(0)
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
#pragma nounroll
            for (int rep = 0; rep < (NHALF + 511) / 512; ++rep) {
                const int hi_ = rep * 512 + (int)threadIdx.x;
                if (hi_ < NHALF && EXP != 1) {
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
            for (int ky = 0; ky < (EXP == 2 ? 0 : KY); ++ky) {
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
                    if (ky == 4 && UPT > 6) { f32x4 rw2; QGX_RAW_LOAD(6, n_tile, n_ch, rw2) QGX_RAW_STORE(6, rw2) }
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
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                const int item = rep * 512 + threadIdx.x;       // (pair, j, octet g)
                const int g = item & 7, j = (item >> 3) & 3, pl = item >> 5;
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float c = ep[3 * COUT + j * 8 + q];
                    const f32x4 m0 = *reinterpret_cast<const f32x4 *>(vt + (size_t)(q * 32 + pl) * MREC + g * 32);
                    const f32x4 m1 = *reinterpret_cast<const f32x4 *>(vt + (size_t)(q * 32 + pl) * MREC + g * 32 + 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { y[e] = fmaf(c, m0[e], y[e]); y[4 + e] = fmaf(c, m1[e], y[4 + e]); }
                }
                float mx = 0.f;
                unsigned hw[4], lw[4];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = g * 8 + e;
                    y[e] = fmaxf(y[e] + ep[c], 0.f) * ep[COUT + c] + ep[2 * COUT + c];
                    mx = fmaxf(mx, fabsf(y[e]));
                }
                range_guard(mx * a.ascale, a.range, a.range_bit);
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const float v0 = y[2 * e2] * a.ascale, v1 = y[2 * e2 + 1] * a.ascale;
                    hw[e2] = pack_h2(v0, v1);
                    lw[e2] = pack_h2(mix_rest<0>(hw[e2], v0), mix_rest<1>(hw[e2], v1));
                }
                const int row = mt * RM + pl / NQT, col = 4 * (pl % NQT) + j;
                char *o = ob + ((size_t)row * N + col) * OPIXB + g * 32;
                const u32x4 oh = {hw[0], hw[1], hw[2], hw[3]}, ol = {lw[0], lw[1], lw[2], lw[3]};
                *reinterpret_cast<u32x4 *>(o) = oh;
                *reinterpret_cast<u32x4 *>(o + 16) = ol;
            }
        }
    }
#undef QGX_W_LOAD
#undef QGX_RAW_LOAD
#undef QGX_RAW_STORE
}

// LDS bytes of k_convw<NN, TW, R>
constexpr size_t convw_lds_bytes(int NN, int TW, int R) {
    const size_t vtb = (size_t)8 * (R + 4) * (TW / 4) * 64, st = (size_t)8 * 32 * (64 * 4 + 16);
    return (vtb > st ? vtb : st) + (size_t)(R + 4) * 4 * (TW == NN ? NN : TW + 4) * 16 + (3 * 64 + 32) * sizeof(float);
}
