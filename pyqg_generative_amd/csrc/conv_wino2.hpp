// Generator layer 2 as the 1-D Winograd convolution of conv_wino.hpp (same arithmetic, same operand layouts, BIT-identical
// results), re-organised so that its input transform and raw-patch copy run UNDER its matrix instructions instead of
// between them.  Included by conv.hip after conv_wino.hpp.
//
// k_convw (conv_wino.hpp) runs all eight waves of a workgroup through the same barrier-separated phases: raw-patch copy /
// input transform / 120 MFMAs per wave / output transform.  Only 56 % of its time has an MFMA in flight (DESIGN.md section
// 3.2c, round 3: 0.166 of 0.30 ms).  Overlapping chunk c + 1's transform with chunk c's MFMAs by double-buffering the
// transformed patch does not fit (2 x 96 KB + 48 KB of raw patch against 160 KB of LDS).  Here the eight POSITIONS are
// split into two teams of four waves — A = {0, 1, 2, 7}, B = {3, 4, 5, 6}: the rows of BT pair up that way, team B needs
// only x1..x6 of a quad's eight pixels — and the teams alternate roles every phase:
//
//      phase 2c     (even):  A multiplies chunk c (its four positions of the transformed patch),   B transforms chunk c
//      phase 2c + 1 (odd):   A transforms chunk c + 1,                                            B multiplies chunk c
//
// A workgroup's waves are dealt to the SIMDs cyclically, so every SIMD hosts one wave of each team: while one issues MFMAs
// its partner issues the VALU / LDS work of the transform — the complementary pairing of MI355X_MICROARCH.md ("Two waves
// per SIMD").  The live LDS set stays what k_convw uses: at any time one team's half of the transformed patch is being
// read and the other's written (96 KB together), and every weight fragment is still fetched from L2 ONCE per tile and
// chunk (a split by pixels instead of positions would fetch them twice: 39 B/clk/CU, above what L2 serves a CU).
// The raw patch stays single-buffered (51 KB): chunk c is read by A in phase 2c - 1 and by B in phase 2c, so chunk c + 1
// is written in the tail of phase 2c, behind a mid-phase barrier that B reaches when its transform has read the last
// pixel; the global loads of that copy are issued by A a phase earlier (at the start of its transform phase) and by B at
// the start of phase 2c, into 8 x UPP registers of every thread.  The copy adds the hi and lo halves (x = hi + lo in
// float32, the first step of k_convw's transform: v_fma_mix_f32) ONCE and stores float32 — the same 4 bytes per value —
// so neither team's transform repeats the 32 conversions per half-item, and the column slots of the transposed patch
// are padded to an odd pitch: the copy's 16-byte stores were 8-way bank-conflicted in k_convw.
// The output transform (all eight positions of a pair are needed) remains a workgroup-wide epoch per tile.
#pragma once

// LDS-only barrier: __syncthreads() also drains vmcnt(0), i.e. every prefetch in flight (weight fragments, raw patch)
#define QGX_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// The thread index through an opaque asm: what is computed from it is computed HERE, every time — hipcc otherwise hoists
// every thread-constant address of every phase (some thirty registers) out of the loops and keeps them live beside the 128
// accumulators, and the spills that follow are reloaded behind `s_waitcnt vmcnt(0)`, i.e. behind the prefetches in flight
#define QGX_OPAQUE_TID(VAR) int VAR = (int)threadIdx.x; asm volatile("" : "+v"(VAR));

// diagnostic builds of bench_tools/wino2_dev.hip only: s_memtime trace of wave 0 (team A) and wave 4 (team B), 512 slots each
#ifdef QGX_W2_STAMPS
__device__ unsigned long long *g_w2_stamps;
#define QGX_W2_STAMP(ID)                                                                                        \
    if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 8) {                                                       \
        unsigned long long *sp_ = g_w2_stamps + ((size_t)blockIdx.x * 2 + team) * 512;                          \
        if (stamp_i < 512) sp_[stamp_i++] = ((unsigned long long)(ID) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); \
    }
#else
#define QGX_W2_STAMP(ID)
#endif

// BT of F(4, 5), the rows of team A (positions 0, 1, 2, 7) and of team B (3, 4, 5, 6): the expressions of wino_bt8
__device__ __forceinline__ void wino_bt_a(const float (&x)[8][4], int e, float (&v)[4]) {
    const float x0 = x[0][e], x1 = x[1][e], x2 = x[2][e], x3 = x[3][e], x4 = x[4][e], x5 = x[5][e], x6 = x[6][e], x7 = x[7][e];
    v[0] = fmaf(5.25f, x2 - x4, x6) - x0;
    v[3] = fmaf(5.25f, x3 - x5, x7) - x1;
    const float a1 = fmaf(-4.25f, x4, x2 + x6), b1 = fmaf(-4.25f, x3, x1 + x5);
    v[1] = a1 + b1; v[2] = a1 - b1;
}
__device__ __forceinline__ void wino_bt_b(const float (&x)[8][4], int e, float (&v)[4]) {
    const float x1 = x[1][e], x2 = x[2][e], x3 = x[3][e], x4 = x[4][e], x5 = x[5][e], x6 = x[6][e];
    const float a3 = fmaf(-1.25f, x4, fmaf(0.25f, x2, x6)), b3 = fmaf(2.f, x5, fmaf(-2.5f, x3, 0.5f * x1));
    v[0] = a3 + b3; v[1] = a3 - b3;
    const float a5 = fmaf(-5.f, x4, fmaf(4.f, x2, x6)), b5 = fmaf(0.5f, x5, fmaf(-2.5f, x3, 2.f * x1));
    v[2] = a5 + b5; v[3] = a5 - b5;
}

// EXP (bench_tools/wino2_dev.hip, timing experiments only — wrong results): 1 no MFMAs, 2 no input transform, 3 = 2 + weights
// loaded once, 4 = 2 + B fragments read once, 5 = 3 + 4
template <int NN, int TW, int R, int EXP = 0>
__global__ __launch_bounds__(512) void k_convw2(ConvWArgs a, int total_tiles) {
    constexpr int N = NN, CIN = 128, COUT = 64, NCH = CIN / 16, KY = 5;
    constexpr int NQT = TW / 4;                     // quads per tile row
    constexpr int RM = 32 / NQT;                    // rows per M-tile of 32 (row, quad) pairs
    constexpr int MT = R / RM, PR = R + 4;          // M-tiles, patch rows of a tile
    constexpr int XT = N / TW;
    constexpr bool FULLW = TW == N;
    constexpr int PW = FULLW ? N : TW + 4;          // patch columns held in LDS: x0 - 2 ... x0 + TW + 1 (wrapped)
    constexpr int SQ = PW / 4;                      // quads of the raw patch
    // ... their pitch in 16-byte slots.  The copy stores 8 lanes = 4 consecutive pixels x 2 octets per LDS cycle (slots SQP
    // apart, unit rows 2 apart), the transform reads 16 consecutive quads of ONE pixel-in-quad and unit row.  With 16 quads
    // (64 columns) both are conflict-free when the unit rows are 1024 bytes and the quad index is XORed with 2 (pixel in
    // quad) + octet: a permutation inside the 256 bytes a read covers, eight different bank groups for the eight lanes of
    // a store (k_convw's plain layout stores 8-way conflicted).  Other widths: an odd pitch (stores spread, reads 2-way)
    constexpr bool SWZ = SQ == 16;
    constexpr int SQP = SWZ ? 16 : (SQ | 1);
    constexpr int URB = 4 * SQP * 16;               // bytes of one unit row: column slot = SQP (xl & 3) + (xl >> 2)
    constexpr int PIXB = CIN * 4, OPIXB = COUT * 4;
    constexpr int REC = 64;
    constexpr int VPS = PR * NQT * REC;
    constexpr int VT_BYTES = 8 * VPS;
    constexpr int MREC = COUT * 4 + 16;
    constexpr int A_BYTES = VT_BYTES > 8 * 32 * MREC ? VT_BYTES : 8 * 32 * MREC;
    constexpr int RAW_BYTES = PR * 4 * URB;         // [row][unit = (octet, half): 4 channels float32][column slot]
    constexpr int RR = 256 / (4 * NQT);             // patch rows a team's 256 threads transform per round
    constexpr int NREP = (PR + RR - 1) / RR;
    constexpr int NPAIR = PR * PW * 2;              // (row, column, octet) hi + lo unit pairs of a chunk's raw patch
    // ... per thread: team A carries its pairs through a multiply phase (beside 128 accumulator and 48 weight registers), so it
    // takes a third of them and team B, which loads and stores inside its transform phase, two thirds
    constexpr int NP256 = (NPAIR + 255) / 256;
    constexpr int UA = NP256 / 3 > 0 ? NP256 / 3 : 1, UB = NP256 - UA, UPP = UA > UB ? UA : UB;
    static_assert(32 % NQT == 0 && R % RM == 0 && N % TW == 0 && N % R == 0 && UPP <= 5 && 256 % (4 * NQT) == 0, "tile shape");
    char *const vt = conv_smem;
    char *const rawb = conv_smem + A_BYTES;
    float *const ep = reinterpret_cast<float *>(conv_smem + A_BYTES + RAW_BYTES);         // bias | scale | shift | u_p
    const char *const inb = reinterpret_cast<const char *>(a.in);
    const char *const wb = reinterpret_cast<const char *>(a.w);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int team = wave >> 2;                                              // 0: positions {0, 1, 2, 7}, 1: {3, 4, 5, 6}
    const int p = team == 0 ? (wave == 3 ? 7 : wave) : wave - 1;             // the position this wave multiplies
    const int li = lane & 31, h = lane >> 5;
    constexpr int tiles_per_img = (N / R) * XT;
    const int n_my = blockIdx.x < (unsigned)total_tiles ? (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (n_my == 0) return;
    const int n_chunks = n_my * NCH;                                         // linear chunk sequence g of this workgroup
    for (int i = threadIdx.x; i < 3 * COUT + 32; i += 512)
        ep[i] = i < COUT ? a.bias[i] : (i < 2 * COUT ? a.scale[i - COUT] : (i < 3 * COUT ? a.shift[i - 2 * COUT] : a.pscale[(i - 3 * COUT) & 7]));

    // B fragment of this lane: pair li = (row li / NQT, quad li % NQT) of an M-tile, octet h (units 2h | 2h + 1, swizzled by
    // bits 2..3 of the pair index).  A row offset adds a multiple of NQT * 64 bytes and flips swizzle bits that depend on
    // the offset alone, so every fragment address is one of a few per-lane bases plus a COMPILE-TIME offset (ds_read ...
    // offset:) — formed as (base + offset) ^ 16 each address is a register of its own, and twenty of those spill here
    const int fq = li % NQT, fr = li / NQT;
    constexpr int NFB = NQT == 16 ? 1 : (NQT == 8 ? 2 : 4);
    int fb[NFB];
#pragma unroll
    for (int k = 0; k < NFB; ++k) {
        const int prow = fr + k;                                             // swizzle of patch row fr + row_off, row_off = k mod NFB
        fb[k] = p * VPS + (fr * NQT + fq) * REC + (((2 * h) ^ (((prow * NQT + fq) >> 2) & 3)) * 16);
    }
    auto frag = [&](int row_off, int lo) -> int {    // byte address of the hi (lo = 0) / lo (lo = 1) unit of patch row fr + row_off
        return (fb[row_off % NFB] ^ (lo * 16)) + row_off * NQT * REC;
    };
    const int wofs = (p * 4 + h) * COUT * 16 + li * 16;
    constexpr int WSLICE = 8 * 4 * COUT * 16;

    // ---- raw patch of linear chunk G: pair J of this thread = the hi and the lo unit of (row, column, octet): global ->
    //      registers; registers -> x = hi + lo -> two float32 units in LDS.  Team A's threads own the pairs
    //      [0, 256 UA), team B's the rest ----
#define QGX_RAW_LOAD(J, G)                                                                                      \
    {                                                                                                           \
        int q_ = (J) * 256 + q_base;                                                                            \
        q_ = q_ < NPAIR ? q_ : NPAIR - 1;                                                                       \
        const int gt_ = (G) / NCH, gc_ = (G) - gt_ * NCH;                                                       \
        const int tile_ = (int)blockIdx.x + gt_ * (int)gridDim.x;                                               \
        const int b_ = tile_ / tiles_per_img, tr_ = tile_ - b_ * tiles_per_img;                                 \
        const int o_ = q_ & 1, xl_ = (q_ >> 1) % PW, r_ = q_ / (2 * PW);                                        \
        int gy_ = (tr_ / XT) * R - 2 + r_, gx_ = FULLW ? xl_ : (tr_ % XT) * TW - 2 + xl_;                       \
        gy_ = gy_ < 0 ? gy_ + N : (gy_ >= N ? gy_ - N : gy_);                                                   \
        gx_ = gx_ < 0 ? gx_ + N : (gx_ >= N ? gx_ - N : gx_);                                                   \
        const char *gp_ = inb + (((size_t)b_ * N + gy_) * N + gx_) * PIXB + gc_ * 64 + o_ * 32;                 \
        rwh[J] = *reinterpret_cast<const u32x4 *>(gp_);                                                         \
        rwl[J] = *reinterpret_cast<const u32x4 *>(gp_ + 16);                                                    \
    }
#define QGX_RAW_STORE(J)                                                                                        \
    {                                                                                                           \
        const int q_ = (J) * 256 + q_base;                                                                      \
        if (q_ < NPAIR) {                                                                                       \
            const int o_ = q_ & 1, xl_ = (q_ >> 1) % PW, r_ = q_ / (2 * PW);                                    \
            char *lp_ = rawb + (r_ * 4 + 2 * o_) * URB + ((xl_ & 3) * SQP + ((xl_ >> 2) ^ (SWZ ? 2 * (xl_ & 3) + o_ : 0))) * 16; \
            const f32x4 f0_ = {mix_sum<0>(rwh[J][0], rwl[J][0]), mix_sum<1>(rwh[J][0], rwl[J][0]),              \
                               mix_sum<0>(rwh[J][1], rwl[J][1]), mix_sum<1>(rwh[J][1], rwl[J][1])};             \
            const f32x4 f1_ = {mix_sum<0>(rwh[J][2], rwl[J][2]), mix_sum<1>(rwh[J][2], rwl[J][2]),              \
                               mix_sum<0>(rwh[J][3], rwl[J][3]), mix_sum<1>(rwh[J][3], rwl[J][3])};             \
            *reinterpret_cast<f32x4 *>(lp_) = f0_;                                                              \
            *reinterpret_cast<f32x4 *>(lp_ + URB) = f1_;                                                        \
        }                                                                                                       \
    }
#define QGX_RAW_LOADS(G, J0, J1)  { if ((G) < n_chunks && EXP != 13 && EXP != 14 && EXP != 15) { QGX_OPAQUE_TID(tq_) const int q_base = (tq_ & 255) + (team == 0 ? 0 : 256 * UA); \
                                                           _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_) QGX_RAW_LOAD(j_, (G)) } }
#define QGX_RAW_STORES(G, U) { if ((G) < n_chunks && EXP != 13 && EXP != 14 && EXP != 15) { QGX_OPAQUE_TID(tq_) const int q_base = (tq_ & 255) + (team == 0 ? 0 : 256 * UA); \
                                                      _Pragma("unroll") for (int j_ = 0; j_ < (U); ++j_) QGX_RAW_STORE(j_) } }

    // ---- input transform of the chunk in the raw patch: this team's four positions.  Half-item (row, quad, octet, half
    //      of the octet's 8 channels) = thread tt + 256 round: the lanes run over the quads first, a round advances RR
    //      whole rows, so quad / octet / half / swizzle of a thread are the same in every round ----
    // (the transform runs at raised priority: beside a partner whose next MFMA is always pending, vector instructions of a
    //  wave of equal or lower priority wait for issue slots — measured 15 to 20 cycles per instruction)
#ifndef QGX_W2_TPRIO
#define QGX_W2_TPRIO 3
#endif
#define QGX_TRANSFORM(TEAM, WSTEP)                                                                              \
    __builtin_amdgcn_s_setprio(QGX_W2_TPRIO);                                                                   \
    if (EXP == 11 || EXP == 14 || EXP == 15) {      /* 14: clean MFMA loop + clean VALU transform, no raw copy; 15: ... and no weight fetch */ \
        float xf_[16];                                                                                          \
        _Pragma("unroll") for (int k_ = 0; k_ < 16; ++k_) xf_[k_] = 1.0f + 0.001f * (lane + k_);                \
        _Pragma("unroll") for (int r_ = 0; r_ < 19; ++r_)                                                       \
            _Pragma("unroll") for (int k_ = 0; k_ < 16; ++k_) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf_[k_]) : "v"(0.999f), "v"(0.001f)); \
        float sf_ = 0.f;                                                                                        \
        _Pragma("unroll") for (int k_ = 0; k_ < 16; ++k_) sf_ += xf_[k_];                                       \
        if (sf_ == 12345.f) ep[0] = sf_;                                                                        \
        if (EXP != 15) { QGX_W_LOAD(0, WSTEP) }                                                                 \
    } else {                                                                                                    \
    QGX_OPAQUE_TID(tt_)                                                                                         \
    tt_ &= 255;                                                                                                 \
    const int it_t = tt_ % NQT, it_o = (tt_ / NQT) & 1, it_hf = (tt_ / (2 * NQT)) & 1, it_r0 = tt_ / (4 * NQT);  \
    const int t_src0 = (it_r0 * 4 + it_o * 2 + it_hf) * URB;                                                    \
    /* column slots of the window's pixels k = 0..7: x-tiled: local column 4 t + k -> slot SQP (k & 3) + t + (k >> 2); */ \
    /* full width: x = 4 t - 2 + k wrapped -> slot SQP ((k + 2) & 3) + (t - 1 | t | t + 1 mod NQT) */           \
    const int t_sm = FULLW ? ((it_t + NQT - 1) & (NQT - 1)) : it_t, t_s0 = it_t, t_sp = FULLW ? ((it_t + 1) & (NQT - 1)) : it_t + 1; \
    const int t_dsth = (it_r0 * NQT + it_t) * REC + ((2 * it_o) ^ (((it_r0 * NQT + it_t) >> 2) & 3)) * 16 + it_hf * 8; \
    const int t_dstl = t_dsth ^ 16;                 /* the lo unit of the record */                             \
    _Pragma("nounroll") for (int rep = 0; rep < ((EXP >= 2 && EXP <= 6) ? 0 : NREP); ++rep) {                                 \
        /* block 0 of the coming multiply phase: fetched one round (~1000 cycles) before the phase barrier */   \
        if (rep == NREP - 1 && EXP != 6) { QGX_W_LOAD(0, WSTEP) }                                                         \
        if (NREP * RR == PR || it_r0 + rep * RR < PR) {                                                         \
            const char *src = rawb + t_src0 + rep * (RR * 4 * URB);                                             \
            char *dsth = vt + t_dsth + rep * (RR * NQT * REC), *dstl = vt + t_dstl + rep * (RR * NQT * REC);    \
            /* two passes of two channels each (16 + 8 live registers instead of 32 + 16: beside 128 accumulators, the */ \
            /* raw patch in flight and a weight block the four-channel form spills — and a spilled LOAD is a vmcnt(0)) */ \
            unsigned hw[2][4], lw[2][4];                                                                        \
            _Pragma("unroll") for (int ps = 0; ps < 2; ++ps) {                                                  \
                float x[8][4];                                                                                  \
                _Pragma("unroll") for (int k = (TEAM); k < 8 - (TEAM); ++k) {                                   \
                    const int sl = FULLW ? ((k + 2) & 3) * SQP + ((k < 2 ? t_sm : (k < 6 ? t_s0 : t_sp)) ^ (SWZ ? 2 * ((k + 2) & 3) + it_o : 0)) \
                                         : (k & 3) * SQP + ((k < 4 ? t_s0 : t_sp) ^ (SWZ ? 2 * (k & 3) + it_o : 0)); \
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));                                   \
                    f32x2_ xv;                                                                                  \
                    if (EXP == 8 || EXP == 9) { const float f_ = __builtin_bit_cast(float, (unsigned)(size_t)src + k); xv = f32x2_{f_, f_ + 1.f}; } \
                    else xv = *reinterpret_cast<const f32x2_ *>(src + sl * 16 + ps * 8);                        \
                    x[k][0] = xv[0]; x[k][1] = xv[1]; x[k][2] = 0.f; x[k][3] = 0.f;                             \
                }                                                                                               \
                if ((TEAM) == 1) { _Pragma("unroll") for (int e = 0; e < 4; ++e) { x[0][e] = 0.f; x[7][e] = 0.f; } } \
                float v[2][4];                         /* [channel of the pass][position of the team] */        \
                _Pragma("unroll") for (int e = 0; e < 2; ++e) { if ((TEAM) == 0) wino_bt_a(x, e, v[e]); else wino_bt_b(x, e, v[e]); } \
                _Pragma("unroll") for (int ql = 0; ql < 4; ++ql) {                                              \
                    const float v0 = v[0][ql], v1 = v[1][ql];                                                   \
                    hw[ps][ql] = pack_h2(v0, v1);                                                               \
                    lw[ps][ql] = pack_h2(mix_rest<0>(hw[ps][ql], v0), mix_rest<1>(hw[ps][ql], v1));             \
                }                                                                                               \
            }                                                                                                   \
            _Pragma("unroll") for (int ql = 0; ql < 4; ++ql) {                                                  \
                const int q = (TEAM) == 0 ? (ql == 3 ? 7 : ql) : ql + 3;                                        \
                const u32x2 oh = {hw[0][ql], hw[1][ql]}, ol = {lw[0][ql], lw[1][ql]};                           \
                if (EXP == 7 || EXP == 9) { if (oh[0] == 0x12345678u && ol[1] == 0x9abcdef0u) *reinterpret_cast<u32x2 *>(dsth + q * VPS) = oh; } \
                else {                                                                                          \
                    *reinterpret_cast<u32x2 *>(dsth + q * VPS) = oh;                                            \
                    *reinterpret_cast<u32x2 *>(dstl + q * VPS) = ol;                                            \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }                                                                                                           \
    }                                                                                                           \
    __builtin_amdgcn_s_setprio(0);

    f32x16 acc[MT][2];
    u32x4 rwh[UPP], rwl[UPP];                       // raw patch of a coming chunk, in flight through a phase
    // weight fragments [nt][part] of three (chunk, ky) steps: a lone wave per SIMD issues a block of 24 MFMAs in ~770 cycles,
    // less than an L2 round trip under load, so the fragments are fetched TWO blocks ahead into a ring of three slots (block k
    // of a phase uses slot k % 3 and loads block k + 2 into slot (k + 2) % 3); blocks 0 and 1 of a multiply phase are fetched
    // at the END of the team's transform phase in front of it (the transform's registers are free by then, and the team
    // usually waits at the phase barrier for its partner's MFMAs anyway), so nothing of the ring is live during a transform
    h8 W3[3][2][2];
#ifndef QGX_W2_SPREAD
#define QGX_W2_SPREAD 0
#endif
#define QGX_W_LOAD(SLOT, S)                                                                                    \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
            W3[SLOT][nt][j] = *reinterpret_cast<const h8 *>(wb + (size_t)(S) * WSLICE + wofs + j * 2 * COUT * 16 + nt * 32 * 16);

    // ---- the 5 row offsets x MT M-tiles x 2 output-channel tiles x 3 MFMAs of chunk CH on this wave's position ----
    // MIDBAR (team A, even phase): the workgroup's mid-phase barrier and this thread's share of the next raw patch (linear
    // chunk GS) after the fourth block.  LOADS (team B, odd phase): the first half of this thread's share of the raw patch of
    // chunk GL, behind the third block's weight fetch (the other half follows at the start of the team's transform phase:
    // a CU that issues a whole chunk's 48 KB of loads at once waits ~3 us for the last of them, and its L2-hit weight
    // fetches queue behind them)
#define QGX_MULTIPLY(CH, MIDBAR, GS, LOADS, GL)                                                                            \
    {                                                                                                           \
        if ((CH) == 0) {                                                                                        \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                   \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;                        \
        }                                                                                                       \
        /* one stream of KY x MT groups (row offset ky, M-tile mt) of 6 MFMAs; the two B fragments of group i + 1 are */ \
        /* read from LDS BEFORE the MFMAs of group i, across block boundaries too: a lone wave has nobody to cover an */ \
        /* LDS latency.  sched_group_barrier pins that order (hipcc otherwise sinks the reads to two MFMAs before use) */ \
        h8 Pb[2][2];                                /* [group parity][hi | lo] */                               \
        Pb[0][0] = *reinterpret_cast<const h8 *>(vt + frag(0, 0));                                              \
        Pb[0][1] = *reinterpret_cast<const h8 *>(vt + frag(0, 1));                                              \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                      \
        _Pragma("unroll") for (int gi = 0; gi < KY * MT; ++gi) {                                                \
            const int ky = gi / MT, mt = gi % MT;                                                               \
            if (mt == 0) { QGX_W2_STAMP(10 + ky) }                                                              \
            if (ky + 2 < KY && EXP != 3 && EXP != 5) {                                                          \
                /* the four 1-KB fetches of the block after next: in a row at the head of the block, or (QGX_W2_SPREAD) */ \
                /* one in front of each group.  A fetch costs the matrix pipe ~18 cycles (bench_tools/coissue.hip: 35.1  */ \
                /* against 32.2 cycles per MFMA), more while the partner wave fetches too (41 in a row, 36 spread in the */ \
                /* microbenchmark); in this kernel the two orders time the same (301.7 / 303.1 us), the row is kept      */ \
                const int q0_ = QGX_W2_SPREAD ? mt * 4 / MT : (mt == 0 ? 0 : 4);                            \
                const int q1_ = QGX_W2_SPREAD ? (mt + 1) * 4 / MT : 4;                                      \
                _Pragma("unroll") for (int q = q0_; q < q1_; ++q)                                               \
                    W3[(ky + 2) % 3][q >> 1][q & 1] = *reinterpret_cast<const h8 *>(                            \
                        wb + (size_t)((CH) * KY + ky + 2) * WSLICE + wofs + (q & 1) * 2 * COUT * 16 + (q >> 1) * 32 * 16); \
                if (q1_ - q0_ == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                          \
                if (q1_ - q0_ == 2) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                          \
                if (q1_ - q0_ == 4) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                          \
            }                                                                                                   \
            if (gi == 2 * MT && (LOADS)) {          /* behind the last weight fetch of the phase: vmcnt retires in order */ \
                __builtin_amdgcn_sched_barrier(0);                                                              \
                QGX_RAW_LOADS(GL, 0, UB / 2)                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                              \
            }                                                                                                   \
            if (gi + 1 < KY * MT && (EXP < 4 || gi == 0)) {                                                     \
                Pb[(gi + 1) & 1][0] = *reinterpret_cast<const h8 *>(vt + frag(((gi + 1) % MT) * RM + (gi + 1) / MT, 0)); \
                Pb[(gi + 1) & 1][1] = *reinterpret_cast<const h8 *>(vt + frag(((gi + 1) % MT) * RM + (gi + 1) / MT, 1)); \
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                              \
            }                                                                                                   \
            const h8 Ph = Pb[gi & 1][0], Pl = Pb[gi & 1][1];                                                      \
            _Pragma("unroll") for (int nt = 0; nt < (EXP == 1 ? 0 : 2); ++nt) {                                 \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W3[ky % 3][nt][1], Ph, acc[mt][nt], 0, 0, 0);  \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W3[ky % 3][nt][0], Pl, acc[mt][nt], 0, 0, 0);  \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W3[ky % 3][nt][0], Ph, acc[mt][nt], 0, 0, 0);  \
            }                                                                                                   \
            if (EXP != 1) __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                    \
            if (gi == 4 * MT - 1 && (MIDBAR)) {                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                              \
                QGX_W2_STAMP(7)                                                                                 \
                QGX_LDS_BARRIER();                  /* team B's transform has read the last pixel of this chunk */ \
                QGX_RAW_STORES(GS, UA)                                                                          \
                __builtin_amdgcn_sched_barrier(0);                                                              \
            }                                                                                                   \
        }                                                                                                       \
    }

#ifdef QGX_W2_STAMPS
    int stamp_i = 0;
#endif
#define QGX_TILE_COORDS(TI)                                                                                     \
            const int tile_g = blockIdx.x + (TI) * gridDim.x;                                                   \
            const int b = tile_g / tiles_per_img, tr = tile_g - b * tiles_per_img;                              \
            const int y0 = (tr / XT) * R, x0 = (tr % XT) * TW;
#define QGX_OUTPUT_EPOCH()  \
        /* ---- output transform + epilogue, one M-tile at a time through the (now free) patch region: conv_wino.hpp ---- */ \
        char *const ob = reinterpret_cast<char *>(a.out) + (((size_t)b * N + y0) * N + x0) * OPIXB; \
        QGX_W2_STAMP(8) \
        QGX_OPAQUE_TID(to_) \
        _Pragma("unroll") \
        for (int mt = 0; mt < MT; ++mt) { \
            if (mt) QGX_LDS_BARRIER(); \
        _Pragma("unroll") \
            for (int nt = 0; nt < 2; ++nt) \
        _Pragma("unroll") \
                for (int q = 0; q < 4; ++q) { \
                    const f32x4 vv = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]}; \
                    *reinterpret_cast<f32x4 *>(vt + (size_t)(p * 32 + (to_ & 31)) * MREC + (nt * 32 + 8 * q + 4 * ((to_ >> 5) & 1)) * 4) = vv; \
                } \
            QGX_LDS_BARRIER(); \
            { \
                const int hf = to_ & 1, g8 = (to_ >> 1) & 7, pl = to_ >> 4; \
                const float *const u = ep + 3 * COUT; \
                f32x4 m[8]; \
        _Pragma("unroll") \
                for (int q = 0; q < 8; ++q) \
                    m[q] = *reinterpret_cast<const f32x4 *>(vt + (size_t)(q * 32 + pl) * MREC + g8 * 32 + hf * 16); \
                float y[4][4]; \
        _Pragma("unroll") \
                for (int e = 0; e < 4; ++e) { \
                    const float t1 = u[1] * m[1][e], t2 = u[3] * m[3][e], t3 = u[5] * m[5][e]; \
                    const float s1 = fmaf(u[2], m[2][e], t1), d1 = fmaf(-u[2], m[2][e], t1); \
                    const float s2 = fmaf(u[4], m[4][e], t2), d2 = fmaf(-u[4], m[4][e], t2); \
                    const float s3 = fmaf(u[6], m[6][e], t3), d3 = fmaf(-u[6], m[6][e], t3); \
                    y[0][e] = fmaf(u[0], m[0][e], (s1 + s2) + s3); \
                    y[1][e] = fmaf(.5f, d3, fmaf(2.f, d2, d1)); \
                    y[2][e] = fmaf(.25f, s3, fmaf(4.f, s2, s1)); \
                    y[3][e] = fmaf(u[7], m[7][e], fmaf(.125f, d3, fmaf(8.f, d2, d1))); \
                } \
                const int c0 = g8 * 8 + hf * 4; \
                const f32x4 bi = *reinterpret_cast<const f32x4 *>(ep + c0); \
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(ep + COUT + c0); \
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(ep + 2 * COUT + c0); \
                float mx = 0.f; \
        _Pragma("unroll") \
                for (int j = 0; j < 4; ++j) \
        _Pragma("unroll") \
                    for (int e = 0; e < 4; ++e) { \
                        y[j][e] = fmaxf(y[j][e] + bi[e], 0.f) * sc[e] + sh[e]; \
                        mx = fmaxf(mx, fabsf(y[j][e])); \
                    } \
                range_guard(mx * a.ascale, a.range, a.range_bit); \
                const int row = mt * RM + pl / NQT, col = 4 * (pl % NQT); \
                char *o = ob + ((size_t)row * N + col) * OPIXB + g8 * 32 + hf * 8; \
        _Pragma("unroll") \
                for (int j = 0; j < 4; ++j) { \
                    unsigned hw[2], lw[2]; \
        _Pragma("unroll") \
                    for (int e2 = 0; e2 < 2; ++e2) { \
                        const float v0 = y[j][2 * e2] * a.ascale, v1 = y[j][2 * e2 + 1] * a.ascale; \
                        hw[e2] = pack_h2(v0, v1); \
                        lw[e2] = pack_h2(mix_rest<0>(hw[e2], v0), mix_rest<1>(hw[e2], v1)); \
                    } \
                    const u32x2 oh = {hw[0], hw[1]}, ol = {lw[0], lw[1]}; \
                    *reinterpret_cast<u32x2 *>(o + (size_t)j * OPIXB) = oh; \
                    *reinterpret_cast<u32x2 *>(o + (size_t)j * OPIXB + 16) = ol; \
                } \
            } \
        } \

    // prologue: the first tile's first chunk, synchronously
    if (team == 0) { QGX_RAW_LOADS(0, 0, UA) QGX_RAW_STORES(0, UA) }
    else { QGX_RAW_LOADS(0, 0, UB) QGX_RAW_STORES(0, UB) }
    QGX_LDS_BARRIER();

    // The two teams run two separate programs with the same sequence of workgroup barriers: per tile a start-up phase (team A
    // transforms the tile's first chunk alone: the output epoch of the previous tile staged through the transformed patch),
    // an even and an odd phase per chunk, the output epoch.
    if (team == 0) {
#pragma nounroll
        for (int ti = 0; ti < n_my; ++ti) {
            QGX_TILE_COORDS(ti)
            QGX_W2_STAMP(1)
            QGX_RAW_LOADS(ti * NCH + 1, 0, UA)
            { QGX_TRANSFORM(0, 0) }
            if ((EXP != 6 || ti == 0) && EXP != 15) { QGX_W_LOAD(1, 1) }
            if (EXP == 6 && ti == 0) { QGX_W_LOAD(0, 0) }
            QGX_W2_STAMP(3)
            QGX_LDS_BARRIER();
#pragma nounroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int g = ti * NCH + ch;
                // even phase: multiply chunk g; behind the mid-phase barrier the raw patch of chunk g + 1 (loaded a phase ago)
                QGX_W2_STAMP(1)
                QGX_MULTIPLY(ch, true, g + 1, false, 0)
                QGX_W2_STAMP(2)
                QGX_LDS_BARRIER();
                // odd phase: transform chunk g + 1 (of this tile), load this team's share of the raw patch of chunk g + 2
                QGX_W2_STAMP(1)
                if (ch + 1 < NCH) {
                    QGX_RAW_LOADS(g + 2, 0, UA)
                    { QGX_TRANSFORM(0, (ch + 1) * KY) }
                    if (EXP != 6 && EXP != 15) { QGX_W_LOAD(1, (ch + 1) * KY + 1) }
                }
                QGX_W2_STAMP(3)
                QGX_LDS_BARRIER();
            }
            QGX_OUTPUT_EPOCH()
            QGX_W2_STAMP(9)
            QGX_LDS_BARRIER();                      // the staging reads are over: the patch region is free again
        }
    } else {
        QGX_RAW_LOADS(1, 0, UB / 2)                 // (later chunks: in the multiply phase two phases ahead)
#pragma nounroll
        for (int ti = 0; ti < n_my; ++ti) {
            QGX_TILE_COORDS(ti)
            QGX_W2_STAMP(1)
            QGX_LDS_BARRIER();                      // start-up phase: nothing to do
#pragma nounroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int g = ti * NCH + ch;
                // even phase: load this team's share of the raw patch of chunk g + 1, transform chunk g, store the share
                // behind the mid-phase barrier, fetch the first two weight blocks of the multiply phase
                QGX_W2_STAMP(1)
                QGX_RAW_LOADS(g + 1, UB / 2, UB)
                { QGX_TRANSFORM(1, ch * KY) }
                QGX_W2_STAMP(4)
                QGX_LDS_BARRIER();                  // this team has read the last pixel of the chunk
                QGX_W2_STAMP(5)
                QGX_RAW_STORES(g + 1, UB)
                if ((EXP != 6 || g == 0) && EXP != 15) { QGX_W_LOAD(1, ch * KY + 1) }
                if (EXP == 6 && g == 0) { QGX_W_LOAD(0, 0) }
                QGX_W2_STAMP(6)
                QGX_LDS_BARRIER();
                // odd phase: multiply chunk g
                QGX_W2_STAMP(1)
                QGX_MULTIPLY(ch, false, 0, true, g + 2)
                QGX_W2_STAMP(2)
                QGX_LDS_BARRIER();
            }
            QGX_OUTPUT_EPOCH()
            QGX_W2_STAMP(9)
            QGX_LDS_BARRIER();
        }
    }
#undef QGX_W_LOAD
#undef QGX_RAW_LOAD
#undef QGX_RAW_STORE
#undef QGX_RAW_LOADS
#undef QGX_RAW_STORES
#undef QGX_TRANSFORM
#undef QGX_MULTIPLY
#undef QGX_OUTPUT_EPOCH
#undef QGX_TILE_COORDS
}

// LDS bytes of k_convw2<NN, TW, R>
constexpr size_t convw2_lds_bytes(int NN, int TW, int R) {
    const size_t vtb = (size_t)8 * (R + 4) * (TW / 4) * 64, st = (size_t)8 * 32 * (64 * 4 + 16);
    const size_t pw = TW == NN ? NN : TW + 4;
    const size_t sqp = pw / 4 == 16 ? 16 : ((pw / 4) | 1);
    const size_t raw = (size_t)(R + 4) * 4 * (4 * sqp * 16);
    return (vtb > st ? vtb : st) + raw + (3 * 64 + 32) * sizeof(float);
}
