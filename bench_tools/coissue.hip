// Microbenchmark: what a lone MFMA wave and a lone VALU wave on the SAME SIMD cost each other (MI355X).
// One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) issue 32x32x16 f16 MFMAs from registers, waves 4-7 issue
// vector-ALU work (independent v_fma_f32, or packed f32, or f16 conversions), with or without LDS reads.  Each role is
// timed alone and together (s_memtime per wave).
//     hipcc -O3 --offload-arch=gfx950 bench_tools/coissue.hip -o bench_tools/_build/coissue && bench_tools/_build/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// mode bit 0: MFMA waves active; bit 1: VALU waves active; vkind: 0 v_fma_f32, 1 v_pk_fma_f32, 2 cvt + fma_mix like the
// transform's split, 3 = 0 with LDS reads / writes interleaved; prio: s_setprio of the VALU waves (MFMA waves stay 0)
template <int VKIND, int MK>
__global__ __launch_bounds__(512) void k_co(int mode, int iters, int prio, float *sink, unsigned long long *times, const float *gbuf) {
    constexpr int mkind = MK;
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc[4];
            for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
            h8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); }
            t0 = __builtin_amdgcn_s_memtime();
            if (mkind == 0) {
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int r = 0; r < 30; ++r)
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
                }
            } else if (mkind == 1) {          // groups of 6 on two accumulators (3 dependent each, interleaved): the conv kernel's order
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int r = 0; r < 20; ++r)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
#pragma unroll
                            for (int k = 0; k < 2; ++k) acc[2 * (r & 1) + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[2 * (r & 1) + k], 0, 0, 0);
                }
            } else if (mkind >= 3) {          // as 2, and the A operands from global memory (L2-resident), two blocks ahead, per 24 MFMAs: 3: four 1-KB wave loads in a group,
                                              // 4: two, 5: four, one before each group of 6 MFMAs, 6: four dword loads (256 B) in a group, 7: eight dwordx2 loads in a group
                const char *lp = reinterpret_cast<const char *>(lds) + lane * 16;
                const char *gp0 = reinterpret_cast<const char *>(gbuf) + threadIdx.x * 16;
                h8 pb[2][2];
                h8 wa[3][4];
                for (int q = 0; q < 4; ++q) { wa[0][q] = *reinterpret_cast<const h8 *>(gp0 + q * 8192); wa[1][q] = *reinterpret_cast<const h8 *>(gp0 + 32768 + q * 8192); wa[2][q] = wa[0][q]; }
                pb[0][0] = *reinterpret_cast<const h8 *>(lp); pb[0][1] = *reinterpret_cast<const h8 *>(lp + 1024);
                for (int it = 0; it < iters; it += 3) {
#pragma unroll
                    for (int blk = 0; blk < 15; ++blk) {                 // 3 "phases" of 5 blocks: the slot ring stays static
                        const char *gp = gp0 + (((it * 5 + blk) & 31) * 32768);
                        if (mkind == 3 || mkind == 4) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) if (q < 2 || mkind == 3) wa[(blk + 2) % 3][q] = *reinterpret_cast<const h8 *>(gp + q * 8192);
                            if (mkind == 3) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0); else __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                        } else if (mkind == 6) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { h2 t = *reinterpret_cast<const h2 *>(gp + q * 8192); wa[(blk + 2) % 3][q][0] = t[0]; wa[(blk + 2) % 3][q][1] = t[1]; }
                            __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
                        } else if (mkind == 7) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                h4 t0 = *reinterpret_cast<const h4 *>(gp + q * 8192), t1 = *reinterpret_cast<const h4 *>(gp + q * 8192 + 4096);
                                for (int e = 0; e < 4; ++e) { wa[(blk + 2) % 3][q][e] = t0[e]; wa[(blk + 2) % 3][q][4 + e] = t1[e]; }
                            }
                            __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (mkind == 5) {
                                wa[(blk + 2) % 3][r] = *reinterpret_cast<const h8 *>(gp + r * 8192);
                                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                            }
                            pb[(r + 1) & 1][0] = *reinterpret_cast<const h8 *>(lp + ((blk * 4 + r + 1) & 15) * 2048);
                            pb[(r + 1) & 1][1] = *reinterpret_cast<const h8 *>(lp + ((blk * 4 + r + 1) & 15) * 2048 + 1024);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                acc[2 * (r & 1) + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[blk % 3][2 * k + 1], pb[r & 1][0], acc[2 * (r & 1) + k], 0, 0, 0);
                                acc[2 * (r & 1) + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[blk % 3][2 * k], pb[r & 1][1], acc[2 * (r & 1) + k], 0, 0, 0);
                                acc[2 * (r & 1) + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[blk % 3][2 * k], pb[r & 1][0], acc[2 * (r & 1) + k], 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                        }
                    }
                }
            } else {                          // ... with the B operands of every group read from LDS one group ahead
                const char *lp = reinterpret_cast<const char *>(lds) + lane * 16;
                h8 pb[2][2];
                pb[0][0] = *reinterpret_cast<const h8 *>(lp); pb[0][1] = *reinterpret_cast<const h8 *>(lp + 1024);
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int r = 0; r < 20; ++r) {
                        pb[(r + 1) & 1][0] = *reinterpret_cast<const h8 *>(lp + ((r + 1) & 15) * 2048);
                        pb[(r + 1) & 1][1] = *reinterpret_cast<const h8 *>(lp + ((r + 1) & 15) * 2048 + 1024);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            acc[2 * (r & 1) + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[r & 1][0], acc[2 * (r & 1) + k], 0, 0, 0);
                            acc[2 * (r & 1) + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, pb[r & 1][1], acc[2 * (r & 1) + k], 0, 0, 0);
                            acc[2 * (r & 1) + k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, pb[r & 1][0], acc[2 * (r & 1) + k], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    }
                }
            }
            t1 = __builtin_amdgcn_s_memtime();
            float s = 0.f;
            for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
            if (s == 12345.f) sink[0] = s;
        }
    } else if (mode & 2) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        float x[16];
        f32x4 gvv[8];
        for (int q = 0; q < 8; ++q) gvv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 16; ++k) x[k] = 1.0f + 0.001f * (lane + k);
        const float c0 = 0.999f, c1 = 0.001f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 20; ++r) {
                if (VKIND == 0 || VKIND == 3 || VKIND == 4 || VKIND == 5 || VKIND == 6) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(c0), "v"(c1));
                } else if (VKIND == 1) {
#pragma unroll
                    for (int k = 0; k < 16; k += 2) {
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        f2 v = {x[k], x[k + 1]}, cc0 = {c0, c0}, cc1 = {c1, c1};
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(cc0), "v"(cc1));
                        x[k] = v[0]; x[k + 1] = v[1];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 16; k += 2) {
                        unsigned hp;
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(x[k]), "v"(x[k + 1]));
                        asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x[k]) : "v"(hp));
                    }
                }
                if (VKIND == 4 || VKIND == 5) {
                    // plain loads (hipcc keeps the waitcnt book): eight 1-KB wave loads in flight, consumed together
                    const char *gp = reinterpret_cast<const char *>(gbuf) + ((((it * 20 + r) * (VKIND == 5 ? 4 : 1)) & 255) * 4096) + threadIdx.x * 16;
                    gvv[r & 7] = *reinterpret_cast<const f32x4 *>(gp);
                    if ((r & 7) == 7) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) x[q] += gvv[q][0];
                    }
                }
                if (VKIND == 6) {
                    // the same 1-KB wave load straight into LDS (global_load_lds_dwordx4: no VGPR write-back)
                    const char *gp = reinterpret_cast<const char *>(gbuf) + (((it * 20 + r) & 255) * 4096) + threadIdx.x * 16;
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(gp),
                                                     (void __attribute__((address_space(3))) *)(&lds[8192 + (wave - 4) * 1024 + (r & 3) * 256]), 16, 0, 0);
                }
                if (VKIND == 3 && (r & 3) == 0) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(&lds[((threadIdx.x * 4 + r * 64) & 16380)]);
                    x[r & 15] += v[0];
                    *reinterpret_cast<f32x4 *>(&lds[((threadIdx.x * 4 + r * 64 + 8192) & 16380)]) = v;
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += x[k];
        if (s == 12345.f) sink[1] = s;
    }
    if (lane == 0 && blockIdx.x == 0) times[wave] = t1 - t0;
}

static float *g_gbuf = nullptr;
template <int VKIND, int MK = 0>
static int run(const char *what, float *sink, unsigned long long *dt) {
    const int iters = 201;
    for (int prio = 0; prio < 4; prio += 3)
        for (int mode = 1; mode <= 3; ++mode) {
            if (prio && mode != 3) continue;
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_co<VKIND, MK>), dim3(256), dim3(512), 0, 0, mode, iters, prio, sink, dt, g_gbuf);
            CK(hipDeviceSynchronize());
            unsigned long long t[8];
            CK(hipMemcpy(t, dt, sizeof(t), hipMemcpyDeviceToHost));
            const double nm = 120.0 * iters, nv = (VKIND == 0 || VKIND >= 3 ? 320.0 : (VKIND == 1 ? 160.0 : 320.0)) * iters;
            printf("%-28s prio %d mode %d: MFMA wave %.1f cycles per MFMA, VALU wave %.2f cycles per instruction\n", what, prio, mode,
                   (mode & 1) ? t[0] / nm : 0.0, (mode & 2) ? t[4] / nv : 0.0);
        }
    return 0;
}

int main() {
    float *sink; unsigned long long *dt;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&dt, 64));
    CK(hipMalloc(&g_gbuf, 2 << 20)); CK(hipMemset(g_gbuf, 0, 2 << 20));
    if (run<6>("v_fma + 1 global_load_lds / 16", sink, dt)) return 1;
    if (run<6, 2>("... MFMA pairs + LDS", sink, dt)) return 1;
    if (run<6, 3>("v_fma+lds-dma; MFMA+LDS+4 loads/24", sink, dt)) return 1;
    if (run<4>("v_fma + 1 global_load / 16 (vaddr)", sink, dt)) return 1;
    if (run<5>("v_fma + 1 global_load / 16 (other lines)", sink, dt)) return 1;
    if (run<4, 2>("... MFMA pairs + LDS", sink, dt)) return 1;
    if (run<0, 3>("v_fma; MFMA + LDS + 4 loads / 24", sink, dt)) return 1;
    if (run<0, 4>("v_fma; MFMA + LDS + 2 loads / 24", sink, dt)) return 1;
    if (run<4, 3>("v_fma+loads; MFMA+LDS+4 loads/24", sink, dt)) return 1;
    if (run<0, 5>("v_fma; MFMA + LDS + 4 loads spread", sink, dt)) return 1;
    if (run<0, 6>("v_fma; MFMA + LDS + 4 dword loads", sink, dt)) return 1;
    if (run<0, 7>("v_fma; MFMA + LDS + 8 dwordx2 loads", sink, dt)) return 1;
    if (run<4, 5>("v_fma+loads; MFMA+LDS+4 spread", sink, dt)) return 1;
    if (run<0>("v_fma_f32", sink, dt)) return 1;
    if (run<1>("v_pk_fma_f32", sink, dt)) return 1;
    if (run<2>("cvt_pk_f16 + fma_mix", sink, dt)) return 1;
    if (run<3>("v_fma_f32 + LDS b128", sink, dt)) return 1;
    if (run<0, 1>("v_fma_f32, MFMA pairs", sink, dt)) return 1;
    if (run<0, 2>("v_fma_f32, MFMA pairs + LDS", sink, dt)) return 1;
    if (run<3, 2>("v_fma+LDS, MFMA pairs + LDS", sink, dt)) return 1;
    return 0;
}
