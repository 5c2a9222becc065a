// LDS-resident spectral kernels: one workgroup advances one ensemble member.
//
// Restates pyqg 0.7.2 kernel.pyx::{_invert,_do_advection,_do_friction,
// _do_q_subgrid_parameterization,_forward_timestep} (call sites in the reference:
// pyqg_generative/tools/simulate.py:132,137,168; operators.py:233) for the
// ensemble-batched layout (B,2,N,N).  The twelve real 2-D FFTs of one step are
// executed as six complex N x N transforms of packed pairs
//   (u_k + i v_k), ((u_k+U_k) q_k + i v_k q_k), (S_1 + i S_2), (q_1 + i q_2)
// entirely inside one CU's LDS (N <= 96: N*(N+1)*16 B <= 149 KB).
#include "common.hpp"
#include "fft_lds.hpp"
#include "philox.hpp"
#include "diag_acc.hpp"
#include <cstdlib>

namespace qgx {

__device__ __forceinline__ int neg_mod(int j, int N) { return j == 0 ? 0 : N - j; }

struct Grid {
    int N, NK, LD, nrad;
    const int *rad;     // registers/param space
    const int *pos;     // LDS copy
    const double2 *tw;
};

// half-spectra (A,B) of the real parts packed as A + iB, read out of the DIF-ordered field
__device__ __forceinline__ void unpack_pair(const double2 *Z, const Grid &g, int j, int i,
                                            double2 &A, double2 &Bv) {
    const int jm = neg_mod(j, g.N), im = neg_mod(i, g.N);
    const double2 a = Z[g.pos[j] * g.LD + g.pos[i]];
    const double2 b = cconj(Z[g.pos[jm] * g.LD + g.pos[im]]);
    A = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
    // -i/2 * (a - b)
    Bv = make_double2(0.5 * (a.y - b.y), -0.5 * (a.x - b.x));
}

// store the Hermitian extension of (Ah + i Bh) at (j,i) [and its mirror], DIT-input order.
// For the self-conjugate columns the caller passes already symmetrised values.
__device__ __forceinline__ void pack_store(double2 *Z, const Grid &g, int j, int i, double2 Ah,
                                           double2 Bh, double scale) {
    Z[g.pos[j] * g.LD + g.pos[i]] = make_double2((Ah.x - Bh.y) * scale, (Ah.y + Bh.x) * scale);
    if (i != 0 && 2 * i != g.N) {
        const int jm = neg_mod(j, g.N);
        // conj(Ah) + i conj(Bh)
        Z[g.pos[jm] * g.LD + g.pos[g.N - i]] =
            make_double2((Ah.x + Bh.y) * scale, (Bh.x - Ah.y) * scale);
    }
}

__device__ __forceinline__ double2 invert_layer(const SpecDev &d, int k, int idx, double2 q0, double2 q1) {
    const int sz = d.N * d.NK;
    const double a0 = d.a[(2 * k) * sz + idx], a1 = d.a[(2 * k + 1) * sz + idx];
    return make_double2(a0 * q0.x + a1 * q1.x, a0 * q0.y + a1 * q1.y);
}

// Build the packed spectrum of (u_k + i v_k) from qh; optionally store ph_k.
__device__ __forceinline__ void build_uv(double2 *Z, const Grid &g, const SpecDev &d, int k,
                                         const double2 *qh0, const double2 *qh1, double2 *ph_out) {
    const int N = g.N, NK = g.NK;
    for (int idx = threadIdx.x; idx < N * NK; idx += blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        const double2 ph = invert_layer(d, k, idx, qh0[idx], qh1[idx]);
        if (ph_out) ph_out[idx] = ph;
        const double kx = d.kk[i], ly = d.ll[j];
        // uh = -i l ph ; vh = i k ph
        double2 uh = make_double2(ly * ph.y, -ly * ph.x);
        double2 vh = make_double2(-kx * ph.y, kx * ph.x);
        if (i == 0 || 2 * i == N) {
            const int jm = neg_mod(j, N);
            const int idm = jm * NK + i;
            const double2 pm = invert_layer(d, k, idm, qh0[idm], qh1[idm]);
            const double lm = d.ll[jm];
            const double2 um = make_double2(lm * pm.y, -lm * pm.x);
            const double2 vm = make_double2(-kx * pm.y, kx * pm.x);
            uh = make_double2(0.5 * (uh.x + um.x), 0.5 * (uh.y - um.y));
            vh = make_double2(0.5 * (vh.x + vm.x), 0.5 * (vh.y - vm.y));
        }
        pack_store(Z, g, j, i, uh, vh, d.invN2);
    }
}

// Build the packed spectrum of (A + i B) from two half spectra in global memory (Bh == nullptr: B = 0).
__device__ __forceinline__ void build_pair(double2 *Z, const Grid &g, const double2 *Ah,
                                           const double2 *Bh, double scale) {
    const int N = g.N, NK = g.NK;
    for (int idx = threadIdx.x; idx < N * NK; idx += blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        double2 a = Ah[idx], b = Bh ? Bh[idx] : make_double2(0., 0.);
        if (i == 0 || 2 * i == N) {
            const int idm = neg_mod(j, N) * NK + i;
            const double2 am = Ah[idm], bm = Bh ? Bh[idm] : make_double2(0., 0.);
            a = make_double2(0.5 * (a.x + am.x), 0.5 * (a.y - am.y));
            b = make_double2(0.5 * (b.x + bm.x), 0.5 * (b.y - bm.y));
        }
        pack_store(Z, g, j, i, a, b, scale);
    }
}

__device__ __forceinline__ Grid make_grid(const SpecDev &d, double2 *Z, int *&pos_lds) {
    Grid g;
    g.N = d.N; g.NK = d.NK; g.LD = d.LD; g.nrad = d.nrad; g.rad = d.rad; g.tw = d.tw;
    pos_lds = reinterpret_cast<int *>(Z + d.N * d.LD);
    // twiddle table in LDS too: read from global memory, every butterfly of a transform's first pass waits for an
    // L1 / L2 round trip (the large-grid kernels measured 60-70 % wait that way)
    double2 *tw_lds = reinterpret_cast<double2 *>(pos_lds + ((d.N + 3) & ~3));
    for (int t = threadIdx.x; t < d.N; t += blockDim.x) { pos_lds[t] = d.pos[t]; tw_lds[t] = d.tw[t]; }
    g.pos = pos_lds;
    g.tw = tw_lds;
    return g;
}

extern __shared__ __attribute__((aligned(16))) char qgx_smem[];

// ------------------------------------------------------------------ one time step
// LSPLIT: two workgroups per member, one per layer.  Nothing couples the layers inside a step except the
// inversion (both workgroups read both layers of qh_in), so workgroup k runs the chain of layer k only:
// forcing S_k, (u_k, v_k), advection of layer k, its tendency and AB3 update, q_k — four 2-D FFTs instead
// of six, with the single real fields packed as (x + 0 i).  Used while 2B workgroups are all resident
// (the kernel is a latency chain then: B = 128 at 64x64 leaves half of the CUs idle with one workgroup per
// member); 8 FFTs per member instead of 6 make it slower once members queue up.
// PART (LSPLIT only): the step as two kernels.  1 = the half that needs nothing of the forcing — inversion, (u, v), the
// advection products and their transform, the tendency without its forcing term, stored in the slot of the new tendency;
// 2 = the rest — forcing S_k and its transform, tendency + forcing, AB3 update, q_k (and the generator work of GenFuse).
// Half 1 depends on the previous step alone, so it runs on a side stream UNDER the generator's layers (model.hip); the
// arithmetic and its order are those of the whole kernel: bit-identical results.
template <int NN, bool LSPLIT = false, int PART = 0>
__global__ void k_step_small(SpecDev d, StepArgs a) {
    static_assert(PART == 0 || LSPLIT, "the two-kernel step exists in layer-split form only");
    // PART 3: both halves in ONE launch of 4 B workgroups.  Blocks [0, 2 B) transform the forcing of their (member, layer),
    // publish it (agent-scope release: fence by every thread, barrier, flag) and leave; blocks [2 B, 4 B) run the
    // inversion / advection chain meanwhile and wait for the flag (acquire) in front of the time-step loop: three transforms
    // in the chain instead of four.  Used while all 4 B workgroups are resident at once (one per CU), so nobody waits for a
    // workgroup that has no CU; the wait is bounded all the same, and a timed-out step poisons its state (NaN: loud).
    constexpr bool SIB = PART == 3;
    const bool roleF = SIB && blockIdx.x < 2u * (unsigned)d.B;
    // the chain workgroups start at a multiple of 8, so that (workgroups being dealt to the 8 XCDs round-robin) a pair of
    // siblings shares an XCD and its L2 — checked at run time, see below; the blocks in between have nothing to do
    const unsigned m0 = (2u * (unsigned)d.B + 7u) & ~7u;
    if (SIB && !roleF && blockIdx.x < m0) return;
    const unsigned bid = SIB && !roleF ? blockIdx.x - m0 : blockIdx.x;
    const unsigned my_xcc = SIB ? (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u) : 0u;     // HW_REG_XCC_ID
    if (SIB && !roleF && threadIdx.x == 0)      // where the chain workgroup runs, tagged with the launch's epoch
        __hip_atomic_store(a.sib_flag + 2 * (size_t)d.B + bid, (a.sib_epoch << 4) | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool fside = PART != 1 && (!SIB || roleF);     // this workgroup does the forcing side (noise, output kernel, forcing transform)
    double2 *Z = reinterpret_cast<double2 *>(qgx_smem);
    int *pos_lds;
    Grid g = make_grid(d, Z, pos_lds);
    if (NN) { g.N = NN; g.NK = NN / 2 + 1; g.LD = NN + 1; }
    const int N = NN ? NN : d.N, NK = NN ? NN / 2 + 1 : d.NK, LD = NN ? NN + 1 : d.LD;
    const int b = LSPLIT ? (int)(bid >> 1) : (int)blockIdx.x;
    const int kown = LSPLIT ? (int)(bid & 1) : 0;
    const size_t so = (size_t)b * 2 * N * NK, ro = (size_t)b * 2 * N * N;
    const int sz = N * NK, rz = N * N;
    const double2 *qh0 = a.qh_in + so, *qh1 = qh0 + sz;
    float in_max = 0.f;
    // Latency chain (one workgroup per CU, nobody to hide a round trip to memory behind): the raw generator output of the
    // folded output kernel is fetched BEFORE the Philox arithmetic of the next step's noise, the real-space q of the advection
    // products before the inverse transform in front of them, and the new spectral state goes from the time-step loop to
    // the last inverse transform in registers.  The arithmetic and its order are untouched.
    constexpr bool PF = LSPLIT && NN != 0;           // (compile-time sizes, 1024 threads)
    constexpr int NPXL = PF ? (NN * NN + 1023) / 1024 : 1, NSPL = PF ? (NN * (NN / 2 + 1) + 1023) / 1024 : 1;
    constexpr bool PFY = PF && NN <= 64;             // (96 x 96: nine pixels per thread leave no registers for it)
    float yraw[PFY ? 16 : 1];
    if (PFY && fside && a.has_S && a.gf.y) {
        const float *yk = a.gf.y + ((size_t)b * 2 + kown) * rz;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = u * 1024 + (int)threadIdx.x;
            yraw[u] = i < rz ? yk[i] : 0.f;
        }
    }
    if (LSPLIT && fside && a.gf.X) {
        // next step's latent channel first: it depends on nothing this kernel computes, and here its arithmetic
        // (Philox rounds, log, sincos) runs while the first global loads of the step are in flight
        // latent channel kown: z = b * xi, white in time (k_prep_noise with a == 0; quads of the flat (2, N*N) field)
        for (int ql = threadIdx.x; ql < rz / 4; ql += blockDim.x) {
            const int quad = kown * (rz / 4) + ql;
            float xi[4];
            philox_normal4(a.gf.seed, a.gf.member_offset + b, a.gf.step, (uint32_t)quad, xi);
            const size_t o = (size_t)b * 2 * rz + 4 * (size_t)quad;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float zn = a.gf.b * xi[e];
                a.gf.z[o + e] = zn;
                a.gf.X[(size_t)b * 4 * rz + 2 * (size_t)rz + 4 * (size_t)quad + e] = zn;
                in_max = fmaxf(in_max, zn != zn ? __uint_as_float(0x7f800000u) : fabsf(zn));
            }
        }
    }
    __syncthreads();

    // ---- subgrid forcing: Sh_k = rfft2(weight * S_k), pair packed (pyqg _do_q_subgrid_parameterization)
    if (fside && a.has_S) {
        const double *S0 = a.S + ro, *S1 = S0 + rz;
        if (LSPLIT && a.gf.y) {
            // the generator's output kernel folded in (k_finish<FIN_PLAIN>, conv.hip: same arithmetic, same summation order —
            // 1024 threads, per-thread partial sums over u, wave shuffles, waves in order): S = double(y * y_std) - mean
            constexpr int KEEP = 16;
            __shared__ double fin_sm[16];
            __shared__ double fin_mean;
            const float *yk = a.gf.y + ((size_t)b * 2 + kown) * rz;
            const float *y1k = a.gf.y1 ? a.gf.y1 + ((size_t)b * 2 + kown) * rz : nullptr;   // (k_finish<FIN_SUM>)
            const float ys = a.gf.ys[kown];
            double keep[KEEP];
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < KEEP; ++u) {
                const int i = u * (int)blockDim.x + (int)threadIdx.x;
                if constexpr (PFY) keep[u] = i < rz ? (double)((y1k ? yraw[u] + y1k[i] : yraw[u]) * ys) : 0.0;
                else keep[u] = i < rz ? (double)((y1k ? yk[i] + y1k[i] : yk[i]) * ys) : 0.0;
                acc += keep[u];
            }
            double mu = 0.0;
            if (a.gf.demean) {
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
                if ((threadIdx.x & 63) == 0) fin_sm[threadIdx.x >> 6] = acc;
                __syncthreads();
                if (threadIdx.x == 0) {
                    double t = 0;
                    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += fin_sm[w];
                    fin_mean = t / (double)rz;
                }
                __syncthreads();
                mu = fin_mean;
            }
            double *Sk = const_cast<double *>(kown ? S1 : S0);
            bool bad = false;
#pragma unroll
            for (int u = 0; u < KEEP; ++u) {
                const int i = u * (int)blockDim.x + (int)threadIdx.x;
                if (i < rz) {
                    const double sv = keep[u] - mu;
                    Sk[i] = sv;
                    const int y = i / N, x = i - y * N;
                    Z[y * LD + x] = make_double2(a.weight * sv, 0.);
                }
                bad |= !(fabs(keep[u]) <= 1.79e308);
            }
            if (bad) atomicOr(a.gf.range, 0x80000000u);      // a non-finite forcing never reaches the model unnoticed
        } else {
            for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
                const int y = idx / N, x = idx - y * N;
                if (LSPLIT) Z[y * LD + x] = make_double2(a.weight * (kown ? S1 : S0)[idx], 0.);
                else Z[y * LD + x] = make_double2(a.weight * S0[idx], a.weight * S1[idx]);
            }
        }
        __syncthreads();
        fft2d_fwd_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
        for (int idx = threadIdx.x; idx < sz; idx += blockDim.x) {
            const int j = idx / NK, i = idx - j * NK;
            double2 s0, s1;
            unpack_pair(Z, g, j, i, s0, s1);
            if (a.demean && idx == 0) { s0 = make_double2(0., 0.); s1 = s0; }
            if (LSPLIT) {
                a.dqh[so + kown * sz + idx] = s0;
            } else {
                a.dqh[so + idx] = s0;
                a.dqh[so + sz + idx] = s1;
            }
        }
        __syncthreads();
    }

    if (SIB && roleF) {
        // the forcing side is done: largest |latent noise| for the range guard, then publish
        if (a.gf.X) {
            for (int o = 32; o > 0; o >>= 1) in_max = fmaxf(in_max, __shfl_down(in_max, o));
            if ((threadIdx.x & 63) == 0 && __float_as_uint(in_max) > __builtin_nontemporal_load(a.gf.range + 1))
                atomicMax(a.gf.range + 1, __float_as_uint(in_max));
        }
        // Same XCD as the sibling (it said so for THIS launch): both ends share one L2, so it is enough that every wave's
        // stores have left the CU (write-through L1, vmcnt drained) before the flag goes to that L2 — the exchange of
        // spectral_large.hip's team barrier.  Otherwise (another XCD, or the sibling has not started yet): agent-scope
        // release, i.e. the L2's dirty lines written back, and the flag tells the sibling to invalidate on its side.
        __shared__ int sib_same;
        if (threadIdx.x == 0)
            sib_same = __hip_atomic_load(a.sib_flag + 2 * (size_t)d.B + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ((a.sib_epoch << 4) | my_xcc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool same = sib_same != 0 && !a.sib_full;
        if (!same) { __threadfence(); __syncthreads(); }
        if (threadIdx.x == 0) {
            if (same) __hip_atomic_store(a.sib_flag + bid, a.sib_epoch << 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(a.sib_flag + bid, (a.sib_epoch << 1) | 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    bool sib_lost = false;
    double2 qnew[NSPL];
    for (int k = kown; k < (LSPLIT ? kown + 1 : 2); ++k) {
        if constexpr (PART != 2) {
            // ---- _invert: ph_k, (u_k, v_k) = irfft2(-il ph, ik ph)
            build_uv(Z, g, d, k, qh0, qh1, a.diag ? a.ph + so + k * sz : nullptr);
            const double *qk = a.q + ro + k * rz;
            double qpre[NPXL];
            if constexpr (PF) {
#pragma unroll
                for (int r = 0; r < NPXL; ++r) { const int idx = (int)threadIdx.x + r * 1024; qpre[r] = idx < rz ? qk[idx] : 0.0; }
            }
            __syncthreads();
            fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
            // ---- _do_advection, real space: uq = (u+U) q, vq = v q
            {
                const double Uk = d.U[k];
                if constexpr (PF) {
#pragma unroll
                    for (int r = 0; r < NPXL; ++r) {
                        const int idx = (int)threadIdx.x + r * 1024;
                        if (idx < rz) {
                            const int y = idx / N, x = idx - y * N;
                            const double2 uv = Z[y * LD + x];
                            if (a.diag) { a.u[ro + k * rz + idx] = uv.x; a.v[ro + k * rz + idx] = uv.y; }
                            const double qv = qpre[r];
                            Z[y * LD + x] = make_double2((uv.x + Uk) * qv, uv.y * qv);
                        }
                    }
                } else {
                    for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
                        const int y = idx / N, x = idx - y * N;
                        const double2 uv = Z[y * LD + x];
                        if (a.diag) { a.u[ro + k * rz + idx] = uv.x; a.v[ro + k * rz + idx] = uv.y; }
                        const double qv = qk[idx];
                        Z[y * LD + x] = make_double2((uv.x + Uk) * qv, uv.y * qv);
                    }
                }
            }
            __syncthreads();
            fft2d_fwd_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
        }
        if (SIB && a.has_S) {
            // the forcing's spectrum comes from the sibling workgroup
            __shared__ int sib_ok;
            if (threadIdx.x == 0) {
                int spins = 0, ok = 0;
                for (;;) {
                    const unsigned long long f = __hip_atomic_load(a.sib_flag + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((f >> 1) == a.sib_epoch) { ok = 1 + (int)(f & 1ull); break; }
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 23)) break;
                }
                sib_ok = ok;
            }
            __syncthreads();
            // 2: published across XCDs — every wave invalidates (agent-scope acquire); 1: the sibling shares this L2 and this
            // CU's L1 holds no line of dqh yet (never read in this launch): nothing to do
            if (sib_ok == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            sib_lost = sib_ok == 0;
        }
        // ---- spectral tendency, friction, forcing, AB3 + filter (_forward_timestep)
        // (PF: element r of a thread is idx = tid + 1024 r, which is also what this loop visits in its r-th trip)
        int trip = 0;
        for (int idx = threadIdx.x; idx < sz; idx += blockDim.x, ++trip) {
            const int j = idx / NK, i = idx - j * NK;
            const size_t o = so + k * sz + idx;
            const double2 q0 = qh0[idx], q1 = qh1[idx];
            double tx, ty;
            if constexpr (PART != 2) {
                double2 uqh, vqh;
                unpack_pair(Z, g, j, i, uqh, vqh);
                const double2 ph = invert_layer(d, k, idx, q0, q1);
                const double kx = d.kk[i], ly = d.ll[j];
                const double kq = kx * d.Qy[k];
                // -(ik uqh + il vqh + ik Qy ph)
                tx = (kx * uqh.y + ly * vqh.y + kq * ph.y);
                ty = -(kx * uqh.x + ly * vqh.x + kq * ph.x);
                if (k == 1 && d.rek != 0.0) {
                    const double f = d.rek * d.wv2[idx];
                    tx += f * ph.x;
                    ty += f * ph.y;
                }
            } else {
                const double2 t = a.dq_new[o];      // what half 1 stored
                tx = t.x; ty = t.y;
            }
            if constexpr (PART == 1) {
                a.dq_new[o] = make_double2(tx, ty);
            } else {
                if (a.has_S) {
                    const double2 s = a.dqh[so + k * sz + idx];
                    tx += s.x;
                    ty += s.y;
                }
                const double2 p = a.dq_p[o], pp = a.dq_pp[o];
                const double2 qk = k == 0 ? q0 : q1;
                const double f = d.filtr[idx];
                a.dq_new[o] = make_double2(tx, ty);
                double2 qn = make_double2(f * (qk.x + a.dt1 * tx + a.dt2 * p.x + a.dt3 * pp.x),
                                          f * (qk.y + a.dt1 * ty + a.dt2 * p.y + a.dt3 * pp.y));
                if (SIB && sib_lost) qn = make_double2(__longlong_as_double(0x7ff8000000000000ll), 0.);   // never a silently wrong state
                a.qh_out[o] = qn;
                if constexpr (PF) {
#pragma unroll
                    for (int r = 0; r < NSPL; ++r) if (r == trip) qnew[r] = qn;
                }
            }
        }
        __syncthreads();
    }
    if constexpr (PART == 1) return;
    // ---- q^{n+1} = irfft2(qh^{n+1}), both layers packed (LSPLIT: the own layer alone)
    if constexpr (PF) {
        // build_pair(qh_out_k, nullptr) with the own elements from registers; the mirrors of the two self-conjugate columns
        // were written by other threads (behind the barrier above)
        const double2 *Ah = a.qh_out + so + kown * sz;
#pragma unroll
        for (int r = 0; r < NSPL; ++r) {
            const int idx = (int)threadIdx.x + r * 1024;
            if (idx < sz) {
                const int j = idx / NK, i = idx - j * NK;
                double2 av = qnew[r], bv = make_double2(0., 0.);
                if (i == 0 || 2 * i == N) {
                    const int idm = neg_mod(j, N) * NK + i;
                    const double2 am = Ah[idm], bm = make_double2(0., 0.);
                    av = make_double2(0.5 * (av.x + am.x), 0.5 * (av.y - am.y));
                    bv = make_double2(0.5 * (bv.x + bm.x), 0.5 * (bv.y - bm.y));
                }
                pack_store(Z, g, j, i, av, bv, d.invN2);
            }
        }
    } else if (LSPLIT) build_pair(Z, g, a.qh_out + so + kown * sz, nullptr, d.invN2);
    else build_pair(Z, g, a.qh_out + so, a.qh_out + so + sz, d.invN2);
    __syncthreads();
    fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
    for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
        const int y = idx / N, x = idx - y * N;
        const double2 w = Z[y * LD + x];
        if (LSPLIT) {
            a.q[ro + kown * rz + idx] = w.x;
            if (a.gf.X) {       // the next step's network input, channel kown: float(q) / x_std (k_prep_noise, conv.hip)
                const float xq = (float)w.x / a.gf.xs[kown];
                a.gf.X[((size_t)b * 4 + kown) * rz + idx] = xq;
                in_max = fmaxf(in_max, xq != xq ? __uint_as_float(0x7f800000u) : fabsf(xq));
            }
        } else {
            a.q[ro + idx] = w.x;
            a.q[ro + rz + idx] = w.y;
        }
    }
    if (LSPLIT && a.gf.X) {
        // largest |network input| for the f16x3 range guard (input_absmax, conv.hip)
        for (int o = 32; o > 0; o >>= 1) in_max = fmaxf(in_max, __shfl_down(in_max, o));
        if ((threadIdx.x & 63) == 0 && __float_as_uint(in_max) > __builtin_nontemporal_load(a.gf.range + 1))
            atomicMax(a.gf.range + 1, __float_as_uint(in_max));
    }
}

// ------------------------------------------------------------------ q -> qh (property q setter)
template <int NN>
__global__ void k_q_to_qh_small(SpecDev d, const double *q, double2 *qh) {
    double2 *Z = reinterpret_cast<double2 *>(qgx_smem);
    int *pos_lds;
    Grid g = make_grid(d, Z, pos_lds);
    if (NN) { g.N = NN; g.NK = NN / 2 + 1; g.LD = NN + 1; }
    const int N = NN ? NN : d.N, NK = NN ? NN / 2 + 1 : d.NK, LD = NN ? NN + 1 : d.LD, b = blockIdx.x;
    const size_t so = (size_t)b * 2 * N * NK, ro = (size_t)b * 2 * N * N;
    const int sz = N * NK, rz = N * N;
    for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
        const int y = idx / N, x = idx - y * N;
        Z[y * LD + x] = make_double2(q[ro + idx], q[ro + rz + idx]);
    }
    __syncthreads();
    fft2d_fwd_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
    for (int idx = threadIdx.x; idx < sz; idx += blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        double2 s0, s1;
        unpack_pair(Z, g, j, i, s0, s1);
        qh[so + idx] = s0;
        qh[so + sz + idx] = s1;
    }
}

// ------------------------------------------------------------------ qh -> q
template <int NN>
__global__ void k_qh_to_q_small(SpecDev d, const double2 *qh, double *q) {
    double2 *Z = reinterpret_cast<double2 *>(qgx_smem);
    int *pos_lds;
    Grid g = make_grid(d, Z, pos_lds);
    if (NN) { g.N = NN; g.NK = NN / 2 + 1; g.LD = NN + 1; }
    const int N = NN ? NN : d.N, NK = NN ? NN / 2 + 1 : d.NK, LD = NN ? NN + 1 : d.LD, b = blockIdx.x;
    const size_t so = (size_t)b * 2 * N * NK, ro = (size_t)b * 2 * N * N;
    const int sz = N * NK, rz = N * N;
    __syncthreads();
    build_pair(Z, g, qh + so, qh + so + sz, d.invN2);
    __syncthreads();
    fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
    for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
        const int y = idx / N, x = idx - y * N;
        const double2 w = Z[y * LD + x];
        q[ro + idx] = w.x;
        q[ro + rz + idx] = w.y;
    }
}

// ------------------------------------------------------------------ _invert only
template <int NN>
__global__ void k_invert_small(SpecDev d, const double2 *qh, double2 *ph, double *u, double *v) {
    double2 *Z = reinterpret_cast<double2 *>(qgx_smem);
    int *pos_lds;
    Grid g = make_grid(d, Z, pos_lds);
    if (NN) { g.N = NN; g.NK = NN / 2 + 1; g.LD = NN + 1; }
    const int N = NN ? NN : d.N, NK = NN ? NN / 2 + 1 : d.NK, LD = NN ? NN + 1 : d.LD, b = blockIdx.x;
    const size_t so = (size_t)b * 2 * N * NK, ro = (size_t)b * 2 * N * N;
    const int sz = N * NK, rz = N * N;
    __syncthreads();
    for (int k = 0; k < 2; ++k) {
        build_uv(Z, g, d, k, qh + so, qh + so + sz, ph + so + k * sz);
        __syncthreads();
        fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
        for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
            const int y = idx / N, x = idx - y * N;
            const double2 uv = Z[y * LD + x];
            u[ro + k * rz + idx] = uv.x;
            v[ro + k * rz + idx] = uv.y;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ one increment of the time-averaged diagnostics
// model.py::_calc_diagnostics on the small grids, a workgroup per member: _invert (psi, u, v of the current state),
// p = irfft2(psi), xi = irfft2(-K^2 psi), the three pairs of real-space products, their transforms and the forcing's,
// accumulation — what diag.hip runs as nine launches on the large grids.  Real-space fields pass through the model's own
// arrays (u, v: kept, as _invert leaves them) and two scratch arrays; every global value is re-read by the thread that
// wrote it or after a workgroup barrier.  The arithmetic is that of k_invert_small, k_qh_to_q_small, k_diag_xih,
// k_diag_products, k_q_to_qh_small, k_diag_scale_S and k_diag_accumulate, expression by expression.
template <int NN>
__global__ void k_diag_small(SpecDev d, DiagConst c, const double2 *qh, double2 *ph, double *u, double *v, double *P, double *XI,
                             double2 *S3, double2 *S4, double2 *S5, double2 *Sh, double2 *S6, double2 *S7, const double *S, double weight,
                             const double *q, const double2 *dq_p, const double2 *dq_pp, DiagAcc acc) {
    double2 *Z = reinterpret_cast<double2 *>(qgx_smem);
    int *pos_lds;
    Grid g = make_grid(d, Z, pos_lds);
    if (NN) { g.N = NN; g.NK = NN / 2 + 1; g.LD = NN + 1; }
    const int N = NN ? NN : d.N, NK = NN ? NN / 2 + 1 : d.NK, LD = NN ? NN + 1 : d.LD, b = blockIdx.x;
    const size_t so = (size_t)b * 2 * N * NK, ro = (size_t)b * 2 * N * N;
    const int sz = N * NK, rz = N * N;
    __syncthreads();
    // ---- _invert
    for (int k = 0; k < 2; ++k) {
        build_uv(Z, g, d, k, qh + so, qh + so + sz, ph + so + k * sz);
        __syncthreads();
        fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
        for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
            const int y = idx / N, x = idx - y * N;
            const double2 uv = Z[y * LD + x];
            u[ro + k * rz + idx] = uv.x;
            v[ro + k * rz + idx] = uv.y;
        }
        __syncthreads();
    }
    // ---- p = irfft2(psi), both layers as one packed pair
    build_pair(Z, g, ph + so, ph + so + sz, d.invN2);
    __syncthreads();
    fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
    for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
        const int y = idx / N, x = idx - y * N;
        const double2 w = Z[y * LD + x];
        P[ro + idx] = w.x;
        P[ro + rz + idx] = w.y;
    }
    __syncthreads();
    // ---- xi = irfft2(-K^2 psi): the spectrum goes through S3 (free until the first product transform)
    for (int idx = threadIdx.x; idx < 2 * sz; idx += blockDim.x) {
        const double w = -d.wv2[idx % sz];
        const double2 p = ph[so + idx];
        S3[so + idx] = make_double2(w * p.x, w * p.y);
    }
    __syncthreads();
    build_pair(Z, g, S3 + so, S3 + so + sz, d.invN2);
    __syncthreads();
    fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
    for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
        const int y = idx / N, x = idx - y * N;
        const double2 w = Z[y * LD + x];
        XI[ro + idx] = w.x;
        XI[ro + rz + idx] = w.y;
    }
    __syncthreads();
    // ---- the five pairs of products and the forcing, each: stage the pair, forward transform, unpack
    for (int which = 0; which < 6; ++which) {
        if (which == 3 && !S) continue;
        for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
            const int y = idx / N, x = idx - y * N;
            const size_t o = ro + idx;
            double ra, rb;
            if (which == 0) {
                const double u1 = u[o], u2 = u[o + rz], v1 = v[o], v2 = v[o + rz];
                const double ptpc = P[o] - P[o + rz];
                const double ub = c.del1 * u1 + c.del2 * u2, vb = c.del1 * v1 + c.del2 * v2;
                ra = ub * ptpc; rb = vb * ptpc;
            } else if (which == 1) {
                const double x1 = XI[o];
                ra = u[o] * x1; rb = v[o] * x1;
            } else if (which == 2) {
                const double x2 = XI[o + rz];
                ra = u[o + rz] * x2; rb = v[o + rz] * x2;
            } else if (which == 3) {
                ra = weight * S[o]; rb = weight * S[o + rz];
            } else if (which == 4) {            // (u_1 q_1, v_1 q_1): the enstrophy flux and the tendency of Dissspec
                const double q1 = q[o];
                ra = u[o] * q1; rb = v[o] * q1;
            } else {
                const double q2 = q[o + rz];
                ra = u[o + rz] * q2; rb = v[o + rz] * q2;
            }
            Z[y * LD + x] = make_double2(ra, rb);
        }
        __syncthreads();
        fft2d_fwd_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
        double2 *dst = which == 0 ? S3 : (which == 1 ? S4 : (which == 2 ? S5 : (which == 3 ? Sh : (which == 4 ? S6 : S7))));
        for (int idx = threadIdx.x; idx < sz; idx += blockDim.x) {
            const int j = idx / NK, i = idx - j * NK;
            double2 s0, s1;
            unpack_pair(Z, g, j, i, s0, s1);
            dst[so + idx] = s0;
            dst[so + sz + idx] = s1;
        }
        __syncthreads();
    }
    // ---- accumulate (each thread re-reads the transforms it unpacked itself)
    for (int idx = threadIdx.x; idx < sz; idx += blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        const size_t o = so + idx, o2 = (size_t)b * sz + idx;
        const double2 zero = make_double2(0., 0.);
        diag_accumulate_elem(d, c, acc, idx, i, j, o, o2, sz, qh[o], qh[o + sz], ph[o], ph[o + sz], S3[o], S3[o + sz], S4[o], S4[o + sz],
                             S5[o], S5[o + sz], S != nullptr, S ? Sh[o] : zero, S ? Sh[o + sz] : zero, S6[o], S6[o + sz], S7[o], S7[o + sz],
                             dq_p[o], dq_p[o + sz], dq_pp[o], dq_pp[o + sz]);
    }
}

// ------------------------------------------------------------------ the same increment with its work fields in registers
// k_diag_small moves 3.6 MB per member (468 MB per 128-member increment, PMC) where the increment's own bytes are 0.94 MB
// (qh, q, S, the two older tendencies, sixteen accumulators read and written once): every real-space field (u, v, p, xi: 260
// KB) goes to HBM and comes back two or three times, the six product spectra (810 KB) go out and come back for the
// accumulation.  Here a thread keeps the pixels it owns of (u_1, v_1, u_2, v_2) in registers for the whole kernel, forms each
// product pair straight into the LDS field it has just read, and accumulates as soon as a transform is in LDS: APEflux
// after the (ub ptpc, vb ptpc) transform, KEflux after the two (u_k xi_k, v_k xi_k) ones (J_1 held in registers), the rest
// after the last transform (the forcing's and the layer-1 enstrophy flux's spectra held in registers).  Same device
// functions, same expressions, same element -> thread map: the accumulators are BIT-identical to k_diag_small's.  It stores
// ph but no u, v (the caller marks them stale: qgx_get inverts on demand).  Grids up to 64 x 64 (at 96 x 96 nine pixels per
// thread do not fit the 128 registers of a 1024-thread workgroup).
// LDS of k_diag_small_reg: the field + tables of small_lds_bytes(), then the forcing's two half spectra
__host__ __device__ constexpr size_t diag_reg_side_offset(int N) {
    return (((size_t)N * (N + 1) * sizeof(double2) + (size_t)((N + 3) & ~3) * sizeof(int) + (size_t)N * sizeof(double2)) + 15) & ~(size_t)15;
}
__host__ __device__ constexpr size_t diag_reg_lds_bytes(int N) { return diag_reg_side_offset(N) + (size_t)2 * N * (N / 2 + 1) * sizeof(double2); }
// HALF: the increment of a member as two workgroups (blockIdx = 2 member + half) — 1: the inversion, p, xi and the three
// product transforms behind APEflux and KEflux (7 transforms in a chain), 2: the inversion, the forcing's transform and the
// two enstrophy-flux ones, with every other accumulator (5) — instead of ten in one chain on half of the CUs (128 members).
// Both halves invert and both store the same psi; the accumulators they add to are disjoint.  0: one workgroup per member.
template <int NN, int HALF = 0>
__global__ __launch_bounds__(1024) void k_diag_small_reg(SpecDev d, DiagConst c, const double2 *qh, double2 *ph, const double *S, double weight,
                                                         const double *q, const double2 *dq_p, const double2 *dq_pp, DiagAcc acc) {
    double2 *Z = reinterpret_cast<double2 *>(qgx_smem);
    int *pos_lds;
    Grid g = make_grid(d, Z, pos_lds);
    g.N = NN; g.NK = NN / 2 + 1; g.LD = NN + 1;
    constexpr int N = NN, NK = NN / 2 + 1, LD = NN + 1, sz = N * NK, rz = N * N;
    constexpr int NPX = (rz + 1023) / 1024, NSP = (sz + 1023) / 1024;
    const int b = HALF ? blockIdx.x >> 1 : blockIdx.x;
    const int half = HALF ? 1 + (int)(blockIdx.x & 1) : 0;      // (workgroup-uniform)
    const size_t so = (size_t)b * 2 * sz, ro = (size_t)b * 2 * rz;
    const double2 *qh0 = qh + so, *qh1 = qh + so + sz;
    double u1[NPX], v1[NPX], u2[NPX], v2[NPX];
    __syncthreads();
    // ---- _invert: (u_k + i v_k) of both layers into registers
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        build_uv(Z, g, d, k, qh0, qh1, ph + so + k * sz);
        __syncthreads();
        fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
#pragma unroll
        for (int r = 0; r < NPX; ++r) {
            const int idx = threadIdx.x + r * 1024;
            if (idx < rz) {
                const int y = idx / N, x = idx - y * N;
                const double2 uv = Z[y * LD + x];
                if (k == 0) { u1[r] = uv.x; v1[r] = uv.y; } else { u2[r] = uv.x; v2[r] = uv.y; }
            }
        }
        __syncthreads();
    }
    // psi is stored once (68 KB per member) and read back wherever k_diag_small reads it: forming it anew from qh at each use
    // would leave it to the compiler which of the two products of a0 q0 + a1 q1 it fuses — a last-bit difference between
    // uses.  The packed spectrum of (f(psi_1) + i f(psi_2)); XIH: f = -K^2 psi (k_diag_small stages that through S3)
    const double2 *ph0 = ph + so, *ph1 = ph + so + sz;
    auto build_psi_pair = [&](bool XIH) {
        for (int idx = threadIdx.x; idx < sz; idx += 1024) {
            const int j = idx / NK, i = idx - j * NK;
            double2 a = ph0[idx], bb = ph1[idx];
            if (XIH) { const double w = -d.wv2[idx]; a = make_double2(w * a.x, w * a.y); bb = make_double2(w * bb.x, w * bb.y); }
            if (i == 0 || 2 * i == N) {
                const int idm = neg_mod(j, N) * NK + i;
                double2 am = ph0[idm], bm = ph1[idm];
                if (XIH) { const double w = -d.wv2[idm]; am = make_double2(w * am.x, w * am.y); bm = make_double2(w * bm.x, w * bm.y); }
                a = make_double2(0.5 * (a.x + am.x), 0.5 * (a.y - am.y));
                bb = make_double2(0.5 * (bb.x + bm.x), 0.5 * (bb.y - bm.y));
            }
            pack_store(Z, g, j, i, a, bb, d.invN2);
        }
    };
    const double2 zero = make_double2(0., 0.);
    if (half != 2) {
    // ---- p = irfft2(psi): ptpc = p_1 - p_2 is used once, by the first product pair, which goes straight back into the field
    build_psi_pair(false);
    __syncthreads();
    fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
#pragma unroll
    for (int r = 0; r < NPX; ++r) {
        const int idx = threadIdx.x + r * 1024;
        if (idx < rz) {
            const int y = idx / N, x = idx - y * N;
            const double2 w = Z[y * LD + x];
            const double ptpc = w.x - w.y;
            const double ub = c.del1 * u1[r] + c.del2 * u2[r], vb = c.del1 * v1[r] + c.del2 * v2[r];
            Z[y * LD + x] = make_double2(ub * ptpc, vb * ptpc);
        }
    }
    __syncthreads();
    fft2d_fwd_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
#pragma unroll
    for (int r = 0; r < NSP; ++r) {
        const int idx = threadIdx.x + r * 1024;
        if (idx < sz) {
            const int j = idx / NK, i = idx - j * NK;
            double2 A3, B3;
            unpack_pair(Z, g, j, i, A3, B3);
            const double2 p1 = ph0[idx], p2 = ph1[idx];
            diag_accumulate_elem<1>(d, c, acc, idx, i, j, so + idx, (size_t)b * sz + idx, sz, zero, zero, p1, p2, A3, B3, zero, zero, zero, zero,
                                    false, zero, zero, zero, zero, zero, zero, zero, zero, zero, zero);
        }
    }
    __syncthreads();
    // ---- xi = irfft2(-K^2 psi); (u_1 xi_1, v_1 xi_1) goes back into the field, xi_2 waits in registers
    build_psi_pair(true);
    __syncthreads();
    fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
    double xi2[NPX];
#pragma unroll
    for (int r = 0; r < NPX; ++r) {
        const int idx = threadIdx.x + r * 1024;
        if (idx < rz) {
            const int y = idx / N, x = idx - y * N;
            const double2 w = Z[y * LD + x];
            xi2[r] = w.y;
            Z[y * LD + x] = make_double2(u1[r] * w.x, v1[r] * w.x);
        }
    }
    __syncthreads();
    fft2d_fwd_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
    double2 J1[NSP];                                // Jacobian of layer 1, i k A + i l B (diag_acc.hpp), held for KEflux
#pragma unroll
    for (int r = 0; r < NSP; ++r) {
        const int idx = threadIdx.x + r * 1024;
        if (idx < sz) {
            const int j = idx / NK, i = idx - j * NK;
            double2 A4, B4;
            unpack_pair(Z, g, j, i, A4, B4);
            J1[r] = diag_jacobian(d.kk[i], d.ll[j], A4, B4);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NPX; ++r) {
        const int idx = threadIdx.x + r * 1024;
        if (idx < rz) { const int y = idx / N, x = idx - y * N; Z[y * LD + x] = make_double2(u2[r] * xi2[r], v2[r] * xi2[r]); }
    }
    __syncthreads();
    fft2d_fwd_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
#pragma unroll
    for (int r = 0; r < NSP; ++r) {
        const int idx = threadIdx.x + r * 1024;
        if (idx < sz) {
            const int j = idx / NK, i = idx - j * NK;
            double2 A5, B5;
            unpack_pair(Z, g, j, i, A5, B5);
            const double2 p1 = ph0[idx], p2 = ph1[idx];
            diag_accumulate_elem<2, true>(d, c, acc, idx, i, j, so + idx, (size_t)b * sz + idx, sz, zero, zero, p1, p2, zero, zero, J1[r], zero, A5, B5,
                                    false, zero, zero, zero, zero, zero, zero, zero, zero, zero, zero);
        }
    }
    __syncthreads();
    }       // half != 2
    if (half == 1) return;
    // ---- the forcing's spectrum (held in LDS behind the field and its tables: every thread re-reads what it wrote itself),
    //      the enstrophy flux of layer 1 (held in registers), of layer 2, and everything else
    double2 *SH = reinterpret_cast<double2 *>(qgx_smem + diag_reg_side_offset(NN));
    if (S) {
#pragma unroll
        for (int r = 0; r < NPX; ++r) {
            const int idx = threadIdx.x + r * 1024;
            if (idx < rz) { const int y = idx / N, x = idx - y * N; Z[y * LD + x] = make_double2(weight * S[ro + idx], weight * S[ro + rz + idx]); }
        }
        __syncthreads();
        fft2d_fwd_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
#pragma unroll
        for (int r = 0; r < NSP; ++r) {
            const int idx = threadIdx.x + r * 1024;
            if (idx < sz) {
                const int j = idx / NK, i = idx - j * NK;
                double2 s1, s2;
                unpack_pair(Z, g, j, i, s1, s2);
                SH[idx] = s1; SH[sz + idx] = s2;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < NPX; ++r) {
        const int idx = threadIdx.x + r * 1024;
        if (idx < rz) { const int y = idx / N, x = idx - y * N; const double q1 = q[ro + idx]; Z[y * LD + x] = make_double2(u1[r] * q1, v1[r] * q1); }
    }
    __syncthreads();
    fft2d_fwd_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
    double2 G1[NSP];                                // Jacobian of (u_1 q_1, v_1 q_1), held for the enstrophy budget
#pragma unroll
    for (int r = 0; r < NSP; ++r) {
        const int idx = threadIdx.x + r * 1024;
        if (idx < sz) {
            const int j = idx / NK, i = idx - j * NK;
            double2 A6, B6;
            unpack_pair(Z, g, j, i, A6, B6);
            G1[r] = diag_jacobian(d.kk[i], d.ll[j], A6, B6);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NPX; ++r) {
        const int idx = threadIdx.x + r * 1024;
        if (idx < rz) { const int y = idx / N, x = idx - y * N; const double q2 = q[ro + rz + idx]; Z[y * LD + x] = make_double2(u2[r] * q2, v2[r] * q2); }
    }
    __syncthreads();
    fft2d_fwd_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
#pragma unroll
    for (int r = 0; r < NSP; ++r) {
        const int idx = threadIdx.x + r * 1024;
        if (idx < sz) {
            const int j = idx / NK, i = idx - j * NK;
            double2 A7, B7;
            unpack_pair(Z, g, j, i, A7, B7);
            const double2 q1 = qh0[idx], q2 = qh1[idx];
            const double2 p1 = ph0[idx], p2 = ph1[idx];
            const size_t o = so + idx;
            diag_accumulate_elem<4, true>(d, c, acc, idx, i, j, o, (size_t)b * sz + idx, sz, q1, q2, p1, p2, zero, zero, zero, zero, zero, zero,
                                          S != nullptr, S ? SH[idx] : zero, S ? SH[sz + idx] : zero, G1[r], zero, A7, B7, dq_p[o], dq_p[o + sz],
                                          dq_pp[o], dq_pp[o + sz]);
        }
    }
}

// ------------------------------------------------------------------ the same increment spread over several workgroups per member
// One workgroup per member is a chain of ten dependent 2-D transforms on ONE CU (about 110 us for a single member, and half
// of the CUs idle at 128).  The four inverse transforms depend on qh alone and the six forward transforms on their results
// alone, so they are run as two launches of (member, transform) workgroups followed by diag.hip's accumulation kernel:
// three short launches instead of one long one.  Same device functions, same expressions: bit-identical to k_diag_small.
//   which 0, 1: _invert of layer k -> ph_k, u_k, v_k;  2: p = irfft2(psi) (both layers packed);  3: xi = irfft2(-K^2 psi)
template <int NN>
__global__ void k_diag_inv_small(SpecDev d, const double2 *qh, double2 *ph, double *u, double *v, double *P, double *XI,
                                 double2 *T3, double2 *T4) {
    double2 *Z = reinterpret_cast<double2 *>(qgx_smem);
    int *pos_lds;
    Grid g = make_grid(d, Z, pos_lds);
    if (NN) { g.N = NN; g.NK = NN / 2 + 1; g.LD = NN + 1; }
    const int N = NN ? NN : d.N, NK = NN ? NN / 2 + 1 : d.NK, LD = NN ? NN + 1 : d.LD;
    const int b = blockIdx.x >> 2, which = blockIdx.x & 3;
    const size_t so = (size_t)b * 2 * N * NK, ro = (size_t)b * 2 * N * N;
    const int sz = N * NK, rz = N * N;
    __syncthreads();
    if (which < 2) {
        const int k = which;
        build_uv(Z, g, d, k, qh + so, qh + so + sz, ph + so + k * sz);
        __syncthreads();
        fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
        for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
            const int y = idx / N, x = idx - y * N;
            const double2 uv = Z[y * LD + x];
            u[ro + k * rz + idx] = uv.x;
            v[ro + k * rz + idx] = uv.y;
        }
        return;
    }
    // psi (which == 2) or -K^2 psi (which == 3) of both layers into this workgroup's own scratch, then one packed pair
    double2 *T = which == 2 ? T4 : T3;
    for (int idx = threadIdx.x; idx < 2 * sz; idx += blockDim.x) {
        const int k = idx / sz, r = idx - k * sz;
        const double2 p = invert_layer(d, k, r, qh[so + r], qh[so + sz + r]);
        if (which == 2) T[so + idx] = p;
        else {
            const double w = -d.wv2[r];
            T[so + idx] = make_double2(w * p.x, w * p.y);
        }
    }
    __syncthreads();
    build_pair(Z, g, T + so, T + so + sz, d.invN2);
    __syncthreads();
    fft2d_inv_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
    double *dst = which == 2 ? P : XI;
    for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
        const int y = idx / N, x = idx - y * N;
        const double2 w = Z[y * LD + x];
        dst[ro + idx] = w.x;
        dst[ro + rz + idx] = w.y;
    }
}

// which 0..5: the product pair (or the forcing, which == 3) of k_diag_small, transformed and unpacked into its spectrum array
template <int NN>
__global__ void k_diag_fwd_small(SpecDev d, DiagConst c, const double *u, const double *v, const double *P, const double *XI,
                                 const double *q, const double *S, double weight, double2 *S3, double2 *S4, double2 *S5, double2 *Sh,
                                 double2 *S6, double2 *S7, int nwhich) {
    double2 *Z = reinterpret_cast<double2 *>(qgx_smem);
    int *pos_lds;
    Grid g = make_grid(d, Z, pos_lds);
    if (NN) { g.N = NN; g.NK = NN / 2 + 1; g.LD = NN + 1; }
    const int N = NN ? NN : d.N, NK = NN ? NN / 2 + 1 : d.NK, LD = NN ? NN + 1 : d.LD;
    const int b = blockIdx.x / nwhich;
    int which = blockIdx.x - b * nwhich;
    if (!S && which >= 3) ++which;                 // no forcing: five pairs, the slot of the forcing is skipped
    const size_t so = (size_t)b * 2 * N * NK, ro = (size_t)b * 2 * N * N;
    const int sz = N * NK, rz = N * N;
    __syncthreads();
    for (int idx = threadIdx.x; idx < rz; idx += blockDim.x) {
        const int y = idx / N, x = idx - y * N;
        const size_t o = ro + idx;
        double ra, rb;
        if (which == 0) {
            const double u1 = u[o], u2 = u[o + rz], v1 = v[o], v2 = v[o + rz];
            const double ptpc = P[o] - P[o + rz];
            const double ub = c.del1 * u1 + c.del2 * u2, vb = c.del1 * v1 + c.del2 * v2;
            ra = ub * ptpc; rb = vb * ptpc;
        } else if (which == 1) {
            const double x1 = XI[o];
            ra = u[o] * x1; rb = v[o] * x1;
        } else if (which == 2) {
            const double x2 = XI[o + rz];
            ra = u[o + rz] * x2; rb = v[o + rz] * x2;
        } else if (which == 3) {
            ra = weight * S[o]; rb = weight * S[o + rz];
        } else if (which == 4) {
            const double q1 = q[o];
            ra = u[o] * q1; rb = v[o] * q1;
        } else {
            const double q2 = q[o + rz];
            ra = u[o + rz] * q2; rb = v[o + rz] * q2;
        }
        Z[y * LD + x] = make_double2(ra, rb);
    }
    __syncthreads();
    fft2d_fwd_x<NN>(Z, N, LD, g.nrad, g.rad, g.tw);
    double2 *dst = which == 0 ? S3 : (which == 1 ? S4 : (which == 2 ? S5 : (which == 3 ? Sh : (which == 4 ? S6 : S7))));
    for (int idx = threadIdx.x; idx < sz; idx += blockDim.x) {
        const int j = idx / NK, i = idx - j * NK;
        double2 s0, s1;
        unpack_pair(Z, g, j, i, s0, s1);
        dst[so + idx] = s0;
        dst[so + sz + idx] = s1;
    }
}

// ------------------------------------------------------------------ host launchers
static size_t small_lds_bytes(const SpecDev &d) {
    size_t bytes = (size_t)d.N * d.LD * sizeof(double2) + (size_t)((d.N + 3) & ~3) * sizeof(int) + (size_t)d.N * sizeof(double2);
    return (bytes + 15) & ~(size_t)15;
}
static int small_threads(const SpecDev &d, const ModelOpts &o) {
    if (o.spec_threads > 0) return o.spec_threads;
    // One workgroup advances one member.  With at most one member per CU the kernel is a latency chain
    // (about 50 barriers): 1024 threads shorten it (B=128, N=64: 119 -> 71 us); with several members
    // queued per CU 512 threads give the best throughput (B=1024: 293 us vs 385 @256 / 337 @1024).
    // Grids above 64x64 leave room for one workgroup per CU only: always 1024 threads there.
    return (d.B <= 256 || d.N > 64) ? 1024 : 512;
}

bool small_path_fits(int N) {
    // field + digit-reversal table + twiddle table (small_lds_bytes) + the static reduction scratch of k_step_small
    return (size_t)N * (N + 1) * 16 + (size_t)((N + 3) & ~3) * 4 + (size_t)N * 16 + 16 + 160 <= 160 * 1024;
}

// kernels specialised for the grid sizes the reference runs (compile-time index arithmetic and FFT
// plan), generic run-time-N kernels for every other 2^a 3^b size
#define QGX_DISPATCH_N(N_, CALL)                \
    switch (N_) {                               \
        case 32: { constexpr int NN = 32; CALL; } break; \
        case 48: { constexpr int NN = 48; CALL; } break; \
        case 64: { constexpr int NN = 64; CALL; } break; \
        case 96: { constexpr int NN = 96; CALL; } break; \
        default: { constexpr int NN = 0; CALL; } break;  \
    }

int small_prepare(const SpecDev &d) {
    const int bytes = (int)small_lds_bytes(d);
    QGX_DISPATCH_N(d.N, {
        QGX_HIP(hipFuncSetAttribute((const void *)k_step_small<NN>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        QGX_HIP(hipFuncSetAttribute((const void *)(k_step_small<NN, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        QGX_HIP(hipFuncSetAttribute((const void *)(k_step_small<NN, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        QGX_HIP(hipFuncSetAttribute((const void *)(k_step_small<NN, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        QGX_HIP(hipFuncSetAttribute((const void *)(k_step_small<NN, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        QGX_HIP(hipFuncSetAttribute((const void *)k_q_to_qh_small<NN>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        QGX_HIP(hipFuncSetAttribute((const void *)k_qh_to_q_small<NN>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        QGX_HIP(hipFuncSetAttribute((const void *)k_invert_small<NN>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        QGX_HIP(hipFuncSetAttribute((const void *)k_diag_small<NN>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        if constexpr (NN == 32 || NN == 48 || NN == 64)
        {
            QGX_HIP(hipFuncSetAttribute((const void *)k_diag_small_reg<NN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_reg_lds_bytes(NN)));
            QGX_HIP(hipFuncSetAttribute((const void *)(k_diag_small_reg<NN, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_reg_lds_bytes(NN)));
        }
        QGX_HIP(hipFuncSetAttribute((const void *)k_diag_inv_small<NN>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        QGX_HIP(hipFuncSetAttribute((const void *)k_diag_fwd_small<NN>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    })
    return QGX_OK;
}

// two workgroups per member (one per layer) while all of them are resident at once
bool small_layer_split(const SpecDev &d, const ModelOpts &o) {
    if (o.lsplit >= 0) return o.lsplit != 0;
    return 2 * d.B <= 256;      // one workgroup per CU (measured at 64x64: B = 128 44 -> 33 us, B = 256 55 -> 63 us)
}

// part: 0 the whole step; 1 / 2 its two halves; 3 both in one launch of 4 B workgroups (layer-split form only; k_step_small PART)
int small_step(const SpecDev &d, const ModelOpts &o, const StepArgs &a, hipStream_t st, int part) {
    QGX_REQUIRE(part == 0 || small_layer_split(d, o), "small_step: the two-kernel step needs the layer-split form");
    if (part == 1) {
        QGX_DISPATCH_N(d.N, hipLaunchKernelGGL((k_step_small<NN, true, 1>), dim3(2 * d.B), dim3(1024), small_lds_bytes(d), st, d, a))
    } else if (part == 2) {
        QGX_DISPATCH_N(d.N, hipLaunchKernelGGL((k_step_small<NN, true, 2>), dim3(2 * d.B), dim3(1024), small_lds_bytes(d), st, d, a))
    } else if (part == 3) {
        QGX_REQUIRE(a.sib_flag && a.sib_epoch, "small_step: the four-workgroup step needs its flag words");
        QGX_DISPATCH_N(d.N, hipLaunchKernelGGL((k_step_small<NN, true, 3>), dim3(((2 * d.B + 7) & ~7) + 2 * d.B), dim3(1024), small_lds_bytes(d), st, d, a))
    } else if (small_layer_split(d, o)) {
        QGX_DISPATCH_N(d.N, hipLaunchKernelGGL((k_step_small<NN, true>), dim3(2 * d.B), dim3(1024), small_lds_bytes(d), st, d, a))
    } else {
        QGX_DISPATCH_N(d.N, hipLaunchKernelGGL(k_step_small<NN>, dim3(d.B), dim3(small_threads(d, o)), small_lds_bytes(d), st, d, a))
    }
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}
int small_q_to_qh(const SpecDev &d, const ModelOpts &o, const double *q, double2 *qh, hipStream_t st) {
    QGX_DISPATCH_N(d.N, hipLaunchKernelGGL(k_q_to_qh_small<NN>, dim3(d.B), dim3(small_threads(d, o)), small_lds_bytes(d), st, d, q, qh))
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}
int small_qh_to_q(const SpecDev &d, const ModelOpts &o, const double2 *qh, double *q, hipStream_t st) {
    QGX_DISPATCH_N(d.N, hipLaunchKernelGGL(k_qh_to_q_small<NN>, dim3(d.B), dim3(small_threads(d, o)), small_lds_bytes(d), st, d, qh, q))
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}
int small_invert(const SpecDev &d, const ModelOpts &o, const double2 *qh, double2 *ph, double *u, double *v, hipStream_t st) {
    QGX_DISPATCH_N(d.N, hipLaunchKernelGGL(k_invert_small<NN>, dim3(d.B), dim3(small_threads(d, o)), small_lds_bytes(d), st, d, qh, ph, u, v))
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}
// the one-workgroup-per-member increment with its work fields in registers (k_diag_small_reg): grids up to 64 x 64; stores no
// ph, u, v.  -> false: no such kernel for this grid
bool small_diag_increment_reg_ok(const SpecDev &d) { return d.N == 32 || d.N == 48 || d.N == 64; }
int small_diag_increment_reg(const SpecDev &d, const DiagConst &c, const double2 *qh, double2 *ph, const double *S, double weight, const double *q,
                             const double2 *dq_p, const double2 *dq_pp, const DiagAcc &a, hipStream_t st, bool halves) {
    if (halves) {      // two workgroups per member (k_diag_small_reg HALF)
        switch (d.N) {
            case 32: hipLaunchKernelGGL((k_diag_small_reg<32, 1>), dim3(2 * d.B), dim3(1024), diag_reg_lds_bytes(32), st, d, c, qh, ph, S, weight, q, dq_p, dq_pp, a); break;
            case 48: hipLaunchKernelGGL((k_diag_small_reg<48, 1>), dim3(2 * d.B), dim3(1024), diag_reg_lds_bytes(48), st, d, c, qh, ph, S, weight, q, dq_p, dq_pp, a); break;
            case 64: hipLaunchKernelGGL((k_diag_small_reg<64, 1>), dim3(2 * d.B), dim3(1024), diag_reg_lds_bytes(64), st, d, c, qh, ph, S, weight, q, dq_p, dq_pp, a); break;
            default: QGX_REQUIRE(false, "small_diag_increment_reg: no kernel for N = %d", d.N);
        }
        QGX_HIP(hipGetLastError());
        return QGX_OK;
    }
    switch (d.N) {
        case 32: hipLaunchKernelGGL(k_diag_small_reg<32>, dim3(d.B), dim3(1024), diag_reg_lds_bytes(32), st, d, c, qh, ph, S, weight, q, dq_p, dq_pp, a); break;
        case 48: hipLaunchKernelGGL(k_diag_small_reg<48>, dim3(d.B), dim3(1024), diag_reg_lds_bytes(48), st, d, c, qh, ph, S, weight, q, dq_p, dq_pp, a); break;
        case 64: hipLaunchKernelGGL(k_diag_small_reg<64>, dim3(d.B), dim3(1024), diag_reg_lds_bytes(64), st, d, c, qh, ph, S, weight, q, dq_p, dq_pp, a); break;
        default: QGX_REQUIRE(false, "small_diag_increment_reg: no kernel for N = %d", d.N);
    }
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}
int small_diag_increment(const SpecDev &d, const DiagConst &c, const double2 *qh, double2 *ph, double *u, double *v, double *P,
                         double *XI, double2 *S3, double2 *S4, double2 *S5, double2 *Sh, double2 *S6, double2 *S7, const double *S,
                         double weight, const double *q, const double2 *dq_p, const double2 *dq_pp, const DiagAcc &a, hipStream_t st) {
    QGX_DISPATCH_N(d.N, hipLaunchKernelGGL(k_diag_small<NN>, dim3(d.B), dim3(1024), small_lds_bytes(d), st, d, c, qh, ph, u, v, P, XI,
                                           S3, S4, S5, Sh, S6, S7, S, weight, q, dq_p, dq_pp, a))
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}
// the transforms of one increment as (member, transform) workgroups: inverse set, then forward set (k_diag_accumulate follows)
int small_diag_transforms_wide(const SpecDev &d, const DiagConst &c, const double2 *qh, double2 *ph, double *u, double *v, double *P,
                               double *XI, double2 *S3, double2 *S4, double2 *S5, double2 *Sh, double2 *S6, double2 *S7, const double *S,
                               double weight, const double *q, hipStream_t st) {
    const int nwhich = S ? 6 : 5;
    QGX_DISPATCH_N(d.N, {
        hipLaunchKernelGGL(k_diag_inv_small<NN>, dim3(4 * d.B), dim3(1024), small_lds_bytes(d), st, d, qh, ph, u, v, P, XI, S3, S4);
        hipLaunchKernelGGL(k_diag_fwd_small<NN>, dim3(nwhich * d.B), dim3(1024), small_lds_bytes(d), st, d, c, (const double *)u,
                           (const double *)v, (const double *)P, (const double *)XI, q, S, weight, S3, S4, S5, Sh, S6, S7, nwhich);
    })
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}
#undef QGX_DISPATCH_N

}  // namespace qgx
