// Developer tool (not part of the product library): price of keeping a 256 x 256 member's row <-> column exchange on
// chip.  One workgroup per CU; the workgroups that find themselves on the same XCD (HW_REG_XCC_ID, read at run time)
// form a team, and a team walks its members through the three phases of the large-grid step
//     A  write 3 fields by rows      | barrier |  B  read 3 fields by columns, write 2 back  | barrier |
//     C  read 2 fields by rows       | barrier |
// in ONE 3 MB exchange buffer per team, with data tags that are checked (stale or torn reads are counted).
// MODE 0: plain stores, agent-scope release fence before the arrive, acquire fence after the wait (documented protocol)
// MODE 1: sc1 (write-through) stores, vmcnt(0), arrive; acquire fence after the wait             (documented protocol)
// MODE 2: plain stores, vmcnt(0), arrive; NO fences; consumers load with sc1 (L1 bypass) -- relies on the team sharing
//         one L2, which HW_REG_XCC_ID establishes
// MODE 3: plain stores, vmcnt(0), arrive; acquire fence after the wait (L1 invalidate), plain loads
// MODE 4: MODE 2 with TWO members in flight per team (two exchange buffers = 6.4 MB live per XCD, beyond its 4 MB L2):
//         arrive for one member, work on the other, then wait — the barrier latency hides behind the other member's phase.
//         Measured: 18.5 us per member against 12.2 us with one in flight — two live buffers overflow the L2 and cost more
//         than the hidden barriers give.  (Timing probe only: its tag check is not reliable — the hand-written asm loads
//         leave their destination registers unprotected until the s_waitcnt, which is why the product kernel issues its
//         sc1 loads through the compiler's buffer-load builtin.)
//   hipcc --offload-arch=gfx950 -O3 -o bench_tools/_build/xcd_exchange bench_tools/xcd_exchange.hip && bench_tools/_build/xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N = 256;
constexpr int NT = 1024;
constexpr size_t FIELD = (size_t)N * N;            // double2 per field

struct Ctl {
    unsigned arrived, err, bad, pad[29];
    unsigned team_n[8][32];                          // [x][0]: workgroups registered on XCD x
    unsigned bar[8][32];                             // [x][0]: monotonic arrive counter of team x
    unsigned long long t_total[8][4];
};

__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ inline void store16(double2 *p, double2 v, bool sc1) {
    if (sc1) {
        v2d w = {v.x, v.y};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
    } else *p = v;
}

template <int K>
__device__ inline void load16_sc1(double2 (&v)[K], const double2 *const (&p)[K]) {
    v2d w[K];
#pragma unroll
    for (int i = 0; i < K; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(w[i]) : "v"(p[i]) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < K; ++i) v[i] = make_double2(w[i].x, w[i].y);
}

template <int MODE>
__device__ inline bool team_barrier(unsigned *ctr, unsigned target, Ctl *c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int good = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 21)) { atomicExch(&c->err, 1u); good = 0; break; }
        }
        if (MODE != 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

__device__ inline double2 tag(int m, int f, int row, int col, int gen) {
    return make_double2((double)(((m * 4 + f) * N + row) * N + col), (double)gen);
}

template <int MODE>
__global__ __launch_bounds__(NT) void k_exchange(double2 *X, const double2 *stream_in, double2 *stream_out, Ctl *c,
                                                  int members, int stream_lines) {
    extern __shared__ double2 lds[];
    __shared__ unsigned s_x, s_rank, s_size;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned x = xcc_id() & 7u;
        s_x = x;
        s_rank = atomicAdd(&c->team_n[x][0], 1u);
        __hip_atomic_fetch_add(&c->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&c->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 21)) { atomicExch(&c->err, 2u); break; }
        }
        s_size = __hip_atomic_load(&c->team_n[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned x = s_x, rank = s_rank, size = s_size;
    if (c->err) return;
    double2 *Xt = X + (size_t)x * 3 * FIELD;
    unsigned *ctr = &c->bar[x][0];
    unsigned phase = 0, bad = 0;
    const bool sc1_st = MODE == 1;
    long long t0 = wall_clock64();
    for (int it = 0; it < members; ++it) {
        const int m = it * 8 + x;
        // optional streaming traffic beside the exchange (stands for qh / dqhdt history in, qh / dqhdt out)
        if (stream_lines > 0) {
            const size_t base = ((size_t)m * size + rank) * stream_lines * 256;
            double2 acc = make_double2(0, 0);
            for (int i = tid; i < stream_lines * 256; i += NT) {
                v2d v = __builtin_nontemporal_load((const v2d *)&stream_in[base + i]);
                acc.x += v.x; acc.y += v.y;
                if (i < stream_lines * 256 * 2 / 3) __builtin_nontemporal_store(v, (v2d *)&stream_out[base + i]);
            }
            if (acc.x == 12345.678) lds[tid] = acc;
        }
        // A: rows
        for (int row = rank; row < N; row += size)
            for (int i = tid; i < 3 * N; i += NT) {
                int f = i >> 8, col = i & 255;
                store16(&Xt[f * FIELD + (size_t)row * N + col], tag(m, f, row, col, 0), sc1_st);
            }
        if (!team_barrier<MODE>(ctr, ++phase * size, c)) return;
        // B: columns, 8 at a time (128-byte segments of a row), all 3 fields; write fields 0 and 1 back
        for (int g = rank; g < N / 8; g += size) {
            // 256 rows x 3 fields x 8 columns = 6144 values, 6 per thread
            double2 v[6];
            const double2 *p[6];
            int rr[6], ff[6], cc[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                int i = tid + j * NT;
                int col = g * 8 + (i & 7), row = (i >> 3) & 255, f = i >> 11;
                rr[j] = row; ff[j] = f; cc[j] = col;
                p[j] = &Xt[f * FIELD + (size_t)row * N + col];
            }
            if (MODE == 2) load16_sc1<6>(v, p);
            else {
#pragma unroll
                for (int j = 0; j < 6; ++j) v[j] = *p[j];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double2 e = tag(m, ff[j], rr[j], cc[j], 0);
                bad += (v[j].x != e.x || v[j].y != e.y);
                if (ff[j] < 2) store16(const_cast<double2 *>(p[j]), tag(m, ff[j], rr[j], cc[j], 1), sc1_st);
            }
        }
        if (!team_barrier<MODE>(ctr, ++phase * size, c)) return;
        // C: rows of fields 0 and 1
        for (int row = rank; row < N; row += size) {
            if (tid < 2 * N) {
                int f = tid >> 8, col = tid & 255;
                double2 v[1];
                const double2 *p[1] = {&Xt[f * FIELD + (size_t)row * N + col]};
                if (MODE == 2) load16_sc1<1>(v, p); else v[0] = *p[0];
                double2 e = tag(m, f, row, col, 1);
                bad += (v[0].x != e.x || v[0].y != e.y);
            }
        }
        if (!team_barrier<MODE>(ctr, ++phase * size, c)) return;
    }
    long long t1 = wall_clock64();
    if (bad) atomicAdd(&c->bad, bad);
    if (tid == 0 && rank == 0) { c->t_total[x][0] = (unsigned long long)(t1 - t0); c->t_total[x][1] = size; }
}


__device__ inline void arrive(unsigned *ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline bool wait_for(unsigned *ctr, unsigned target, Ctl *c) {
    __shared__ int ok2;
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        int good = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 21)) { atomicExch(&c->err, 1u); good = 0; break; }
        }
        ok2 = good;
    }
    __syncthreads();
    return ok2 != 0;
}

__global__ __launch_bounds__(NT) void k_exchange2(double2 *X, Ctl *c, int members, int slots) {
    extern __shared__ double2 lds[];
    __shared__ unsigned s_x, s_rank, s_size;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned x = xcc_id() & 7u;
        s_x = x;
        s_rank = atomicAdd(&c->team_n[x][0], 1u);
        __hip_atomic_fetch_add(&c->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&c->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 21)) { atomicExch(&c->err, 2u); break; }
        }
        s_size = __hip_atomic_load(&c->team_n[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned x = s_x, rank = s_rank, size = s_size;
    if (c->err) return;
    unsigned bad = 0;
    unsigned ph[2] = {0, 0};
    long long t0 = wall_clock64();
    auto phaseA = [&](int sl, int m) {
        double2 *Xt = X + ((size_t)sl * 8 + x) * 3 * FIELD;
        for (int row = rank; row < N; row += size)
            for (int i = tid; i < 3 * N; i += NT) {
                int f = i >> 8, col = i & 255;
                Xt[f * FIELD + (size_t)row * N + col] = tag(m, f, row, col, 0);
            }
    };
    auto phaseB = [&](int sl, int m) {
        double2 *Xt = X + ((size_t)sl * 8 + x) * 3 * FIELD;
        for (int g = rank; g < N / 8; g += size) {
            double2 v[6];
            const double2 *p[6];
            int rr[6], ff[6], cc[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                int i = tid + j * NT;
                int col = g * 8 + (i & 7), row = (i >> 3) & 255, f = i >> 11;
                rr[j] = row; ff[j] = f; cc[j] = col;
                p[j] = &Xt[f * FIELD + (size_t)row * N + col];
            }
            load16_sc1<6>(v, p);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double2 e = tag(m, ff[j], rr[j], cc[j], 0);
                bad += (v[j].x != e.x || v[j].y != e.y);
                if (ff[j] < 2) *const_cast<double2 *>(p[j]) = tag(m, ff[j], rr[j], cc[j], 1);
            }
        }
    };
    auto phaseC = [&](int sl, int m) {
        double2 *Xt = X + ((size_t)sl * 8 + x) * 3 * FIELD;
        for (int r0 = rank; r0 < N; r0 += 4 * size) {
            double2 v[4];
            const double2 *p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int row = r0 + j * size, f = tid >> 8, col = tid & 255;
                p[j] = &Xt[(tid < 2 * N ? f : 0) * FIELD + (size_t)(row < N ? row : 0) * N + col];
            }
            load16_sc1<4>(v, p);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int row = r0 + j * size, f = tid >> 8, col = tid & 255;
                if (tid < 2 * N && row < N) { double2 e = tag(m, f, row, col, 1); bad += (v[j].x != e.x || v[j].y != e.y); }
            }
        }
    };
    for (int it = 0; it < members; it += slots) {
        const int m0 = it * 8 + x, m1 = (it + 1) * 8 + x;
        unsigned *c0 = &c->bar[x][0], *c1 = &c->bar[x][16];
        if (slots == 2) {
            phaseA(0, m0); arrive(c0); ++ph[0];
            phaseA(1, m1); arrive(c1); ++ph[1];
            if (!wait_for(c0, ph[0] * size, c)) return;
            phaseB(0, m0); arrive(c0); ++ph[0];
            if (!wait_for(c1, ph[1] * size, c)) return;
            phaseB(1, m1); arrive(c1); ++ph[1];
            if (!wait_for(c0, ph[0] * size, c)) return;
            phaseC(0, m0); arrive(c0); ++ph[0];
            if (!wait_for(c1, ph[1] * size, c)) return;
            phaseC(1, m1); arrive(c1); ++ph[1];
            if (!wait_for(c0, ph[0] * size, c)) return;
            if (!wait_for(c1, ph[1] * size, c)) return;
        } else {
            phaseA(0, m0); arrive(c0); ++ph[0];
            if (!wait_for(c0, ph[0] * size, c)) return;
            phaseB(0, m0); arrive(c0); ++ph[0];
            if (!wait_for(c0, ph[0] * size, c)) return;
            phaseC(0, m0); arrive(c0); ++ph[0];
            if (!wait_for(c0, ph[0] * size, c)) return;
        }
    }
    long long t1 = wall_clock64();
    if (bad) atomicAdd(&c->bad, bad);
    if (tid == 0 && rank == 0) { c->t_total[x][0] = (unsigned long long)(t1 - t0); c->t_total[x][1] = size; }
}

static void run2(int members, int slots, double2 *X, Ctl *c, int ncu) {
    CK(hipMemset(c, 0, sizeof(Ctl)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    size_t lds = 96 * 1024;
    CK(hipFuncSetAttribute((const void *)k_exchange2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_exchange2, dim3(ncu), dim3(NT), lds, 0, X, c, members, slots);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    Ctl h;
    CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("mode 4 members/xcd %3d, %d in flight: %8.1f us total, %6.2f us per member per XCD  err %u bad %u\n", members, slots,
           ms * 1e3, ms * 1e3 / members, h.err, h.bad);
    fflush(stdout);
}

template <int MODE>
static void run(int members, int stream_lines, double2 *X, double2 *sin, double2 *sout, Ctl *c, int ncu) {
    CK(hipMemset(c, 0, sizeof(Ctl)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    size_t lds = 96 * 1024;
    CK(hipFuncSetAttribute((const void *)k_exchange<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_exchange<MODE>, dim3(ncu), dim3(NT), lds, 0, X, sin, sout, c, members, stream_lines);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    Ctl h;
    CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("mode %d members/xcd %3d stream_lines %3d: %8.1f us total, %6.2f us per member per XCD  err %u bad %u  teams",
           MODE, members, stream_lines, ms * 1e3, ms * 1e3 / members, h.err, h.bad);
    for (int x = 0; x < 8; ++x) printf(" %llu(%.1fus)", h.t_total[x][1], h.t_total[x][0] / 100.0);
    printf("\n");
    fflush(stdout);
}

int main(int argc, char **argv) {
    int members = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.name, ncu);
    double2 *X, *sin, *sout;
    Ctl *c;
    CK(hipMalloc(&X, 16 * 3 * FIELD * sizeof(double2)));
    const int max_lines = 40;                          // lines of 256 double2 per workgroup and member
    size_t stream = (size_t)members * 8 * 40 * max_lines * 256;
    CK(hipMalloc(&sin, stream * sizeof(double2)));
    CK(hipMalloc(&sout, stream * sizeof(double2)));
    CK(hipMemset(sin, 0, stream * sizeof(double2)));
    CK(hipMalloc(&c, sizeof(Ctl)));
    for (int rep = 0; rep < 3; ++rep) { run2(members, 1, X, c, ncu); run2(members, 2, X, c, ncu); }
    for (int rep = 0; rep < 1; ++rep) {
        for (int sl : {0}) {
            run<0>(members, sl, X, sin, sout, c, ncu);
            run<1>(members, sl, X, sin, sout, c, ncu);
            run<2>(members, sl, X, sin, sout, c, ncu);
            run<3>(members, sl, X, sin, sout, c, ncu);
        }
    }
    return 0;
}
