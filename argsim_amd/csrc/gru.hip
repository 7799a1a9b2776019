// gru.hip -- the recurrent part of cuDNN-style (reset-after) GRU layers, forward and BPTT,
// as persistent kernels for gfx950.
//
// Replaces tf.contrib.cudnn_rnn.CudnnGRU (reference src/model.py:15,120-121,160):
//   r = s(gi_r + R_r h + bR_r)   u = s(gi_u + R_u h + bR_u)
//   n = tanh(gi_n + r * (R_n h + bR_n))          h' = (1-u) n + u h
// gi = W x + bW is hoisted out of the time loop (one MFMA GEMM, gemm_f32.hip).
//
// Decomposition.  One workgroup (4 waves) owns 16 hidden units (x3 gates) of one job
// (= one layer-direction) for one batch group.  Its 48xD slice of R lives in REGISTERS for
// the whole launch, already in v_mfma_f32_16x16x4_f32 B-operand order (3*D/16 VGPRs per lane);
// the four waves split K = D, partial sums meet in LDS, then 256 threads do the gate math.
// Per step a workgroup reads only its group's h_{t-1} rows (A operand) and publishes its
// 16-column slice of h_t.  Workgroups that share a batch group synchronise through one
// monotonic counter per (job, group) with agent-scope release/acquire -- never a grid
// barrier.  blockIdx % G selects the group, so under round-robin dispatch a group's
// workgroups share an XCD (speed only; correctness is placement independent).
// Every spin is bounded; on time-out the error word is set and all waits fall through.
#include "kernels.h"

namespace avae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int pos_map(int p, int len, int reverse) { return (reverse && p < len) ? (len - 1 - p) : p; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// k offset (inside one wave's contiguous K range) of MFMA step ks for lane quarter kh
template <int NKS>
__device__ __forceinline__ int kperm(int ks, int kh)
{
    if (NKS % 4 == 0) return 16 * (ks >> 2) + 4 * kh + (ks & 3);
    return 4 * ks + kh;
}

// Hand-off protocol (MI355X_MICROARCH.md, "Valid forms", first table row): every exchanged byte is
// stored write-through (sc1) and loaded with sc1 loads (L1 bypass); each storing wave drains its
// stores (vmcnt(0)), the workgroup meets at a barrier, ONE lane adds to the group counter with an
// agent-scope atomic; the consumer polls that counter with a relaxed agent-scope (sc1) load from one
// lane and releases its workgroup through a barrier.  No cache-wide fence is involved: an agent
// release would write back every dirty L2 line of the XCD (the saved gates) on every step.
__device__ __forceinline__ void group_wait(unsigned* ctr, unsigned target, int* err)
{
    if (threadIdx.x == 0) {
        unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ULL) {   // 2 s
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler ordering only
    __syncthreads();
}

__device__ __forceinline__ void group_publish(unsigned* ctr)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// 16-byte sc1 (agent-coherent, L1-bypassing) load through a raw buffer descriptor
__device__ __forceinline__ float4 load16_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off)
{
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float load4_sc1(const float* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store4_sc1(float* p, float v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, -1, 0x00020000);
}

// ------------------------------------------------------------------------------ forward
template <int KS>   // D = 16*KS
__global__ __launch_bounds__(256, 2) void gru_fwd_kernel(GruArgs a)
{
    constexpr int D = 16 * KS, HT = KS;
    __shared__ __attribute__((aligned(16))) float part[2][2][4][3][256];   // [buf][chunk][wave][gate][lane*4+reg]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kh = lane >> 4;
    const int per_job = a.G * HT;
    const int jb = blockIdx.x / per_job, rem = blockIdx.x - jb * per_job;
    const int g = rem % a.G, ht = rem / a.G;
    const GruJob& J = a.job[jb];
    unsigned* ctr = a.counters + jb * a.G + g;
    const int B = a.B;
    const int row_beg = g * a.rows_per_group;
    const int row_end = min(B, row_beg + a.rows_per_group);

    // weights -> registers, MFMA B-operand order: B[k][n] = R'[ht*48 + gate*16 + n][k]
    float w[3][KS];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) {
        const float* rp = J.R + (size_t)(ht * 48 + gate * 16 + n) * D + wave * 4 * KS;
        if (KS % 4 == 0) {
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                float4 v = *reinterpret_cast<const float4*>(rp + 16 * q + 4 * kh);
                w[gate][4 * q + 0] = v.x; w[gate][4 * q + 1] = v.y; w[gate][4 * q + 2] = v.z; w[gate][4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w[gate][ks] = rp[kperm<KS>(ks, kh)];
        }
    }
    const int gn = tid & 15, gr = tid >> 4;        // gate-phase item: unit gn, row gr (+16 per chunk)
    float bR[3];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) bR[gate] = J.bR[ht * 48 + gate * 16 + gn];

    const __amdgpu_buffer_rsrc_t rs_hs = make_rsrc(J.hs), rs_h0 = make_rsrc(J.h0 ? J.h0 : J.hs);
    int buf = 0;
    for (int p = a.p_begin; p < a.p_end; ++p) {
        if (p > a.p_begin) group_wait(ctr, (unsigned)(HT * (p - a.p_begin)), a.err);

        for (int rb = row_beg; rb < row_end; rb += 32, buf ^= 1) {
            // ---- MFMA phase: partial gh = h_{p-1}[rows, wave's K range] x R'^T
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (rb + 16 * c >= row_end) break;           // wave uniform
                const int row = min(rb + 16 * c + n, B - 1);   // clamped: rows >= B are never stored
                unsigned aoff = 0; bool have = true;
                if (p == 0) { have = J.h0 != nullptr; aoff = (unsigned)((size_t)row * D * 4); }
                else {
                    int pp = pos_map(p - 1, J.reverse ? a.lens[row] : 0, J.reverse);
                    aoff = (unsigned)((((size_t)pp * B + row) * a.ldh) * 4);
                }
                const __amdgpu_buffer_rsrc_t rs = (p == 0) ? rs_h0 : rs_hs;
                float av[KS];
                if (KS % 4 == 0) {
#pragma unroll
                    for (int q = 0; q < KS / 4; ++q) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (have) v = load16_sc1(rs, aoff + (wave * 4 * KS + 16 * q + 4 * kh) * 4);
                        av[4 * q + 0] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
                    }
                } else {
                    const float* hsrc = (p == 0) ? J.h0 : J.hs;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        av[ks] = have ? load4_sc1(hsrc + aoff / 4 + wave * 4 * KS + kperm<KS>(ks, kh)) : 0.f;
                }
                f32x4 acc[3];
#pragma unroll
                for (int gate = 0; gate < 3; ++gate) acc[gate] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int gate = 0; gate < 3; ++gate)
                        acc[gate] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], w[gate][ks], acc[gate], 0, 0, 0);
#pragma unroll
                for (int gate = 0; gate < 3; ++gate)
                    *reinterpret_cast<f32x4*>(&part[buf][c][wave][gate][lane * 4]) = acc[gate];
            }
            __syncthreads();
            // ---- gate phase: 32 rows x 16 units, two items per thread
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int row = rb + 16 * c + gr;
                if (row >= row_end) continue;
                const int len = J.reverse ? a.lens[row] : 0;
                const int pos = pos_map(p, len, J.reverse);
                const int j = ht * 16 + gn;
                const int pidx = ((gr >> 2) * 16 + gn) * 4 + (gr & 3);
                float gh[3];
#pragma unroll
                for (int gate = 0; gate < 3; ++gate)
                    gh[gate] = bR[gate] + ((part[buf][c][0][gate][pidx] + part[buf][c][1][gate][pidx]) +
                                           (part[buf][c][2][gate][pidx] + part[buf][c][3][gate][pidx]));
                const float* gi = J.gi + ((size_t)pos * B + row) * a.ldg + ht * 48 + gn;
                float hprev = 0.f;
                if (p == 0) { if (J.h0) hprev = J.h0[(size_t)row * D + j]; }
                else {
                    int pp = pos_map(p - 1, len, J.reverse);
                    hprev = load4_sc1(J.hs + ((size_t)pp * B + row) * a.ldh + j);
                }
                float r = sigmoidf_(gi[0] + gh[0]);
                float u = sigmoidf_(gi[16] + gh[1]);
                float nn = tanhf(gi[32] + r * gh[2]);
                float hnew = (1.f - u) * nn + u * hprev;
                store4_sc1(J.hs + ((size_t)pos * B + row) * a.ldh + j, hnew);
                if (J.sv) {
                    float* sv = J.sv + (((size_t)pos * B + row) * HT + ht) * 64 + gn;
                    sv[0] = r; sv[16] = u; sv[32] = nn; sv[48] = gh[2];
                }
                if (J.hp) J.hp[((size_t)pos * B + row) * D + j] = hprev;
            }
        }
        if (p + 1 < a.p_end) group_publish(ctr);
    }
}

// ------------------------------------------------------------------------------ backward
// step p (descending):  dH_p = dh_out[p] + dH_{p+1} u_{p+1} + dgh_{p+1} R'
//   dn = dH (1-u)(1-n^2)   du = dH (h_{p-1} - n) u (1-u)   dr = dn hn r (1-r)
//   dgi = [dr,du,dn]   dgh = [dr,du,dn r]
template <int KS>
__global__ __launch_bounds__(256, 2) void gru_bwd_kernel(GruArgs a)
{
    constexpr int D = 16 * KS, HT = KS, NKS = 3 * KS;   // wave K range = 3D/4 = 12*KS floats
    __shared__ __attribute__((aligned(16))) float part[2][2][4][256];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kh = lane >> 4;
    const int per_job = a.G * HT;
    const int jb = blockIdx.x / per_job, rem = blockIdx.x - jb * per_job;
    const int g = rem % a.G, ht = rem / a.G;
    const GruJob& J = a.job[jb];
    unsigned* ctr = a.counters + jb * a.G + g;
    const int B = a.B, S = a.S;
    const int row_beg = g * a.rows_per_group;
    const int row_end = min(B, row_beg + a.rows_per_group);

    // B operand: B[k = c'][n] = R'[c'][ht*16 + n]
    float w[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
        w[ks] = J.R[(size_t)(wave * 12 * KS + kperm<NKS>(ks, kh)) * D + ht * 16 + n];

    const int gn = tid & 15, gr = tid >> 4;
    const int j = ht * 16 + gn;
    const bool want_dh0 = (J.dh0 != nullptr) && a.p_begin == 0;
    const int p_last = want_dh0 ? -1 : a.p_begin;       // p == -1: only dh0 = carry + dgh_0 R'
    const __amdgpu_buffer_rsrc_t rs_dgh = make_rsrc(J.dgh);
    int buf = 0, done = 0;
    for (int p = a.p_end - 1; p >= p_last; --p, ++done) {
        if (done > 0) group_wait(ctr, (unsigned)(HT * done), a.err);
        const bool have_next = (p + 1 < S);

        for (int rb = row_beg; rb < row_end; rb += 32, buf ^= 1) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (rb + 16 * c >= row_end) break;
                f32x4 acc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (have_next) {
                    const int row = min(rb + 16 * c + n, B - 1);
                    const int pos1 = pos_map(p + 1, J.reverse ? a.lens[row] : 0, J.reverse);
                    const unsigned aoff = (unsigned)((((size_t)pos1 * B + row) * a.ldg + wave * 12 * KS) * 4);
                    float av[NKS];
                    if (NKS % 4 == 0) {
#pragma unroll
                        for (int q = 0; q < NKS / 4; ++q) {
                            float4 v = load16_sc1(rs_dgh, aoff + (16 * q + 4 * kh) * 4);
                            av[4 * q + 0] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int ks = 0; ks < NKS; ++ks) av[ks] = load4_sc1(J.dgh + aoff / 4 + kperm<NKS>(ks, kh));
                    }
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks)
                        acc[ks & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], w[ks], acc[ks & 3], 0, 0, 0);
                }
                f32x4 s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                *reinterpret_cast<f32x4*>(&part[buf][c][wave][lane * 4]) = s;
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int row = rb + 16 * c + gr;
                if (row >= row_end) continue;
                const int pidx = ((gr >> 2) * 16 + gn) * 4 + (gr & 3);
                float carried = (part[buf][c][0][pidx] + part[buf][c][1][pidx]) + (part[buf][c][2][pidx] + part[buf][c][3][pidx]);
                float* carryp = J.carry + (size_t)row * D + j;
                if (have_next) carried += *carryp;
                if (p < 0) { J.dh0[(size_t)row * D + j] = carried; continue; }
                const int len = J.reverse ? a.lens[row] : 0;
                const int pos = pos_map(p, len, J.reverse);
                const size_t rix = (size_t)pos * B + row;
                float dH = carried + (J.dh_out ? J.dh_out[rix * a.ldh + j] : 0.f);
                const float* sv = J.sv + (rix * HT + ht) * 64 + gn;
                float r = sv[0], u = sv[16], nn = sv[32], hn = sv[48];
                float hprev = J.hp[rix * D + j];
                float dn = dH * (1.f - u) * (1.f - nn * nn);
                float du = dH * (hprev - nn) * u * (1.f - u);
                float dr = dn * hn * r * (1.f - r);
                float* dgi = J.dgi + rix * a.ldg + ht * 48 + gn;
                float* dgh = J.dgh + rix * a.ldg + ht * 48 + gn;
                store4_sc1(dgh, dr); store4_sc1(dgh + 16, du); store4_sc1(dgh + 32, dn * r);   // exchanged
                dgi[0] = dr; dgi[16] = du; dgi[32] = dn;
                *carryp = dH * u;
            }
        }
        if (p > p_last) group_publish(ctr);
    }
}

// ------------------------------------------------------------------------------ launchers
bool gru_dim_supported(int D) { return D == 16 || D == 64 || D == 256 || D == 512; }

template <bool FWD>
static hipError_t launch(hipStream_t st, const GruArgs& a, int grid)
{
#define AVAE_GRU_CASE(KSV)                                                                          \
    case KSV:                                                                                       \
        if (FWD) hipLaunchKernelGGL((gru_fwd_kernel<KSV>), dim3(grid), dim3(256), 0, st, a);        \
        else     hipLaunchKernelGGL((gru_bwd_kernel<KSV>), dim3(grid), dim3(256), 0, st, a);        \
        break;
    switch (a.D / 16) {
        AVAE_GRU_CASE(1)
        AVAE_GRU_CASE(4)
        AVAE_GRU_CASE(16)
        AVAE_GRU_CASE(32)
        default: return hipErrorInvalidValue;
    }
#undef AVAE_GRU_CASE
    return hipGetLastError();
}

static hipError_t check(const GruArgs& a, int* grid)
{
    if (!gru_dim_supported(a.D) || a.njobs < 1 || a.njobs > kMaxGruJobs) return hipErrorInvalidValue;
    if (a.rows_per_group % 16 || a.G * a.rows_per_group < a.B) return hipErrorInvalidValue;
    *grid = a.njobs * a.G * (a.D / 16);
    if (*grid > 512) return hipErrorInvalidValue;     // residency: 2 workgroups per CU x 256 CUs
    return hipSuccess;
}

hipError_t gru_forward(hipStream_t st, const GruArgs& a, bool persistent)
{
    int grid; hipError_t e = check(a, &grid); if (e != hipSuccess) return e;
    if (persistent) {
        e = hipMemsetAsync(a.counters, 0, sizeof(unsigned) * 64, st); if (e != hipSuccess) return e;
        return launch<true>(st, a, grid);
    }
    for (int p = a.p_begin; p < a.p_end; ++p) {
        GruArgs b = a; b.p_begin = p; b.p_end = p + 1;
        e = launch<true>(st, b, grid); if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t gru_backward(hipStream_t st, const GruArgs& a, bool persistent)
{
    int grid; hipError_t e = check(a, &grid); if (e != hipSuccess) return e;
    if (persistent) {
        e = hipMemsetAsync(a.counters, 0, sizeof(unsigned) * 64, st); if (e != hipSuccess) return e;
        return launch<false>(st, a, grid);
    }
    // one launch per step (descending); the dh0 tail (p = -1) rides with p = 0 only in the
    // persistent form, so split it: steps S-1..1, then [0,1) which also emits dh0 after a sync
    for (int p = a.p_end - 1; p >= a.p_begin; --p) {
        GruArgs b = a; b.p_begin = p; b.p_end = p + 1;
        if (p == 0) {
            // run step 0 without the tail, then the tail alone
            GruArgs c = b; for (int i = 0; i < c.njobs; ++i) c.job[i].dh0 = nullptr;
            e = hipMemsetAsync(a.counters, 0, sizeof(unsigned) * 64, st); if (e != hipSuccess) return e;
            e = launch<false>(st, c, grid); if (e != hipSuccess) return e;
            bool any = false; for (int i = 0; i < a.njobs; ++i) any |= a.job[i].dh0 != nullptr;
            if (any) {
                GruArgs d = b; d.p_begin = 0; d.p_end = 0;   // loop runs p = -1 only
                e = launch<false>(st, d, grid); if (e != hipSuccess) return e;
            }
        } else {
            e = launch<false>(st, b, grid); if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

}  // namespace avae
