// gru_dev.h -- device-side helpers shared by the GRU kernel files (gru.hip, gru_rs.hip): fragment loads with the in-band
// "not yet written" check, buffer-instruction wrappers, bounded spins, the same-XCD rendezvous, the tiled exchange index maps,
// team barriers and the team-kernel workgroup map.  Internal; not part of the C ABI.
#pragma once
#include "kernels.h"

namespace avae {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two floats -> two bf16 (round to nearest even, v_cvt_pk_bf16_f32), `lo` in the low half
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// 16-bit exchange (bf16 mode, backward team kernels): the exchanged operand itself is stored as bf16, eight k per 16-byte
// chunk -- element (pos, row, k) at halfword
//     X[(((pos * B/16 + row/16) * K/8 + k/8) * 16 + row%16) * 8 + k%8]
// -- so one load instruction of a wave still reads 1 KB of contiguous memory and IS the v_mfma_f32_16x16x32_bf16 A operand
// (half the bytes through the texture addresser, no conversion, the whole K range of a wave in flight at once).  "Not
// yet written" is the halfword 0xFFFF (the launcher's fill); a stored value with that pattern (a NaN) becomes 0x7FC0.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned xch16_index(int pos, int row, int k, int B, int K)
{
    return ((((unsigned)pos * (unsigned)(B >> 4) + (unsigned)(row >> 4)) * (unsigned)(K >> 3) + (unsigned)(k >> 3)) * 16u + (unsigned)(row & 15)) * 8u + (unsigned)(k & 7);
}
__device__ __forceinline__ unsigned short bf16_not_sentinel(float v)
{
    const unsigned short h = (unsigned short)(pack_bf16(v, 0.f) & 0xffffu);
    return h == 0xffffu ? (unsigned short)0x7fc0u : h;
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ bool any_half_sentinel(unsigned mx) { return (mx & 0xffffu) == 0xffffu || (mx >> 16) == 0xffffu; }
// bf16-operand mode in the register-form kernels: the operand keeps its fp32 register but carries a bf16 value (products of
// two such values are exact in fp32, so these kernels and the team kernels' bf16 MFMA differ only in summation order)
__device__ __forceinline__ float opnd(float x, bool bf) { return bf ? __uint_as_float(pack_bf16(0.f, x)) : x; }
// eight floats (two 16-byte pieces of an MFMA A fragment) -> one v_mfma_f32_16x16x32_bf16 operand.  Which k a slot
// stands for is the caller's business: the B operand in LDS is packed with the same slot order.
__device__ __forceinline__ bf16x8 pack_bf16x8(const u32x4& p0, const u32x4& p1)
{
    const u32x4 v = {pack_bf16(__uint_as_float(p0.x), __uint_as_float(p0.y)), pack_bf16(__uint_as_float(p0.z), __uint_as_float(p0.w)),
                     pack_bf16(__uint_as_float(p1.x), __uint_as_float(p1.y)), pack_bf16(__uint_as_float(p1.z), __uint_as_float(p1.w))};
    return __builtin_bit_cast(bf16x8, v);
}

constexpr unsigned kSentinel = 0xFFFFFFFFu;

// An exchanged value must never alias the "not yet written" pattern: a NaN that reaches the gate math with every
// payload bit set (e.g. from a poisoned parameter) would otherwise make every consumer spin until its 2 s bound.
// Such a value is stored as the canonical quiet NaN instead: still a NaN, never the sentinel.
__device__ __forceinline__ float not_sentinel(float v)
{
    return __float_as_uint(v) == kSentinel ? __uint_as_float(0x7FC00000u) : v;
}
__device__ __forceinline__ int pos_map(int p, int len, int reverse) { return (reverse && p < len) ? (len - 1 - p) : p; }
// sigmoid / tanh on v_exp_f32 + v_rcp_f32 (each <= 1 ulp): absolute error ~1e-7, far inside the
// fp32 parity tolerances, at a fraction of the libm cost that sat on the per-step critical path
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return __builtin_fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)), -1.f); }
// One cell update, shared by every forward kernel form.  Which multiply-adds are fused is written out: left to the
// compiler's contraction the choice followed the surrounding code, and two forms of the same step (the team kernels and
// the one-launch-per-step kernels the tests compare them with) differed in the last bit.
struct GruCellOut { float r, u, n, h; };
__device__ __forceinline__ GruCellOut gru_cell(float gi_r, float gi_u, float gi_n, float gh_r, float gh_u, float gh_n, float hprev)
{
    GruCellOut c;
    c.r = sigmoidf_(gi_r + gh_r);
    c.u = sigmoidf_(gi_u + gh_u);
    c.n = tanhf_(__builtin_fmaf(c.r, gh_n, gi_n));
    float t = (1.f - c.u) * c.n;
    asm volatile("" : "+v"(t));                   // a product of its own, not a candidate for fusion with the sum below
    c.h = __builtin_fmaf(c.u, hprev, t);
    return c;
}

// k offset (inside one wave's contiguous K range) of MFMA step ks for lane quarter kh
template <int NKS>
__device__ __forceinline__ int kperm(int ks, int kh)
{
    if (NKS % 4 == 0) return 16 * (ks >> 2) + 4 * kh + (ks & 3);
    return 4 * ks + kh;
}

// diagnostic phase stamps (ablate bit 32): 100 MHz real-time counter deltas summed per workgroup
// (compile-time: the production instantiation carries no stamp code at all)
#define AVAE_STAMP(i) do { if constexpr (STAMPS) { unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); ph[i] += t_ - tprev; tprev = t_; } } while (0)

__device__ __forceinline__ u32x4 load16_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off)
{
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 16);     // aux 16 = sc1
}
// Buffer forms of the gate-phase accesses: 32-bit byte offsets against a wave-uniform descriptor.  On the texture
// addresser a buffer instruction costs about half of the global (64-bit address per lane) form of the same access
// (scripts/micro/ta_cost.hip: x4 store of 4 rows x 256 B 43 vs 83 cycles, dword store 31 vs 63).
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t rs, unsigned off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0)); }
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t rs, unsigned off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void bstore1(float v, __amdgpu_buffer_rsrc_t rs, unsigned off) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)off, 0, 0); }
__device__ __forceinline__ void bstore1_sc1(float v, __amdgpu_buffer_rsrc_t rs, unsigned off) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)off, 0, 16); }
__device__ __forceinline__ void bstore3(float x, float y, float z, __amdgpu_buffer_rsrc_t rs, unsigned off)
{
    const u32x3 v = {__float_as_uint(x), __float_as_uint(y), __float_as_uint(z)};
    __builtin_amdgcn_raw_buffer_store_b96(v, rs, (int)off, 0, 0);
}
__device__ __forceinline__ void bstore4(float x, float y, float z, float w, __amdgpu_buffer_rsrc_t rs, unsigned off)
{
    const u32x4 v = {__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), __float_as_uint(w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0);
}
__device__ __forceinline__ unsigned load4_sc1(const float* p)
{
    return __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store4_sc1(float* p, float v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, -1, 0x00020000);
}

// bounded spin bookkeeping shared by the data polls: returns true when the wave must give up
struct SpinGuard {
    unsigned long long t0 = 0; unsigned n = 0;
    __device__ __forceinline__ bool expired(int* err)
    {
        __builtin_amdgcn_s_sleep(4);
        if ((++n & 31u) != 0) return false;
        if (t0 == 0) t0 = __builtin_amdgcn_s_memrealtime();
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ULL) {   // 2 s at 100 MHz
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        return false;
    }
};

// Loads NQ 16-byte pieces (stride 64 B) of one row per lane and repeats until no dword of the
// WAVE's fragment equals the sentinel (poll == false: single pass).
template <int NQ>
__device__ __forceinline__ void load_frag(float* dst, __amdgpu_buffer_rsrc_t rs, unsigned off, bool poll, int* err)
{
    SpinGuard sg;
    for (;;) {
        u32x4 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = load16_sc1(rs, off + 64u * q);
        bool bad = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q) bad |= (v[q].x == kSentinel) | (v[q].y == kSentinel) | (v[q].z == kSentinel) | (v[q].w == kSentinel);
        if (!poll || !__any(bad) || sg.expired(err)) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                dst[4 * q + 0] = __uint_as_float(v[q].x); dst[4 * q + 1] = __uint_as_float(v[q].y);
                dst[4 * q + 2] = __uint_as_float(v[q].z); dst[4 * q + 3] = __uint_as_float(v[q].w);
            }
            return;
        }
    }
}
// Pipelined form: the loads of a fragment are ISSUED (no wait) and the fragment is verified later, at its
// first use.  Written so that hipcc's waitcnt insertion stays counted on the fast path: younger fragments
// issued in straight-line code before the check stay in flight (vmcnt(N)); only the rare re-load loop,
// which consumes what it loads inside the loop, waits for everything.
template <int NQ>
__device__ __forceinline__ void frag_issue(u32x4 (&v)[NQ], __amdgpu_buffer_rsrc_t rs, unsigned off)
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = load16_sc1(rs, off + 64u * q);
}
// the same with a run-time (wave-uniform) distance between the pieces: it travels in the scalar offset operand
template <int NQ>
__device__ __forceinline__ void frag_issue(u32x4 (&v)[NQ], __amdgpu_buffer_rsrc_t rs, unsigned off, unsigned qstride)
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, (int)(qstride * q), 16);
}
template <int NQ>
__device__ __forceinline__ bool frag_bad(const u32x4 (&v)[NQ])
{
    unsigned mx = 0u;                     // the sentinel is the largest unsigned value: one running maximum
#pragma unroll
    for (int q = 0; q < NQ; ++q) mx = max(max(mx, max(v[q].x, v[q].y)), max(v[q].z, v[q].w));
    return __any(mx == kSentinel);
}
// verify `v`; on sentinels re-load it until complete (bounded)
template <int NQ>
__device__ __forceinline__ void frag_ensure(u32x4 (&v)[NQ], __amdgpu_buffer_rsrc_t rs, unsigned off, int* err)
{
    if (frag_bad<NQ>(v)) {
        SpinGuard sg;
        do { frag_issue<NQ>(v, rs, off); } while (frag_bad<NQ>(v) && !sg.expired(err));
    }
}
template <int NQ>
__device__ __forceinline__ void frag_ensure(u32x4 (&v)[NQ], __amdgpu_buffer_rsrc_t rs, unsigned off, int* err, unsigned qstride)
{
    if (frag_bad<NQ>(v)) {
        SpinGuard sg;
        do { frag_issue<NQ>(v, rs, off, qstride); } while (frag_bad<NQ>(v) && !sg.expired(err));
    }
}
// the same for a 16-bit exchange: "not yet written" is the halfword 0xFFFF (one running v_pk_max_u16)
template <int NQ>
__device__ __forceinline__ bool frag_bad16(const u32x4 (&v)[NQ])
{
    unsigned mx = 0u;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const unsigned a = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(unsigned short __attribute__((ext_vector_type(2))), v[q].x), __builtin_bit_cast(unsigned short __attribute__((ext_vector_type(2))), v[q].y)));
        const unsigned b = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(unsigned short __attribute__((ext_vector_type(2))), v[q].z), __builtin_bit_cast(unsigned short __attribute__((ext_vector_type(2))), v[q].w)));
        const unsigned c = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(unsigned short __attribute__((ext_vector_type(2))), a), __builtin_bit_cast(unsigned short __attribute__((ext_vector_type(2))), b)));
        mx = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(unsigned short __attribute__((ext_vector_type(2))), mx), __builtin_bit_cast(unsigned short __attribute__((ext_vector_type(2))), c)));
    }
    return __any((mx & 0xffffu) == 0xffffu || (mx >> 16) == 0xffffu);
}
template <int NQ>
__device__ __forceinline__ void frag_ensure16(u32x4 (&v)[NQ], __amdgpu_buffer_rsrc_t rs, unsigned off, int* err, unsigned qstride)
{
    if (frag_bad16<NQ>(v)) {
        SpinGuard sg;
        do { frag_issue<NQ>(v, rs, off, qstride); } while (frag_bad16<NQ>(v) && !sg.expired(err));
    }
}

// Hand-counted form of the same pipeline for the D = 512 backward pass.  hipcc's waitcnt insertion falls
// back to vmcnt(0) around the re-load branches of frag_ensure, which serialises every piece behind a
// full L2 round trip; these loads are invisible to it (inline asm), are waited for by explicit counted
// s_waitcnt statements that name the destination registers, and are fully retired before the code leaves
// the block that issued them (cdna_hip_programming.md section 5.7 forms (ii)/(iii)).
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void asm_issue6(u32x4 (&v)[6], unsigned voff, i32x4 srd)
{
    asm volatile("buffer_load_dwordx4 %0, %6, %7, 0 offen offset:%c8 sc1\n\t"
                 "buffer_load_dwordx4 %1, %6, %7, 0 offen offset:%c9 sc1\n\t"
                 "buffer_load_dwordx4 %2, %6, %7, 0 offen offset:%c10 sc1\n\t"
                 "buffer_load_dwordx4 %3, %6, %7, 0 offen offset:%c11 sc1\n\t"
                 "buffer_load_dwordx4 %4, %6, %7, 0 offen offset:%c12 sc1\n\t"
                 "buffer_load_dwordx4 %5, %6, %7, 0 offen offset:%c13 sc1"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                 : "v"(voff), "s"(srd), "i"(OFF), "i"(OFF + 64), "i"(OFF + 128), "i"(OFF + 192), "i"(OFF + 256), "i"(OFF + 320)
                 : "memory");
}
// the same six loads from the tiled exchange: 1 KB between consecutive 16-float groups (soffset carries what the 12-bit
// immediate cannot)
__device__ __forceinline__ void asm_issue6x(u32x4 (&v)[6], unsigned voff, i32x4 srd, int soff0, int soff1)
{
    asm volatile("buffer_load_dwordx4 %0, %6, %7, %8 offen offset:0 sc1\n\t"
                 "buffer_load_dwordx4 %1, %6, %7, %8 offen offset:1024 sc1\n\t"
                 "buffer_load_dwordx4 %2, %6, %7, %8 offen offset:2048 sc1\n\t"
                 "buffer_load_dwordx4 %3, %6, %7, %8 offen offset:3072 sc1\n\t"
                 "buffer_load_dwordx4 %4, %6, %7, %9 offen offset:0 sc1\n\t"
                 "buffer_load_dwordx4 %5, %6, %7, %9 offen offset:1024 sc1"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                 : "v"(voff), "s"(srd), "s"(soff0), "s"(soff1)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void asm_wait6(u32x4 (&v)[6])
{
    asm volatile("s_waitcnt vmcnt(%c6)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) : "i"(N) : "memory");
}

// scalar-element variant for the small test dimensions (wave K range not a multiple of 16)
template <int N, int NKS>
__device__ __forceinline__ void load_frag_scalar(float* dst, const float* base, int ks0, int kh, bool poll, int* err)
{
    SpinGuard sg;
    for (;;) {
        unsigned v[N]; bool bad = false;
#pragma unroll
        for (int i = 0; i < N; ++i) { v[i] = load4_sc1(base + kperm<NKS>(ks0 + i, kh)); bad |= v[i] == kSentinel; }
        if (!poll || !__any(bad) || sg.expired(err)) {
#pragma unroll
            for (int i = 0; i < N; ++i) dst[i] = __uint_as_float(v[i]);
            return;
        }
    }
}

// Same-XCD detection (speed only): every workgroup of a group publishes the id of the XCD it runs
// on with the placement-independent sc1 protocol; if all HT ids agree the group's per-step stores
// may stay in that XCD's L2 (plain stores) instead of being written through to the memory side.
// All members read the same HT words, so they all take the same decision.
// first half of the rendezvous, for kernels that have work to do before they need the answer (the team kernels copy their
// weight slice to LDS in between: ~6 us per launch during which the other workgroups' publications arrive)
__device__ __forceinline__ void xcd_publish(unsigned* ctr2, unsigned* ids, int ht)
{
    if (threadIdx.x == 0) {
        const unsigned my = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu;     // HW_REG_XCC_ID
        __hip_atomic_store(ids + ht, my + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ bool group_same_xcd(unsigned* ctr2, unsigned* ids, int ht, int HT, int* err, int force_slow, bool published = false)
{
    __shared__ int s_fast;
    if (threadIdx.x < 64) {          // the first wave: lane 0 publishes and waits, then lane i reads id i (one round trip, not HT)
        const unsigned my = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu;     // HW_REG_XCC_ID
        bool ok = true;
        if (threadIdx.x == 0) {
            if (!published) {
            __hip_atomic_store(ids + ht, my + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(ctr2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            SpinGuard sg;
            while (__hip_atomic_load(ctr2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)HT)
                if (sg.expired(err)) { ok = false; break; }
        }
        ok = __all(ok);
        bool same = true;
        for (int i = threadIdx.x; i < HT; i += 64)
            same = same && __hip_atomic_load(ids + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my + 1u;
        same = __all(same) && ok && !force_slow;
        if (threadIdx.x == 0) s_fast = same ? 1 : 0;
    }
    __syncthreads();
    return s_fast != 0;
}

// blockIdx -> (job, group, hidden tile).  Under round-robin dispatch blocks b and b + grid/2 share a
// CU (2 workgroups per CU): the map gives them DIFFERENT chains (the other direction, or another batch
// group) so one's MFMA phase can overlap the other's exchange latency, keeps blockIdx % 8 = group % 8
// (a group's workgroups on one XCD), and reports which half a block is in so that half can be
// staggered by part of a step.  Any placement is correct; this is speed only.
__device__ __forceinline__ bool wg_map(const GruArgs& a, int HT, int* jb, int* g, int* ht)
{
    const int per_job = a.G * HT, bid = blockIdx.x;
    // alt = which of two interleaved sets a block belongs to: blocks b and b+8 (consecutive blocks of one
    // XCD, which the dispatcher was observed to place on the same CU) get different chains
    const int alt = (bid >> 3) & 1, idx = ((bid >> 4) << 3) | (bid & 7);
    if (a.njobs == 2 && (per_job & 7) == 0) {                       // the two encoder directions
        *jb = alt; *g = idx % a.G; *ht = idx / a.G;
        return alt != 0;
    }
    if (a.njobs == 1 && a.G == 16) {                                // one decoder layer: groups g and g+8
        *jb = 0; *g = (bid & 7) + 8 * alt; *ht = bid >> 4;
        return alt != 0;
    }
    *jb = bid / per_job;
    const int rem = bid - *jb * per_job;
    *g = rem % a.G; *ht = rem / a.G;
    return false;
}

// ---- tiled exchange (team kernels).  The MFMA A operand puts 16 different batch rows in 16 adjacent lanes.  Read from
// a row-major (pos, row, k) buffer, every lane of a 16-byte load touches another cache line (rows are 2-6 KB apart) and
// the CU's texture addresser spends ~64 cycles per load instruction: TA_BUSY was 93 % of the backward kernel and 48 %
// of the forward (gpurun_out/pp1_p5), the real limiter of the step.  The exchanged operand therefore lives in a
// scratch buffer in fragment order -- element (pos, row, k) at
//     X[(((pos * B/16 + row/16) * K/4 + k/4) * 16 + row%16) * 4 + k%4]
// -- so that one load instruction of a wave reads 1 KB of contiguous memory (lane (n, kh): 16 bytes at n*16 + kh*256).
// The row-major hs / dgh the GEMMs consume are written beside it with plain stores and are no longer polled.
// (float index; a job's buffer is below 2^32 bytes -- team_geometry checks -- so 32-bit arithmetic is exact)
__device__ __forceinline__ unsigned xch_index(int pos, int row, int k, int B, int K)
{
    return ((((unsigned)pos * (unsigned)(B >> 4) + (unsigned)(row >> 4)) * (unsigned)(K >> 2) + (unsigned)(k >> 2)) * 16u + (unsigned)(row & 15)) * 4u + (unsigned)(k & 3);
}
// byte offset of lane (n, kh)'s 16 bytes of chunk kc0 + kh, rows row0 .. row0+15 (row0 % 16 == 0) at position pos
__device__ __forceinline__ unsigned xch_lane_offset(int pos, int row0, int kc0, int n, int kh, int B, int K)
{
    return ((((unsigned)pos * (unsigned)(B >> 4) + (unsigned)(row0 >> 4)) * (unsigned)(K >> 2) + (unsigned)(kc0 + kh)) * 16u + (unsigned)n) * 16u;
}

__device__ __forceinline__ void team_barrier(unsigned* word, unsigned target)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // this wave's LDS traffic is done
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}


// blockIdx -> (chain group, hidden tile): the grid is C x 32 workgroups, C <= 8 chain groups.  A chain group is a job
// plus the 64-row blocks slot, slot + cpj, ... of that job (cpj = chain groups per job); the 32 workgroups of a chain
// group exchange with each other and share blockIdx % C (one XCD under round-robin dispatch when C = 8; speed only).
struct TeamMap { int cid, ht, jb, slot, cpj, nrb; };
__device__ __forceinline__ TeamMap team_map(const GruArgs& a, int rows_per_block)
{
    TeamMap m;
    const int C = gridDim.x / 32;
    m.cid = blockIdx.x % C; m.ht = blockIdx.x / C;
    m.cpj = C / a.njobs;
    m.jb = m.cid / m.cpj; m.slot = m.cid % m.cpj;
    m.nrb = ((a.Bx > a.B ? a.Bx : a.B) / rows_per_block) / m.cpj;
    return m;
}

// Padding skipped (GruArgs::slens / perm, optional): slens[slot] = the steps row `slot` really has (its length; decoder: + 1),
// perm[slot] = the batch row that sits in that slot -- rows sorted by length and dealt over the workgroups by the caller
// (ops.hip row_order), so that a team's 16 rows are of similar length.  A team runs a row block for
//     nst = max over its 16 rows of slens (made non-increasing over a workgroup's row blocks by a suffix maximum)
// steps instead of S; the positions behind them -- padding in both time orientations -- are zero-filled in the row-major
// outputs (hs / hp forward, dgi / dgh backward: the GEMMs that sum over all rows must meet finite zeros there, as they did
// when the steps were computed) and never touch the exchange.  All 32 workgroups of a chain group read the same slens, so
// they agree on every team's step count; any perm is correct, a sorted one is fast.  The exchange scratch is indexed by
// slot, every external array by perm[slot].
__device__ __forceinline__ int team_steps(const int* slens, int row0, int lane)
{
    int v = slens[row0 + (lane & 15)];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

// ---- host side: residency check and launch of the 1024-thread team kernels
// Residency.  The persistent kernels exchange data between workgroups inside one launch, so every workgroup of the
// grid must be resident at once.  A plain launch checks nothing (and the occupancy API is advisory), so the grid is
// compared here with (workgroups per CU the occupancy query admits) x (CU count) once per kernel and launch shape,
// and an oversize grid is refused with hipErrorCooperativeLaunchTooLarge instead of being left to the 2 s spin
// bound.  The kernels used need one (1024-thread team kernels) or two (256-thread kernels, <= 80 SGPRs... 102 SGPRs:
// the hardware admits >= 6 such blocks) workgroups per CU, far from the edge where the API over-reports by one.
template <class K>
static hipError_t resident(K kernel, int threads, int dyn_lds, int grid)
{
    struct Entry { const void* k; int threads, lds, cap; };
    static Entry cache[32]; static int ncache = 0;
    const void* kp = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < ncache; ++i)
        if (cache[i].k == kp && cache[i].threads == threads && cache[i].lds == dyn_lds)
            return grid <= cache[i].cap ? hipSuccess : hipErrorCooperativeLaunchTooLarge;
    int per_cu = 0, dev = 0, cus = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, (size_t)dyn_lds);
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    const int cap = per_cu * cus;
    if (ncache < 32) cache[ncache++] = Entry{kp, threads, dyn_lds, cap};
    return grid <= cap ? hipSuccess : hipErrorCooperativeLaunchTooLarge;
}

template <class K>
static hipError_t launch_team(hipStream_t st, K kernel, const GruArgs& a, int lds_bytes, int C)
{
    static const void* attr_done[64]; static int nattr = 0;
    const void* kp = reinterpret_cast<const void*>(kernel);
    bool seen = false;
    for (int i = 0; i < nattr; ++i) seen |= attr_done[i] == kp;
    if (!seen) {
        hipError_t e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        if (nattr < 64) attr_done[nattr++] = kp;
    }
    const int grid = C * 32;
    hipError_t e = resident(kernel, 1024, lds_bytes, grid);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(1024), lds_bytes, st, a);
    return hipGetLastError();
}


}  // namespace avae
