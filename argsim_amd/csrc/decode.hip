// decode.hip -- the greedy decoding loop of reference src/model.py:204-219 as ONE persistent launch.
//
// The reference feeds one token per session call: x, s = sess.run((pred, state_ex), {lead: x, state_in: s}) until
// every row has emitted eos or `steps` tokens were produced (explore_centroids.py:40 uses steps = 512).  Here the
// whole loop runs inside one kernel, one workgroup per CU:
//   * every workgroup keeps ITS slice of every weight in LDS for the whole call -- per decoder layer the three gate
//     rows of W and R of the hidden units it owns (unit u belongs to workgroup u mod G), its columns of the `out`
//     affine and its block of embedding rows for the tied logits (D = 512, V = 8192, G = 256: 74 + 4 + 64 KB);
//     per token only activations move (a few KB per row), no weight byte is re-read;
//   * a token is six phases -- GRU layer 1..L, out affine, logits + partial argmax, final argmax -- separated by a
//     grid-wide barrier.  Everything one workgroup writes for another (states, `out` rows, partial maxima, the new
//     token ids) is stored write-through (sc1), every storing wave drains its stores, the workgroup meets at a
//     barrier and ONE lane adds to a monotonic counter; consumers poll that counter with sc1 loads, meet at a
//     workgroup barrier and read the data with sc1 loads only (MI355X_MICROARCH.md, valid hand-off forms).  No fence.
//   * every spin is bounded (error word, as in gru.hip); the grid must be resident at once, which the launcher checks
//     against the occupancy query.
// Arithmetic: fp32 dot products with lane-split K and a butterfly reduction; same gate equations and activation
// approximations as gru.hip.  The summation order differs from the MFMA path of avae_decode_step, so logits agree to
// rounding, token ids exactly unless two logits tie to the last bit.
#include "kernels.h"

namespace avae {

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

// 16-byte L1-bypassing load of base[idx .. idx+3]; `base` must be wave-uniform (it becomes the buffer resource: a
// per-lane base would be serialised lane by lane in a waterfall loop)
__device__ __forceinline__ float4 ld16_sc1(const float* base, unsigned idx)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(idx * 4u), 0, 16);          // aux 16 = sc1
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float ld4_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld4i_sc1(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st4_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st4i_sc1(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// sum over the 64 lanes, returned in every lane: DPP adds inside the 16-lane rows, row broadcasts across them, one
// readlane (vector ALU only -- the shuffle form goes through the LDS pipe, which the weight reads already load)
__device__ __forceinline__ float wave_sum(float v)
{
#define AVAE_DPP_ADD(ctrl, rows) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rows, 0xf, false))
    AVAE_DPP_ADD(0xB1, 0xf);        // quad_perm [1,0,3,2]
    AVAE_DPP_ADD(0x4E, 0xf);        // quad_perm [2,3,0,1]
    AVAE_DPP_ADD(0x141, 0xf);       // row_half_mirror
    AVAE_DPP_ADD(0x140, 0xf);       // row_mirror: every lane of a row now holds the row's sum
    AVAE_DPP_ADD(0x142, 0xa);       // row_bcast15 into rows 1 and 3
    AVAE_DPP_ADD(0x143, 0xc);       // row_bcast31 into rows 2 and 3: lane 63 holds the total
#undef AVAE_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// grid-wide barrier (see the header): returns false once the error word is set
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned& target, int G, int* err)
{
    __shared__ int s_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        target += (unsigned)G;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long t0 = 0; unsigned n = 0; int ok = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++n & 63u) != 0) continue;
            if (t0 == 0) t0 = __builtin_amdgcn_s_memrealtime();
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ULL) {                  // 2 s at 100 MHz
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break;
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

}  // namespace

constexpr int kRows = 1;            // batch rows per wave and pass (2 was measured slower: register spills at 16 waves per CU)
constexpr int kDecThreads = 1024, kDecWaves = kDecThreads / 64;     // sixteen rows of the batch in flight per workgroup

__global__ __launch_bounds__(kDecThreads) void decode_greedy_kernel(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x, w = blockIdx.x;
    const int D = a.D, V = a.V, L = a.L, b = a.b;
    const int nu = (D + G - 1) / G;                 // hidden units (and `out` columns) per workgroup
    const int vp = (V + G - 1) / G;                 // vocabulary rows per workgroup
    const int v0 = min(V, w * vp), v1 = min(V, v0 + vp);
    // LDS: [layer][slot][6][D] gate rows (W_r W_u W_n R_r R_u R_n) | [slot][D] columns of the out affine | [vp][D] embedding rows
    float* Wl = lds;
    float* Kl = Wl + (size_t)L * nu * 6 * D;
    float* El = Kl + (size_t)nu * D;
    for (int l = 0; l < L; ++l)
        for (int s = 0; s < nu; ++s) {
            const int u = w + s * G;
            for (int i = tid; i < 6 * D; i += kDecThreads) {
                const int g = i / D, k = i - g * D;
                float v = 0.f;
                if (u < D) {
                    const int row = (u / 16) * 48 + (u % 16) * 3 + (g % 3);            // G16 row order (kernels.h)
                    v = (g < 3 ? a.W[l] : a.R[l])[(size_t)row * D + k];
                }
                Wl[((size_t)(l * nu + s) * 6 + g) * D + k] = v;
            }
        }
    for (int s = 0; s < nu; ++s) {
        const int u = w + s * G;
        for (int k = tid; k < D; k += kDecThreads) Kl[(size_t)s * D + k] = u < D ? a.Kout[(size_t)k * D + u] : 0.f;      // kernel is (in, out)
    }
    if (a.cache_e)
        for (int i = tid; i < (v1 - v0) * D; i += kDecThreads) El[i] = a.E[(size_t)v0 * D + i];
    __syncthreads();

    unsigned target = 0;
    int cur = 0, kept = a.steps;
    bool ok = true;
    for (int t = 0; t < a.steps && ok; ++t) {
        const int32_t* lead = a.ids_tm + (size_t)t * b;
        // ---- GRU layers: h'_l = GRU(x_l, h_l), x_1 = E[lead], x_l = h'_{l-1}
        for (int l = 0; l < L && ok; ++l) {
            const float* hcur = a.state[cur] + (size_t)l * b * D;
            float* hnew = a.state[cur ^ 1] + (size_t)l * b * D;
            const float* xsrc = l ? a.state[cur ^ 1] + (size_t)(l - 1) * b * D : nullptr;
            // kRows rows of the batch per wave and pass: their loads are issued together (one L2 round trip per pass)
            for (int bb0 = wave; bb0 < b; bb0 += kDecWaves * kRows) {
                float acc[kRows][2][6], hp[kRows][2];
                int id[kRows];
#pragma unroll
                for (int i = 0; i < kRows; ++i) {
                    const int bb = min(bb0 + i * kDecWaves, b - 1);      // clamped: a duplicate row is computed, not stored
                    id[i] = 0;
                    if (l == 0) { id[i] = ld4i_sc1(lead + bb); id[i] = id[i] < 0 ? 0 : (id[i] >= V ? V - 1 : id[i]); }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int u = w + s * G;
                        hp[i][s] = (s < nu && u < D) ? ld4_sc1(hcur + (size_t)bb * D + u) : 0.f;
#pragma unroll
                        for (int g = 0; g < 6; ++g) acc[i][s][g] = 0.f;
                    }
                }
                for (int k = lane * 4; k < D; k += 256) {
                    float4 x[kRows], hh[kRows];
#pragma unroll
                    for (int i = 0; i < kRows; ++i) {
                        const int bb = min(bb0 + i * kDecWaves, b - 1);
                        x[i] = l ? ld16_sc1(xsrc, (unsigned)(bb * D + k)) : *reinterpret_cast<const float4*>(a.E + (size_t)id[i] * D + k);
                        hh[i] = ld16_sc1(hcur, (unsigned)(bb * D + k));
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        if (s >= nu) break;
                        const float* wr = Wl + ((size_t)(l * nu + s) * 6) * D + k;
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            const float4 wv = *reinterpret_cast<const float4*>(wr + (size_t)g * D);
                            const float4 rv = *reinterpret_cast<const float4*>(wr + (size_t)(3 + g) * D);
#pragma unroll
                            for (int i = 0; i < kRows; ++i) { acc[i][s][g] += dot4(x[i], wv); acc[i][s][3 + g] += dot4(hh[i], rv); }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < kRows; ++i) {
                    const int bb = bb0 + i * kDecWaves;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        if (s >= nu) break;
                        const int u = w + s * G;
                        float gsum[6];
#pragma unroll
                        for (int g = 0; g < 6; ++g) gsum[g] = wave_sum(acc[i][s][g]);
                        if (lane == 0 && u < D && bb < b) {
                            const int c = (u / 16) * 48 + (u % 16) * 3;
                            const float gi_r = gsum[0] + a.bW[l][c], gi_u = gsum[1] + a.bW[l][c + 1], gi_n = gsum[2] + a.bW[l][c + 2];
                            const float gh_r = gsum[3] + a.bR[l][c], gh_u = gsum[4] + a.bR[l][c + 1], gh_n = gsum[5] + a.bR[l][c + 2];
                            const float r = sigm(gi_r + gh_r), z = sigm(gi_u + gh_u), n = tanh_(gi_n + r * gh_n);
                            st4_sc1(hnew + (size_t)bb * D + u, (1.f - z) * n + z * hp[i][s]);
                        }
                    }
                }
            }
            ok = grid_barrier(a.bar, target, G, a.err);
        }
        if (!ok) break;
        // ---- out affine (model.py:162): o = h'_L Kout + bout
        {
            const float* hl = a.state[cur ^ 1] + (size_t)(L - 1) * b * D;
            for (int bb = wave; bb < b; bb += kDecWaves) {
                float acc[2] = {0.f, 0.f};
                for (int k = lane * 4; k < D; k += 256) {
                    const float4 hh = ld16_sc1(hl, (unsigned)(bb * D + k));
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        if (s < nu) acc[s] += dot4(hh, *reinterpret_cast<const float4*>(Kl + (size_t)s * D + k));
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (s >= nu) break;
                    const int u = w + s * G;
                    const float v = wave_sum(acc[s]);
                    if (lane == 0 && u < D) st4_sc1(a.o + (size_t)bb * D + u, v + a.bout[u]);
                }
            }
            ok = grid_barrier(a.bar, target, G, a.err);
            if (!ok) break;
        }
        // ---- tied logits (model.py:166) over this workgroup's vocabulary rows, partial argmax (first maximum)
        for (int bb0 = wave; bb0 < b; bb0 += kDecWaves * kRows) {
            float best[kRows]; int besti[kRows];
            float4 ov[kRows][2];                                  // this lane's slice of the `out` rows (D <= 512: two pieces)
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const int bb = min(bb0 + r * kDecWaves, b - 1);
                best[r] = -INFINITY; besti[r] = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = lane * 4 + 256 * i;
                    ov[r][i] = k < D ? ld16_sc1(a.o, (unsigned)(bb * D + k)) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            for (int vb = v0; vb < v1; vb += 8) {                 // eight rows of the table per pass, each read once for all kRows
                float p[kRows][8];
#pragma unroll
                for (int r = 0; r < kRows; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) p[r][j] = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = lane * 4 + 256 * i;
                    if (k >= D) break;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (vb + j >= v1) break;
                        const float* er = a.cache_e ? El + (size_t)(vb + j - v0) * D + k : a.E + (size_t)(vb + j) * D + k;
                        const float4 ev = *reinterpret_cast<const float4*>(er);
#pragma unroll
                        for (int r = 0; r < kRows; ++r) p[r][j] += dot4(ov[r][i], ev);
                    }
                }
#pragma unroll
                for (int r = 0; r < kRows; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (vb + j >= v1) break;
                        const float v = wave_sum(p[r][j]) * a.isd;
                        if (v > best[r]) { best[r] = v; besti[r] = vb + j; }      // ascending v: the first maximum stays
                    }
            }
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const int bb = bb0 + r * kDecWaves;
                if (lane == 0 && bb < b) { st4_sc1(a.part_val + (size_t)w * b + bb, best[r]); st4i_sc1(a.part_idx + (size_t)w * b + bb, besti[r]); }
            }
        }
        ok = grid_barrier(a.bar, target, G, a.err);
        if (!ok) break;
        // ---- final argmax of row bb by workgroup bb mod G (vocabulary blocks ascend with the workgroup index)
        if (wave == 0)
            for (int bb = w; bb < b; bb += G) {
                float best = -INFINITY; int besti = 0x7fffffff;
                for (int g = lane; g < G; g += 64) {
                    const float v = ld4_sc1(a.part_val + (size_t)g * b + bb);
                    const int vi = ld4i_sc1(a.part_idx + (size_t)g * b + bb);
                    if (v > best || (v == best && vi < besti)) { best = v; besti = vi; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(besti, o, 64);
                    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
                }
                if (lane == 0) st4i_sc1(a.ids_tm + (size_t)(t + 1) * b + bb, besti);
            }
        ok = grid_barrier(a.bar, target, G, a.err);
        if (!ok) break;
        cur ^= 1;
        // model.py:217: stop when every row emitted eos; that all-eos step is not appended
        int all = 1;
        for (int bb = tid; bb < b; bb += kDecThreads) all &= ld4i_sc1(a.ids_tm + (size_t)(t + 1) * b + bb) == a.eos;
        if (__syncthreads_and(all)) { kept = t; break; }
    }
    // (b, steps) row-major output, eos beyond the kept tokens
    for (size_t i = (size_t)w * kDecThreads + tid; i < (size_t)b * a.steps; i += (size_t)G * kDecThreads) {
        const int bb = (int)(i / a.steps), s = (int)(i % a.steps);
        a.out_ids[i] = s < kept ? ld4i_sc1(a.ids_tm + (size_t)(s + 1) * b + bb) : a.eos;
    }
    if (w == 0 && tid == 0) *a.kept = kept;
}

int decode_workgroups()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
}

hipError_t decode_greedy(hipStream_t st, DecodeArgs a, int* grid_out)
{
    int dev = 0, cus = 0, lds_max = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    if (e != hipSuccess) return e;
    if (lds_max < 160 * 1024) lds_max = 160 * 1024;            // gfx950: one workgroup may own the CU's whole LDS
    const int G = cus;
    const int nu = (a.D + G - 1) / G, vp = (a.V + G - 1) / G;
    if (nu > 2 || a.L > 8 || (a.D & 3) || a.D > 512) return hipErrorInvalidValue;
    size_t floats = (size_t)a.L * nu * 6 * a.D + (size_t)nu * a.D;
    const size_t with_e = floats + (size_t)vp * a.D;
    a.cache_e = with_e * 4 + 64 <= (size_t)lds_max - 1024 ? 1 : 0;
    if (a.cache_e) floats = with_e;
    const int lds_bytes = (int)(floats * 4 + 64);
    if ((size_t)lds_bytes > (size_t)lds_max) return hipErrorInvalidValue;
    static int attr_set = 0;
    if (attr_set < lds_bytes) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_greedy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = lds_bytes;
    }
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_greedy_kernel, kDecThreads, (size_t)lds_bytes);
    if (e != hipSuccess) return e;
    if (per_cu < 1) return hipErrorCooperativeLaunchTooLarge;     // the grid barrier needs every workgroup resident
    if (grid_out) *grid_out = G;
    hipLaunchKernelGGL(decode_greedy_kernel, dim3(G), dim3(kDecThreads), lds_bytes, st, a);
    return hipGetLastError();
}

}  // namespace avae
