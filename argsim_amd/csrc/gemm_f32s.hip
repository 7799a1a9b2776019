// gemm_f32s.hip -- fp32-accurate GEMM on the bf16 matrix cores by operand splitting ("bf16x3 / 6 products").
//
// gfx950 multiplies bf16 sixteen times faster than fp32 on the matrix cores (v_mfma_f32_32x32x16_bf16: 32 768
// FLOP in 32 cycles; v_mfma_f32_32x32x2_f32: 4 096 FLOP in 64 cycles), and CDNA4 has no xf32.  Every fp32 operand
// element x is split, in registers, on its way from global memory to LDS, into three bf16 values
//     hi = rne(x),  mid = rne(x - hi),  lo = rne(x - hi - mid)          (both subtractions are exact)
// so that x = hi + mid + lo up to 2^-27 |x|.  A product a*b is then accumulated in fp32 from the six partial
// products whose weight is >= 2^-18:  lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi  (each bf16 x bf16 product
// is exact in fp32; the three dropped products are <= 2^-23 |a b| in the worst case, ~2^-27 |a b| rms with zero
// mean: below the 2^-24 rounding of the fp32 accumulation they would be added into).  The result carries the same kind
// and size of error as an fp32 GEMM -- the fp32 accumulation -- and the parity tests hold it to the fp32
// tolerances; it is not a reduced-precision mode.  Six 32-cycle MFMAs replace eight 64-cycle ones per 16-deep
// K step: 2.67x the matrix-core rate of the exact-fp32 kernel (peak 2.5 PFLOP/s / 6 = 417 TFLOP/s fp32-equivalent).
//
// Opt-in (compute_dtype = 2); the default stays v_mfma_f32_32x32x2_f32 (gemm_f32.hip).  Same operand
// conventions as gemm_f32 (GemmArgs, kernels.h): all four layouts, predicated edges, device-side row counts,
// split-K with float atomics.  Operands stay fp32 in HBM: no conversion pass, no extra traffic.
//
// Block tile 128x128x32, 4 waves (2x2), each 2x2 MFMA tiles of 32x32.  LDS holds three bf16 planes per operand,
// 64-byte k-contiguous rows, XOR-swizzled (sw_off) so that the ds_read_b128 fragment reads (lane (r, h): 8
// consecutive k at 16 s + 8 h of row r) and both kinds of staging stores are bank-conflict free.  Operands stored
// [k][x] are transposed in registers (a thread loads a 4x4 patch) so that LDS never sees a 2-byte access.
// Two kernels: gemm_f32s_kernel (three workgroups per CU alternate staging and MFMA phases; all shapes) and
// gemm_f32s_ws_kernel (wave-specialised: 4 MFMA waves + 8 staging waves per CU, double-buffered LDS; long-K
// shapes).  s_setprio on either phase, deeper load prefetch and a hand-interleaved single-wave-per-SIMD form were
// measured and brought nothing.
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

namespace avae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SBK = 32;               // K tile
constexpr int SLD = 32;               // LDS row stride in bf16: 64-byte rows, no padding, XOR-swizzled (below)
constexpr int SPLANE = 128 * SLD;     // bf16 elements of one plane of one operand

// LDS image of one plane: logical (row r, 16-byte chunk c of the row's 64 bytes) lives at physical row
// r ^ ((r >> 2) & 1), chunk c ^ ((r >> 2) & 3).  With the lane groups and bank widths of MI355X_MICROARCH.md
// (ds_read_b128: 16-lane groups {0-3,12-15,20-27}.. over 64 banks; ds_write_b64: 16 contiguous lanes over 32
// banks) the fragment reads (32 rows, one chunk), the k-contiguous stores (2 rows x 8 lanes) and the
// transposing stores (rows 4 apart) are all conflict free.
__device__ __forceinline__ int sw_off(int r, int c)      // bf16 element offset of (row, chunk)
{
    return ((r ^ ((r >> 2) & 1)) << 5) + ((c ^ ((r >> 2) & 3)) << 3);
}

__device__ __forceinline__ unsigned cvt_pk(float a, float b)      // v_cvt_pk_bf16_f32: RNE, a -> low half
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// x - y as one v_sub_f32: written in C the compiler pairs the residuals into v_pk_add_f32, which issues several
// times slower than two v_sub_f32 here (measured on the split phase)
__device__ __forceinline__ float sub_f32(float x, float y)
{
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float lo_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f32(unsigned p) { return __uint_as_float(p & 0xFFFF0000u); }

// four consecutive-k fp32 values -> three planes of four bf16 (two dwords each)
__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, uint2& ph, uint2& pm, uint2& pl)
{
    ph.x = cvt_pk(x0, x1); ph.y = cvt_pk(x2, x3);
    const float r0 = x0 - lo_f32(ph.x), r1 = x1 - hi_f32(ph.x), r2 = x2 - lo_f32(ph.y), r3 = x3 - hi_f32(ph.y);
    pm.x = cvt_pk(r0, r1); pm.y = cvt_pk(r2, r3);
    const float s0 = r0 - lo_f32(pm.x), s1 = r1 - hi_f32(pm.x), s2 = r2 - lo_f32(pm.y), s3 = r3 - hi_f32(pm.y);
    pl.x = cvt_pk(s0, s1); pl.y = cvt_pk(s2, s3);
}

__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// stage one 128 x 32 operand tile global -> 4 float4 registers per thread
//   k-contiguous storage [x][k]: float4 = 4 k of one row, 8 lanes cover the 128-byte row segment
//   x-contiguous storage [k][x]: a 4(k) x 4(x) patch per thread, 8 lanes (same k rows) cover 128 bytes of x
template <bool XC>
__device__ __forceinline__ void s_load(float4 (&r)[4], const float* __restrict__ P, int ld, int x0, int X, int k0, int K1, int tid)
{
    if (XC) {
        const int lane = tid & 63, kg = lane & 7, xg = (lane >> 3) + 8 * (tid >> 6);
        const int x = x0 + 4 * xg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + 4 * kg + e;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < K1 && x < X) v = *reinterpret_cast<const float4*>(P + (size_t)k * ld + x);
            r[e] = v;
        }
    } else {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            const int f = tid + 256 * rep, x = x0 + (f >> 3), k = k0 + ((f & 7) << 2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x < X && k < K1) v = *reinterpret_cast<const float4*>(P + (size_t)x * ld + k);
            r[rep] = v;
        }
    }
}

// Fast staging path (shapes whose tiles need no per-element predicate, see gemm_f32s()): raw buffer loads with
// per-thread byte offsets computed once and a uniform (SGPR) base that advances per K tile; rows beyond the valid
// region read as zero through the buffer's num_records, so the loop carries no exec-mask branch and no 64-bit
// vector address arithmetic (which cost more vector issue slots than the split itself in the generic path).
template <bool XC>
struct FastSrc {
    const float* base;      // tile origin at the first K tile of this slice
    long long step;         // elements per K tile
    long long valid;        // elements from base to the end of the valid region
    unsigned voff[4];       // per-thread byte offsets of the four 16-byte pieces
    __device__ __forceinline__ void init(const float* P, int ld, int x0, int X, int kb, int ke, int tid)
    {
        if (XC) {           // [k][x]: rows k < ke are valid
            const int lane = tid & 63, kg = lane & 7, xg = (lane >> 3) + 8 * (tid >> 6);
            base = P + (size_t)kb * ld + x0; step = (long long)SBK * ld; valid = (long long)(ke - kb) * ld - x0;
#pragma unroll
            for (int e = 0; e < 4; ++e) voff[e] = (unsigned)(((4 * kg + e) * ld + 4 * xg) * 4);
        } else {            // [x][k]: rows x < X are valid, K is a multiple of the K tile
            base = P + (size_t)x0 * ld + kb; step = SBK; valid = (long long)(X - x0) * ld - kb;
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) { const int f = tid + 256 * rep; voff[rep] = (unsigned)(((f >> 3) * ld + ((f & 7) << 2)) * 4); }
        }
    }
    __device__ __forceinline__ void load(float4 (&r)[4], int it) const
    {
        const long long rem = valid - it * step;
        const unsigned bytes = rem <= 0 ? 0u : (rem >= (1ll << 30) ? 0xFFFFFFFFu : (unsigned)(rem * 4));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + it * step), 0, (int)bytes, 0x00020000);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[e], 0, 0);
            r[e] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    }
};

// split + store of one operand's four 16-byte pieces.  The four pieces are carried through the split stage by
// stage (all converts, then all residuals, ...) with scheduling barriers between the stages: written piece by
// piece the compiler reuses ten registers and every instruction waits for its predecessor (measured: the
// dependent chain cost ~14 cycles per instruction under the other waves' MFMA traffic, 3x the whole MFMA phase).
template <bool XC, int ABL = 0, int NP = 3>     // NP = 1: the bf16-operand mode (compute_dtype 1) keeps the hi plane only -- RNE to bf16, one product
__device__ __forceinline__ void s_split_store(unsigned short* __restrict__ s, const float4 (&r)[4], int tid)
{
    float v[4][4];          // v[j][i]: piece j, i-th of its four consecutive k
    int off[4];             // LDS element offset of piece j
    if (XC) {
        const int lane = tid & 63, kg = lane & 7, xg = (lane >> 3) + 8 * (tid >> 6);
        // r[e] component xi = value at (k = 4 kg + e, x = 4 xg + xi): piece j = row 4 xg + j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[j][i] = comp(r[i], j);
            off[j] = sw_off(4 * xg + j, kg >> 1) + 4 * (kg & 1);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[j][i] = comp(r[j], i);
            off[j] = sw_off(f >> 3, (f & 7) >> 1) + 4 * (f & 1);
        }
    }
    uint2 ph[4], pm[4], pl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ph[j].x = cvt_pk(v[j][0], v[j][1]); ph[j].y = cvt_pk(v[j][2], v[j][3]); }
    if constexpr (NP == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<uint2*>(s + off[j]) = ph[j];
        return;
    } else {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j][0] = sub_f32(v[j][0], lo_f32(ph[j].x)); v[j][1] = sub_f32(v[j][1], hi_f32(ph[j].x));
            v[j][2] = sub_f32(v[j][2], lo_f32(ph[j].y)); v[j][3] = sub_f32(v[j][3], hi_f32(ph[j].y));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pm[j].x = cvt_pk(v[j][0], v[j][1]); pm[j].y = cvt_pk(v[j][2], v[j][3]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j][0] = sub_f32(v[j][0], lo_f32(pm[j].x)); v[j][1] = sub_f32(v[j][1], hi_f32(pm[j].x));
            v[j][2] = sub_f32(v[j][2], lo_f32(pm[j].y)); v[j][3] = sub_f32(v[j][3], hi_f32(pm[j].y));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pl[j].x = cvt_pk(v[j][0], v[j][1]); pl[j].y = cvt_pk(v[j][2], v[j][3]); }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned short* d = s + off[j];
        *reinterpret_cast<uint2*>(d) = ph[j];
        *reinterpret_cast<uint2*>(d + SPLANE) = pm[j];
        *reinterpret_cast<uint2*>(d + 2 * SPLANE) = pl[j];
    }
}

template <bool A_MC, bool B_NC, int ABL = 0, bool FAST = false, int NP = 3>
__global__ __launch_bounds__(256, NP == 1 ? 3 : 2) void gemm_f32s_kernel(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * NP * SPLANE];      // 48 KB: A hi|mid|lo, B hi|mid|lo (NP = 1: 16 KB)
    unsigned short* As = smem;
    unsigned short* Bs = smem + NP * SPLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    int M = g.M, K = g.K;
    if (g.dyn_kind == 1) M = min(M, *g.dyn);
    if (g.dyn_kind == 2) K = min(K, *g.dyn);

    // XCD-aware tile order (as gemm_f32): blocks b, b+8, .. share an XCD; each XCD gets a contiguous run of tiles
    // (runs cut over the EFFECTIVE tile count -- the device-side row count of a ragged batch -- so that no XCD idles)
    const int tiles_n = (g.N + 127) / 128;
    const int tiles_m = (M + 127) / 128;
    int bid = blockIdx.x;
    {
        const int nblk = tiles_m * tiles_n, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        if (slot >= q + (xcd < r ? 1 : 0)) return;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    // grouped order inside the run: 8 row panels x all column panels, rows fastest, so that the ~64 tiles an XCD
    // works on at one time form an 8 x 8 block: 16 operand panels per K step instead of 3 + all of B
    // (the exact-fp32 kernel needs a quarter of this kernel's operand bandwidth and gets away without)
    const int grp = bid / (8 * tiles_n), rem = bid - grp * 8 * tiles_n;
    const int gm = min(8, tiles_m - 8 * grp);
    const int tn = rem / gm, tm = 8 * grp + rem - tn * gm;
    const int m0 = tm * 128, n0 = tn * 128;
    if (m0 >= M) return;

    int kb = 0, ke = K;
    if (g.split_k > 1) {
        const int ktiles = (K + SBK - 1) / SBK, per = (ktiles + g.split_k - 1) / g.split_k;
        kb = blockIdx.z * per * SBK;
        ke = min(K, kb + per * SBK);
        if (kb >= ke) return;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[4];
    FastSrc<A_MC> fsa; FastSrc<B_NC> fsb;
    if (FAST) {
        fsa.init(g.A, g.lda, m0, M, kb, ke, tid); fsb.init(g.B, g.ldb, n0, g.N, kb, ke, tid);
        fsa.load(ra, 0); fsb.load(rb, 0);
    } else {
        s_load<A_MC>(ra, g.A, g.lda, m0, M, kb, ke, tid);
        s_load<B_NC>(rb, g.B, g.ldb, n0, g.N, kb, ke, tid);
    }

    // fragment (tile t, step s): row 64 w + 32 t + l31, chunk 2 s + h.  32 t flips neither swizzle term's low bits
    // beyond (r >> 2): rows r and r + 32 share r & 31, so (r >> 2) & 3 and (r >> 2) & 1 are equal for both tiles
    int offa[2], offb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { offa[s] = sw_off(64 * wm + l31, 2 * s + h); offb[s] = sw_off(64 * wn + l31, 2 * s + h); }

    // ABL 16: phase stamps (100 MHz s_memrealtime) summed into g.bias (as 5 x uint64) by wave 0 of every workgroup
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tp = 0;
#define AVAE_STAMP(i) do { if (ABL & 16) { unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); ph[i] += t_ - tp; tp = t_; } } while (0)
    if (ABL & 16) tp = __builtin_amdgcn_s_memrealtime();
    for (int k0 = kb; k0 < ke; k0 += SBK) {
        if (ABL & 16) { __builtin_amdgcn_s_waitcnt(0x0F70 & 0xC07F); }      // vmcnt(0) only (gfx9 encoding: vmcnt lo [3:0], hi [15:14])
        AVAE_STAMP(0);
        s_split_store<A_MC, ABL, NP>(As, ra, tid);
        s_split_store<B_NC, ABL, NP>(Bs, rb, tid);
        AVAE_STAMP(1);
        __syncthreads();
        AVAE_STAMP(2);
        if (k0 + SBK < ke) {
            if (FAST) {
                const int it = (k0 - kb) / SBK + 1;
                fsa.load(ra, it); fsb.load(rb, it);
            } else {
                s_load<A_MC>(ra, g.A, g.lda, m0, M, k0 + SBK, ke, tid);
                s_load<B_NC>(rb, g.B, g.ldb, n0, g.N, k0 + SBK, ke, tid);
            }
        }
        AVAE_STAMP(3);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[2][NP], b[2][NP];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    a[t][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + offa[s] + p * SPLANE + 32 * t * SLD));
                    b[t][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + offb[s] + p * SPLANE + 32 * t * SLD));
                }
            // smallest partial products first; consecutive MFMAs target different accumulators
#define AVAE_PROD(pa, pb)                                                                                     \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                     \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                 \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
            if constexpr (NP == 3) { AVAE_PROD(2, 0) AVAE_PROD(0, 2) AVAE_PROD(1, 1) AVAE_PROD(1, 0) AVAE_PROD(0, 1) }
            AVAE_PROD(0, 0)
#undef AVAE_PROD
        }
        __syncthreads();
        if (ABL & 16) asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[1][1][15]));     // the stamp waits for the last MFMA
        AVAE_STAMP(4);
    }
    if (ABL & 16) {
        if (tid == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(g.bias));
            for (int i = 0; i < 5; ++i) atomicAdd(o + i, ph[i]);
            atomicAdd(o + 5, (unsigned long long)((ke - kb) / SBK));
        }
    }
#undef AVAE_STAMP

    // epilogue.  C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool atomic = g.split_k > 1;
    const bool add_bias = g.bias != nullptr && (!atomic || blockIdx.z == 0) && !(ABL & 16);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + 64 * wn + 32 * j + l31;
        if (col >= g.N) continue;
        const float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= M) continue;
                const float v = g.alpha * acc[i][j][r] + bv;
                float* c = g.C + (size_t)row * g.ldc + col;
                if (atomic) atomicAdd(c, v);
                else if (g.accumulate) *c += v;
                else *c = v;
            }
        }
    }
}

// ---------------------------------------------------------------- wave-specialised form (fast-path shapes)
// One 768-thread workgroup per CU: waves 0-3 ("consumers", one per SIMD) only read fragments and issue MFMAs,
// waves 4-11 ("producers", two per SIMD) only load, split and store the NEXT K tile into the other half of a
// double-buffered LDS image (2 x 48 KB).  One s_barrier per K tile, persistent over the workgroup's tiles.  In the
// phase-structured kernel above each wave alternates between ~1900 cycles of vector work and 1536 cycles of MFMA
// and three workgroups per CU keep the matrix pipe ~60 % busy; here the MFMA wave never leaves its stream.  What
// was measured: the producers finish early and wait at the barrier, but the consumer's 48 MFMAs
// take ~2400 cycles instead of 1536 -- the producers' vector instructions on the same SIMD cost the MFMA stream
// their issue time -- so this form wins ~5 % on long-K shapes and loses on short K, where one workgroup per CU
// leaves a tile's 64 KB C store un-overlapped.

// the tiles one workgroup works through: virtual block ids blockIdx.x, + gridDim.x, ... through the XCD-aware grouped
// order; tiles beyond the device-side row count are skipped (both roles walk the same sequence)
struct TileIter {
    int vb, step, nblk, tiles_m, tiles_n, M, m0, n0; bool ok;
    __device__ __forceinline__ void seek()
    {
        for (ok = false; vb < nblk; vb += step) {
            const int q = nblk >> 3, r = nblk & 7, xcd = vb & 7, slot = vb >> 3;
            const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
            const int grp = bid / (8 * tiles_n), rem = bid - grp * 8 * tiles_n;
            const int gm = min(8, tiles_m - 8 * grp);
            const int tn = rem / gm, tm = 8 * grp + rem - tn * gm;
            m0 = tm * 128; n0 = tn * 128;
            if (m0 < M) { ok = true; return; }
        }
    }
    __device__ __forceinline__ void first(int vb0, int step_, int nblk_, int tm_, int tn_, int M_)
    {
        vb = vb0; step = step_; nblk = nblk_; tiles_m = tm_; tiles_n = tn_; M = M_; seek();
    }
    __device__ __forceinline__ void next() { vb += step; seek(); }
};

template <bool A_MC, bool B_NC, int NP = 3>
__global__ __launch_bounds__(768, 1) void gemm_f32s_ws_kernel(GemmArgs g, int nblk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short dsm[];      // 2 stages x (A hi|mid|lo, B hi|mid|lo)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;

    int M = g.M, K = g.K;
    if (g.dyn_kind == 1) M = min(M, *g.dyn);
    if (g.dyn_kind == 2) K = min(K, *g.dyn);
    const int tiles_n = (g.N + 127) / 128, tiles_m = (M + 127) / 128;       // effective row tiles (ragged batches)
    nblk = min(nblk, tiles_m * tiles_n);
    int kb = 0, ke = K;
    if (g.split_k > 1) {
        const int ktiles = (K + SBK - 1) / SBK, per = (ktiles + g.split_k - 1) / g.split_k;
        kb = blockIdx.z * per * SBK;
        ke = min(K, kb + per * SBK);
        if (kb >= ke) return;
    }
    const int nk = (ke - kb + SBK - 1) / SBK;
    // PERSISTENT over tiles (gridDim.x = min(tiles, CUs)): the K-tile pipeline runs across tile boundaries, so a
    // tile's first loads, split and dispatch hide behind the previous tile's MFMAs and only its C stores remain
    // un-overlapped (one workgroup per CU paid ~9 us per tile for them before)
    TileIter ti;
    ti.first(blockIdx.x, gridDim.x, nblk, tiles_m, tiles_n, M);
    if (!ti.ok) return;                                   // uniform over the workgroup

    if (producer) {
        // waves 4-7 stage the A operand, waves 8-11 the B operand (256 threads each, the staging map of the
        // phase-structured kernel): two producer waves per SIMD run their dependent split chains side by side
        const bool isb = wave >= 8;
        const int ptid = tid - (isb ? 512 : 256);
        float4 r[4];
        auto run = [&](auto& fs, auto xc, const float* P, int ld, bool is_a, unsigned short* base) __attribute__((always_inline)) {
            constexpr bool XC = decltype(xc)::value;
            int kt = 0;                                   // load cursor: (ti, kt) = the next K tile to load
            auto issue = [&]() __attribute__((always_inline)) {
                if (kt == 0) fs.init(P, ld, is_a ? ti.m0 : ti.n0, is_a ? M : g.N, kb, ke, ptid);
                fs.load(r, kt);
                if (++kt == nk) { kt = 0; ti.next(); }
            };
            issue();
            s_split_store<XC, 0, NP>(base, r, ptid);
            bool nxt = ti.ok;
            if (nxt) issue();
            __syncthreads();
            for (int G = 0; ; ++G) {                      // consumers work on K tile G of this workgroup's sequence
                if (!nxt) { __syncthreads(); break; }
                s_split_store<XC, 0, NP>(base + ((G + 1) & 1) * 2 * NP * SPLANE, r, ptid);
                nxt = ti.ok;
                if (nxt) issue();
                __syncthreads();
            }
        };
        if (!isb) { FastSrc<A_MC> fs; run(fs, std::integral_constant<bool, A_MC>{}, g.A, g.lda, true, dsm); }
        else      { FastSrc<B_NC> fs; run(fs, std::integral_constant<bool, B_NC>{}, g.B, g.ldb, false, dsm + NP * SPLANE); }
        return;
    }

    // ---- consumers
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    int offa[2], offb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { offa[s] = sw_off(64 * wm + l31, 2 * s + h); offb[s] = NP * SPLANE + sw_off(64 * wn + l31, 2 * s + h); }
    const bool atomic = g.split_k > 1;
    const bool add_bias = g.bias != nullptr && (!atomic || blockIdx.z == 0);
    __syncthreads();
    for (int G = 0; ti.ok; ti.next()) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < nk; ++it, ++G) {
            const unsigned short* st = dsm + (G & 1) * 2 * NP * SPLANE;
            bf16x8 a[2][2][NP], b[2][2][NP];          // [step][tile][plane]
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        a[s][t][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + offa[s] + p * SPLANE + 32 * t * SLD));
                        b[s][t][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + offb[s] + p * SPLANE + 32 * t * SLD));
                    }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#define AVAE_PROD(pa, pb)                                                                                     \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i][pa], b[s][j][pb], acc[i][j], 0, 0, 0);
                if constexpr (NP == 3) { AVAE_PROD(2, 0) AVAE_PROD(0, 2) AVAE_PROD(1, 1) AVAE_PROD(1, 0) AVAE_PROD(0, 1) }
                AVAE_PROD(0, 0)
#undef AVAE_PROD
            }
            __syncthreads();
        }
        const int m0 = ti.m0, n0 = ti.n0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + l31;
            if (col >= g.N) continue;
            const float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row >= M) continue;
                    const float v = g.alpha * acc[i][j][r] + bv;
                    float* c = g.C + (size_t)row * g.ldc + col;
                    if (atomic) atomicAdd(c, v);
                    else if (g.accumulate) *c += v;
                    else *c = v;
                }
            }
        }
    }
}

template <bool A_MC, bool B_NC, int NP>
static hipError_t launch_ws(hipStream_t st, dim3 grid, const GemmArgs& g)
{
    constexpr int lds_bytes = 2 * 2 * NP * SPLANE * 2;
    static bool attr_set = false;
    static int ncu = 256;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32s_ws_kernel<A_MC, B_NC, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
        attr_set = true;
    }
    const int tiles = (int)grid.x;
    // one workgroup per CU walks its tiles; with split-K every (tile, slice) keeps its own workgroup
    if (grid.z == 1 && tiles > ncu) grid.x = ncu / 8 * 8;
    hipLaunchKernelGGL((gemm_f32s_ws_kernel<A_MC, B_NC, NP>), grid, dim3(768), lds_bytes, st, g, tiles);
    return hipGetLastError();
}

template <int NP>
static hipError_t gemm_split(hipStream_t st, bool a_mc, bool b_nc, const GemmArgs& g)
{
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if ((g.lda | g.ldb) & 3) return hipErrorInvalidValue;
    if (((uintptr_t)g.A | (uintptr_t)g.B) & 15) return hipErrorInvalidValue;
    if (!a_mc && (g.K & 3)) return hipErrorInvalidValue;
    if (a_mc && (g.M & 3)) return hipErrorInvalidValue;
    if (!b_nc && (g.K & 3)) return hipErrorInvalidValue;
    if (b_nc && (g.N & 3)) return hipErrorInvalidValue;
    const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
    dim3 grid(tiles, 1, g.split_k > 1 ? g.split_k : 1);
    const GemmArgs& gp = g;
    static const int abl = getenv("AVAE_F32S_ABLATE") ? atoi(getenv("AVAE_F32S_ABLATE")) : 0;     // diagnostics only
    if constexpr (NP == 3) {
    if (abl && !a_mc && !b_nc) {         // 16 / 17: phase stamps of the predicated / fast-staging main loop (scripts/gemm_stamps.py)
        if (abl == 16)      hipLaunchKernelGGL((gemm_f32s_kernel<false, false, 16>), grid, dim3(256), 0, st, gp);
        else if (abl == 17) hipLaunchKernelGGL((gemm_f32s_kernel<false, false, 16, true>), grid, dim3(256), 0, st, gp);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    }
    // fast staging path: no element of any tile needs a predicate the buffer bounds cannot express
    //   k-contiguous operand: K a multiple of the K tile and known on the host;  [k][x] operand: x extent a multiple of 128
    const bool fast = (a_mc ? (g.M % 128 == 0) : (g.K % SBK == 0 && g.dyn_kind != 2)) &&
                      (b_nc ? (g.N % 128 == 0) : (g.K % SBK == 0 && g.dyn_kind != 2)) && abl != 99;
    static const int ws = getenv("AVAE_F32S_WS") ? atoi(getenv("AVAE_F32S_WS")) : 1;      // 0: phase-structured kernel (A/B experiments)
    static const int ws1 = getenv("AVAE_BF16D_WS") ? atoi(getenv("AVAE_BF16D_WS")) : 0;   // the one-plane form: wave-specialised kernel off by default
    // one 768-thread workgroup per CU pays ~9 us per tile that nothing overlaps (first loads, C stores, dispatch):
    // it wins where a workgroup's K extent is long (measured cross-over between K = 1024 and 1536)
    const int k_per_wg = g.split_k > 1 ? (g.K + g.split_k - 1) / g.split_k : g.K;
    const int wsel = NP == 3 ? ws : ws1;
    if (fast && (wsel == 2 || (wsel == 1 && k_per_wg >= 1536))) {
        if (!a_mc && !b_nc)      return launch_ws<false, false, NP>(st, grid, gp);
        else if (!a_mc && b_nc)  return launch_ws<false, true, NP>(st, grid, gp);
        else if (a_mc && b_nc)   return launch_ws<true, true, NP>(st, grid, gp);
        else                     return launch_ws<true, false, NP>(st, grid, gp);
    }
    if (fast) {
        if (!a_mc && !b_nc)      hipLaunchKernelGGL((gemm_f32s_kernel<false, false, 0, true, NP>), grid, dim3(256), 0, st, gp);
        else if (!a_mc && b_nc)  hipLaunchKernelGGL((gemm_f32s_kernel<false, true, 0, true, NP>), grid, dim3(256), 0, st, gp);
        else if (a_mc && b_nc)   hipLaunchKernelGGL((gemm_f32s_kernel<true, true, 0, true, NP>), grid, dim3(256), 0, st, gp);
        else                     hipLaunchKernelGGL((gemm_f32s_kernel<true, false, 0, true, NP>), grid, dim3(256), 0, st, gp);
        return hipGetLastError();
    }
    if (!a_mc && !b_nc)      hipLaunchKernelGGL((gemm_f32s_kernel<false, false, 0, false, NP>), grid, dim3(256), 0, st, gp);
    else if (!a_mc && b_nc)  hipLaunchKernelGGL((gemm_f32s_kernel<false, true, 0, false, NP>), grid, dim3(256), 0, st, gp);
    else if (a_mc && b_nc)   hipLaunchKernelGGL((gemm_f32s_kernel<true, true, 0, false, NP>), grid, dim3(256), 0, st, gp);
    else                     hipLaunchKernelGGL((gemm_f32s_kernel<true, false, 0, false, NP>), grid, dim3(256), 0, st, gp);
    return hipGetLastError();
}

hipError_t gemm_f32s(hipStream_t st, bool a_mc, bool b_nc, const GemmArgs& g) { return gemm_split<3>(st, a_mc, b_nc, g); }
// bf16-operand GEMM straight from fp32 operands in any layout: each element rounded to bf16 (RNE) on its way into LDS, ONE
// product on v_mfma_f32_32x32x16_bf16, fp32 accumulate -- the values of cvt_bf16 + gemm_bf16_nt without the conversion passes
hipError_t gemm_bf16_direct(hipStream_t st, bool a_mc, bool b_nc, const GemmArgs& g) { return gemm_split<1>(st, a_mc, b_nc, g); }

}  // namespace avae
