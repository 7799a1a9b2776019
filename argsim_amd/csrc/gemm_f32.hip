// gemm_f32.hip -- exact-fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the cuBLAS calls under tf.layers.dense / tf.tensordot / CudnnGRU input projections
// of the reference graph (src/model.py:120-121,149-156,160-168) and their autodiff duals.
//
// Block tile 128x128, BK = 32, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles.
// Global -> registers -> LDS staging with the next K-tile's loads in flight during the MFMAs;
// 2-3 workgroups per CU overlap each other's staging phases.
//
// MFMA operand map (lane l, h = l>>5):  A[i = l&31][k = h],  B[k = h][j = l&31].
// Within a 32-deep K tile, MFMA step j (0..15) of lane-half h consumes k = 8*(j>>2) + 4*h + (j&3):
// a k-contiguous operand row then feeds four consecutive steps from ONE ds_read_b128.
#include "kernels.h"
#include <cstdlib>

namespace avae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDK = BK + 4;     // k-contiguous tile row stride (floats): conflict-free b128 reads

// stage one operand tile (ROWS x 32) global -> ROWS/32 float4 registers per thread
template <bool XC, int ROWS>   // XC: x(m or n)-contiguous storage [k][x];  else k-contiguous [x][k]
__device__ __forceinline__ void load_tile(float4 (&r)[ROWS / 32], const float* __restrict__ P, int ld,
                                          int x0, int X, int k0, int K1, int tid)
{
#pragma unroll
    for (int rep = 0; rep < ROWS / 32; ++rep) {
        int f = tid + 256 * rep;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (XC) {
            int k = k0 + f / (ROWS / 4), x = x0 + ((f % (ROWS / 4)) << 2);
            if (k < K1 && x < X) v = *reinterpret_cast<const float4*>(P + (size_t)k * ld + x);
        } else {
            int x = x0 + (f >> 3), k = k0 + ((f & 7) << 2);
            if (x < X && k < K1) v = *reinterpret_cast<const float4*>(P + (size_t)x * ld + k);
        }
        r[rep] = v;
    }
}

template <bool XC, int ROWS>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const float4 (&r)[ROWS / 32], int tid)
{
#pragma unroll
    for (int rep = 0; rep < ROWS / 32; ++rep) {
        int f = tid + 256 * rep;
        if (XC) *reinterpret_cast<float4*>(s + (f / (ROWS / 4)) * ROWS + ((f % (ROWS / 4)) << 2)) = r[rep];
        else    *reinterpret_cast<float4*>(s + (f >> 3) * LDK + ((f & 7) << 2)) = r[rep];
    }
}

// Fast staging (shapes whose tiles need no per-element predicate, see gemm_f32()): raw buffer loads with per-thread
// byte offsets computed once and a uniform (SGPR) base that advances per K tile; rows beyond the valid region read
// as zero through the buffer's num_records.  The predicated path above spends ~650 vector instructions per wave
// and K tile on exec-mask branches and 64-bit address arithmetic; this one ~20.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool XC, int ROWS>
struct FastTile {
    const float* base; long long step, valid; unsigned voff[ROWS / 32];
    __device__ __forceinline__ void init(const float* P, int ld, int x0, int X, int kb, int ke, int tid)
    {
        if (XC) {           // [k][x]: rows k < ke valid; x extent a multiple of ROWS
            base = P + (size_t)kb * ld + x0; step = (long long)BK * ld; valid = (long long)(ke - kb) * ld - x0;
#pragma unroll
            for (int rep = 0; rep < ROWS / 32; ++rep) { const int f = tid + 256 * rep; voff[rep] = (unsigned)(((f / (ROWS / 4)) * ld + ((f % (ROWS / 4)) << 2)) * 4); }
        } else {            // [x][k]: rows x < X valid; K a multiple of BK
            base = P + (size_t)x0 * ld + kb; step = BK; valid = (long long)(X - x0) * ld - kb;
#pragma unroll
            for (int rep = 0; rep < ROWS / 32; ++rep) { const int f = tid + 256 * rep; voff[rep] = (unsigned)(((f >> 3) * ld + ((f & 7) << 2)) * 4); }
        }
    }
    __device__ __forceinline__ void load(float4 (&r)[ROWS / 32], int it) const
    {
        const long long rem = valid - it * step;
        const unsigned bytes = rem <= 0 ? 0u : (rem >= (1ll << 30) ? 0xFFFFFFFFu : (unsigned)(rem * 4));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + it * step), 0, (int)bytes, 0x00020000);
#pragma unroll
        for (int rep = 0; rep < ROWS / 32; ++rep) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[rep], 0, 0);
            r[rep] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    }
};

// Epilogue of a (32 WM TM) x (32 WN TN) block tile.  C/D map of a 32x32 MFMA tile: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  MODE 0 store, 1 accumulate (+=), 2 float atomics (split-K).  A tile that lies
// wholly inside the matrix takes the straight-line form (one address computation per 32x32 tile, no per-element
// predicate or mode branch): the branchy generic form below costs ~40 instructions per element, which was ~15 % of
// a K = 512 output tile.
template <int MODE>
__device__ __forceinline__ void put(float* c, float v)
{
    if (MODE == 2) atomicAdd(c, v);
    else if (MODE == 1) *c += v;
    else *c = v;
}
template <int TM, int TN, int WN, int MODE>
__device__ __forceinline__ void epilogue_full(const GemmArgs& g, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int h, int l31, bool add_bias)
{
    // (the bias values are fetched before the first store: a load issued between the stores makes hipcc wait for it
    //  with vmcnt(0), i.e. for every store before it -- one exposed store round trip per 32x32 tile)
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = add_bias ? g.bias[n0 + 32 * (wn * TN + j) + l31] : 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + 32 * (wn * TN + j) + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float* c = g.C + (size_t)(m0 + 32 * (wm * TM + i) + 4 * h) * g.ldc + col;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                put<MODE>(c + (size_t)((r & 3) + 8 * (r >> 2)) * g.ldc, g.alpha * acc[i][j][r] + bv[j]);
        }
    }
}
template <int TM, int TN, int WN>
__device__ __forceinline__ void epilogue(const GemmArgs& g, const f32x16 (&acc)[TM][TN], int M, int m0, int n0, int BM, int BN,
                                         int wm, int wn, int h, int l31, bool atomic, bool add_bias)
{
    // (split-K float atomics keep the generic form: 64 atomics per lane issued back to back ran 25 % SLOWER than the
    //  same atomics spaced by the generic form's bookkeeping -- the memory-side atomic rate is the limit there)
    if (!atomic && m0 + BM <= M && n0 + BN <= g.N) {            // uniform
        if (g.accumulate) epilogue_full<TM, TN, WN, 1>(g, acc, m0, n0, wm, wn, h, l31, add_bias);
        else epilogue_full<TM, TN, WN, 0>(g, acc, m0, n0, wm, wn, h, l31, add_bias);
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int col = n0 + 32 * (wn * TN + j) + l31;
        if (col >= g.N) continue;
        float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + 32 * (wm * TM + i) + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= M) continue;
                float v = g.alpha * acc[i][j][r] + bv;
                float* c = g.C + (size_t)row * g.ldc + col;
                if (atomic) atomicAdd(c, v);
                else if (g.accumulate) *c += v;
                else *c = v;
            }
        }
    }
}

// WM x WN waves (WM*WN = 4), each TM x TN MFMA tiles of 32x32: block tile BM = 32 WM TM, BN = 32 WN TN.
// <2,2,2,2> = 128x128 is the workhorse; <1,4,1,1> = 32x128 serves thin row panels (M <= 512 and the
// remainder rows of a tile count just above a multiple of 256) deterministically, without split-K.
// (register budget pinned to the residency the launch shaping assumes: 3 workgroups per CU, 2 for the double-buffered form)
template <bool A_MC, bool B_NC, int WM, int WN, int TM, int TN, bool FAST = false, bool DB = false>
__global__ __launch_bounds__(256, DB ? 2 : 3) void gemm_f32_kernel(GemmArgs g)
{
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int kStage = (BM + BN) * LDK;
    __shared__ __attribute__((aligned(16))) float smem[kStage * (DB ? 2 : 1)];
    float* As = smem;
    float* Bs = smem + BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;

    if (blockIdx.y) { g.A = g.A2; g.B = g.B2; g.C = g.C2; g.bias = g.bias2; }      // second problem of a pair (kernels.h)
    int M = g.M, K = g.K;
    if (g.dyn_kind == 1) M = min(M, *g.dyn);
    if (g.dyn_kind == 2) K = min(K, *g.dyn);

    // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch), give each
    // XCD a contiguous run of tiles; tiles advance along N fastest so a run shares A panels.
    // The runs are cut over the EFFECTIVE tile count (device-side row count of a ragged batch): cut over the grid,
    // the XCDs whose runs lie beyond the real rows would idle while the others carry the whole GEMM.
    const int tiles_n = (g.N + BN - 1) / BN;
    const int nblk = ((M + BM - 1) / BM) * tiles_n;          // <= gridDim.x
    int bid = blockIdx.x;
    {
        int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        if (slot >= q + (xcd < r ? 1 : 0)) return;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;

    // K range of this split
    int kb = 0, ke = K;
    if (g.split_k > 1) {
        int ktiles = (K + BK - 1) / BK;
        int per = (ktiles + g.split_k - 1) / g.split_k;
        kb = blockIdx.z * per * BK;
        ke = min(K, kb + per * BK);
        if (kb >= ke) return;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[BM / 32], rb[BN / 32];
    FastTile<A_MC, BM> fa; FastTile<B_NC, BN> fb;
    if (FAST) {
        fa.init(g.A, g.lda, m0, M, kb, ke, tid); fb.init(g.B, g.ldb, n0, g.N, kb, ke, tid);
        fa.load(ra, 0); fb.load(rb, 0);
    } else {
        load_tile<A_MC, BM>(ra, g.A, g.lda, m0, M, kb, ke, tid);
        load_tile<B_NC, BN>(rb, g.B, g.ldb, n0, g.N, kb, ke, tid);
    }

    if (DB) {
        store_tile<A_MC, BM>(As, ra, tid);
        store_tile<B_NC, BN>(Bs, rb, tid);
        __syncthreads();
    }
    for (int k0 = kb; k0 < ke; k0 += BK) {
        // The next K tile's loads go out BEFORE the barrier that publishes this one (the staging registers are free as soon as the LDS
        // writes have been issued): the barrier wait is time the loads are already in flight -- the load latency is what a K = 512 tile
        // of the persistent form was exposed to (ablation, gpurun_out/s2_f32_abl.log: no loads +9 %; early issue +2..6 % on the NT shapes).
        auto next_loads = [&]() __attribute__((always_inline)) {
            if (k0 + BK < ke) {
                if (FAST) { const int it = (k0 - kb) / BK + 1; fa.load(ra, it); fb.load(rb, it); }
                else {
                    load_tile<A_MC, BM>(ra, g.A, g.lda, m0, M, k0 + BK, ke, tid);
                    load_tile<B_NC, BN>(rb, g.B, g.ldb, n0, g.N, k0 + BK, ke, tid);
                }
            }
        };
        if (!DB) {
            store_tile<A_MC, BM>(As, ra, tid);
            store_tile<B_NC, BN>(Bs, rb, tid);
            next_loads();
            __syncthreads();
        } else next_loads();
        // the waves inside their MFMA burst issue ahead of the co-resident workgroups' staging phases (address arithmetic, LDS writes, the
        // loads' issue): +1-3 % on every shape of the step (gpurun_out/s2_f32_prio2.log); a static priority per workgroup measured nothing
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a[TM][4], b[TN][4];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if (A_MC) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[t][e] = As[(8 * q + 4 * h + e) * BM + 32 * (wm * TM + t) + l31];
                } else {
                    float4 v = *reinterpret_cast<const float4*>(As + (32 * (wm * TM + t) + l31) * LDK + 8 * q + 4 * h);
                    a[t][0] = v.x; a[t][1] = v.y; a[t][2] = v.z; a[t][3] = v.w;
                }
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (B_NC) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[t][e] = Bs[(8 * q + 4 * h + e) * BN + 32 * (wn * TN + t) + l31];
                } else {
                    float4 v = *reinterpret_cast<const float4*>(Bs + (32 * (wn * TN + t) + l31) * LDK + 8 * q + 4 * h);
                    b[t][0] = v.x; b[t][1] = v.y; b[t][2] = v.z; b[t][3] = v.w;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (DB) {
            // the other stage: its readers passed the barrier that ended the previous K tile
            As = (As == smem) ? smem + kStage : smem;
            Bs = As + BM * LDK;
            if (k0 + BK < ke) {
                store_tile<A_MC, BM>(As, ra, tid);
                store_tile<B_NC, BN>(Bs, rb, tid);
            }
        }
        __syncthreads();
    }

    const bool atomic = g.split_k > 1;
    const bool add_bias = g.bias != nullptr && (!atomic || blockIdx.z == 0);
    epilogue<TM, TN, WN>(g, acc, M, m0, n0, BM, BN, wm, wn, h, l31, atomic, add_bias);
}

// Persistent form of the 128x128 kernel for GEMMs of more than one round of tiles (no split-K): 768 workgroups
// (3 per CU) walk the tiles of their XCD's run round-robin, and the first K tile of a workgroup's NEXT output tile is
// put in flight before the epilogue of the current one, so the global-load latency of a tile's prologue -- which all
// workgroups of a round otherwise pay together -- hides behind the last MFMAs and the C stores.  Measured on the
// step's short-K shapes (K = 512: 16 K tiles per output tile) the per-tile overhead was ~14 % of the tile.
template <bool A_MC, bool B_NC, bool FAST>
__global__ __launch_bounds__(256, 3) void gemm_f32_persist_kernel(GemmArgs g)
{
    static_assert(FAST, "buffer-load staging only (register budget of three workgroups per CU)");
    constexpr int TM = 2, TN = 2, WN = 2, BM = 128, BN = 128;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
    float* As = smem;
    float* Bs = smem + BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;

    int M = g.M;
    const int K = g.K;
    if (g.dyn_kind == 1) M = min(M, *g.dyn);
    const int tiles_n = (g.N + BN - 1) / BN;
    const int nblk = ((M + BM - 1) / BM) * tiles_n;
    // XCD-aware order (see gemm_f32_kernel): workgroups g, g + 8, ... share an XCD and walk its contiguous run of tiles
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int run0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, run_n = q + (xcd < r ? 1 : 0);
    int local = blockIdx.x >> 3;
    if (local >= run_n) return;

    float4 ra[BM / 32], rb[BN / 32];
    FastTile<A_MC, BM> fa; FastTile<B_NC, BN> fb;
    auto first_tile = [&](int bid) __attribute__((always_inline)) {          // loads K tile 0 of output tile `bid`
        const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
        if (FAST) {
            fa.init(g.A, g.lda, tm * BM, M, 0, K, tid); fb.init(g.B, g.ldb, tn * BN, g.N, 0, K, tid);
            fa.load(ra, 0); fb.load(rb, 0);
        } else {
            load_tile<A_MC, BM>(ra, g.A, g.lda, tm * BM, M, 0, K, tid);
            load_tile<B_NC, BN>(rb, g.B, g.ldb, tn * BN, g.N, 0, K, tid);
        }
    };
    first_tile(run0 + local);
    for (; local < run_n; local += per_xcd) {
        const int bid = run0 + local;
        const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
        const bool more = local + per_xcd < run_n;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int k0 = 0; k0 < K; k0 += BK) {
            store_tile<A_MC, BM>(As, ra, tid);
            store_tile<B_NC, BN>(Bs, rb, tid);
            // (the next loads ahead of the barrier: see gemm_f32_kernel)
            if (k0 + BK < K) {
                if (FAST) { const int it = k0 / BK + 1; fa.load(ra, it); fb.load(rb, it); }
                else {
                    load_tile<A_MC, BM>(ra, g.A, g.lda, m0, M, k0 + BK, K, tid);
                    load_tile<B_NC, BN>(rb, g.B, g.ldb, n0, g.N, k0 + BK, K, tid);
                }
            } else if (more) {
                first_tile(bid + per_xcd);                    // the next output tile's first K tile, behind these MFMAs
            }
            __syncthreads();
            __builtin_amdgcn_s_setprio(1);                    // (see gemm_f32_kernel)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float a[TM][4], b[TN][4];
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    if (A_MC) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[t][e] = As[(8 * qq + 4 * h + e) * BM + 32 * (wm * TM + t) + l31];
                    } else {
                        float4 v = *reinterpret_cast<const float4*>(As + (32 * (wm * TM + t) + l31) * LDK + 8 * qq + 4 * h);
                        a[t][0] = v.x; a[t][1] = v.y; a[t][2] = v.z; a[t][3] = v.w;
                    }
                }
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    if (B_NC) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) b[t][e] = Bs[(8 * qq + 4 * h + e) * BN + 32 * (wn * TN + t) + l31];
                    } else {
                        float4 v = *reinterpret_cast<const float4*>(Bs + (32 * (wn * TN + t) + l31) * LDK + 8 * qq + 4 * h);
                        b[t][0] = v.x; b[t][1] = v.y; b[t][2] = v.z; b[t][3] = v.w;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
        }
        epilogue<TM, TN, WN>(g, acc, M, m0, n0, BM, BN, wm, wn, h, l31, false, g.bias != nullptr);
    }
}

// Skinny problems (a few hundred rows: the latent block, the one-step top layer, the remainder rows of a big GEMM): the tiled
// kernel above spends a K tile's staging latency on a handful of MFMAs and runs at 4-40 TFLOP/s there.  Here a workgroup owns
// ONE 32x32 output tile and its 8 waves split K eight ways: every wave loads its operand slices straight into the MFMA
// register layout (no LDS staging, no barrier in the loop: the operands are a few hundred KB and live in L2), the eight partial
// tiles meet in LDS once and wave 0 adds them in a fixed order -- deterministic, no atomics.  A k-contiguous ([M][K]); B either
// k-contiguous ([N][K]) or [K][N].
template <bool B_NC>
__global__ __launch_bounds__(512) void gemm_f32_skinny_kernel(GemmArgs g)
{
    __shared__ float red[7][16][64];
    if (blockIdx.y) { g.A = g.A2; g.B = g.B2; g.C = g.C2; g.bias = g.bias2; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int tiles_n = (g.N + 31) / 32, tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 32, n0 = tn * 32;
    const int kchunk = ((g.K + 63) / 64) * 8;                       // ceil(K / 8) rounded up to the 8 k of one iteration
    const int k0 = wave * kchunk, k1 = min(g.K, k0 + kchunk);
    const float* __restrict__ ap = g.A + (size_t)min(m0 + l31, g.M - 1) * g.lda;
    const int bcol = min(n0 + l31, g.N - 1);
    const float* __restrict__ bp = B_NC ? g.B + bcol : g.B + (size_t)bcol * g.ldb;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // lane (l31, h) feeds k = kk + 4 h + {0..3} of an 8-wide step: A[row l31][k] and B[k][col l31] (any pairing of k indices is a
    // valid contraction order as long as both operands use the same one)
    // four 8-wide steps per pass: their eight operand loads are issued together (one L2 round trip per 32 k instead of one per 8),
    // the MFMAs follow in the k order of the plain loop -- the same sums, bit for bit
    for (int kk = k0; kk < k1; kk += 32) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kk + 8 * u + 4 * h;
            a[u] = make_float4(0.f, 0.f, 0.f, 0.f); b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < k1) {                                           // (K and the chunks are multiples of 4: a whole float4 or nothing)
                a[u] = *reinterpret_cast<const float4*>(ap + k);
                if constexpr (B_NC) {
                    b[u].x = bp[(size_t)k * g.ldb]; b[u].y = bp[(size_t)(k + 1) * g.ldb]; b[u].z = bp[(size_t)(k + 2) * g.ldb]; b[u].w = bp[(size_t)(k + 3) * g.ldb];
                } else b[u] = *reinterpret_cast<const float4*>(bp + k);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (kk + 8 * u >= k1) break;                            // (uniform: a step wholly behind the chunk adds nothing -- skipped so that the sums keep the plain loop's order)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
    const int col = n0 + l31;
    if (col >= g.N) return;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r];
#pragma unroll
        for (int w = 0; w < 7; ++w) v += red[w][r][lane];           // fixed order: the same bits every run
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        float* c = g.C + (size_t)row * g.ldc + col;
        v = g.alpha * v + bv;
        if (g.accumulate) *c += v; else *c = v;
    }
}
static bool skinny_ok(bool a_mc, const GemmArgs& g)
{
    return !a_mc && g.split_k <= 1 && g.dyn_kind == 0 && (g.K & 3) == 0 && (g.lda & 3) == 0 && g.K >= 64;
}
static void launch_skinny(hipStream_t st, bool b_nc, const GemmArgs& g)
{
    dim3 grid(((g.M + 31) / 32) * ((g.N + 31) / 32), g.A2 ? 2 : 1);
    if (b_nc) hipLaunchKernelGGL(gemm_f32_skinny_kernel<true>, grid, dim3(512), 0, st, g);
    else      hipLaunchKernelGGL(gemm_f32_skinny_kernel<false>, grid, dim3(512), 0, st, g);
}

template <int WM, int WN, int TM, int TN>
static void launch_variant(hipStream_t st, bool a_mc, bool b_nc, const GemmArgs& g)
{
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    dim3 grid(tiles, g.A2 ? 2 : 1, g.split_k > 1 ? g.split_k : 1);
    // fast staging: k-contiguous operand: K a multiple of BK, known on the host; [k][x] operand: x extent a multiple of the tile
    static const bool allow_fast = !getenv("AVAE_F32_NOFAST");
    // (measured: +4..9 % where an operand is stored [k][x] or the K extent is long; the NT shapes with K <= 1024
    //  lose 1..7 % to the longer prologue and stay on the predicated path)
    const int k_per_wg = g.split_k > 1 ? (g.K + g.split_k - 1) / g.split_k : g.K;
    const bool fast = allow_fast && (a_mc ? (g.M % BM == 0) : (g.K % BK == 0 && g.dyn_kind != 2)) &&
                      (b_nc ? (g.N % BN == 0) : (g.K % BK == 0 && g.dyn_kind != 2)) && (a_mc || b_nc || k_per_wg >= 1536);
    // Two launch forms of the 128x128 kernel.  Default: single LDS stage, two barriers per K tile, 3 workgroups per CU
    // (768 slots).  Double-buffered: one barrier per K tile, 2 workgroups per CU (512 slots); measured 3-8 % slower at
    // equal balance, but a tile count in (768, 1024] runs as ONE balanced round instead of a full round plus a round
    // that leaves most slots empty (1024 tiles: 112 -> 130 TFLOP/s, gpurun_out/gb_db.log).
    // more than one round of tiles, no split-K: the persistent form (cross-tile prefetch)
    static const char* ps_env = getenv("AVAE_F32_PERSIST");
    if constexpr (WM == 2 && WN == 2) {
        // (only with the buffer-load staging: the predicated staging needs more registers than three workgroups per
        //  CU leave, and its longer prologue no longer matters once the prologue is prefetched)
        const bool aligned = (a_mc ? (g.M % BM == 0) : (g.K % BK == 0)) && (b_nc ? (g.N % BN == 0) : (g.K % BK == 0));
        const bool persist = (ps_env ? atoi(ps_env) != 0 : true) && allow_fast && aligned && !g.A2 && g.split_k <= 1 && g.dyn_kind != 2 && tiles > 1024;
        if (persist) {
            dim3 pg(768);
            if (!a_mc && !b_nc)      hipLaunchKernelGGL((gemm_f32_persist_kernel<false, false, true>), pg, dim3(256), 0, st, g);
            else if (!a_mc && b_nc)  hipLaunchKernelGGL((gemm_f32_persist_kernel<false, true, true>), pg, dim3(256), 0, st, g);
            else if (a_mc && b_nc)   hipLaunchKernelGGL((gemm_f32_persist_kernel<true, true, true>), pg, dim3(256), 0, st, g);
            else                     hipLaunchKernelGGL((gemm_f32_persist_kernel<true, false, true>), pg, dim3(256), 0, st, g);
            return;
        }
    }
    static const char* db_env = getenv("AVAE_F32_DB");
    if constexpr (WM == 2 && WN == 2) {
        // (a device-side row count: where the host expects more than one round of the 768 single-stage slots to be real)
        const int eff_tiles = (g.dyn_kind == 1 && g.dyn_expect > 0) ? std::min(tiles, ((g.dyn_expect + BM - 1) / BM) * ((g.N + BN - 1) / BN)) : (g.dyn_kind == 1 ? 0 : tiles);
        const bool use_db = db_env ? atoi(db_env) != 0 && !g.A2 : (!g.A2 && g.split_k <= 1 && eff_tiles > 768 && tiles <= 1024);
        if (use_db) {
            if (fast) {
                if (!a_mc && !b_nc)      hipLaunchKernelGGL((gemm_f32_kernel<false, false, WM, WN, TM, TN, true, true>), grid, dim3(256), 0, st, g);
                else if (!a_mc && b_nc)  hipLaunchKernelGGL((gemm_f32_kernel<false, true, WM, WN, TM, TN, true, true>), grid, dim3(256), 0, st, g);
                else if (a_mc && b_nc)   hipLaunchKernelGGL((gemm_f32_kernel<true, true, WM, WN, TM, TN, true, true>), grid, dim3(256), 0, st, g);
                else                     hipLaunchKernelGGL((gemm_f32_kernel<true, false, WM, WN, TM, TN, true, true>), grid, dim3(256), 0, st, g);
            } else {
                if (!a_mc && !b_nc)      hipLaunchKernelGGL((gemm_f32_kernel<false, false, WM, WN, TM, TN, false, true>), grid, dim3(256), 0, st, g);
                else if (!a_mc && b_nc)  hipLaunchKernelGGL((gemm_f32_kernel<false, true, WM, WN, TM, TN, false, true>), grid, dim3(256), 0, st, g);
                else if (a_mc && b_nc)   hipLaunchKernelGGL((gemm_f32_kernel<true, true, WM, WN, TM, TN, false, true>), grid, dim3(256), 0, st, g);
                else                     hipLaunchKernelGGL((gemm_f32_kernel<true, false, WM, WN, TM, TN, false, true>), grid, dim3(256), 0, st, g);
            }
            return;
        }
    }
    if (fast) {
        if (!a_mc && !b_nc)      hipLaunchKernelGGL((gemm_f32_kernel<false, false, WM, WN, TM, TN, true>), grid, dim3(256), 0, st, g);
        else if (!a_mc && b_nc)  hipLaunchKernelGGL((gemm_f32_kernel<false, true, WM, WN, TM, TN, true>), grid, dim3(256), 0, st, g);
        else if (a_mc && b_nc)   hipLaunchKernelGGL((gemm_f32_kernel<true, true, WM, WN, TM, TN, true>), grid, dim3(256), 0, st, g);
        else                     hipLaunchKernelGGL((gemm_f32_kernel<true, false, WM, WN, TM, TN, true>), grid, dim3(256), 0, st, g);
        return;
    }
    if (!a_mc && !b_nc)      hipLaunchKernelGGL((gemm_f32_kernel<false, false, WM, WN, TM, TN>), grid, dim3(256), 0, st, g);
    else if (!a_mc && b_nc)  hipLaunchKernelGGL((gemm_f32_kernel<false, true, WM, WN, TM, TN>), grid, dim3(256), 0, st, g);
    else if (a_mc && b_nc)   hipLaunchKernelGGL((gemm_f32_kernel<true, true, WM, WN, TM, TN>), grid, dim3(256), 0, st, g);
    else                     hipLaunchKernelGGL((gemm_f32_kernel<true, false, WM, WN, TM, TN>), grid, dim3(256), 0, st, g);
}

hipError_t gemm_f32(hipStream_t st, bool a_mc, bool b_nc, const GemmArgs& g)
{
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if ((g.lda | g.ldb) & 3) return hipErrorInvalidValue;
    if (((uintptr_t)g.A | (uintptr_t)g.B | (uintptr_t)g.A2 | (uintptr_t)g.B2) & 15) return hipErrorInvalidValue;
    // contiguous extents must be multiples of 4 (float4 staging)
    if (!a_mc && (g.K & 3)) return hipErrorInvalidValue;
    if (a_mc && (g.M & 3)) return hipErrorInvalidValue;
    if (!b_nc && (g.K & 3)) return hipErrorInvalidValue;
    if (b_nc && (g.N & 3)) return hipErrorInvalidValue;
    // (256x128 and 128x256 tiles were measured 5-20 % slower: 1 workgroup per CU cannot hide its own staging)
    if (g.thin == 3 && skinny_ok(a_mc, g) && (b_nc || (g.ldb & 3) == 0)) launch_skinny(st, b_nc, g);      // a few rows: one 32x32 tile per workgroup, K split over its waves
    else if (g.thin == 3) launch_variant<1, 4, 1, 1>(st, a_mc, b_nc, g);
    else if (g.thin == 2) launch_variant<2, 2, 1, 1>(st, a_mc, b_nc, g);      // 64x64 tiles: 4x the tiles of the 128x128 form -> a quarter of the split-K slices
    else if (g.thin) launch_variant<1, 4, 1, 1>(st, a_mc, b_nc, g);
    else        launch_variant<2, 2, 2, 2>(st, a_mc, b_nc, g);
    return hipGetLastError();
}

}  // namespace avae
