// gemm_bf16_p8.hip -- the phased bf16 NT GEMM of compute_dtype 1 (BASELINE configs[2]): C[M,N] = alpha * A[M,K] * B[N,K]^T
// (+ bias, + C), both operands k-contiguous bf16, fp32 accumulate on v_mfma_f32_16x16x32_bf16.
//
// Why a second 256x256 kernel: gemm_bf16_nt256_kernel (gemm_bf16.hip) stages through registers behind ONE barrier per K
// tile, all eight waves in step -- the matrix pipe idles while they read LDS, write LDS and wait for the loads (0.9
// PFLOP/s at best, 0.55-0.75 on the shapes of the step).  Here
//   * the operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), four
//     128-row x 64-k pieces ("half tiles", 16 KB) per K tile into a two-tile ring (128 KB), each piece issued FOUR phases
//     before its first read and retired by a COUNTED s_waitcnt vmcnt(8) -- the queue is never drained inside the K loop;
//   * a K tile is four phases {ds_read a register sub-tile | issue one half tile || barrier | 16 MFMAs | barrier}, and
//     the two wave groups (the M halves of the tile) run one barrier apart: while one group's 16 MFMAs run, the other
//     group reads its fragments and issues its DMA pieces -- every SIMD holds one wave of each group;
//   * the LDS image of a half tile is lane-linear (a DMA piece of one wave = 8 rows x 128 B), so the bank swizzle sits on
//     the per-lane SOURCE address: 16-byte chunk c of row r is stored at chunk position c ^ (r & 7), which makes the
//     fragment reads (16 rows x 4 chunks per wave-instruction) conflict free;
//   * the MFMA takes B's fragment as its first operand: a lane then holds four CONSECUTIVE columns of one row of C and
//     the epilogue stores 16 bytes per lane.
//
// Half tiles of K tile t, ring slot (t & 1): AE = the 64 + 64 rows the two M groups read in phase 1, BE = the 4 x 32
// columns read in phase 1 (kept in registers for phase 4), BL = the other 4 x 32 columns (phase 2), AL = the other
// 64 + 64 rows (phase 3).  A slot piece is re-filled two or more phases after its last read:
//   phase 1 of tile t issues BL(t+1), phase 2 AL(t+1), phase 3 AE(t+2), phase 4 BE(t+2);
// phases 4, 1 and 2 wait vmcnt(8) after their issue (= everything but the last four half tiles has landed: the pieces the
// NEXT phase reads), every wave waits before the phase's first barrier, and the read follows one phase later -- the order
// LDS-DMA data needs (cdna_hip_programming.md section 5, "Read a staged buffer one phase AFTER the wait that retires it").
// Beyond the last K tile the issue slots re-load the last tile (into pieces nobody reads any more): the counts stay
// uniform.  K must be a multiple of 64 and N of 4; rows beyond M / columns beyond N are clamped on the load side and
// masked on the store side.  Everything else (K tails, tiny shapes) stays on gemm_bf16.hip's kernels.
#include <algorithm>
#include "kernels.h"

namespace avae {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kP8Tile = 256, kP8K = 64, kP8Piece = 16384, kP8Lds = 8 * kP8Piece, kP8Group = 8;
enum { AE = 0, BE = 1, BL = 2, AL = 3 };

struct P8Args {
    const unsigned short* A; const unsigned short* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    float alpha; int accumulate, split_k; const int* dyn; int dyn_kind;
};

__device__ __forceinline__ void dma16(const unsigned short* g, unsigned char* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ bf16x8 frag(const unsigned char* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p)); }

#define P8_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

__global__ __launch_bounds__(512) void gemm_bf16_p8_kernel(P8Args g)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char L[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    int M = g.M;
    if (g.dyn_kind == 1) M = min(M, *g.dyn);
    const int tiles_n = (g.N + kP8Tile - 1) / kP8Tile;
    int bid = blockIdx.x;
    {
        const int nblk = ((M + kP8Tile - 1) / kP8Tile) * tiles_n;       // effective tiles (device-side row count), <= gridDim.x
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        if (slot >= q + (xcd < r ? 1 : 0)) return;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    // tiles in groups of kP8Group tile rows, column by column inside a group: the ~32 tiles an XCD works on at a time are an
    // 8 x 4 block (12 operand panels through its L2 per K tile) instead of one tile row (33)
    int tm, tn;
    {
        const int tiles_m = (M + kP8Tile - 1) / kP8Tile, per = kP8Group * tiles_n, grp = bid / per, in = bid - grp * per;
        const int rows = min(kP8Group, tiles_m - grp * kP8Group);
        tn = in / rows; tm = grp * kP8Group + in - tn * rows;
    }
    const int m0 = tm * kP8Tile, n0 = tn * kP8Tile;
    int kt0 = 0, nkt = g.K / kP8K;
    if (g.split_k > 1) {
        const int per = (nkt + g.split_k - 1) / g.split_k;
        kt0 = blockIdx.z * per; nkt = min(nkt, kt0 + per) - kt0;
        if (nkt <= 0) return;
    }

    // ---- DMA sources: instruction j of a half tile covers its rows j*64 + wave*8 + (lane >> 3); this lane moves chunk
    // (lane & 7) ^ (row & 7) of its row into chunk position lane & 7
    const unsigned short* src[4][2];
    {
        const int chunk = ((lane & 7) ^ (lane >> 3)) * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int R = j * 64 + wave * 8 + (lane >> 3);
            const int ra = m0 + (R >> 6) * 128 + (R & 63), cb = n0 + (R >> 5) * 64 + (R & 31);
            src[AE][j] = g.A + (size_t)min(ra, M - 1) * g.lda + (size_t)kt0 * kP8K + chunk;
            src[AL][j] = g.A + (size_t)min(ra + 64, M - 1) * g.lda + (size_t)kt0 * kP8K + chunk;
            src[BE][j] = g.B + (size_t)min(cb, g.N - 1) * g.ldb + (size_t)kt0 * kP8K + chunk;
            src[BL][j] = g.B + (size_t)min(cb + 32, g.N - 1) * g.ldb + (size_t)kt0 * kP8K + chunk;
        }
    }
    unsigned char* const dst = L + wave * 1024;          // + slot * 65536 + piece * 16384 + j * 8192
    const int last = nkt - 1;
#define P8_STAGE(piece, t) do { const int t_ = (t); const int kt_ = min(t_, last) * kP8K; unsigned char* d_ = dst + (t_ & 1) * 65536 + (piece) * kP8Piece; \
        dma16(src[piece][0] + kt_, d_); dma16(src[piece][1] + kt_, d_ + 8192); } while (0)

    // ---- fragment addresses: row fr of a 16-row group, chunk (ks * 4 + fq) ^ (fr & 7); ks = 1 flips byte bit 6
    const int offA = (wr * 64 + fr) * 128 + ((fq ^ (fr & 7)) << 4);
    const int offB = (wc * 32 + fr) * 128 + ((fq ^ (fr & 7)) << 4);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4][2], b0[2][2], b1[2][2];

    // ---- prologue: tile 0 whole, tile 1's AE and BE; AE(0), BE(0) landed before the first read
    P8_STAGE(AE, 0); P8_STAGE(BE, 0); P8_STAGE(BL, 0); P8_STAGE(AL, 0); P8_STAGE(AE, 1); P8_STAGE(BE, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    P8_BARRIER();
    if (wr == 1) P8_BARRIER();                           // the second M group runs one barrier behind the first

#define P8_MFMA(AI, BJ, BV) do { __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) \
            acc[(AI) + mt][(BJ) + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BV[nt][ks], a[mt][ks], acc[(AI) + mt][(BJ) + nt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0); } while (0)

    for (int t = 0; t < nkt; ++t) {
        const unsigned char* const S = L + (t & 1) * 65536;
        // phase 1: AE, BE -> registers; issue BL(t+1)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b0[nt][ks] = frag(S + BE * kP8Piece + ((offB + nt * 2048) ^ (ks * 64)));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a[mt][ks] = frag(S + AE * kP8Piece + ((offA + mt * 2048) ^ (ks * 64)));
        P8_STAGE(BL, t + 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        P8_BARRIER();
        P8_MFMA(0, 0, b0);
        P8_BARRIER();
        // phase 2: BL -> registers; issue AL(t+1)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b1[nt][ks] = frag(S + BL * kP8Piece + ((offB + nt * 2048) ^ (ks * 64)));
        P8_STAGE(AL, t + 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        P8_BARRIER();
        P8_MFMA(0, 2, b1);
        P8_BARRIER();
        // phase 3: AL -> registers; issue AE(t+2)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a[mt][ks] = frag(S + AL * kP8Piece + ((offA + mt * 2048) ^ (ks * 64)));
        P8_STAGE(AE, t + 2);
        P8_BARRIER();
        P8_MFMA(4, 2, b1);
        P8_BARRIER();
        // phase 4: no reads (BE is still in registers); issue BE(t+2)
        P8_STAGE(BE, t + 2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        P8_BARRIER();
        P8_MFMA(4, 0, b0);
        P8_BARRIER();
    }
    if (wr == 0) P8_BARRIER();                           // (every wave passes the same number of barriers)
#undef P8_STAGE
#undef P8_MFMA

    // ---- epilogue: lane (fr, fq) of tile (mt, nt) holds C[row fr][columns 4 fq .. 4 fq + 3]
    const bool atomic = g.split_k > 1;
    const bool add_bias = g.bias != nullptr && (!atomic || blockIdx.z == 0);
    const int rbase = m0 + wr * 128 + fr, cbase = n0 + wc * 64 + fq * 4;
    const bool inside = m0 + kP8Tile <= M && n0 + kP8Tile <= g.N;
    if (inside && !atomic && !g.accumulate) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = cbase + (j >> 1) * 32 + (j & 1) * 16;
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (add_bias) bv = *reinterpret_cast<const f32x4*>(g.bias + col);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = rbase + (i >> 2) * 64 + (i & 3) * 16;
                *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + col) = g.alpha * acc[i][j] + bv;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = cbase + (j >> 1) * 32 + (j & 1) * 16;
            if (col >= g.N) continue;                    // (N is a multiple of 4: a lane's four columns are in or out together)
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (add_bias) bv = *reinterpret_cast<const f32x4*>(g.bias + col);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = rbase + (i >> 2) * 64 + (i & 3) * 16;
                if (row >= M) continue;
                float* c = g.C + (size_t)row * g.ldc + col;
                const f32x4 v = g.alpha * acc[i][j] + bv;
                if (atomic) { atomicAdd(c, v[0]); atomicAdd(c + 1, v[1]); atomicAdd(c + 2, v[2]); atomicAdd(c + 3, v[3]); }
                else if (g.accumulate) *reinterpret_cast<f32x4*>(c) = *reinterpret_cast<const f32x4*>(c) + v;
                else *reinterpret_cast<f32x4*>(c) = v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the re-loads of the last tile must have landed before the LDS is given back)
}

bool gemm_bf16_p8_ok(const GemmArgs& g, int lda, int ldb)
{
    return g.nt8 && g.K % kP8K == 0 && g.K >= 2 * kP8K && (g.N & 3) == 0 && (g.ldc & 3) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && g.dyn_kind != 2 &&
           (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 && (g.bias == nullptr || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0);
}

// s2: the K split (>= 1) as gemm_bf16_nt derived it; the caller's g.split_k > 1 says C holds the value to add onto
hipError_t gemm_bf16_p8(hipStream_t st, const unsigned short* A, int lda, const unsigned short* B, int ldb, const GemmArgs& g, int s2)
{
    P8Args a{A, B, g.C, g.bias, g.M, g.N, g.K, lda, ldb, g.ldc, g.alpha, g.accumulate, s2, g.dyn, g.dyn_kind};
    if (g.split_k > 1 && s2 == 1) a.accumulate = 1;       // the caller's slices were going to ADD into C
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_p8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kP8Lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = ((g.M + kP8Tile - 1) / kP8Tile) * ((g.N + kP8Tile - 1) / kP8Tile);
    hipLaunchKernelGGL(gemm_bf16_p8_kernel, dim3(tiles, 1, s2), dim3(512), kP8Lds, st, a);
    return hipGetLastError();
}

}  // namespace avae
