// gemm_bf16.hip -- bf16-operand GEMM on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, fp32 output) plus
// the conversion kernels that bring fp32 operands into k-contiguous bf16 panels.
//
// This is the opt-in precision mode of BASELINE.json configs[2] ("bf16 ... MFMA vocab-projection GEMM"):
// every dense contraction of the step (input projections, out affine, tied logits and all their
// gradients) rounds its two operands to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) and accumulates
// in fp32; parameters, activations, the recurrent state, the gate math, the losses and Adam stay fp32.
// The default mode is exact fp32 (gemm_f32.hip); nothing here runs unless compute_dtype = 1.
//
// One GEMM layout only: C[M,N] = alpha * A[M,K] * B[N,K]^T with both operands k-contiguous bf16.  fp32
// operands that are stored [k][x] (the NN / TN cases of the fp32 kernel) are transposed while they are
// converted, so the GEMM never needs a transposing LDS access.
#include <algorithm>
#include "kernels.h"

namespace avae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// ---------------------------------------------------------------- fp32 [rows][cols] -> bf16 [rows][ldd]
// ldd = cols rounded up to 8; the pad columns are written as zeros.  One thread = 8 output elements.
__global__ __launch_bounds__(256) void cvt_rows_bf16_kernel(const float* __restrict__ src, int ld, int rows, int cols,
                                                            unsigned short* __restrict__ dst, int ldd)
{
    const int groups = ldd >> 3;
    const size_t total = (size_t)rows * groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / groups), c = (int)(i - (size_t)r * groups) << 3;
        const float* s = src + (size_t)r * ld + c;
        float v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (c + 4 * q + 4 <= cols) {
                float4 t = *reinterpret_cast<const float4*>(s + 4 * q);
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] = (c + 4 * q + e < cols) ? s[4 * q + e] : 0.f;
            }
        }
        uint4 o = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
        *reinterpret_cast<uint4*>(dst + (size_t)r * ldd + c) = o;
    }
}

// ---------------------------------------------------------------- fp32 [K][X] -> bf16 [X][ldk] (transpose)
// 64x64 tiles through LDS; ldk = K rounded up to 8, pad written as zeros.
__global__ __launch_bounds__(256) void cvt_transpose_bf16_kernel(const float* __restrict__ src, int ld, int K, int X,
                                                                 unsigned short* __restrict__ dst, int ldk)
{
    __shared__ float tile[64][65];
    const int k0 = blockIdx.y * 64, x0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
        int f = tid + 256 * rep, k = f >> 4, xq = (f & 15) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 + k < K && x0 + xq < X) v = *reinterpret_cast<const float4*>(src + (size_t)(k0 + k) * ld + x0 + xq);
        tile[k][xq] = v.x; tile[k][xq + 1] = v.y; tile[k][xq + 2] = v.z; tile[k][xq + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        int f = tid + 256 * rep, x = f >> 3, kg = (f & 7) << 3;       // 8 consecutive lanes write one row's 128 contiguous bytes
        if (x0 + x < X && k0 + kg < ldk) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[kg + e][x];       // rows beyond K were loaded as zeros
            uint4 o = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
            *reinterpret_cast<uint4*>(dst + (size_t)(x0 + x) * ldk + k0 + kg) = o;
        }
    }
}

// ---------------------------------------------------------------- bf16 [K][X] -> bf16 [X][ldk] (transpose of an operand
// that already is bf16: the softmax gradient written by softmax_ce_kernel).  64x64 tiles through LDS, 16-byte accesses on
// both sides; rows beyond K are zero-filled.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const unsigned short* __restrict__ src, int ld, int K, int X,
                                                             unsigned short* __restrict__ dst, int ldk)
{
    __shared__ unsigned short tile[64][72];
    const int k0 = blockIdx.y * 64, x0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int f = tid + 256 * rep, k = f >> 3, xq = (f & 7) << 3;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (k0 + k < K && x0 + xq < X) v = *reinterpret_cast<const uint4*>(src + (size_t)(k0 + k) * ld + x0 + xq);
        *reinterpret_cast<uint4*>(&tile[k][xq]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int f = tid + 256 * rep, x = f >> 3, kg = (f & 7) << 3;
        if (x0 + x < X && k0 + kg < ldk) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (unsigned)tile[kg + 2 * e][x] | ((unsigned)tile[kg + 2 * e + 1][x] << 16);
            *reinterpret_cast<uint4*>(dst + (size_t)(x0 + x) * ldk + k0 + kg) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}
hipError_t transpose_bf16(hipStream_t st, const unsigned short* src, int ld, int K, int X, unsigned short* dst, int ldk)
{
    if ((ld | ldk) & 7 || X & 7) return hipErrorInvalidValue;
    dim3 grid((X + 63) / 64, (ldk + 63) / 64);
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, st, src, ld, K, X, dst, ldk);
    return hipGetLastError();
}

hipError_t cvt_bf16(hipStream_t st, const float* src, int ld, bool transpose, int rows_or_K, int cols_or_X,
                    unsigned short* dst, int ldd)
{
    if (!transpose) {
        size_t total = (size_t)rows_or_K * (ldd >> 3);
        unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(cvt_rows_bf16_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, ld, rows_or_K, cols_or_X, dst, ldd);
    } else {
        dim3 grid((cols_or_X + 63) / 64, (ldd + 63) / 64);
        hipLaunchKernelGGL(cvt_transpose_bf16_kernel, grid, dim3(256), 0, st, src, ld, rows_or_K, cols_or_X, dst, ldd);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------- the GEMM
// Block tile 128x128x64, 4 waves (2x2), each 2x2 MFMA tiles of 32x32; operand lane map of
// v_mfma_f32_32x32x16_bf16: lane l (r = l&31, h = l>>5) holds A[r][k = 8h + j], B[k = 8h + j][r], j = 0..7,
// i.e. one ds_read_b128 per operand tile and k-step.  LDS row stride 72 bf16 = 36 dwords (conflict free).
constexpr int BKH = 64, LDH = BKH + 8;

__device__ __forceinline__ void load_panel(uint4 (&r)[4], const unsigned short* __restrict__ P, int ld, int x0, int X,
                                           int k0, int K, int tid)
{
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
        int f = tid + 256 * rep, x = x0 + (f >> 3), k = k0 + ((f & 7) << 3);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (x < X && k < K) {
            v = *reinterpret_cast<const uint4*>(P + (size_t)x * ld + k);
            if (k + 8 > K) {            // K tail inside this vector (device-side K): drop the elements >= K
                unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k + e >= K) w[e >> 1] &= (e & 1) ? 0x0000FFFFu : 0xFFFF0000u;
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        r[rep] = v;
    }
}
__device__ __forceinline__ void store_panel(unsigned short* __restrict__ s, const uint4 (&r)[4], int tid)
{
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
        int f = tid + 256 * rep;
        *reinterpret_cast<uint4*>(s + (f >> 3) * LDH + ((f & 7) << 3)) = r[rep];
    }
}

struct GemmBf16Args {
    const unsigned short* A; const unsigned short* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    float alpha; int accumulate, split_k; const int* dyn; int dyn_kind;
};

__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(GemmBf16Args g)
{
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * 128 * LDH];
    unsigned short* As = smem;
    unsigned short* Bs = smem + 128 * LDH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    int M = g.M, K = g.K;
    if (g.dyn_kind == 1) M = min(M, *g.dyn);
    if (g.dyn_kind == 2) K = min(K, *g.dyn);
    const int tiles_n = (g.N + 127) / 128;
    int bid = blockIdx.x;
    {
        const int nblk = ((M + 127) / 128) * tiles_n;       // effective tiles (device-side row count), <= gridDim.x
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        if (slot >= q + (xcd < r ? 1 : 0)) return;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    if (m0 >= M) return;
    int kb = 0, ke = K;
    if (g.split_k > 1) {
        int ktiles = (K + BKH - 1) / BKH, per = (ktiles + g.split_k - 1) / g.split_k;
        kb = blockIdx.z * per * BKH; ke = min(K, kb + per * BKH);
        if (kb >= ke) return;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[4], rb[4];
    load_panel(ra, g.A, g.lda, m0, M, kb, ke, tid);
    load_panel(rb, g.B, g.ldb, n0, g.N, kb, ke, tid);
    for (int k0 = kb; k0 < ke; k0 += BKH) {
        store_panel(As, ra, tid);
        store_panel(Bs, rb, tid);
        __syncthreads();
        if (k0 + BKH < ke) {
            load_panel(ra, g.A, g.lda, m0, M, k0 + BKH, ke, tid);
            load_panel(rb, g.B, g.ldb, n0, g.N, k0 + BKH, ke, tid);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + (64 * wm + 32 * t + l31) * LDH + 16 * s + 8 * h));
                b[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + (64 * wn + 32 * t + l31) * LDH + 16 * s + 8 * h));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const bool atomic = g.split_k > 1;
    const bool add_bias = g.bias != nullptr && (!atomic || blockIdx.z == 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int col = n0 + 64 * wn + 32 * j + l31;
        if (col >= g.N) continue;
        float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
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

// ---------------------------------------------------------------- the large-tile form
// The 128x128 kernel above asks the CU's vector-memory path for 64 B/clk at full matrix rate (32 KB of operands per 16 MFMAs
// of 32 cycles with four workgroups sharing the SIMDs) -- exactly what that path delivers, so it cannot pass ~half rate
// (measured 630 TFLOP/s on the shapes of configs[2]).  Block tile 256x256x64 halves the operand bytes per FLOP: one
// 512-thread workgroup per CU, 8 waves as 2 (M) x 4 (N), each 4 x 2 MFMA tiles of 32x32 (128 accumulator registers); the
// same padded k-contiguous LDS image, two buffers (144 KB), ONE barrier per K tile: the next tile's global loads are
// issued before the 32 MFMAs of this tile and written to the other buffer after them.
constexpr int T2 = 256;
constexpr int kBigLds = 2 * 2 * T2 * LDH * 2;        // bytes: [buffer][A | B][256 rows][LDH] bf16

__device__ __forceinline__ void load_panel256(uint4 (&r)[4], const unsigned short* __restrict__ P, int ld, int x0, int X,
                                              int k0, int K, int tid)
{
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
        const int f = tid + 512 * rep, x = x0 + (f >> 3), k = k0 + ((f & 7) << 3);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (x < X && k < K) {
            v = *reinterpret_cast<const uint4*>(P + (size_t)x * ld + k);
            if (k + 8 > K) {            // K tail inside this vector (device-side K): drop the elements >= K
                unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k + e >= K) w[e >> 1] &= (e & 1) ? 0x0000FFFFu : 0xFFFF0000u;
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        r[rep] = v;
    }
}
__device__ __forceinline__ void store_panel256(unsigned short* __restrict__ s, const uint4 (&r)[4], int tid)
{
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
        const int f = tid + 512 * rep;
        *reinterpret_cast<uint4*>(s + (f >> 3) * LDH + ((f & 7) << 3)) = r[rep];
    }
}

__global__ __launch_bounds__(512) void gemm_bf16_nt256_kernel(GemmBf16Args g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short smem2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;

    int M = g.M, K = g.K;
    if (g.dyn_kind == 1) M = min(M, *g.dyn);
    if (g.dyn_kind == 2) K = min(K, *g.dyn);
    const int tiles_n = (g.N + T2 - 1) / T2;
    int bid = blockIdx.x;
    {
        const int nblk = ((M + T2 - 1) / T2) * tiles_n;     // effective tiles (device-side row count), <= gridDim.x
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
        if (slot >= q + (xcd < r ? 1 : 0)) return;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * T2, n0 = tn * T2;
    if (m0 >= M) return;
    int kb = 0, ke = K;
    if (g.split_k > 1) {
        const int ktiles = (K + BKH - 1) / BKH, per = (ktiles + g.split_k - 1) / g.split_k;
        kb = blockIdx.z * per * BKH; ke = min(K, kb + per * BKH);
        if (kb >= ke) return;
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[4], rb[4];
    load_panel256(ra, g.A, g.lda, m0, M, kb, ke, tid);
    load_panel256(rb, g.B, g.ldb, n0, g.N, kb, ke, tid);
    store_panel256(smem2, ra, tid);
    store_panel256(smem2 + T2 * LDH, rb, tid);
    __syncthreads();
    int cur = 0;
    for (int k0 = kb; k0 < ke; k0 += BKH, cur ^= 1) {
        const bool more = k0 + BKH < ke;
        if (more) {
            load_panel256(ra, g.A, g.lda, m0, M, k0 + BKH, ke, tid);
            load_panel256(rb, g.B, g.ldb, n0, g.N, k0 + BKH, ke, tid);
        }
        const unsigned short* As = smem2 + cur * (2 * T2 * LDH);
        const unsigned short* Bs = As + T2 * LDH;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                a[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + (128 * wm + 32 * t + l31) * LDH + 16 * s + 8 * h));
#pragma unroll
            for (int t = 0; t < 2; ++t)
                b[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + (64 * wn + 32 * t + l31) * LDH + 16 * s + 8 * h));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            unsigned short* An = smem2 + (cur ^ 1) * (2 * T2 * LDH);
            store_panel256(An, ra, tid);
            store_panel256(An + T2 * LDH, rb, tid);
        }
        __syncthreads();
    }
    const bool atomic = g.split_k > 1;
    const bool add_bias = g.bias != nullptr && (!atomic || blockIdx.z == 0);
    if (m0 + T2 <= M && n0 + T2 <= g.N && !atomic) {
        // a tile wholly inside the matrix: straight-line stores, one address computation per 32x32 tile (the per-element
        // form below -- row predicate, three-way mode branch -- is ~40 instructions x 128 elements per lane: longer than the
        // K = 512 main loop, and with one workgroup per CU nothing runs beside it)
        const size_t ldc = (size_t)g.ldc;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + l31;
            const float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* c0 = g.C + (size_t)(m0 + 128 * wm + 32 * i + 4 * h) * ldc + col;
                if (g.accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { float* c = c0 + (size_t)((r & 3) + 8 * (r >> 2)) * ldc; *c += g.alpha * acc[i][j][r] + bv; }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c0[(size_t)((r & 3) + 8 * (r >> 2)) * ldc] = g.alpha * acc[i][j][r] + bv;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + 64 * wn + 32 * j + l31;
        if (col >= g.N) continue;
        const float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 128 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
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

// ---------------------------------------------------------------- the TN form: both operands stored [k][x]
// C[M,N] = alpha * A^T B with A [K][lda] (m contiguous) and B [K][ldb] (n contiguous), bf16: the weight-gradient GEMMs
// dW = dgi^T x, dR = dgh^T h_prev, dE = dlogits^T ho, whose operands are activations stored row-major by their producers
// ([row = k][column = m or n]).  The k-contiguous form above needed both operands transposed first (cvt_transpose /
// transpose_bf16 passes over 1.6 GB arrays at configs[2]); here a K tile goes to LDS as it lies in memory ([64 k][256 x],
// rows padded to 288 bf16 = 144 dwords = 16 mod 64 banks) and the MFMA fragments are read with ds_read_b64_tr_b16, the
// CDNA4 transposing LDS read: per 16-lane group a 4 (k) x 16 (x) block, lane 4q + p supplying the address of (row q,
// columns 4p..4p+3), lane i receiving column i's four k (cdna_hip_programming.md T10).  Two reads give a lane its eight
// consecutive k of v_mfma_f32_32x32x16_bf16's operand map (lane (r, h): row r, k = 8h..8h+7).  With the 16-dword row
// skew the two groups of a 32-lane half touch 8 disjoint sets of 8 banks: conflict free.  Same 256x256x64 tile, 8 waves,
// double-buffered LDS and epilogue as gemm_bf16_nt256_kernel.  M, N, lda, ldb multiples of 8.
constexpr int LDT = T2 + 32;
constexpr int kTnLds = 2 * 2 * BKH * LDT * 2;        // bytes: [buffer][A | B][64 k][LDT] bf16 = 144 KB
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void load_panel_kx(uint4 (&r)[4], const unsigned short* __restrict__ P, int ld, int x0, int X, int k0, int K, int tid)
{
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
        const int f = tid + 512 * rep, k = k0 + (f >> 5), x = x0 + ((f & 31) << 3);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (k < K && x < X) v = *reinterpret_cast<const uint4*>(P + (size_t)k * ld + x);
        r[rep] = v;
    }
}
__device__ __forceinline__ void store_panel_kx(unsigned short* __restrict__ s, const uint4 (&r)[4], int tid)
{
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
        const int f = tid + 512 * rep;
        *reinterpret_cast<uint4*>(s + (f >> 5) * LDT + ((f & 31) << 3)) = r[rep];
    }
}
// the fragment of the 32 x-columns starting at xb for the 16 k starting at ks: `toff` = this lane's offset inside the block pair
__device__ __forceinline__ bf16x8 tr_frag(const unsigned short* img, int off)
{
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + off + 4 * LDT));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(512) void gemm_bf16_tn256_kernel(GemmBf16Args g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short smem3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;

    int K = g.K;
    const int M = g.M;
    if (g.dyn_kind == 2) K = min(K, *g.dyn);
    const int tiles_n = (g.N + T2 - 1) / T2;
    const int bid = blockIdx.x;
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * T2, n0 = tn * T2;
    int kb = 0, ke = K;
    if (g.split_k > 1) {
        const int ktiles = (K + BKH - 1) / BKH, per = (ktiles + g.split_k - 1) / g.split_k;
        kb = blockIdx.z * per * BKH; ke = min(K, kb + per * BKH);
        if (kb >= ke) return;                       // (uniform over the workgroup: EXEC stays full for the transposing reads)
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // this lane's place in the 4 x 16 blocks of a transposing read: group gq = lane / 16 -> k half (gq >> 1) and x half (gq & 1)
    const int gq = lane >> 4, li = lane & 15;
    const int toff = (8 * (gq >> 1) + (li >> 2)) * LDT + 16 * (gq & 1) + 4 * (li & 3);

    uint4 ra[4], rb[4];
    load_panel_kx(ra, g.A, g.lda, m0, M, kb, ke, tid);
    load_panel_kx(rb, g.B, g.ldb, n0, g.N, kb, ke, tid);
    store_panel_kx(smem3, ra, tid);
    store_panel_kx(smem3 + BKH * LDT, rb, tid);
    __syncthreads();
    int cur = 0;
    for (int k0 = kb; k0 < ke; k0 += BKH, cur ^= 1) {
        const bool more = k0 + BKH < ke;
        if (more) {
            load_panel_kx(ra, g.A, g.lda, m0, M, k0 + BKH, ke, tid);
            load_panel_kx(rb, g.B, g.ldb, n0, g.N, k0 + BKH, ke, tid);
        }
        const unsigned short* As = smem3 + cur * (2 * BKH * LDT);
        const unsigned short* Bs = As + BKH * LDT;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = tr_frag(As, toff + 16 * s * LDT + 128 * wm + 32 * t);
#pragma unroll
            for (int t = 0; t < 2; ++t) b[t] = tr_frag(Bs, toff + 16 * s * LDT + 64 * wn + 32 * t);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            unsigned short* An = smem3 + (cur ^ 1) * (2 * BKH * LDT);
            store_panel_kx(An, ra, tid);
            store_panel_kx(An + BKH * LDT, rb, tid);
        }
        __syncthreads();
    }
    const bool atomic = g.split_k > 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + 64 * wn + 32 * j + l31;
        if (col >= g.N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 128 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= M) continue;
                const float v = g.alpha * acc[i][j][r];
                float* c = g.C + (size_t)row * g.ldc + col;
                if (atomic) atomicAdd(c, v);
                else if (g.accumulate) *c += v;
                else *c = v;
            }
        }
    }
}

// C (M x N) = alpha * A^T B (+ C): A [K][lda], B [K][ldb] bf16 row-major as their producers wrote them.  K split over
// ~2 rounds of one workgroup per CU with float atomics INTO C (which must hold the value to add onto: the zero-filled
// gradient) when g.split_k > 1, as gemm_bf16_nt re-derives it; g.dyn / dyn_kind 2: device-side K.
hipError_t gemm_bf16_tn(hipStream_t st, const unsigned short* A, int lda, const unsigned short* B, int ldb, const GemmArgs& g)
{
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if ((lda | ldb | g.M | g.N) & 7 || g.dyn_kind == 1 || g.bias) return hipErrorInvalidValue;
    GemmBf16Args a{A, B, g.C, nullptr, g.M, g.N, g.K, lda, ldb, g.ldc, g.alpha, g.accumulate, g.split_k, g.dyn, g.dyn_kind};
    const int big_tiles = ((g.M + T2 - 1) / T2) * ((g.N + T2 - 1) / T2);
    int s2 = g.split_k > 1 ? (512 + big_tiles / 2) / big_tiles : 1;
    s2 = std::max(1, std::min(std::min(s2, 16), std::max(1, g.K / (4 * BKH))));
    if (gemm_bf16_p8_tn_ok(g, lda, ldb)) return gemm_bf16_p8_tn(st, A, lda, B, ldb, g, s2);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_tn256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kTnLds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    a.split_k = s2;
    if (g.split_k > 1 && s2 == 1) a.accumulate = 1;       // the caller's slices were going to ADD into C
    dim3 grid(big_tiles, 1, s2);
    hipLaunchKernelGGL(gemm_bf16_tn256_kernel, grid, dim3(512), kTnLds, st, a);
    return hipGetLastError();
}

bool gemm_bf16_c16_ok(const GemmArgs& g, int lda, int ldb)
{
    const int big_tiles = ((g.M + T2 - 1) / T2) * ((g.N + T2 - 1) / T2);
    return ((lda | ldb) & 7) == 0 && g.N >= 192 && g.M >= 192 && big_tiles >= 200 && g.split_k <= 1 && !g.bias && !g.accumulate && (g.N & 7) == 0 && (g.ldc & 7) == 0 &&
           gemm_bf16_p8_ok(g, lda, ldb);
}

// operands already converted: A [M][lda] bf16, B [N][ldb] bf16, lda/ldb multiples of 8
hipError_t gemm_bf16_nt(hipStream_t st, const unsigned short* A, int lda, const unsigned short* B, int ldb, const GemmArgs& g)
{
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if ((lda | ldb) & 7) return hipErrorInvalidValue;
    if (g.c16 && !gemm_bf16_c16_ok(g, lda, ldb)) return hipErrorInvalidValue;      // (only the phased kernel writes the 2-byte panel: the caller asks first)
    GemmBf16Args a{A, B, g.C, g.bias, g.M, g.N, g.K, lda, ldb, g.ldc, g.alpha, g.accumulate, g.split_k, g.dyn, g.dyn_kind};
    // large tiles where they fill the chip, by themselves or through the K split (a caller's split was sized for 128x128
    // tiles at three workgroups per CU; here one workgroup per CU, so it is re-derived: ~2 rounds of 256 workgroups)
    const int big_tiles = ((g.M + T2 - 1) / T2) * ((g.N + T2 - 1) / T2);
    int s2 = g.split_k > 1 ? (512 + big_tiles / 2) / big_tiles : 1;
    s2 = std::max(1, std::min(std::min(s2, 16), std::max(1, g.K / (4 * BKH))));      // (more slices: the float atomics dominate)
    if (g.N >= 192 && g.M >= 192 && big_tiles * s2 >= 200) {
        if (gemm_bf16_p8_ok(g, lda, ldb)) return gemm_bf16_p8(st, A, lda, B, ldb, g, s2);
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_nt256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kBigLds);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        a.split_k = s2;
        if (g.split_k > 1 && s2 == 1) a.accumulate = 1;       // the caller's slices were going to ADD into C
        dim3 grid(big_tiles, 1, s2);
        hipLaunchKernelGGL(gemm_bf16_nt256_kernel, grid, dim3(512), kBigLds, st, a);
        return hipGetLastError();
    }
    int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
    dim3 grid(tiles, 1, g.split_k > 1 ? g.split_k : 1);
    hipLaunchKernelGGL(gemm_bf16_nt_kernel, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace avae
