// gemm_bf16_p8.hip -- the phased, persistent bf16 NT GEMM of compute_dtype 1 (BASELINE configs[2]):
// C[M,N] = alpha * A[M,K] * B[N,K]^T (+ bias, + C), both operands k-contiguous bf16, fp32 accumulate on
// v_mfma_f32_16x16x32_bf16.
//
// Why a second 256x256 kernel: gemm_bf16_nt256_kernel (gemm_bf16.hip) stages through registers behind ONE barrier per K
// tile, all eight waves in step -- the matrix pipe idles while they read LDS, write LDS and wait for the loads -- and it
// is one output tile per workgroup at one workgroup per CU: a CU sits idle from the moment a tile's stores are issued
// until they have drained (a wave ends when its stores are acknowledged) and again until the next workgroup's first
// operands arrive.  At K = 512 (logits, decoder projections) that was 2/3 of the time.  Here
//   * the operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), four
//     128-row x 64-k pieces ("half tiles", 16 KB) per K tile into a two-tile ring (128 KB), each piece issued FOUR phases
//     before its first read and retired by a COUNTED s_waitcnt vmcnt -- the queue is never drained inside the loop;
//   * a K tile is four phases {ds_read a register sub-tile | issue one half tile || barrier | 16 MFMAs | barrier}, and
//     the two wave groups (the M halves of the tile) run one barrier apart: while one group's 16 MFMAs run, the other
//     group reads its fragments and issues its DMA pieces -- every SIMD holds one wave of each group;
//   * the LDS image of a half tile is lane-linear (a DMA piece of one wave = 8 rows x 128 B), so the bank swizzle sits on
//     the per-lane SOURCE address: 16-byte chunk c of row r is stored at chunk position c ^ (r & 7), which makes the
//     fragment reads (16 rows x 4 chunks per wave-instruction) conflict free;
//   * the MFMA takes B's fragment as its first operand: a lane then holds four CONSECUTIVE columns of one row of C and
//     the epilogue stores 16 bytes per lane;
//   * one workgroup per CU walks output tiles (grid = min(work items, 256)); the K-tile stream runs ACROSS output tiles:
//     while a tile's last K tiles are multiplied the ring already fills with the next tile's first ones, and the tile's 32
//     stores per wave are issued between two phases and drain while the next tile is multiplied.  Loads and stores retire
//     in issue order on one counter, so the three waits that follow an epilogue allow for its operations (vmcnt(8 + E),
//     E = 32 stores + the bias piece); an epilogue that may skip stores (edge tiles), read C or use atomics (K split) ends
//     with vmcnt(0) instead, which makes any later count safe;
//   * the bias row of a tile (4 x 64 floats) rides the same DMA queue into a per-wave LDS strip: no ordinary load -- the
//     compiler would wait for it with vmcnt(0) and drain the ring.
//
// Half tiles of K tile u (a flat count over this workgroup's tiles), ring slot (u & 1): AE = the 64 + 64 rows the two M
// groups read in phase 1, BE = the 4 x 32 columns read in phase 1 (kept in registers for phase 4), BL = the other 4 x 32
// columns (phase 2), AL = the other 64 + 64 rows (phase 3).  A slot piece is re-filled two or more phases after its last
// read:  phase 1 of u issues BL(u+1), phase 2 AL(u+1), phase 3 AE(u+2), phase 4 BE(u+2);
// phases 4, 1 and 2 wait vmcnt(8) after their issue (= everything but the last four half tiles has landed: the pieces the
// NEXT phase reads), every wave waits before the phase's first barrier, and the read follows one phase later -- the order
// LDS-DMA data needs (cdna_hip_programming.md section 5, "Read a staged buffer one phase AFTER the wait that retires it").
// Beyond the last K tile of the last work item the issue slots re-load that item's first tiles (into pieces nobody reads
// any more): the counts stay uniform.  K must be a multiple of 64 (>= 128) and N of 4; rows beyond M / columns beyond N
// are clamped on the load side and masked on the store side.  K tails and tiny shapes stay on gemm_bf16.hip's kernels.
#include <algorithm>
#include <type_traits>
#include <utility>
#include "kernels.h"

namespace avae {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kP8Tile = 256, kP8K = 64, kP8Piece = 16384, kP8Ring = 8 * kP8Piece, kP8Lds = kP8Ring + 8 * 256, kP8Group = 8;
constexpr int kP8Stores = 32;                            // stores per wave of a whole-tile epilogue
enum { AE = 0, BE = 1, BL = 2, AL = 3 };

struct P8Args {
    const unsigned short* A; const unsigned short* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    float alpha; int accumulate, split_k; const int* dyn; int dyn_kind;
    unsigned short* C16; // NT, whole problem on this kernel: the result as an fp16 panel [M][ldc] instead of fp32 C (GemmArgs::c16)
    float* slab;         // TN, K split: the slices' partial tiles go here as plain stores ([slice][tile][256][256] fp32), p8_slab_reduce_kernel sums them into C
};

__device__ __forceinline__ void dma16(const unsigned short* g, unsigned char* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void dma16b(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, unsigned char* l)      // bounds-checked: zeros beyond num_records
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, (int)byte_off, 0, 0, 0);
}
__device__ __forceinline__ void dma4(const float* g, unsigned char* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}
// eight consecutive k of one column out of a [k][x] image (256-byte rows): two ds_read_b64_tr_b16, four k rows apart.  Inline asm:
// behind the intrinsic hipcc (ROCm 7.2) waits vmcnt(0) before every such read while an LDS-DMA is in flight (it cannot tell the
// read from the DMA's target), which drains the ring each phase.  The caller waits lgkmcnt(0) itself before the first use.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ bf16x8 frag_tr(unsigned lds_addr)
{
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(lds_addr), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(lds_addr), "n"(OFF + 1024));
    return __builtin_bit_cast(bf16x8, u32x4v{lo[0], lo[1], hi[0], hi[1]});
}
template <class F, int... I> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }
__device__ __forceinline__ bf16x8 frag(const unsigned char* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p)); }

#define P8_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define P8_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory")

// The accumulator layout gives lane (fr, fq) columns 4 fq .. 4 fq + 3 of row fr in each of two 16-column tiles (x: columns 0-15, y: 16-31 of
// the wave's 32): stored as they stand, one instruction writes 16 rows x 64 B -- half lines, 3.0-3.3 TB/s chip-wide on a store-only
// stream (scripts/micro/store_pattern.hip), and a K = 512 output tile's 256 KB took longer to drain than its MFMAs to run.  Lanes fr and
// fr ^ 1 trade one register set (quad-permute DPP): the even lane ends up with columns 4 fq.. of rows fr and fr + 1 out of x, the odd lane with
// those of rows fr - 1 and fr out of y -- s0 goes to row (fr & ~1), s1 to the row below, both at 16-byte chunk (fr & 1) * 4 + fq of the 32
// columns: one instruction = 8 rows x 128 B, whole lines, 5.0-5.2 TB/s.
__device__ __forceinline__ void p8_whole_lines(f32x4 x, f32x4 y, int odd, f32x4& s0, f32x4& s1)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float send = odd ? x[e] : y[e];
        const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, false));      // quad_perm [1, 0, 3, 2]
        s0[e] = odd ? recv : x[e];
        s1[e] = odd ? y[e] : recv;
    }
}

constexpr bool p8_of_a(int piece) { return piece == AE || piece == AL; }
struct P8Item { int m0, n0, kt0, nkt, z, tl; };

// two floats -> one dword of two fp16 (round to nearest even)
__device__ __forceinline__ unsigned p8_pk_f16(float a, float b)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, h2));
}

// H16: the output tile leaves as fp16 (P8Args::C16; no bias, no accumulate, no K split): half the bytes of the store drain that a K = 512 tile waits for
template <bool BIAS, bool TN, bool H16 = false>
__global__ __launch_bounds__(512) void gemm_bf16_p8_kernel(P8Args g)
{
    static_assert(!H16 || (!BIAS && !TN), "the fp16 panel: plain NT products only");
    extern __shared__ __attribute__((aligned(1024))) unsigned char L[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int E = (H16 ? kP8Stores / 2 : kP8Stores) + (BIAS ? 1 : 0);

    int M = g.M, K = g.K;
    if (g.dyn_kind == 1) M = min(M, *g.dyn);
    if (g.dyn_kind == 2) K = min(K, *g.dyn);             // (TN only: the K tail is zero filled by the buffer loads' bounds check)
    const int tiles_m = (M + kP8Tile - 1) / kP8Tile, tiles_n = (g.N + kP8Tile - 1) / kP8Tile, tiles = tiles_m * tiles_n;
    // every item holds two K tiles or more (the look-ahead crosses ONE item boundary): NT -- the host keeps 2 * split <= K / 64;
    // TN -- K tiles beyond K read zeros, so a short K is padded to two tiles and the split shrinks with a device-side K
    const int nkt_all = TN ? max((K + kP8K - 1) / kP8K, 2) : K / kP8K, nslice = TN ? max(min(g.split_k, nkt_all >> 1), 1) : max(g.split_k, 1);
    const int work = tiles * nslice, G = gridDim.x;
    // this workgroup's place in a round of G work items: the (up to) 32 workgroups of an XCD take consecutive items
    int place;
    {
        const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        place = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    if (place >= work) return;
    // item v -> K slice v / tiles, tile v % tiles in groups of kP8Group tile rows walked column by column: the ~32 tiles an
    // XCD works on at a time are an 8 x 4 block (12 operand panels through its L2 per K tile) instead of a tile row (33)
    auto item = [&](int v) {
        P8Item it;
        it.z = v / tiles;
        const int tl = it.tl = v - it.z * tiles, per = kP8Group * tiles_n, grp = tl / per, in = tl - grp * per;
        const int rows = min(kP8Group, tiles_m - grp * kP8Group), tn = in / rows;
        it.m0 = (grp * kP8Group + in - tn * rows) * kP8Tile; it.n0 = tn * kP8Tile;
        it.kt0 = it.z * nkt_all / nslice; it.nkt = (it.z + 1) * nkt_all / nslice - it.kt0;
        return it;
    };

    // ---- DMA sources (byte offsets from A / B as 32-bit values -- the host checks both operands are below 4 GB --: 16 registers
    // for the current and the next item).  Pieces: AE = rows m0 .. m0+127 of the tile, AL = the other 128, BE / BL likewise in n.
    // NT: instruction j of a piece covers its rows j*64 + wave*8 + (lane >> 3); this lane moves 16-byte chunk (lane & 7) ^ (row & 7)
    // of the row's 64 k into chunk position lane & 7.
    // TN: a piece is [64 k][128 x] (256-byte rows); instruction j covers k rows j*32 + wave*4 + (lane >> 4); a 32-byte chunk c of k row
    // kr sits at chunk position c ^ key(kr), key = (kr & 3) | ((kr >> 3) & 1) << 2: the eight rows a 32-lane half reads with one
    // transposing ds_read_b64_tr_b16 (k = 8 fq + 0..3 for two neighbouring fq) then fall on eight different 32-byte bank groups.
    const unsigned strideA = TN ? (unsigned)g.lda * (kP8K * 2) : kP8K * 2, strideB = TN ? (unsigned)g.ldb * (kP8K * 2) : kP8K * 2;      // bytes per K tile
    auto sources = [&](const P8Item& it, unsigned (&src)[4][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (!TN) {
                const int R = j * 64 + wave * 8 + (lane >> 3);
                const unsigned k = (unsigned)(it.kt0 * kP8K + ((lane & 7) ^ (lane >> 3)) * 8) * 2u;
                src[AE][j] = (unsigned)min(it.m0 + R, M - 1) * (unsigned)g.lda * 2u + k;
                src[AL][j] = (unsigned)min(it.m0 + 128 + R, M - 1) * (unsigned)g.lda * 2u + k;
                src[BE][j] = (unsigned)min(it.n0 + R, g.N - 1) * (unsigned)g.ldb * 2u + k;
                src[BL][j] = (unsigned)min(it.n0 + 128 + R, g.N - 1) * (unsigned)g.ldb * 2u + k;
            } else {
                const int kr = j * 32 + wave * 4 + (lane >> 4), p16 = lane & 15;
                const int key = (kr & 3) | (((kr >> 3) & 1) << 2);
                const int x = ((p16 >> 1) ^ key) * 16 + (p16 & 1) * 8;
                const unsigned ka = (unsigned)(it.kt0 * kP8K + kr) * (unsigned)g.lda * 2u, kb = (unsigned)(it.kt0 * kP8K + kr) * (unsigned)g.ldb * 2u;
                src[AE][j] = ka + (unsigned)min(it.m0 + x, M - 8) * 2u;
                src[AL][j] = ka + (unsigned)min(it.m0 + 128 + x, M - 8) * 2u;
                src[BE][j] = kb + (unsigned)min(it.n0 + x, g.N - 8) * 2u;
                src[BL][j] = kb + (unsigned)min(it.n0 + 128 + x, g.N - 8) * 2u;
            }
        }
    };
    // TN: bounds-checked LDS-DMA (rows k >= K read as zeros)
    __amdgpu_buffer_rsrc_t rsA, rsB;
    if constexpr (TN) {
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(g.A), 0, (int)((unsigned)K * (unsigned)g.lda * 2u), 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(g.B), 0, (int)((unsigned)K * (unsigned)g.ldb * 2u), 0x00020000);
    }
    unsigned char* const dst = L + wave * 1024;          // + slot * 65536 + piece * 16384 + j * 8192
    unsigned char* const bias_lds = L + kP8Ring + wave * 256;
    auto bias_dma = [&](const P8Item& it) {      // this wave's 32 + 32 columns
        if constexpr (BIAS) dma4(g.bias + min(it.n0 + wc * 32 + (lane & 31) + (lane >> 5) * 128, g.N - 1), bias_lds);
    };

    // ---- fragment addresses.  NT: row fr of a 16-row group, chunk (ks * 4 + fq) ^ (fr & 7); ks = 1 flips byte bit 6.
    // TN: lane (fr = 4 q + p, fq) supplies the address of (k row 8 fq + q, columns 4 p .. 4 p + 3) of a 16-column group and gets
    // column fr's four k (rows + 4: the other four); the group's 32-byte chunk sits at position chunk ^ key, key = q | (fq & 1) << 2
    const int offA = TN ? (8 * fq + (fr >> 2)) * 256 + ((wr ^ (fq & 1)) << 7) + 8 * (fr & 3) : (wr * 64 + fr) * 128 + ((fq ^ (fr & 7)) << 4);
    const int offB = TN ? (8 * fq + (fr >> 2)) * 256 + 8 * (fr & 3) : (wc * 32 + fr) * 128 + ((fq ^ (fr & 7)) << 4);
    const int tkey = (fr >> 2) | ((fq & 1) << 2);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)L;
    // a piece's A (4 x 16 rows) or B (2 x 16 columns) sub-tile for both 32-deep k steps
    auto read_a = [&](int slot, auto piece, bf16x8 (&r)[4][2]) {
        constexpr int P = decltype(piece)::value;
        if constexpr (TN) {
            sfor<4>([&](auto mt) { sfor<2>([&](auto ks) {
                r[mt][ks] = frag_tr<P * kP8Piece + decltype(ks)::value * 8192>(lds0 + slot * 65536 + offA + ((decltype(mt)::value ^ (fr >> 2)) << 5)); }); });
        } else {
            const unsigned char* const S = L + slot * 65536 + P * kP8Piece;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) r[mt][ks] = frag(S + ((offA + mt * 2048) ^ (ks * 64)));
        }
    };
    auto read_b = [&](int slot, auto piece, bf16x8 (&r)[2][2]) {
        constexpr int P = decltype(piece)::value;
        if constexpr (TN) {
            sfor<2>([&](auto nt) { sfor<2>([&](auto ks) {
                r[nt][ks] = frag_tr<P * kP8Piece + decltype(ks)::value * 8192>(lds0 + slot * 65536 + offB + (((wc * 2 + decltype(nt)::value) ^ tkey) << 5)); }); });
        } else {
            const unsigned char* const S = L + slot * 65536 + P * kP8Piece;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) r[nt][ks] = frag(S + ((offB + nt * 2048) ^ (ks * 64)));
        }
    };
#define P8_P(x) std::integral_constant<int, (x)>{}

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4][2], b0[2][2], b1[2][2];

    P8Item cur = item(place), nxt = place + G < work ? item(place + G) : cur;
    unsigned sc[4][2], sn[4][2];
    sources(cur, sc); sources(nxt, sn);
    int u = 0;                                           // flat K-tile count: ring slot u & 1
    // piece `piece` of the K tile d (1 or 2) after tile kt of the current item (flat index ub): from the next item beyond the current one's end
#define P8_STAGE(piece, kt, d, ub) do { const int k_ = (kt) + (d); const bool in_ = k_ < cur.nkt; \
        const unsigned o_ = (unsigned)(in_ ? k_ : k_ - cur.nkt) * (p8_of_a(piece) ? strideA : strideB); \
        unsigned char* d_ = dst + (((ub) + (d)) & 1) * 65536 + (piece) * kP8Piece; \
        const unsigned s0_ = (in_ ? sc[piece][0] : sn[piece][0]) + o_, s1_ = (in_ ? sc[piece][1] : sn[piece][1]) + o_; \
        if constexpr (TN) { dma16b(p8_of_a(piece) ? rsA : rsB, s0_, d_); dma16b(p8_of_a(piece) ? rsA : rsB, s1_, d_ + 8192); } \
        else { const unsigned char* base_ = reinterpret_cast<const unsigned char*>(p8_of_a(piece) ? g.A : g.B); \
               dma16(reinterpret_cast<const unsigned short*>(base_ + (size_t)s0_), d_); dma16(reinterpret_cast<const unsigned short*>(base_ + (size_t)s1_), d_ + 8192); } } while (0)

    // ---- prologue: the bias strip, K tiles 0 and 1 whole; AE(0), BE(0) landed before the first read
    bias_dma(cur);
    P8_STAGE(AE, -1, 1, -1); P8_STAGE(BE, -1, 1, -1); P8_STAGE(BL, -1, 1, -1); P8_STAGE(AL, -1, 1, -1);
    P8_STAGE(AE, -1, 2, -1); P8_STAGE(BE, -1, 2, -1); P8_STAGE(BL, -1, 2, -1); P8_STAGE(AL, -1, 2, -1);
    P8_WAIT(12);
    P8_BARRIER();
    if (wr == 1) P8_BARRIER();                           // the second M group runs one barrier behind the first

#define P8_MFMA(AI, BJ, BV) do { if constexpr (TN) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) \
            acc[(AI) + mt][(BJ) + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BV[nt][ks], a[mt][ks], acc[(AI) + mt][(BJ) + nt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0); } while (0)

    // one K tile.  first: the K tile that opens an item -- its BL(u+1), AL(u+1) were issued ahead of the previous item's epilogue (or
    // by the prologue), phases 1 and 2 issue nothing and phase 1's wait allows for those two pieces.  plus: the waits allow for the E
    // operations of the whole-tile epilogue that were issued behind the pieces they retire (the item's first two K tiles)
    auto ktile = [&](int kt, bool first, bool plus) {
        const int slot = u & 1;
        // phase 1: AE, BE -> registers; issue BL(u+1)
        read_b(slot, P8_P(BE), b0);
        read_a(slot, P8_P(AE), a);
        if (!first) P8_STAGE(BL, kt, 1, u);
        if (plus) { if (first) P8_WAIT(10 + E); else P8_WAIT(8 + E); } else { if (first) P8_WAIT(10); else P8_WAIT(8); }
        P8_BARRIER();
        P8_MFMA(0, 0, b0);
        P8_BARRIER();
        // phase 2: BL -> registers; issue AL(u+1)
        read_b(slot, P8_P(BL), b1);
        if (!first) P8_STAGE(AL, kt, 1, u);
        if (plus) P8_WAIT(8 + E); else P8_WAIT(8);
        P8_BARRIER();
        P8_MFMA(0, 2, b1);
        P8_BARRIER();
        // phase 3: AL -> registers; issue AE(u+2)
        read_a(slot, P8_P(AL), a);
        P8_STAGE(AE, kt, 2, u);
        P8_BARRIER();
        P8_MFMA(4, 2, b1);
        P8_BARRIER();
        // phase 4: no reads (BE is still in registers); issue BE(u+2)
        P8_STAGE(BE, kt, 2, u);
        if (plus && first) P8_WAIT(8 + E); else P8_WAIT(8);
        P8_BARRIER();
        P8_MFMA(4, 0, b0);
        P8_BARRIER();
        ++u;
    };
    bool counted = false;                                // did the previous item end with a whole-tile epilogue (E operations, none waited for)?
    for (int v = place;;) {
        for (int kt = 0; kt < cur.nkt; ++kt) ktile(kt, kt == 0, counted && kt < 2);
        // the next item's second K tile: its BL and AL go out AHEAD of this item's stores -- the first wait for a piece issued behind
        // the stores is then two K tiles away (loads and stores retire in issue order)
        P8_STAGE(BL, cur.nkt - 1, 2, u - 1); P8_STAGE(AL, cur.nkt - 1, 2, u - 1);

        // ---- epilogue: lane (fr, fq) of tile (i, j) holds C[row fr][columns 4 fq .. 4 fq + 3]
        const bool atomic = nslice > 1;
        const bool add_bias = BIAS && cur.z == 0;
        // rows m0 + wr*64 + (i >> 2)*128 + (i & 3)*16 + fr, columns n0 + wc*32 + (j >> 1)*128 + (j & 1)*16 + 4 fq .. + 3
        const int rbase = cur.m0 + wr * 64 + fr, cbase = cur.n0 + wc * 32 + fq * 4;
        counted = !TN && cur.m0 + kP8Tile <= M && cur.n0 + kP8Tile <= g.N && !atomic && !g.accumulate;
        if (TN && atomic && g.slab) {
            // a K slice's partial tile, whole, into its slab (edge tiles too: what lies beyond M / N is never read back): 32 plain stores,
            // no float atomics -- in this accumulator layout they come as 16 rows x 4 scattered dwords per instruction, far off their rate
            counted = true;
            float* const t0 = g.slab + ((size_t)(cur.z * tiles + cur.tl) << 16) + (wr * 64 + (fr & ~1)) * kP8Tile + wc * 32 + ((fr & 1) * 4 + fq) * 4;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x4 s0, s1;
                    p8_whole_lines(g.alpha * acc[i][2 * jp], g.alpha * acc[i][2 * jp + 1], fr & 1, s0, s1);
                    float* const t = t0 + ((i >> 2) * 128 + (i & 3) * 16) * kP8Tile + jp * 128;
                    *reinterpret_cast<f32x4*>(t) = s0;
                    *reinterpret_cast<f32x4*>(t + kP8Tile) = s1;
                }
        } else
        if (H16 && counted) {
            // fp16: lane (fr, fq) packs its four columns of x (tile 2 jp) and y (tile 2 jp + 1) into two dwords each; v_permlane16_swap trades the odd
            // 16-lane rows of x with the even rows of y, which leaves fq = 0 / 2 with columns 0-7 / 8-15 and fq = 1 / 3 with columns 16-23 / 24-31 of
            // the wave's 32: one 16-byte store per lane, 16 rows x 64 B per instruction, 16 instructions per wave and tile
            unsigned short* const h0 = g.C16 + (size_t)rbase * g.ldc + cur.n0 + wc * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 x = g.alpha * acc[i][2 * jp], y = g.alpha * acc[i][2 * jp + 1];
                    const auto r0 = __builtin_amdgcn_permlane16_swap(p8_pk_f16(x[0], x[1]), p8_pk_f16(y[0], y[1]), false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(p8_pk_f16(x[2], x[3]), p8_pk_f16(y[2], y[3]), false, false);
                    *reinterpret_cast<u32x4v*>(h0 + (size_t)((i >> 2) * 128 + (i & 3) * 16) * g.ldc + jp * 128) = u32x4v{r0[0], r1[0], r0[1], r1[1]};
                }
        } else if (H16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = cbase + (j >> 1) * 128 + (j & 1) * 16;
                if (col >= g.N) continue;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = rbase + (i >> 2) * 128 + (i & 3) * 16;
                    if (row >= M) continue;
                    const f32x4 val = g.alpha * acc[i][j];
                    *reinterpret_cast<u32x2*>(g.C16 + (size_t)row * g.ldc + col) = u32x2{p8_pk_f16(val[0], val[1]), p8_pk_f16(val[2], val[3])};
                }
            }
            P8_WAIT(0);
        } else
        if (counted) {
            // whole 128-byte lines per store instruction (8 rows x 128 B instead of 16 rows x 64 B: p8_whole_lines)
            float* const c0 = g.C + (size_t)(cur.m0 + wr * 64 + (fr & ~1)) * g.ldc + cur.n0 + wc * 32 + ((fr & 1) * 4 + fq) * 4;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                f32x4 bv0 = f32x4{0.f, 0.f, 0.f, 0.f}, bv1 = bv0;
                if constexpr (BIAS) {
                    bv0 = *reinterpret_cast<const f32x4*>(bias_lds + (jp * 32 + fq * 4) * 4);
                    bv1 = *reinterpret_cast<const f32x4*>(bias_lds + (jp * 32 + 16 + fq * 4) * 4);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x4 s0, s1;
                    p8_whole_lines(g.alpha * acc[i][2 * jp] + bv0, g.alpha * acc[i][2 * jp + 1] + bv1, fr & 1, s0, s1);
                    float* const c = c0 + (size_t)((i >> 2) * 128 + (i & 3) * 16) * g.ldc + jp * 128;
                    *reinterpret_cast<f32x4*>(c) = s0;
                    *reinterpret_cast<f32x4*>(c + g.ldc) = s1;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = cbase + (j >> 1) * 128 + (j & 1) * 16;
                if (col >= g.N) continue;                // (N is a multiple of 4: a lane's four columns are in or out together)
                f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (BIAS) { if (add_bias) bv = *reinterpret_cast<const f32x4*>(bias_lds + ((j >> 1) * 32 + (j & 1) * 16 + fq * 4) * 4); }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = rbase + (i >> 2) * 128 + (i & 3) * 16;
                    if (row >= M) continue;
                    float* c = g.C + (size_t)row * g.ldc + col;
                    const f32x4 val = g.alpha * acc[i][j] + bv;
                    if (atomic) { atomicAdd(c, val[0]); atomicAdd(c + 1, val[1]); atomicAdd(c + 2, val[2]); atomicAdd(c + 3, val[3]); }
                    else if (g.accumulate) *reinterpret_cast<f32x4*>(c) = *reinterpret_cast<const f32x4*>(c) + val;
                    else *reinterpret_cast<f32x4*>(c) = val;
                }
            }
            P8_WAIT(0);                                  // (an unknown number of operations: drain, whatever count follows is then safe)
        }
        v += G;
        if (v >= work) break;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        cur = nxt;
#pragma unroll
        for (int p = 0; p < 4; ++p) { sc[p][0] = sn[p][0]; sc[p][1] = sn[p][1]; }
        if (v + G < work) { nxt = item(v + G); sources(nxt, sn); }
        bias_dma(cur);                                   // (the strip's reads above are complete: their values went into the stores)
    }
    if (wr == 0) P8_BARRIER();                           // (every wave passes the same number of barriers)
    P8_WAIT(0);                                          // (the re-loads beyond the end must have landed before the LDS is given back)
#undef P8_STAGE
#undef P8_MFMA
#undef P8_P
}

// sums the K slices' partial tiles into C: C (+)= sum over slices of slab[slice][tile] (fixed order: deterministic).  One thread = four
// columns of one row.  The slice count is re-derived from the (device-side) K exactly as the GEMM did; one slice = the GEMM wrote C itself.
__global__ __launch_bounds__(256) void p8_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int M, int N, int ldc, int K,
                                                             const int* __restrict__ dyn, int dyn_kind, int split, int add)
{
    if (dyn_kind == 2) K = min(K, *dyn);
    const int nkt_all = max((K + kP8K - 1) / kP8K, 2), nslice = max(min(split, nkt_all >> 1), 1);
    if (nslice == 1) return;
    const int tiles_n = (N + kP8Tile - 1) / kP8Tile, tiles = ((M + kP8Tile - 1) / kP8Tile) * tiles_n, n4 = N >> 2;
    const size_t total = (size_t)M * n4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(e / n4), n = (int)(e - (size_t)m * n4) << 2;
        // the tile index as the GEMM numbers it: groups of kP8Group tile rows, column by column inside a group
        const int tm = m >> 8, tn = n >> 8, tiles_m = (M + kP8Tile - 1) / kP8Tile, grp = tm / kP8Group, rows = min(kP8Group, tiles_m - grp * kP8Group);
        const int tl = grp * kP8Group * tiles_n + tn * rows + (tm - grp * kP8Group);
        const float* p = slab + ((size_t)tl << 16) + (m & 255) * kP8Tile + (n & 255);
        f32x4 sum = *reinterpret_cast<const f32x4*>(p);
        for (int z = 1; z < nslice; ++z) sum += *reinterpret_cast<const f32x4*>(p + ((size_t)z * tiles << 16));
        f32x4* c = reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n);
        *c = add ? *c + sum : sum;
    }
}

bool gemm_bf16_p8_ok(const GemmArgs& g, int lda, int ldb)
{
    return g.nt8 && g.K % kP8K == 0 && g.K >= 2 * kP8K && (g.N & 3) == 0 && (g.ldc & 3) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && g.dyn_kind != 2 &&
           (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 && (size_t)g.M * lda < ((size_t)1 << 31) && (size_t)g.N * ldb < ((size_t)1 << 31);
}
// the TN form (both operands [k][x] row-major): any K (device-side too), M, N, lda, ldb multiples of 8
bool gemm_bf16_p8_tn_ok(const GemmArgs& g, int lda, int ldb)
{
    return g.nt8 && ((g.M | g.N | lda | ldb) & 7) == 0 && (g.ldc & 3) == 0 && g.dyn_kind != 1 && !g.bias && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
           (size_t)g.K * lda < ((size_t)1 << 31) && (size_t)g.K * ldb < ((size_t)1 << 31) && g.M >= 8 && g.N >= 8;
}

static hipError_t p8_attrs()
{
    static bool done = false;
    if (done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_p8_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kP8Lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_p8_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kP8Lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_p8_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kP8Lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_p8_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kP8Lds);
    if (e == hipSuccess) done = true;
    return e;
}

// s2: the K split (>= 1) as gemm_bf16_nt derived it; the caller's g.split_k > 1 says C holds the value to add onto
hipError_t gemm_bf16_p8(hipStream_t st, const unsigned short* A, int lda, const unsigned short* B, int ldb, const GemmArgs& g, int s2)
{
    P8Args a{A, B, g.C, g.bias, g.M, g.N, g.K, lda, ldb, g.ldc, g.alpha, g.accumulate, s2, g.dyn, g.dyn_kind, g.c16, nullptr};
    if (g.c16 && (g.bias || g.accumulate || g.split_k > 1 || s2 > 1)) return hipErrorInvalidValue;
    if (g.split_k > 1 && s2 == 1) a.accumulate = 1;       // the caller's slices were going to ADD into C
    const int nkt = g.K / kP8K;
    if (2 * s2 > nkt) a.split_k = s2 = std::max(1, nkt / 2);       // (every slice holds two K tiles or more)
    hipError_t e = p8_attrs();
    if (e != hipSuccess) return e;
    const int tiles = ((g.M + kP8Tile - 1) / kP8Tile) * ((g.N + kP8Tile - 1) / kP8Tile);
    const int grid = std::min(tiles * s2, 256);
    if (g.c16) hipLaunchKernelGGL((gemm_bf16_p8_kernel<false, false, true>), dim3(grid), dim3(512), kP8Lds, st, a);
    else if (g.bias) hipLaunchKernelGGL((gemm_bf16_p8_kernel<true, false>), dim3(grid), dim3(512), kP8Lds, st, a);
    else hipLaunchKernelGGL((gemm_bf16_p8_kernel<false, false>), dim3(grid), dim3(512), kP8Lds, st, a);
    return hipGetLastError();
}

// C (M x N) = alpha * A^T B (+ C): A [K][lda], B [K][ldb] bf16 row-major.  s2 > 1: K slices added into C with float atomics (C holds
// the value to add onto); the device shrinks the split when a device-side K leaves fewer than two K tiles per slice
hipError_t gemm_bf16_p8_tn(hipStream_t st, const unsigned short* A, int lda, const unsigned short* B, int ldb, const GemmArgs& g, int s2)
{
    const int tiles = ((g.M + kP8Tile - 1) / kP8Tile) * ((g.N + kP8Tile - 1) / kP8Tile);
    if (g.split_k > 1) {
        // the K split of a persistent grid: the smallest one whose tiles x slices fill whole rounds of 256 workgroups to 93 % or more
        // (16 K tiles or more per slice; fewer slices = fewer float atomics), else the best-filling one
        const int nkt = (g.K + kP8K - 1) / kP8K;
        int best = 1; double best_eff = 0.0;
        for (int c = 1; c <= 32 && c * 16 <= std::max(nkt, 16); ++c) {
            const int items = tiles * c, rounds = (items + 255) / 256;
            const double eff = (double)items / (rounds * 256.0);
            if (eff > best_eff + 1e-9) { best = c; best_eff = eff; }
            if (eff >= 0.93) { best = c; break; }
        }
        s2 = best;
    }
    // (a caller's split says C holds the value to add onto -- also when the device shrinks the split to one slice)
    const int add = g.accumulate || g.split_k > 1;
    const bool use_slab = s2 > 1 && g.slab && (size_t)tiles * s2 * (kP8Tile * kP8Tile) <= g.slab_floats;
    P8Args a{A, B, g.C, nullptr, g.M, g.N, g.K, lda, ldb, g.ldc, g.alpha, add, s2, g.dyn, g.dyn_kind, nullptr, use_slab ? g.slab : nullptr};
    hipError_t e = p8_attrs();
    if (e != hipSuccess) return e;
    const int grid = std::min(tiles * s2, 256);
    hipLaunchKernelGGL((gemm_bf16_p8_kernel<false, true>), dim3(grid), dim3(512), kP8Lds, st, a);
    if (use_slab) {
        const size_t total = (size_t)g.M * (g.N >> 2);
        const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 2048);
        hipLaunchKernelGGL(p8_slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, g.slab, g.C, g.M, g.N, g.ldc, g.K, g.dyn, g.dyn_kind, s2, add);
    }
    return hipGetLastError();
}

}  // namespace avae
