// ops.hip -- the HBM-bound kernels of the VAE step: id preparation, embedding gather /
// scatter-add, compaction, final-state pick, latent elementwise, softmax cross-entropy,
// column sums, TF-style Adam, layout permutation.  Each cites the reference lines it replaces.
#include <algorithm>
#include "kernels.h"

namespace avae {

// ---------------------------------------------------------------- counter RNG
// stateless: value = f(seed, stream, index).  (Not TF's Philox stream: the reference's draws
// are unreproducible anyway; parity tests inject keep_mask / eps.)
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t stream, uint64_t idx)
{
    uint64_t r = mix64(mix64(seed ^ (stream * 0xD6E8FEB86659FD93ULL)) + idx);
    return (float)((r >> 40) + 0.5) * (1.0f / 16777216.0f);      // (0,1)
}
__device__ __forceinline__ float normal01(uint64_t seed, uint64_t stream, uint64_t idx)
{
    float u1 = uniform01(seed, stream, 2 * idx), u2 = uniform01(seed, stream, 2 * idx + 1);
    return sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// ---------------------------------------------------------------- prep_ids
// src/model.py:82-95 + src/util_tf.py:40-57: transpose to time-major, lengths, decoder mask,
// gold = tgt + [eos], lead = [bos] + word-dropout(tgt), and the time-major compaction index
// that tf.boolean_mask (model.py:161,174) implies.
// Two launches over the whole chip (one workgroup took 73 us at 256 x 64 and 570 us at 1024 x 128): the first does
// everything elementwise -- lengths (a wave per row), transposes, lead / gold -- and counts the kept positions of each
// contiguous chunk of the flat time-major order; the second turns the chunk counts into the compaction index (every
// workgroup sums the counts before its own chunk, then scans its chunk).
__device__ __forceinline__ bool prep_kept(const PrepArgs& p, int i)     // decoder mask of flat time-major position i
{
    const int t = i / p.B, b = i - t * p.B;
    return t == 0 || p.tgt[(size_t)b * p.St + t - 1] != p.eos;
}
__device__ __forceinline__ int block_sum256(int v, int* red)            // sum over the 256 threads, returned to all
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const int s = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return s;
}
__global__ __launch_bounds__(256) void prep_ids_kernel(PrepArgs p, int per, int nchunks)
{
    __shared__ int red[4];
    const int tid = threadIdx.x, lane = tid & 63, B = p.B, T = p.St + 1, total = T * B;
    const int nthr = gridDim.x * 256, gid = blockIdx.x * 256 + tid;
    for (int b = gid >> 6; b < B; b += nthr >> 6) {
        int ls = 0, lt = 0;
        for (int s = lane; s < p.Ss; s += 64) ls += p.src[(size_t)b * p.Ss + s] != p.eos;
        // (source: the reference's length = the COUNT of non-eos ids, model.py:84; target: the decoder mask is per position, so the
        //  steps that can reach the loss end one behind the LAST non-eos id, wherever other eos ids sit)
        for (int s = lane; s < p.St; s += 64) if (p.tgt[(size_t)b * p.St + s] != p.eos) lt = s + 1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ls += __shfl_xor(ls, o); lt = max(lt, __shfl_xor(lt, o)); }
        if (lane == 0) { p.lens_src[b] = ls; p.lens_tgt[b] = lt; }
    }
    for (int i = gid; i < p.Ss * B; i += nthr) {
        const int s = i / B, b = i - s * B;
        p.src_tm[i] = p.src[(size_t)b * p.Ss + s];
    }
    for (int i = gid; i < total; i += nthr) {
        const int t = i / B, b = i - t * B;
        p.gold[i] = t < p.St ? p.tgt[(size_t)b * p.St + t] : p.eos;
        int lead = p.bos;
        if (t > 0) {
            lead = p.tgt[(size_t)b * p.St + t - 1];
            if (p.train) {
                const int j = (t - 1) * B + b;
                const bool keep = p.keep_mask ? p.keep_mask[j] != 0 : uniform01(p.seed, 1, j) < p.keepwd;
                if (!keep) lead = 0;     // unk, model.py:94
            }
        }
        p.lead[i] = lead;
    }
    if (gid < 2 && p.zero2) p.zero2[gid] = 0.f;
    if ((int)blockIdx.x < nchunks) {
        const int beg = blockIdx.x * per, end = min(total, beg + per);
        int cnt = 0;
        for (int i = beg + tid; i < end; i += 256) cnt += prep_kept(p, i);
        cnt = block_sum256(cnt, red);
        if (tid == 0) p.chunk_counts[blockIdx.x] = cnt;
    }
}
__global__ __launch_bounds__(256) void prep_rank_kernel(PrepArgs p, int per)
{
    __shared__ int red[4];
    __shared__ int scan[256];
    const int tid = threadIdx.x, total = (p.St + 1) * p.B;
    int pre = 0;
    for (int c = tid; c < (int)blockIdx.x; c += 256) pre += p.chunk_counts[c];
    pre = block_sum256(pre, red);
    // a contiguous run of per / 256 positions per thread, exclusive scan of the run counts over the workgroup
    const int ept = per >> 8, beg = min(total, (int)blockIdx.x * per + tid * ept), end = min(total, beg + ept);
    int cnt = 0;
    for (int i = beg; i < end; ++i) cnt += prep_kept(p, i);
    scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int run = pre + scan[tid] - cnt;
    for (int i = beg; i < end; ++i) {
        const bool m = prep_kept(p, i);
        p.rank[i] = m ? run : -1;
        if (m) p.cidx[run++] = i;
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 255) p.ntok[0] = pre + scan[255];
}
hipError_t prep_ids(hipStream_t st, const PrepArgs& p)
{
    const int total = (p.St + 1) * p.B;
    int per = 2048;                                         // positions per chunk: a multiple of 256, at most kPrepChunks chunks
    while ((total + per - 1) / per > kPrepChunks) per *= 2;
    const int nchunks = (total + per - 1) / per;
    const int grid = std::max(nchunks, std::min(1024, (total + 255) / 256));
    hipLaunchKernelGGL(prep_ids_kernel, dim3(grid), dim3(256), 0, st, p, per, nchunks);
    hipLaunchKernelGGL(prep_rank_kernel, dim3(nchunks), dim3(256), 0, st, p, per);
    return hipGetLastError();
}

// ---------------------------------------------------------------- embedding gather / scatter-add
// tf.gather(embedding, ids) (src/model.py:111-112) and its gradient.  One wave per row:
// a D-float row is read/written as 16-byte pieces by consecutive lanes (whole 256-B segments).
__global__ __launch_bounds__(256) void embed_gather_kernel(const float* __restrict__ E, const int32_t* __restrict__ ids,
                                                           float* __restrict__ out, int n, int D, int V)
{
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < n; row += gridDim.x * wpb) {
        int id = ids[row];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        const float4* s = reinterpret_cast<const float4*>(E + (size_t)id * D);
        float4* d = reinterpret_cast<float4*>(out + (size_t)row * D);
        for (int c = lane; c < D / 4; c += 64) d[c] = s[c];
    }
}
hipError_t embed_gather(hipStream_t st, const float* E, const int32_t* ids, float* out, int n, int D, int V)
{
    if (n <= 0) return hipSuccess;
    int blocks = min((n + 3) / 4, 2048);
    hipLaunchKernelGGL(embed_gather_kernel, dim3(blocks), dim3(256), 0, st, E, ids, out, n, D, V);
    return hipGetLastError();
}

// Gradient of the two gathers (source and decoder-input embeddings), accumulated into dE.  One float atomic per element
// and token ran at 0.8 TB/s: a Zipf batch sends a tenth of its tokens to one row and the adders of a row serialise.  The
// tokens are grouped by id first (histogram -> offsets -> lists: three small integer kernels), then a wave sums one
// SEGMENT of at most 32 same-id rows in registers (each row one coalesced 2 KB read) and adds the sum to dE once: a plain
// read-modify-write where the id has a single segment (most ids), one atomic row-add per segment for the hot ids.
constexpr int kScatterSeg = 32;
struct Scatter2 {
    float* dE; const int32_t* ids0; const float* dout0; int n0; const int32_t* ids1; const float* dout1; int n1; int D, V;
    int32_t* cnt;      // [V]   tokens per id           (zeroed by the launcher)
    int32_t* cur;      // [V]   fill cursors            (zeroed by the launcher)
    int32_t* off;      // [V+1] first list slot of an id
    int32_t* segoff;   // [V+1] first segment of an id
    int32_t* list;     // [n0+n1] token numbers grouped by id (token t >= n0 is row t - n0 of the second source)
    int32_t* segid;    // [<= (n0+n1)/kScatterSeg + V] id of every segment
    int32_t* rank;     // optional [V]: index of an id among the ids PRESENT in the batch (-1: absent), with
    int32_t* uid;      //          [V]: the present ids in ascending order, and
    int32_t* nuniq;    //          [1]: how many there are
};
__device__ __forceinline__ int scatter_id(const Scatter2& a, int t)
{
    int id = t < a.n0 ? a.ids0[t] : a.ids1[t - a.n0];
    return id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);
}
// Counting and list filling go through a per-workgroup histogram in LDS: a Zipf batch puts a tenth of its tokens on ONE
// id, and same-address global atomics serialise at ~18 ns each (62 us for 33 K tokens, measured); in LDS the hot bin costs a
// few cycles per token and the global counter sees one add per workgroup.  kScatterLdsV bins fit the static LDS limit.
constexpr int kScatterLdsV = 12288, kScatterTok = 1024;      // tokens per workgroup (4 per thread)
__global__ __launch_bounds__(256) void scatter_hist_kernel(Scatter2 a)
{
    __shared__ int lc[kScatterLdsV];
    const int n = a.n0 + a.n1, tid = threadIdx.x, t0 = blockIdx.x * kScatterTok;
    for (int i = tid; i < a.V; i += 256) lc[i] = 0;
    __syncthreads();
    for (int t = t0 + tid; t < min(n, t0 + kScatterTok); t += 256) atomicAdd(&lc[scatter_id(a, t)], 1);
    __syncthreads();
    for (int i = tid; i < a.V; i += 256) { const int c = lc[i]; if (c) atomicAdd(a.cnt + i, c); }
}
// inclusive scan of one int per thread over a 1024-thread workgroup: wave scans by lane shuffles, the 16 wave totals by
// the first wave (two barriers instead of the twenty of a shared-memory doubling scan)
__device__ __forceinline__ int block_scan1024(int v, int* wsum /* [16] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    if (wave == 0) {
        int w = lane < 16 ? wsum[lane] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { const int t = __shfl_up(w, o, 64); if (lane >= o) w += t; }
        if (lane < 16) wsum[lane] = w;
    }
    __syncthreads();
    const int r = v + (wave ? wsum[wave - 1] : 0);
    __syncthreads();                                 // (wsum is reused by the next scan)
    return r;
}
__global__ __launch_bounds__(1024) void scatter_scan_kernel(Scatter2 a)        // one workgroup: V is a few thousand
{
    __shared__ int wsum[16];
    const int tid = threadIdx.x, per = (a.V + 1023) / 1024, beg = min(a.V, tid * per), end = min(a.V, beg + per);
    int tok = 0, seg = 0, uni = 0;
    for (int i = beg; i < end; ++i) { const int c = a.cnt[i]; tok += c; seg += (c + kScatterSeg - 1) / kScatterSeg; uni += c > 0; }
    const int itok = block_scan1024(tok, wsum), iseg = block_scan1024(seg, wsum), iuni = block_scan1024(uni, wsum);
    int rt = itok - tok, rs = iseg - seg, ru = iuni - uni;
    for (int i = beg; i < end; ++i) {
        const int c = a.cnt[i];
        a.off[i] = rt; a.segoff[i] = rs;
        rt += c;
        for (int k = (c + kScatterSeg - 1) / kScatterSeg; k > 0; --k) a.segid[rs++] = i;
        if (a.rank) { a.rank[i] = c > 0 ? ru : -1; if (c > 0) a.uid[ru++] = i; }
    }
    if (tid == 1023) { a.off[a.V] = itok; a.segoff[a.V] = iseg; if (a.rank) a.nuniq[0] = iuni; }
}
__global__ __launch_bounds__(256) void scatter_fill_kernel(Scatter2 a)
{
    __shared__ int lc[kScatterLdsV];
    const int n = a.n0 + a.n1, tid = threadIdx.x, t0 = blockIdx.x * kScatterTok;
    for (int i = tid; i < a.V; i += 256) lc[i] = 0;
    __syncthreads();
    int id[kScatterTok / 256], rk[kScatterTok / 256];          // rank of each of this thread's tokens inside the workgroup's share of its id
#pragma unroll
    for (int q = 0; q < kScatterTok / 256; ++q) {
        const int t = t0 + tid + 256 * q;
        id[q] = -1; rk[q] = 0;
        if (t < n) { id[q] = scatter_id(a, t); rk[q] = atomicAdd(&lc[id[q]], 1); }
    }
    __syncthreads();
    for (int i = tid; i < a.V; i += 256) { const int c = lc[i]; if (c) lc[i] = atomicAdd(a.cur + i, c); }     // the share's first slot
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kScatterTok / 256; ++q)
        if (id[q] >= 0) a.list[a.off[id[q]] + lc[id[q]] + rk[q]] = t0 + tid + 256 * q;
}
__global__ __launch_bounds__(256) void scatter_reduce_kernel(Scatter2 a)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6, nseg = a.segoff[a.V], D = a.D;
    for (int sg = blockIdx.x * wpb + (threadIdx.x >> 6); sg < nseg; sg += gridDim.x * wpb) {
        const int id = a.segid[sg], first = a.segoff[id], cnt = a.off[id + 1] - a.off[id];
        const int beg = a.off[id] + (sg - first) * kScatterSeg, end = min(a.off[id] + cnt, beg + kScatterSeg);
        float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};     // D <= 512: two 16-byte pieces per lane
        for (int i = beg; i < end; i += 4) {          // four rows in flight: the token numbers first, then their rows
            int t[4]; float4 v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = a.list[min(i + u, end - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* s = t[u] < a.n0 ? a.dout0 + (size_t)t[u] * D : a.dout1 + (size_t)(t[u] - a.n0) * D;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c = 4 * lane + 256 * q;
                    v[u][q] = (c < D && i + u < end) ? *reinterpret_cast<const float4*>(s + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 2; ++q) { acc[q].x += v[u][q].x; acc[q].y += v[u][q].y; acc[q].z += v[u][q].z; acc[q].w += v[u][q].w; }
        }
        float* d = a.dE + (size_t)id * D;
        const bool alone = cnt <= kScatterSeg;       // the only segment of its id: no other wave touches this row
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = 4 * lane + 256 * q;
            if (c >= D) continue;
            if (alone) {
                float4 v = *reinterpret_cast<float4*>(d + c);
                v.x += acc[q].x; v.y += acc[q].y; v.z += acc[q].z; v.w += acc[q].w;
                *reinterpret_cast<float4*>(d + c) = v;
            } else {
                atomicAdd(d + c, acc[q].x); atomicAdd(d + c + 1, acc[q].y); atomicAdd(d + c + 2, acc[q].z); atomicAdd(d + c + 3, acc[q].w);
            }
        }
    }
}
// fallback (vocabularies beyond the LDS histogram, rows longer than two pieces per lane): one float atomic per element
__global__ __launch_bounds__(256) void embed_scatter_atomic_kernel(float* __restrict__ dE, const int32_t* __restrict__ ids,
                                                                   const float* __restrict__ dout, int n, int D, int V)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < n; row += gridDim.x * wpb) {
        int id = ids[row];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        const float* s = dout + (size_t)row * D;
        float* d = dE + (size_t)id * D;
        for (int c = lane; c < D; c += 64) atomicAdd(d + c, s[c]);
    }
}
// zero fill as a plain kernel (hipMemsetAsync goes through the runtime's blit path: ~10 us of stream gap per call, several
// calls per step); p 16-byte aligned, bytes a multiple of 4
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* __restrict__ p16, size_t n16, unsigned* __restrict__ tail, int ntail)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p16[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0u;
}
hipError_t zero_fill(hipStream_t st, void* p, size_t bytes)
{
    if (!bytes) return hipSuccess;
    if (((uintptr_t)p & 15) || (bytes & 3)) return hipMemsetAsync(p, 0, bytes, st);
    const size_t n16 = bytes >> 4; const int ntail = (int)((bytes & 15) >> 2);
    const size_t want = (n16 + 255) / 256;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)std::max<size_t>(1, std::min<size_t>(want, 2048))), dim3(256), 0, st,
                       reinterpret_cast<uint4*>(p), n16, reinterpret_cast<unsigned*>(p) + 4 * n16, ntail);
    return hipGetLastError();
}

hipError_t embed_scatter_add2(hipStream_t st, float* dE, const int32_t* ids0, const float* dout0, int n0, const int32_t* ids1,
                              const float* dout1, int n1, int D, int V, int32_t* scratch)
{
    const int n = n0 + n1;
    if (n <= 0) return hipSuccess;
    if (D > 512 || (D & 3) || V > kScatterLdsV) {
        if (n0 > 0) hipLaunchKernelGGL(embed_scatter_atomic_kernel, dim3(std::min((n0 + 3) / 4, 4096)), dim3(256), 0, st, dE, ids0, dout0, n0, D, V);
        if (n1 > 0) hipLaunchKernelGGL(embed_scatter_atomic_kernel, dim3(std::min((n1 + 3) / 4, 4096)), dim3(256), 0, st, dE, ids1, dout1, n1, D, V);
        return hipGetLastError();
    }
    Scatter2 a{dE, ids0, dout0, n0, ids1, dout1, n1, D, V, scratch, scratch + V, scratch + 2 * V, scratch + 3 * V + 1, scratch + 4 * V + 2, scratch + 4 * V + 2 + n, nullptr, nullptr, nullptr};
    hipError_t e = zero_fill(st, scratch, sizeof(int32_t) * 2 * (size_t)V);
    if (e != hipSuccess) return e;
    const int blocks = (n + kScatterTok - 1) / kScatterTok;
    hipLaunchKernelGGL(scatter_hist_kernel, dim3(blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(scatter_scan_kernel, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(scatter_fill_kernel, dim3(blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(scatter_reduce_kernel, dim3(std::min((n / 4 + V + 3) / 4, 4096)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------- token groups of an id source; rows by id and their transpose
// A layer whose input is an embedding row computes W E[id].  With more tokens than vocabulary entries the projection is
// taken once over the ids PRESENT in the batch (U <= V rows: E rows gathered by `uid`, one GEMM with a device-side row
// count) and the per-token result is a row gather through `rank` (rows_gather_ranked); in the backward the per-token
// gradients are summed by id first (rows_group_sum: the grouped scatter above on wide rows, written to row rank[id]) and the
// two weight-side products run over U rows instead of over every token.  Exact algebra: the same products, grouped by id.
// id_groups_build runs once per step and source (forward); the lists serve the backward of the same step.
static Scatter2 id_groups_view(const int32_t* ids, int n, int V, int32_t* scratch)
{
    int32_t* rank = scratch + 4 * (size_t)V + 2 + n + ((size_t)n / kScatterSeg + V + 1);
    return Scatter2{nullptr, ids, nullptr, n, nullptr, nullptr, 0, 0, V, scratch, scratch + V, scratch + 2 * V, scratch + 3 * V + 1,
                    scratch + 4 * V + 2, scratch + 4 * (size_t)V + 2 + n, rank, rank + V, rank + 2 * (size_t)V};
}
const int32_t* id_groups_rank(int32_t* scratch, int n, int V) { return id_groups_view(nullptr, n, V, scratch).rank; }
const int32_t* id_groups_uid(int32_t* scratch, int n, int V) { return id_groups_view(nullptr, n, V, scratch).uid; }
const int32_t* id_groups_count(int32_t* scratch, int n, int V) { return id_groups_view(nullptr, n, V, scratch).nuniq; }
bool id_groups_supported(int V) { return V <= kScatterLdsV; }
hipError_t id_groups_build(hipStream_t st, const int32_t* ids, int n, int V, int32_t* scratch, bool lists)
{
    if (n <= 0 || V > kScatterLdsV) return hipErrorInvalidValue;
    Scatter2 a = id_groups_view(ids, n, V, scratch);
    hipError_t e = zero_fill(st, scratch, sizeof(int32_t) * 2 * (size_t)V);
    if (e != hipSuccess) return e;
    const int blocks = (n + kScatterTok - 1) / kScatterTok;
    hipLaunchKernelGGL(scatter_hist_kernel, dim3(blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(scatter_scan_kernel, dim3(1), dim3(1024), 0, st, a);
    if (lists) hipLaunchKernelGGL(scatter_fill_kernel, dim3(blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void rows_gather_ranked_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                                 const int32_t* __restrict__ ids, const int32_t* __restrict__ rank, int n, int W, int V)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < n; row += gridDim.x * wpb) {
        int id = ids[row];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        const float4* s = reinterpret_cast<const float4*>(src + (size_t)rank[id] * W);
        float4* d = reinterpret_cast<float4*>(dst + (size_t)row * W);
        for (int c = lane; c < W / 4; c += 64) d[c] = s[c];
    }
}
__global__ __launch_bounds__(256) void rank_rows_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ ids, const int32_t* __restrict__ rank, int n, int V)
{
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
        int id = ids[t];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        out[t] = rank[id];
    }
}
hipError_t rank_rows(hipStream_t st, int32_t* out, const int32_t* ids, const int32_t* rank, int n, int V)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(rank_rows_kernel, dim3(std::min((n + 255) / 256, 2048)), dim3(256), 0, st, out, ids, rank, n, V);
    return hipGetLastError();
}
hipError_t rows_gather_ranked(hipStream_t st, float* dst, const float* src, const int32_t* ids, const int32_t* rank, int n, int W, int V)
{
    if (n <= 0) return hipSuccess;
    if (W & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rows_gather_ranked_kernel, dim3(std::min((n + 3) / 4, 8192)), dim3(256), 0, st, dst, src, ids, rank, n, W, V);
    return hipGetLastError();
}
// work item = (segment, block of 256 columns); dst was zero-filled: a single-segment id stores, a hot id's segments add
__global__ __launch_bounds__(256) void scatter_reduce_wide_kernel(Scatter2 a, float* __restrict__ dst, const float* __restrict__ src, int W)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6, nseg = a.segoff[a.V], ncb = (W + 255) / 256;
    const long long items = (long long)nseg * ncb;
    for (long long it = (long long)blockIdx.x * wpb + (threadIdx.x >> 6); it < items; it += (long long)gridDim.x * wpb) {
        const int sg = (int)(it / ncb), c = (int)(it - (long long)sg * ncb) * 256 + 4 * lane;
        const int id = a.segid[sg], first = a.segoff[id], cnt = a.off[id + 1] - a.off[id];
        const int beg = a.off[id] + (sg - first) * kScatterSeg, end = min(a.off[id] + cnt, beg + kScatterSeg);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = beg; i < end; i += 4) {
            int t[4]; float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = a.list[min(i + u, end - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = (c < W && i + u < end) ? *reinterpret_cast<const float4*>(src + (size_t)t[u] * W + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        if (c >= W) continue;
        float* d = dst + (size_t)a.rank[id] * W + c;
        if (cnt <= kScatterSeg) *reinterpret_cast<float4*>(d) = acc;
        else { atomicAdd(d, acc.x); atomicAdd(d + 1, acc.y); atomicAdd(d + 2, acc.z); atomicAdd(d + 3, acc.w); }
    }
}
// only the rows that several segments ADD into need zeros first (the hot ids: a few rows); every other present row is stored
__global__ __launch_bounds__(256) void zero_hot_rows_kernel(Scatter2 a, float* __restrict__ dst, int W)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int id = blockIdx.x * wpb + (threadIdx.x >> 6); id < a.V; id += gridDim.x * wpb) {
        if (a.off[id + 1] - a.off[id] <= kScatterSeg) continue;
        float4* d = reinterpret_cast<float4*>(dst + (size_t)a.rank[id] * W);
        for (int c = lane; c < W / 4; c += 64) d[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// dst (rows of the present ids, W floats each; rows beyond their count are left untouched) = per-id sums of src's rows
hipError_t rows_group_sum(hipStream_t st, float* dst, const int32_t* ids, const float* src, int n, int W, int V, int32_t* scratch)
{
    if ((W & 3) || n <= 0 || V > kScatterLdsV) return hipErrorInvalidValue;
    Scatter2 a = id_groups_view(ids, n, V, scratch);
    hipLaunchKernelGGL(zero_hot_rows_kernel, dim3(std::min((V + 3) / 4, 2048)), dim3(256), 0, st, a, dst, W);
    const long long items = ((long long)n / 8 + V) * ((W + 255) / 256);
    hipLaunchKernelGGL(scatter_reduce_wide_kernel, dim3((unsigned)std::min<long long>((items + 3) / 4, 8192)), dim3(256), 0, st, a, dst, src, W);
    return hipGetLastError();
}
// dst[uid[r], :] += src[r, :] for r < *nuniq (distinct rows: a plain read-modify-write)
__global__ __launch_bounds__(256) void rows_add_indexed_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                               const int32_t* __restrict__ uid, const int32_t* __restrict__ nuniq, int n_max, int D)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6, n = min(n_max, *nuniq);
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < n; row += gridDim.x * wpb) {
        const float4* s = reinterpret_cast<const float4*>(src + (size_t)row * D);
        float4* d = reinterpret_cast<float4*>(dst + (size_t)uid[row] * D);
        for (int c = lane; c < D / 4; c += 64) { float4 v = d[c]; const float4 x = s[c]; v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w; d[c] = v; }
    }
}
hipError_t rows_add_indexed(hipStream_t st, float* dst, const float* src, const int32_t* uid, const int32_t* nuniq, int n_max, int D)
{
    if (n_max <= 0) return hipSuccess;
    if (D & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rows_add_indexed_kernel, dim3(std::min((n_max + 3) / 4, 4096)), dim3(256), 0, st, dst, src, uid, nuniq, n_max, D);
    return hipGetLastError();
}

// ---------------------------------------------------------------- compaction (tf.boolean_mask, model.py:161)
__global__ __launch_bounds__(256) void rows_gather_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                          const int32_t* __restrict__ idx, const int32_t* __restrict__ n_dev,
                                                          int n_max, int D, const int32_t* __restrict__ map)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int n = min(n_max, *n_dev);
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < n; row += gridDim.x * wpb) {
        const int sr = map ? map[idx[row]] : idx[row];          // (map: src is stored over the real rows only, row_map)
        const float4* s = reinterpret_cast<const float4*>(src + (size_t)sr * D);
        float4* d = reinterpret_cast<float4*>(dst + (size_t)row * D);
        for (int c = lane; c < D / 4; c += 64) d[c] = s[c];
    }
}
hipError_t rows_gather(hipStream_t st, float* dst, const float* src, const int32_t* idx, const int32_t* n_dev, int n_max, int D, const int32_t* map)
{
    if (n_max <= 0) return hipSuccess;
    hipLaunchKernelGGL(rows_gather_kernel, dim3(min((n_max + 3) / 4, 2048)), dim3(256), 0, st, dst, src, idx, n_dev, n_max, D, map);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void rows_expand_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                          const int32_t* __restrict__ rank, int rows, int D, const int32_t* __restrict__ map)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        int r = rank[row];
        const int dr = map ? map[row] : row;                    // (map: dst is stored over the real rows only; padding has no row)
        if (dr < 0) continue;
        float4* d = reinterpret_cast<float4*>(dst + (size_t)dr * D);
        if (r >= 0) {
            const float4* s = reinterpret_cast<const float4*>(src + (size_t)r * D);
            for (int c = lane; c < D / 4; c += 64) d[c] = s[c];
        } else {
            for (int c = lane; c < D / 4; c += 64) d[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
hipError_t rows_expand(hipStream_t st, float* dst, const float* src, const int32_t* rank, int rows, int D, const int32_t* map)
{
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(rows_expand_kernel, dim3(min((rows + 3) / 4, 2048)), dim3(256), 0, st, dst, src, rank, rows, D, map);
    return hipGetLastError();
}

// ---------------------------------------------------------------- final-state pick (tf.gather_nd, model.py:135)
// ---------------------------------------------------------------- row order for the padding-skipping GRU kernels
struct RowOrders { RowOrder o[3]; };
__global__ __launch_bounds__(1024) void row_order_kernel(RowOrders ro, int Breal, int B, int S, int32_t* __restrict__ steps_sum, int sum_rows)
{
    extern __shared__ int steps[];                       // B ints
    const RowOrder o = ro.o[blockIdx.x];
    // (B > Breal: the launch geometry has more slots than the batch has rows -- the rows beyond are PHANTOM rows of one step, sorted
    //  behind every real row; GruArgs::Bx)
    for (int b = threadIdx.x; b < B; b += blockDim.x) steps[b] = b < Breal ? min(max(o.lens[b] + o.add, 1), S) : 1;
    __syncthreads();
    if (steps_sum && blockIdx.x == 0 && threadIdx.x < 64) {      // real positions of the first order's id source (the host's fill hint)
        int v = 0;
        for (int b = threadIdx.x; b < Breal; b += 64) v += steps[b];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (threadIdx.x == 0) { steps_sum[0] = v; steps_sum[1] = sum_rows; }      // (real positions, padded positions)
    }
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int mine = steps[b];
        int rank = 0;                                     // rows with more steps, or as many and a smaller index: stable, descending
        for (int c = 0; c < B; ++c) { const int v = steps[c]; rank += (v > mine) || (v == mine && c < b); }
        const int g = rank >> 4, w = g % o.cpj, jj = g / o.cpj;
        const int slot = (((w + (jj / o.T) * o.cpj) * o.T + jj % o.T) << 4) + (rank & 15);
        o.perm[slot] = b;
        o.slens[slot] = mine;
    }
}
hipError_t row_order(hipStream_t st, const RowOrder* orders, int n, int Breal, int B, int S, int32_t* steps_sum, int sum_rows)
{
    if (n <= 0) return hipSuccess;
    if (n > 3 || B % 16 || B < Breal || (size_t)B * 4 > 64 * 1024) return hipErrorInvalidValue;
    RowOrders ro{};
    for (int i = 0; i < n; ++i) ro.o[i] = orders[i];
    hipLaunchKernelGGL(row_order_kernel, dim3(n), dim3(1024), (size_t)B * sizeof(int), st, ro, Breal, B, S, steps_sum, sum_rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------- compact row map of an id source
// map[t * B + b] = the place of position (t, b) among the real positions (t < lens[b] + add) in time-major order, -1 for
// padding; count[0] = how many.  Every array between the encoder's GEMMs and GRU launches is then stored over the real rows
// only (GruArgs::rowmap): the padded rows of a ragged batch cost nothing anywhere.
__global__ __launch_bounds__(256) void row_map_local_kernel(const int32_t* __restrict__ lens, int add, int B, int32_t* __restrict__ map, int32_t* __restrict__ nact)
{
    __shared__ int scan[256];
    __shared__ int base_s;
    const int t = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int b = b0 + tid;
        const int act = (b < B && t < lens[b] + add) ? 1 : 0;
        scan[tid] = act;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int v = tid >= off ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const int base = base_s;
        if (b < B) map[(size_t)t * B + b] = act ? base + scan[tid] - 1 : -1;
        __syncthreads();
        if (tid == 255) base_s = base + scan[255];
        __syncthreads();
    }
    if (tid == 0) nact[t] = base_s;
}
__global__ __launch_bounds__(256) void row_map_offset_kernel(int B, int S, int32_t* __restrict__ map, const int32_t* __restrict__ nact, int32_t* __restrict__ count)
{
    __shared__ int red[4];
    const int t = blockIdx.x, tid = threadIdx.x;
    int pre = 0;
    for (int c = tid; c < t; c += 256) pre += nact[c];
    pre = block_sum256(pre, red);
    __shared__ int off_s;
    if (tid == 0) off_s = pre;
    __syncthreads();
    const int off = off_s;
    for (int b = tid; b < B; b += 256) { const int v = map[(size_t)t * B + b]; if (v >= 0) map[(size_t)t * B + b] = v + off; }
    if (t == S - 1 && tid == 0) count[0] = off + nact[t];
}
hipError_t row_map(hipStream_t st, const int32_t* lens, int add, int S, int B, int32_t* map, int32_t* nact, int32_t* count)
{
    hipLaunchKernelGGL(row_map_local_kernel, dim3(S), dim3(256), 0, st, lens, add, B, map, nact);
    hipLaunchKernelGGL(row_map_offset_kernel, dim3(S), dim3(256), 0, st, B, S, map, nact, count);
    return hipGetLastError();
}
// X[r, :] = 0 for r < min(rows_max, *count)
__global__ __launch_bounds__(256) void zero_rows_dyn_kernel(float4* __restrict__ X, const int32_t* __restrict__ count, int rows_max, int W4)
{
    const size_t n = (size_t)min(rows_max, *count) * W4, stride = (size_t)gridDim.x * blockDim.x;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) X[i] = z;
}
hipError_t zero_rows_dyn(hipStream_t st, float* X, const int32_t* count, int rows_max, int W)
{
    hipLaunchKernelGGL(zero_rows_dyn_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<float4*>(X), count, rows_max, W / 4);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void pick_last_kernel(float* __restrict__ h, const float* __restrict__ hs,
                                                        const int32_t* __restrict__ lens, int B, int W, const int32_t* __restrict__ map)
{
    const int b = blockIdx.x;
    int t = max(lens[b] - 1, 0);
    // A row without a single non-eos id (len 0; compact layout: no place among the real positions, map = -1): the reference is
    // undefined there (gather_nd at index -1, model.py:135).  The build picks zeros and passes no gradient, in every layout --
    // never another sentence's row
    const long long mrow = map ? (long long)map[(size_t)t * B + b] : (long long)t * B + b;
    float4* d = reinterpret_cast<float4*>(h + (size_t)b * W);
    if (mrow < 0 || lens[b] <= 0) { for (int c = threadIdx.x; c < W / 4; c += blockDim.x) d[c] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const float4* s = reinterpret_cast<const float4*>(hs + (size_t)mrow * W);
    for (int c = threadIdx.x; c < W / 4; c += blockDim.x) d[c] = s[c];
}
hipError_t pick_last(hipStream_t st, float* h, const float* hs, const int32_t* lens, int B, int W, const int32_t* map)
{
    hipLaunchKernelGGL(pick_last_kernel, dim3(B), dim3(256), 0, st, h, hs, lens, B, W, map);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void pick_last16_kernel(float* __restrict__ h, const unsigned short* __restrict__ hs, const int32_t* __restrict__ lens, int B, int W,
                                                          const int32_t* __restrict__ map)
{
    const int b = blockIdx.x;
    const int t = max(lens[b] - 1, 0);
    const long long mrow = map ? (long long)map[(size_t)t * B + b] : (long long)t * B + b;      // (-1: see pick_last_kernel)
    float4* d = reinterpret_cast<float4*>(h + (size_t)b * W);
    if (mrow < 0 || lens[b] <= 0) { for (int c = threadIdx.x; c < W / 4; c += blockDim.x) d[c] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const uint2* s = reinterpret_cast<const uint2*>(hs + (size_t)mrow * W);      // four bf16 per access
    for (int c = threadIdx.x; c < W / 4; c += blockDim.x) {
        const uint2 v = s[c];
        d[c] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    }
}
hipError_t pick_last16(hipStream_t st, float* h, const unsigned short* hs16, const int32_t* lens, int B, int W, const int32_t* map)
{
    hipLaunchKernelGGL(pick_last16_kernel, dim3(B), dim3(256), 0, st, h, hs16, lens, B, W, map);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void pick_last_add_kernel(float* __restrict__ dhs, const float* __restrict__ d,
                                                            const int32_t* __restrict__ lens, int B, int W, const int32_t* __restrict__ map)
{
    const int b = blockIdx.x;
    const int t = max(lens[b] - 1, 0);
    const long long mrow = map ? (long long)map[(size_t)t * B + b] : (long long)t * B + b;
    if (mrow < 0 || lens[b] <= 0) return;                   // (a row without a real position: nothing to add to, see pick_last_kernel)
    float4* dst = reinterpret_cast<float4*>(dhs + (size_t)mrow * W);
    const float4* s = reinterpret_cast<const float4*>(d + (size_t)b * W);
    for (int c = threadIdx.x; c < W / 4; c += blockDim.x) {
        float4 v = dst[c]; const float4 a = s[c];
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        dst[c] = v;
    }
}
hipError_t pick_last_add(hipStream_t st, float* dhs, const float* d, const int32_t* lens, int B, int W, const int32_t* map)
{
    hipLaunchKernelGGL(pick_last_add_kernel, dim3(B), dim3(256), 0, st, dhs, d, lens, B, W, map);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void pick_last_bwd_kernel(float* __restrict__ dhs, const float* __restrict__ dh,
                                                            const int32_t* __restrict__ lens, int S, int B, int W)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int rows = S * B;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        int s = row / B, b = row - s * B;
        float4* d = reinterpret_cast<float4*>(dhs + (size_t)row * W);
        if (s == lens[b] - 1) {       // (a row of length 0 has no such position: see pick_last_kernel)
            const float4* src = reinterpret_cast<const float4*>(dh + (size_t)b * W);
            for (int c = lane; c < W / 4; c += 64) d[c] = src[c];
        } else {
            for (int c = lane; c < W / 4; c += 64) d[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
hipError_t pick_last_bwd(hipStream_t st, float* dhs, const float* dh, const int32_t* lens, int S, int B, int W)
{
    hipLaunchKernelGGL(pick_last_bwd_kernel, dim3(min((S * B + 3) / 4, 2048)), dim3(256), 0, st, dhs, dh, lens, S, B, W);
    return hipGetLastError();
}

// ---------------------------------------------------------------- latent (model.py:151-155,183-184)
__device__ __forceinline__ float block_sum(float v, float* sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
    return s;    // valid on thread 0
}
struct LatentK { const float* mu; const float* lv; const float* eps_in; float* eps_out; float* z; float* kld; int n;
                 int train; uint64_t seed; float free_bits; float* acc; };
__global__ __launch_bounds__(256) void latent_fwd_kernel(LatentK a)
{
    __shared__ float sh[4];
    float part = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
        float mu = a.mu[i], lv = a.lv[i];
        float z = mu;
        if (a.train) {
            float e = a.eps_in ? a.eps_in[i] : normal01(a.seed, 2, i);
            if (a.eps_out) a.eps_out[i] = e;
            z = mu + expf(0.5f * lv) * e;
        }
        a.z[i] = z;
        float k = 0.5f * (mu * mu + expf(lv) - lv - 1.f);
        if (a.kld) a.kld[i] = k;
        part += fmaxf(k, a.free_bits);
    }
    float s = block_sum(part, sh);
    if (threadIdx.x == 0 && a.acc) atomicAdd(a.acc, s);
}
hipError_t latent_fwd(hipStream_t st, const float* mu, const float* lv, const float* eps_in, float* eps_out, float* z,
                       float* kld, int n, int train, uint64_t seed, float free_bits, float* acc)
{
    LatentK k{mu, lv, eps_in, eps_out, z, kld, n, train, seed, free_bits, acc};
    hipLaunchKernelGGL(latent_fwd_kernel, dim3(min((n + 255) / 256, 1024)), dim3(256), 0, st, k);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ mu,
                                                         const float* __restrict__ lv, const float* __restrict__ eps,
                                                         float* __restrict__ dmu, float* __restrict__ dlv, int n,
                                                         float coef, float free_bits)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float m = mu[i], l = lv[i], g = dz[i];
        float el = expf(l);
        float k = 0.5f * (m * m + el - l - 1.f);
        float c = (k >= free_bits) ? coef : 0.f;
        dmu[i] = g + c * m;
        dlv[i] = (eps ? g * eps[i] * 0.5f * expf(0.5f * l) : 0.f) + c * 0.5f * (el - 1.f);
    }
}
hipError_t latent_bwd(hipStream_t st, const float* dz, const float* mu, const float* lv, const float* eps_used,
                      float* dmu, float* dlv, int B, int R, float coef, float free_bits)
{
    int n = B * R;
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(min((n + 255) / 256, 1024)), dim3(256), 0, st, dz, mu, lv, eps_used, dmu, dlv, n, coef, free_bits);
    return hipGetLastError();
}

// ---------------------------------------------------------------- softmax cross-entropy (model.py:170-181)
// softmax cross-entropy, argmax and error flags of one compact row per workgroup (model.py:170-181); optionally the
// gradient (softmax - onehot) * scale in place of the logits, or as the bf16 GEMM operand (grad16).
// The V <= 8192 form (row in registers).  It was VALU-bound, not HBM-bound: the online max / sum-exp update evaluated two
// libm expf per element and the gradient pass a third (~80 instructions per element, 214 us for 545 MB).  Now: pass A finds
// the row maximum and first argmax (no exponential), pass B replaces every kept logit by exp2((x - m) log2 e) -- ONE v_exp_f32
// per element -- and sums, pass C writes kept * scale / sum - onehot * scale.  exp2 of a non-positive argument on the hardware
// unit is accurate to 1 ulp; the loss uses m + log(sum).
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float4 half4_to_float4(uint2 v)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 lo = __builtin_bit_cast(h2, v.x), hi = __builtin_bit_cast(h2, v.y);
    return make_float4((float)lo[0], (float)lo[1], (float)hi[0], (float)hi[1]);
}
__global__ __launch_bounds__(256) void softmax_ce_reg_kernel(CeArgs a)
{
    __shared__ float s_m[4], s_s[4]; __shared__ int s_bi[4];
    __shared__ float s_acc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(a.n_max, *a.n_dev);
    const float scale = a.inv_n > 0.f ? a.inv_n : 1.f / (float)max(n, 1);
    if (tid == 0) s_acc = 0.f;
    __syncthreads();
    for (int row = blockIdx.x; row < n; row += gridDim.x) {
        float* x = a.logits + (size_t)row * a.V;
        const _Float16* x16 = reinterpret_cast<const _Float16*>(a.grad16) + (size_t)row * a.V;      // logits16: the row as the GEMM left it, fp16, where the gradient goes
        const int label = a.gold[a.cidx[row]];
        float4 keep[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = tid * 4 + 1024 * q;
            if (a.logits16) keep[q] = c < a.V ? half4_to_float4(*reinterpret_cast<const uint2*>(x16 + c)) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            else
            keep[q] = c < a.V ? *reinterpret_cast<const float4*>(x + c) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
        const float xl = a.logits16 ? (float)x16[label] : x[label];                       // (before anything overwrites the row)
        // pass A: maximum and its first position
        float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = tid * 4 + 1024 * q;
            const float e[4] = {keep[q].x, keep[q].y, keep[q].z, keep[q].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) if (e[k] > bv) { bv = e[k]; bi = c + k; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_down(bv, o, 64); const int oi = __shfl_down(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();                                 // (s_* of the previous row have been read)
        if (lane == 0) { s_m[wave] = bv; s_bi[wave] = bi; }
        __syncthreads();
        bv = s_m[0]; bi = s_bi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) if (s_m[w] > bv || (s_m[w] == bv && s_bi[w] < bi)) { bv = s_m[w]; bi = s_bi[w]; }
        const float m = bv;
        // pass B: one exponential per element, kept
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            keep[q].x = exp_fast(keep[q].x - m); keep[q].y = exp_fast(keep[q].y - m);
            keep[q].z = exp_fast(keep[q].z - m); keep[q].w = exp_fast(keep[q].w - m);
            s += (keep[q].x + keep[q].y) + (keep[q].z + keep[q].w);
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) s_s[wave] = s;
        __syncthreads();
        s = (s_s[0] + s_s[1]) + (s_s[2] + s_s[3]);
        if (tid == 0) {
            const float loss = m + logf(s) - xl;
            if (a.loss_samp) a.loss_samp[row] = loss;
            if (a.errt_samp) a.errt_samp[row] = (label != bi) ? 1.f : 0.f;
            if (a.pred) a.pred[row] = bi;
            s_acc += loss;
        }
        if (a.write_grad) {
            const float g = scale / s;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = tid * 4 + 1024 * q;
                if (c >= a.V) break;
                float e[4] = {keep[q].x, keep[q].y, keep[q].z, keep[q].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = e[k] * g - ((c + k) == label ? scale : 0.f);
                if (a.grad16) {      // bf16 mode: the GEMM operand panel itself (RNE, what the conversion pass would have written)
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const f2 lo = {e[0], e[1]}, hi = {e[2], e[3]};
                    const uint2 pk = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf2)), __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf2)));
                    *reinterpret_cast<uint2*>(a.grad16 + (size_t)row * a.V + c) = pk;
                } else
                *reinterpret_cast<float4*>(x + c) = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    }
    __syncthreads();
    if (tid == 0 && a.loss_acc) atomicAdd(a.loss_acc, s_acc);
}

// logits16 (bf16 mode, V <= 8192, V % 8 == 0): the row is the fp16 panel the phased GEMM left where the bf16 gradient goes.  Half the bytes per row
// means half the loads in flight per workgroup, and the plain one-row-at-a-time form above turned latency bound on it (1.30 ms against 1.21 ms on
// the 4-byte logits at configs[2]): here a thread moves 16 bytes per access, and the NEXT row's panel and label are fetched as soon as this row's
// values have left the staging registers -- the three reductions and barriers of a row run under the next row's loads.  The label's logit comes
// from the owning thread's registers (through LDS), not from a second global read.
__global__ __launch_bounds__(256) void softmax_ce_h16_kernel(CeArgs a)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    __shared__ float s_m[4], s_s[4], s_xl; __shared__ int s_bi[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(a.n_max, *a.n_dev);
    const float scale = a.inv_n > 0.f ? a.inv_n : 1.f / (float)max(n, 1);
    uint4 raw[4]; int label = 0;
    auto fetch = [&](int row) __attribute__((always_inline)) {
        const uint4* p = reinterpret_cast<const uint4*>(a.grad16 + (size_t)row * a.V);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tid * 8 + 2048 * q;
            raw[q] = c < a.V ? p[c >> 3] : make_uint4(0xFC00FC00u, 0xFC00FC00u, 0xFC00FC00u, 0xFC00FC00u);      // (-inf, -inf)
        }
        label = a.gold[a.cidx[row]];
    };
    int row = blockIdx.x;
    if (row < n) fetch(row);
    for (; row < n; row += gridDim.x) {
        float e[32];
        const int lab = label;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned w[4] = {raw[q].x, raw[q].y, raw[q].z, raw[q].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { const h2 v = __builtin_bit_cast(h2, w[k]); e[q * 8 + 2 * k] = (float)v[0]; e[q * 8 + 2 * k + 1] = (float)v[1]; }
        }
        if (row + (int)gridDim.x < n) fetch(row + gridDim.x);
        // pass A: maximum and its first position
        float bv = -INFINITY; int bi = 0x7fffffff; float xl = 0.f; bool mine = false;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = tid * 8 + 2048 * q + k;
                if (e[q * 8 + k] > bv) { bv = e[q * 8 + k]; bi = c; }
                if (c == lab) { xl = e[q * 8 + k]; mine = true; }
            }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_down(bv, o, 64); const int oi = __shfl_down(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();                                 // (s_* of the previous row have been read)
        if (lane == 0) { s_m[wave] = bv; s_bi[wave] = bi; }
        if (mine) s_xl = xl;
        __syncthreads();
        bv = s_m[0]; bi = s_bi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) if (s_m[w] > bv || (s_m[w] == bv && s_bi[w] < bi)) { bv = s_m[w]; bi = s_bi[w]; }
        const float m = bv;
        // pass B: one exponential per element, kept
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { e[q * 8 + k] = exp_fast(e[q * 8 + k] - m); t += e[q * 8 + k]; }
            s += t;
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) s_s[wave] = s;
        __syncthreads();
        s = (s_s[0] + s_s[1]) + (s_s[2] + s_s[3]);
        if (tid == 0) {
            const float loss = m + logf(s) - s_xl;
            if (a.loss_samp) a.loss_samp[row] = loss;
            if (a.errt_samp) a.errt_samp[row] = (lab != bi) ? 1.f : 0.f;
            if (a.pred) a.pred[row] = bi;
        }
        // pass C: (softmax - onehot) * scale as bf16 (RNE), over the panel
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const float g = scale / s;
        uint4* const out = reinterpret_cast<uint4*>(a.grad16 + (size_t)row * a.V);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tid * 8 + 2048 * q;
            if (c >= a.V) break;
            unsigned pk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f2 v = {e[q * 8 + 2 * k] * g - ((c + 2 * k) == lab ? scale : 0.f), e[q * 8 + 2 * k + 1] * g - ((c + 2 * k + 1) == lab ? scale : 0.f)};
                pk[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
            }
            out[c >> 3] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
}

// the streaming form for V > 8192 (row re-read for the gradient pass; online max / sum-exp)
__global__ __launch_bounds__(256) void softmax_ce_kernel(CeArgs a)
{
    __shared__ float s_m[4], s_s[4], s_bv[4]; __shared__ int s_bi[4];
    __shared__ float s_acc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(a.n_max, *a.n_dev);
    const float scale = a.inv_n > 0.f ? a.inv_n : 1.f / (float)max(n, 1);
    if (tid == 0) s_acc = 0.f;
    __syncthreads();
    for (int row = blockIdx.x; row < n; row += gridDim.x) {
        float* x = a.logits + (size_t)row * a.V;
        const _Float16* x16 = reinterpret_cast<const _Float16*>(a.grad16) + (size_t)row * a.V;      // (logits16, see the register form)
        const int label = a.gold[a.cidx[row]];
        float m = -INFINITY, s = 0.f, bv = -INFINITY; int bi = 0x7fffffff;
        for (int c = tid * 4; c < a.V; c += 1024) {
            const float4 v = a.logits16 ? half4_to_float4(*reinterpret_cast<const uint2*>(x16 + c)) : *reinterpret_cast<const float4*>(x + c);
            float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (e[k] > bv) { bv = e[k]; bi = c + k; }
                float nm = fmaxf(m, e[k]);
                s = s * expf(m - nm) + expf(e[k] - nm);
                m = nm;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            float om = __shfl_down(m, o, 64), os = __shfl_down(s, o, 64);
            float ov = __shfl_down(bv, o, 64); int oi = __shfl_down(bi, o, 64);
            float nm = fmaxf(m, om);
            s = (nm == -INFINITY) ? 0.f : s * expf(m - nm) + os * expf(om - nm);
            m = nm;
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { s_m[wave] = m; s_s[wave] = s; s_bv[wave] = bv; s_bi[wave] = bi; }
        __syncthreads();
        m = s_m[0]; s = s_s[0]; bv = s_bv[0]; bi = s_bi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            float nm = fmaxf(m, s_m[w]);
            s = s * expf(m - nm) + s_s[w] * expf(s_m[w] - nm);
            m = nm;
            if (s_bv[w] > bv || (s_bv[w] == bv && s_bi[w] < bi)) { bv = s_bv[w]; bi = s_bi[w]; }
        }
        const float lse = m + logf(s);
        if (tid == 0) {
            float loss = lse - (a.logits16 ? (float)x16[label] : x[label]);
            if (a.loss_samp) a.loss_samp[row] = loss;
            if (a.errt_samp) a.errt_samp[row] = (label != bi) ? 1.f : 0.f;
            if (a.pred) a.pred[row] = bi;
            s_acc += loss;
        }
        if (a.write_grad) {
            __syncthreads();     // x[label] read before overwrite
            for (int c = tid * 4; c < a.V; c += 1024) {
                const float4 v = a.logits16 ? half4_to_float4(*reinterpret_cast<const uint2*>(x16 + c)) : *reinterpret_cast<const float4*>(x + c);
                float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = (expf(e[k] - lse) - ((c + k) == label ? 1.f : 0.f)) * scale;
                if (a.grad16) {      // bf16 mode: the GEMM operand panel itself (RNE, what the conversion pass would have written)
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const f2 lo = {e[0], e[1]}, hi = {e[2], e[3]};
                    const uint2 pk = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf2)), __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf2)));
                    *reinterpret_cast<uint2*>(a.grad16 + (size_t)row * a.V + c) = pk;
                } else
                *reinterpret_cast<float4*>(x + c) = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    }
    __syncthreads();
    if (tid == 0 && a.loss_acc) atomicAdd(a.loss_acc, s_acc);
}
hipError_t softmax_ce(hipStream_t st, const CeArgs& a)
{
    if (a.n_max <= 0) return hipSuccess;
    if (a.V & 3) return hipErrorInvalidValue;
    if (a.logits16 && (!a.grad16 || !a.write_grad)) return hipErrorInvalidValue;      // (the fp16 panel is read where the bf16 gradient is written)
    if (a.logits16 && a.V <= 8192 && (a.V & 7) == 0 && !a.loss_acc) hipLaunchKernelGGL(softmax_ce_h16_kernel, dim3(min(a.n_max, 8192)), dim3(256), 0, st, a);
    else if (a.V <= 8192) hipLaunchKernelGGL(softmax_ce_reg_kernel, dim3(min(a.n_max, 8192)), dim3(256), 0, st, a);
    else             hipLaunchKernelGGL(softmax_ce_kernel, dim3(min(a.n_max, 8192)), dim3(256), 0, st, a);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ logits, int32_t* __restrict__ pred, int n, int V)
{
    __shared__ float s_bv[4]; __shared__ int s_bi[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int row = blockIdx.x; row < n; row += gridDim.x) {
        const float* x = logits + (size_t)row * V;
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int c = tid; c < V; c += 256) { float v = x[c]; if (v > bv) { bv = v; bi = c; } }
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_down(bv, o, 64); int oi = __shfl_down(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { s_bv[wave] = bv; s_bi[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) if (s_bv[w] > bv || (s_bv[w] == bv && s_bi[w] < bi)) { bv = s_bv[w]; bi = s_bi[w]; }
            pred[row] = bi;
        }
    }
}
hipError_t argmax_rows(hipStream_t st, const float* logits, int32_t* pred, int n, int V)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(min(n, 4096)), dim3(256), 0, st, logits, pred, n, V);
    return hipGetLastError();
}

// ---------------------------------------------------------------- column sums (bias gradients)
// block = 64 columns x 4 row lanes; grid.y chunks of `chunk` rows, sized so that ~1024 workgroups share the rows (with
// 1024-row chunks the 16 640 x 512 sum ran on 136 workgroups of 256 dependent loads each: 62 us for 34 MB); float atomics
// into out (+=).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int M, int N, int ldx,
                                                     float* __restrict__ out, const int32_t* __restrict__ m_dev, int chunk)
{
    __shared__ float sh[4][64];
    if (m_dev) M = min(M, *m_dev);
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * chunk, r1 = min(M, r0 + chunk);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < N) {
        int r = r0 + rl;
        for (; r + 12 < r1; r += 16) {         // four independent loads in flight per thread
            s0 += X[(size_t)r * ldx + c]; s1 += X[(size_t)(r + 4) * ldx + c];
            s2 += X[(size_t)(r + 8) * ldx + c]; s3 += X[(size_t)(r + 12) * ldx + c];
        }
        for (; r < r1; r += 4) s0 += X[(size_t)r * ldx + c];
    }
    sh[rl][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && c < N && r0 < r1) atomicAdd(out + c, (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]));
}
hipError_t colsum(hipStream_t st, const float* X, int M, int N, int ldx, float* out, const int32_t* m_dev)
{
    if (M <= 0 || N <= 0) return hipSuccess;
    const int gx = (N + 63) / 64, want = std::max(1, 1024 / gx);
    int chunk = std::max(16, (M + want - 1) / want);
    chunk = (chunk + 3) & ~3;
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, (M + chunk - 1) / chunk), dim3(256), 0, st, X, M, N, ldx, out, m_dev, chunk);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void add3_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b,
                                                   const float* __restrict__ c, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] + (b ? b[i] : 0.f) + (c ? c[i] : 0.f);
}
hipError_t add3(hipStream_t st, float* out, const float* a, const float* b, const float* c, int64_t n)
{
    hipLaunchKernelGGL(add3_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, st, out, a, b, c, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------- Adam, TF formulation (model.py:189)
// m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_t m / (sqrt(v) + eps), lr_t bias-corrected
// on the host.  28 B of HBM traffic per parameter, one launch over the flat state.
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a)
{
    const int64_t n4 = a.n >> 2;
    float4* p = reinterpret_cast<float4*>(a.p);
    const float4* g = reinterpret_cast<const float4*>(a.g);
    float4* m = reinterpret_cast<float4*>(a.m);
    float4* v = reinterpret_cast<float4*>(a.v);
    const float c1 = 1.f - a.b1, c2 = 1.f - a.b2;
    // a persistent GRU launch of this step gave up on a wait: its gradients are garbage, the update is skipped
    // (uniform branch; the host raises at its next synchronising call)
    if (a.skip_if && *a.skip_if != 0) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 P = p[i], G = g[i], Mv = m[i], V = v[i];
#define AVAE_ADAM1(f) Mv.f = a.b1 * Mv.f + c1 * G.f; V.f = a.b2 * V.f + c2 * G.f * G.f; P.f -= a.lr_t * Mv.f / (sqrtf(V.f) + a.eps);
        AVAE_ADAM1(x) AVAE_ADAM1(y) AVAE_ADAM1(z) AVAE_ADAM1(w)
#undef AVAE_ADAM1
        p[i] = P; m[i] = Mv; v[i] = V;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n4 << 2; i < a.n; ++i) {
            float G = a.g[i];
            a.m[i] = a.b1 * a.m[i] + c1 * G; a.v[i] = a.b2 * a.v[i] + c2 * G * G;
            a.p[i] -= a.lr_t * a.m[i] / (sqrtf(a.v[i]) + a.eps);
        }
}
hipError_t adam_tf(hipStream_t st, const AdamArgs& a)
{
    hipLaunchKernelGGL(adam_kernel, dim3(2048), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------- G16 row permutation
__global__ __launch_bounds__(256) void g16_permute_kernel(float* __restrict__ dst, const float* __restrict__ src, int D, int cols, int to_g16)
{
    const int64_t total = (int64_t)3 * D * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t row = i / cols; int col = (int)(i - row * cols);
        // row is a G16 row index c' = ht*48 + u*3 + gate  <->  natural gate*D + ht*16 + u
        int ht = (int)(row / 48), rem = (int)(row % 48), u = rem / 3, gate = rem % 3;
        int64_t nat = (int64_t)gate * D + ht * 16 + u;
        if (to_g16) dst[i] = src[nat * cols + col];
        else dst[nat * cols + col] = src[i];
    }
}
hipError_t g16_permute(hipStream_t st, float* dst, const float* src, int D, int cols, bool to_g16)
{
    int64_t total = (int64_t)3 * D * cols;
    hipLaunchKernelGGL(g16_permute_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, st, dst, src, D, cols, to_g16 ? 1 : 0);
    return hipGetLastError();
}

// The three scalars of model.py:181-185 from the per-sample arrays, summed in a FIXED order (thread i takes elements i, i + 1024,
// ...; then a tree over the block): the same inputs give the same bits run after run -- no float atomics (SURVEY section 5 asks
// for a same-seed => bit-identical-loss determinism test).  loss_samp (n tokens) and kld (nk = B R elements) are written by
// softmax_ce / latent_fwd anyway; 16 640 + 32 768 floats at configs[1], one 1024-thread block, ~6 us.
__global__ __launch_bounds__(1024) void finalize_losses_kernel(float* losses, const float* __restrict__ loss_samp, const int32_t* n_dev, int n_max,
                                                               const float* __restrict__ kld, int nk, float free_bits, float inv_br, float anneal)
{
    __shared__ float sh[2][1024];
    const int tid = threadIdx.x, n = min(n_max, *n_dev);
    float sg = 0.f, sk = 0.f;
    for (int i = tid; i < n; i += 1024) sg += loss_samp[i];
    for (int i = tid; i < nk; i += 1024) sk += fmaxf(kld[i], free_bits);
    sh[0][tid] = sg; sh[1][tid] = sk;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        const float gen = sh[0][0] / (float)max(n, 1), kl = sh[1][0] * inv_br;
        losses[0] = gen; losses[1] = kl; losses[2] = anneal * kl + gen;
    }
}
hipError_t finalize_losses(hipStream_t st, float* losses, const float* loss_samp, const int32_t* n_dev, int n_max,
                           const float* kld, int nk, float free_bits, float inv_br, float anneal)
{
    hipLaunchKernelGGL(finalize_losses_kernel, dim3(1), dim3(1024), 0, st, losses, loss_samp, n_dev, n_max, kld, nk, free_bits, inv_br, anneal);
    return hipGetLastError();
}

}  // namespace avae
