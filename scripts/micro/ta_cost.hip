// TA / L1 cost of one vector-memory instruction per access pattern: 256 workgroups x 16 waves issue the same
// instruction ITERS times; cycles per instruction per CU = time * clock / (ITERS * 16).  (diagnostic, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// pattern -> byte offset of lane `l` within the wave's region for iteration it
__device__ __forceinline__ unsigned pat_off(int pat, int l)
{
    switch (pat) {
    case 0: return l * 4;                                  // dword, 256 B contiguous
    case 1: return (l >> 2) * 256 + (l & 3) * 4;            // dword, quads of 16 B, 256 B apart
    case 2: return (l & 15) * 12 + (l >> 4) * 6144;          // dword, stride 12 B, 4 rows 6 KB apart
    case 3: return (l & 15) * 12 + (l >> 4) * 6144;          // x3, 16 lanes = 192 B contiguous, 4 rows
    case 4: return (l >> 2) * 6144 + (l & 3) * 12;           // x3, 4 lanes = 48 B, 16 rows
    case 5: return (l & 15) * 16 + (l >> 4) * 8192;          // x4, 256 B per row, 4 rows
    case 6: return (l >> 2) * 8192 + (l & 3) * 16;           // x4, 64 B per row, 16 rows
    case 7: return (l & 31) * 768 + 188;                     // dword, 32 distinct lines (probe)
    case 8: return l * 16;                                   // x4, 1 KB contiguous
    case 9: return (l & 15) * 6144 + (l >> 4) * 16;          // x4, 16 rows x 64 B (the old operand pattern)
    case 10: return (l & 15) * 4 + (l >> 4) * 2048;          // dword, 64 B per row, 4 rows
    }
    return 0;
}
template <int KIND>   // 0 global load, 1 global store, 2 buffer load, 3 buffer store, 4 global store sc1, 5 global load sc1
__global__ __launch_bounds__(1024) void k(float* base, int pat, int width, int iters, float* sink)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    char* region = reinterpret_cast<char*>(base) + ((size_t)blockIdx.x * 16 + w) * (1u << 18);   // 256 KB per wave
    const unsigned lo = pat_off(pat, l);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(region, 0, -1, 0x00020000);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const unsigned o = lo + (unsigned)(it & 1) * 131072u;
        char* p = region + o;
        if (KIND == 0 || KIND == 5) {
            if (width == 1) { unsigned v = KIND == 5 ? __hip_atomic_load((unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *(volatile unsigned*)p; acc += v; }
            else if (width == 3) { f32x3 v = *(volatile f32x3*)p; acc += v.x + v.z; }
            else { u32x4 v = __builtin_nontemporal_load((u32x4*)p); acc += v.x + v.w; }
        } else if (KIND == 1 || KIND == 4) {
            if (width == 1) { if (KIND == 4) __hip_atomic_store((float*)p, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *(volatile float*)p = acc; }
            else if (width == 3) { f32x3 v = {acc, acc, acc}; *(volatile f32x3*)p = v; }
            else { u32x4 v = {1u, 2u, 3u, 4u}; *(volatile u32x4*)p = v; }
        } else if (KIND == 2) {
            if (width == 1) acc += __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0);
            else if (width == 4) { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 16); acc += v.x + v.w; }
        } else {
            if (width == 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc), rs, o, 0, 0);
            else if (width == 4) { u32x4 v = {1u, 2u, 3u, 4u}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, o, 0, 0); }
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}
// does a clock read right after vector stores report the time of the read, or does it wait behind them?
__global__ __launch_bounds__(1024) void clock_after_store(float* base, int nstores, unsigned long long* out)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* region = base + ((size_t)blockIdx.x * 16 + w) * 65536;
    unsigned long long d01 = 0, d12 = 0, d23 = 0;
    for (int it = 0; it < 200; ++it) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < nstores; ++i) __hip_atomic_store(region + ((it & 7) * 8 + i) * 256 + (l & 15) * 3 + (l >> 4) * 64, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_branch 1f\n1:" ::: "memory");
        const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
        d01 += t1 - t0; d12 += t2 - t1; d23 += t3 - t2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = d01; out[1] = d12; out[2] = d23; }
}
int main()
{
    {
        float* b2; CK(hipMalloc(&b2, (size_t)256 * 16 * 65536 * 4)); unsigned long long* o; CK(hipMalloc(&o, 64)); unsigned long long h[3];
        for (int ns : {0, 2, 6}) {
            hipLaunchKernelGGL(clock_after_store, dim3(256), dim3(1024), 0, 0, b2, ns, o); CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, o, 24, hipMemcpyDeviceToHost));
            printf("clock reads around %d stores (16 waves/CU): stores %.3f us, back-to-back read %.3f us, read after a branch %.3f us\n", ns, h[0] / 200.0 * 0.01, h[1] / 200.0 * 0.01, h[2] / 200.0 * 0.01);
        }
        CK(hipFree(b2)); CK(hipFree(o));
    }
    float* buf; CK(hipMalloc(&buf, (size_t)256 * 16 * (1u << 18) + 4096)); CK(hipMemset(buf, 0, (size_t)256 * 16 * (1u << 18)));
    float* sink; CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int kind, pat, width; };
    std::vector<Case> cs = {
        {"gload  dword 256B contiguous", 0, 0, 1}, {"gload  dword 4 rows x 64 B", 0, 10, 1}, {"gload  dword 32 lines (probe)", 0, 7, 1}, {"gload.sc1 dword 32 lines (probe)", 5, 7, 1},
        {"gload  x3 4 rows x 192 B", 0, 3, 3}, {"gload  x4 4 rows x 256 B", 0, 5, 4}, {"gload  x4 16 rows x 64 B", 0, 6, 4}, {"gload  x4 1 KB contiguous", 0, 8, 4},
        {"bload  x4 1 KB contiguous", 2, 8, 4}, {"bload  x4 16 rows x 64 B (old operand)", 2, 9, 4}, {"bload  dword 32 lines (probe)", 2, 7, 1},
        {"gstore dword 256B contiguous", 1, 0, 1}, {"gstore dword quads 16 B", 1, 1, 1}, {"gstore dword stride 12 B", 1, 2, 1}, {"gstore.sc1 dword stride 12 B", 4, 2, 1}, {"gstore.sc1 dword 256B contiguous", 4, 0, 1},
        {"gstore dword 4 rows x 64 B", 1, 10, 1}, {"gstore x3 4 rows x 192 B", 1, 3, 3}, {"gstore x3 16 rows x 48 B", 1, 4, 3}, {"gstore x4 4 rows x 256 B", 1, 5, 4}, {"gstore x4 16 rows x 64 B", 1, 6, 4},
        {"bstore dword stride 12 B", 3, 2, 1}, {"bstore dword 256B contiguous", 3, 0, 1}, {"bstore x4 4 rows x 256 B", 3, 5, 4},
    };
    const int iters = 4000;
    for (auto& c : cs) {
        auto launch = [&](int it) {
            switch (c.kind) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, buf, c.pat, c.width, it, sink); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, buf, c.pat, c.width, it, sink); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, buf, c.pat, c.width, it, sink); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, buf, c.pat, c.width, it, sink); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(1024), 0, 0, buf, c.pat, c.width, it, sink); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(1024), 0, 0, buf, c.pat, c.width, it, sink); break;
            }
        };
        launch(200); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-42s %7.1f cycles / instruction / CU (at 2.4 GHz)   %.3f ms\n", c.name, ms * 1e-3 * 2.4e9 / (iters * 16.0), ms);
        fflush(stdout);
    }
    return 0;
}
