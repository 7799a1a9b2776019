// what does the C-store pattern of a 256x256-tile epilogue cost on its own?  One workgroup (8 waves) per CU walks the output tiles of a
// (132096 x 8192) fp32 matrix in gemm_bf16_p8's item order and only STORES them (no loads, no MFMAs), in several lane -> address maps:
//   A  as gemm_bf16_p8's epilogue: one instruction = 16 rows x 64 B (half lines), 32 instructions per wave and tile
//   B  whole lines: one instruction = 8 rows x 128 B (neighbour-lane exchange in front of it), 32 instructions
//   C  pattern A, non-temporal
//   D  bf16 output in pattern A's shape: 16 rows x 64 B, 16 instructions (half the bytes)
//   E  upper bound: one instruction = 1 row x 1 KB
//   F  pattern B with pauses (s_sleep) between groups of 4 stores: a drain spread over time instead of a burst
// build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern;  run: ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(float* C, int tiles_m, int tiles_n, int ldc)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3, fr = lane & 15, fq = lane >> 4;
    const int tiles = tiles_m * tiles_n, G = gridDim.x;
    int place;
    { const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3; place = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot; }
    const f32x4 val = {1.f + lane, 2.f, 3.f, 4.f + wave};
    for (int v = place; v < tiles; v += G) {
        const int per = 8 * tiles_n, grp = v / per, in = v - grp * per, rows = min(8, tiles_m - grp * 8), tn = in / rows;
        const int m0 = (grp * 8 + in - tn * rows) * 256, n0 = tn * 256;
        if (PAT == 0 || PAT == 2) {
            float* c0 = C + (size_t)(m0 + wr * 64 + fr) * ldc + n0 + wc * 32 + fq * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x4* p = reinterpret_cast<f32x4*>(c0 + (size_t)((i >> 2) * 128 + (i & 3) * 16) * ldc + (j >> 1) * 128 + (j & 1) * 16);
                    if (PAT == 2) __builtin_nontemporal_store(val, p); else *p = val;
                }
        } else if (PAT == 1 || PAT == 5) {
            // row (fr & ~1) | s, chunk (fr & 1) * 4 + fq of the wave's 32 columns
            float* c0 = C + (size_t)(m0 + wr * 64 + (fr & ~1)) * ldc + n0 + wc * 32 + ((fr & 1) * 4 + fq) * 4;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        *reinterpret_cast<f32x4*>(c0 + (size_t)((i >> 2) * 128 + (i & 3) * 16 + s) * ldc + jp * 128) = val;
                    if (PAT == 5 && (i & 1)) __builtin_amdgcn_s_sleep(8);
                }
        } else if (PAT == 3) {
            unsigned short* c0 = reinterpret_cast<unsigned short*>(C) + (size_t)(m0 + wr * 64 + fr) * ldc + n0 + wc * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *reinterpret_cast<f32x4*>(c0 + (size_t)((i >> 2) * 128 + (i & 3) * 16) * ldc + jp * 128) = val;
        } else {
            // wave w stores rows w*32 .. w*32+31 of the tile, one 1-KB row per instruction
#pragma unroll
            for (int i = 0; i < 32; ++i)
                *reinterpret_cast<f32x4*>(C + (size_t)(m0 + wave * 32 + i) * ldc + n0 + lane * 4) = val;
        }
    }
}

int main()
{
    const int M = 132096, N = 8192, tiles_m = M / 256, tiles_n = N / 256;
    float* C; hipMalloc(&C, (size_t)M * N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[6] = {"A p8 epilogue (16 rows x 64 B)", "B whole lines (8 rows x 128 B)", "C pattern A non-temporal", "D bf16, 16 rows x 64 B, half the bytes", "E 1 row x 1 KB", "F whole lines, paced"};
    for (int rep = 0; rep < 2; ++rep)
        for (int pat = 0; pat < 6; ++pat) {
            hipEventRecord(e0);
            for (int it = 0; it < 3; ++it) {
                switch (pat) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, C, tiles_m, tiles_n, N); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, C, tiles_m, tiles_n, N); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, C, tiles_m, tiles_n, N); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, C, tiles_m, tiles_n, N); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, C, tiles_m, tiles_n, N); break;
                default: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, C, tiles_m, tiles_n, N); break;
                }
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
            const double bytes = (double)M * N * (pat == 3 ? 2 : 4);
            printf("%-42s %8.1f us  %6.2f TB/s  (%.1f us per 256x256 tile and CU)\n", names[pat], ms * 1e3, bytes / ms / 1e9, ms * 1e3 / ((double)tiles_m * tiles_n / 256));
        }
    return 0;
}
