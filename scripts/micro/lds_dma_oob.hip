// does an LDS-DMA buffer load (buffer_load_dwordx4 ... lds) write ZEROS for lanes beyond num_records?  build: hipcc --offload-arch=gfx950 -O2 lds_dma_oob.hip -o /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* g, unsigned nbytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char L[];
    for (int i = threadIdx.x; i < 1024; i += 64) ((unsigned*)L)[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)L, 16, threadIdx.x * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((unsigned*)L)[i];
}
int main() {
    unsigned *g, *o; hipMalloc(&g, 4096); hipMalloc(&o, 1024);
    unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1;
    hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, 600u, o);
    unsigned r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    for (int i = 140; i < 160; ++i) printf("%d:%x ", i, r[i]); printf("\n");
    printf("last in-bounds dword index 149 (600 bytes): r[149]=%x r[150]=%x r[151]=%x r[152]=%x r[255]=%x\n", r[149], r[150], r[151], r[152], r[255]);
    return 0;
}
