// Probe (round 6): the instruction's immediate offset of an LDS-DMA advances the global AND the LDS address; and does LDS-DMA (buffer_load ... lds) reach every 1 KB block of a 160 KB dynamic LDS allocation?  One workgroup DMAs block b of
// a global pattern to LDS block b for b = 0 .. 159 and reads the LDS back with ds_read.   hipcc --offload-arch=gfx950 -O2 -o /tmp/p tools/lds_dma_range.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 1) void probe(const unsigned* src, unsigned* out, int nblk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < nblk * 256; i += 256) ((unsigned*)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)nblk * 1024u, 0x00020000);
    // groups of 4 blocks: ONE LDS base / scalar offset, the block inside the group selected by the instruction's immediate offset --
    // it must advance BOTH the global and the LDS address
    for (int g = wave; g < nblk / 4; g += 4) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + g * 4096), 16, lane * 16, g * 4096, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + g * 4096), 16, lane * 16, g * 4096, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + g * 4096), 16, lane * 16, g * 4096, 2048, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + g * 4096), 16, lane * 16, g * 4096, 3072, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < nblk * 256; i += 256) out[i] = ((unsigned*)smem)[i];
}
int main() {
    for (int nblk : {120, 128, 144, 156, 160}) {
        std::vector<unsigned> h(nblk * 256);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i * 2654435761u;
        unsigned *d, *o;
        hipMalloc(&d, h.size() * 4); hipMalloc(&o, h.size() * 4);
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipError_t e0 = hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, nblk * 1024);
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), nblk * 1024, 0, d, o, nblk);
        hipError_t e1 = hipDeviceSynchronize();
        std::vector<unsigned> r(h.size());
        hipMemcpy(r.data(), o, h.size() * 4, hipMemcpyDeviceToHost);
        int badblk = 0, first = -1;
        for (int b = 0; b < nblk; ++b) {
            bool bad = false;
            for (int i = 0; i < 256; ++i) bad |= r[b * 256 + i] != h[b * 256 + i];
            if (bad) { ++badblk; if (first < 0) first = b; }
        }
        printf("LDS %3d KB: attr %d sync %d bad blocks %d (first %d)\n", nblk, (int)e0, (int)e1, badblk, first);
        hipFree(d); hipFree(o);
    }
    return 0;
}
