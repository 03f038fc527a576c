// dmabench: what does one LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KB per wave) cost a wave of the implicit-GEMM K loop,
// as a function of (a) how many pieces a load segment carries and (b) the SHAPE of a piece in memory --
//     PAT 0: 8 rows x 128 B (whole cache lines; the 256x256 kernel's pieces)
//     PAT 1: 16 rows x 64 B (half lines: what a K-half / BK = 32 refill of the 256x320 tile would issue)
// The skeleton is the K loop's: 8 waves, two wave groups one barrier apart, a phase = {P pieces, counted vmcnt, s_barrier,
// NM MFMAs, s_barrier}; no fragment reads (their cost is known: 4 LDS cycles per ds_read_b128).  One workgroup per CU
// (148 KB of LDS), source matrix L2 / MALL resident.  Standalone:
//   hipcc --offload-arch=gfx950 -O3 tools/dmabench.hip -o tools/dmabench.bin && tools/dmabench.bin
// Prints wall time per phase in ns and shader cycles per phase (s_memtime) for P = 0..6: the slope is the cost of a piece.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int PAT, int P, int NM>
__global__ __launch_bounds__(512, 2) void dma_kernel(const char* __restrict__ A, unsigned a_bytes, int ld, int nrows, int iters,
                                                    unsigned long long* __restrict__ cyc, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    // per-lane source offset of piece q of this wave in a 256-row tile (swizzled chunk as the real kernels do)
    unsigned off[6];
    const int row0 = (blockIdx.x * 256) % nrows;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        if (PAT == 0) {
            const int row = (wave * 4 + (q & 3)) * 8 + (lane >> 3);
            off[q] = (unsigned)(row0 + row) * (unsigned)ld + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16) + (q >> 2) * 128;
        } else {
            const int row = (wave * 2 + (q & 1)) * 16 + (lane >> 2);
            off[q] = (unsigned)(row0 + row) * (unsigned)ld + (unsigned)((((lane & 3) ^ ((row >> 2) & 3)) * 16) + ((q >> 1) & 1) * 64) +
                     (q >> 2) * 128;
        }
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 fa, fb;
#pragma unroll
    for (int e = 0; e < 8; ++e) { fa[e] = (f16)(0.001f * (lane + e)); fb[e] = (f16)(0.002f * (lane - e)); }
    if (grp == 1) __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int soff = (it & 15) * 256;                       // walk along K (two K tiles of 128 B per step pair)
#pragma unroll
        for (int q = 0; q < P; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + ((it & 1) * 8 + wave) * 6144 + q * 1024),
                                                     16, off[q], soff, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(P < 1 ? 0 : 2 * P) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[m & 3], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[tid] = s + smem[tid];
}

template <int PAT, int P, int NM>
static void run(const char* A, unsigned a_bytes, int ld, int nrows, unsigned long long* cyc, float* sink) {
    auto kern = dma_kernel<PAT, P, NM>;
    const int lds = 16 * 6144 + 53248;                          // 148 KB: one workgroup per CU
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, A, a_bytes, ld, nrows, iters, cyc, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, A, a_bytes, ld, nrows, iters, cyc, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double c = 0;
    for (int i = 0; i < 256; ++i) c += (double)h[i];
    c /= 256.0 * iters;
    // one loop iteration = one phase of EACH wave group = two barrier intervals on every SIMD
    printf("pat %d  pieces/phase %d  mfma/phase %2d : %8.1f ns / iteration   %7.1f cycles / iteration  (MFMA-bound floor %d)  %.2f GHz\n",
           PAT, P, NM, ms / 3.0 / iters * 1e6, c, 2 * 32 * NM, c / (ms / 3.0 / iters * 1e6));
}

int main() {
    const int nrows = 8192, ld = 5120;                          // 40 MB: MALL resident, per-XCD L2 partly
    const unsigned a_bytes = (unsigned)nrows * ld;
    char* A; unsigned long long* cyc; float* sink;
    CK(hipMalloc(&A, a_bytes)); CK(hipMemset(A, 0x11, a_bytes));
    CK(hipMalloc(&cyc, 256 * 8)); CK(hipMalloc(&sink, 4096));
    printf("8 waves / CU, two staggered wave groups; a phase = {P LDS-DMA pieces, vmcnt(2P), barrier, NM MFMA 32x32x16, barrier}\n");
    run<0, 0, 16>(A, a_bytes, ld, nrows, cyc, sink);
    run<0, 2, 16>(A, a_bytes, ld, nrows, cyc, sink);
    run<0, 4, 16>(A, a_bytes, ld, nrows, cyc, sink);
    run<0, 6, 16>(A, a_bytes, ld, nrows, cyc, sink);
    run<1, 2, 16>(A, a_bytes, ld, nrows, cyc, sink);
    run<1, 4, 16>(A, a_bytes, ld, nrows, cyc, sink);
    run<1, 6, 16>(A, a_bytes, ld, nrows, cyc, sink);
    run<0, 0, 20>(A, a_bytes, ld, nrows, cyc, sink);
    run<0, 4, 20>(A, a_bytes, ld, nrows, cyc, sink);
    run<0, 5, 20>(A, a_bytes, ld, nrows, cyc, sink);
    run<1, 4, 20>(A, a_bytes, ld, nrows, cyc, sink);
    run<1, 5, 20>(A, a_bytes, ld, nrows, cyc, sink);
    return 0;
}
