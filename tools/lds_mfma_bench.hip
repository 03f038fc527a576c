// Micro-benchmark (r05): what bounds the implicit-GEMM K loop on gfx950 -- LDS fragment reads, LDS-DMA writes, or the matrix pipe?
// A workgroup keeps ONE K tile (64 deep) of X [TM rows] and W [TN rows] in LDS (144-byte rows: conflict-free ds_read_b128) and runs the
// K-loop body of an implicit GEMM on it over and over: per 16-deep K step MI + NJ fragment reads, MI x NJ v_mfma_f32_32x32x16_f16;
// optionally it also issues the LDS-DMA (buffer_load ... lds, 16 bytes per lane) that would refill a ring of K tiles from an L2-resident
// source (no waits: timing only).  Wave layouts:
//     8 waves (2 per SIMD): 4 x 2 tiles per wave (igemm8: 256 x 256)          8 waves: 2 x 5 (igemm320: 256 x 320)
//     4 waves (1 per SIMD, up to 512 VGPRs): 4 x 4 (256 x 256) and 4 x 5 (256 x 320)
// Random fp16 operands (the chip is power limited: profiles/r05_issue_overlap.log).  Build + run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 tools/lds_mfma_bench.hip -o /tmp/lm.bin && /tmp/lm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RS = 144;                                   // LDS row stride, bytes (64 halves + 8 pad)

template <int NW, int WM, int WN, int MI, int NJ, int DMA>
__global__ void __launch_bounds__(NW * 64, 1) k(const f16* __restrict__ src, float* out, int iters) {
    constexpr int TM = WM * MI * 32, TN = WN * NJ * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sx = smem;
    char* sw = smem + TM * RS;
    char* dma_dst = smem + (TM + TN) * RS;                // the "other ring slot": DMA target, never read
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // fill the tile with pseudo-random fp16 in (-0.5, 0.5)
    unsigned h = tid * 2654435761u + 777u;
    for (int i = tid; i < (TM + TN) * RS / 2; i += NW * 64) {
        h = h * 1664525u + 1013904223u;
        ((f16*)smem)[i] = (f16)(((h >> 8) & 0xffff) / 65536.0f - 0.5f);
    }
    __syncthreads();
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* xa = sx + (wm * MI * 32 + l31) * RS + lh * 16;
    const char* wa = sw + (wn * NJ * 32 + l31) * RS + lh * 16;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1u << 24, 0x00020000);
    constexpr int PIECES = (TM + TN) * 128 / 1024 / NW;   // 1 KB DMA pieces per wave and K tile (the real 128-byte rows)
    for (int it = 0; it < iters; ++it) {
        if (DMA) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dma_dst + (wave * PIECES + p) * 1024), 16,
                                                         (unsigned)(lane * 16), ((it & 63) * NW * PIECES + wave * PIECES + p) * 1024, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 xf[MI], wf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) xf[i] = *(const f16x8*)(xa + i * 32 * RS + ks * 32);
#pragma unroll
            for (int j = 0; j < NJ; ++j) wf[j] = *(const f16x8*)(wa + j * 32 * RS + ks * 32);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        if (DMA == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) r += acc[i][j][q];
    if (r == 123.456f) out[0] = r;
}

template <int NW, int WM, int WN, int MI, int NJ, int DMA>
static void run(const char* name, const f16* src, float* out, int iters) {
    constexpr int TM = WM * MI * 32, TN = WN * NJ * 32;
    const size_t lds = (size_t)(TM + TN) * RS + (DMA ? (size_t)(TM + TN) * 128 : 0);
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    if (lds > 160 * 1024) { printf("%-58s LDS %zu KB: skipped\n", name, lds / 1024); return; }
    (void)hipFuncSetAttribute((const void*)k<NW, WM, WN, MI, NJ, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NW, WM, WN, MI, NJ, DMA>), dim3(pr.multiProcessorCount), dim3(NW * 64), lds, 0, src, out, iters / 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NW, WM, WN, MI, NJ, DMA>), dim3(pr.multiProcessorCount), dim3(NW * 64), lds, 0, src, out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    const double flop = 2.0 * TM * TN * 64 * (double)iters * pr.multiProcessorCount;
    printf("%-58s tile %3d x %3d  %d waves  reads/MFMA %.2f  LDS %3zu KB : %7.1f TF/s%s\n", name, TM, TN, NW, (double)(MI + NJ) / (MI * NJ), lds / 1024,
           flop / (ms * 1e-3) / 1e12, err == hipSuccess ? "" : "  (LAUNCH ERROR)");
}


// ---- part 2: K loop + epilogue per output tile: ONE 8-wave workgroup per CU (K loop and epilogue serialise on the CU) against TWO
// independent 4-wave workgroups per CU with half-size tiles and 32-deep K tiles (their LDS must fit twice: one's epilogue runs under the
// other's K loop) ------------------------------------------------------------------------------------------------------------------------
template <int NW, int WM, int WN, int MI, int NJ, int BK>
__global__ void __launch_bounds__(NW * 64, (NW == 4 ? 2 : 1)) kt(const f16* __restrict__ src, f16* __restrict__ dst, int ntiles, int K) {
    constexpr int TM = WM * MI * 32, TN = WN * NJ * 32, RSK = BK * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sx = smem;
    char* sw = smem + TM * RSK;
    char* dma_dst = smem + (TM + TN) * RSK;
    char* scr = (NW == 8 ? dma_dst : dma_dst + (TM + TN) * BK * 2) + (threadIdx.x >> 6) * 4096;   // (8 waves: the scratch aliases the DMA target -- timing only -- to stay inside 160 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    unsigned h = (tid + 977 * blockIdx.x) * 2654435761u + 777u;
    for (int i = tid; i < (TM + TN) * RSK / 2; i += NW * 64) {
        h = h * 1664525u + 1013904223u;
        ((f16*)smem)[i] = (f16)(((h >> 8) & 0xffff) / 65536.0f - 0.5f);
    }
    __syncthreads();
    const char* xa = sx + (wm * MI * 32 + l31) * RSK + lh * 16;
    const char* wa = sw + (wn * NJ * 32 + l31) * RSK + lh * 16;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1u << 24, 0x00020000);
    constexpr int PIECES = (TM + TN) * BK * 2 / 1024 / NW;
    const int nk = K / BK;
    for (int tile = 0; tile < ntiles; ++tile) {
        f32x16 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < nk; ++it) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dma_dst + (wave * PIECES + p) * 1024), 16,
                                                         (unsigned)(lane * 16), (((tile * nk + it) & 63) * NW * PIECES + wave * PIECES + p) * 1024, 0, 0);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                f16x8 xf[MI], wf[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) xf[i] = *(const f16x8*)(xa + i * 32 * RSK + ks * 32);
#pragma unroll
                for (int j = 0; j < NJ; ++j) wf[j] = *(const f16x8*)(wa + j * 32 * RSK + ks * 32);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // epilogue as the light kinds do it: fp16, 32 x 64 transposes through a private 4 KB scratch, 16-byte row-major stores
        f16* out = dst + ((size_t)((blockIdx.x * ntiles + tile) & 1023) * TM * TN);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jp = 0; jp < NJ; jp += 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                        f16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (f16)acc[i][jp + j < NJ ? jp + j : jp][4 * g + e];
                        const int c8 = 8 * j + 2 * g + lh;
                        *(f16x4*)(scr + l31 * 128 + (((c8 >> 1) ^ ((l31 >> 1) & 7)) << 4) + (((c8 & 1) ^ (l31 & 1)) << 3)) = o;
                    }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = 8 * p + (lane >> 3), blk = lane & 7;
                    const f16x8 v = *(const f16x8*)(scr + row * 128 + ((blk ^ ((row >> 1) & 7)) << 4));
                    if (jp + 2 <= NJ || blk < 4)
                        *(f16x8*)(out + (size_t)((wm * MI + i) * 32 + row) * TN + (wn * NJ + jp) * 32 + 8 * blk) = v;
                }
            }
    }
}

template <int NW, int WM, int WN, int MI, int NJ, int BK>
static void run_tiles(const char* name, const f16* src, f16* dst, int K, int ntiles) {
    constexpr int TM = WM * MI * 32, TN = WN * NJ * 32, RSK = BK * 2 + 16;
    const size_t lds = (size_t)(TM + TN) * RSK + (size_t)(TM + TN) * BK * 2 + (NW == 8 ? 0 : NW * 4096);
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    (void)hipFuncSetAttribute((const void*)kt<NW, WM, WN, MI, NJ, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int per_cu = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kt<NW, WM, WN, MI, NJ, BK>, NW * 64, lds);
    const int wgs = pr.multiProcessorCount * (NW == 4 ? 2 : 1);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kt<NW, WM, WN, MI, NJ, BK>), dim3(wgs), dim3(NW * 64), lds, 0, src, dst, ntiles / 8 + 1, K);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kt<NW, WM, WN, MI, NJ, BK>), dim3(wgs), dim3(NW * 64), lds, 0, src, dst, ntiles, K);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    const double flop = 2.0 * TM * TN * (double)K * ntiles * wgs;
    printf("%-52s tile %3d x %3d, BK %2d, %d waves, %d WG/CU resident, LDS %3zu KB, K = %4d: %7.1f TF/s%s\n", name, TM, TN, BK, NW, per_cu, lds / 1024, K,
           flop / (ms * 1e-3) / 1e12, err == hipSuccess ? "" : "  (LAUNCH ERROR)");
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    f16* src;
    float* out;
    (void)hipMalloc(&src, 1u << 25);
    (void)hipMemset(src, 0x3c, 1u << 25);
    (void)hipMalloc(&out, 64);
    printf("K-loop body on one LDS-resident K tile, random fp16; DMA 0 = none, 1 = ring refill issued (never waited for), 2 = + vmcnt(0) per K tile\n");
    run<8, 2, 4, 4, 2, 0>("8 waves, 4 x 2 per wave (igemm8)", src, out, iters);
    run<8, 2, 4, 4, 2, 1>("8 waves, 4 x 2 per wave (igemm8) + DMA", src, out, iters);
    run<8, 2, 4, 4, 2, 2>("8 waves, 4 x 2 per wave (igemm8) + DMA + wait", src, out, iters);
    run<8, 4, 2, 2, 5, 0>("8 waves, 2 x 5 per wave (igemm320)", src, out, iters);
    run<8, 4, 2, 2, 5, 1>("8 waves, 2 x 5 per wave (igemm320) + DMA", src, out, iters);
    run<4, 2, 2, 4, 4, 0>("4 waves, 4 x 4 per wave", src, out, iters);
    run<4, 2, 2, 4, 4, 1>("4 waves, 4 x 4 per wave + DMA", src, out, iters);
    run<4, 2, 2, 4, 4, 2>("4 waves, 4 x 4 per wave + DMA + wait", src, out, iters);
    run<4, 2, 2, 4, 5, 0>("4 waves, 4 x 5 per wave", src, out, iters);
    run<4, 2, 2, 4, 5, 1>("4 waves, 4 x 5 per wave + DMA", src, out, iters);
    run<4, 1, 4, 8, 2, 0>("4 waves, 8 x 2 per wave", src, out, iters);
    run<8, 2, 4, 2, 2, 0>("8 waves, 2 x 2 per wave (128 x 256 tile)", src, out, iters);
    printf("K loop + epilogue per output tile (fp16 transposes through LDS, 16-byte stores to a 512 MB window), DMA issued, random operands\n");
    f16* dst;
    (void)hipMalloc(&dst, (size_t)1025 * 256 * 320 * 2);
    for (int K : {320, 640, 1280, 2880}) {
        const int nt = 128000 / K + 8;
        run_tiles<8, 2, 4, 4, 2, 64>("one 8-wave WG per CU (igemm8 shape)", src, dst, K, nt);
        run_tiles<8, 4, 2, 2, 5, 64>("one 8-wave WG per CU (igemm320 shape)", src, dst, K, nt);
        run_tiles<4, 2, 2, 2, 4, 32>("two 4-wave WGs per CU, 128 x 256, 2 x 4 per wave", src, dst, K, 2 * nt);
        run_tiles<4, 1, 4, 4, 2, 32>("two 4-wave WGs per CU, 128 x 256, 4 x 2 per wave", src, dst, K, 2 * nt);
        run_tiles<4, 2, 2, 2, 5, 32>("two 4-wave WGs per CU, 128 x 320, 2 x 5 per wave", src, dst, K, 2 * nt);
    }
    return 0;
}
