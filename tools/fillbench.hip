// fillbench: what does the global -> (LDS | VGPR) fill path of one CU sustain on gfx950, for the access pattern of the
// implicit-GEMM K loop (tiles of ROWS rows x 128 B, row stride LD bytes, walking along K)?  Standalone (no torch):
//   hipcc --offload-arch=gfx950 -O3 tools/fillbench.hip -o /tmp/fillbench && /tmp/fillbench
// Prints bytes/clk/CU (at the measured wall time and an assumed 2.4 GHz) for each variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16;
typedef int v4i __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One block = NW waves; one "stage" = ROWS rows x 128 B of matrix A (row stride ld bytes) at K offset ks*128.
// Tile t covers rows [t*ROWS, (t+1)*ROWS) (mod nrows); nk stages per tile.  MODE 0: LDS-DMA ring of NST stages.
// MODE 1: global_load_dwordx4 into registers, D = NST-1 stages in flight, XOR-reduced.  MODE 2: MODE 1 + ds_write_b128.
template <int MODE, int NW, int ROWS, int NST>
__global__ __launch_bounds__(64 * NW, 2) void fill_kernel(const char* __restrict__ A, long ld, int nrows, int nk, int tiles_per_block,
                                                         int* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IPW = ROWS / 8 / NW;         // 1 KB instructions per wave per stage (8 rows x 128 B each)
    constexpr int STB = ROWS * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    v4i accx = {0, 0, 0, 0};
    for (int tt = 0; tt < tiles_per_block; ++tt) {
        const long tile = (long)blockIdx.x * tiles_per_block + tt;
        const long row0 = (tile * ROWS) % nrows;
        const char* src[IPW];
#pragma unroll
        for (int q = 0; q < IPW; ++q) {
            const int row = (wave * IPW + q) * 8 + lane / 8;
            src[q] = A + (row0 + row) * ld + ((lane % 8) ^ (row & 7)) * 16;
        }
        if (MODE == 0) {
            constexpr int D = NST - 1;
            for (int t = 0; t < D && t < nk; ++t)
#pragma unroll
                for (int q = 0; q < IPW; ++q) glds16(src[q] + (long)t * 128, smem + t * STB + (wave * IPW + q) * 1024);
            for (int ks = 0; ks < nk; ++ks) {
                const int rem = nk - 1 - ks;
                if (D >= 3 && rem >= 2) wait_vmcnt<2 * IPW>();
                else if (D >= 2 && rem >= 1) wait_vmcnt<IPW>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (ks + D < nk) {
#pragma unroll
                    for (int q = 0; q < IPW; ++q)
                        glds16(src[q] + (long)(ks + D) * 128, smem + ((ks + D) % NST) * STB + (wave * IPW + q) * 1024);
                }
                // touch one LDS word per stage so the ring is observable
                accx[0] ^= *(const int*)(smem + (ks % NST) * STB + tid * 4);
            }
            __syncthreads();
        } else {
            // register path: D stages in flight held in a register ring (fully unrolled in groups of NST-1)
            constexpr int D = NST - 1;
            v4i r[D][IPW];
#pragma unroll
            for (int t = 0; t < D; ++t)
#pragma unroll
                for (int q = 0; q < IPW; ++q) r[t][q] = *(const v4i*)(src[q] + (long)(t < nk ? t : 0) * 128);
            for (int ks = 0; ks < nk; ks += D) {
#pragma unroll
                for (int t = 0; t < D; ++t) {
                    v4i cur[IPW];
#pragma unroll
                    for (int q = 0; q < IPW; ++q) cur[q] = r[t][q];
                    const int nxt = ks + t + D;
#pragma unroll
                    for (int q = 0; q < IPW; ++q) r[t][q] = *(const v4i*)(src[q] + (long)(nxt < nk ? nxt : 0) * 128);
                    if (MODE == 2) {
                        char* sb = smem + ((ks + t) & 1) * STB;
#pragma unroll
                        for (int q = 0; q < IPW; ++q) *(v4i*)(sb + (wave * IPW + q) * 1024 + lane * 16) = cur[q];
                        __builtin_amdgcn_s_barrier();
                        accx[0] ^= *(const int*)(sb + tid * 4);
                    } else {
#pragma unroll
                        for (int q = 0; q < IPW; ++q) accx ^= cur[q];
                    }
                }
            }
        }
    }
    if ((accx[0] ^ accx[1] ^ accx[2] ^ accx[3]) == 0x12345678) sink[0] = 1;
}

template <int MODE, int NW, int ROWS, int NST>
static void run(const char* name, const char* A, long ld, int nrows, int nk, int* sink, int nblocks, int tpb) {
    auto kern = fill_kernel<MODE, NW, ROWS, NST>;
    const int lds = (MODE == 0) ? NST * ROWS * 128 : (MODE == 2 ? 2 * ROWS * 128 : 0);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds > 0 ? lds : 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64 * NW), lds, 0, A, ld, nrows, nk, tpb, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64 * NW), lds, 0, A, ld, nrows, nk, tpb, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)nblocks * tpb * nk * ROWS * 128.0;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-44s lds %6d B  %8.3f ms  %6.2f TB/s  %5.1f B/clk/CU\n", name, lds, ms, tbs, tbs * 1e12 / 256 / 2.4e9);
}

int main() {
    // matrix: 131072 rows x 5120 halves (10240 B rows) = 1.3 GB; each 256-row tile is re-read by ~8 blocks ("N tiles")
    const int nrows = 131072; const long ld = 10240; const int nk = 80;   // 512 distinct 256-row tiles, each read ~8x
    char* A; int* sink;
    CK(hipMalloc(&A, (size_t)(nrows + 512) * ld)); CK(hipMemset(A, 1, (size_t)(nrows + 512) * ld)); CK(hipMalloc(&sink, 64));
    // each tile of 256 rows is read by several blocks (as N tiles would): blocks = 2048, 2 tiles each
    const int nb = 2048;
    printf("pattern A: rows of 128 B at stride %ld B, %d K steps per tile\n", ld, nk);
    run<0, 4, 256, 2>("dma  4w 32KB/stage ring2 (cfg2-like)", A, ld, nrows, nk, sink, nb, 2);
    run<0, 4, 256, 3>("dma  4w 32KB/stage ring3", A, ld, nrows, nk, sink, nb, 2);
    run<0, 4, 256, 4>("dma  4w 32KB/stage ring4 (1 WG/CU)", A, ld, nrows, nk, sink, nb, 2);
    run<0, 8, 512, 2>("dma  8w 64KB/stage ring2 (cfg3-like)", A, ld, nrows, nk, sink, nb / 2, 2);
    run<0, 8, 256, 4>("dma  8w 32KB/stage ring4", A, ld, nrows, nk, sink, nb, 2);
    run<0, 8, 256, 5>("dma  8w 32KB/stage ring5", A, ld, nrows, nk, sink, nb, 2);
    run<1, 4, 256, 2>("regs 4w 32KB/stage 1 in flight", A, ld, nrows, nk, sink, nb, 2);
    run<1, 4, 256, 3>("regs 4w 32KB/stage 2 in flight", A, ld, nrows, nk, sink, nb, 2);
    run<1, 4, 256, 5>("regs 4w 32KB/stage 4 in flight", A, ld, nrows, nk, sink, nb, 2);
    run<1, 8, 512, 3>("regs 8w 64KB/stage 2 in flight", A, ld, nrows, nk, sink, nb / 2, 2);
    run<2, 4, 256, 3>("regs+ds_write 4w 32KB/stage 2 in flight", A, ld, nrows, nk, sink, nb, 2);
    run<2, 8, 512, 3>("regs+ds_write 8w 64KB/stage 2 in flight", A, ld, nrows, nk, sink, nb / 2, 2);
    // L2-resident variant: only 2048 rows (21 MB) -> mostly L2/MALL hits
    printf("pattern B: same, matrix limited to 1024 rows (10 MB: L2 + MALL resident)\n");
    run<0, 4, 256, 2>("dma  4w 32KB/stage ring2", A, ld, 1024, nk, sink, nb, 2);
    run<0, 8, 512, 2>("dma  8w 64KB/stage ring2", A, ld, 1024, nk, sink, nb / 2, 2);
    run<1, 4, 256, 3>("regs 4w 32KB/stage 2 in flight", A, ld, 1024, nk, sink, nb, 2);
    run<1, 4, 256, 5>("regs 4w 32KB/stage 4 in flight", A, ld, 1024, nk, sink, nb, 2);
    return 0;
}
