// Micro-benchmark (r05): how much do MFMA and VALU / transcendental issue of ONE SIMD overlap on gfx950 -- inside one wave and between
// the two waves of a SIMD?  Decides whether a software-pipelined spatial-attention loop (one query block's softmax under the other's
// MFMAs) can beat the shipped kernel, whose waves run QK^T -> softmax -> PV strictly in sequence and rely on the partner wave.
//
// Per iteration a wave issues the per-key-tile mix of attn_spatial_kernel<64, 2>: 32 v_mfma_f32_32x32x16_f16, 64 v_exp_f32, 64 v_add_f32,
// 32 v_cvt_pkrtz (all on registers, no memory), in one of these orders (volatile asm keeps program order):
//   0  MFMAs only                       1  VALU only
//   2  blocked: 16 MFMA, 160 VALU, 16 MFMA           (the shipped kernel's order)
//   3  interleaved: (1 MFMA, 5 VALU) x 32            (software-pipelined order)
//   4  half-blocked: 8 MFMA, 40 VALU, ... x 4
// run with 4 waves per CU (1 per SIMD) and 8 (2 per SIMD; 512-thread workgroups or two 256-thread workgroups).
// build + run ON the GPU box (binaries do not travel with gpurun): hipcc --offload-arch=gfx950 -O3 tools/issue_overlap.hip -o /tmp/io.bin && /tmp/io.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA_(acc, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B))
#define MFMA(acc) MFMA_(acc, a[(m) & 3], b[((m) >> 2) & 3])
#define MFMA0(acc, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(A), "v"(B))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define ADD(x, y) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define CVT(d, x, y) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
// one "softmax unit" = 2 exp + 2 add + 1 cvt on rotating registers (5 VALU)
#define SM5(i) do { EXP(e[(i) & 7]); EXP(e[((i) + 1) & 7]); ADD(s0, e[(i) & 7]); ADD(s1, e[((i) + 1) & 7]); CVT(p[(i) & 3], e[(i) & 7], e[((i) + 1) & 7]); } while (0)

template <int MODE>
__global__ void __launch_bounds__(512, 1) k(float* out, long long* cyc, int iters, int rnd) {
    extern __shared__ char lds[];
    f16x8 a[4], b[4];
    {
        unsigned h = threadIdx.x * 2654435761u + 12345u;
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 8; ++i) {
                h = h * 1664525u + 1013904223u;
                const float u = rnd ? ((h >> 8) & 0xffff) / 65536.0f - 0.5f : 0.001f * (threadIdx.x & 63);
                a[j][i] = (_Float16)u;
                h = h * 1664525u + 1013904223u;
                b[j][i] = (_Float16)(rnd ? ((h >> 8) & 0xffff) / 65536.0f - 0.5f : 0.002f);
            }
    }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float e[8], s0 = 0.f, s1 = 0.f;
    unsigned p[4] = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) e[i] = -1.0f - 0.01f * i;
    f32x16 sa[2], sb[2];
    unsigned pa[2][8], pb_[2][8];
    if (MODE == 6) {                                              // pipeline prologue: the first half tile's scores
#pragma unroll
        for (int j = 0; j < 2; ++j) for (int r = 0; r < 8; ++r) { pa[j][r] = 0; pb_[j][r] = 0; }
#pragma unroll
        for (int m = 0; m < 8; ++m) { if (m < 2) MFMA0(sa[m & 1], a[m & 3], b[0]); else MFMA_(sa[m & 1], a[m & 3], b[(m >> 1) & 3]); }
    }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 32; ++m) MFMA(acc[m & 3]);
        } else if (MODE == 1) {
#pragma unroll
            for (int m = 0; m < 32; ++m) SM5(2 * m);
        } else if (MODE == 2) {
#pragma unroll
            for (int m = 0; m < 16; ++m) MFMA(acc[m & 3]);
#pragma unroll
            for (int m = 0; m < 32; ++m) SM5(2 * m);
#pragma unroll
            for (int m = 0; m < 16; ++m) MFMA(acc[m & 3]);
        } else if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < 32; ++m) { MFMA(acc[m & 3]); SM5(2 * m); }
        } else if (MODE == 5) {
            // true dependencies, blocked (the shipped order): S = K.Q (16 MFMAs, C = 0) -> p = exp2(S), row sums, fp16 pack -> O += V.P
            f32x16 sc[4];
            unsigned pk[4][8];
#pragma unroll
            for (int m = 0; m < 16; ++m) { if (m < 4) MFMA0(sc[m & 3], a[m & 3], b[0]); else MFMA_(sc[m & 3], a[m & 3], b[(m >> 2) & 3]); }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    float x = sc[j][r], y = sc[j][r + 1];
                    EXP(x); EXP(y); ADD(s0, x); ADD(s1, y); CVT(pk[j][r >> 1], x, y);
                }
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                f16x8 pb = __builtin_bit_cast(f16x8, *(const uint4*)&pk[m & 3][4 * ((m >> 2) & 1)]);
                MFMA_(acc[m & 3], a[m & 3], pb);
            }
        } else if (MODE == 6) {
            // true dependencies, software-pipelined by half tiles: per half h = 2 score tiles (8 QK MFMAs), 32 exps, 8 PV MFMAs;
            // body: QK(h+1) + PV(h-1) MFMAs interleaved with the softmax of h
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x16 (&cur)[2] = half ? sb : sa;
                f32x16 (&nxt)[2] = half ? sa : sb;
                unsigned (&pc)[2][8] = half ? pb_ : pa;
                unsigned (&pp)[2][8] = half ? pa : pb_;
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    if (m < 8) { if (m < 2) MFMA0(nxt[m & 1], a[m & 3], b[0]); else MFMA_(nxt[m & 1], a[m & 3], b[(m >> 1) & 3]); }
                    else { f16x8 q = __builtin_bit_cast(f16x8, *(const uint4*)&pp[m & 1][4 * ((m >> 1) & 1)]); MFMA_(acc[m & 3], a[m & 3], q); }
                    const int j = m >> 3, r = (m & 7) * 2;
                    float x = cur[j][r], y = cur[j][r + 1];
                    EXP(x); EXP(y); ADD(s0, x); ADD(s1, y); CVT(pc[j][r >> 1], x, y);
                }
            }
            // (the pipeline's carried state -- the next iteration's first QK and this one's last PV -- is folded: same instruction counts)
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int m = 0; m < 8; ++m) MFMA(acc[m & 3]);
#pragma unroll
                for (int m = 0; m < 8; ++m) SM5(2 * m);
            }
        }
        for (int i = 0; i < 8; ++i) e[i] = e[i] * 0.f - 1.0f;     // keep the exps' inputs finite (8 VALU per iteration, all modes)
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 6) p[0] ^= pa[0][0] ^ pb_[1][7];
    float r = s0 + s1 + (float)(p[0] ^ p[1] ^ p[2] ^ p[3]);
    for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) r += acc[j][q];
    if (r == 123.456f) out[0] = r;
    (void)lds;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int threads, int wg_per_cu, int iters, int rnd) {
    float* out;
    long long* cyc;
    hipMalloc(&out, 64);
    hipMalloc(&cyc, 8);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int grid = pr.multiProcessorCount * wg_per_cu;
    const size_t lds = wg_per_cu == 1 ? 100 * 1024 : 60 * 1024;       // forces exactly wg_per_cu workgroups per CU
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), lds, 0, out, cyc, iters / 10, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), lds, 0, out, cyc, iters, rnd);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double ns_it = ms * 1e6 / iters, waves_per_simd = threads / 256.0 * wg_per_cu;
    const double mfma = MODE == 1 ? 0 : 32;
    // TFLOP/s of the whole chip: waves/SIMD * 4 SIMDs * CUs * 32 MFMAs * 32768 flop per iteration time
    const double tf = mfma * 32768.0 * waves_per_simd * 4 * pr.multiProcessorCount / (ns_it * 1e-9) / 1e12;
    printf("%-44s waves/SIMD %.0f (%d x %d thr): %8.1f ns/iter  %8.0f memtime ticks/iter (%.2f GHz if ticks = cycles)  %7.1f TF/s MFMA\n", name,
           waves_per_simd, wg_per_cu, threads, ns_it, (double)c / iters, (double)c / iters / ns_it, tf);
    hipFree(out);
    hipFree(cyc);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    for (int rnd = 0; rnd < 2; ++rnd) {
        printf("---- MFMA operands: %s\n", rnd ? "random per lane, rotating between consecutive MFMAs" : "constant");
        for (int cfg = 0; cfg < 3; cfg += 2) {
            const int threads = 256, wpc = cfg == 2 ? 2 : 1;
            run<0>("0 MFMA only (32)", threads, wpc, iters, rnd);
            run<1>("1 VALU only (64 exp + 64 add + 32 cvt)", threads, wpc, iters, rnd);
            run<2>("2 blocked 16 MFMA | 160 VALU | 16 MFMA", threads, wpc, iters, rnd);
            run<3>("3 interleaved (1 MFMA + 5 VALU) x 32", threads, wpc, iters, rnd);
            run<5>("5 true deps, blocked (shipped order)", threads, wpc, iters, rnd);
            run<6>("6 true deps, pipelined by half tiles", threads, wpc, iters, rnd);
        }
    }
    return 0;
}
