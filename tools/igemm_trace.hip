// igemm_trace: per-workgroup phase timing of the implicit-GEMM kernel (prologue / K loop / epilogue), standalone.
//   hipcc --offload-arch=gfx950 -O3 -DMOFA_IGEMM_TRACE -I include tools/igemm_trace.hip -o tools/igemm_trace.bin
//   MOFA_IGEMM_CFG=2 tools/igemm_trace.bin M N K [act]
#include "../mofa_video_amd/csrc/igemm.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    const int act = argc > 4 ? atoi(argv[4]) : 0;
    f16 *x, *w, *o;
    const int nout = act == 2 ? N / 2 : N;
    hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&o, (size_t)M * nout * 2);
    std::vector<f16> h((size_t)std::max(M, N) * K);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (f16)(((int)(s >> 16) % 2001 - 1000) / 1000.0f); }
    hipMemcpy(x, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    mofa_igemm_args a = {};
    a.x = x; a.w = w; a.out = o; a.M = M; a.N = N; a.Cin = K; a.ldx = K; a.ldo = nout; a.mode = 0; a.act = act;
    a.s_acc = 1.0f; a.rv_div = a.rv_mul = a.rv_mod_in = a.rv_mod_out = 1;
    if (argc > 5) a.ldx = atoi(argv[5]);   // timing diagnostic: ldx = 0 makes every activation row alias row 0 (all refill reads hit L1 / L2)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) if (mofa_igemm_f16(&a, nullptr)) { printf("launch failed\n"); return 1; }
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) mofa_igemm_f16(&a, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    static unsigned long long tr[8 * 1024];
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace), sizeof(tr));
    int nb = 0; while (nb < 1024 && tr[nb * 8 + 3] != 0) ++nb;
    double p = 0, l = 0, e = 0, nt = 0;
    double e4 = 0, e5 = 0, e6 = 0;
    for (int b = 0; b < nb; ++b) { p += tr[b * 8]; l += tr[b * 8 + 1]; e += tr[b * 8 + 2]; nt += tr[b * 8 + 3]; e4 += tr[b * 8 + 4]; e5 += tr[b * 8 + 5]; e6 += tr[b * 8 + 6]; }
    const double fl = 2.0 * M * (double)N * K;
    printf("M %d N %d K %d act %d: %.3f ms  %.0f TF/s  workgroups %d  tiles %.0f\n", M, N, K, act, ms, fl / ms / 1e9, nb, nt);
    printf("  ticks per tile: first-stage wait %.0f  K loop %.0f  epilogue %.0f  (per K step %.0f)\n", p / nt, l / nt, e / nt,
           l / nt / (K / 64));
    printf("  epilogue parts per tile: bias wait %.0f  phase 1 (acc -> slab) %.0f  phase 2 (passes) %.0f  rest %.0f\n", e4 / nt, e5 / nt, e6 / nt, e / nt);
    return 0;
}
