// igemm_det: determinism check of one conv3x3 + residual launch shape (standalone; variants via -D flags)
#include "../mofa_video_amd/csrc/igemm.hip"
#include <cstdio>
#include <vector>
#include <cstring>
int main(int argc, char** argv) {
    const int n = 8, H = 128, W = 128, C = 64, N = argc > 1 ? atoi(argv[1]) : 64;
    const int use_r1 = argc > 2 ? atoi(argv[2]) : 1, use_bias = argc > 3 ? atoi(argv[3]) : 0;
    const int M = n * H * W, K = 9 * C;
    f16 *x, *w, *r, *o; float* b;
    hipMalloc(&x, (size_t)M * C * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&r, (size_t)M * N * 2); hipMalloc(&o, (size_t)M * N * 2);
    hipMalloc(&b, N * 4);
    std::vector<f16> h((size_t)M * std::max(C, N)); unsigned s = 1;
    auto fill = [&](f16* d, size_t cnt, float sc) { for (size_t i = 0; i < cnt; ++i) { s = s * 1664525u + 1013904223u; h[i] = (f16)(((int)(s >> 16) % 2001 - 1000) / 1000.0f * sc); } hipMemcpy(d, h.data(), cnt * 2, hipMemcpyHostToDevice); };
    fill(x, (size_t)M * C, 1.0f); fill(w, (size_t)N * K, 0.05f); fill(r, (size_t)M * N, 1.0f);
    if (argc > 4 && atoi(argv[4])) { for (size_t i = 0; i < (size_t)M * N; ++i) h[i] = (f16)1.0f; hipMemcpy(r, h.data(), (size_t)M * N * 2, hipMemcpyHostToDevice); }
    std::vector<float> hb(N, 0.5f); hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
    mofa_igemm_args a = {};
    a.x = x; a.w = w; a.out = o; a.M = M; a.N = N; a.Cin = C; a.ldx = C; a.ldo = N; a.mode = MOFA_MODE_CONV3X3;
    a.Hin = a.Hout = H; a.Win = a.Wout = W; a.stride = 1; a.up = 1; a.ksize = 3;
    a.s_acc = 1.0f; a.s1 = 1.0f; a.rv_div = a.rv_mul = a.rv_mod_in = a.rv_mod_out = 1;
    if (use_r1) { a.r1 = r; a.ldr1 = N; }
    if (use_bias) a.bias = b;
    const int use_r2 = argc > 5 ? atoi(argv[5]) : 0, use_rv = argc > 6 ? atoi(argv[6]) : 0;
    if (use_r2) { a.r2 = x; a.ldr2 = C; a.s2 = 0.5f; }          // (only valid when N <= C; residual 2 = the input tensor)
    float* rvp; hipMalloc(&rvp, (size_t)n * N * 4);
    { std::vector<float> hv((size_t)n * N); for (size_t i = 0; i < hv.size(); ++i) hv[i] = 0.01f * (float)(i % 97); hipMemcpy(rvp, hv.data(), hv.size() * 4, hipMemcpyHostToDevice); }
    if (use_rv) { a.rowvec = rvp; a.rv_div = H * W; a.rv_mul = 1; a.rv_mod_in = 1; a.rv_mod_out = n; }
    std::vector<f16> ref((size_t)M * N), cur((size_t)M * N);
    int bad_runs = 0; size_t bad_elems = 0;
    for (int it = 0; it < 8; ++it) {
        hipMemset(o, 0, (size_t)M * N * 2);
        if (mofa_igemm_f16(&a, nullptr)) { printf("launch failed\n"); return 1; }
        hipDeviceSynchronize();
        hipMemcpy(it ? cur.data() : ref.data(), o, (size_t)M * N * 2, hipMemcpyDeviceToHost);
        if (it) {
            size_t d = 0; size_t first = 0;
            for (size_t i = 0; i < (size_t)M * N; ++i) if (memcmp(&cur[i], &ref[i], 2)) { if (!d) first = i; if (0) printf("    (%zu,%zu) tile row %zu: %.3f vs %.3f\n", i / N, i % N, (i / N) % 128, (float)cur[i], (float)ref[i]); ++d; }
            if (d && bad_runs < 1) {
                std::vector<f16> hx((size_t)M * C), hw((size_t)N * K), hr_((size_t)M * N);
                hipMemcpy(hx.data(), x, hx.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hw.data(), w, hw.size() * 2, hipMemcpyDeviceToHost);
                hipMemcpy(hr_.data(), r, hr_.size() * 2, hipMemcpyDeviceToHost);
                int shown = 0;
                for (size_t i = 0; i < (size_t)M * N && shown < 12; ++i) if (memcmp(&cur[i], &ref[i], 2)) {
                    const size_t m = i / N, nn = i % N; const int img = m / (H * W), oy = (m / W) % H, ox = m % W;
                    double acc = 0;
                    for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) { const int iy = oy + ky - 1, ix = ox + kx - 1; if (iy < 0 || ix < 0 || iy >= H || ix >= W) continue;
                        for (int c = 0; c < C; ++c) acc += (double)hx[((size_t)(img * H + iy) * W + ix) * C + c] * (double)hw[nn * K + (ky * 3 + kx) * C + c]; }
                    printf("    (%zu,%zu): this run %.4f  first run %.4f  | conv %.4f  r1 %.4f  conv+r1 %.4f\n", m, nn, (float)cur[i], (float)ref[i], acc, (float)hr_[i], acc + (float)hr_[i]);
                    ++shown;
                }
            }
            if (d && bad_runs < 2) {
                int hr[128] = {0}, hc[8] = {0};
                for (size_t i = 0; i < (size_t)M * N; ++i) if (memcmp(&cur[i], &ref[i], 2)) { hr[(i / N) % 128]++; hc[(i % N) % 8]++; }
                printf("    rows(mod 128):"); for (int q = 0; q < 128; ++q) if (hr[q]) printf(" %d:%d", q, hr[q]);
                printf("\n    cols(mod 8):"); for (int q = 0; q < 8; ++q) printf(" %d", hc[q]); printf("\n");
            }
            if (d) { ++bad_runs; bad_elems += d; if (bad_runs <= 2) printf("  run %d: %zu differing elements, first at row %zu col %zu\n", it, d, first / N, first % N); }
        }
    }
    printf("N %d r1 %d bias %d r2 %d rv %d: %d of 7 repeat runs differ (%zu elements)\n", N, use_r1, use_bias, use_r2, use_rv, bad_runs, bad_elems);
    return 0;
}
