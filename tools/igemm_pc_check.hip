// igemm_pc_check: bit-exactness / determinism / time of the experimental producer-consumer kernel (mofa_video_amd/csrc/igemm_pc.inc)
// For a list of shapes / epilogue kinds it launches the shipped kernels (reference) and the producer-consumer kernel on the same
// operands, compares the outputs bit for bit (every output element accumulates its K terms in the same order in both, so
// they must be identical), repeats its launch for determinism, and times both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMOFA_IGEMM_PC -I include tools/igemm_pc_check.hip -o tools/igemm_pc_check.bin
//   timeout 60 tools/igemm_pc_check.bin            (always under a timeout: a wrong barrier / wait can hang the GPU)
#include "../mofa_video_amd/csrc/igemm.hip"
#include <cstdio>
#include <cstring>
#include <vector>

struct Case { const char* name; int M, N, K, act, r1, rv, bias, conv; };

int main() {
    const Case cases[] = {                                       // kinds the experiment instantiates: plain, r1, GEGLU
        {"plain 115200x640x2560", 115200, 640, 2560, 0, 0, 0, 0, 0},
        {"plain+bias+r1 28800x1280x5120", 28800, 1280, 5120, 0, 1, 0, 1, 0},
        {"GEGLU 115200x5120x640", 115200, 5120, 640, 2, 0, 0, 1, 0},
        {"ragged M/N 7200x328x1280 +bias", 7200, 328, 1280, 0, 0, 0, 1, 0},
        {"nk=1 4096x256x64 +r1", 4096, 256, 64, 0, 1, 0, 0, 0},
        {"conv3x3 8x64x64 C320 -> 320 +bias", 8 * 64 * 64, 320, 320, 0, 0, 0, 1, 1},
    };
    size_t maxMK = 0, maxNK = 0, maxMN = 0;
    for (const Case& c : cases) {
        const size_t kt = (size_t)c.K * (c.conv ? 9 : 1);
        maxMK = std::max(maxMK, (size_t)c.M * c.K); maxNK = std::max(maxNK, (size_t)c.N * kt); maxMN = std::max(maxMN, (size_t)c.M * c.N);
    }
    f16 *x, *w, *r, *o; float *b, *rv;
    hipMalloc(&x, maxMK * 2); hipMalloc(&w, maxNK * 2); hipMalloc(&r, maxMN * 2); hipMalloc(&o, maxMN * 2);
    hipMalloc(&b, 16384 * 4); hipMalloc(&rv, 64 * 16384 * 4);
    std::vector<f16> h(std::max(std::max(maxMK, maxNK), maxMN)); unsigned s = 7;
    auto fill = [&](f16* d, size_t cnt, float sc) { for (size_t i = 0; i < cnt; ++i) { s = s * 1664525u + 1013904223u; h[i] = (f16)(((int)(s >> 16) % 2001 - 1000) / 1000.0f * sc); } hipMemcpy(d, h.data(), cnt * 2, hipMemcpyHostToDevice); };
    fill(x, maxMK, 1.0f); fill(w, maxNK, 0.03f); fill(r, maxMN, 1.0f);
    { std::vector<float> hb(16384); for (int i = 0; i < 16384; ++i) hb[i] = 0.01f * (float)(i % 53) - 0.2f; hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice); }
    { std::vector<float> hv((size_t)64 * 16384); for (size_t i = 0; i < hv.size(); ++i) hv[i] = 0.002f * (float)(i % 211); hipMemcpy(rv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int failures = 0;
    for (const Case& c : cases) {
        mofa_igemm_args a = {};
        const int nout = c.act == 2 ? c.N / 2 : c.N;
        a.x = x; a.w = w; a.out = o; a.M = c.M; a.N = c.N; a.Cin = c.K; a.ldx = c.K; a.ldo = nout; a.act = c.act;
        a.s_acc = 1.0f; a.s1 = 0.5f; a.rv_div = a.rv_mul = a.rv_mod_in = a.rv_mod_out = 1;
        if (c.conv) { a.mode = MOFA_MODE_CONV3X3; a.Hin = a.Hout = 64; a.Win = a.Wout = 64; a.stride = 1; a.up = 1; a.ksize = 3; }
        if (c.r1) { a.r1 = r; a.ldr1 = nout; }
        if (c.bias) a.bias = b;
        if (c.rv) { a.rowvec = rv; a.rv_div = c.conv ? 64 * 64 : 900; a.rv_mul = 1; a.rv_mod_in = 1; a.rv_mod_out = 8; }
        const size_t cnt = (size_t)c.M * nout;
        std::vector<f16> ref(cnt), cur(cnt);
        float ms[2] = {0, 0};
        size_t diff = 0, nondet = 0;
        for (int mode = 0; mode < 2; ++mode) {
            s_pc_on = mode;
            for (int it = 0; it < (mode ? 4 : 1); ++it) {
                hipMemset(o, 0xff, cnt * 2);
                if (mofa_igemm_f16(&a, nullptr)) { printf("%s: launch failed (mode %d)\n", c.name, mode); return 2; }
                if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel fault (mode %d)\n", c.name, mode); return 3; }
                std::vector<f16>& dst = (mode == 0) ? ref : cur;
                std::vector<f16> tmp;
                if (mode && it) { tmp.resize(cnt); hipMemcpy(tmp.data(), o, cnt * 2, hipMemcpyDeviceToHost); nondet += memcmp(tmp.data(), cur.data(), cnt * 2) ? 1 : 0; }
                else hipMemcpy(dst.data(), o, cnt * 2, hipMemcpyDeviceToHost);
            }
            hipEventRecord(e0);
            for (int it = 0; it < 5; ++it) mofa_igemm_f16(&a, nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1); ms[mode] /= 5;
        }
        size_t first = 0;
        for (size_t i = 0; i < cnt; ++i) if (memcmp(&ref[i], &cur[i], 2)) { if (!diff) first = i; ++diff; }
        const double fl = 2.0 * c.M * (double)c.N * c.K * (c.conv ? 9 : 1);
        printf("%-40s shipped %.3f ms (%.0f TF/s)  pc %.3f ms (%.0f TF/s)  differing %zu of %zu%s  nondeterministic repeats %zu/3\n", c.name,
               ms[0], fl / ms[0] / 1e9, ms[1], fl / ms[1] / 1e9, diff, cnt, diff ? "" : " (bit-identical)", nondet);
        if (diff) printf("    first difference at row %zu col %zu: shipped %.4f pc %.4f\n", first / nout, first % nout, (float)ref[first], (float)cur[first]);
        failures += (diff || nondet) ? 1 : 0;
    }
    printf(failures ? "FAILED: %d case(s)\n" : "all cases bit-identical and deterministic\n", failures);
    return failures ? 1 : 0;
}
