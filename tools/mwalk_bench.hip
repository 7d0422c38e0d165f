// Development harness for mwalk_kernel (mwalk_kernels.hip): checks it against the LDS-ring block kernel on the 48x48x32 layer
// shape, times both, and times a run of `nb` dependent blocks (BackCamera has seven in a row).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Iinclude -Irs-face-detection-tflite_amd/csrc tools/mwalk_bench.hip -o tools/bin/bb_mwalk   Run: tools/bin/bb_mwalk [B H act has_res nb]
#include "../rs-face-detection-tflite_amd/csrc/block_kernels.hip"
#include "../rs-face-detection-tflite_amd/csrc/mwalk_kernels.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mi;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.f - 1.f; }

int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 48;
    int act = argc > 3 ? atoi(argv[3]) : ACT_RELU, has_res = argc > 4 ? atoi(argv[4]) : 1, nb = argc > 5 ? atoi(argv[5]) : 7;
    const int W = argc > 6 ? atoi(argv[6]) : 48, C = argc > 7 ? atoi(argv[7]) : 32, Co = C;
    const bool ms = true;
    auto supports = [&](const BlockArgs& x) { return mwalk_kernel_supports(x); };
    auto launch = [&](const BlockArgs& x, hipStream_t st) { return launch_mwalk(x, st); };
    size_t n = (size_t)B * H * W * C;
    float *din, *dout, *dref;
    CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&dout, n * 4)); CK(hipMalloc(&dref, n * 4));
    unsigned seed = 4242;
    std::vector<float> hin(n);
    for (auto& v : hin) v = rnd(seed);
    CK(hipMemcpy(din, hin.data(), n * 4, hipMemcpyHostToDevice));
    auto up = [&](float** d, const std::vector<float>& h) { CK(hipMalloc(d, h.size() * 4)); CK(hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); };
    std::vector<BlockArgs> blk(nb);
    float* tmp[2]; CK(hipMalloc(&tmp[0], n * 4)); CK(hipMalloc(&tmp[1], n * 4));
    for (int k = 0; k < nb; k++) {
        int Cp, Cop; block_weight_dims(C, Co, &Cp, &Cop);
        std::vector<float> w(9 * C), bdw(C), bias(C), alpha(C), pw((size_t)Co * C), packed((size_t)Cop * Cp, 0.f), sc(mwalk_consts_floats(W, C, Co));
        for (auto& v : w) v = 0.3f * rnd(seed);
        for (auto& v : bdw) v = 0.1f * rnd(seed);
        for (auto& v : bias) v = 0.1f * rnd(seed);
        for (auto& v : alpha) v = 0.25f * rnd(seed);
        for (auto& v : pw) v = 0.15f * rnd(seed);
        const int Ch = Cp / 2, MT = Cop / 32;  // same A-fragment packing as engine.cpp
        for (int mt = 0; mt < MT; mt++) for (int j = 0; j < Ch / 4; j++) for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++) {
            int o = mt * 32 + (l & 31), c = (l >> 5) * Ch + 4 * j + e;
            if (o < Co && c < C) packed[(((size_t)mt * (Ch / 4) + j) * 64 + l) * 4 + e] = pw[(size_t)o * C + c];
        }
        mwalk_pack_consts(W, C, Co, w.data(), bdw.data(), pw.data(), bias.data(), alpha.data(), act, sc.data());
        float *dw, *db, *dpw, *dbp, *dal, *dst;
        up(&dw, w); up(&db, bdw); up(&dpw, packed); up(&dbp, bias); up(&dal, alpha); up(&dst, sc);
        BlockArgs a;
        a.in = k == 0 ? din : tmp[(k - 1) & 1]; a.out = tmp[k & 1]; a.in_fs = (long)H * W * C; a.out_fs = a.in_fs;
        a.w_dw = dw; a.b_dw = db; a.w_pw = dpw; a.w_strip = nullptr; a.w_mwalk = dst;
        a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = H; a.Wo = W; a.Co = Co; a.sh = a.sw = 1; a.pt = a.pl = 1;
        a.ep.bias = dbp; a.ep.act = act; a.ep.alpha = dal;
        if (has_res) { a.ep.res = a.in; a.ep.res_fs = a.in_fs; a.ep.res_C = C; a.ep.res_mode = RES_DIRECT; }
        blk[k] = a;
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    // single block: block kernel vs wide kernel on the same input
    BlockArgs a = blk[0]; a.out = dref;
    BlockArgs a2 = blk[0]; a2.out = dout;
    if (launch_block(a, s)) { printf("block launch failed\n"); return 1; }
    if (!supports(a2)) { printf("%s: unsupported\n", ms ? "mwalk " : "wide"); return 1; }
    CK(hipMemset(dout, 0xff, n * 4));
    if (int rc = launch(a2, s)) { printf("launch failed %d\n", rc); return 1; }
    CK(hipStreamSynchronize(s));
    std::vector<float> r0(n), r1(n);
    CK(hipMemcpy(r0.data(), dref, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), dout, n * 4, hipMemcpyDeviceToHost));
    double maxd = 0; size_t bad = 0, first = (size_t)-1;
    for (size_t i = 0; i < n; i++) {
        double d = std::fabs((double)r0[i] - r1[i]);
        if (!(d <= 1e-4 * std::max(1.0, std::fabs((double)r0[i])))) { if (!bad) first = i; bad++; }
        if (d > maxd || d != d) maxd = d;
    }
    printf("check: max |diff| %.3g, %zu of %zu outside 1e-4", maxd, bad, n);
    if (bad) { size_t p = first / C; printf("  first at frame %zu y %zu x %zu c %zu: ref %g got %g", p / ((size_t)H * W), p / W % H, p % W, first % C, r0[first], r1[first]); }
    printf("\n");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; which++) {
        const int it = 20;
        auto run = [&] { for (int k = 0; k < nb; k++) which ? launch(blk[k], s) : launch_block(blk[k], s); };
        for (int i = 0; i < 3; i++) run();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < it; i++) run();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        printf("%s x%d B %d %dx%d C %d : %.4f ms (%.4f per block)  %.1f GB/s  %.1f TFLOP/s algorithmic\n", which ? (ms ? "mwalk " : "wide ") : "block", nb, B, H, W, C, ms, ms / nb,
               2.0 * n * 4 * nb / ms / 1e6, 2.0 * B * H * W * C * (9.0 + Co) * nb / ms / 1e9);
    }
    return bad ? 2 : 0;
}
