// Development harness for mpipe_kernel (mpipe_kernels.hip): a chain of nb stride-1 24-channel blocks on the MFMA row pipeline
// against the packed-FMA row pipeline (strip_pipe2_kernel), results compared, both timed.
// Build: tools/build_mpipe_bench.sh [TAG -DFLAG ...]   Run: tools/bin/bb_mpipe [B H nb has_res act]
#include "../rs-face-detection-tflite_amd/csrc/block_kernels.hip"
#include "../rs-face-detection-tflite_amd/csrc/strip_kernels.hip"
#include "../rs-face-detection-tflite_amd/csrc/mpipe_kernels.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mi;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.f - 1.f; }

int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 128, nb = argc > 3 ? atoi(argv[3]) : 4;
    int has_res = argc > 4 ? atoi(argv[4]) : 1, act = argc > 5 ? atoi(argv[5]) : ACT_RELU;
    const int C = 24, W = H;
    size_t n = (size_t)B * H * W * C;
    float *din, *dref, *dout;
    CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&dref, n * 4)); CK(hipMalloc(&dout, n * 4));
    unsigned seed = 777;
    std::vector<float> hin(n);
    for (auto& v : hin) v = rnd(seed);
    CK(hipMemcpy(din, hin.data(), n * 4, hipMemcpyHostToDevice));
    auto up = [&](float** d, const std::vector<float>& h) { CK(hipMalloc(d, h.size() * 4)); CK(hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); };
    std::vector<BlockArgs> blk(nb);
    for (int k = 0; k < nb; k++) {
        std::vector<float> w(9 * C), bdw(C), bias(C), alpha(C), pw((size_t)C * C), sc(strip_consts_floats(C)), mc(mpipe_consts_floats(C));
        for (auto& v : w) v = 0.3f * rnd(seed);
        for (auto& v : bdw) v = 0.1f * rnd(seed);
        for (auto& v : bias) v = 0.1f * rnd(seed);
        for (auto& v : alpha) v = 0.25f * rnd(seed);
        for (auto& v : pw) v = 0.2f * rnd(seed);
        strip_pack_consts(C, w.data(), bdw.data(), pw.data(), bias.data(), alpha.data(), act, sc.data());
        mpipe_pack_consts(C, w.data(), bdw.data(), pw.data(), bias.data(), alpha.data(), act, mc.data());
        float *dsc, *dmc; up(&dsc, sc); up(&dmc, mc);
        BlockArgs a;
        a.in = din; a.out = dref; a.in_fs = (long)H * W * C; a.out_fs = a.in_fs;
        a.w_strip = dsc; a.w_mfma = dmc;
        a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = H; a.Wo = W; a.Co = C; a.sh = a.sw = 1; a.pt = a.pl = 1;
        a.ep.act = act;
        if (has_res) { a.ep.res = din; a.ep.res_fs = a.in_fs; a.ep.res_C = C; a.ep.res_mode = RES_DIRECT; }
        blk[k] = a;
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    if (!strip_pipe_supports(blk.data(), nb)) { printf("strip pipe: unsupported\n"); return 1; }
    if (int rc = launch_strip_pipe(blk.data(), nb, s)) { printf("strip pipe launch failed %d\n", rc); return 1; }
    std::vector<BlockArgs> mb = blk; mb[nb - 1].out = dout;
    if (!mpipe_supports(mb.data(), nb)) { printf("mpipe: unsupported\n"); return 1; }
    CK(hipMemset(dout, 0xff, n * 4));
    if (int rc = launch_mpipe(mb.data(), nb, s)) { printf("mpipe launch failed %d\n", rc); return 1; }
    CK(hipStreamSynchronize(s));
    std::vector<float> r0(n), r1(n);
    CK(hipMemcpy(r0.data(), dref, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), dout, n * 4, hipMemcpyDeviceToHost));
    double maxd = 0; size_t bad = 0, first = (size_t)-1;
    for (size_t i = 0; i < n; i++) {
        double d = std::fabs((double)r0[i] - r1[i]);
        if (!(d <= 1e-4 * std::max(1.0, std::fabs((double)r0[i])))) { if (!bad) first = i; bad++; }
        if (d > maxd || d != d) maxd = d;
    }
    size_t nz = 0, eq = 0; for (size_t i = 0; i < n; i++) { nz += r0[i] != 0.f; eq += r0[i] == r1[i]; }
    printf("samples: ref %g %g %g  got %g %g %g  nonzero %zu identical %zu\n", r0[12345], r0[n / 2 + 7], r0[n - 5], r1[12345], r1[n / 2 + 7], r1[n - 5], nz, eq);
    printf("check: max |diff| %.3g, %zu of %zu outside 1e-4", maxd, bad, n);
    if (bad) { size_t p = first / C; printf("  first at frame %zu y %zu x %zu c %zu: ref %g got %g", p / ((size_t)H * W), p / W % H, p % W, first % C, r0[first], r1[first]); }
    printf("\n");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; which++) {
        const int it = 20;
        auto run = [&] { which ? launch_mpipe(mb.data(), nb, s) : launch_strip_pipe(blk.data(), nb, s); };
        for (int i = 0; i < 3; i++) run();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < it; i++) run();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        printf("%s x%d B %d %dx%d C %d : %.4f ms (%.4f per block)  %.1f TFLOP/s algorithmic\n", which ? "mpipe " : "strip2", nb, B, H, W, C, ms, ms / nb,
               2.0 * B * H * W * C * (9.0 + C) * nb / ms / 1e9);
    }
    return bad ? 2 : 0;
}
