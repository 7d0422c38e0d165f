// Development harness for the fused BlazeBlock kernel: times one layer shape and (with -DMI_BLOCK_STAMPS) prints the
// share of wave cycles per phase.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off [-DMI_BLOCK_STAMPS] tools/block_bench.hip -o /tmp/bb
#include "../rs-face-detection-tflite_amd/csrc/block_kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mi;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 128, C = argc > 3 ? atoi(argv[3]) : 24, Co = argc > 4 ? atoi(argv[4]) : 24;
    int S = argc > 5 ? atoi(argv[5]) : 1;
    const bool skip = argc > 6 ? atoi(argv[6]) != 0 : true;  // 0: no skip connection (the contracting blocks of full_range)
    int W = H, Ho = H / S, Wo = W / S;
    size_t nin = (size_t)B * H * W * C, nout = (size_t)B * Ho * Wo * Co;
    float *din, *dout, *dw, *db, *dpw, *dbp;
    CK(hipMalloc(&din, nin * 4)); CK(hipMalloc(&dout, nout * 4));
    std::vector<float> hin(nin);
    for (size_t i = 0; i < nin; i++) hin[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice));
    int Cp, Cop; block_weight_dims(C, Co, &Cp, &Cop);
    std::vector<float> w(9 * C, 0.1f), bz(std::max(C, Co), 0.01f), pw((size_t)Cop * Cp, 0.02f);
    CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&db, bz.size() * 4)); CK(hipMalloc(&dpw, pw.size() * 4)); CK(hipMalloc(&dbp, bz.size() * 4));
    CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bz.data(), bz.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpw, pw.data(), pw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbp, bz.data(), bz.size() * 4, hipMemcpyHostToDevice));
    BlockArgs a;
    a.in = din; a.out = dout; a.in_fs = (long)H * W * C; a.out_fs = (long)Ho * Wo * Co;
    a.w_dw = dw; a.b_dw = db; a.w_pw = dpw;
    a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.sh = a.sw = S; a.pt = a.pl = S == 1 ? 1 : 0;
    a.ep.bias = dbp; a.ep.act = ACT_RELU; a.ep.res = din; a.ep.res_fs = a.in_fs; a.ep.res_C = C; a.ep.res_mode = S == 1 ? RES_DIRECT : RES_MAXPOOL;
    if (!skip || Co < C) { a.ep.res = nullptr; a.ep.res_mode = RES_NONE; a.ep.res_C = 0; }
    BlockGeom g;
    if (!make_geom(a, &g)) { printf("unsupported\n"); return 1; }
    int PG = g.PG;
    printf("geom: Cp %d PS %d R %d NR %d band %d bands %d MT %d a_lds %d lds %d B  grid %d  PG %d  nsplit %d wmod %d mt_per %d\n", g.Cp, g.PS, g.R, g.NR, g.band, g.bands, g.MT, g.a_lds, g.lds_bytes, B * g.bands * g.nsplit, PG, g.nsplit, g.wmod, g.mt_per);
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int i = 0; i < 3; i++) if (launch_block(a, s)) { printf("launch failed\n"); return 1; }
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int n = 20;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < n; i++) launch_block(a, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
    double bytes = (double)(nin + nout) * 4;
    printf("B %d %dx%d C %d->%d s%d : %.3f ms  %.1f GB/s algorithmic  %.2f TFLOP/s\n", B, H, W, C, Co, S, ms, bytes / ms / 1e6, 2.0 * B * Ho * Wo * C * (9.0 + Co) / ms / 1e9);
#ifdef MI_BLOCK_STAMPS
    {
        size_t nw = (size_t)B * g.bands * g.nsplit * 4;
        unsigned long long* dst; CK(hipMalloc(&dst, nw * 8 * 8)); CK(hipMemset(dst, 0, nw * 64));
        // relaunch once with stamps through the same dispatch table
        g_stamp_ptr = dst;
        launch_block(a, s); CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(nw * 8);
        CK(hipMemcpy(h.data(), dst, nw * 64, hipMemcpyDeviceToHost));
        double acc[6] = {0}; 
        for (size_t i = 0; i < nw; i++) for (int k = 0; k < 6; k++) acc[k] += h[i * 8 + k];
        const char* nm[6] = {"prefetch-issue", "dw+mfma", "epilogue", "barrier1", "lds-write", "barrier2"};
        double tot = 0; for (int k = 0; k < 6; k++) tot += acc[k];
        int steps = (g.band + g.R - 1) / g.R;
        for (int k = 0; k < 6; k++) printf("  %-15s %6.1f%%  %8.0f cycles/step\n", nm[k], 100 * acc[k] / tot, acc[k] / nw / steps);
        double pro = 0; for (size_t i = 0; i < nw; i++) pro += h[i * 8 + 6];
        printf("  total %.0f cycles/step/wave; prologue %.0f cycles/wave = %.1f steps; steps/band %d\n", tot / nw / steps, pro / nw, pro / nw / (tot / nw / steps), steps);
    }
#endif
    return 0;
}
