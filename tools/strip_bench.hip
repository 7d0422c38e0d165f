// Development harness for the register-resident strip kernel: checks it against the LDS-ring block kernel on one layer
// shape and times both.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Irs-face-detection-tflite_amd/csrc tools/strip_bench.hip -o tools/bb_strip
#include "../rs-face-detection-tflite_amd/csrc/block_kernels.hip"
#include "../rs-face-detection-tflite_amd/csrc/strip_kernels.hip"

#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mi;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.f - 1.f; }

int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 128, C = argc > 3 ? atoi(argv[3]) : 24;
    int W = argc > 4 ? atoi(argv[4]) : H, act = argc > 5 ? atoi(argv[5]) : ACT_RELU, has_res = argc > 6 ? atoi(argv[6]) : 1;
    int Co = C;
    size_t n = (size_t)B * H * W * C;
    float *din, *dout, *dref, *dw, *db, *dpw, *dbp, *dal, *dst;
    CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&dout, n * 4)); CK(hipMalloc(&dref, n * 4));
    unsigned seed = 12345;
    std::vector<float> hin(n);
    for (auto& v : hin) v = rnd(seed);
    CK(hipMemcpy(din, hin.data(), n * 4, hipMemcpyHostToDevice));
    int Cp, Cop; block_weight_dims(C, Co, &Cp, &Cop);
    std::vector<float> w(9 * C), bdw(C), bias(C), alpha(C), pw((size_t)Co * C), packed((size_t)Cop * Cp, 0.f), sc(strip_consts_floats(C));
    for (auto& v : w) v = 0.3f * rnd(seed);
    for (auto& v : bdw) v = 0.1f * rnd(seed);
    for (auto& v : bias) v = 0.1f * rnd(seed);
    for (auto& v : alpha) v = 0.25f * rnd(seed);
    for (auto& v : pw) v = 0.2f * rnd(seed);
    {   // same A-fragment packing as engine.cpp
        const int Ch = Cp / 2, MT = Cop / 32;
        for (int mt = 0; mt < MT; mt++) for (int j = 0; j < Ch / 4; j++) for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++) {
            int o = mt * 32 + (l & 31), c = (l >> 5) * Ch + 4 * j + e;
            if (o < Co && c < C) packed[(((size_t)mt * (Ch / 4) + j) * 64 + l) * 4 + e] = pw[(size_t)o * C + c];
        }
    }
    strip_pack_consts(C, w.data(), bdw.data(), pw.data(), bias.data(), alpha.data(), act, sc.data());
    auto up = [&](float** d, const std::vector<float>& h) { CK(hipMalloc(d, h.size() * 4)); CK(hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); };
    up(&dw, w); up(&db, bdw); up(&dpw, packed); up(&dbp, bias); up(&dal, alpha); up(&dst, sc);
    BlockArgs a;
    a.in = din; a.out = dref; a.in_fs = (long)H * W * C; a.out_fs = a.in_fs;
    a.w_dw = dw; a.b_dw = db; a.w_pw = dpw; a.w_strip = dst;
    a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = H; a.Wo = W; a.Co = Co; a.sh = a.sw = 1; a.pt = a.pl = 1;
    a.ep.bias = dbp; a.ep.act = act; a.ep.alpha = dal;
    if (has_res) { a.ep.res = din; a.ep.res_fs = a.in_fs; a.ep.res_C = C; a.ep.res_mode = RES_DIRECT; }
    hipStream_t s; CK(hipStreamCreate(&s));
    if (launch_block(a, s)) { printf("block launch failed\n"); return 1; }
    CK(hipStreamSynchronize(s));
    BlockArgs a2 = a; a2.out = dout;
    if (!strip_kernel_supports(a2)) { printf("strip: unsupported\n"); return 1; }
    CK(hipMemset(dout, 0xff, n * 4));
    if (launch_strip(a2, s)) { printf("strip launch failed\n"); return 1; }
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
        for (int i = 0; i < 3; i++) which ? launch_strip(a2, s) : launch_block(a, s);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < it; i++) which ? launch_strip(a2, s) : launch_block(a, s);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        printf("%s B %d %dx%d C %d : %.4f ms  %.1f GB/s algorithmic\n", which ? "strip" : "block", B, H, W, C, ms, 2.0 * n * 4 / ms / 1e6);
    }
    int nb = argc > 7 ? atoi(argv[7]) : 1;
    int tail = argc > 8 ? atoi(argv[8]) : 0;  // 1 / 2: the last of the nb blocks is a stride-2 block with Co = tail * C
    if (nb > 1) {  // row-pipelined chain of nb blocks vs one launch per block
        std::vector<BlockArgs> blk(nb);
        const int Ho = H / 2, Wo = W / 2, Co2 = tail * C;
        const size_t nlast = tail ? (size_t)B * Ho * Wo * Co2 : n;
        float *t0, *t1, *dpipe, *dlast; CK(hipMalloc(&t0, n * 4)); CK(hipMalloc(&t1, n * 4)); CK(hipMalloc(&dpipe, nlast * 4)); CK(hipMalloc(&dlast, nlast * 4));
        for (int k = 0; k < nb; k++) {
            const bool is_tail = tail && k == nb - 1;
            const int Cok = is_tail ? Co2 : C;
            std::vector<float> w2(9 * C), bdw2(C), bias2(Cok), pw2((size_t)Cok * C), al2(Cok, 0.1f);
            for (auto& v : w2) v = 0.3f * rnd(seed);
            for (auto& v : bdw2) v = 0.1f * rnd(seed);
            for (auto& v : bias2) v = 0.1f * rnd(seed);
            for (auto& v : pw2) v = 0.2f * rnd(seed);
            blk[k] = a2;
            blk[k].in = k == 0 ? din : (k % 2 ? t0 : t1);
            blk[k].out = k == nb - 1 ? dlast : (k % 2 ? t1 : t0);
            if (!is_tail) {
                std::vector<float> sc2(strip_consts_floats(C));
                strip_pack_consts(C, w2.data(), bdw2.data(), pw2.data(), bias2.data(), alpha.data(), act, sc2.data());
                float* dsc; up(&dsc, sc2);
                blk[k].w_strip = dsc;
                if (has_res) blk[k].ep.res = blk[k].in;
            } else {
                std::vector<float> sc2(strip_consts_s2_floats(C, Co2));
                strip_pack_consts_s2(C, Co2, w2.data(), bdw2.data(), pw2.data(), bias2.data(), al2.data(), act, sc2.data());
                float *dsc, *dw2, *db2, *dp2, *dbb; up(&dsc, sc2); up(&dw2, w2); up(&db2, bdw2); up(&dbb, bias2);
                int Cp2, Cop2; block_weight_dims(C, Co2, &Cp2, &Cop2);
                std::vector<float> packed2((size_t)Cop2 * Cp2, 0.f);
                const int Ch = Cp2 / 2, MT = Cop2 / 32;
                for (int mt = 0; mt < MT; mt++) for (int j = 0; j < Ch / 4; j++) for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++) {
                    int o = mt * 32 + (l & 31), c = (l >> 5) * Ch + 4 * j + e;
                    if (o < Co2 && c < C) packed2[(((size_t)mt * (Ch / 4) + j) * 64 + l) * 4 + e] = pw2[(size_t)o * C + c];
                }
                up(&dp2, packed2);
                BlockArgs& t = blk[k];
                t.w_strip = dsc; t.w_dw = dw2; t.b_dw = db2; t.w_pw = dp2;
                t.Ho = Ho; t.Wo = Wo; t.Co = Co2; t.sh = t.sw = 2; t.pt = t.pl = 0; t.out_fs = (long)Ho * Wo * Co2;
                t.ep.bias = dbb; t.ep.res = t.in; t.ep.res_fs = t.in_fs; t.ep.res_C = C; t.ep.res_H = H; t.ep.res_W = W; t.ep.res_mode = RES_MAXPOOL;
            }
        }
        for (int k = 0; k < nb; k++) {
            const bool is_tail = tail && k == nb - 1;
            if (is_tail ? launch_block(blk[k], s) : launch_strip(blk[k], s)) { printf("reference launch failed\n"); return 1; }
        }
        CK(hipStreamSynchronize(s));
        if (!strip_pipe_supports(blk.data(), nb)) { printf("pipe: unsupported\n"); return 1; }
        std::vector<BlockArgs> pb = blk; pb[nb - 1].out = dpipe;
        CK(hipMemset(dpipe, 0xff, nlast * 4));
        int rc = launch_strip_pipe(pb.data(), nb, s);
        if (rc) { printf("pipe launch failed %d\n", rc); return 1; }
        CK(hipStreamSynchronize(s));
        r0.resize(nlast); r1.resize(nlast);
        CK(hipMemcpy(r0.data(), dlast, nlast * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), dpipe, nlast * 4, hipMemcpyDeviceToHost));
        const double tol = tail ? 1e-4 : 1e-5;
        const int Cl = tail ? Co2 : C, Wl = tail ? Wo : W, Hl = tail ? Ho : H;
        double md = 0; size_t nbad = 0, fi = (size_t)-1;
        for (size_t i = 0; i < nlast; i++) { double d = std::fabs((double)r0[i] - r1[i]); if (!(d <= tol * std::max(1.0, std::fabs((double)r0[i])))) { if (!nbad) fi = i; nbad++; } if (d > md || d != d) md = d; }
        printf("pipe x%d%s check: max |diff| %.3g, %zu of %zu outside %.0e", nb, tail ? " (stride-2 tail)" : "", md, nbad, nlast, tol);
        if (nbad) { size_t px = fi / Cl; printf("  first at frame %zu y %zu x %zu c %zu: ref %g got %g", px / ((size_t)Hl * Wl), px / Wl % Hl, px % Wl, fi % Cl, r0[fi], r1[fi]); }
        printf("\n");
        for (int hint : {2, 4}) {   // the packed-FMA two-row form (hint 2) and the one-row MFMA form (hint 4) against the default (two rows, MFMA): bit for bit
            std::vector<BlockArgs> pv = pb;
            for (auto& b : pv) b.pipe_rows = hint;
            float* dpv; CK(hipMalloc(&dpv, nlast * 4)); CK(hipMemset(dpv, 0xff, nlast * 4));
            pv[nb - 1].out = dpv;
            if (launch_strip_pipe(pv.data(), nb, s)) { printf("pipe (hint %d) launch failed\n", hint); return 1; }
            CK(hipStreamSynchronize(s));
            std::vector<float> r2(nlast);
            CK(hipMemcpy(r2.data(), dpv, nlast * 4, hipMemcpyDeviceToHost));
            size_t ndiff = 0;
            for (size_t i = 0; i < nlast; i++) ndiff += memcmp(&r2[i], &r1[i], 4) != 0;
            char l0[96], l1[96];
            printf("%s vs %s: %zu of %zu values differ in any bit\n", strip_pipe_label(pb.data(), nb, l0, sizeof l0), strip_pipe_label(pv.data(), nb, l1, sizeof l1), ndiff, nlast);
            if (ndiff) nbad += ndiff;
            const int it = 20;
            for (int i = 0; i < 3; i++) launch_strip_pipe(pv.data(), nb, s);
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < it; i++) launch_strip_pipe(pv.data(), nb, s);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
            printf("pipe (%s) x%d B %d %dx%d C %d : %.4f ms (%.4f per block)\n", strip_pipe_label(pv.data(), nb, l1, sizeof l1), nb, B, H, W, C, ms, ms / nb);
        }
        for (int which = 0; which < 2; which++) {
            const int it = 20;
            auto run = [&] { if (which) launch_strip_pipe(pb.data(), nb, s); else for (int k = 0; k < nb; k++) { if (tail && k == nb - 1) launch_block(blk[k], s); else launch_strip(blk[k], s); } };
            for (int i = 0; i < 3; i++) run();
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < it; i++) run();
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
            printf("%s x%d B %d %dx%d C %d : %.4f ms (%.4f per block)\n", which ? "pipe " : "separate", nb, B, H, W, C, ms, ms / nb);
        }
#ifdef MI_PIPE_STAMPS
        for (int mf = 0; mf < 2; mf++) {   // where a wave's time goes, per role: s_memtime ticks per step
            std::vector<BlockArgs> pv = pb;
            for (auto& b : pv) b.pipe_rows = mf ? 0 : 2;
            const size_t nw = (size_t)B * 64 * 8;  // upper bound on workgroups x 8 waves
            unsigned long long* dstp; CK(hipMalloc(&dstp, nw * 64)); CK(hipMemset(dstp, 0, nw * 64));
            g_strip_stamps = dstp;
            launch_strip_pipe(pv.data(), nb, s); CK(hipStreamSynchronize(s));
            g_strip_stamps = nullptr;
            std::vector<unsigned long long> h(nw * 8);
            CK(hipMemcpy(h.data(), dstp, nw * 64, hipMemcpyDeviceToHost));
            char lb[96];
            printf("stamps %s: ticks per step per wave [compute | barrier 1 | hand-over / store | barrier 2]\n", strip_pipe_label(pv.data(), nb, lb, sizeof lb));
            for (int w = 0; w < 8; w++) {
                double acc[4] = {0, 0, 0, 0}, steps = 0; size_t cnt = 0;
                for (size_t g = 0; g < nw / 8; g++) { const unsigned long long* d = &h[(g * 8 + w) * 8]; if (!d[4]) continue; cnt++; steps += d[4]; for (int k = 0; k < 4; k++) acc[k] += d[k]; }
                if (!cnt) continue;
                double sub[3] = {0, 0, 0};
                for (size_t g = 0; g < nw / 8; g++) { const unsigned long long* d = &h[(g * 8 + w) * 8]; if (!d[4]) continue; for (int k = 0; k < 3; k++) sub[k] += d[5 + k]; }
                printf("  wave %d (block %d): %7.0f | %7.0f | %7.0f | %7.0f   total %7.0f per step, %d steps, %zu waves", w, w / 2, acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps,
                       (acc[0] + acc[1] + acc[2] + acc[3]) / steps, (int)(steps / cnt), cnt);
                if (sub[1] > 0) printf("   compute = %5.0f before + %6.0f row function + %5.0f behind", sub[0] / steps, sub[1] / steps, sub[2] / steps);
                printf("\n");
            }
            CK(hipFree(dstp));
        }
#endif
        return nbad ? 2 : 0;
    }
#ifdef MI_STRIP_STAMPS
    {
        int strips = (W + 63) / 64, rows = strips_band_rows(a2, strips), bands = (H + rows - 1) / rows;
        size_t nw = ((size_t)B * strips * bands + 3) / 4 * 4;
        unsigned long long* dstp; CK(hipMalloc(&dstp, nw * 64)); CK(hipMemset(dstp, 0, nw * 64));
        g_strip_stamps = dstp;
        launch_strip(a2, s); CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(nw * 8);
        CK(hipMemcpy(h.data(), dstp, nw * 64, hipMemcpyDeviceToHost));
        double acc[4] = {0}, tot = 0;
        for (size_t i = 0; i < nw; i++) for (int k = 0; k < 4; k++) acc[k] += h[i * 8 + k];
        for (int k = 0; k < 4; k++) tot += acc[k];
        const char* nm[4] = {"wait-dma", "fix+issue", "dw+mfma", "epilogue"};
        for (int k = 0; k < 4; k++) printf("  %-10s %5.1f%%  %8.0f cycles/row\n", nm[k], 100 * acc[k] / tot, acc[k] / ((double)B * strips * bands) / (rows + 2));
        printf("  total %.0f cycles/row/wave (memtime ticks = 100 MHz? see ratio), band rows %d\n", tot / ((double)B * strips * bands) / (rows + 2), rows);
    }
#endif
    return bad ? 2 : 0;
}
