// Development harness for the frame-resident stage-program kernel: a synthetic iris-like tail (N bottlenecks at HxH:
// pointwise C -> C/2, then DW3x3 -> pointwise C/2 -> C + skip, in place) timed as one launch; with -DMI_RES_STAMPS prints
// where a workgroup's time goes per stage.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off [-DMI_RES_STAMPS] tools/res_bench.hip -o /tmp/rb
#include "../rs-face-detection-tflite_amd/csrc/resident_kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
using namespace mi;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 1024, H = argc > 2 ? atoi(argv[2]) : 8, C = argc > 3 ? atoi(argv[3]) : 128, NB = argc > 4 ? atoi(argv[4]) : 6;
    if (NB < 1 || H < 1 || C % 8 || B < 1) { printf("bad arguments\n"); return 1; }
    const int NB_arg = NB; (void)NB_arg;
    const bool banded = argc > 5 && std::string(argv[5]) == "band";   // row-band bottleneck (iris 32x32x64): one pointwise + one depthwise stage
    const int R = C / 2;
    size_t fs = (size_t)H * H * C;
    float *din, *dout, *dw;
    CK(hipMalloc(&din, B * fs * 4)); CK(hipMalloc(&dout, B * fs * 4));
    std::vector<float> hin(B * fs);
    for (size_t i = 0; i < hin.size(); i++) hin[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
    // weights blob: per bottleneck [pw1 R x C][b1 R][a1 R][dw 9R][bdw R][pw2 C x R][b2 C][a2 C]
    std::vector<float> wb;
    auto put = [&](size_t n, float v) { size_t o = (wb.size() + 63) & ~63ul; wb.resize(o + n, v); return (long)o; };
    std::vector<ResStage> prog;
    const int PSX = C + 4, PSR = R + 4;
    const int offX = 0, offR = H * H * PSX;                 // X unbordered, R bordered
    const int sizeR = (H + 2) * (H + 2) * PSR;
    int bands = 1;
    if (banded) {
        const int BR = 8;
        bands = (H + BR - 1) / BR;
        ResStage a;
        a.kind = RES_STAGE_GATHER; a.src_H = H; a.src_W = H; a.src_C = C; a.KH = a.KW = 1; a.S = 1; a.Kv = C;
        a.Ho = BR + 2; a.Wo = H; a.Co = R; a.act = ACT_PRELU; a.band_role = 1; a.band_rows = BR; a.band_H = H;
        a.src_g.base = 1; a.src_g.fs = (long)fs;
        a.kblk = (argc > 7 && atoi(argv[7]) == 0) ? 0 : 16;   // argv[7] = 0: direct gather (mode 1) instead of the LDS-staged K blocks
        a.dst_off = 0; a.dst_PS = R + 4; a.dst_b = 1; a.zero_dst = (BR + 2) * (H + 2) * (R + 4);
        a.w_pw = put((size_t)((R + 31) / 32 * 32) * C, 0.01f); a.cblob = put(resident_const_floats(a), 0.01f);
        prog.push_back(a);
        ResStage b;
        b.kind = RES_STAGE_DW; b.src_off = 0; b.src_H = BR; b.src_W = H; b.src_C = R; b.src_PS = R + 4; b.src_b = 1;
        b.KH = b.KW = 3; b.S = 1; b.pt = b.pl = 1; b.Kv = R; b.Ho = BR; b.Wo = H; b.Co = C; b.act = ACT_PRELU;
        b.res_mode = RES_DIRECT; b.res_C = C; b.res_H = H; b.res_W = H; b.band_role = 2; b.band_rows = BR; b.band_H = H;
        b.res_g.base = 1; b.res_g.fs = (long)fs; b.dst_g.base = 2; b.dst_g.fs = (long)fs;
        b.w_pw = put((size_t)((C + 31) / 32 * 32) * R, 0.01f); b.cblob = put(resident_const_floats(b), 0.01f);
        b.dw_pg = std::max(1, std::min((BR * H + 31) / 32, (8 + (C + 31) / 32 - 1) / ((C + 31) / 32)));
        prog.push_back(b);
        NB = 0;
    } else
    {   // X <- input (pointwise identity would need weights; use a LOAD with border 0)
        ResStage st; st.kind = RES_STAGE_LOAD; st.src_H = H; st.src_W = H; st.src_C = C;
        st.src_g.base = 1; st.src_g.fs = (long)fs;
        st.dst_off = offX; st.dst_PS = PSX; st.dst_b = 0; st.zero_dst = H * H * PSX;
        prog.push_back(st);
    }
    for (int k = 0; k < NB; k++) {
        ResStage a;  // R = prelu(W1 X)
        a.kind = RES_STAGE_GATHER; a.src_off = offX; a.src_H = H; a.src_W = H; a.src_C = C; a.src_PS = PSX; a.src_b = 0;
        a.KH = a.KW = 1; a.S = 1; a.Kv = C; a.Ho = H; a.Wo = H; a.Co = R;
        a.dst_off = offR; a.dst_PS = PSR; a.dst_b = 1; a.zero_dst = k == 0 ? sizeR : 0;
        a.act = ACT_PRELU; a.w_pw = put((size_t)((R + 31) / 32 * 32) * C, 0.01f); a.cblob = put(resident_const_floats(a), 0.01f);
        prog.push_back(a);
        ResStage b;  // X = prelu(W2 (DW R) + X)
        b.kind = RES_STAGE_DW; b.src_off = offR; b.src_H = H; b.src_W = H; b.src_C = R; b.src_PS = PSR; b.src_b = 1;
        b.KH = b.KW = 3; b.S = 1; b.pt = b.pl = 1; b.Kv = R; b.Ho = H; b.Wo = H; b.Co = C;
        b.dst_off = offX; b.dst_PS = PSX; b.dst_b = 0;
        b.res_mode = RES_DIRECT; b.res_C = C; b.res_H = H; b.res_W = H; b.res_off = offX; b.res_PS = PSX; b.res_b = 0;
        b.act = ACT_PRELU;
        b.w_pw = put((size_t)((C + 31) / 32 * 32) * R, 0.01f); b.cblob = put(resident_const_floats(b), 0.01f);
        b.dw_pg = std::max(1, std::min((H * H + 31) / 32, (8 + (C + 31) / 32 - 1) / ((C + 31) / 32)));
        if (k == NB - 1) { b.dst_g.base = 2; b.dst_g.fs = (long)fs; }
        prog.push_back(b);
    }
    int cmax = 0;
    for (auto& st : prog) cmax = std::max(cmax, resident_const_floats(st));
    ResLaunch L;
    L.bands = bands;
    L.const_off = banded ? ((prog[0].zero_dst + 3) & ~3) : (offR + sizeR + 3) & ~3;
    L.const_floats = cmax;
    int scratch = 0;
    for (auto& st : prog)
        if (st.kind == RES_STAGE_DW) { st.dw_off = L.const_off + 2 * cmax; scratch = std::max(scratch, st.dw_pg * 32 * (((st.Kv + 7) & ~7) + 4)); }
        else if (st.kblk) { st.dw_off = L.const_off + 2 * cmax; scratch = std::max(scratch, kResSlabFloats); }
    L.lds_bytes = (L.const_off + 2 * cmax + scratch) * 4;
    wb.resize(wb.size() + 4096, 0.f);  // slack for the A-fragment prefetch
    CK(hipMalloc(&dw, wb.size() * 4)); CK(hipMemcpy(dw, wb.data(), wb.size() * 4, hipMemcpyHostToDevice));
    ResStage* dprog; CK(hipMalloc(&dprog, prog.size() * sizeof(ResStage))); CK(hipMemcpy(dprog, prog.data(), prog.size() * sizeof(ResStage), hipMemcpyHostToDevice));
    L.prog = dprog; L.nstages = (int)prog.size(); L.B = B;
    for (int k = 0; k < kResBases; k++) { L.bases.p[k] = nullptr; L.bases.scale[k] = 0; L.bases.frame0[k] = 0; }
    L.bases.p[1] = din; L.bases.p[2] = dout; L.bases.weights = dw;
    printf("B %d  %dx%dx%d  %d bottlenecks = %d stages, LDS %d B, weights %.1f KB\n", B, H, H, C, NB, L.nstages, L.lds_bytes, wb.size() * 4 / 1024.0);
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int i = 0; i < 3; i++) if (launch_resident(L, s)) { printf("launch failed\n"); return 1; }
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int n = 20;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < n; i++) launch_resident(L, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
    double macs = (double)B * H * H * NB * ((double)C * R + 9.0 * R + (double)R * C);
    printf("%.3f ms per launch = %.2f us per stage per round-of-512  %.2f TFLOP/s\n", ms, ms * 1e3 / L.nstages / std::max(1.0, B / 512.0), 2 * macs / ms / 1e9);
#ifdef MI_RES_STAMPS
    {
        const int WG = B * L.bands;
        size_t n8 = (size_t)WG * L.nstages * 8;
        unsigned long long* dst; CK(hipMalloc(&dst, n8 * 8)); CK(hipMemset(dst, 0, n8 * 8));
        g_res_stamps = dst;
        hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
        CK(hipEventRecord(f0, s));
        launch_resident(L, s);
        CK(hipEventRecord(f1, s)); CK(hipStreamSynchronize(s));
        float kms; CK(hipEventElapsedTime(&kms, f0, f1));
        std::vector<unsigned long long> h(n8);
        CK(hipMemcpy(h.data(), dst, n8 * 8, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int b = 0; b < WG; b++) { tmin = std::min(tmin, h[(size_t)b * L.nstages * 8]); tmax = std::max(tmax, h[((size_t)b * L.nstages + L.nstages - 1) * 8 + 3]); }
        size_t zeros[8] = {0};
        for (size_t i = 0; i < n8; i++) zeros[i & 7] += h[i] == 0;
        printf("zero stamps per slot: %zu %zu %zu %zu %zu %zu %zu %zu of %zu\n", zeros[0], zeros[1], zeros[2], zeros[3], zeros[4], zeros[5], zeros[6], zeros[7], n8 / 8);
        // the counters of different XCDs are not aligned: calibrate on the mean workgroup lifetime, which equals the launch's
        // event time when every workgroup is resident at once (B <= 256)
        double life = 0;
        for (int b = 0; b < WG; b++) life += h[((size_t)b * L.nstages + L.nstages - 1) * 8 + 3] - h[(size_t)b * L.nstages * 8];
        const double tick_us = WG <= 256 ? kms * 1e3 / (life / WG) : (argc > 6 ? atof(argv[6]) : 0.00059);
        (void)tmin; (void)tmax;
        printf("stamped launch %.3f ms, mean workgroup lifetime %.0f ticks -> %.5f us per tick\n", kms, life / WG, tick_us);
        printf("stage kind: prefetch-issue | work | closing barrier || first unit of wave 0: ring issue | mfma loop | epilogue   [us]\n");
        for (int st = 0; st < L.nstages; st++) {
            double a[6] = {0, 0, 0, 0, 0, 0};
            for (int b = 0; b < WG; b++) {
                const unsigned long long* q = &h[((size_t)b * L.nstages + st) * 8];
                a[0] += q[1] - q[0]; a[1] += q[2] - q[1]; a[2] += q[3] - q[2];
                if (q[7] > q[4]) { a[3] += q[5] - q[4]; a[4] += q[6] - q[5]; a[5] += q[7] - q[6]; }
            }
            printf("  %2d: %d   %6.2f  %6.2f  %6.2f  ||  %6.2f  %6.2f  %6.2f\n", st, prog[st].kind, a[0] / WG * tick_us, a[1] / WG * tick_us, a[2] / WG * tick_us,
                   a[3] / WG * tick_us, a[4] / WG * tick_us, a[5] / WG * tick_us);
        }
        double tot = 0;
        for (int b = 0; b < WG; b++) tot += h[((size_t)b * L.nstages + L.nstages - 1) * 8 + 3] - h[(size_t)b * L.nstages * 8];
        printf("  workgroup lifetime %.1f us\n", tot / WG * tick_us);
    }
#endif
    return 0;
}
