// Development harness for mdblock_kernel (mdblock_kernels.hip): checks it against the LDS-tensor double-block kernel
// (dblock_kernels.hip) on a full_range double-block shape and times both.
// Build: tools/build_mdb_bench.sh [TAG -DFLAG ...]   Run: tools/bin/bb_mdb [B H W C Cm Co]
#include "../rs-face-detection-tflite_amd/csrc/dblock_kernels.hip"
#include "../rs-face-detection-tflite_amd/csrc/mdblock_kernels.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mi;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.f - 1.f; }

// the block kernel's A-fragment packing (engine.cpp pack_pw): W [O][I] -> [mt][j][lane][4]
static std::vector<float> pack_pw(const std::vector<float>& w, int O, int I) {
    const int Ip = (I + 7) & ~7, MT = (O + 31) / 32, Ch = Ip / 2;
    std::vector<float> r((size_t)MT * 32 * Ip, 0.f);
    for (int mt = 0; mt < MT; mt++) for (int j = 0; j < Ch / 4; j++) for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++) {
        const int o = mt * 32 + (l & 31), c = (l >> 5) * Ch + 4 * j + e;
        if (o < O && c < I) r[(((size_t)mt * (Ch / 4) + j) * 64 + l) * 4 + e] = w[(size_t)o * I + c];
    }
    return r;
}

int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 96, W = argc > 3 ? atoi(argv[3]) : 96;
    int C = argc > 4 ? atoi(argv[4]) : 32, Cm = argc > 5 ? atoi(argv[5]) : 8, Co = argc > 6 ? atoi(argv[6]) : 32;
    int act = argc > 7 ? atoi(argv[7]) : ACT_RELU;
    const bool pair = argc > 8 ? atoi(argv[8]) != 0 : false;  // two plain BlazeBlocks (each adds its own input) instead of the double block
    const size_t nin = (size_t)B * H * W * C, nout = (size_t)B * H * W * Co;
    float *din, *dout, *dref;
    CK(hipMalloc(&din, nin * 4)); CK(hipMalloc(&dout, nout * 4)); CK(hipMalloc(&dref, nout * 4));
    unsigned seed = 777;
    std::vector<float> hin(nin);
    for (auto& v : hin) v = rnd(seed);
    CK(hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice));
    auto up = [&](const std::vector<float>& h) { float* d; CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); return d; };
    std::vector<float> wdw1(9 * C), bdw1(C), pw1((size_t)Cm * C), b1(Cm), al1(Cm), wdw2(9 * Cm), bdw2(Cm), pw2((size_t)Co * Cm), b2(Co), al2(Co);
    for (auto& v : wdw1) v = 0.3f * rnd(seed);
    for (auto& v : bdw1) v = 0.1f * rnd(seed);
    for (auto& v : pw1) v = 0.2f * rnd(seed);
    for (auto& v : b1) v = 0.1f * rnd(seed);
    for (auto& v : al1) v = 0.25f * rnd(seed);
    for (auto& v : wdw2) v = 0.3f * rnd(seed);
    for (auto& v : bdw2) v = 0.1f * rnd(seed);
    for (auto& v : pw2) v = 0.2f * rnd(seed);
    for (auto& v : b2) v = 0.1f * rnd(seed);
    for (auto& v : al2) v = 0.25f * rnd(seed);
    // dblock constants blob (engine.cpp): [dw1 9 x C][b_dw1 C][b1 32 MTA][slope1 32 MTA][dw2 9 x Cmp][b_dw2 Cmp][b2 32 MT][slope2 32 MT]
    const int Cmp = (Cm + 7) & ~7, MT = (Co + 31) / 32, MTA = (Cm + 31) / 32;
    std::vector<float> cb((size_t)dblock_const_floats(C, Cm, Co), 0.f);
    auto slope = [&](const std::vector<float>& al, int c) { return act == ACT_PRELU ? al[c] : (act == ACT_NONE ? 1.f : 0.f); };
    size_t o = 0;
    for (int i = 0; i < 9 * C; i++) cb[o + i] = wdw1[i];
    o += 9 * C;
    for (int c = 0; c < C; c++) cb[o + c] = bdw1[c];
    o += C;
    for (int c = 0; c < Cm; c++) { cb[o + c] = b1[c]; cb[o + 32 * MTA + c] = slope(al1, c); }
    o += 64 * MTA;
    for (int t = 0; t < 9; t++) for (int c = 0; c < Cm; c++) cb[o + t * Cmp + c] = wdw2[t * Cm + c];
    o += 9 * Cmp;
    for (int c = 0; c < Cm; c++) cb[o + c] = bdw2[c];
    o += Cmp;
    for (int c = 0; c < Co; c++) { cb[o + c] = b2[c]; cb[o + 32 * MT + c] = slope(al2, c); }
    std::vector<float> mc((size_t)std::max(1, mdblock_consts_floats(W, C, Cm, Co, pair)), 0.f);
    mdblock_pack_consts(W, C, Cm, Co, wdw1.data(), bdw1.data(), pw1.data(), b1.data(), al1.data(), act, wdw2.data(), bdw2.data(), pw2.data(), b2.data(), al2.data(), act, mc.data(), pair);
    DblockArgs a;
    a.in = din; a.out = dref; a.in_fs = (long)H * W * C; a.out_fs = (long)H * W * Co;
    a.B = B; a.H = H; a.W = W; a.C = C; a.Cm = Cm; a.Co = Co;
    a.consts = up(cb); a.w1 = up(pack_pw(pw1, Cm, C)); a.w2 = up(pack_pw(pw2, Co, Cm));
    a.hi1 = a.hi2 = act == ACT_RELU6 ? 6.f : INFINITY;
    a.mconsts = up(mc); a.act1 = a.act2 = act;
    if (pair) a.skip1 = a.skip2_from_a = 1;
    DblockArgs a2 = a; a2.out = dout;
    hipStream_t s; CK(hipStreamCreate(&s));
    if (!dblock_kernel_supports(a)) { printf("dblock: unsupported\n"); return 1; }
    if (int rc = launch_dblock(a, s)) { printf("dblock launch failed %d\n", rc); return 1; }
    if (!mdblock_kernel_supports(a2)) { printf("mdblock: unsupported\n"); return 1; }
    CK(hipMemset(dout, 0xff, nout * 4));
    if (int rc = launch_mdblock(a2, s)) { printf("mdblock launch failed %d\n", rc); return 1; }
    CK(hipStreamSynchronize(s));
    std::vector<float> r0(nout), r1(nout);
    CK(hipMemcpy(r0.data(), dref, nout * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), dout, nout * 4, hipMemcpyDeviceToHost));
    double maxd = 0; size_t bad = 0, first = (size_t)-1;
    for (size_t i = 0; i < nout; i++) {
        double d = std::fabs((double)r0[i] - r1[i]);
        if (!(d <= 1e-4 * std::max(1.0, std::fabs((double)r0[i])))) { if (!bad) first = i; bad++; }
        if (d > maxd || d != d) maxd = d;
    }
    printf("check: max |diff| %.3g, %zu of %zu outside 1e-4", maxd, bad, nout);
    if (bad) { size_t px = first / Co; printf("  first at frame %zu y %zu x %zu c %zu: ref %g got %g", px / ((size_t)H * W), px / W % H, px % W, first % Co, r0[first], r1[first]); }
    printf("\n");
    if (bad) {  // where the mismatches are
        size_t by_row[8] = {0}, by_col[8] = {0};
        for (size_t i = 0; i < nout; i++) {
            double d = std::fabs((double)r0[i] - r1[i]);
            if (!(d <= 1e-4 * std::max(1.0, std::fabs((double)r0[i])))) { size_t px = i / Co; by_row[(px / W % H) * 8 / H]++; by_col[(px % W) * 8 / W]++; }
        }
        printf("  mismatches by row octant:"); for (int k = 0; k < 8; k++) printf(" %zu", by_row[k]);
        printf("\n  mismatches by column octant:"); for (int k = 0; k < 8; k++) printf(" %zu", by_col[k]);
        printf("\n");
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; which++) {
        const int it = 20;
        auto run = [&] { which ? launch_mdblock(a2, s) : launch_dblock(a, s); };
        for (int i = 0; i < 3; i++) run();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < it; i++) run();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        printf("%s B %d %dx%d %d -> %d -> %d : %.4f ms  %.1f GB/s algorithmic\n", which ? "mdblock" : "dblock ", B, H, W, C, Cm, Co, ms, (double)(nin + nout) * 4 / ms / 1e6);
    }
    return bad ? 2 : 0;
}
