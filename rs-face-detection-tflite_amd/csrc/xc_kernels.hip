// xc_kernels.hip — runs of BlazeBlocks on tiny frames whose channel count swings between a narrow and a wide tensor, as ONE launch:
//
//   wide   = act( W . (DW3x3(narrow) + b_dw) + b + skip )      expand   (full_range: 6x6x96 -> 384; skip = the wide tensor of the pair before
//                                                                        [the double block: wide -> narrow -> wide, skip around both], the
//                                                                        2x2 max-pool of the previous resolution's tensor, or the block's input)
//   narrow = act( W . (DW3x3(wide) + b_dw) + b )                contract (6x6x384 -> 96)
//
// (face_detection_full_range.tflite's three 96 -> 384 -> 96 pairs at 6x6: DEPTHWISE_CONV_2D, CONV_2D + RELU, [MAX_POOL_2D, PAD,] ADD behind
// `interpreter.invoke()`, /root/reference/src/face_detection_lite/face_detection.rs:235)  As six block-kernel
// launches these are 25 - 35 us each at 128 frames for ~1 MFLOP per frame: prologue, one step, epilogue — latency.  The stage programs
// (resident_kernels.hip) cannot take them: a depthwise stage there writes its result to an LDS scratch first, and scratch + a bordered
// 384-channel tensor + the double-buffered constants exceed a CU's LDS.
//
// Here one 512-thread workgroup owns a frame: the narrow and the wide tensor are zero-bordered LDS tensors, every stage is the block
// kernel's inner loop — depthwise 3x3 on the VALU in the MFMA operand layout (lane = pixel x k-half), computed on the fly per wave,
// v_mfma_f32_32x32x2_f32 over the output tiles, the pointwise weights streamed from L2 through a register ring — and writes its
// result into the other tensor.  A stage's 32-pixel groups x output tiles are spread over the 8 waves (4 waves per group).
// Exact f32; same operations as the block kernel, the depthwise bias added first (as in dblock_kernels.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "kernels.hpp"
#include "launch.hpp"

namespace mi {

typedef float xf32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ float4 xld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 xmax4(float4 a, float4 b, float4 c, float4 d) {
    return make_float4(fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y)),
                       fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)), fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w)));
}

struct XcGeom {
    int PSn, PSw;          // pixel strides (floats) of the narrow / wide LDS tensor
    int off_w, off_c;      // LDS offsets (floats): wide tensor, constants
    int cst_floats;
    int lds_bytes;
    unsigned long long* stamps;  // diagnostic builds only (MI_XC_STAMPS): s_memtime stamps of workgroup 0, wave 0
};
#ifdef MI_XC_STAMPS
#define XC_STAMP(k) if (g.stamps && blockIdx.x == 0 && threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); g.stamps[(k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define XC_STAMP(k)
#endif

constexpr int kXcRing = 2;   // k-chunks of A fragments in flight per tile (registers: 4 x MTG per chunk, beside 16 x MTG accumulators)

// One unit: 32 output pixels (group g) x MTG output tiles from mt0 on, of one stage.
// KSPLIT: the `nsh` waves of the group share the CONTRACTION instead of the tiles (a contract stage has 3 tiles and 48 k-chunks: as
// tile shares its waves would each walk all 48 chunks and recompute the depthwise stage; as k shares each computes its own quarter of
// it once): wave `share` takes chunks [share * nch / nsh, ...), all MTG tiles; the partial sums meet in `scratch` (LDS, tile by tile, added
// in a fixed order) and share 0 finishes.  Every wave of the workgroup must come here (barriers inside).
template <int MTG, bool KSPLIT>
__device__ __forceinline__ void xc_unit(const XcStage& st, const XcArgs& a, int frame, const float* src, int PSs, float* dst, int PSd, const float* cst, int g, int mt0,
                                        float* gout, int share = 0, int nsh = 1, float* scratch = nullptr, float* zero_base = nullptr, int zero_f4 = 0) {
    const int lane = threadIdx.x & 63, pl = lane & 31, h = lane >> 5;
    const int W = a.W, row = W + 2, npix = a.H * W;
    const int Cp = (st.C + 7) & ~7, Ch = Cp >> 1, nch = Ch >> 2, MT = (st.Co + 31) >> 5, Cop = MT * 32;
    const float* wdw = cst;             // [9][Cp]
    const float* bdw = cst + 9 * Cp;    // [Cp]
    const float* biasL = cst + 10 * Cp; // [Cop]
    const int q = g * 32 + pl;
    const bool valid = q < npix;
    const int qq = valid ? q : 0, oy = qq / W, ox = qq - oy * W;
    const float* t0 = src + (oy * row + ox) * PSs + h * Ch;   // tap (0, 0) of this pixel (bordered tensor: pixel (y, x) sits at (y + 1, x + 1)), this lane's k-half
    xf32x16 D[MTG];
#pragma unroll
    for (int m = 0; m < MTG; m++)
#pragma unroll
        for (int e = 0; e < 16; e++) D[m][e] = 0.f;
    // A fragments: [mt][j][lane][4]; a ring kXcRing chunks ahead per tile (a chunk is 4 MTG dependent MFMAs behind an L2 round trip)
    const float* wa[MTG];
    float4 ring[kXcRing][MTG];
#pragma unroll
    for (int m = 0; m < MTG; m++) {
        wa[m] = st.w_pw + ((long)min(mt0 + m, MT - 1) * nch * 64 + lane) * 4;
#pragma unroll
        for (int r = 0; r < kXcRing; r++) ring[r][m] = xld4(wa[m] + (long)min((KSPLIT ? share * (nch / nsh) : 0) + r, nch - 1) * 256);
    }
    const int jb = KSPLIT ? share * (nch / nsh) : 0, je = KSPLIT ? jb + nch / nsh : nch;
    // one k-chunk: depthwise 3x3 (+ bias) of 4 channels of this lane's k-half (or the pixel itself: pointwise stage), then the MFMAs
    auto chunk = [&](int j, const float4 (&av)[MTG]) {
        const int c0 = 4 * j;
        float4 bf;
        if (st.has_dw) {   // wave-uniform
            bf = xld4(bdw + h * Ch + c0);
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const float4 w = xld4(wdw + (ky * 3 + kx) * Cp + h * Ch + c0);
                    const float4 d = xld4(t0 + (ky * row + kx) * PSs + c0);
                    bf.x = fmaf(d.x, w.x, bf.x);
                    bf.y = fmaf(d.y, w.y, bf.y);
                    bf.z = fmaf(d.z, w.z, bf.z);
                    bf.w = fmaf(d.w, w.w, bf.w);
                }
        } else {
            bf = xld4(t0 + (row + 1) * PSs + c0);   // pointwise stage: the pixel itself
        }
#pragma unroll
        for (int m = 0; m < MTG; m++) {
            D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].x, bf.x, D[m], 0, 0, 0);
            D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].y, bf.y, D[m], 0, 0, 0);
            D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].z, bf.z, D[m], 0, 0, 0);
            D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].w, bf.w, D[m], 0, 0, 0);
        }
    };
    // whole ring rounds: no control flow around the loads (a branch per chunk makes every wait a full one — each chunk then paid an L2
    // round trip for the fragments asked for a moment before: 3.9 k cycles per chunk); slot r is consumed, then refilled kXcRing chunks ahead
    const int nfull = (je - jb) / kXcRing;
    for (int it = 0; it < nfull; it++) {
        const int j0 = jb + it * kXcRing;
#pragma unroll
        for (int r = 0; r < kXcRing; r++) {
            float4 av[MTG];
#pragma unroll
            for (int m = 0; m < MTG; m++) {
                av[m] = ring[r][m];
                ring[r][m] = xld4(wa[m] + (long)min(j0 + r + kXcRing, nch - 1) * 256);
            }
            chunk(j0 + r, av);
        }
    }
    for (int j = jb + nfull * kXcRing; j < je; j++) {   // the rest (k-chunk counts that are no multiple of the ring): plain loads
        float4 av[MTG];
#pragma unroll
        for (int m = 0; m < MTG; m++) av[m] = xld4(wa[m] + (long)j * 256);
        chunk(j, av);
    }
    if constexpr (KSPLIT) {
        // ---- the shares' partial sums, tile by tile: shares 1 .. nsh-1 write, share 0 adds them in order
        float* sc = scratch + (long)g * (nsh - 1) * 16 * 64;
#pragma unroll
        for (int m = 0; m < MTG; m++) {
            if (share) {
#pragma unroll
                for (int e = 0; e < 16; e++) sc[((share - 1) * 16 + e) * 64 + lane] = D[m][e];
            }
            __syncthreads();
            if (!share) {
                for (int k = 1; k < nsh; k++)
#pragma unroll
                    for (int e = 0; e < 16; e++) D[m][e] += sc[((k - 1) * 16 + e) * 64 + lane];
            }
            __syncthreads();
        }
        // the scratch lay over the destination tensor: clear it again (borders and pads are zeros) before share 0 writes the result
        for (int i = threadIdx.x; i < zero_f4; i += 512) reinterpret_cast<float4*>(zero_base)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        if (share) return;
    }
    // ---- epilogue: lane holds pixel q, output channels mt*32 + 8*gq + 4*h .. +3 in D[m][4*gq .. 4*gq+3]
    if (!valid) return;
    const float hi = st.act == ACT_RELU6 ? 6.f : INFINITY;
    const float lo_slope = st.act == ACT_NONE ? 1.f : 0.f;   // ReLU / ReLU6: 0; no activation: 1
    const float* own = src + ((oy + 1) * row + ox + 1) * PSs;   // the stage's own input pixel (skip == 1)
    float* dl = dst ? dst + ((oy + 1) * row + ox + 1) * PSd : nullptr;
    float* dg = gout ? gout + (long)q * st.Co : nullptr;
#pragma unroll
    for (int m = 0; m < MTG; m++) {
        if (mt0 + m >= MT) continue;
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            const int ch = (mt0 + m) * 32 + 8 * gq + 4 * h;
            if (ch >= st.Co) continue;
            const float4 bb = xld4(biasL + ch);
            float4 v = make_float4(D[m][4 * gq] + bb.x, D[m][4 * gq + 1] + bb.y, D[m][4 * gq + 2] + bb.z, D[m][4 * gq + 3] + bb.w);
            float4 sk = make_float4(0.f, 0.f, 0.f, 0.f);
            if (st.skip == 2 && ch < st.res_C) {   // 2x2 max-pool of the previous resolution's tensor (the run's first stage only: read here, once)
                const float* r = st.res + (long)frame * st.res_fs + ((long)(2 * oy) * st.res_W + 2 * ox) * st.res_C + ch;
                sk = xmax4(xld4(r), xld4(r + st.res_C), xld4(r + (long)st.res_W * st.res_C), xld4(r + (long)(st.res_W + 1) * st.res_C));
            }
            if (st.skip == 1 && ch < st.C) sk = xld4(own + ch);   // channels >= C: the zero channel-pad of the skip
            if (st.skip == 3) sk = xld4(dl + ch);                 // the wide tensor of the pair before: this pixel's old content, replaced in place
            v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w;
            v.x = fminf(fmaxf(v.x, 0.f) + lo_slope * fminf(v.x, 0.f), hi);
            v.y = fminf(fmaxf(v.y, 0.f) + lo_slope * fminf(v.y, 0.f), hi);
            v.z = fminf(fmaxf(v.z, 0.f) + lo_slope * fminf(v.z, 0.f), hi);
            v.w = fminf(fmaxf(v.w, 0.f) + lo_slope * fminf(v.w, 0.f), hi);
            if (dl) *reinterpret_cast<float4*>(dl + ch) = v;
            if (dg) *reinterpret_cast<float4*>(dg + ch) = v;
        }
    }
    (void)Cop;
}

__global__ __launch_bounds__(512) void xc_kernel(XcArgs a, XcGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    XC_STAMP(0)
    float* nar = lds;               // [(H + 2)][(W + 2)][PSn]
    float* wide = lds + g.off_w;    // [(H + 2)][(W + 2)][PSw]
    float* cst = lds + g.off_c;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int frame = blockIdx.x;
    const int W = a.W, row = W + 2, npx = (a.H + 2) * row, npix = a.H * W;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- clear both tensors (borders and channel pads stay zero), bring the frame in
    for (int i = tid; i < (npx * g.PSn) >> 2; i += 512) reinterpret_cast<float4*>(nar)[i] = z4;
    for (int i = tid; i < (npx * g.PSw) >> 2; i += 512) reinterpret_cast<float4*>(wide)[i] = z4;
    __syncthreads();
    {
        const int C0 = a.st[0].C, C4 = C0 >> 2;
        const float* in = a.in + (long)frame * a.in_fs;
        for (int i = tid; i < npix * C4; i += 512) {
            const int px = i / C4, c4 = i - px * C4, y = px / W, x = px - y * W;
            *reinterpret_cast<float4*>(nar + ((y + 1) * row + x + 1) * g.PSn + 4 * c4) = xld4(in + 4 * (long)i);
        }
    }
    XC_STAMP(1)
    const int groups = (npix + 31) >> 5;            // 32-pixel groups of a stage (<= 2: the planner)
    const int wpg = 8 / groups;                     // waves per group
    const int grp = wave / wpg, share = wave - grp * wpg;
    for (int s = 0; s < a.nstages; s++) {
        const XcStage& st = a.st[s];
        const bool expand = (s & 1) == 0;
        const float* src = expand ? nar : wide;
        float* dst = expand ? wide : nar;
        const int PSs = expand ? g.PSn : g.PSw, PSd = expand ? g.PSw : g.PSn;
        const int Cp = (st.C + 7) & ~7, MT = (st.Co + 31) >> 5, Cop = MT * 32;
        // ---- this stage's small constants (depthwise taps [9][Cp], depthwise bias [Cp], pointwise bias [Cop]: one packed blob, copied as it is)
        {
            const int n4 = (10 * Cp + Cop) >> 2;
            for (int i = tid; i < n4; i += 512) reinterpret_cast<float4*>(cst)[i] = xld4(st.cblob + 4 * (long)i);
        }
        XC_STAMP(2 + 4 * s)
        __syncthreads();
        XC_STAMP(3 + 4 * s)
        const bool last = s + 1 == a.nstages;
        float* gout = last ? a.out + (long)frame * a.out_fs : nullptr;
        float* dl = last ? nullptr : dst;
        const int nch = (Cp >> 1) >> 2;
        // contract stages (odd; their destination, the narrow tensor, is dead until the stage's result is written: it serves as the scratch of
        // the split contraction): the group's waves share the k-chunks when the tiles are few and the chunks divide evenly
        const bool ksplit = !expand && groups == 2 && MT <= 3 && nch % wpg == 0 && (wpg - 1) * groups * 16 * 64 <= npx * g.PSn;
        if (ksplit) {
            switch (MT) {
                case 1: xc_unit<1, true>(st, a, frame, src, PSs, dl, PSd, cst, grp, 0, gout, share, wpg, nar, nar, (npx * g.PSn) >> 2); break;
                case 2: xc_unit<2, true>(st, a, frame, src, PSs, dl, PSd, cst, grp, 0, gout, share, wpg, nar, nar, (npx * g.PSn) >> 2); break;
                default: xc_unit<3, true>(st, a, frame, src, PSs, dl, PSd, cst, grp, 0, gout, share, wpg, nar, nar, (npx * g.PSn) >> 2); break;
            }
        } else {
            // the group's output tiles over its waves: per = ceil(MT / wpg) tiles each (<= 3)
            const int per = (MT + wpg - 1) / wpg, mt0 = share * per;
            if (grp < groups && mt0 < MT) {
                switch (per) {
                    case 1: xc_unit<1, false>(st, a, frame, src, PSs, dl, PSd, cst, grp, mt0, gout); break;
                    case 2: xc_unit<2, false>(st, a, frame, src, PSs, dl, PSd, cst, grp, mt0, gout); break;
                    default: xc_unit<3, false>(st, a, frame, src, PSs, dl, PSd, cst, grp, mt0, gout); break;
                }
            }
        }
        XC_STAMP(4 + 4 * s)
        __syncthreads();
        XC_STAMP(5 + 4 * s)
    }
}

bool make_xc_geom(const XcArgs& a, XcGeom* out) {
    if (a.nstages < 2 || a.nstages > kMaxXc || (a.nstages & 1) || a.B < 1 || a.H < 1 || a.W < 1) return false;
    const int npix = a.H * a.W, groups = (npix + 31) / 32;
    if (groups > 2) return false;
    const int wpg = 8 / groups;
    int Cn = 0, Cw = 0, cmax = 0;
    for (int s = 0; s < a.nstages; s++) {
        const XcStage& st = a.st[s];
        if (st.C % 8 || st.Co % 4 || st.C < 8 || st.Co < 4) return false;
        const int MT = (st.Co + 31) / 32;
        if ((MT + wpg - 1) / wpg > 3) return false;
        if (s + 1 < a.nstages && a.st[s + 1].C != st.Co) return false;
        if (s & 1) { if (st.skip != 0) return false; } else if (st.skip == 1 && st.Co < st.C) return false;
        if (st.skip == 3 && (s < 2 || st.Co != a.st[s - 2].Co)) return false;
        if (st.skip < 0 || st.skip > 3) return false;
        if (st.skip == 2 && (st.res_C % 4 || !st.res)) return false;
        if (st.act == ACT_PRELU) return false;
        int& side = (s & 1) ? Cw : Cn;           // the stage's INPUT tensor: narrow for expand stages, wide for contract stages
        if (side && side != st.C) return false;
        side = st.C;
        cmax = std::max(cmax, 10 * ((st.C + 7) & ~7) + MT * 32);
        auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        if (!aligned16(st.w_pw) || !st.cblob || !aligned16(st.cblob) || (st.skip == 2 && (!aligned16(st.res) || (st.res_fs & 3)))) return false;
    }
    if (a.st[a.nstages - 1].Co != Cn) return false;   // the run ends on the narrow side
    XcGeom g{};
    g.PSn = Cn + 4; g.PSw = Cw + 4;
    const int npx = (a.H + 2) * (a.W + 2);
    int off = npx * g.PSn;
    off = (off + 3) & ~3;
    g.off_w = off; off += npx * g.PSw; off = (off + 3) & ~3;
    g.off_c = off; off += (cmax + 3) & ~3;
    g.cst_floats = cmax;
    g.lds_bytes = off * 4;
    if (g.lds_bytes > 160 * 1024 - 256) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!aligned16(a.in) || !aligned16(a.out) || (a.in_fs & 3) || (a.out_fs & 3)) return false;
    *out = g;
    return true;
}

}  // namespace

int xc_const_floats(int C, int Co) { return 10 * ((C + 7) & ~7) + (Co + 31) / 32 * 32; }

bool xc_kernel_supports(const XcArgs& a) {
    XcGeom g;
    return make_xc_geom(a, &g);
}

int launch_xc(const XcArgs& a, void* stream) {
    XcGeom g;
    if (!make_xc_geom(a, &g)) return (int)hipErrorInvalidValue;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(xc_kernel)); e != hipSuccess) return (int)e;
    g.stamps = nullptr;
#ifdef MI_XC_STAMPS
    if (getenv("MI_XC_STAMPS")) {   // stamps build: run the launch eagerly and print where workgroup 0's first wave spent it
        static unsigned long long* d = nullptr;
        if (!d) (void)hipMalloc(&d, 64 * 8);
        g.stamps = d;
        hipLaunchKernelGGL(xc_kernel, dim3((unsigned)a.B), dim3(512), (size_t)g.lds_bytes, (hipStream_t)stream, a, g);
        (void)hipDeviceSynchronize();
        unsigned long long h[64];
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "xc stamps (wg 0 wave 0): clear %llu, load %llu;", h[1] - h[0], h[2] - h[1]);
        for (int s = 0; s < a.nstages; s++) fprintf(stderr, " | s%d constants %llu barrier %llu units %llu barrier %llu", s, s ? h[2 + 4 * s] - h[1 + 4 * s] : 0ULL, h[3 + 4 * s] - h[2 + 4 * s], h[4 + 4 * s] - h[3 + 4 * s], h[5 + 4 * s] - h[4 + 4 * s]);
        fprintf(stderr, "\n");
        return 0;
    }
#endif
    return (int)launch_kernel(xc_kernel, dim3((unsigned)a.B), dim3(512), (size_t)g.lds_bytes, (hipStream_t)stream, a, g);
}

}  // namespace mi
