// bneck_kernels.hip — the iris network's bottleneck blocks with the wide tensor held in REGISTERS.
//
//   r = act1( W1 . x + b1 )                          pointwise C -> Cm            (CONV_2D 1x1 + PRELU)
//   y = act2( W2 . (DW3x3(r) + b_dw) + b2 + x )      depthwise, pointwise Cm -> C (DEPTHWISE_CONV_2D, CONV_2D 1x1, ADD, PRELU)
//
// (iris_landmark.tflite: 64 -> 32 -> 64 at 32x32, 128 -> 64 -> 128 at 16x16; `interpreter.invoke()` at
// /root/reference/src/face_detection_lite/iris_landmark.rs:203)  The stage programs of
// resident_kernels.hip run a pair as pointwise-from-global -> LDS -> depthwise block with the skip read from and the result written
// to global memory in 16-byte pieces (one cache-line lookup per lane and piece): 40 us per workgroup for 5 us of MFMA work.
//
// Here a 512-thread workgroup owns 256 pixels (a whole frame of <= 256 pixels, or a band of 256 / W rows of a larger one) and
// wave w keeps the C channels of its 32 pixels in the MFMA RESULT layout (lane = pixel x half; 4 consecutive channels per register
// quad) for as long as the workgroup lives:
//   * the result layout IS an operand layout: with the contraction index of W1 permuted to match (host packing:
//     channel(chunk j, half h, e) = 32 (j / 4) + 8 (j % 4) + 4 h + e) the registers of x are the B operands of
//     v_mfma_f32_32x32x2_f32 as they stand — the first pointwise conv reads nothing but registers and weights;
//   * r goes to a zero-bordered LDS tensor (the only activation in LDS), the depthwise 3x3 reads its taps from there in the
//     block kernel's operand layout, the second pointwise conv accumulates over all C output channels of the wave's pixels;
//   * + b2 + x (registers) -> activation -> x (registers): a run of blocks on a resident frame never touches memory between blocks.
//   * row bands (frames of more than 256 pixels, one block per launch): the halo rows of r above and below the band are recomputed
//     from x in global memory by the last two waves.
// Exact f32 (MFMA f32 = fmaf chain); matches resident_kernels.hip / block_kernels.hip to reassociation of the contractions.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "kernels.hpp"
#include "launch.hpp"

namespace mi {

typedef float nf32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ float4 nld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct BneckGeom {
    int R;          // rows of the frame a workgroup owns (R * W <= 256)
    int PSa, RSa;   // LDS pixel / row stride of r (floats)
    int PSx;        // pixel stride of the wave-private slabs that x and y pass through ([32 pixels][C + 4] per wave)
    int off_w;      // row bands: the two pointwise matrices of the (single) block, staged once per workgroup (W1 then W2), -1: streamed from L2
    int off_slab;   // where the slabs start: behind the constants (slab_sep) or on top of r's region (frames with 128 channels: no room for both)
    int slab_sep;
    int off_c;      // constants: two halves of cfl floats
    int cfl;
    int lds_bytes;
    unsigned long long* stamps;  // diagnostic builds only (MI_BNECK_STAMPS): 8 s_memtime stamps per wave
};
#ifdef MI_BNECK_STAMPS
unsigned long long* g_bneck_stamps = nullptr;
#define MI_BN_STAMP(k) if (g.stamps && (threadIdx.x & 63) == 0) { __builtin_amdgcn_sched_barrier(0); g.stamps[((long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_BN_STAMP(k)
#endif

// WL: the (single) block's pointwise matrices are staged in LDS (row bands).  A compile-time switch: a run-time choice between an LDS
// and a global pointer makes the loads FLAT, and a flat load waits on vmcnt AND lgkmcnt — i.e. for the next item's x as well.
template <int MT, int MTA, bool WL>
__global__ __launch_bounds__(512, 2) void bneck_kernel(BneckArgs a, BneckGeom g) {
    constexpr int C = 32 * MT, Cm = 32 * MTA, Chm = Cm / 2;
    constexpr int NCH1 = C / 8;   // chunks of the first contraction (4 channels of each half per chunk)
    constexpr int NCH2 = Cm / 8;  // chunks of the second
    constexpr int PSa = Cm + 4, PSx = C + 4;  // = g.PSa, g.PSx: compile-time so that tap / chunk offsets fold into the ds_read immediates
    constexpr int OFF_SL1 = Cm, OFF_WDW = 2 * Cm, OFF_BDW = 11 * Cm, OFF_B2 = 12 * Cm, OFF_SL2 = 12 * Cm + C;
    constexpr int NCONST = 12 * Cm + 2 * C, NCR = (NCONST + 511) / 512;
    constexpr int C4 = C / 4, NLD = 32 * C4 / 64;  // float4s per pixel; lane-linear loads / stores per lane and 32-pixel group
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rt = lds;  // r: [(R + 2)][(W + 2)][PSa], zero border; row 0 = image row r0 - 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int W = a.W;
    const int q = wave * 32 + pl;
    const bool in_tile = q < g.R * W;
    const int oy = in_tile ? q / W : 0, ox = in_tile ? q - (q / W) * W : 0;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float* slab = lds + g.off_slab + wave * (32 * PSx);  // wave-private: x comes in and y goes out through it
    float* my_r = rt + (oy + 1) * g.RSa + (ox + 1) * PSa;
    const int total = a.B * a.bands;

    // workgroup barrier that orders LDS traffic only (raw s_barrier behind lgkmcnt(0)): __syncthreads() also waits for every global
    // load and store in flight — the next item's x, the previous item's stores — which is exactly what must stay in flight
    auto wg_barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    float cr[NCR];
    auto fetch_consts = [&](const BneckBlock& bk) {
#pragma unroll
        for (int k = 0; k < NCR; k++) cr[k] = bk.consts[min(tid + 512 * k, NCONST - 1)];
    };
    auto commit_consts = [&](int half) {
#pragma unroll
        for (int k = 0; k < NCR; k++)
            if (tid + 512 * k < NCONST) lds[g.off_c + half * g.cfl + tid + 512 * k] = cr[k];
    };
    auto sl_off = [&](int ln, int k) {  // float4 number ln + 64 k of a 32-pixel group -> float offset in the slab
        const int i = ln + 64 * k, px = i / C4;
        return px * PSx + 4 * (i - px * C4);
    };
    // valid pixels of this wave's group in work item `it` (a prefix of the group: whole rows inside the image)
    auto nvalid_of = [&](int it) {
        const int band = it % a.bands, r0 = band * g.R;
        return min(32, max(0, min(g.R * W, (a.H - r0) * W) - wave * 32));
    };
    // The wave's 32 pixels are 32 C consecutive floats of the tensor: they travel lane-linear (1 KiB per instruction; in the MFMA
    // result layout every lane would ask for 16-byte pieces 4 C bytes apart, one cache-line lookup each: a third of a band's time)
    // and are turned through the slab.
    constexpr bool PERSIST = WL;  // 64-channel kernels walk over several items with the next x in flight; with 128 channels there are no registers for that
    float4 xv[NLD];
    auto prefetch_x = [&](int it) {
        const int frame = it / a.bands, band = it - frame * a.bands, r0 = band * g.R, nv = nvalid_of(it);
        const float* xg = a.in + (long)frame * a.in_fs + ((long)r0 * W + wave * 32) * C;
#pragma unroll
        for (int k = 0; k < NLD; k++)  // unconditional loads at clamped addresses; pieces past the valid pixels are zeroed where xv is consumed
            xv[k] = nld4(xg + 4 * min(lane + 64 * k, max(nv * C4 - 1, 0)));  // (a select here would make the wave wait for the data at once)
    };

    // r = act1(W1 . x + b1) for one 32-pixel group whose x comes from `xb` (static register indices: the chunk loop is unrolled);
    // weights stream from L2 through a ring 4 chunks deep.  Result (zero where !keep) to the LDS tensor at `dst`, if `store`.
    auto pw1 = [&](const BneckBlock& bk, const float* cst, auto&& xb, float* dst, bool store, bool keep) {
        // weights: from LDS in the walking (row-band) kernels — a global load here would have to wait for the next item's x, which was
        // issued before it (vmcnt retires in order) — from L2 otherwise
        const float* w1p = WL ? lds + g.off_w : bk.w1;
        nf32x16 Da[MTA];
#pragma unroll
        for (int t = 0; t < MTA; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) Da[t][e] = 0.f;
        float4 ring[4][MTA];
        auto a_frag = [&](int j, float4 (&av)[MTA]) {
#pragma unroll
            for (int t = 0; t < MTA; t++) av[t] = nld4(w1p + ((t * NCH1 + j) * 64 + lane) * 4);
        };
#pragma unroll
        for (int j = 0; j < 4; j++) a_frag(j, ring[j]);
#pragma unroll
        for (int j = 0; j < NCH1; j++) {
            const float4 bf = xb(j);
#pragma unroll
            for (int t = 0; t < MTA; t++) {
                const float4 av = ring[j & 3][t];
                Da[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf.x, Da[t], 0, 0, 0);
                Da[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf.y, Da[t], 0, 0, 0);
                Da[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf.z, Da[t], 0, 0, 0);
                Da[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf.w, Da[t], 0, 0, 0);
            }
            if (j + 4 < NCH1) a_frag(j + 4, ring[j & 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < MTA; t++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const int ch = 32 * t + 8 * gq + 4 * h;
                const float4 bb = nld4(cst + ch), sl = nld4(cst + OFF_SL1 + ch);
                float4 v = make_float4(Da[t][4 * gq] + bb.x, Da[t][4 * gq + 1] + bb.y, Da[t][4 * gq + 2] + bb.z, Da[t][4 * gq + 3] + bb.w);
                v.x = fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), bk.hi1);
                v.y = fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), bk.hi1);
                v.z = fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), bk.hi1);
                v.w = fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), bk.hi1);
                if (!keep) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (store) *reinterpret_cast<float4*>(dst + ch) = v;
            }
    };

    MI_BN_STAMP(0)
    int item = blockIdx.x;
    fetch_consts(a.blocks[0]);
    if (PERSIST) prefetch_x(item);
    if (g.slab_sep) {  // r has its own region: cleared once (the borders stay zero for every item of this workgroup)
        for (int i = tid; i < ((g.R + 2) * g.RSa) >> 2; i += 512) reinterpret_cast<float4*>(rt)[i] = zero4;
        commit_consts(0);
    }
    if (WL) {
        for (int i = tid; i < (C * Cm) >> 2; i += 512) {
            reinterpret_cast<float4*>(lds + g.off_w)[i] = nld4(a.blocks[0].w1 + 4 * i);
            reinterpret_cast<float4*>(lds + g.off_w + C * Cm)[i] = nld4(a.blocks[0].w2 + 4 * i);
        }
    }
    for (;;) {  // one item (a frame, or a band of one) per trip; only the walking kernels come round again
        const int frame = item / a.bands, band = item - frame * a.bands, r0 = band * g.R;
        const int nvalid = nvalid_of(item);
        const bool valid = pl < nvalid;
        // ---- x: this lane's pixel, channels 32 m + 8 gq + 4 h + e in X[m][4 gq + e]
        nf32x16 X[MT];
        if (PERSIST) {
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                float4 t = xv[k];
                if (lane + 64 * k >= nvalid * C4) t = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(slab + sl_off(lane, k)) = t;
            }
        } else {  // 8 loads at a time: 16 in flight cost more registers than the kernel has to spare
            const float* xg = a.in + (long)frame * a.in_fs + ((long)r0 * W + wave * 32) * C;
#pragma unroll
            for (int k0 = 0; k0 < NLD; k0 += 8) {
                float4 t[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = lane + 64 * (k0 + k);
                    t[k] = nld4(xg + 4 * min(i, max(nvalid * C4 - 1, 0)));
                    if (i >= nvalid * C4) t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < 8; k++) *reinterpret_cast<float4*>(slab + sl_off(lane, k0 + k)) = t[k];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const float4 t = nld4(slab + pl * PSx + 32 * m + 8 * gq + 4 * h);
                X[m][4 * gq] = t.x; X[m][4 * gq + 1] = t.y; X[m][4 * gq + 2] = t.z; X[m][4 * gq + 3] = t.w;
            }
        // ---- row bands: this wave's piece of the rows above / below the band (the last 2 ceil(W / 32) waves have one each); its x is
        // asked for now ...
        const int npieces = (W + 31) >> 5, hu = wave - (8 - 2 * npieces);
        const bool has_halo = PERSIST && a.bands > 1 && hu >= 0;  // row bands exist for 64 channels only (make_bneck_geom)
        const int hside = has_halo ? hu / npieces : 0, hx = (has_halo ? hu - hside * npieces : 0) * 32 + pl;
        const int hiy = hside == 0 ? r0 - 1 : r0 + g.R;
        const bool hrow = has_halo && hiy >= 0 && hiy < a.H, hv = hx < W;
        float4 xq[PERSIST ? NCH1 : 1];
        if (PERSIST && hrow) {  // wave-uniform
            const float* xh = a.in + (long)frame * a.in_fs + ((long)hiy * W + (hv ? hx : 0)) * C + 4 * h;
#pragma unroll
            for (int j = 0; j < NCH1; j++) xq[PERSIST ? j : 0] = nld4(xh + 32 * (j >> 2) + 8 * (j & 3));
        }
        // ... and, behind it, the next item's x: it has this whole item to arrive.  (vmcnt retires in order: any global load issued
        // after the prefetch would have to wait for it — which is why the walking kernels keep their weights in LDS.)
        if (PERSIST && item + (int)gridDim.x < total) prefetch_x(item + gridDim.x);
        if (!g.slab_sep) {  // the slabs lie over r's region: every wave has its x before r is cleared
            wg_barrier();
            for (int i = tid; i < ((g.R + 2) * g.RSa) >> 2; i += 512) reinterpret_cast<float4*>(rt)[i] = zero4;
            commit_consts(0);
        }
        wg_barrier();
        MI_BN_STAMP(1)

        for (int blk = 0; blk < a.nblocks; blk++) {
            const BneckBlock& bk = a.blocks[blk];
            const float* cst = lds + g.off_c + (blk & 1) * g.cfl;
            const bool more = blk + 1 < a.nblocks;
            if (more) fetch_consts(a.blocks[blk + 1]);
            // ---- first pointwise conv on this wave's own pixels: the operands are the registers of x
            pw1(bk, cst, [&](int j) { return make_float4(X[j >> 2][4 * (j & 3)], X[j >> 2][4 * (j & 3) + 1], X[j >> 2][4 * (j & 3) + 2], X[j >> 2][4 * (j & 3) + 3]); },
                my_r, in_tile, valid);
            MI_BN_STAMP(2)
            // ---- row bands: r of the rows above and below the band
            if (PERSIST && has_halo && blk == 0) {  // wave-uniform
                float* dst = rt + (hside == 0 ? 0 : g.R + 1) * g.RSa + ((hv ? hx : 0) + 1) * PSa;
                if (hrow) {
                    pw1(bk, cst, [&](int j) { return xq[PERSIST ? j : 0]; }, dst, hv, true);
                } else if (hv) {  // outside the image: zero (an earlier band of this workgroup may have left a row here)
#pragma unroll
                    for (int t = 0; t < MTA; t++)
#pragma unroll
                        for (int gq = 0; gq < 4; gq++) *reinterpret_cast<float4*>(dst + 32 * t + 8 * gq + 4 * h) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            MI_BN_STAMP(3)
            if (more) commit_consts((blk + 1) & 1);  // the other half: its last readers finished before the previous block's closing barrier
            wg_barrier();
            MI_BN_STAMP(4)
            // ---- depthwise 3x3 on r (this lane: 4 channels of its half per chunk) -> second pointwise conv over all C output channels
            nf32x16 D[MT];
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int e = 0; e < 16; e++) D[m][e] = 0.f;
            {
                const float* t0 = rt + oy * g.RSa + ox * PSa + h * Chm;  // tap (ky, kx) of this lane's pixel: + ky RSa + kx PSa
                const float* wdw = cst + OFF_WDW + h * Chm;
                auto dw = [&](int j, float4& bf) {
                    bf = nld4(cst + OFF_BDW + h * Chm + 4 * j);
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            const float4 w = nld4(wdw + (ky * 3 + kx) * Cm + 4 * j);
                            const float4 d = nld4(t0 + ky * g.RSa + kx * PSa + 4 * j);
                            bf.x = fmaf(d.x, w.x, bf.x);
                            bf.y = fmaf(d.y, w.y, bf.y);
                            bf.z = fmaf(d.z, w.z, bf.z);
                            bf.w = fmaf(d.w, w.w, bf.w);
                        }
                };
                const float* w2p = WL ? lds + g.off_w + C * Cm : bk.w2;
                auto a_frag = [&](int j, float4 (&av)[MT]) {
#pragma unroll
                    for (int m = 0; m < MT; m++) av[m] = nld4(w2p + ((m * NCH2 + j) * 64 + lane) * 4);
                };
                // weights one chunk ahead: a chunk is 4 MT MFMAs (512 - 1024 cycles), an L2 round trip
                float4 bf, av[MT];
                dw(0, bf);
                a_frag(0, av);
#pragma unroll
                for (int j = 0; j < NCH2; j++) {
                    float4 bn = bf, an[MT];
                    if (j + 1 < NCH2) { a_frag(j + 1, an); dw(j + 1, bn); }
#pragma unroll
                    for (int m = 0; m < MT; m++) {
                        D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].x, bf.x, D[m], 0, 0, 0);
                        D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].y, bf.y, D[m], 0, 0, 0);
                        D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].z, bf.z, D[m], 0, 0, 0);
                        D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].w, bf.w, D[m], 0, 0, 0);
                    }
                    bf = bn;
                    if (j + 1 < NCH2) {
#pragma unroll
                        for (int m = 0; m < MT; m++) av[m] = an[m];
                    }
                    __builtin_amdgcn_sched_barrier(0);  // chunk by chunk: hoisting every chunk's LDS reads to the top costs more registers than there are
                }
            }
            MI_BN_STAMP(5)
            // ---- + b2 + x -> activation -> x
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int ch = 32 * m + 8 * gq + 4 * h;
                    const float4 bb = nld4(cst + OFF_B2 + ch), sl = nld4(cst + OFF_SL2 + ch);
                    const float4 v = make_float4(D[m][4 * gq] + bb.x + X[m][4 * gq], D[m][4 * gq + 1] + bb.y + X[m][4 * gq + 1],
                                                 D[m][4 * gq + 2] + bb.z + X[m][4 * gq + 2], D[m][4 * gq + 3] + bb.w + X[m][4 * gq + 3]);
                    X[m][4 * gq] = fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), bk.hi2);
                    X[m][4 * gq + 1] = fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), bk.hi2);
                    X[m][4 * gq + 2] = fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), bk.hi2);
                    X[m][4 * gq + 3] = fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), bk.hi2);
                }
            if (more) wg_barrier();  // every wave has read r: the next block may overwrite it
        }
        MI_BN_STAMP(6)
        // ---- the result leaves the registers the way x came in
        if (!g.slab_sep) wg_barrier();  // every wave is done with r: the slabs may overwrite it
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++)
                *reinterpret_cast<float4*>(slab + pl * PSx + 32 * m + 8 * gq + 4 * h) = make_float4(X[m][4 * gq], X[m][4 * gq + 1], X[m][4 * gq + 2], X[m][4 * gq + 3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            float* yg = a.out + (long)frame * a.out_fs + ((long)r0 * W + wave * 32) * C;
            int ln = lane;
            asm volatile("" : "+v"(ln));  // recomputed here: shared with the write at the top, the slab offsets would stay live across the item
#pragma unroll
            for (int k = 0; k < NLD; k++)
                if (ln + 64 * k < nvalid * C4) *reinterpret_cast<float4*>(yg + 4 * (ln + 64 * k)) = nld4(slab + sl_off(ln, k));
        }
        MI_BN_STAMP(7)
        if (!PERSIST) break;
        item += gridDim.x;
        if (item >= total) break;
        wg_barrier();  // the next item overwrites r
    }
}

bool make_bneck_geom(const BneckArgs& a, BneckGeom* out) {
    BneckGeom g{};
    if (a.C % 32 || a.Cm % 32 || a.C < 64 || a.C > 128 || a.Cm < 32 || a.Cm > 64 || a.nblocks < 1 || a.nblocks > kMaxBneck) return false;
    if (a.H < 1 || a.W < 1 || a.W > 256 || a.bands < 1) return false;
    g.R = (a.H + a.bands - 1) / a.bands;
    if (g.R * a.W > 256 || (a.bands > 1 && (a.nblocks != 1 || a.C > 64 || (g.R * a.W) % 32 || 2 * ((a.W + 31) / 32) > 8))) return false;
    g.PSa = a.Cm + 4;
    g.RSa = (a.W + 2) * g.PSa;
    g.PSx = a.C + 4;
    g.cfl = (12 * a.Cm + 2 * a.C + 3) & ~3;
    const int rfl = ((g.R + 2) * g.RSa + 3) & ~3, sfl = 8 * 32 * g.PSx;
    g.slab_sep = (rfl + 2 * g.cfl + sfl) * 4 <= 160 * 1024 - 256;
    int off = g.slab_sep ? rfl : std::max(rfl, sfl);
    g.off_c = off;
    off += 2 * g.cfl;
    g.off_slab = g.slab_sep ? off : 0;
    if (g.slab_sep) off += sfl;
    g.off_w = -1;
    if (a.bands > 1) {  // row bands: the walking kernel (weights in LDS, slabs beside r)
        if (!g.slab_sep || (off + 2 * a.C * a.Cm) * 4 > 160 * 1024 - 256) return false;
        g.off_w = off;
        off += 2 * a.C * a.Cm;
    }
    g.lds_bytes = off * 4;
    if (g.lds_bytes > 160 * 1024 - 256) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!aligned16(a.in) || !aligned16(a.out) || (a.in_fs & 3) || (a.out_fs & 3)) return false;
    for (int k = 0; k < a.nblocks; k++)
        if (!a.blocks[k].w1 || !a.blocks[k].w2 || !a.blocks[k].consts || !aligned16(a.blocks[k].w1) || !aligned16(a.blocks[k].w2)) return false;
    *out = g;
    return true;
}

int bneck_cu_count() { return device_cu_count(); }  // launch.hpp: per device

template <int MT, int MTA, bool WL>
int launch_bneck_inst(const BneckArgs& a, const BneckGeom& g, hipStream_t s) {
    auto kern = bneck_kernel<MT, MTA, WL>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    // row bands: one workgroup per CU walks over its share of the bands (the next band's x in flight under the current band's MFMAs)
    unsigned grid = (unsigned)(a.B * a.bands);
    if (WL) grid = std::min(grid, (unsigned)bneck_cu_count());  // walking kernels: one workgroup per CU
    return (int)launch_kernel(kern, dim3(grid), dim3(512), (size_t)g.lds_bytes, s, a, g);
}

}  // namespace

bool bneck_kernel_supports(const BneckArgs& a) {
    BneckGeom g;
    return make_bneck_geom(a, &g);
}

int bneck_const_floats(int C, int Cm) { return 12 * Cm + 2 * C; }

int launch_bneck(const BneckArgs& a, void* stream) {
    BneckGeom g;
    if (!make_bneck_geom(a, &g)) return (int)hipErrorInvalidValue;
#ifdef MI_BNECK_STAMPS
    g.stamps = g_bneck_stamps ? g_bneck_stamps + (a.bands > 1 ? 0 : (1 << 20)) : nullptr;  // row-band launches and frame runs stamp separate areas
#endif
    hipStream_t s = (hipStream_t)stream;
    const int MT = a.C / 32, MTA = a.Cm / 32;
    const bool wl = g.off_w >= 0;
    if (MT == 2 && MTA == 1) return wl ? launch_bneck_inst<2, 1, true>(a, g, s) : launch_bneck_inst<2, 1, false>(a, g, s);
    if (MT == 2 && MTA == 2) return wl ? launch_bneck_inst<2, 2, true>(a, g, s) : launch_bneck_inst<2, 2, false>(a, g, s);
    if (wl) return (int)hipErrorInvalidValue;
    if (MT == 4 && MTA == 2) return launch_bneck_inst<4, 2, false>(a, g, s);
    if (MT == 4 && MTA == 1) return launch_bneck_inst<4, 1, false>(a, g, s);
    return (int)hipErrorInvalidValue;
}

}  // namespace mi

#ifdef MI_BNECK_STAMPS
// stamps build only (tools/bneck_stamps.py)
extern "C" void mi_debug_set_bneck_stamps(unsigned long long* p) { mi::g_bneck_stamps = p; }
#endif
