// dblock_kernels.hip — full_range's "double BlazeBlock" as ONE launch:
//
//   a = act1( W1 . (DW3x3(x) + b_dw1) + b1 )         depthwise, pointwise C -> Cm (Cm = C / 4 in the reference's graph)
//   y = act2( W2 . (DW3x3(a) + b_dw2) + b2 + x )     depthwise, pointwise Cm -> C, skip around both
//
// (face_detection_full_range.tflite: 32 -> 8 -> 32 at 96x96, 48 -> 12 -> 48 / 64 -> 16 -> 64 at 48x48, 96 -> 24 -> 96 / 128 -> 32 -> 128 at
// 24x24; DEPTHWISE_CONV_2D, CONV_2D + fused RELU, DEPTHWISE_CONV_2D, CONV_2D, ADD, RELU behind `interpreter.invoke()`,
// /root/reference/src/face_detection_lite/face_detection.rs:235)  As two block-kernel launches the pair reads
// x twice (input of the first, skip of the second — the latter in 16-byte pieces) and round-trips the narrow tensor a through HBM;
// at 128 frames both launches are latency-bound (0.18 ms per pair at 96x96 for 0.3 GB of traffic).
//
// Here one 512-thread workgroup per CU WALKS over row bands: a band of RB output rows needs a on RB + 2 rows and x on RB + 4 rows.
//   * the band's rows of x arrive lane-linear (one contiguous piece of the frame) in registers — asked for while the PREVIOUS band
//     is being computed — and are laid out as a zero-bordered LDS tensor; a lives in a second LDS tensor and never leaves the CU;
//   * both stages are the block kernel's inner loop: depthwise 3x3 on the VALU in the MFMA operand layout (lane = pixel x k-half),
//     v_mfma_f32_32x32x2_f32 over the output tiles, pointwise weights in LDS (a global load inside the loop would have to wait for the
//     next band's rows: vmcnt retires in order); stage 2 adds the skip from the centre pixel of x in LDS;
//   * halo rows of a (one above, one below the band) are recomputed; rows outside the image are written as zeros (TF SAME padding).
// Exact f32 (MFMA f32 = fmaf chain); matches two block-kernel launches to reassociation of nothing: same operations, same order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"
#include "launch.hpp"

namespace mi {

typedef float df32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ float4 dld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

constexpr int kDbPrefetch = 12;  // float4 registers per thread for the next band's rows of x

struct DblockGeom {
    int RB, bands;            // output rows per band, bands per frame
    int pair1, pair2;         // stage 1 / 2 computes two rows per lane (worth it when it shortens the busiest wave's share)
    int Cp, Cmp;              // channel counts padded to 8
    int PSx, RSx, PSa, RSa;   // LDS pixel / row strides (floats)
    int off_a, off_c, off_w1, off_w2;  // LDS offsets (floats): a tensor, constants, the two pointwise matrices
    int cfl;                  // floats of the constants blob
    int lds_bytes;
    unsigned long long* stamps;  // diagnostic builds only (MI_DBLOCK_STAMPS): 8 s_memtime stamps per wave, last band of the workgroup
};
#ifdef MI_DBLOCK_STAMPS
unsigned long long* g_dblock_stamps = nullptr;
#define MI_DB_STAMP(k) if (g.stamps && (threadIdx.x & 63) == 0) { __builtin_amdgcn_sched_barrier(0); g.stamps[((long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_DB_STAMP(k)
#endif

// constants blob (floats): [dw1 taps 9 x Cp][b_dw1 Cp][b1 32 MTA][slope1 32 MTA][dw2 taps 9 x Cmp][b_dw2 Cmp][b2 32 MT][slope2 32 MT]
template <int MT, int MTA>
__global__ __launch_bounds__(512, 2) void dblock_kernel(DblockArgs a, DblockGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xt = lds;               // x: [(RB + 4)][(W + 2)][PSx], row 0 = image row r0 - 2
    float* at = lds + g.off_a;     // a: [(RB + 2)][(W + 2)][PSa], row 0 = image row r0 - 1
    const float* cst = lds + g.off_c;
    const float* w1L = lds + g.off_w1;
    const float* w2L = lds + g.off_w2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int W = a.W, C4 = a.C >> 2;
    const int Ch = g.Cp >> 1, Chm = g.Cmp >> 1, nch1 = Ch >> 2, nch2 = Chm >> 2;
    const int o_bdw1 = 9 * g.Cp, o_b1 = 10 * g.Cp, o_sl1 = o_b1 + 32 * MTA, o_wdw2 = o_sl1 + 32 * MTA, o_bdw2 = o_wdw2 + 9 * g.Cmp, o_b2 = o_bdw2 + g.Cmp, o_sl2 = o_b2 + 32 * MT;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int total = a.B * g.bands;
    // LDS-only workgroup barrier: __syncthreads() would also wait for the next band's rows and for this band's stores
    auto wg_barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // the rows of x a band needs: image rows [r0 - 2, r0 + RB + 2) clipped to the image = one contiguous piece of the frame
    auto band_rows = [&](int it, int& frame, int& r0, int& lo, int& hi) {
        frame = it / g.bands;
        r0 = (it - frame * g.bands) * g.RB;
        lo = max(0, r0 - 2);
        hi = min(a.H, r0 + g.RB + 2);
    };
    float4 xv[kDbPrefetch];
    auto prefetch = [&](int it) {
        int frame, r0, lo, hi;
        band_rows(it, frame, r0, lo, hi);
        const int n4 = (hi - lo) * W * C4;
        const float* src = a.in + (long)frame * a.in_fs + (long)lo * W * a.C;
#pragma unroll
        for (int k = 0; k < kDbPrefetch; k++) xv[k] = dld4(src + 4 * (long)min(tid + 512 * k, n4 - 1));  // clamped; pieces past the end are not used
    };

    // ---- once per workgroup: clear both tensors (their border columns stay zero), stage the constants and the weights
    for (int i = tid; i < ((g.RB + 4) * g.RSx) >> 2; i += 512) reinterpret_cast<float4*>(xt)[i] = zero4;
    for (int i = tid; i < ((g.RB + 2) * g.RSa) >> 2; i += 512) reinterpret_cast<float4*>(at)[i] = zero4;
    for (int i = tid; i < g.cfl; i += 512) lds[g.off_c + i] = a.consts[i];
    for (int i = tid; i < (32 * MTA * g.Cp) >> 2; i += 512) reinterpret_cast<float4*>(lds + g.off_w1)[i] = dld4(a.w1 + 4 * i);
    for (int i = tid; i < (32 * MT * g.Cmp) >> 2; i += 512) reinterpret_cast<float4*>(lds + g.off_w2)[i] = dld4(a.w2 + 4 * i);
    int item = blockIdx.x;
    prefetch(item);

    // depthwise 3x3 (+ bias) of 4 channels of this lane's k-half at tap origin `t0`, then MFMAs: one 32-pixel group, NT output tiles
    auto contract = [&](const float* t0, int RS, int PS, int Cpad, int nchk, const float* wdw, const float* bdw, const float* wL, auto& D, auto NTc) {
        constexpr int NT = decltype(NTc)::value;
        const int Chh = Cpad >> 1;
#pragma unroll
        for (int m = 0; m < NT; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) D[m][e] = 0.f;
        for (int j = 0; j < nchk; j++) {
            const int c0 = h * Chh + 4 * j;
            float4 bf = dld4(bdw + c0);
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const float4 w = dld4(wdw + (ky * 3 + kx) * Cpad + c0);
                    const float4 d = dld4(t0 + ky * RS + kx * PS + c0);
                    bf.x = fmaf(d.x, w.x, bf.x);
                    bf.y = fmaf(d.y, w.y, bf.y);
                    bf.z = fmaf(d.z, w.z, bf.z);
                    bf.w = fmaf(d.w, w.w, bf.w);
                }
#pragma unroll
            for (int m = 0; m < NT; m++) {
                const float4 av = dld4(wL + ((m * nchk + j) * 64 + lane) * 4);
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf.x, D[m], 0, 0, 0);
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf.y, D[m], 0, 0, 0);
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf.z, D[m], 0, 0, 0);
                D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf.w, D[m], 0, 0, 0);
            }
        }
    };

    // One stage over the 32-pixel groups of `nrows` output rows (local row la reads the source tensor's rows la .. la + 2):
    //   out = act( W . (DW3x3(src) + b_dw) + bias + skip )
    // skip (or null): tensor K, the output pixel's own position at K's local row la + krow, channels < Ck.  TOLDS: the result goes to the
    // LDS tensor dstL (local row la; zero where the image row img0 + la lies outside the image), else to the frame in global memory.
    // pair: a lane computes TWO vertically adjacent pixels — the 4 x 3 window they share is 12 LDS reads instead of 18 and the nine
    // depthwise taps are read once for both; the LDS pipe bounds these stages (per channel chunk and pixel 9 + 9 + 2 reads, paired 6 + 4.5 + 1).
    auto stage = [&](auto NTc, auto TOLDSc, auto PAIROKc, bool pair, int nrows, int img0, const float* S, int RSs, int PSs, int Cpad, int nchk, const float* wdw,
                     const float* bdw, const float* wL, const float* bias, const float* slope, float hi, int Cout, const float* K, int RSk, int PSk,
                     int krow, int Ck, float* dstL, float* dstG) {
        constexpr int NT = decltype(NTc)::value;
        constexpr bool TOLDS = decltype(TOLDSc)::value;
        const int Chh = Cpad >> 1;
        auto finish = [&](const df32x16 (&D)[NT], int la, int ox, bool valid) {
            const int iy = img0 + la;
            const bool inside = iy >= 0 && iy < a.H;
            const float* ks = K ? K + (la + krow) * RSk + (ox + 1) * PSk : nullptr;
            float* dl = TOLDS ? dstL + la * g.RSa + (ox + 1) * g.PSa : nullptr;
            float* dg = TOLDS ? nullptr : dstG + ((long)la * W + ox) * Cout;
#pragma unroll
            for (int m = 0; m < NT; m++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int ch = 32 * m + 8 * gq + 4 * h;
                    if (ch >= (TOLDS ? g.Cmp : Cout) || !valid) continue;
                    const float4 bb = dld4(bias + ch), sl = dld4(slope + ch);
                    float4 sk = make_float4(0.f, 0.f, 0.f, 0.f);  // channels above Ck: the zero channel-pad of the skip
                    if (ks && ch < Ck) sk = dld4(ks + ch);
                    float4 v = make_float4(D[m][4 * gq] + bb.x + sk.x, D[m][4 * gq + 1] + bb.y + sk.y, D[m][4 * gq + 2] + bb.z + sk.z, D[m][4 * gq + 3] + bb.w + sk.w);
                    v.x = fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), hi);
                    v.y = fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), hi);
                    v.z = fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), hi);
                    v.w = fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), hi);
                    if (TOLDS) {
                        if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);  // rows outside the image are the next depthwise conv's zero padding
                        *reinterpret_cast<float4*>(dl + ch) = v;
                    } else {
                        *reinterpret_cast<float4*>(dg + ch) = v;
                    }
                }
        };
        constexpr bool PAIROK = decltype(PAIROKc)::value;  // the two-row form exists for <= 2 output tiles (accumulators)
        if (!PAIROK || !pair) {
            const int npx = nrows * W, ngr = (npx + 31) >> 5;
            for (int grp = wave; grp < ngr; grp += 8) {  // wave-uniform
                const int q = grp * 32 + pl;
                const bool valid = q < npx;
                const int la = valid ? q / W : 0, ox = valid ? q - (q / W) * W : 0;
                df32x16 D[NT];
                contract(S + la * RSs + ox * PSs, RSs, PSs, Cpad, nchk, wdw, bdw, wL, D, NTc);
                finish(D, la, ox, valid);
            }
        } else if constexpr (PAIROK) {
            const int prows = (nrows + 1) >> 1, npq = prows * W, ngr = (npq + 31) >> 5;
            for (int grp = wave; grp < ngr; grp += 8) {  // wave-uniform
                const int qq = grp * 32 + pl;
                const bool vq = qq < npq;
                const int pr = vq ? qq / W : 0, ox = vq ? qq - (qq / W) * W : 0, la0 = 2 * pr;
                const float* t0 = S + la0 * RSs + ox * PSs;  // tap rows la0 .. la0 + 3
                df32x16 D0[NT], D1[NT];
#pragma unroll
                for (int m = 0; m < NT; m++)
#pragma unroll
                    for (int e = 0; e < 16; e++) { D0[m][e] = 0.f; D1[m][e] = 0.f; }
                for (int j = 0; j < nchk; j++) {
                    const int c0 = h * Chh + 4 * j;
                    const float4 bb = dld4(bdw + c0);
                    float4 bf0 = bb, bf1 = bb;
                    // data row r feeds tap row r of the upper pixel and tap row r - 1 of the lower one (same tap order as the one-row form)
                    float4 wprev[3];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float4 d[3], wr[3];
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            d[kx] = dld4(t0 + r * RSs + kx * PSs + c0);
                            if (r < 3) wr[kx] = dld4(wdw + (r * 3 + kx) * Cpad + c0);
                        }
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            if (r < 3) { bf0.x = fmaf(d[kx].x, wr[kx].x, bf0.x); bf0.y = fmaf(d[kx].y, wr[kx].y, bf0.y); bf0.z = fmaf(d[kx].z, wr[kx].z, bf0.z); bf0.w = fmaf(d[kx].w, wr[kx].w, bf0.w); }
                            if (r > 0) { bf1.x = fmaf(d[kx].x, wprev[kx].x, bf1.x); bf1.y = fmaf(d[kx].y, wprev[kx].y, bf1.y); bf1.z = fmaf(d[kx].z, wprev[kx].z, bf1.z); bf1.w = fmaf(d[kx].w, wprev[kx].w, bf1.w); }
                        }
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) wprev[kx] = wr[kx];
                    }
#pragma unroll
                    for (int m = 0; m < NT; m++) {
                        const float4 av = dld4(wL + ((m * nchk + j) * 64 + lane) * 4);
                        D0[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf0.x, D0[m], 0, 0, 0);
                        D1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf1.x, D1[m], 0, 0, 0);
                        D0[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf0.y, D0[m], 0, 0, 0);
                        D1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf1.y, D1[m], 0, 0, 0);
                        D0[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf0.z, D0[m], 0, 0, 0);
                        D1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf1.z, D1[m], 0, 0, 0);
                        D0[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf0.w, D0[m], 0, 0, 0);
                        D1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf1.w, D1[m], 0, 0, 0);
                    }
                }
                finish(D0, la0, ox, vq);
                finish(D1, la0 + 1, ox, vq && la0 + 1 < nrows);
            }
        }
    };

    MI_DB_STAMP(0)
    for (;;) {
        int frame, r0, lo, hi;
        band_rows(item, frame, r0, lo, hi);
        MI_DB_STAMP(1)
        // ---- x: registers -> zero-bordered LDS tensor; row slots without an image row are cleared
        {
            const int rowf4 = W * C4, n4 = (hi - lo) * rowf4;
            // float4 number tid + 512 k -> (row, pixel, channel quad) without a division per element (two software divisions per
            // element were a third of a band's time): the position advances by 512 float4s = dpx pixels + dc4 quads per step
            int px = tid / C4, c4 = tid - px * C4;          // position within the band's rows, in pixels (all rows concatenated)
            const int dpx = 512 / C4, dc4 = 512 - dpx * C4;
            int rr = px / W;
            px -= rr * W;
#pragma unroll
            for (int k = 0; k < kDbPrefetch; k++) {
                if (tid + 512 * k < n4) *reinterpret_cast<float4*>(xt + (lo - (r0 - 2) + rr) * g.RSx + (px + 1) * g.PSx + 4 * c4) = xv[k];
                c4 += dc4; px += dpx;
                if (c4 >= C4) { c4 -= C4; px++; }
                while (px >= W) { px -= W; rr++; }
            }
            for (int s = 0; s < g.RB + 4; s++) {  // uniform
                const int iy = r0 - 2 + s;
                if (iy >= lo && iy < hi) continue;
                for (int e = tid; e < (W * g.PSx) >> 2; e += 512) *reinterpret_cast<float4*>(xt + s * g.RSx + g.PSx + 4 * e) = zero4;
            }
        }
        if (item + (int)gridDim.x < total) prefetch(item + gridDim.x);  // the next band's rows travel under this band's work
        MI_DB_STAMP(2)
        wg_barrier();
        MI_DB_STAMP(3)
        // ---- stage 1: a on rows [r0 - 1, r0 + RB + 1) (rows outside the image: zeros)
        stage(std::integral_constant<int, MTA>{}, std::true_type{}, std::true_type{}, g.pair1 != 0, g.RB + 2, r0 - 1, xt, g.RSx, g.PSx, g.Cp, nch1, cst, cst + o_bdw1, w1L,
              cst + o_b1, cst + o_sl1, a.hi1, a.Cm, a.skip1 ? xt : nullptr, g.RSx, g.PSx, 1, a.C, at, nullptr);
        MI_DB_STAMP(4)
        wg_barrier();
        MI_DB_STAMP(5)
        // ---- stage 2: y on rows [r0, r0 + RB) inside the image; its skip is x (double block) or a (a pair of BlazeBlocks)
        stage(std::integral_constant<int, MT>{}, std::false_type{}, std::integral_constant<bool, (MT <= 2)>{}, g.pair2 != 0, min(g.RB, a.H - r0), r0, at, g.RSa, g.PSa, g.Cmp, nch2, cst + o_wdw2, cst + o_bdw2, w2L,
              cst + o_b2, cst + o_sl2, a.hi2, a.Co, a.skip2_from_a ? at : xt, a.skip2_from_a ? g.RSa : g.RSx, a.skip2_from_a ? g.PSa : g.PSx, a.skip2_from_a ? 1 : 2,
              a.skip2_from_a ? a.Cm : a.C, nullptr, a.out + (long)frame * a.out_fs + (long)r0 * W * a.Co);
        MI_DB_STAMP(6)
        item += gridDim.x;
        if (item >= total) break;
        wg_barrier();  // the next band overwrites both tensors
    }
}

bool make_dblock_geom(const DblockArgs& a, DblockGeom* out) {
    DblockGeom g{};
    if (a.C % 8 || a.C < 8 || a.C > 128 || a.Cm % 4 || a.Cm < 4 || a.Cm > 64 || a.H < 1 || a.W < 2 || a.B < 1) return false;
    if (a.Co > 128 || a.Co % 4) return false;
    if (a.skip2_from_a ? a.Co != a.Cm : a.Co < a.C) return false;
    if (a.skip1 && a.Cm != a.C) return false;
    g.Cp = a.C;
    g.Cmp = (a.Cm + 7) & ~7;
    g.PSx = g.Cp + 4; g.RSx = (a.W + 2) * g.PSx;
    g.PSa = g.Cmp + 4; g.RSa = (a.W + 2) * g.PSa;
    const int MT = (a.Co + 31) / 32, MTA = (a.Cm + 31) / 32;
    if (MTA == 2 && MT != 2) return false;  // instantiated: one stage-1 tile with 1 .. 4 stage-2 tiles, or two and two
    g.cfl = 10 * g.Cp + 64 * MTA + 10 * g.Cmp + 64 * MT;
    const int wfl = 32 * MTA * g.Cp + 32 * MT * g.Cmp;
    // rows per band: as many as LDS and the prefetch registers hold (fewer bands = less halo work)
    // ... but a handful of frames (a single-image call, face_detection.rs:205) would then be a handful of workgroups on 256 CUs: there the
    // bands get as short as it takes to have about a workgroup per CU (at least two rows; the halo rows of stage 1 are recomputed per band)
    int RB = 0;
    int cap = 32;
    {
        static const int env_cap = getenv("MI_DBLOCK_RBCAP") ? atoi(getenv("MI_DBLOCK_RBCAP")) : 0;   // tuning aid
        const long cus = device_cu_count();
        if ((long)a.B * ((a.H + 31) / 32) * 4 <= cus) cap = std::max(2, (int)(((long)a.H * a.B + cus - 1) / cus));
        if (env_cap > 0) cap = env_cap;
    }
    for (int r = std::min(std::min(a.H, 32), cap); r >= 1; r--) {
        const long fl = (long)(r + 4) * g.RSx + (long)(r + 2) * g.RSa + g.cfl + wfl + 16;
        const long n4 = (long)(r + 4) * a.W * (a.C / 4);
        if (fl * 4 <= 160 * 1024 - 256 && n4 <= (long)kDbPrefetch * 512) { RB = r; break; }
    }
    if (RB < 2) return false;
    // stage 1 with two rows per lane needs an even band height; worth it when it shortens the busiest wave's share (a pair group
    // costs about 1.3 single groups) and giving up an odd row does not cost a short band a third of its height
    g.pair1 = 0;
    if (RB % 2 == 0 || RB >= 5) {
        const int RBe = RB & ~1, bands = (a.H + RBe - 1) / RBe, rb = (((a.H + bands - 1) / bands) + 1) & ~1;
        const int single = ((rb + 2) * a.W + 31) / 32, pairs = ((rb + 2) / 2 * a.W + 31) / 32;
        if (13 * ((pairs + 7) / 8) < 10 * ((single + 7) / 8)) { g.pair1 = 1; g.bands = bands; g.RB = rb; }
    }
    if (!g.pair1) {
        g.bands = (a.H + RB - 1) / RB;
        g.RB = (a.H + g.bands - 1) / g.bands;  // bands of equal height
    }
    {
        const int single = (g.RB * a.W + 31) / 32, pairs = ((g.RB + 1) / 2 * a.W + 31) / 32;
        g.pair2 = MT <= 2 && 13 * ((pairs + 7) / 8) < 10 * ((single + 7) / 8);  // the accumulators of 2 x 2 tiles fit, those of 2 x 4 do not
    }
    int off = (g.RB + 4) * g.RSx;
    off = (off + 3) & ~3;
    g.off_a = off; off += (g.RB + 2) * g.RSa; off = (off + 3) & ~3;
    g.off_c = off; off += (g.cfl + 3) & ~3;
    g.off_w1 = off; off += 32 * MTA * g.Cp;
    g.off_w2 = off; off += 32 * MT * g.Cmp;
    g.lds_bytes = off * 4;
    if (g.lds_bytes > 160 * 1024 - 256) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!aligned16(a.in) || !aligned16(a.out) || !aligned16(a.w1) || !aligned16(a.w2) || !a.consts || (a.in_fs & 3) || (a.out_fs & 3)) return false;
    *out = g;
    return true;
}

int dblock_cu_count() { return device_cu_count(); }  // launch.hpp: per device

template <int MT, int MTA>
int launch_dblock_inst(const DblockArgs& a, const DblockGeom& g, hipStream_t s) {
    auto kern = dblock_kernel<MT, MTA>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    const unsigned grid = (unsigned)std::min<long>((long)a.B * g.bands, dblock_cu_count());
    return (int)launch_kernel(kern, dim3(grid), dim3(512), (size_t)g.lds_bytes, s, a, g);
}

}  // namespace

bool dblock_kernel_supports(const DblockArgs& a) {
    DblockGeom g;
    return make_dblock_geom(a, &g);
}

int dblock_const_floats(int C, int Cm, int Co) {
    const int Cmp = (Cm + 7) & ~7, MT = (Co + 31) / 32, MTA = (Cm + 31) / 32;
    return 10 * C + 64 * MTA + 10 * Cmp + 64 * MT;
}

int launch_dblock(const DblockArgs& a, void* stream) {
    DblockGeom g;
    if (!make_dblock_geom(a, &g)) return (int)hipErrorInvalidValue;
#ifdef MI_DBLOCK_STAMPS
    static const int only_h = getenv("MI_DB_H") ? atoi(getenv("MI_DB_H")) : 0;  // stamp the launches on frames of this height only
    g.stamps = (only_h == 0 || a.H == only_h) ? g_dblock_stamps : nullptr;
#endif
    hipStream_t s = (hipStream_t)stream;
    if ((a.Cm + 31) / 32 == 2) return launch_dblock_inst<2, 2>(a, g, s);
    switch ((a.Co + 31) / 32) {
        case 1: return launch_dblock_inst<1, 1>(a, g, s);
        case 2: return launch_dblock_inst<2, 1>(a, g, s);
        case 3: return launch_dblock_inst<3, 1>(a, g, s);
        case 4: return launch_dblock_inst<4, 1>(a, g, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace mi

#ifdef MI_DBLOCK_STAMPS
// stamps build only (tools/dblock_stamps.py)
extern "C" void mi_debug_set_dblock_stamps(unsigned long long* p) { mi::g_dblock_stamps = p; }
#endif
