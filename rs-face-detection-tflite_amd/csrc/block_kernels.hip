// block_kernels.hip — the hot kernel: one fused BlazeBlock per launch on gfx950 (CDNA4).
//
//   out = act( PW1x1( DW3x3(in) + b_dw ) + b_pw + skip )            (KS = 3)      or
//   out = act( PW1x1( in ) + b_pw + skip )                           (KS = 1, pointwise only)
//
// replaces the DEPTHWISE_CONV_2D -> CONV_2D(1x1) -> ADD -> RELU/PRELU op chain (+ MAX_POOL_2D / channel PAD on the skip
// path) that TensorFlow-Lite runs op by op behind `interpreter.invoke()` (/root/reference/src/face_detection_lite/
// face_detection.rs:235, face_landmark.rs:265, iris_landmark.rs:203; graphs: SURVEY.md Appendix A).
//
// Design (HBM-bound: every activation byte is read once and written once):
//   * one 256-thread workgroup owns a band of output rows of ONE frame and walks down it.  Input rows live in an LDS
//     ring (NR = (R-1)*S + KS rows for R output rows per step); each input row is fetched from HBM exactly once per band
//     (KS-S halo rows per band are the only re-reads).  The rows a step adds are one contiguous block of the NHWC frame,
//     so the 16 B/lane loads (and the stores) are fully coalesced.  Bands are sized so that the grid is about one
//     resident wave of workgroups (2 per CU): the prologue is paid once per ~64 rows.
//   * the next step's rows are issued into registers BEFORE the current step is computed and written to LDS after it
//     (issue-early / write-late), so HBM latency hides under the DW + MFMA work.  The hot variants contain no other
//     global load in the compute phase (bias/slopes/weights sit in LDS): vmcnt is in-order, a stray load would drain it.
//   * LDS pixel stride is Cp + 4 floats: with Cp a multiple of 8 the 16 lanes of a ds_read_b128 group hit 16 distinct
//     4-bank slots (conflict-free) when each lane reads its own pixel.
//   * depthwise 3x3 on the VALU, 4 channels (one float4) at a time, computed directly in the MFMA operand layout:
//     lane l = (pixel l&31, k-half l>>5) owns the channels [h*Cp/2, (h+1)*Cp/2) of its pixel, so the DW result IS the
//     B operand of v_mfma_f32_32x32x2_f32 (B[k = l>>5][n = l&31]) with the contraction index permuted consistently in
//     the pre-packed A operand (pointwise weights, M = 32 output channels per tile).  No LDS round trip for DW outputs.
//     LDS reads run one filter row ahead of the FMAs that consume them.
//   * exact f32 MFMA (same numerics as an fmaf chain); D layout gives each lane 4 consecutive output channels per
//     register quad -> float4 epilogue: + bias + skip (from the LDS ring when the skip is the block input, incl. the
//     fused 2x2 max-pool of stride-2 blocks and the zero channel-pad) -> activation -> 16-byte stores.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "kernels.hpp"
#include "launch.hpp"

namespace mi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kPrefetch = 8;          // float4 registers per thread for the next step's rows
constexpr int kALdsMax = 40 * 1024;   // pointwise weights are staged in LDS when they fit in this many bytes
static int cu_count() { return device_cu_count(); }  // launch.hpp: per device

int lds_budget() {
    static const int v = getenv("MI_BLOCK_LDS") ? atoi(getenv("MI_BLOCK_LDS")) : 80 * 1024;  // tuning aid
    return v;
}

struct BlockGeom {
    int Cp, Ch, C4;        // padded channels (mult of 8), channels per k-half, float4s per real pixel
    int PS, RS;            // LDS pixel stride / row stride in floats
    int R, NR;             // output rows per step, ring rows
    int band, bands;       // output rows per workgroup, workgroups per frame
    int MT;                // 32-row output-channel tiles
    int a_lds;             // pointwise weights staged in LDS
    int off_wdw, off_bdw, off_a, off_bias, off_alpha;  // LDS offsets in floats
    int res_lds;           // skip connection served from the LDS ring
    int lds_bytes;
    int PG;
    // Small grids (few frames x few pixels, many output channels: the 12x12 / 6x6 layers of full_range at 128 frames): the output
    // tiles of a band are spread over nsplit workgroups (the depthwise stage is recomputed per pass anyway) and, when a step has
    // at most two 32-pixel groups, over the waves that would have no pixels (wmod groups; wave -> group wave % wmod, tile share wave / wmod)
    int nsplit, wmod, mt_per;  // mt_per: tiles per (workgroup, tile share)
    unsigned long long* stamps;  // diagnostic builds only (MI_BLOCK_STAMPS): 8 accumulators per wave
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 max4(float4 a, float4 b, float4 c, float4 d) {
    return make_float4(fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y)),
                       fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)), fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w)));
}

__device__ __forceinline__ float4 res_from_global(const Epilogue& ep, int b, int oy, int ox, int Wo, int ch) {
    // generic float4 skip read (channels ch..ch+3 < res_C guaranteed by the caller)
    const float* r = ep.res + (long)b * ep.res_fs;
    if (ep.res_mode == RES_DIRECT) return ld4(r + ((long)oy * Wo + ox) * ep.res_C + ch);
    if (ep.res_mode == RES_MAXPOOL) {
        const float* p = r + ((long)(2 * oy) * ep.res_W + 2 * ox) * ep.res_C + ch;
        return max4(ld4(p), ld4(p + ep.res_C), ld4(p + (long)ep.res_W * ep.res_C), ld4(p + (long)ep.res_W * ep.res_C + ep.res_C));
    }
    // RES_UP2X (TFLite ResizeBilinear, half_pixel_centers)
    float iy = ((float)oy + 0.5f) * 0.5f - 0.5f, ix = ((float)ox + 0.5f) * 0.5f - 0.5f;
    int y0 = max((int)floorf(iy), 0), y1 = min((int)ceilf(iy), ep.res_H - 1);
    int x0 = max((int)floorf(ix), 0), x1 = min((int)ceilf(ix), ep.res_W - 1);
    float dy = iy - (float)y0, dx = ix - (float)x0;
    float4 p00 = ld4(r + ((long)y0 * ep.res_W + x0) * ep.res_C + ch), p01 = ld4(r + ((long)y0 * ep.res_W + x1) * ep.res_C + ch);
    float4 p10 = ld4(r + ((long)y1 * ep.res_W + x0) * ep.res_C + ch), p11 = ld4(r + ((long)y1 * ep.res_W + x1) * ep.res_C + ch);
    float w00 = (1 - dy) * (1 - dx), w10 = dy * (1 - dx), w01 = (1 - dy) * dx, w11 = dy * dx;
    return make_float4(p00.x * w00 + p10.x * w10 + p01.x * w01 + p11.x * w11, p00.y * w00 + p10.y * w10 + p01.y * w01 + p11.y * w11,
                       p00.z * w00 + p10.z * w10 + p01.z * w01 + p11.z * w11, p00.w * w00 + p10.w * w10 + p01.w * w01 + p11.w * w11);
}

// MTG: output-channel tiles kept in accumulators at once; S: stride; KS: 3 (depthwise stage) or 1 (pointwise only);
// PG: 32-pixel groups per wave per step; CPT: compile-time padded channel count (0 = runtime); SLOW: variants that may stream the pointwise weights from global/L2 and read
// the skip from global memory (large-channel / cross-tensor-skip layers; small, never on the hot path).
// ALDS: the pointwise weights are staged in LDS (g.a_lds), else streamed from L2.  A template parameter, not a run-time choice:
// selecting between an LDS and a global pointer at run time turns the load into a FLAT load, whose wait covers vmcnt and lgkmcnt —
// every weight fragment then also waits for the next step's rows, which are meant to stay in flight across the compute phase.
// The SLOW variants with one or two accumulator tiles are held to 256 registers (two workgroups per CU): left alone they take ~300
// (prefetched skip quads + row prefetch + accumulators), one workgroup per CU, and the layers whose tiles are spread over 2-4
// workgroups per frame (6x6 / 12x12 at 128 frames: 384-512 workgroups) ran in two rounds: 36.6 -> 24.6 us for 6x6x96 -> 384.
// With three or four tiles the spills cost more than the second workgroup brings (12x12x64 -> 256: 33.9 -> 44.9 us), so those stay.
template <int MTG, int S, int KS, int PG, bool SLOW, int CPT, bool ALDS>
__global__ __launch_bounds__(256, (SLOW && MTG <= 2) ? 2 : 1) void block_kernel(BlockArgs a, BlockGeom gin) {
    // CPT > 0: padded channel count known at compile time (hot shapes) so that every tap / chunk offset of the LDS reads
    // folds into the ds_read immediate instead of costing a VALU add per load (49 of ~160 VALU ops per chunk at runtime Cp).
    BlockGeom g = gin;
    if (CPT > 0) { g.Cp = CPT; g.Ch = CPT / 2; g.PS = CPT + 4; }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tile = lds;
    const float* wdw = lds + g.off_wdw;
    const float* bdw = lds + g.off_bdw;
    const float* aL = lds + g.off_a;
    const float* biasL = lds + g.off_bias;    // [Cop] pointwise bias (0 when absent)
    const float* alphaL = lds + g.off_alpha;  // [Cop] negative-side slope of the activation

#ifdef MI_BLOCK_STAMPS
    const unsigned long long st_kernel_start = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int bsp = blockIdx.x / g.nsplit, sp = blockIdx.x - bsp * g.nsplit;
    const int b = bsp / g.bands, band = bsp % g.bands;
    const int wq = wave % g.wmod, wshare = wave / g.wmod;
    const int mt_lo = min(g.MT, (sp * (4 / g.wmod) + wshare) * g.mt_per), mt_hi = min(g.MT, mt_lo + g.mt_per);
    const int oy0 = band * g.band, oy1 = min(oy0 + g.band, a.Ho);
    const int iy_base = oy0 * S - a.pt;  // input row held by ring slot 0 at band start
    const float* in = a.in + (long)b * a.in_fs;
    const int rowf4 = a.W * g.C4;        // float4s of one real input row
    const int Cop = g.MT * 32;

    // ---- prologue: clear the ring (pad pixels / out-of-image rows stay zero), stage constants, load the first rows
    for (int i = tid; i < (g.NR * g.RS) >> 2; i += 256) reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KS == 3) {
        for (int i = tid; i < 9 * g.Cp; i += 256) {
            int c = i % g.Cp;
            lds[g.off_wdw + i] = c < a.C ? a.w_dw[(i / g.Cp) * a.C + c] : 0.f;
        }
        for (int i = tid; i < g.Cp; i += 256) lds[g.off_bdw + i] = (i < a.C && a.b_dw) ? a.b_dw[i] : 0.f;
    }
    for (int i = tid; i < Cop; i += 256) {
        lds[g.off_bias + i] = (i < a.Co && a.ep.bias) ? a.ep.bias[i] : 0.f;
        // negative-side slope: PReLU alpha, 0 for ReLU/ReLU6, 1 for "no activation"
        lds[g.off_alpha + i] = (i < a.Co && a.ep.act == ACT_PRELU) ? a.ep.alpha[i] : (a.ep.act == ACT_NONE ? 1.f : 0.f);
    }
    if (ALDS)
        for (int i = tid; i < (g.MT * 32 * g.Cp) >> 2; i += 256) reinterpret_cast<float4*>(lds + g.off_a)[i] = ld4(a.w_pw + 4 * (long)i);

    // prefetch geometry: float4 number (tid + 256k) of the contiguous block of new rows -> (row, LDS offset in its slot)
    int pf_pack[kPrefetch];  // row << 20 | LDS float offset within the row slot
#pragma unroll
    for (int k = 0; k < kPrefetch; k++) {
        int i = tid + k * 256;
        int r = 0;
#pragma unroll
        for (int t = 1; t < kPrefetch; t++) r += (i >= t * rowf4) ? 1 : 0;
        int e = i - r * rowf4;
        int px = e / g.C4;
        pf_pack[k] = (r << 20) | ((px + 1) * g.PS + 4 * (e - px * g.C4));
    }
    __syncthreads();
    {
        const int nrows = min(g.NR, (oy1 - oy0 - 1) * S + KS);
        for (int r = 0; r < nrows; r++) {
            int iy = iy_base + r;
            if (iy < 0 || iy >= a.H) continue;
            const float* src = in + (long)iy * a.W * a.C;
            for (int e = tid; e < rowf4; e += 256) {
                int px = e / g.C4;
                *reinterpret_cast<float4*>(tile + r * g.RS + (px + 1) * g.PS + 4 * (e - px * g.C4)) = ld4(src + 4 * e);
            }
        }
    }
    __syncthreads();

    // ---- step-invariant lane geometry: pixel slot -> (row within step, column)
    int rp[PG], ox[PG];
#pragma unroll
    for (int p = 0; p < PG; p++) {
        int q = (wq * PG + p) * 32 + pl;
        rp[p] = q / a.Wo;
        ox[p] = q - rp[p] * a.Wo;
    }
    const float hi = a.ep.act == ACT_RELU6 ? 6.f : INFINITY;
    const int rk = a.ep.res_mode == RES_NONE ? 0 : (!g.res_lds ? 3 : (a.ep.res_mode == RES_DIRECT ? 1 : 2));

    const int nsteps = (oy1 - oy0 + g.R - 1) / g.R;
    int ring0 = 0;  // ring slot of the first input row of the current step
#ifdef MI_BLOCK_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long st_prologue = st_prev - st_kernel_start;
#define MI_STAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); st_acc[k] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_STAMP(k)
#endif
    for (int step = 0; step < nsteps; step++) {
        const int oys = oy0 + step * g.R;                  // first output row of this step
        auto wrap = [&](int x) { return x >= g.NR ? x - g.NR : x; };  // x < 2*NR
        // ---- issue the loads of the rows the NEXT step adds to the ring (kept in registers across the compute phase).
        // They are consecutive rows of the frame = one contiguous block; rows below the image stay zero.
        float4 pf[kPrefetch];
        const int new_lo = (oys - oy0 + g.R - 1) * S + KS; // band-relative index of the first new row
        const bool more = step + 1 < nsteps;
        const int new_n = more ? min(g.R, oy1 - (oys + g.R)) * S : 0;  // rows the next step really needs
        if (more) {
            const int iy_first = iy_base + new_lo;          // >= 0 always (new rows lie below rows already loaded)
            const int in_image = min(new_n, a.H - iy_first) * rowf4;
            const float* src = in + (long)iy_first * a.W * a.C;
#pragma unroll
            for (int k = 0; k < kPrefetch; k++) {
                int i = tid + k * 256;
                pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < in_image) pf[k] = ld4(src + 4 * i);
            }
        }

        MI_STAMP(0)
        // ---- compute: this wave's PG groups of 32 output pixels
        bool valid[PG];
        int base[PG][KS];  // LDS float offset of tap row ky, kx = 0, channel h*Ch
#pragma unroll
        for (int p = 0; p < PG; p++) {
            valid[p] = rp[p] < g.R && oys + rp[p] < oy1;
            int rr = valid[p] ? rp[p] : 0, oxc = valid[p] ? ox[p] : 0;
#pragma unroll
            for (int ky = 0; ky < KS; ky++) base[p][ky] = wrap(ring0 + rr * S + ky) * g.RS + (oxc * S - a.pl + 1) * g.PS + h * g.Ch;
        }
        for (int mt0 = mt_lo; mt0 < mt_hi; mt0 += MTG) {
            // Skip connection from another tensor (SLOW variants): its 16-byte pieces are asked for BEFORE the contraction and land under
            // the MFMAs; read in the epilogue, each of the MTG * 4 quads paid a full memory round trip after the last MFMA.
            float4 rvq[SLOW ? PG : 1][SLOW ? MTG : 1][4];
            if constexpr (SLOW) {
                if (rk == 3) {
#pragma unroll
                    for (int p = 0; p < PG; p++)
#pragma unroll
                        for (int m = 0; m < MTG; m++)
#pragma unroll
                            for (int gq = 0; gq < 4; gq++) {
                                const int ch = (mt0 + m) * 32 + 8 * gq + 4 * h;
                                rvq[p][m][gq] = make_float4(0.f, 0.f, 0.f, 0.f);
                                if (valid[p] && mt0 + m < mt_hi && ch < a.Co && ch < a.ep.res_C) rvq[p][m][gq] = res_from_global(a.ep, b, oys + rp[p], ox[p], a.Wo, ch);
                            }
                }
            }
            f32x16 D[PG][MTG];
#pragma unroll
            for (int p = 0; p < PG; p++)
#pragma unroll
                for (int m = 0; m < MTG; m++)
#pragma unroll
                    for (int e = 0; e < 16; e++) D[p][m][e] = 0.f;
            if constexpr (MTG < 2) {
#pragma unroll 1
            for (int j = 0; j < (g.Ch >> 2); j++) {
                // A fragments of tiles mt0.., k-steps 4j..4j+3: packed [mt][j][lane][4]
                float4 av[MTG];
#pragma unroll
                for (int m = 0; m < MTG; m++) {
                    int mt = min(mt0 + m, g.MT - 1);
                    long ao = (((long)mt * (g.Ch >> 2) + j) * 64 + lane) * 4;
                    av[m] = ALDS ? ld4(aL + ao) : ld4(a.w_pw + ao);
                }
                float4 bf[PG];
                if (KS == 3) {
                    // depthwise 3x3 for 4 channels of this lane's k-half; loads are issued one filter row ahead of the
                    // FMAs that consume them (explicit double buffer: the LDS latency hides under the previous row's math)
                    const float* wj = wdw + h * g.Ch + 4 * j;
                    float4 wq[3], dq[PG][3];
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        wq[kx] = ld4(wj + kx * g.Cp);
#pragma unroll
                        for (int p = 0; p < PG; p++) dq[p][kx] = ld4(tile + base[p][0] + kx * g.PS + 4 * j);
                    }
#pragma unroll
                    for (int p = 0; p < PG; p++) bf[p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int ky = 0; ky < 3; ky++) {
                        float4 wn[3], dn[PG][3];
                        if (ky < 2) {
#pragma unroll
                            for (int kx = 0; kx < 3; kx++) {
                                wn[kx] = ld4(wj + ((ky + 1) * 3 + kx) * g.Cp);
#pragma unroll
                                for (int p = 0; p < PG; p++) dn[p][kx] = ld4(tile + base[p][ky + 1 < KS ? ky + 1 : 0] + kx * g.PS + 4 * j);
                            }
                        }
#pragma unroll
                        for (int kx = 0; kx < 3; kx++)
#pragma unroll
                            for (int p = 0; p < PG; p++) {
                                bf[p].x = fmaf(dq[p][kx].x, wq[kx].x, bf[p].x);
                                bf[p].y = fmaf(dq[p][kx].y, wq[kx].y, bf[p].y);
                                bf[p].z = fmaf(dq[p][kx].z, wq[kx].z, bf[p].z);
                                bf[p].w = fmaf(dq[p][kx].w, wq[kx].w, bf[p].w);
                            }
                        if (ky < 2) {
#pragma unroll
                            for (int kx = 0; kx < 3; kx++) {
                                wq[kx] = wn[kx];
#pragma unroll
                                for (int p = 0; p < PG; p++) dq[p][kx] = dn[p][kx];
                            }
                        }
                    }
                    float4 bb = ld4(bdw + h * g.Ch + 4 * j);
#pragma unroll
                    for (int p = 0; p < PG; p++) { bf[p].x += bb.x; bf[p].y += bb.y; bf[p].z += bb.z; bf[p].w += bb.w; }
                } else {
#pragma unroll
                    for (int p = 0; p < PG; p++) bf[p] = ld4(tile + base[p][0] + 4 * j);
                }
#pragma unroll
                for (int m = 0; m < MTG; m++) {
#pragma unroll
                    for (int p = 0; p < PG; p++) {
                        D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].x, bf[p].x, D[p][m], 0, 0, 0);
                        D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].y, bf[p].y, D[p][m], 0, 0, 0);
                        D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].z, bf[p].z, D[p][m], 0, 0, 0);
                        D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].w, bf[p].w, D[p][m], 0, 0, 0);
                    }
                }
            }
            } else {
            // Two or more output-channel tiles: software pipeline over channel chunks — the MFMAs of chunk j (matrix pipe)
            // are issued in the same basic block as the depthwise math of chunk j+1 (VALU + LDS) and interleaved with it
            // (sched_group_barrier), +4..8 % on the 48- and 96-channel layers.  With one tile the extra live registers
            // cost a resident workgroup, so the single-tile variants keep the simple loop above.
                const int nch = g.Ch >> 2;
                auto dw_chunk = [&](int j, float4 (&bf)[PG]) {
                    if (KS == 3) {
                        const float* wj = wdw + h * g.Ch + 4 * j;
#pragma unroll
                        for (int p = 0; p < PG; p++) bf[p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int ky = 0; ky < 3; ky++)
#pragma unroll
                            for (int kx = 0; kx < 3; kx++) {
                                const float4 w = ld4(wj + (ky * 3 + kx) * g.Cp);
#pragma unroll
                                for (int p = 0; p < PG; p++) {
                                    const float4 d = ld4(tile + base[p][ky] + kx * g.PS + 4 * j);
                                    bf[p].x = fmaf(d.x, w.x, bf[p].x);
                                    bf[p].y = fmaf(d.y, w.y, bf[p].y);
                                    bf[p].z = fmaf(d.z, w.z, bf[p].z);
                                    bf[p].w = fmaf(d.w, w.w, bf[p].w);
                                }
                            }
                        const float4 bb = ld4(bdw + h * g.Ch + 4 * j);
#pragma unroll
                        for (int p = 0; p < PG; p++) { bf[p].x += bb.x; bf[p].y += bb.y; bf[p].z += bb.z; bf[p].w += bb.w; }
                    } else {
#pragma unroll
                        for (int p = 0; p < PG; p++) bf[p] = ld4(tile + base[p][0] + 4 * j);
                    }
                };
                auto a_frag = [&](int j, float4 (&av)[MTG]) {
#pragma unroll
                    for (int m = 0; m < MTG; m++) {
                        int mt = min(mt0 + m, g.MT - 1);
                        long ao = (((long)mt * nch + j) * 64 + lane) * 4;
                        av[m] = ALDS ? ld4(aL + ao) : ld4(a.w_pw + ao);
                    }
                };
                auto mfma_chunk = [&](const float4 (&av)[MTG], const float4 (&bf)[PG]) {
#pragma unroll
                    for (int m = 0; m < MTG; m++)
#pragma unroll
                        for (int p = 0; p < PG; p++) {
                            D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].x, bf[p].x, D[p][m], 0, 0, 0);
                            D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].y, bf[p].y, D[p][m], 0, 0, 0);
                            D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].z, bf[p].z, D[p][m], 0, 0, 0);
                            D[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m].w, bf[p].w, D[p][m], 0, 0, 0);
                        }
                };
                float4 bf[PG], av[MTG];
                dw_chunk(0, bf);
                a_frag(0, av);
                for (int j = 0; j + 1 < nch; j++) {
                    float4 bn[PG], an[MTG];
                    mfma_chunk(av, bf);
                    a_frag(j + 1, an);
                    dw_chunk(j + 1, bn);
#pragma unroll
                    for (int p = 0; p < PG; p++) bf[p] = bn[p];
#pragma unroll
                    for (int m = 0; m < MTG; m++) av[m] = an[m];
                    if (KS == 3) {
                        // program order: one MFMA (64 cycles of matrix pipe), then a slice of the next chunk's LDS reads + FMAs
                        constexpr int NM = 4 * MTG * PG, ND = 10 + 9 * PG + MTG, NV = 20 * PG;
#pragma unroll
                        for (int k = 0; k < NM; k++) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // MFMA
                            __builtin_amdgcn_sched_group_barrier(0x100, (ND + NM - 1) / NM, 0);  // DS read
                            __builtin_amdgcn_sched_group_barrier(0x002, (NV + NM - 1) / NM, 0);  // VALU
                        }
                    }
                }
                mfma_chunk(av, bf);
            }
            MI_STAMP(1)
            // ---- epilogue: lane holds pixel pl, output channels (mt*32 + 8*gq + 4*h .. +3) in D[..][4*gq .. 4*gq+3]
            float* op[PG];
            const float *rd[PG], *r0[PG], *r1[PG];
#pragma unroll
            for (int p = 0; p < PG; p++) {
                op[p] = a.out + (long)b * a.out_fs + ((long)(oys + rp[p]) * a.Wo + ox[p]) * a.Co;
                rd[p] = tile + wrap(ring0 + rp[p] + a.pt) * g.RS + (ox[p] + 1) * g.PS;                       // RES_DIRECT source pixel
                r0[p] = tile + wrap(ring0 + (S == 2 ? 2 * rp[p] : 0)) * g.RS + (2 * ox[p] + 1) * g.PS;       // RES_MAXPOOL rows
                r1[p] = tile + wrap(ring0 + (S == 2 ? 2 * rp[p] + 1 : 0)) * g.RS + (2 * ox[p] + 1) * g.PS;
            }
#pragma unroll
            for (int m = 0; m < MTG; m++) {
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int ch = (mt0 + m) * 32 + 8 * gq + 4 * h;
                    if (mt0 + m >= mt_hi || ch >= a.Co) continue;
                    const float4 bb = ld4(biasL + ch), al = ld4(alphaL + ch);
                    const bool has_res = rk != 0 && ch < a.ep.res_C;
#pragma unroll
                    for (int p = 0; p < PG; p++) {
                        if (!valid[p]) continue;
                        float4 v = make_float4(D[p][m][4 * gq] + bb.x, D[p][m][4 * gq + 1] + bb.y, D[p][m][4 * gq + 2] + bb.z, D[p][m][4 * gq + 3] + bb.w);
                        float4 rva = make_float4(0.f, 0.f, 0.f, 0.f);  // skip that joins behind the activation (Epilogue::res_after)
                        if (has_res) {
                            float4 rv;
                            if (rk == 1) rv = ld4(rd[p] + ch);
                            else if (rk == 2) rv = max4(ld4(r0[p] + ch), ld4(r0[p] + g.PS + ch), ld4(r1[p] + ch), ld4(r1[p] + g.PS + ch));
                            else if (SLOW) rv = rvq[SLOW ? p : 0][SLOW ? m : 0][gq];
                            else rv = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (a.ep.res_after) rva = rv;
                            else { v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                        }
                        // act(v) = min(max(v,0) + slope*min(v,0), hi): ReLU (slope 0), PReLU (alpha), none (1), ReLU6 (hi = 6)
                        v.x = fminf(fmaxf(v.x, 0.f) + al.x * fminf(v.x, 0.f), hi);
                        v.y = fminf(fmaxf(v.y, 0.f) + al.y * fminf(v.y, 0.f), hi);
                        v.z = fminf(fmaxf(v.z, 0.f) + al.z * fminf(v.z, 0.f), hi);
                        v.w = fminf(fmaxf(v.w, 0.f) + al.w * fminf(v.w, 0.f), hi);
                        v.x += rva.x; v.y += rva.y; v.z += rva.z; v.w += rva.w;
                        *reinterpret_cast<float4*>(op[p] + ch) = v;  // plain store: the 3 x 16 B pieces of a pixel merge in L2 (nontemporal stores measured 2.3x slower)
                    }
                }
            }
        }

        // ---- retire this step's dead rows: write the prefetched rows into their ring slots
        MI_STAMP(2)
        __syncthreads();
        MI_STAMP(3)
        if (more) {
            const int slot_lo = wrap(ring0 + (g.R - 1) * S + KS);  // first dead slot
#pragma unroll
            for (int k = 0; k < kPrefetch; k++) {
                const int r = pf_pack[k] >> 20;
                if (r < new_n) {
                    int sl = slot_lo + r;
                    sl = sl >= g.NR ? sl - g.NR : sl;
                    *reinterpret_cast<float4*>(tile + sl * g.RS + (pf_pack[k] & 0xFFFFF)) = pf[k];
                }
            }
        }
        ring0 = wrap(ring0 + g.R * S);
        MI_STAMP(4)
        __syncthreads();
        MI_STAMP(5)
    }
#ifdef MI_BLOCK_STAMPS
    if (g.stamps && lane == 0)
    {
        for (int k = 0; k < 6; k++) g.stamps[((long)blockIdx.x * 4 + wave) * 8 + k] = st_acc[k];
        g.stamps[((long)blockIdx.x * 4 + wave) * 8 + 6] = st_prologue;
    }
#endif
}

bool make_geom_pg(const BlockArgs& a, int PG, BlockGeom* out) {
    BlockGeom g{};
    g.stamps = nullptr;
    g.PG = PG;
    const int S = a.sh, KS = a.has_dw ? 3 : 1;
    if (a.C % 4 || a.Co % 4 || a.sh != a.sw || (S != 1 && S != 2)) return false;
    if (!a.has_dw && S != 1) return false;
    g.Cp = (a.C + 7) & ~7;
    g.Ch = g.Cp / 2;
    g.C4 = a.C / 4;
    g.PS = g.Cp + 4;
    g.RS = (a.W + 2) * g.PS;
    g.MT = (a.Co + 31) / 32;
    const long a_bytes = (long)g.MT * 32 * g.Cp * 4;
    static const int alds_max = getenv("MI_BLOCK_ALDS") ? atoi(getenv("MI_BLOCK_ALDS")) : kALdsMax;  // tuning aid
    g.a_lds = a_bytes <= alds_max;
    const int fixed = (a.has_dw ? 10 * g.Cp * 4 : 0) + (g.a_lds ? (int)a_bytes : 0) + g.MT * 32 * 8 + 64;
    auto lds_for = [&](int R) { return ((R - 1) * S + KS) * g.RS * 4 + fixed; };
    auto pf_ok = [&](int R) { return (long)R * S * a.W * g.C4 <= (long)kPrefetch * 256; };
    // ... of a handful of frames (a single-image call of full_range, face_detection.rs:205: 1 - 2 such steps per frame on 256 CUs) only 32 pixels'
    // worth: a step's MFMA time goes with its 32-pixel groups, and the shorter bands are more workgroups (full_range at one frame: its 19
    // block_kernel launches 0.260 -> 0.226 ms; 16 pixels: 0.217, 64: 0.242).  The arithmetic of a pixel does not depend on the band it is in.
    static const int rcap_px = getenv("MI_BLOCK_RCAP_PX") ? atoi(getenv("MI_BLOCK_RCAP_PX")) : 32;  // tuning aid (0: off)
    int Rfull = std::max(1, std::min(a.Ho, (128 * PG) / a.Wo));  // rows that fill every lane of the 4 waves
    if (rcap_px > 0 && (long)a.B * ((a.Ho + Rfull - 1) / Rfull) * 8 <= cu_count()) Rfull = std::max(1, std::min(Rfull, rcap_px / a.Wo));
    int R = Rfull;
    while (R > 1 && (lds_for(R) > lds_budget() || !pf_ok(R))) R--;
    if (R * a.Wo * 4 < 128 * PG * 3) {  // step under 75% full: small, channel-heavy layer -> spend the whole LDS of a CU on one workgroup
        int R2 = Rfull;
        while (R2 > 1 && (lds_for(R2) > 150 * 1024 || !pf_ok(R2))) R2--;
        if (R2 > R) R = R2;
    }
    if (lds_for(R) > 150 * 1024 || !pf_ok(R)) return false;
    if (g.RS >= (1 << 20)) return false;
    // Small, channel-heavy layers (16x16x96 -> 8x8, 32x32x48 -> 16x16): the prefetch registers cap R at 2 rows, i.e. 16-32
    // of the 128 pixel slots of a step.  A band that is finished in ONE step needs no prefetch at all, so take as many rows
    // as the LDS holds and make each band exactly one such step.
    bool single_step = false;
    {
        int R1 = Rfull;
        while (R1 > 1 && lds_for(R1) > 150 * 1024) R1--;
        if (a.Ho * a.Wo <= 256 && R1 >= 2 * R && lds_for(R1) <= 150 * 1024) { R = R1; single_step = true; }
    }
    g.R = R;
    g.NR = (R - 1) * S + KS;
    int off = g.NR * g.RS;
    g.off_wdw = off; off += a.has_dw ? 9 * g.Cp : 0;
    g.off_bdw = off; off += a.has_dw ? g.Cp : 0;
    off = (off + 3) & ~3;
    g.off_a = off; off += g.a_lds ? g.MT * 32 * g.Cp : 0;
    g.off_bias = off; off += g.MT * 32;
    g.off_alpha = off; off += g.MT * 32;
    g.lds_bytes = off * 4;
    // bands: about one resident wave of workgroups over the chip (prologue + halo paid once per band)
    static const int max_per_cu = getenv("MI_BLOCK_PERCU") ? atoi(getenv("MI_BLOCK_PERCU")) : 2;  // tuning aid
    const int per_cu = std::max(1, std::min(max_per_cu, (160 * 1024) / g.lds_bytes));
    const int max_bands = (a.Ho + R - 1) / R;
    int bands = std::min(max_bands, std::max(1, (cu_count() * per_cu + a.B / 2) / std::max(1, a.B)));
    g.band = single_step ? R : ((a.Ho + bands - 1) / bands + R - 1) / R * R;
    g.bands = (a.Ho + g.band - 1) / g.band;
    g.nsplit = 1; g.wmod = 4; g.mt_per = g.MT;
    {
        static const int no_split = getenv("MI_BLOCK_NOSPLIT") ? atoi(getenv("MI_BLOCK_NOSPLIT")) : 0;  // tuning aid
        const long wgs = (long)a.B * g.bands;
        const int groups = (std::min(R, a.Ho) * a.Wo + 32 * PG - 1) / (32 * PG);  // waves of a step that have pixels
        if (!no_split && g.MT >= 2 && wgs * 2 <= cu_count()) {
            const int wmod = groups <= 1 ? 1 : (groups == 2 ? 2 : 4), wdiv = 4 / wmod;
            int nsplit = (int)std::min<long>((2L * cu_count() + wgs - 1) / wgs, (g.MT + wdiv - 1) / wdiv);
            nsplit = std::max(1, nsplit);
            const int per = (g.MT + nsplit * wdiv - 1) / (nsplit * wdiv);
            nsplit = (g.MT + per * wdiv - 1) / (per * wdiv);
            g.nsplit = nsplit; g.wmod = wmod; g.mt_per = per;
        }
    }
    // skip connection straight from the ring when it is the block's own input tensor
    g.res_lds = 0;
    if (a.ep.res == a.in && a.ep.res_fs == a.in_fs && a.ep.res_C == a.C) {
        if (a.ep.res_mode == RES_DIRECT && S == 1) g.res_lds = 1;
        if (a.ep.res_mode == RES_MAXPOOL && S == 2 && a.has_dw && a.pt == 0 && a.pl == 0 && a.H == 2 * a.Ho && a.W == 2 * a.Wo) g.res_lds = 1;
    }
    if (a.ep.res_mode != RES_NONE && a.ep.res_C % 4) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!aligned16(a.in) || !aligned16(a.out) || (a.in_fs & 3) || (a.out_fs & 3)) return false;
    if (a.ep.res && (!aligned16(a.ep.res) || (a.ep.res_fs & 3))) return false;
    *out = g;
    return true;
}

// Two 32-pixel groups per wave halve the depthwise-weight reads per pixel, but only pay when a step has >= 192 pixels.
bool make_geom(const BlockArgs& a, BlockGeom* out) {
    static const int forced = getenv("MI_BLOCK_PG") ? atoi(getenv("MI_BLOCK_PG")) : 0;  // tuning aid
    const int MT = (a.Co + 31) / 32;
    BlockGeom g2;
    // ... and not when the skip comes from another tensor: those variants carry the prefetched skip quads as well, two groups per wave
    // push them past 256 registers (one workgroup per CU); measured on full_range's 96x96x8 -> 32 blocks: 0.130 -> 0.103 ms
    const bool foreign_skip = a.ep.res_mode != RES_NONE && !(a.ep.res == a.in && a.ep.res_fs == a.in_fs && a.ep.res_C == a.C);
    if (forced != 1 && (forced == 2 || !foreign_skip) && MT <= 2 && make_geom_pg(a, 2, &g2) && g2.R * a.Wo >= 192) {
        *out = g2;
        return true;
    }
    return make_geom_pg(a, 1, out);
}

template <int MTG, int S, int KS, int PG, bool SLOW, int CPT, bool ALDS>
int launch_inst4(const BlockArgs& a, const BlockGeom& g, hipStream_t s) {
    auto kern = block_kernel<MTG, S, KS, PG, SLOW, CPT, ALDS>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(kern, dim3((unsigned)(a.B * g.bands * g.nsplit)), dim3(256), (size_t)g.lds_bytes, s, a, g);
}

template <int MTG, int S, int KS, int PG, bool SLOW, int CPT>
int launch_inst3(const BlockArgs& a, const BlockGeom& g, hipStream_t s) {
    if constexpr (SLOW) {
        if (!g.a_lds) return launch_inst4<MTG, S, KS, PG, SLOW, CPT, false>(a, g, s);
    }
    return launch_inst4<MTG, S, KS, PG, SLOW, CPT, true>(a, g, s);
}

template <int MTG, int S, int KS, int PG, bool SLOW>
int launch_inst2(const BlockArgs& a, const BlockGeom& g, hipStream_t s) {
    if constexpr (!SLOW && KS == 3 && MTG <= 2) {  // hot BackCamera shapes get compile-time channel counts
        if (g.Cp == 24) return launch_inst3<MTG, S, KS, PG, SLOW, 24>(a, g, s);
        if (g.Cp == 48) return launch_inst3<MTG, S, KS, PG, SLOW, 48>(a, g, s);
    }
    return launch_inst3<MTG, S, KS, PG, SLOW, 0>(a, g, s);
}

template <int MTG, int S, int KS, int PG>
int launch_inst(const BlockArgs& a, const BlockGeom& g, hipStream_t s) {
    const bool slow = !g.a_lds || (a.ep.res_mode != RES_NONE && !g.res_lds);
    return slow ? launch_inst2<MTG, S, KS, PG, true>(a, g, s) : launch_inst2<MTG, S, KS, PG, false>(a, g, s);
}

}  // namespace

// Pointwise weights are packed on the host straight into MFMA A-fragment order (engine.cpp).
void block_weight_dims(int C, int Co, int* Cp, int* Cop) {
    *Cp = (C + 7) & ~7;
    *Cop = (Co + 31) / 32 * 32;
}

bool block_kernel_supports(const BlockArgs& a) {
    BlockGeom g;
    return make_geom(a, &g);
}

const char* block_kernel_label(const BlockArgs& a, char* buf, size_t cap) {
    BlockGeom g;
    if (!make_geom(a, &g)) return "block_kernel<?>";
    const bool slow = !g.a_lds || (a.ep.res_mode != RES_NONE && !g.res_lds);
    snprintf(buf, cap, "block_kernel<%d,%d,%d,%d,%d>", std::min(4, g.mt_per), a.has_dw ? a.sh : 1, a.has_dw ? 3 : 1, g.PG, slow ? 1 : 0);
    return buf;
}

#ifdef MI_BLOCK_STAMPS
unsigned long long* g_stamp_ptr = nullptr;
#endif

int launch_block(const BlockArgs& a, void* stream) {
    BlockGeom g;
    if (!make_geom(a, &g)) return (int)hipErrorInvalidValue;
#ifdef MI_BLOCK_STAMPS
    g.stamps = g_stamp_ptr;
#endif
    hipStream_t s = (hipStream_t)stream;
    const int MTG = std::min(4, g.mt_per);
    const int S = a.sh, K = a.has_dw ? 3 : 1, PG = g.PG;
#define MI_BLOCK_CASE(M, SS, KK, P) \
    if (MTG == M && S == SS && K == KK && PG == P) return launch_inst<M, SS, KK, P>(a, g, s);
    MI_BLOCK_CASE(1, 1, 3, 2) MI_BLOCK_CASE(2, 1, 3, 2) MI_BLOCK_CASE(1, 1, 3, 1) MI_BLOCK_CASE(2, 1, 3, 1) MI_BLOCK_CASE(3, 1, 3, 1) MI_BLOCK_CASE(4, 1, 3, 1)
    MI_BLOCK_CASE(1, 2, 3, 2) MI_BLOCK_CASE(2, 2, 3, 2) MI_BLOCK_CASE(1, 2, 3, 1) MI_BLOCK_CASE(2, 2, 3, 1) MI_BLOCK_CASE(3, 2, 3, 1) MI_BLOCK_CASE(4, 2, 3, 1)
    MI_BLOCK_CASE(1, 1, 1, 2) MI_BLOCK_CASE(2, 1, 1, 2) MI_BLOCK_CASE(1, 1, 1, 1) MI_BLOCK_CASE(2, 1, 1, 1) MI_BLOCK_CASE(3, 1, 1, 1) MI_BLOCK_CASE(4, 1, 1, 1)
#undef MI_BLOCK_CASE
    return (int)hipErrorInvalidValue;
}

}  // namespace mi
