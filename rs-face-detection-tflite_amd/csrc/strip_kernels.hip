// strip_kernels.hip — register-resident fused BlazeBlocks for the narrow layers (C in {16, 24, 32}), one block per launch
// (strip_kernel) or a run of 2..4 blocks per launch with the rows handed from block to block through LDS
// (strip_pipe_kernel), optionally ended by the stride-2 block that halves the resolution.
//
//   out = act( PW1x1( DW3x3(in) + b_dw ) + b_pw + in )
//
// Same operator chain as block_kernels.hip (DEPTHWISE_CONV_2D -> CONV_2D 1x1 -> ADD -> RELU/PRELU behind
// `interpreter.invoke()`, /root/reference/src/face_detection_lite/face_detection.rs:235; the 24-channel blocks at 128^2
// and 64^2 are 16 of BackCamera's 41 block-fused launches and two thirds of its time), different mapping to the machine.
// The block kernel keeps rows in a workgroup-shared LDS ring and re-reads the 3x3 window and the depthwise weights from
// LDS per lane: ~100 ds_read_b128 per 64 pixels, which keeps the CU's LDS pipe ~60 % busy at 2 waves/SIMD — its wall.
// Here:
//   * one WAVE owns a strip of 64 pixel columns and walks down a band of rows; lane = pixel.
//   * a row of the strip (64 + 2 halo pixels, all channels) arrives by LDS-DMA (1 KiB contiguous per instruction) in a
//     wave-private buffer; the lane reads its own pixel and its left / right neighbours from it: 3 ds_read_b128 per
//     channel quad and row — each input element is read from LDS three times instead of nine — and vertical reuse is in
//     registers: the partial output rows an input row contributes to (ky = 2, 1, 0) stay in VGPR accumulators.
//   * depthwise AND pointwise weights are wave-uniform in this layout: they arrive through the scalar cache as SGPR-pair
//     operands of v_pk_fma_f32 (two channels per VALU lane-op), no LDS or VGPR cost.  The 1x1 conv is NOT on the matrix
//     cores: on gfx950 the f32 MFMA has the packed-FMA peak, does not overlap VALU work, and M = 32 wastes a quarter of it
//     on 24 channels (the MFMA variant of this kernel, with v_permlane32_swap layout changes, is in the git history).
//   * + bias (folded W b_dw + b_pw) + skip (the centre input row, still in VGPRs) -> activation -> transposed back
//     through LDS so that every store instruction writes 1 KiB of consecutive bytes (full 128-byte lines instead of
//     16-byte pieces that have to meet in L2).
// Exact f32 throughout; results match the block kernel to reassociation of the 3x3 sum and the folded depthwise bias.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"
#include "launch.hpp"

namespace mi {

namespace {

typedef float sf32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float __attribute__((address_space(4))) cfloat;

struct StripArgs {
    const float* in;
    float* out;
    const float* consts;   // strip_pack_consts() blob
    long in_fs, out_fs;
    int B, H, W;
    int strips, bands, band_rows;
    int has_res;
    float hi;              // upper clamp of the activation (6 for ReLU6, +inf otherwise)
    unsigned long long* stamps;  // diagnostic builds only (MI_STRIP_STAMPS): 8 accumulators per wave
};

template <int CQ>
struct SK {
    static constexpr int C = 4 * CQ;          // channels
    static constexpr int K2 = C / 2;          // MFMA k-steps (2 channels each)
    static constexpr int NF = 66 * CQ;        // float4s of one staged input row (64 pixels + 2 halo pixels)
    static constexpr int NL = (NF + 63) / 64; // LDS-DMA instructions per row (the last one partial)
    static constexpr int BUF_F = (66 * C + 63) / 64 * 64;  // floats per row buffer (pixel-major, unpadded: DMA image is lane-linear)
    static constexpr int WAVE_F = 2 * BUF_F;  // two row buffers per wave: row r in use, row r+1 in flight
    // constants blob (floats): per input-channel pair a stage record of ST_F floats = depthwise taps [9][2] at 0, pointwise
    // columns W[0..C)[2st] at 32 and W[0..C)[2st+1] at 32 + C; then the pointwise bias (+ W b_dw) and the negative slopes
    static constexpr int ST_F = 32 + 2 * C + (16 - (2 * C) % 16) % 16;
    // ... then (strip_pipe2m_kernel) the pointwise filter as A operands of v_mfma_f32_4x4x1_16b_f32: NA registers of 64 lanes; the 4 x 1
    // block (input channel k, output channels 4t .. 4t+3) is block n % 16 (lanes 4 (n % 16) + i) of register n / 16, n = k CQ + t
    static constexpr int NA = (C * CQ + 15) / 16;
    static constexpr int OFF_DW = 0, OFF_BIAS = K2 * ST_F, OFF_SLOPE = OFF_BIAS + 32, OFF_A = OFF_SLOPE + 32, TOTAL = OFF_A + 64 * NA;
};

__device__ __forceinline__ float4 sld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void sst4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// LDS traffic between the lanes of ONE wave: the hardware executes a wave's LDS instructions in order; this only has to
// stop the compiler from moving accesses across the hand-over.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ v2f pkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = imm[15:14]:imm[3:0], expcnt and lgkmcnt left at their maxima)
template <int N>
__device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70); }

// One input row r (pixel-major image in LDS; `me` = this lane's left neighbour in it): its ky = 2 / 1 / 0 taps go to the partial output rows r-1 / r / r+1 (aPN on
// entry / aC / aPN on exit: the slot of the row that finishes here is re-used for the row that starts here); the finished
// row r-1 goes straight into the pointwise accumulators.  Everything runs on the VALU as
// v_pk_fma_f32 with SGPR-pair weights: on this part the f32 MFMA does not overlap VALU work (measured: 2 x
// v_mfma_f32_32x32x2_f32 + n x v_fma_f32 costs 129 + 4.5 n cycles, at one or two waves per SIMD), it runs at the
// packed-FMA rate, and M = 32 would waste a quarter of it on 24 output channels.
// Work is cut into 2*CQ stages of one input-channel PAIR.  Scalar loads return out of order, so every wait is
// lgkmcnt(0); each stage therefore waits twice and issues the NEXT chunk's loads right after each wait:
//   wait | load PW weights of the pair's 2nd channel | depthwise (9 pk) + pointwise of the 1st channel (CQ*2 pk)
//   wait | load DW + 1st-channel PW weights of the next stage, next quad's pixels | pointwise of the 2nd channel
template <int CQ, bool EMIT>
__device__ __forceinline__ void strip_row(const float* me, const cfloat* cst, bool has_res, v2f (&aPN)[CQ][2], v2f (&aC)[CQ][2],
                                          const float4 (&xres)[CQ], float4 (&xcur)[CQ], v2f (&oacc)[2 * CQ]) {
    using K = SK<CQ>;
    constexpr int C = K::C;
    float wd[18], wp0[C], wp1[C];
    float4 xbuf[2][3];
    // scalar loads of one group share one opaque copy of the blob pointer: opaque so that the loads stay inside the row
    // loop, just ahead of their use (hoisted, they spill SGPRs); the record offsets stay immediates of the s_load
    auto load_first = [&](int st) {  // what a stage needs at its first wait: depthwise taps + pointwise column of channel 2st
        const cfloat* p = cst;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < 18; i++) wd[i] = p[K::OFF_DW + st * K::ST_F + i];
        if (EMIT) {
#pragma unroll
            for (int i = 0; i < C; i++) wp0[i] = p[K::OFF_DW + st * K::ST_F + 32 + i];
        }
    };
    auto load_second = [&](int st) {  // pointwise column of channel 2st + 1
        const cfloat* p = cst;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < C; i++) wp1[i] = p[K::OFF_DW + st * K::ST_F + 32 + C + i];
    };
    auto load_x = [&](int q, float4 (&x)[3]) {
#pragma unroll
        for (int kx = 0; kx < 3; kx++) x[kx] = sld4(me + kx * C + 4 * q);
    };
    if constexpr (EMIT) {  // accumulators start from skip + bias (bias = b_pw + W b_dw): one v_pk_add_f32 with an SGPR-pair operand each
        const cfloat* bp = cst + K::OFF_BIAS;
        asm volatile("" : "+s"(bp));
        if (has_res) {  // wave-uniform branch
#pragma unroll
            for (int q = 0; q < CQ; q++) {
                oacc[2 * q] = v2f{xres[q].x, xres[q].y} + v2f{bp[4 * q], bp[4 * q + 1]};
                oacc[2 * q + 1] = v2f{xres[q].z, xres[q].w} + v2f{bp[4 * q + 2], bp[4 * q + 3]};
            }
        } else {
#pragma unroll
            for (int q = 0; q < CQ; q++) {
                oacc[2 * q] = v2f{bp[4 * q], bp[4 * q + 1]};
                oacc[2 * q + 1] = v2f{bp[4 * q + 2], bp[4 * q + 3]};
            }
        }
#pragma unroll
        for (int o = 0; o < 2 * CQ; o++) asm volatile("" : "+v"(oacc[o]));  // keeps the two arms from being merged into selects
    }
    load_first(0);
    load_x(0, xbuf[0]);
#pragma unroll
    for (int st = 0; st < 2 * CQ; st++) {
        const int q = st >> 1, h = st & 1;
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        if (EMIT) load_second(st);
        __builtin_amdgcn_sched_barrier(0);
        const float4 (&x)[3] = xbuf[q & 1];
        if (h == 0) xcur[q] = x[1];
        v2f t[3];
#pragma unroll
        for (int kx = 0; kx < 3; kx++) t[kx] = h == 0 ? v2f{x[kx].x, x[kx].y} : v2f{x[kx].z, x[kx].w};
        auto wv = [&](int tap) { return v2f{wd[2 * tap], wd[2 * tap + 1]}; };
        v2f n = t[0] * wv(0), c = aC[q][h], pch = aPN[q][h];  // the depthwise bias is folded into the pointwise bias on the host
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            if (EMIT) pch = pkfma(t[kx], wv(6 + kx), pch);
            if (kx) n = pkfma(t[kx], wv(kx), n);
            c = pkfma(t[kx], wv(3 + kx), c);
        }
        aC[q][h] = c;
        aPN[q][h] = n;  // the slot of the row just finished now starts the row after next
        // pin the partial rows here: left alone, LLVM sinks these updates past the loop-exit test into the next row's
        // block, across the scheduling fences, which keeps every stage's weights alive (SGPR spills)
        asm volatile("" : "+v"(aC[q][h]), "+v"(aPN[q][h]));
        if constexpr (EMIT) {
#pragma unroll
            for (int o = 0; o < 2 * CQ; o++) oacc[o] = pkfma(v2f{pch.x, pch.x}, v2f{wp0[2 * o], wp0[2 * o + 1]}, oacc[o]);
            asm volatile("" : "+v"(pch));
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (st + 1 < 2 * CQ) {
            load_first(st + 1);
            if (h == 1) load_x(q + 1, xbuf[(q + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (EMIT) {
#pragma unroll
            for (int o = 0; o < 2 * CQ; o++) {
                oacc[o] = pkfma(v2f{pch.y, pch.y}, v2f{wp1[2 * o], wp1[2 * o + 1]}, oacc[o]);
                asm volatile("" : "+v"(oacc[o]));  // same pin as above: a caller that emits conditionally must not pull the pointwise FMAs into its branch
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Two input rows r, r+1 per call (images me0, me1): every scalar-loaded weight then feeds two rows' worth of packed
// FMAs, which halves the scalar-cache round trips, waits and LDS hand-overs per FMA (with one row per call the waits are
// only half covered: doubling the FMA work per stage costs 1.58x, not 2x).  Entering: aA = partial row r-1 (rows r-2,
// r-1 in), aB = partial row r (row r-1 in).  The call finishes rows r-1 and r (into oacc0 / oacc1, which start from
// bias + skip: xres0 = centre pixels of row r-1; row r's are added at the end), and leaves aA = partial row r+1, aB =
// partial row r+2, xnext = centre pixels of row r+1.
template <int CQ>
__device__ __forceinline__ void strip_row2(const float* me0, const float* me1, const cfloat* cst, bool has_res, v2f (&aA)[CQ][2], v2f (&aB)[CQ][2],
                                           const float4 (&xres0)[CQ], float4 (&xnext)[CQ], v2f (&oacc0)[2 * CQ], v2f (&oacc1)[2 * CQ]) {
    using K = SK<CQ>;
    constexpr int C = K::C;
    float wd[18], wp0[C], wp1[C];
    float4 xbuf[2][2][3];
#ifdef MI_ABL_NOSMEM  // timing ablation (development harness only): weights come from one scalar register, no scalar loads
    float abl_w = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, aA[0][0].x)));
    auto load_first = [&](int st) {
        asm volatile("" : "+s"(abl_w));
#pragma unroll
        for (int i = 0; i < 18; i++) wd[i] = abl_w;
#pragma unroll
        for (int i = 0; i < C; i++) wp0[i] = abl_w;
    };
    auto load_second = [&](int st) {
        asm volatile("" : "+s"(abl_w));
#pragma unroll
        for (int i = 0; i < C; i++) wp1[i] = abl_w;
    };
#else
    auto load_first = [&](int st) {
        const cfloat* p = cst;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < 18; i++) wd[i] = p[K::OFF_DW + st * K::ST_F + i];
#pragma unroll
        for (int i = 0; i < C; i++) wp0[i] = p[K::OFF_DW + st * K::ST_F + 32 + i];
    };
    auto load_second = [&](int st) {
        const cfloat* p = cst;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < C; i++) wp1[i] = p[K::OFF_DW + st * K::ST_F + 32 + C + i];
    };
#endif
    auto load_x = [&](int q, float4 (&x)[2][3]) {
#ifdef MI_ABL_NOLDSX  // timing ablation: pixels from registers instead of LDS
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            x[0][kx] = xres0[(q + kx) % CQ];
            x[1][kx] = xres0[(q + kx + 1) % CQ];
            asm volatile("" : "+v"(x[0][kx].x), "+v"(x[0][kx].y), "+v"(x[0][kx].z), "+v"(x[0][kx].w), "+v"(x[1][kx].x), "+v"(x[1][kx].y), "+v"(x[1][kx].z), "+v"(x[1][kx].w));
        }
#else
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            x[0][kx] = sld4(me0 + kx * C + 4 * q);
            x[1][kx] = sld4(me1 + kx * C + 4 * q);
        }
#endif
    };
    {   // both rows start from skip + bias, in that order (the same arithmetic as strip_row whichever row of a pair a row is)
        const cfloat* bp = cst + K::OFF_BIAS;
        asm volatile("" : "+s"(bp));
        float4 xm[CQ];
#pragma unroll
        for (int q = 0; q < CQ; q++) xm[q] = has_res ? sld4(me0 + C + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);  // centre pixels of row r
        if (has_res) {  // wave-uniform branch
#pragma unroll
            for (int q = 0; q < CQ; q++) {
                oacc0[2 * q] = v2f{xres0[q].x, xres0[q].y} + v2f{bp[4 * q], bp[4 * q + 1]};
                oacc0[2 * q + 1] = v2f{xres0[q].z, xres0[q].w} + v2f{bp[4 * q + 2], bp[4 * q + 3]};
                oacc1[2 * q] = v2f{xm[q].x, xm[q].y} + v2f{bp[4 * q], bp[4 * q + 1]};
                oacc1[2 * q + 1] = v2f{xm[q].z, xm[q].w} + v2f{bp[4 * q + 2], bp[4 * q + 3]};
            }
        } else {
#pragma unroll
            for (int q = 0; q < CQ; q++) {
                const v2f b0 = v2f{bp[4 * q], bp[4 * q + 1]}, b1 = v2f{bp[4 * q + 2], bp[4 * q + 3]};
                oacc0[2 * q] = b0;
                oacc1[2 * q] = b0;
                oacc0[2 * q + 1] = b1;
                oacc1[2 * q + 1] = b1;
            }
        }
#pragma unroll
        for (int o = 0; o < 2 * CQ; o++) asm volatile("" : "+v"(oacc0[o]), "+v"(oacc1[o]));
    }
    load_first(0);
    load_x(0, xbuf[0]);
#pragma unroll
    for (int st = 0; st < 2 * CQ; st++) {
        const int q = st >> 1, h = st & 1;
#ifndef MI_ABL_NOWAIT
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#endif
        load_second(st);
        __builtin_amdgcn_sched_barrier(0);
        const float4 (&x)[2][3] = xbuf[q & 1];
        if (h == 0) xnext[q] = x[1][1];
        v2f t0[3], t1[3];
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            t0[kx] = h == 0 ? v2f{x[0][kx].x, x[0][kx].y} : v2f{x[0][kx].z, x[0][kx].w};
            t1[kx] = h == 0 ? v2f{x[1][kx].x, x[1][kx].y} : v2f{x[1][kx].z, x[1][kx].w};
        }
        auto wv = [&](int tap) { return v2f{wd[2 * tap], wd[2 * tap + 1]}; };
        v2f p0 = aA[q][h], p1 = aB[q][h], nA = t0[0] * wv(0), nB = t1[0] * wv(0);
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            p0 = pkfma(t0[kx], wv(6 + kx), p0);
            p1 = pkfma(t0[kx], wv(3 + kx), p1);
            if (kx) nA = pkfma(t0[kx], wv(kx), nA);
        }
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            p1 = pkfma(t1[kx], wv(6 + kx), p1);
            nA = pkfma(t1[kx], wv(3 + kx), nA);
            if (kx) nB = pkfma(t1[kx], wv(kx), nB);
        }
        aA[q][h] = nA;
        aB[q][h] = nB;
        asm volatile("" : "+v"(aA[q][h]), "+v"(aB[q][h]));
#pragma unroll
        for (int o = 0; o < 2 * CQ; o++) {
            oacc0[o] = pkfma(v2f{p0.x, p0.x}, v2f{wp0[2 * o], wp0[2 * o + 1]}, oacc0[o]);
            oacc1[o] = pkfma(v2f{p1.x, p1.x}, v2f{wp0[2 * o], wp0[2 * o + 1]}, oacc1[o]);
        }
        asm volatile("" : "+v"(p0), "+v"(p1));
        __builtin_amdgcn_sched_barrier(0);
#ifndef MI_ABL_NOWAIT
        __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
        if (st + 1 < 2 * CQ) {
            load_first(st + 1);
            if (h == 1) load_x(q + 1, xbuf[(q + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int o = 0; o < 2 * CQ; o++) {
            oacc0[o] = pkfma(v2f{p0.y, p0.y}, v2f{wp1[2 * o], wp1[2 * o + 1]}, oacc0[o]);
            oacc1[o] = pkfma(v2f{p1.y, p1.y}, v2f{wp1[2 * o], wp1[2 * o + 1]}, oacc1[o]);
            asm volatile("" : "+v"(oacc0[o]), "+v"(oacc1[o]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// max(x, 0) as ONE v_max_f32: fmaxf() first canonicalises an operand it cannot prove to be quiet (the accumulators pass through empty
// asm statements), two instructions per value
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// activation of a finished output row (lane = pixel, channel quads)
template <int CQ, bool RELU>
__device__ __forceinline__ void strip_act(const v2f (&oacc)[2 * CQ], const cfloat* slopes, float hi, float4 (&o)[CQ]) {
#pragma unroll
    for (int q = 0; q < CQ; q++) {
        o[q] = make_float4(oacc[2 * q].x, oacc[2 * q].y, oacc[2 * q + 1].x, oacc[2 * q + 1].y);
        if (RELU) {
            o[q].x = relu1(o[q].x); o[q].y = relu1(o[q].y); o[q].z = relu1(o[q].z); o[q].w = relu1(o[q].w);
        } else {
            // act(v) = min(max(v,0) + slope*min(v,0), hi): ReLU (slope 0), PReLU (alpha), none (1), ReLU6 (hi = 6)
            const cfloat* sq = slopes + 4 * q;
            o[q].x = fminf(fmaxf(o[q].x, 0.f) + sq[0] * fminf(o[q].x, 0.f), hi);
            o[q].y = fminf(fmaxf(o[q].y, 0.f) + sq[1] * fminf(o[q].y, 0.f), hi);
            o[q].z = fminf(fmaxf(o[q].z, 0.f) + sq[2] * fminf(o[q].z, 0.f), hi);
            o[q].w = fminf(fmaxf(o[q].w, 0.f) + sq[3] * fminf(o[q].w, 0.f), hi);
        }
    }
}

// D = A[block `blk` of `a`, broadcast to all 16 blocks] x B + C (the builtin wants the block as a literal; `blk` is a constant once
// the caller's loops are unrolled, and the switch folds away)
__device__ __forceinline__ v4f mfma4_bcast(float a, float b, v4f c, int blk) {
    switch (blk) {
#define MI_M4(N) case N: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, N, 0);
        MI_M4(0) MI_M4(1) MI_M4(2) MI_M4(3) MI_M4(4) MI_M4(5) MI_M4(6) MI_M4(7) MI_M4(8) MI_M4(9) MI_M4(10) MI_M4(11) MI_M4(12) MI_M4(13) MI_M4(14)
#undef MI_M4
        default: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 15, 0);
    }
}

// strip_row2 with the pointwise conv on the matrix cores, in the layout the depthwise stage already has (lane = pixel).
// v_mfma_f32_4x4x1_16b_f32 is 16 independent (4 x 1)(1 x 4) outer products: block b = lanes 4b .. 4b+3, D[i][lane] = A[block, i] *
// B[lane] + C[i][lane].  With B = the finished depthwise value of input channel k (one register, lane = pixel) and A = four
// output-channel weights W[4t .. 4t+3][k], broadcast from ONE block of a weight register to all sixteen (cbsz = 4, abid =
// block), D = four output channels of every pixel, lane = pixel, register = channel: the accumulator layout of the packed-FMA
// form, no transposition, no padding (24 = 6 x 4), the same k-sequential f32 fma chain per output (bit-identical).  The
// whole pointwise filter is NA = 9 registers per wave (24 x 24), loaded once per launch: no weight stream; what is left in
// the scalar stream are the 18 depthwise taps of a channel pair, fetched a full stage ahead (one wait per stage).
//   me0 / me1: this lane's left neighbour in the images of input rows c0, c0 + 1; ps: pixel stride of those images (floats).
//   cb: pointwise bias (+ W b_dw), wd[0]: the taps of stage 0 — both asked for by the caller before the step's barrier.
//   onext: bias + skip of the first row this call finishes (= centre pixels of the previous call's second row, put there by that
//   call: they are the C operand of this call's first MFMAs, whose D is oacc0 — no copy, no separate add); the call leaves the next one's.
template <int CQ>
__device__ __forceinline__ void strip_row2m(const float* me0, const float* me1, int ps, const cfloat* cst, bool has_res, v2f (&aA)[CQ][2], v2f (&aB)[CQ][2],
                                            float4 (&onext)[CQ], v4f (&oacc0)[CQ], v4f (&oacc1)[CQ], const float (&wa)[SK<CQ>::NA],
                                            const float (&cb)[4 * CQ], float (&wd)[2][18]) {
    using K = SK<CQ>;
    float4 xbuf[2][2][3];
    const float* r0[3] = {me0, me0 + ps, me0 + 2 * ps};
    const float* r1[3] = {me1, me1 + ps, me1 + 2 * ps};
#ifdef MI_ABL_M_NOSMEM  // timing ablations (development harness only; the ablated kernels compute wrong values)
    float abl_w = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, aA[0][0].x)));
    auto load_taps = [&](int st) {
        asm volatile("" : "+s"(abl_w));
#pragma unroll
        for (int i = 0; i < 18; i++) wd[st & 1][i] = abl_w;
    };
#else
    auto load_taps = [&](int st) {
        const cfloat* p = cst;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < 18; i++) wd[st & 1][i] = p[K::OFF_DW + st * K::ST_F + i];
    };
#endif
    auto load_x = [&](int q, float4 (&x)[2][3]) {
#ifdef MI_ABL_M_NOLDS
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            x[0][kx] = onext[(q + kx) % CQ];
            x[1][kx] = onext[(q + kx + 1) % CQ];
            asm volatile("" : "+v"(x[0][kx].x), "+v"(x[0][kx].y), "+v"(x[0][kx].z), "+v"(x[0][kx].w), "+v"(x[1][kx].x), "+v"(x[1][kx].y), "+v"(x[1][kx].z), "+v"(x[1][kx].w));
        }
#else
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            x[0][kx] = sld4(r0[kx] + 4 * q);
            x[1][kx] = sld4(r1[kx] + 4 * q);
        }
#endif
    };
    auto bias4 = [&](int q) { return v4f{cb[4 * q], cb[4 * q + 1], cb[4 * q + 2], cb[4 * q + 3]}; };
    // second finished row: bias + skip (centre pixels of row c0); quad 0's window is asked for behind them and lands while these are added
    if (has_res) {  // wave-uniform branch
        float4 xm[CQ];
#pragma unroll
        for (int q = 0; q < CQ; q++) xm[q] = sld4(r0[1] + 4 * q);
        load_x(0, xbuf[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < CQ; q++) {
            oacc1[q] = v4f{xm[q].x, xm[q].y, xm[q].z, xm[q].w} + bias4(q);
            asm volatile("" : "+v"(oacc1[q]));
        }
    } else {
        load_x(0, xbuf[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < CQ; q++) {
            oacc1[q] = bias4(q);
            asm volatile("" : "+v"(oacc1[q]));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 2 * CQ; st++) {
        const int q = st >> 1, h = st & 1;
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this stage's taps (asked for a stage ago) and pixels
        if (st + 1 < 2 * CQ) {
            load_taps(st + 1);
            if (h == 1) load_x(q + 1, xbuf[(q + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float4 (&x)[2][3] = xbuf[q & 1];
        v2f t0[3], t1[3];
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            t0[kx] = h == 0 ? v2f{x[0][kx].x, x[0][kx].y} : v2f{x[0][kx].z, x[0][kx].w};
            t1[kx] = h == 0 ? v2f{x[1][kx].x, x[1][kx].y} : v2f{x[1][kx].z, x[1][kx].w};
        }
        const float (&w)[18] = wd[st & 1];
        auto wv = [&](int tap) { return v2f{w[2 * tap], w[2 * tap + 1]}; };
        v2f p0 = aA[q][h], p1 = aB[q][h], nA = t0[0] * wv(0), nB = t1[0] * wv(0);
#ifdef MI_ABL_M_NODW
        if (false)
#endif
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            p0 = pkfma(t0[kx], wv(6 + kx), p0);
            p1 = pkfma(t0[kx], wv(3 + kx), p1);
            if (kx) nA = pkfma(t0[kx], wv(kx), nA);
        }
#ifdef MI_ABL_M_NODW
        if (false)
#endif
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            p1 = pkfma(t1[kx], wv(6 + kx), p1);
            nA = pkfma(t1[kx], wv(3 + kx), nA);
            if (kx) nB = pkfma(t1[kx], wv(kx), nB);
        }
        aA[q][h] = nA;
        aB[q][h] = nB;
        asm volatile("" : "+v"(aA[q][h]), "+v"(aB[q][h]));
        // the depthwise stage ends here and the MFMAs run as one block: left to itself the scheduler interleaves the last packed FMAs
        // with the first MFMAs, and every switch between the two costs issue cycles (measured on the harness: 0.384 -> 0.353 ms)
        asm volatile("" : "+v"(p0), "+v"(p1));
        __builtin_amdgcn_sched_barrier(0);
#ifdef MI_ABL_M_NOMFMA
        asm volatile("" : "+v"(p0), "+v"(p1));
        if (false)
#endif
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float b0 = e ? p0.y : p0.x, b1 = e ? p1.y : p1.x;
#pragma unroll
            for (int t = 0; t < CQ; t++) {
                const int n = (2 * st + e) * CQ + t;
                const v4f c0 = (st == 0 && e == 0) ? v4f{onext[t].x, onext[t].y, onext[t].z, onext[t].w} : oacc0[t];
                oacc0[t] = mfma4_bcast(wa[n >> 4], b0, c0, n & 15);
                oacc1[t] = mfma4_bcast(wa[n >> 4], b1, oacc1[t], n & 15);
            }
        }
        if (h == 0 && has_res) {  // the next call's first row starts from bias + centre pixels of row c0 + 1
            const v4f nx = v4f{x[1][1].x, x[1][1].y, x[1][1].z, x[1][1].w} + bias4(q);
            onext[q] = make_float4(nx.x, nx.y, nx.z, nx.w);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// strip_row (one input row per call) with the pointwise conv on v_mfma_f32_4x4x1_16b_f32: see strip_row2m.  Input row r: its ky = 2 / 1 / 0
// taps go to the partial output rows r-1 / r / r+1 (aPN on entry / aC / aPN on exit); the finished row r-1 is the B operand of the MFMAs, whose
// first C operand is `onext` = bias + skip of row r-1 (left by the previous call); the call leaves bias + centre pixels of row r there.
template <int CQ>
__device__ __forceinline__ void strip_row1m(const float* me, int ps, const cfloat* cst, bool has_res, v2f (&aPN)[CQ][2], v2f (&aC)[CQ][2],
                                            float4 (&onext)[CQ], v4f (&oacc)[CQ], const float (&wa)[SK<CQ>::NA], const float (&cb)[4 * CQ],
                                            float (&wd)[2][18]) {
    using K = SK<CQ>;
    float4 xbuf[2][3];
    const float* r[3] = {me, me + ps, me + 2 * ps};
    auto load_taps = [&](int st) {
        const cfloat* p = cst;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < 18; i++) wd[st & 1][i] = p[K::OFF_DW + st * K::ST_F + i];
    };
    auto load_x = [&](int q, float4 (&x)[3]) {
#pragma unroll
        for (int kx = 0; kx < 3; kx++) x[kx] = sld4(r[kx] + 4 * q);
    };
    auto bias4 = [&](int q) { return v4f{cb[4 * q], cb[4 * q + 1], cb[4 * q + 2], cb[4 * q + 3]}; };
    load_x(0, xbuf[0]);
#pragma unroll
    for (int st = 0; st < 2 * CQ; st++) {
        const int q = st >> 1, h = st & 1;
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this stage's taps (asked for a stage ago) and pixels
        if (st + 1 < 2 * CQ) {
            load_taps(st + 1);
            if (h == 1) load_x(q + 1, xbuf[(q + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float4 (&x)[3] = xbuf[q & 1];
        v2f t[3];
#pragma unroll
        for (int kx = 0; kx < 3; kx++) t[kx] = h == 0 ? v2f{x[kx].x, x[kx].y} : v2f{x[kx].z, x[kx].w};
        const float (&w)[18] = wd[st & 1];
        auto wv = [&](int tap) { return v2f{w[2 * tap], w[2 * tap + 1]}; };
        v2f n = t[0] * wv(0), c = aC[q][h], pch = aPN[q][h];
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            pch = pkfma(t[kx], wv(6 + kx), pch);
            if (kx) n = pkfma(t[kx], wv(kx), n);
            c = pkfma(t[kx], wv(3 + kx), c);
        }
        aC[q][h] = c;
        aPN[q][h] = n;
        asm volatile("" : "+v"(aC[q][h]), "+v"(aPN[q][h]));
        asm volatile("" : "+v"(pch));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float bv = e ? pch.y : pch.x;
#pragma unroll
            for (int tq = 0; tq < CQ; tq++) {
                const int nn = (2 * st + e) * CQ + tq;
                const v4f c0 = (st == 0 && e == 0) ? v4f{onext[tq].x, onext[tq].y, onext[tq].z, onext[tq].w} : oacc[tq];
                oacc[tq] = mfma4_bcast(wa[nn >> 4], bv, c0, nn & 15);
            }
        }
        if (h == 0 && has_res) {  // the next call's finished row starts from bias + centre pixels of row r
            const v4f nx = v4f{x[1].x, x[1].y, x[1].z, x[1].w} + bias4(q);
            onext[q] = make_float4(nx.x, nx.y, nx.z, nx.w);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int CQ, bool RELU>
__device__ __forceinline__ void strip_act(const v4f (&oacc)[CQ], const cfloat* slopes, float hi, float4 (&o)[CQ]) {
    v2f t[2 * CQ];
#pragma unroll
    for (int q = 0; q < CQ; q++) {
        t[2 * q] = v2f{oacc[q].x, oacc[q].y};
        t[2 * q + 1] = v2f{oacc[q].z, oacc[q].w};
    }
    strip_act<CQ, RELU>(t, slopes, hi, o);
}

// Stride-2 tail of a row pipeline (the BlazeBlock that halves the resolution: DW3x3 s2 (TF SAME on an even size = taps at
// rows 2oy..2oy+2, columns 2ox..2ox+2) -> PW1x1 C -> NH*C -> + [2x2 max-pool of the input, zero channel-pad] -> act).
// lane = OUTPUT pixel; `me2` = this lane's first tap column in the full-width row image the last stride-1 block wrote.
// Rows arrive one per step: an even row 2k finishes output row k-1 (its ky = 2 taps, then the pointwise conv of this
// wave's C output channels) and starts output row k (ky = 0); an odd row adds the ky = 1 taps.  The max-pool runs beside
// it (max over the kx = 0, 1 taps of rows 2k, 2k+1).  MODE 0: odd row, 1: even row that only starts, 2: even row that
// finishes and starts.  Constants record per channel pair: depthwise taps [9][2] at 0, pointwise columns of the two
// channels at 32 and 32 + NH*C (this wave reads entries [hf*C, hf*C + C) of them).
template <int CQ, int NH, int MODE, int PS = 4 * CQ>
__device__ __forceinline__ void strip_row_s2(const float* me2, const cfloat* cst, int hf, bool has_skip, v2f (&acc)[CQ][2], float4 (&mx)[CQ],
                                             v2f (&oacc)[2 * CQ]) {
    constexpr int C = 4 * CQ, Co = NH * C;
    constexpr int ST2 = (32 + 2 * Co + 15) / 16 * 16, OFF_BIAS2 = (C / 2) * ST2;
    float wd[18], wp0[C], wp1[C];
    float4 xbuf[2][3];
    auto load_first = [&](int st) {
        const cfloat* p = cst;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < 18; i++) wd[i] = p[st * ST2 + i];
        if (MODE == 2) {
            const cfloat* pc = p + hf * C;  // wave-uniform offset
#pragma unroll
            for (int i = 0; i < C; i++) wp0[i] = pc[st * ST2 + 32 + i];
        }
    };
    auto load_second = [&](int st) {
        const cfloat* p = cst + hf * C;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < C; i++) wp1[i] = p[st * ST2 + 32 + Co + i];
    };
    auto load_x = [&](int q, float4 (&x)[3]) {
#pragma unroll
        for (int kx = 0; kx < 3; kx++) x[kx] = sld4(me2 + kx * PS + 4 * q);
    };
    if constexpr (MODE == 2) {  // accumulators start from bias (+ the finished max-pool of rows 2k-2, 2k-1 for channels < C)
        const cfloat* bp = cst + OFF_BIAS2 + hf * C;
        asm volatile("" : "+s"(bp));
        if (has_skip) {
#pragma unroll
            for (int q = 0; q < CQ; q++) {
                oacc[2 * q] = v2f{mx[q].x, mx[q].y} + v2f{bp[4 * q], bp[4 * q + 1]};
                oacc[2 * q + 1] = v2f{mx[q].z, mx[q].w} + v2f{bp[4 * q + 2], bp[4 * q + 3]};
            }
        } else {
#pragma unroll
            for (int q = 0; q < CQ; q++) {
                oacc[2 * q] = v2f{bp[4 * q], bp[4 * q + 1]};
                oacc[2 * q + 1] = v2f{bp[4 * q + 2], bp[4 * q + 3]};
            }
        }
#pragma unroll
        for (int o = 0; o < 2 * CQ; o++) asm volatile("" : "+v"(oacc[o]));
    }
    load_first(0);
    load_x(0, xbuf[0]);
#pragma unroll
    for (int st = 0; st < 2 * CQ; st++) {
        const int q = st >> 1, h = st & 1;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (MODE == 2) load_second(st);
        __builtin_amdgcn_sched_barrier(0);
        const float4 (&x)[3] = xbuf[q & 1];
        if (h == 0) {  // running 2x2 max-pool of this quad (columns 2ox, 2ox+1 = taps 0, 1)
            float4 m = make_float4(fmaxf(x[0].x, x[1].x), fmaxf(x[0].y, x[1].y), fmaxf(x[0].z, x[1].z), fmaxf(x[0].w, x[1].w));
            if (MODE == 0) m = make_float4(fmaxf(m.x, mx[q].x), fmaxf(m.y, mx[q].y), fmaxf(m.z, mx[q].z), fmaxf(m.w, mx[q].w));
            mx[q] = m;
        }
        v2f t[3];
#pragma unroll
        for (int kx = 0; kx < 3; kx++) t[kx] = h == 0 ? v2f{x[kx].x, x[kx].y} : v2f{x[kx].z, x[kx].w};
        auto wv = [&](int tap) { return v2f{wd[2 * tap], wd[2 * tap + 1]}; };
        v2f pch = acc[q][h], n;
        if (MODE == 0) {
            n = pch;
#pragma unroll
            for (int kx = 0; kx < 3; kx++) n = pkfma(t[kx], wv(3 + kx), n);
        } else {
            n = t[0] * wv(0);
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                if (MODE == 2) pch = pkfma(t[kx], wv(6 + kx), pch);
                if (kx) n = pkfma(t[kx], wv(kx), n);
            }
        }
        acc[q][h] = n;
        asm volatile("" : "+v"(acc[q][h]));
        if constexpr (MODE == 2) {
#pragma unroll
            for (int o = 0; o < 2 * CQ; o++) oacc[o] = pkfma(v2f{pch.x, pch.x}, v2f{wp0[2 * o], wp0[2 * o + 1]}, oacc[o]);
            asm volatile("" : "+v"(pch));
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (st + 1 < 2 * CQ) {
            load_first(st + 1);
            if (h == 1) load_x(q + 1, xbuf[(q + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 2) {
#pragma unroll
            for (int o = 0; o < 2 * CQ; o++) {
                oacc[o] = pkfma(v2f{pch.y, pch.y}, v2f{wp1[2 * o], wp1[2 * o + 1]}, oacc[o]);
                asm volatile("" : "+v"(oacc[o]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// strip_row_s2 with the pointwise conv of MODE 2 on v_mfma_f32_4x4x1_16b_f32 (see strip_row2m): lane = OUTPUT pixel, wa = this wave's C x C
// slice of the filter as A operands (strip_pack_consts_s2: OFF_A2 + hf NA 64), one wait per stage (only the 18 taps are streamed).
template <int CQ, int NH, int MODE, int PS>
__device__ __forceinline__ void strip_row_s2m(const float* me2, const cfloat* cst, int hf, bool has_skip, v2f (&acc)[CQ][2], float4 (&mx)[CQ],
                                              v4f (&oacc)[CQ], const float (&wa)[SK<CQ>::NA]) {
    constexpr int C = 4 * CQ, Co = NH * C;
    constexpr int ST2 = (32 + 2 * Co + 15) / 16 * 16, OFF_BIAS2 = (C / 2) * ST2;
    float wd[2][18];
    float4 xbuf[2][3];
    auto load_taps = [&](int st) {
        const cfloat* p = cst;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int i = 0; i < 18; i++) wd[st & 1][i] = p[st * ST2 + i];
    };
    auto load_x = [&](int q, float4 (&x)[3]) {
#pragma unroll
        for (int kx = 0; kx < 3; kx++) x[kx] = sld4(me2 + kx * PS + 4 * q);
    };
    if constexpr (MODE == 2) {  // accumulators start from bias (+ the finished max-pool of rows 2k-2, 2k-1 for channels < C)
        const cfloat* bp = cst + OFF_BIAS2 + hf * C;
        asm volatile("" : "+s"(bp));
        if (has_skip) {
#pragma unroll
            for (int q = 0; q < CQ; q++) oacc[q] = v4f{mx[q].x, mx[q].y, mx[q].z, mx[q].w} + v4f{bp[4 * q], bp[4 * q + 1], bp[4 * q + 2], bp[4 * q + 3]};
        } else {
#pragma unroll
            for (int q = 0; q < CQ; q++) oacc[q] = v4f{bp[4 * q], bp[4 * q + 1], bp[4 * q + 2], bp[4 * q + 3]};
        }
#pragma unroll
        for (int q = 0; q < CQ; q++) asm volatile("" : "+v"(oacc[q]));
    }
    load_taps(0);
    load_x(0, xbuf[0]);
#pragma unroll
    for (int st = 0; st < 2 * CQ; st++) {
        const int q = st >> 1, h = st & 1;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (st + 1 < 2 * CQ) {
            load_taps(st + 1);
            if (h == 1) load_x(q + 1, xbuf[(q + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float4 (&x)[3] = xbuf[q & 1];
        if (h == 0) {  // running 2x2 max-pool of this quad (columns 2ox, 2ox+1 = taps 0, 1)
            float4 m = make_float4(fmaxf(x[0].x, x[1].x), fmaxf(x[0].y, x[1].y), fmaxf(x[0].z, x[1].z), fmaxf(x[0].w, x[1].w));
            if (MODE == 0) m = make_float4(fmaxf(m.x, mx[q].x), fmaxf(m.y, mx[q].y), fmaxf(m.z, mx[q].z), fmaxf(m.w, mx[q].w));
            mx[q] = m;
        }
        v2f t[3];
#pragma unroll
        for (int kx = 0; kx < 3; kx++) t[kx] = h == 0 ? v2f{x[kx].x, x[kx].y} : v2f{x[kx].z, x[kx].w};
        const float (&w)[18] = wd[st & 1];
        auto wv = [&](int tap) { return v2f{w[2 * tap], w[2 * tap + 1]}; };
        v2f pch = acc[q][h], n;
        if (MODE == 0) {
            n = pch;
#pragma unroll
            for (int kx = 0; kx < 3; kx++) n = pkfma(t[kx], wv(3 + kx), n);
        } else {
            n = t[0] * wv(0);
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                if (MODE == 2) pch = pkfma(t[kx], wv(6 + kx), pch);
                if (kx) n = pkfma(t[kx], wv(kx), n);
            }
        }
        acc[q][h] = n;
        asm volatile("" : "+v"(acc[q][h]));
        if constexpr (MODE == 2) {
            asm volatile("" : "+v"(pch));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float bv = e ? pch.y : pch.x;
#pragma unroll
                for (int tq = 0; tq < CQ; tq++) {
                    const int nn = (2 * st + e) * CQ + tq;
                    oacc[tq] = mfma4_bcast(wa[nn >> 4], bv, oacc[tq], nn & 15);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int CQ, bool RELU>
__global__ __launch_bounds__(256, CQ <= 6 ? 3 : 2) void strip_kernel(StripArgs a) {
    using K = SK<CQ>;
    constexpr int C = K::C, NL = K::NL, BUF_F = K::BUF_F;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: everything derived from it (band, rows, branches) stays on the SALU
    float* wbase = lds + wave * K::WAVE_F;
    int gw = blockIdx.x * 4 + wave;
    const int band = gw % a.bands;
    gw /= a.bands;
    const int strip = gw % a.strips, b = gw / a.strips;
    if (b >= a.B) return;  // whole wave; no workgroup-level synchronisation anywhere in this kernel
    const int x0 = strip * 64;
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.H);
    if (y0 >= y1) return;
    const float* in = a.in + (long)b * a.in_fs;
    float* out = a.out + (long)b * a.out_fs;
    // constant address space: the blob is never written while kernels run, and wave-uniform loads from it become s_load
    const cfloat* cst = (const cfloat*)a.consts;

    // Row image in LDS: [66 pixels][C] floats, filled by LDS-DMA (global_load_lds_dwordx4: LDS destination = M0 + offset +
    // 16 * lane, source address per lane), so float4 number f = lane + 64k of the image comes from (pixel f / CQ, quad
    // f % CQ) of the strip.  A lane whose pixel lies outside the image (halo of an edge strip, tail of a partial strip)
    // reads a clamped in-image pixel instead; the one pixel column that must be zero (left of x = 0, right of x = W-1)
    // is cleared in LDS after the DMA has landed — no per-lane branches around the memory instructions.
    // The DMA is issued from inline asm: hipcc would otherwise drain vmcnt(0) before every LDS read that follows a DMA
    // it can see (it cannot tell the three row buffers apart), which serialises the two-rows-ahead prefetch.
    int goff[NL];  // byte offset of the lane's float4 from the row start, per DMA instruction
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int f = min(lane + 64 * k, K::NF - 1), px = f / CQ, qd = f - px * CQ;
        goff[k] = (min(max(x0 - 1 + px, 0), a.W - 1) * C + 4 * qd) * 4;
    }
    const int npx = min(64, a.W - x0);              // valid pixels of this strip
    const bool full = npx == 64;
    const int zl = x0 == 0 ? 0 : -1;                // image pixel columns to clear (-1: none)
    const int zr = x0 + 64 >= a.W ? npx + 1 : -1;
    const long gout = (long)x0 * C + 4 * lane;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)wbase);

    // start the DMA of input row r into row buffer bi (always NL instructions, so that counted vmcnt waits stay valid;
    // a row outside the image reads a clamped row and is zeroed when it is consumed)
    auto issue_row = [&](int r, int bi) {
        const char* src = reinterpret_cast<const char*>(in + (long)min(max(r, 0), a.H - 1) * a.W * C);
        const unsigned dstb = lds_wave + (unsigned)(bi * BUF_F * 4);
        constexpr int TAIL = K::NF - 64 * (NL - 1);  // active lanes of the last instruction
#pragma unroll
        for (int k = 0; k < NL; k++) {
            if (k < NL - 1 || TAIL == 64) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dstb + 1024u * k), "v"(goff[k]), "s"(src) : "memory", "m0");
            } else {
                unsigned long long saved;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(dstb + 1024u * k), "v"(goff[k]), "s"(src), "n"((1ull << (TAIL & 63)) - 1) : "memory", "m0");
            }
        }
    };
    // after the DMA of row r has landed in buffer bi: zero what lies outside the image (zfix: this lane's float offset in
    // the row image of the border-column float4 it clears, -1 for the lanes that clear nothing)
    const int zfix = lane < CQ ? (zl >= 0 ? zl * C + 4 * lane : -1) : (lane < 2 * CQ ? (zr >= 0 ? zr * C + 4 * (lane - CQ) : -1) : -1);
    auto fix_row = [&](int r, int bi) {
        float* buf = wbase + bi * BUF_F;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < 0 || r >= a.H) {  // wave-uniform, only at the top / bottom image edge
#pragma unroll
            for (int k = 0; k < NL; k++)
                if (64 * (k + 1) <= K::NF || lane < K::NF - 64 * k) sst4(buf + 4 * (lane + 64 * k), z);
        } else if (zfix >= 0) {
            sst4(buf + zfix, z);
        }
    };

    v2f oacc[2 * CQ];
    auto row = [&](auto emit_t, int bi, v2f (&aPN)[CQ][2], v2f (&aC)[CQ][2], const float4 (&xres)[CQ], float4 (&xcur)[CQ]) {
        strip_row<CQ, decltype(emit_t)::value>(wbase + bi * BUF_F + lane * C, cst, a.has_res != 0, aPN, aC, xres, xcur, oacc);
    };
    // epilogue of output row y: activation, then transposed through the (now dead) row buffer bi so that every store
    // instruction writes 1 KiB of consecutive bytes
    auto epilogue = [&](int y, int bi) {
        float* obuf = wbase + bi * BUF_F;
        float4 o[CQ];
        strip_act<CQ, RELU>(oacc, cst + K::OFF_SLOPE, a.hi, o);
#pragma unroll
        for (int q = 0; q < CQ; q++) sst4(obuf + lane * C + 4 * q, o[q]);
        wave_sync();
        float* dst = out + (long)y * a.W * C + gout;
        if (full) {  // wave-uniform
#pragma unroll
            for (int k = 0; k < CQ; k++) sst4(dst + 256 * k, sld4(obuf + 4 * lane + 256 * k));
        } else {
#pragma unroll
            for (int k = 0; k < CQ; k++) {
                const float4 o = sld4(obuf + 4 * lane + 256 * k);
                if (lane + 64 * k < npx * CQ) sst4(dst + 256 * k, o);
            }
        }
    };

    // two partial-row sets and two centre rows alternate roles from one row to the next (the loop is unrolled by two)
    v2f acc0[CQ][2], acc1[CQ][2];
    float4 xa[CQ], xb[CQ];
#pragma unroll
    for (int q = 0; q < CQ; q++) {
        acc0[q][0] = acc0[q][1] = acc1[q][0] = acc1[q][1] = v2f{0.f, 0.f};
        xa[q] = xb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // Step r consumes input row r from buffer r & 1 (band-relative), emits output row r-1 through the same buffer and then
    // starts the DMA of row r+2 into it; row r+1 is in flight in the other buffer meanwhile.  vm operations are retired
    // in issue order, so "row r has landed" = at most [stores of step r-1] + [DMA of row r+1] still outstanding; the
    // store count is only relied on when it is exact (full strip, previous step emitted), otherwise the wait also covers
    // those stores.
    // EMIT is a compile-time property of the call site (the two priming rows of a band produce no output): as a run-time
    // branch the compiler sinks a third of the depthwise FMAs into it, across the scheduling fences, and spills SGPRs.
#ifdef MI_STRIP_STAMPS
    unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
#define MI_SSTAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); st_acc[k] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_SSTAMP(k)
#endif
    auto step = [&](auto emit, int r, v2f (&aPN)[CQ][2], v2f (&aC)[CQ][2], const float4 (&xprev)[CQ], float4 (&xcur)[CQ]) {
        const int bi = (r - (y0 - 1)) & 1;
        if (r + 1 <= y1) {
            if (full && r >= y0 + 2) wait_vm<NL + CQ>();
            else wait_vm<NL>();
        } else {
            wait_vm<0>();
        }
        MI_SSTAMP(0)
        fix_row(r, bi);
        wave_sync();
        MI_SSTAMP(1)
        row(emit, bi, aPN, aC, xprev, xcur);
        wave_sync();  // every read of the row image above is issued before the epilogue overwrites it
        MI_SSTAMP(2)
        if constexpr (decltype(emit)::value) epilogue(r - 1, bi);
        wave_sync();
        if (r + 2 <= y1) issue_row(r + 2, bi);
        MI_SSTAMP(3)
    };
    issue_row(y0 - 1, 0);
    issue_row(y0, 1);
    step(std::false_type{}, y0 - 1, acc0, acc1, xb, xa);
    step(std::false_type{}, y0, acc1, acc0, xa, xb);
    for (int r = y0 + 1; r <= y1; r += 2) {
        step(std::true_type{}, r, acc0, acc1, xb, xa);
        if (r + 1 > y1) break;
        step(std::true_type{}, r + 1, acc1, acc0, xa, xb);
    }
#ifdef MI_STRIP_STAMPS
    if (a.stamps && lane == 0)
        for (int k = 0; k < 4; k++) a.stamps[((long)blockIdx.x * 4 + wave) * 8 + k] = st_acc[k];
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-pipelined chain of KB same-shape stride-1 blocks (W <= 128): one workgroup = 2 * KB waves.  Waves (2j, 2j+1) run
// block j of the chain on the two 64-column strips of a band (W <= 64: on two independent bands); block j's finished row
// goes, activated, into a double-buffered full-width row image in LDS instead of HBM, where block j+1's waves pick it
// up one step later (they read their strip's 66-pixel window of it, so the halo columns come from the partner wave for
// free).  Only block 0 reads HBM (LDS-DMA) and only block KB-1 writes it: the HBM traffic of KB layers becomes that of
// one.  Steps are separated by one s_barrier; block j lags block j-1 by two rows, so a band costs band_rows + 3 KB - 1
// steps, and block j computes KB-1-j extra rows above and below the band (the halo of the blocks behind it).
constexpr int kMaxPipe = 4;
struct PipeArgs {
    const float* in;
    float* out;
    const float* consts[kMaxPipe];  // strip_pack_consts() blobs, one per block
    long in_fs, out_fs;
    int B, H, W;
    int strips, bands, band_rows, units;
    int rows_per_step;              // 1: strip_pipe_kernel, 2: strip_pipe2_kernel (even band starts / lengths), 3: strip_pipe2m_kernel
    int has_res[kMaxPipe];
    float hi[kMaxPipe];
    int prio;                       // strip_pipe2m_kernel: which waves run at issue priority 1 (1: both ends of the pipeline, 0: none, 2: block 0, 3: last stage)
#ifdef MI_PIPE_STAMPS  // development harness only: s_memtime accumulators per wave (8 per wave: compute, barrier 1, hand-over / store, barrier 2, steps)
    unsigned long long* stamps;
#endif
};

// NH2 > 0: the chain ends in a stride-2 block (see strip_row_s2) run by NH2 extra waves (one per C output channels); KB
// counts that block too.
template <int CQ, int KB, bool RELU, int NH2>
__global__ __launch_bounds__(NH2 ? 128 * (KB - 1) + 64 * NH2 : 128 * KB, 2) void strip_pipe_kernel(PipeArgs a) {
    using K = SK<CQ>;
    constexpr int C = K::C, NL = K::NL, BUF_F = K::BUF_F;
    constexpr int S = NH2 ? KB - 1 : KB;         // stride-1 blocks
    constexpr int NT = 128 * S + 64 * NH2;       // threads
    constexpr int SLOTS = 2;
    constexpr int IMG = 132 * C;                 // floats of one full-width row image (<= 130 pixels used)
    constexpr int RING_F = (KB - 1) * SLOTS * IMG;   // KB-1 hand-over rings of two row images
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool tail = NH2 && w >= 2 * S;         // wave of the stride-2 block
    const int p = tail ? 0 : (w & 1), role = tail ? S : (w >> 1);
    const int hf = tail ? w - 2 * S : 0;         // which C output channels a tail wave computes
    // wave-private scratch: block 0 waves own two DMA row buffers each, the waves that store one transposition buffer each
    float* scratch = lds + RING_F + (role == 0 ? p * 2 * BUF_F : 4 * BUF_F + (tail ? hf : p) * BUF_F);
    const int unit = a.strips == 2 ? (int)blockIdx.x : 2 * (int)blockIdx.x + p;
    const bool active = unit < a.units;
    const int band = unit % a.bands, b = min(unit / a.bands, a.B - 1);
    const int x0 = a.strips == 2 ? 64 * p : 0;
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.H);
    const float* in = a.in + (long)b * a.in_fs;
    float* out = a.out + (long)b * a.out_fs;
    const cfloat* cst = (const cfloat*)a.consts[role];
    const bool has_res = a.has_res[role] != 0;
    const float hi = a.hi[role];
    // the last stride-1 block produces rows [y0, hi_last): one more row when a stride-2 tail follows (its third tap row)
    const int hi_last = NH2 ? y1 + 1 : y1;
    const int lo_j = y0 - (S - 1 - role), hi_j = hi_last + (S - 1 - role);  // this (stride-1) block produces rows [lo_j, hi_j)
    const int img_p = a.strips == 2 ? 64 * C * p : 66 * C * p;           // this wave's window / sub-image inside a row image

    // clear the hand-over rings once: their border pixel columns (left of x = 0, right of x = W-1) are never written
    for (int i = threadIdx.x; i < RING_F / 4; i += NT) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    int goff[NL];  // block 0: byte offset of the lane's float4 from the row start, per DMA instruction (see strip_kernel)
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int f = min(lane + 64 * k, K::NF - 1), px = f / CQ, qd = f - px * CQ;
        goff[k] = (min(max(x0 - 1 + px, 0), a.W - 1) * C + 4 * qd) * 4;
    }
    const int npx = min(64, a.W - x0);
    const bool full = npx == 64;
    const int zl = x0 == 0 ? 0 : -1;
    const int zr = x0 + 64 >= a.W ? npx + 1 : -1;
    const long gout = (long)x0 * C + 4 * lane;
    const unsigned lds_scratch = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)scratch);
    auto issue_row = [&](int r, int bi) {
        const char* src = reinterpret_cast<const char*>(in + (long)min(max(r, 0), a.H - 1) * a.W * C);
        const unsigned dstb = lds_scratch + (unsigned)(bi * BUF_F * 4);
        constexpr int TAIL = K::NF - 64 * (NL - 1);
#pragma unroll
        for (int k = 0; k < NL; k++) {
            if (k < NL - 1 || TAIL == 64) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dstb + 1024u * k), "v"(goff[k]), "s"(src) : "memory", "m0");
            } else {
                unsigned long long saved;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(dstb + 1024u * k), "v"(goff[k]), "s"(src), "n"((1ull << (TAIL & 63)) - 1) : "memory", "m0");
            }
        }
    };
    const int zfix = lane < CQ ? (zl >= 0 ? zl * C + 4 * lane : -1) : (lane < 2 * CQ ? (zr >= 0 ? zr * C + 4 * (lane - CQ) : -1) : -1);
    auto fix_row = [&](int r, int bi) {
        float* buf = scratch + bi * BUF_F;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < 0 || r >= a.H) {
#pragma unroll
            for (int k = 0; k < NL; k++)
                if (64 * (k + 1) <= K::NF || lane < K::NF - 64 * k) sst4(buf + 4 * (lane + 64 * k), z);
        } else if (zfix >= 0) {
            sst4(buf + zfix, z);
        }
    };

    v2f acc0[CQ][2], acc1[CQ][2], oacc[2 * CQ];
    float4 xa[CQ], xb[CQ];
#pragma unroll
    for (int q = 0; q < CQ; q++) {
        acc0[q][0] = acc0[q][1] = acc1[q][0] = acc1[q][1] = v2f{0.f, 0.f};
        xa[q] = xb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int c_first = y0 - S - 2 * role;  // row this block would consume at step 0 (it starts at step 3 * role)
    if (role == 0 && active) {
        issue_row(c_first, 0);
        issue_row(c_first + 1, 1);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();

    auto wg_barrier = [&]() {
        // raw s_barrier with an LDS-only wait, so that the stores of block KB-1 and the DMA of block 0 stay in flight across it
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // ---- stride-2 tail: one wave serves both strips (W > 64: lane = output column) or both units (W <= 64: 32 lanes each).
    // Its state (partial depthwise row, running max-pool) lives in the registers the stride-1 waves use for acc0 / xa.
    auto tail_step = [&](int t, v2f (&tacc)[CQ][2], float4 (&tmx)[CQ]) {
        if constexpr (NH2 > 0) {
            const int k = t - 3 * S;  // band-relative index of the row the last stride-1 block handed over in step t-1
            if (k < 0 || k > a.band_rows) return;
            constexpr int Co = NH2 * C;
            constexpr int ST2 = (32 + 2 * Co + 15) / 16 * 16, OFF_SLOPE2 = (C / 2) * ST2 + 64;
            const cfloat* cst2 = (const cfloat*)a.consts[S];
            const float* img = lds + ((S - 1) * SLOTS + ((t - 1) & 1)) * IMG;
            const int tu = a.strips == 2 ? 0 : lane >> 5, tox = a.strips == 2 ? lane : lane & 31;
            const float* me2 = img + tu * 66 * C + (2 * tox + 1) * C;
            const bool skip = hf == 0 && a.has_res[S] != 0;
            if (k & 1) {
                strip_row_s2<CQ, NH2, 0>(me2, cst2, hf, skip, tacc, tmx, oacc);
            } else if (k < 2) {
                strip_row_s2<CQ, NH2, 1>(me2, cst2, hf, skip, tacc, tmx, oacc);
            } else {
                strip_row_s2<CQ, NH2, 2>(me2, cst2, hf, skip, tacc, tmx, oacc);
                float4 o[CQ];
                strip_act<CQ, RELU>(oacc, cst2 + OFF_SLOPE2 + hf * C, a.hi[S], o);
                float* obuf = scratch;
#pragma unroll
                for (int q = 0; q < CQ; q++) sst4(obuf + lane * C + 4 * q, o[q]);
                wave_sync();
                // coalesced read-back: float4 f = lane + 64 j of the [64 pixels][C] image -> pixel f / CQ (unit, column)
                const int Wo = a.W >> 1;
#pragma unroll
                for (int j = 0; j < CQ; j++) {
                    const int f = lane + 64 * j, px = f / CQ, qd = f - px * CQ;
                    const int u = a.strips == 2 ? 0 : px >> 5, ox = a.strips == 2 ? px : px & 31;
                    const int un = a.strips == 2 ? (int)blockIdx.x : 2 * (int)blockIdx.x + u;
                    const int ub = un % a.bands, bb = min(un / a.bands, a.B - 1);
                    const int uy0 = ub * a.band_rows, uy1 = min(uy0 + a.band_rows, a.H);
                    const float4 v = sld4(obuf + 4 * f);
                    if (un < a.units && uy0 + k <= uy1 && ox < Wo) {
                        const int oy = (uy0 + k - 2) >> 1;
                        sst4(a.out + (long)bb * a.out_fs + ((long)oy * Wo + ox) * Co + hf * C + 4 * qd, v);
                    }
                }
                wave_sync();
            }
        }
    };
    auto step = [&](int t, v2f (&aPN)[CQ][2], v2f (&aC)[CQ][2], const float4 (&xprev)[CQ], float4 (&xcur)[CQ], v2f (&tacc)[CQ][2], float4 (&tmx)[CQ]) {
        const int c = c_first + t;  // row consumed in this step; the row it completes is c - 1
        const int e = c - 1;
        bool hand_over = false;     // this wave has a finished row for the next block
        float4 o[CQ];
        if (!tail && active && c >= lo_j - 1 && c <= hi_j) {
            const float* me;
            if (role == 0) {
                if (c + 1 <= hi_j) wait_vm<NL>();
                else wait_vm<0>();
                fix_row(c, t & 1);
                wave_sync();
                me = scratch + (t & 1) * BUF_F + lane * C;
            } else {
                me = lds + ((role - 1) * SLOTS + ((t - 1) & 1)) * IMG + img_p + lane * C;
            }
            strip_row<CQ, true>(me, cst, has_res, aPN, aC, xprev, xcur, oacc);
            wave_sync();
            if (e >= lo_j && e < hi_j) {
                strip_act<CQ, RELU>(oacc, cst + K::OFF_SLOPE, hi, o);
                if (NH2 || role < S - 1) {
                    hand_over = true;
                } else {
                    float* obuf = scratch;
#pragma unroll
                    for (int q = 0; q < CQ; q++) sst4(obuf + lane * C + 4 * q, o[q]);
                    wave_sync();
                    float* dst = out + (long)e * a.W * C + gout;
                    if (full) {
#pragma unroll
                        for (int k = 0; k < CQ; k++) sst4(dst + 256 * k, sld4(obuf + 4 * lane + 256 * k));
                    } else {
#pragma unroll
                        for (int k = 0; k < CQ; k++) {
                            const float4 v = sld4(obuf + 4 * lane + 256 * k);
                            if (lane + 64 * k < npx * CQ) sst4(dst + 256 * k, v);
                        }
                    }
                }
            }
            if (role == 0 && c + 2 <= hi_j) {
                wave_sync();
                issue_row(c + 2, t & 1);
            }
        }
        if constexpr (NH2 > 0) {
            if (tail) tail_step(t, tacc, tmx);
        }
        if (hand_over) {
            // hand the row to the next block; rows outside the image are that block's zero padding
            float* dstl = lds + (role * SLOTS + (t & 1)) * IMG + img_p + (1 + lane) * C;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 0 || e >= a.H) {  // wave-uniform
#pragma unroll
                for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, z);
            } else if (full) {
#pragma unroll
                for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, o[q]);
            } else {
#pragma unroll
                for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, lane >= npx ? z : o[q]);
            }
        }
        wg_barrier();
    };
    const int T = NH2 ? a.band_rows + 3 * S + 1 : a.band_rows + 3 * S - 1;
    for (int t = 0; t < T; t += 2) {
        step(t, acc0, acc1, xb, xa, acc0, xa);
        if (t + 1 >= T) break;
        step(t + 1, acc1, acc0, xa, xb, acc0, xa);
    }
}

// strip_pipe_kernel (ONE row per step: double-buffered single-row rings, one barrier per step, the hand-over writes of the waves that
// finish early overlap the others' compute) with the pointwise convs on the matrix cores (strip_row1m) — the resident filters took away
// what made two rows per step pay (a scalar weight stream per row).  Bit-identical to the other pipeline kernels.
template <int CQ, int KB, bool RELU, int NH2>
__global__ __launch_bounds__(NH2 ? 128 * (KB - 1) + 64 * NH2 : 128 * KB, 2) void strip_pipe1m_kernel(PipeArgs a) {
    using K = SK<CQ>;
    constexpr int C = K::C, NL = K::NL, BUF_F = K::BUF_F;
    constexpr int S = NH2 ? KB - 1 : KB;         // stride-1 blocks
    constexpr int NT = 128 * S + 64 * NH2;       // threads
    constexpr int SLOTS = 2;
    constexpr int PXS = C + 4;                   // pixel stride of a hand-over row image (floats): conflict-free 16-byte reads and writes
    constexpr int IMG = 132 * PXS;               // floats of one full-width row image (<= 130 pixels used)
    constexpr int RING_F = (KB - 1) * SLOTS * IMG;   // KB-1 hand-over rings of two row images
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool tail = NH2 && w >= 2 * S;         // wave of the stride-2 block
    const int p = tail ? 0 : (w & 1), role = tail ? S : (w >> 1);
    const int hf = tail ? w - 2 * S : 0;         // which C output channels a tail wave computes
    // wave-private scratch: block 0 waves own two DMA row buffers each, the waves that store one transposition buffer each
    float* scratch = lds + RING_F + (role == 0 ? p * 2 * BUF_F : 4 * BUF_F + (tail ? hf : p) * BUF_F);
    const int unit = a.strips == 2 ? (int)blockIdx.x : 2 * (int)blockIdx.x + p;
    const bool active = unit < a.units;
    const int band = unit % a.bands, b = min(unit / a.bands, a.B - 1);
    const int x0 = a.strips == 2 ? 64 * p : 0;
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.H);
    const float* in = a.in + (long)b * a.in_fs;
    float* out = a.out + (long)b * a.out_fs;
    const cfloat* cst = (const cfloat*)a.consts[role];
    const bool has_res = a.has_res[role] != 0;
    const float hi = a.hi[role];
    // the last stride-1 block produces rows [y0, hi_last): one more row when a stride-2 tail follows (its third tap row)
    const int hi_last = NH2 ? y1 + 1 : y1;
    const int lo_j = y0 - (S - 1 - role), hi_j = hi_last + (S - 1 - role);  // this (stride-1) block produces rows [lo_j, hi_j)
    const int img_p = a.strips == 2 ? 64 * PXS * p : 66 * PXS * p;       // this wave's window / sub-image inside a row image
    {
        const bool first = role == 0, last = tail || (NH2 == 0 && role == S - 1);
        if ((a.prio == 1 && (first || last)) || (a.prio == 2 && first) || (a.prio == 3 && last)) __builtin_amdgcn_s_setprio(1);
    }

    // clear the hand-over rings once: their border pixel columns (left of x = 0, right of x = W-1) are never written
    for (int i = threadIdx.x; i < RING_F / 4; i += NT) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    int goff[NL];  // block 0: byte offset of the lane's float4 from the row start, per DMA instruction (see strip_kernel)
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int f = min(lane + 64 * k, K::NF - 1), px = f / CQ, qd = f - px * CQ;
        goff[k] = (min(max(x0 - 1 + px, 0), a.W - 1) * C + 4 * qd) * 4;
    }
    const int npx = min(64, a.W - x0);
    const bool full = npx == 64;
    const int zl = x0 == 0 ? 0 : -1;
    const int zr = x0 + 64 >= a.W ? npx + 1 : -1;
    const long gout = (long)x0 * C + 4 * lane;
    const unsigned lds_scratch = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)scratch);
    auto issue_row = [&](int r, int bi) {
        const char* src = reinterpret_cast<const char*>(in + (long)min(max(r, 0), a.H - 1) * a.W * C);
        const unsigned dstb = lds_scratch + (unsigned)(bi * BUF_F * 4);
        constexpr int TAIL = K::NF - 64 * (NL - 1);
#pragma unroll
        for (int k = 0; k < NL; k++) {
            if (k < NL - 1 || TAIL == 64) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dstb + 1024u * k), "v"(goff[k]), "s"(src) : "memory", "m0");
            } else {
                unsigned long long saved;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(dstb + 1024u * k), "v"(goff[k]), "s"(src), "n"((1ull << (TAIL & 63)) - 1) : "memory", "m0");
            }
        }
    };
    const int zfix = lane < CQ ? (zl >= 0 ? zl * C + 4 * lane : -1) : (lane < 2 * CQ ? (zr >= 0 ? zr * C + 4 * (lane - CQ) : -1) : -1);
    auto fix_row = [&](int r, int bi) {
        float* buf = scratch + bi * BUF_F;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < 0 || r >= a.H) {
#pragma unroll
            for (int k = 0; k < NL; k++)
                if (64 * (k + 1) <= K::NF || lane < K::NF - 64 * k) sst4(buf + 4 * (lane + 64 * k), z);
        } else if (zfix >= 0) {
            sst4(buf + zfix, z);
        }
    };

    v2f acc0[CQ][2], acc1[CQ][2];
    v4f oacc[CQ], toacc[CQ];
    float4 onext[CQ];    // bias + skip of the row the next call finishes (stride-2 tail wave: its running max-pool)
    float wa[K::NA];     // pointwise filter, A operands
    float cb[C], wd[2][18];
#pragma unroll
    for (int q = 0; q < CQ; q++) {
        acc0[q][0] = acc0[q][1] = acc1[q][0] = acc1[q][1] = v2f{0.f, 0.f};
        onext[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto prefetch_consts = [&]() {  // bias + first-stage taps of the coming step, asked for before the barrier
        const cfloat* pc = cst;
        asm volatile("" : "+s"(pc));
#pragma unroll
        for (int i = 0; i < C; i++) cb[i] = pc[K::OFF_BIAS + i];
#pragma unroll
        for (int i = 0; i < 18; i++) wd[0][i] = pc[K::OFF_DW + i];
    };
    auto pin_consts = [&]() {
#pragma unroll
        for (int i = 0; i < C; i++) asm volatile("" : "+s"(cb[i]));
#pragma unroll
        for (int i = 0; i < 18; i++) asm volatile("" : "+s"(wd[0][i]));
    };
    if (!tail) {
#pragma unroll
        for (int i = 0; i < K::NA; i++) wa[i] = a.consts[role][K::OFF_A + 64 * i + lane];
    } else if constexpr (NH2 > 0) {
        constexpr int ST2h = (32 + 2 * NH2 * C + 15) / 16 * 16, OFF_A2 = (C / 2) * ST2h + 128;
#pragma unroll
        for (int i = 0; i < K::NA; i++) wa[i] = a.consts[S][OFF_A2 + (hf * K::NA + i) * 64 + lane];
    }
    prefetch_consts();
    const int c_first = y0 - S - 2 * role;  // row this block would consume at step 0 (it starts at step 3 * role)
    if (role == 0 && active) {
        issue_row(c_first, 0);
        issue_row(c_first + 1, 1);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    pin_consts();
    __builtin_amdgcn_s_barrier();
    if (!tail && !has_res) {
#pragma unroll
        for (int q = 0; q < CQ; q++) onext[q] = make_float4(cb[4 * q], cb[4 * q + 1], cb[4 * q + 2], cb[4 * q + 3]);
    }

    auto wg_barrier = [&]() {
        // raw s_barrier with an LDS-only wait (and the scalar loads of the next step's constants), so that the stores of block KB-1 and the
        // DMA of block 0 stay in flight across it
        prefetch_consts();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        pin_consts();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // ---- stride-2 tail: one wave serves both strips (W > 64: lane = output column) or both units (W <= 64: 32 lanes each).
    // Its state (partial depthwise row, running max-pool) lives in the registers the stride-1 waves use for acc0 / xa.
    auto tail_step = [&](int t, v2f (&tacc)[CQ][2], float4 (&tmx)[CQ]) {
        if constexpr (NH2 > 0) {
            const int k = t - 3 * S;  // band-relative index of the row the last stride-1 block handed over in step t-1
            if (k < 0 || k > a.band_rows) return;
            constexpr int Co = NH2 * C;
            constexpr int ST2 = (32 + 2 * Co + 15) / 16 * 16, OFF_SLOPE2 = (C / 2) * ST2 + 64;
            const cfloat* cst2 = (const cfloat*)a.consts[S];
            const float* img = lds + ((S - 1) * SLOTS + ((t - 1) & 1)) * IMG;
            const int tu = a.strips == 2 ? 0 : lane >> 5, tox = a.strips == 2 ? lane : lane & 31;
            const float* me2 = img + tu * 66 * PXS + (2 * tox + 1) * PXS;
            const bool skip = hf == 0 && a.has_res[S] != 0;
            if (k & 1) {
                strip_row_s2m<CQ, NH2, 0, PXS>(me2, cst2, hf, skip, tacc, tmx, toacc, wa);
            } else if (k < 2) {
                strip_row_s2m<CQ, NH2, 1, PXS>(me2, cst2, hf, skip, tacc, tmx, toacc, wa);
            } else {
                strip_row_s2m<CQ, NH2, 2, PXS>(me2, cst2, hf, skip, tacc, tmx, toacc, wa);
                float4 o[CQ];
                strip_act<CQ, RELU>(toacc, cst2 + OFF_SLOPE2 + hf * C, a.hi[S], o);
                float* obuf = scratch;
#pragma unroll
                for (int q = 0; q < CQ; q++) sst4(obuf + lane * C + 4 * q, o[q]);
                wave_sync();
                // coalesced read-back: float4 f = lane + 64 j of the [64 pixels][C] image -> pixel f / CQ (unit, column)
                const int Wo = a.W >> 1;
#pragma unroll
                for (int j = 0; j < CQ; j++) {
                    const int f = lane + 64 * j, px = f / CQ, qd = f - px * CQ;
                    const int u = a.strips == 2 ? 0 : px >> 5, ox = a.strips == 2 ? px : px & 31;
                    const int un = a.strips == 2 ? (int)blockIdx.x : 2 * (int)blockIdx.x + u;
                    const int ub = un % a.bands, bb = min(un / a.bands, a.B - 1);
                    const int uy0 = ub * a.band_rows, uy1 = min(uy0 + a.band_rows, a.H);
                    const float4 v = sld4(obuf + 4 * f);
                    if (un < a.units && uy0 + k <= uy1 && ox < Wo) {
                        const int oy = (uy0 + k - 2) >> 1;
                        sst4(a.out + (long)bb * a.out_fs + ((long)oy * Wo + ox) * Co + hf * C + 4 * qd, v);
                    }
                }
                wave_sync();
            }
        }
    };
    auto step = [&](int t, v2f (&aPN)[CQ][2], v2f (&aC)[CQ][2]) {
        v2f (&tacc)[CQ][2] = acc0;
        float4 (&tmx)[CQ] = onext;
        const int c = c_first + t;  // row consumed in this step; the row it completes is c - 1
        const int e = c - 1;
        bool hand_over = false;     // this wave has a finished row for the next block
        float4 o[CQ];
        if (!tail && active && c >= lo_j - 1 && c <= hi_j) {
            const float* me;
            int ps;
            if (role == 0) {
                if (c + 1 <= hi_j) wait_vm<NL>();
                else wait_vm<0>();
                fix_row(c, t & 1);
                wave_sync();
                me = scratch + (t & 1) * BUF_F + lane * C;
                ps = C;
            } else {
                me = lds + ((role - 1) * SLOTS + ((t - 1) & 1)) * IMG + img_p + lane * PXS;
                ps = PXS;
            }
            strip_row1m<CQ>(me, ps, cst, has_res, aPN, aC, onext, oacc, wa, cb, wd);
            wave_sync();
            if (e >= lo_j && e < hi_j) {
                strip_act<CQ, RELU>(oacc, cst + K::OFF_SLOPE, hi, o);
                if (NH2 || role < S - 1) {
                    hand_over = true;
                } else {
                    float* obuf = scratch;
#pragma unroll
                    for (int q = 0; q < CQ; q++) sst4(obuf + lane * C + 4 * q, o[q]);
                    wave_sync();
                    float* dst = out + (long)e * a.W * C + gout;
                    if (full) {
#pragma unroll
                        for (int k = 0; k < CQ; k++) sst4(dst + 256 * k, sld4(obuf + 4 * lane + 256 * k));
                    } else {
#pragma unroll
                        for (int k = 0; k < CQ; k++) {
                            const float4 v = sld4(obuf + 4 * lane + 256 * k);
                            if (lane + 64 * k < npx * CQ) sst4(dst + 256 * k, v);
                        }
                    }
                }
            }
            if (role == 0 && c + 2 <= hi_j) {
                wave_sync();
                issue_row(c + 2, t & 1);
            }
        }
        if constexpr (NH2 > 0) {
            if (tail) tail_step(t, tacc, tmx);
        }
        if (hand_over) {
            // hand the row to the next block; rows outside the image are that block's zero padding
            float* dstl = lds + (role * SLOTS + (t & 1)) * IMG + img_p + (1 + lane) * PXS;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 0 || e >= a.H) {  // wave-uniform
#pragma unroll
                for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, z);
            } else if (full) {
#pragma unroll
                for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, o[q]);
            } else {
#pragma unroll
                for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, lane >= npx ? z : o[q]);
            }
        }
        wg_barrier();
    };
    const int T = NH2 ? a.band_rows + 3 * S + 1 : a.band_rows + 3 * S - 1;
    for (int t = 0; t < T; t += 2) {
        step(t, acc0, acc1);
        if (t + 1 >= T) break;
        step(t + 1, acc1, acc0);
    }
}

// Same pipeline with TWO rows per step (strip_row2): block j consumes the row pair its predecessor finished in the previous
// step.  The rings hold one pair of row images each (single-buffered: a step is compute | barrier | hand over | barrier),
// block 0 keeps two pairs of DMA buffers per wave.  Needs even band starts / lengths.
template <int CQ, int KB, bool RELU, int NH2>
__global__ __launch_bounds__(NH2 ? 128 * (KB - 1) + 64 * NH2 : 128 * KB, 2) void strip_pipe2_kernel(PipeArgs a) {
    using K = SK<CQ>;
    constexpr int C = K::C, NL = K::NL, BUF_F = K::BUF_F;
    constexpr int S = NH2 ? KB - 1 : KB;         // stride-1 blocks
    constexpr int NT = 128 * S + 64 * NH2;       // threads
    constexpr int IMG = 132 * C;                 // floats of one full-width row image (<= 130 pixels used)
    constexpr int RING_F = (KB - 1) * 2 * IMG;   // KB-1 hand-over rings of one pair of row images
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool tail = NH2 && w >= 2 * S;
    const int p = tail ? 0 : (w & 1), role = tail ? S : (w >> 1);
    const int hf = tail ? w - 2 * S : 0;
    // Static issue priority for the two ends of the pipeline: block 0's waves (they also wait for and issue the DMA) and the
    // waves of the last stage (the last block stores to memory; a stride-2 tail wave serves both strips alone).  A step ends
    // when its slowest wave reaches the barrier, and these are the slowest: measured on the harness (tools/strip_bench.hip)
    // -1.5 % (4 blocks at 128^2), -3 % (64^2), -6 % / -7 % (with a stride-2 tail); any other choice of waves is neutral or slower.
    if (role == 0 || tail || (NH2 == 0 && role == S - 1)) __builtin_amdgcn_s_setprio(1);
    // wave-private scratch: block 0 waves own two pairs of DMA row buffers each, the waves that store one transposition buffer
    float* scratch = lds + RING_F + (role == 0 ? p * 4 * BUF_F : 8 * BUF_F + (tail ? hf : p) * BUF_F);
    const int unit = a.strips == 2 ? (int)blockIdx.x : 2 * (int)blockIdx.x + p;
    const bool active = unit < a.units;
    const int band = unit % a.bands, b = min(unit / a.bands, a.B - 1);
    const int x0 = a.strips == 2 ? 64 * p : 0;
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.H);
    const float* in = a.in + (long)b * a.in_fs;
    float* out = a.out + (long)b * a.out_fs;
    const cfloat* cst = (const cfloat*)a.consts[role];
    const bool has_res = a.has_res[role] != 0;
    const float hi = a.hi[role];
    // the last stride-1 block produces rows [y0, hi_last): two more rows when a stride-2 tail follows (its third tap row,
    // and one to keep the count even)
    const int hi_last = NH2 ? y1 + 2 : y1;
    const int lo_j = y0 - (S - 1 - role), hi_j = hi_last + (S - 1 - role);  // this (stride-1) block produces rows [lo_j, hi_j)
    const int P = (hi_j - lo_j + 2) >> 1;                                   // row pairs it consumes: (lo_j - 1 + 2m, lo_j + 2m)
    const int img_p = a.strips == 2 ? 64 * C * p : 66 * C * p;

    for (int i = threadIdx.x; i < RING_F / 4; i += NT) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    int goff[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int f = min(lane + 64 * k, K::NF - 1), px = f / CQ, qd = f - px * CQ;
        goff[k] = (min(max(x0 - 1 + px, 0), a.W - 1) * C + 4 * qd) * 4;
    }
    const int npx = min(64, a.W - x0);
    const bool full = npx == 64;
    const int zl = x0 == 0 ? 0 : -1;
    const int zr = x0 + 64 >= a.W ? npx + 1 : -1;
    const long gout = (long)x0 * C + 4 * lane;
    const unsigned lds_scratch = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)scratch);
    auto issue_row = [&](int r, int bi) {
        const char* src = reinterpret_cast<const char*>(in + (long)min(max(r, 0), a.H - 1) * a.W * C);
        const unsigned dstb = lds_scratch + (unsigned)(bi * BUF_F * 4);
        constexpr int TAIL = K::NF - 64 * (NL - 1);
#pragma unroll
        for (int k = 0; k < NL; k++) {
            if (k < NL - 1 || TAIL == 64) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dstb + 1024u * k), "v"(goff[k]), "s"(src) : "memory", "m0");
            } else {
                unsigned long long saved;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(dstb + 1024u * k), "v"(goff[k]), "s"(src), "n"((1ull << (TAIL & 63)) - 1) : "memory", "m0");
            }
        }
    };
    const int zfix = lane < CQ ? (zl >= 0 ? zl * C + 4 * lane : -1) : (lane < 2 * CQ ? (zr >= 0 ? zr * C + 4 * (lane - CQ) : -1) : -1);
    auto fix_row = [&](int r, int bi) {
        float* buf = scratch + bi * BUF_F;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < 0 || r >= a.H) {
#pragma unroll
            for (int k = 0; k < NL; k++)
                if (64 * (k + 1) <= K::NF || lane < K::NF - 64 * k) sst4(buf + 4 * (lane + 64 * k), z);
        } else if (zfix >= 0) {
            sst4(buf + zfix, z);
        }
    };

    v2f aA[CQ][2], aB[CQ][2], oacc0[2 * CQ], oacc1[2 * CQ];
    float4 xa[CQ], xb[CQ];
#pragma unroll
    for (int q = 0; q < CQ; q++) {
        aA[q][0] = aA[q][1] = aB[q][0] = aB[q][1] = v2f{0.f, 0.f};
        xa[q] = xb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (role == 0 && active) {
        issue_row(lo_j - 1, 0);
        issue_row(lo_j, 1);
        if (P > 1) {
            issue_row(lo_j + 1, 2);
            issue_row(lo_j + 2, 3);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();

#ifdef MI_PIPE_STAMPS
    unsigned long long ps_acc[4] = {0, 0, 0, 0}, ps_prev = __builtin_amdgcn_s_memtime();
#define MI_PSTAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); ps_acc[k] += t_ - ps_prev; ps_prev = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_PSTAMP(k)
#endif
    auto wg_barrier = [&]() {
        // raw s_barrier with an LDS-only wait, so that the stores of the last block and the DMA of block 0 stay in flight across it
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
#ifndef MI_ABL_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
    };
    auto store_row = [&](int e, const float4 (&o)[CQ]) {  // transposed through LDS: 1 KiB of consecutive bytes per store instruction
        float* obuf = scratch;
#pragma unroll
        for (int q = 0; q < CQ; q++) sst4(obuf + lane * C + 4 * q, o[q]);
        wave_sync();
        float* dst = out + (long)e * a.W * C + gout;
        if (full) {
#pragma unroll
            for (int k = 0; k < CQ; k++) sst4(dst + 256 * k, sld4(obuf + 4 * lane + 256 * k));
        } else {
#pragma unroll
            for (int k = 0; k < CQ; k++) {
                const float4 v = sld4(obuf + 4 * lane + 256 * k);
                if (lane + 64 * k < npx * CQ) sst4(dst + 256 * k, v);
            }
        }
        wave_sync();
    };
    auto hand_row = [&](int e, int which, const float4 (&o)[CQ]) {  // rows outside the image are the next block's zero padding
        float* dstl = lds + (role * 2 + which) * IMG + img_p + (1 + lane) * C;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < 0 || e >= a.H) {  // wave-uniform
#pragma unroll
            for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, z);
        } else if (full) {
#pragma unroll
            for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, o[q]);
        } else {
#pragma unroll
            for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, lane >= npx ? z : o[q]);
        }
    };
    // ---- stride-2 tail: consumes the pair (even row, odd row) the last stride-1 block handed over in the previous step
    auto tail_step = [&](int t, v2f (&tacc)[CQ][2], float4 (&tmx)[CQ]) {
        if constexpr (NH2 > 0) {
            const int mt = t - 2 * S;
            if (mt < 0 || 2 * mt > a.band_rows) return;
            const int k = 2 * mt;  // band-relative index of the even row
            constexpr int Co = NH2 * C;
            constexpr int ST2 = (32 + 2 * Co + 15) / 16 * 16, OFF_SLOPE2 = (C / 2) * ST2 + 64;
            const cfloat* cst2 = (const cfloat*)a.consts[S];
            const float* img = lds + ((S - 1) * 2) * IMG;
            const int tu = a.strips == 2 ? 0 : lane >> 5, tox = a.strips == 2 ? lane : lane & 31;
            const float* me2 = img + tu * 66 * C + (2 * tox + 1) * C;
            const bool skip = hf == 0 && a.has_res[S] != 0;
            if (mt == 0) {
                strip_row_s2<CQ, NH2, 1>(me2, cst2, hf, skip, tacc, tmx, oacc0);
            } else {
                strip_row_s2<CQ, NH2, 2>(me2, cst2, hf, skip, tacc, tmx, oacc0);
                float4 o[CQ];
                strip_act<CQ, RELU>(oacc0, cst2 + OFF_SLOPE2 + hf * C, a.hi[S], o);
                float* obuf = scratch;
#pragma unroll
                for (int q = 0; q < CQ; q++) sst4(obuf + lane * C + 4 * q, o[q]);
                wave_sync();
                const int Wo = a.W >> 1;
#pragma unroll
                for (int j = 0; j < CQ; j++) {
                    const int f = lane + 64 * j, px = f / CQ, qd = f - px * CQ;
                    const int u = a.strips == 2 ? 0 : px >> 5, ox = a.strips == 2 ? px : px & 31;
                    const int un = a.strips == 2 ? (int)blockIdx.x : 2 * (int)blockIdx.x + u;
                    const int ub = un % a.bands, bb = min(un / a.bands, a.B - 1);
                    const int uy0 = ub * a.band_rows, uy1 = min(uy0 + a.band_rows, a.H);
                    const float4 v = sld4(obuf + 4 * f);
                    if (un < a.units && uy0 + k <= uy1 && ox < Wo) {
                        const int oy = (uy0 + k - 2) >> 1;
                        sst4(a.out + (long)bb * a.out_fs + ((long)oy * Wo + ox) * Co + hf * C + 4 * qd, v);
                    }
                }
                wave_sync();
            }
            if (k + 1 < a.band_rows) strip_row_s2<CQ, NH2, 0>(me2 + IMG, cst2, hf, skip, tacc, tmx, oacc0);
        }
    };
    auto step = [&](int t, const float4 (&xin)[CQ], float4 (&xout)[CQ], v2f (&tacc)[CQ][2], float4 (&tmx)[CQ]) {
        const int m = t - 2 * role;  // pair index of this block in this step
        const int c0 = lo_j - 1 + 2 * m;
        bool hand_over = false, store_out = false;
        float4 o0[CQ], o1[CQ];
        if (!tail && active && m >= 0 && m < P) {
            const float *me0, *me1;
            if (role == 0) {
                if (m + 1 < P) wait_vm<2 * NL>();
                else wait_vm<0>();
                fix_row(c0, (m & 1) * 2);
                fix_row(c0 + 1, (m & 1) * 2 + 1);
                wave_sync();
                me0 = scratch + ((m & 1) * 2) * BUF_F + lane * C;
                me1 = me0 + BUF_F;
            } else {
                me0 = lds + ((role - 1) * 2) * IMG + img_p + lane * C;
                me1 = me0 + IMG;
            }
            strip_row2<CQ>(me0, me1, cst, has_res, aA, aB, xin, xout, oacc0, oacc1);
            wave_sync();
            if (m >= 1) {  // rows c0 - 1 and c0 are finished
                strip_act<CQ, RELU>(oacc0, cst + K::OFF_SLOPE, hi, o0);
                strip_act<CQ, RELU>(oacc1, cst + K::OFF_SLOPE, hi, o1);
                if (NH2 || role < S - 1) hand_over = true;
#ifdef MI_ABL_STORE_EARLY
                else { store_row(c0 - 1, o0); store_row(c0, o1); }
#else
                else store_out = true;  // the last block's rows go to memory in the hand-over phase, while the others write their rings
#endif
            }
            if (role == 0 && m + 2 < P) {
                wave_sync();
                issue_row(c0 + 4, (m & 1) * 2);
                issue_row(c0 + 5, (m & 1) * 2 + 1);
            }
        }
        if constexpr (NH2 > 0) {
            if (tail) tail_step(t, tacc, tmx);
        }
        MI_PSTAMP(0)
        wg_barrier();  // every reader of the previous pair is done
        MI_PSTAMP(1)
        if (hand_over) {
            hand_row(c0 - 1, 0, o0);
            hand_row(c0, 1, o1);
        } else if (store_out) {
            store_row(c0 - 1, o0);
            store_row(c0, o1);
        }
        MI_PSTAMP(2)
        wg_barrier();
        MI_PSTAMP(3)
    };
    const int T = NH2 ? 2 * S + a.band_rows / 2 + 1 : 2 * S - 1 + a.band_rows / 2;
    for (int t = 0; t < T; t += 2) {
        step(t, xa, xb, aA, xa);
        if (t + 1 >= T) break;
        step(t + 1, xb, xa, aA, xa);
    }
#ifdef MI_PIPE_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long* d = a.stamps + ((size_t)blockIdx.x * 8 + w) * 8;
        for (int k = 0; k < 4; k++) d[k] = ps_acc[k];
        d[4] = T;
    }
#endif
#undef MI_PSTAMP
}

// strip_pipe2_kernel with the pointwise convs of the stride-1 blocks on the matrix cores (strip_row2m): the filters live in registers, the
// only scalar stream left is the depthwise taps (a stage ahead; the bias and the first stage's taps are asked for BEFORE the barrier that
// ends the previous step), and the hand-over rings have a pixel stride of C + 4 floats, which makes their 16-byte reads and writes
// free of bank conflicts (lane stride 24 floats: lanes l and l + 8 meet in the same banks; 28: the 16 lanes of a pass cover all 64).
// Bit-identical to strip_pipe2_kernel (tools/strip_bench.hip compares them).
template <int CQ, int KB, bool RELU, int NH2>
__global__ __launch_bounds__(NH2 ? 128 * (KB - 1) + 64 * NH2 : 128 * KB, 2) void strip_pipe2m_kernel(PipeArgs a) {
    using K = SK<CQ>;
    constexpr int C = K::C, NL = K::NL, BUF_F = K::BUF_F;
    constexpr int S = NH2 ? KB - 1 : KB;         // stride-1 blocks
    constexpr int NT = 128 * S + 64 * NH2;       // threads
    constexpr int PXS = C + 4;                   // pixel stride of a hand-over row image (floats)
    constexpr int IMG = 132 * PXS;               // floats of one full-width row image (<= 130 pixels used)
    constexpr int RING_F = (KB - 1) * 2 * IMG;   // KB-1 hand-over rings of one pair of row images
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool tail = NH2 && w >= 2 * S;
    const int p = tail ? 0 : (w & 1), role = tail ? S : (w >> 1);
    const int hf = tail ? w - 2 * S : 0;
    // Static issue priority for the two ends of the pipeline: block 0's waves (they also wait for and issue the DMA) and the
    // waves of the last stage (the last block stores to memory; a stride-2 tail wave serves both strips alone).  A step ends
    // when its slowest wave reaches the barrier, and these are the slowest: measured on the harness (tools/strip_bench.hip)
    // -1.5 % (4 blocks at 128^2), -3 % (64^2), -6 % / -7 % (with a stride-2 tail); any other choice of waves is neutral or slower.
    {
        const bool first = role == 0, last = tail || (NH2 == 0 && role == S - 1);
        if ((a.prio == 1 && (first || last)) || (a.prio == 2 && first) || (a.prio == 3 && last)) __builtin_amdgcn_s_setprio(1);
    }
    // wave-private scratch: block 0 waves own two pairs of DMA row buffers each, the waves that store one transposition buffer
    float* scratch = lds + RING_F + (role == 0 ? p * 4 * BUF_F : 8 * BUF_F + (tail ? hf : p) * BUF_F);
    const int unit = a.strips == 2 ? (int)blockIdx.x : 2 * (int)blockIdx.x + p;
    const bool active = unit < a.units;
    const int band = unit % a.bands, b = min(unit / a.bands, a.B - 1);
    const int x0 = a.strips == 2 ? 64 * p : 0;
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.H);
    const float* in = a.in + (long)b * a.in_fs;
    float* out = a.out + (long)b * a.out_fs;
    const cfloat* cst = (const cfloat*)a.consts[role];
    const bool has_res = a.has_res[role] != 0;
    const float hi = a.hi[role];
    // the last stride-1 block produces rows [y0, hi_last): two more rows when a stride-2 tail follows (its third tap row,
    // and one to keep the count even)
    const int hi_last = NH2 ? y1 + 2 : y1;
    const int lo_j = y0 - (S - 1 - role), hi_j = hi_last + (S - 1 - role);  // this (stride-1) block produces rows [lo_j, hi_j)
    const int P = (hi_j - lo_j + 2) >> 1;                                   // row pairs it consumes: (lo_j - 1 + 2m, lo_j + 2m)
    const int img_p = a.strips == 2 ? 64 * PXS * p : 66 * PXS * p;

    for (int i = threadIdx.x; i < RING_F / 4; i += NT) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    int goff[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int f = min(lane + 64 * k, K::NF - 1), px = f / CQ, qd = f - px * CQ;
        goff[k] = (min(max(x0 - 1 + px, 0), a.W - 1) * C + 4 * qd) * 4;
    }
    const int npx = min(64, a.W - x0);
    const bool full = npx == 64;
    const int zl = x0 == 0 ? 0 : -1;
    const int zr = x0 + 64 >= a.W ? npx + 1 : -1;
    const long gout = (long)x0 * C + 4 * lane;
    const unsigned lds_scratch = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)scratch);
    auto issue_row = [&](int r, int bi) {
        const char* src = reinterpret_cast<const char*>(in + (long)min(max(r, 0), a.H - 1) * a.W * C);
        const unsigned dstb = lds_scratch + (unsigned)(bi * BUF_F * 4);
        constexpr int TAIL = K::NF - 64 * (NL - 1);
#pragma unroll
        for (int k = 0; k < NL; k++) {
            if (k < NL - 1 || TAIL == 64) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dstb + 1024u * k), "v"(goff[k]), "s"(src) : "memory", "m0");
            } else {
                unsigned long long saved;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(dstb + 1024u * k), "v"(goff[k]), "s"(src), "n"((1ull << (TAIL & 63)) - 1) : "memory", "m0");
            }
        }
    };
    const int zfix = lane < CQ ? (zl >= 0 ? zl * C + 4 * lane : -1) : (lane < 2 * CQ ? (zr >= 0 ? zr * C + 4 * (lane - CQ) : -1) : -1);
    auto fix_row = [&](int r, int bi) {
        float* buf = scratch + bi * BUF_F;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < 0 || r >= a.H) {
#pragma unroll
            for (int k = 0; k < NL; k++)
                if (64 * (k + 1) <= K::NF || lane < K::NF - 64 * k) sst4(buf + 4 * (lane + 64 * k), z);
        } else if (zfix >= 0) {
            sst4(buf + zfix, z);
        }
    };

    v2f aA[CQ][2], aB[CQ][2];
    v4f oacc0[CQ], oacc1[CQ];
    v4f toacc[CQ];       // stride-2 tail wave: accumulators (its partial depthwise row / running max-pool live in aA / onext)
    float4 onext[CQ];    // stride-1 waves: bias + skip of the next step's first finished row
    float wa[K::NA];     // pointwise filter, A operands (SK::OFF_A)
    float cb[C], wd[2][18];
#pragma unroll
    for (int q = 0; q < CQ; q++) {
        aA[q][0] = aA[q][1] = aB[q][0] = aB[q][1] = v2f{0.f, 0.f};
        onext[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // bias and first-stage taps of the coming step: scalar loads issued before a barrier, complete when it opens
    auto prefetch_consts = [&]() {  // (a stride-2 tail wave reads the same offsets of its own, larger blob and ignores them)
        const cfloat* pc = cst;
        asm volatile("" : "+s"(pc));
#pragma unroll
        for (int i = 0; i < C; i++) cb[i] = pc[K::OFF_BIAS + i];
#pragma unroll
        for (int i = 0; i < 18; i++) wd[0][i] = pc[K::OFF_DW + i];
    };
    auto pin_consts = [&]() {  // behind the barrier's lgkmcnt(0): keeps the loads above in front of it
#pragma unroll
        for (int i = 0; i < C; i++) asm volatile("" : "+s"(cb[i]));
#pragma unroll
        for (int i = 0; i < 18; i++) asm volatile("" : "+s"(wd[0][i]));
    };
    if (!tail) {
#pragma unroll
        for (int i = 0; i < K::NA; i++) wa[i] = a.consts[role][K::OFF_A + 64 * i + lane];
    } else if constexpr (NH2 > 0) {  // this tail wave's C x C slice of the stride-2 block's filter (strip_pack_consts_s2)
        constexpr int ST2h = (32 + 2 * NH2 * C + 15) / 16 * 16, OFF_A2 = (C / 2) * ST2h + 128;
#pragma unroll
        for (int i = 0; i < K::NA; i++) wa[i] = a.consts[S][OFF_A2 + (hf * K::NA + i) * 64 + lane];
    }
    prefetch_consts();
    if (role == 0 && active) {
        issue_row(lo_j - 1, 0);
        issue_row(lo_j, 1);
        if (P > 1) {
            issue_row(lo_j + 1, 2);
            issue_row(lo_j + 2, 3);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    pin_consts();
    __builtin_amdgcn_s_barrier();
    if (!tail && !has_res) {  // no skip: the first finished row of every step starts from the bias alone
#pragma unroll
        for (int q = 0; q < CQ; q++) onext[q] = make_float4(cb[4 * q], cb[4 * q + 1], cb[4 * q + 2], cb[4 * q + 3]);
    }

#ifdef MI_PIPE_STAMPS
    unsigned long long ps_acc[7] = {0, 0, 0, 0, 0, 0, 0}, ps_prev = __builtin_amdgcn_s_memtime();
#define MI_PSTAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); ps_acc[k] += t_ - ps_prev; ps_prev = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define MI_PSTAMP(k)
#endif
    auto wg_barrier = [&]() {
        // raw s_barrier with an LDS-only wait, so that the stores of the last block and the DMA of block 0 stay in flight across it
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
#ifndef MI_ABL_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
    };
    auto store_row = [&](int e, const float4 (&o)[CQ]) {  // transposed through LDS: 1 KiB of consecutive bytes per store instruction
        float* obuf = scratch;
#pragma unroll
        for (int q = 0; q < CQ; q++) sst4(obuf + lane * C + 4 * q, o[q]);
        wave_sync();
        float* dst = out + (long)e * a.W * C + gout;
        if (full) {
#pragma unroll
            for (int k = 0; k < CQ; k++) sst4(dst + 256 * k, sld4(obuf + 4 * lane + 256 * k));
        } else {
#pragma unroll
            for (int k = 0; k < CQ; k++) {
                const float4 v = sld4(obuf + 4 * lane + 256 * k);
                if (lane + 64 * k < npx * CQ) sst4(dst + 256 * k, v);
            }
        }
        wave_sync();
    };
    auto hand_row = [&](int e, int which, const float4 (&o)[CQ]) {  // rows outside the image are the next block's zero padding
        float* dstl = lds + (role * 2 + which) * IMG + img_p + (1 + lane) * PXS;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < 0 || e >= a.H) {  // wave-uniform
#pragma unroll
            for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, z);
        } else if (full) {
#pragma unroll
            for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, o[q]);
        } else {
#pragma unroll
            for (int q = 0; q < CQ; q++) sst4(dstl + 4 * q, lane >= npx ? z : o[q]);
        }
    };
    // ---- stride-2 tail: consumes the pair (even row, odd row) the last stride-1 block handed over in the previous step
    auto tail_step = [&](int t, v2f (&tacc)[CQ][2], float4 (&tmx)[CQ]) {
        if constexpr (NH2 > 0) {
            const int mt = t - 2 * S;
            if (mt < 0 || 2 * mt > a.band_rows) return;
            const int k = 2 * mt;  // band-relative index of the even row
            constexpr int Co = NH2 * C;
            constexpr int ST2 = (32 + 2 * Co + 15) / 16 * 16, OFF_SLOPE2 = (C / 2) * ST2 + 64;
            const cfloat* cst2 = (const cfloat*)a.consts[S];
            const float* img = lds + ((S - 1) * 2) * IMG;
            const int tu = a.strips == 2 ? 0 : lane >> 5, tox = a.strips == 2 ? lane : lane & 31;
            const float* me2 = img + tu * 66 * PXS + (2 * tox + 1) * PXS;
            const bool skip = hf == 0 && a.has_res[S] != 0;
            if (mt == 0) {
                strip_row_s2m<CQ, NH2, 1, PXS>(me2, cst2, hf, skip, tacc, tmx, toacc, wa);
            } else {
                strip_row_s2m<CQ, NH2, 2, PXS>(me2, cst2, hf, skip, tacc, tmx, toacc, wa);
                float4 o[CQ];
                strip_act<CQ, RELU>(toacc, cst2 + OFF_SLOPE2 + hf * C, a.hi[S], o);
                float* obuf = scratch;
#pragma unroll
                for (int q = 0; q < CQ; q++) sst4(obuf + lane * C + 4 * q, o[q]);
                wave_sync();
                const int Wo = a.W >> 1;
#pragma unroll
                for (int j = 0; j < CQ; j++) {
                    const int f = lane + 64 * j, px = f / CQ, qd = f - px * CQ;
                    const int u = a.strips == 2 ? 0 : px >> 5, ox = a.strips == 2 ? px : px & 31;
                    const int un = a.strips == 2 ? (int)blockIdx.x : 2 * (int)blockIdx.x + u;
                    const int ub = un % a.bands, bb = min(un / a.bands, a.B - 1);
                    const int uy0 = ub * a.band_rows, uy1 = min(uy0 + a.band_rows, a.H);
                    const float4 v = sld4(obuf + 4 * f);
                    if (un < a.units && uy0 + k <= uy1 && ox < Wo) {
                        const int oy = (uy0 + k - 2) >> 1;
                        sst4(a.out + (long)bb * a.out_fs + ((long)oy * Wo + ox) * Co + hf * C + 4 * qd, v);
                    }
                }
                wave_sync();
            }
            if (k + 1 < a.band_rows) strip_row_s2m<CQ, NH2, 0, PXS>(me2 + IMG, cst2, hf, skip, tacc, tmx, toacc, wa);
        }
    };
    auto step = [&](int t) {
        v2f (&tacc)[CQ][2] = aA;
        float4 (&tmx)[CQ] = onext;
        const int m = t - 2 * role;  // pair index of this block in this step
        const int c0 = lo_j - 1 + 2 * m;
        bool hand_over = false;
        float4 o0[CQ], o1[CQ];
        if (!tail && active && m >= 0 && m < P) {
            const float *me0, *me1;
            int ps;
            if (role == 0) {
                if (m + 1 < P) wait_vm<2 * NL>();
                else wait_vm<0>();
                fix_row(c0, (m & 1) * 2);
                fix_row(c0 + 1, (m & 1) * 2 + 1);
                wave_sync();
                me0 = scratch + ((m & 1) * 2) * BUF_F + lane * C;
                me1 = me0 + BUF_F;
                ps = C;
            } else {
                me0 = lds + ((role - 1) * 2) * IMG + img_p + lane * PXS;
                me1 = me0 + IMG;
                ps = PXS;
            }
            MI_PSTAMP(4)
            strip_row2m<CQ>(me0, me1, ps, cst, has_res, aA, aB, onext, oacc0, oacc1, wa, cb, wd);
            wave_sync();
            MI_PSTAMP(5)
            if (m >= 1) {  // rows c0 - 1 and c0 are finished
                strip_act<CQ, RELU>(oacc0, cst + K::OFF_SLOPE, hi, o0);
                strip_act<CQ, RELU>(oacc1, cst + K::OFF_SLOPE, hi, o1);
                if (NH2 || role < S - 1) hand_over = true;
                // the last block's rows go to memory at once: its waves run at issue priority and reach the barrier thousands of cycles
                // before their SIMD partners, and its transposition buffer is private — stored here, their 24 KB of LDS writes are out of
                // the hand-over phase, which is bound by the LDS write rate (harness: 0.354 -> 0.345 ms)
                else { store_row(c0 - 1, o0); store_row(c0, o1); }
            }
            if (role == 0 && m + 2 < P) {
                wave_sync();
                issue_row(c0 + 4, (m & 1) * 2);
                issue_row(c0 + 5, (m & 1) * 2 + 1);
            }
        }
        if constexpr (NH2 > 0) {
            if (tail) tail_step(t, tacc, tmx);
        }
        MI_PSTAMP(6)
#ifdef MI_ABL_M_ONEBAR  // timing ablation (development harness only; racy, wrong values): what double-buffered rings would allow — the hand-over
                        // writes right behind the wave's own compute phase and ONE barrier per step
        if (hand_over) {
            hand_row(c0 - 1, 0, o0);
            hand_row(c0, 1, o1);
        }
        MI_PSTAMP(1)
        MI_PSTAMP(2)
#else
        wg_barrier();  // every reader of the previous pair is done
        MI_PSTAMP(1)
        if (hand_over) {
            hand_row(c0 - 1, 0, o0);
            hand_row(c0, 1, o1);
        }
        MI_PSTAMP(2)
#endif
        prefetch_consts();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        pin_consts();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        MI_PSTAMP(3)
    };
    const int T = NH2 ? 2 * S + a.band_rows / 2 + 1 : 2 * S - 1 + a.band_rows / 2;
    for (int t = 0; t < T; t++) step(t);
#ifdef MI_PIPE_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long* d = a.stamps + ((size_t)blockIdx.x * 8 + w) * 8;
        d[0] = ps_acc[4] + ps_acc[5] + ps_acc[6];   // compute = before the row function + the row function + behind it
        for (int k = 1; k < 4; k++) d[k] = ps_acc[k];
        d[4] = T; d[5] = ps_acc[4]; d[6] = ps_acc[5]; d[7] = ps_acc[6];
    }
#endif
#undef MI_PSTAMP
}

template <int CQ, int KB, bool RELU, int NH2>
int launch_pipe_inst(const PipeArgs& pa, hipStream_t s) {
    using K = SK<CQ>;
    const bool two = pa.rows_per_step >= 2;
    if (pa.rows_per_step == 4) {  // one row per step, MFMA pointwise convs
        auto k1 = strip_pipe1m_kernel<CQ, KB, RELU, NH2>;
        const size_t lds1 = (size_t)((KB - 1) * 2 * 132 * (K::C + 4) + 6 * K::BUF_F) * 4;
        if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(k1)); e != hipSuccess) return (int)e;
        const int wgs1 = pa.strips == 2 ? pa.units : (pa.units + 1) / 2;
        return (int)launch_kernel(k1, dim3((unsigned)wgs1), dim3(NH2 ? 128 * (KB - 1) + 64 * NH2 : 128 * KB), lds1, s, pa);
    }
    const bool mf = pa.rows_per_step == 3;
    auto kern = mf ? strip_pipe2m_kernel<CQ, KB, RELU, NH2> : (two ? strip_pipe2_kernel<CQ, KB, RELU, NH2> : strip_pipe_kernel<CQ, KB, RELU, NH2>);
    const size_t lds_bytes = (size_t)((KB - 1) * 2 * 132 * (mf ? K::C + 4 : K::C) + (two ? 10 : 6) * K::BUF_F) * 4;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    const int wgs = pa.strips == 2 ? pa.units : (pa.units + 1) / 2;
    return (int)launch_kernel(kern, dim3((unsigned)wgs), dim3(NH2 ? 128 * (KB - 1) + 64 * NH2 : 128 * KB), lds_bytes, s, pa);
}

unsigned long long* g_strip_stamps = nullptr;  // set by the development harness (MI_STRIP_STAMPS builds)

int strips_band_rows(const BlockArgs& a, int strips) {
    // about one resident set of waves over the chip (12 per CU), but bands of at least 12 rows (2 halo rows re-read per band; measured on
    // the face mesh's 48x48x32 blocks at 512 frames: 8-row bands 0.110 ms, 12 rows 0.101 ms, 16 rows 0.123 ms)
    static const int forced = getenv("MI_STRIP_BAND") ? atoi(getenv("MI_STRIP_BAND")) : 0;  // tuning aid
    if (forced > 0) return std::min(forced, a.H);
    const long per_row_waves = (long)a.B * strips;
    long bands = std::max<long>(1, (3072 + per_row_waves / 2) / per_row_waves);
    int rows = (int)((a.H + bands - 1) / bands);
    // ... as long as that leaves enough waves to fill the chip: small batches (the 64 ROIs of the device pipeline) take shorter bands
    // (round 5: down to ONE row per band for a handful of frames — a band costs two priming rows, but a batch of one is latency: 128x128x24 at
    // one frame, four blocks: 53.6 us with 4-row bands, 38.9 with 2, 31.4 with 1; at 16 frames 90 / 84 / 91)
    const int floor_rows = per_row_waves * ((a.H + 11) / 12) >= 2048 ? 12
                         : (per_row_waves * ((a.H + 7) / 8) >= 1024 ? 8 : (per_row_waves * ((a.H + 3) / 4) >= 2048 ? 4 : (per_row_waves * ((a.H + 1) / 2) >= 2048 ? 2 : 1)));
    rows = std::max(rows, std::min(a.H, floor_rows));
    return rows;
}

template <int CQ, bool RELU>
int launch_strip_inst(const BlockArgs& a, hipStream_t s) {
    using K = SK<CQ>;
    auto kern = strip_kernel<CQ, RELU>;
    StripArgs sa;
    sa.in = a.in; sa.out = a.out; sa.consts = a.w_strip; sa.in_fs = a.in_fs; sa.out_fs = a.out_fs;
    sa.B = a.B; sa.H = a.H; sa.W = a.W;
    sa.strips = (a.W + 63) / 64;
    sa.band_rows = strips_band_rows(a, sa.strips);
    sa.bands = (a.H + sa.band_rows - 1) / sa.band_rows;
    sa.has_res = a.ep.res_mode == RES_DIRECT;
    sa.hi = a.ep.act == ACT_RELU6 ? 6.f : INFINITY;
    sa.stamps = g_strip_stamps;
    const long waves = (long)a.B * sa.strips * sa.bands;
    const size_t lds_bytes = (size_t)4 * K::WAVE_F * 4;  // 4 waves x 2 row buffers
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(kern, dim3((unsigned)((waves + 3) / 4)), dim3(256), lds_bytes, s, sa);
}

}  // namespace

// Shapes the strip kernel takes: stride-1 depthwise 3x3 (SAME) + pointwise with C = Co in {16, 24, 32}, skip = the
// block's own input (or none), constants packed by strip_pack_consts().
bool strip_kernel_supports(const BlockArgs& a) {
    static const bool off = getenv("MI_NO_STRIP") != nullptr;  // tuning aid: fall back to the LDS-ring block kernel
    if (off || !a.w_strip || !a.has_dw || a.sh != 1 || a.sw != 1 || a.pt != 1 || a.pl != 1) return false;
    if (a.C != a.Co || (a.C != 16 && a.C != 24 && a.C != 32) || a.H != a.Ho || a.W != a.Wo) return false;
    if (a.ep.res_mode != RES_NONE) {
        if (a.ep.res_after) return false;  // skip behind the activation: block / generic kernels only
        if (a.ep.res_mode != RES_DIRECT || a.ep.res != a.in || a.ep.res_fs != a.in_fs || a.ep.res_C != a.C) return false;
    }
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!aligned16(a.in) || !aligned16(a.out) || (a.in_fs & 3) || (a.out_fs & 3)) return false;
    return true;
}

bool strip_shape_ok(int C, int Co) { return C == Co && (C == 16 || C == 24 || C == 32); }

static int strip_stage_floats(int C) { return 32 + 2 * C + (16 - (2 * C) % 16) % 16; }
int strip_consts_floats(int C) { return C / 2 * strip_stage_floats(C) + 64 + 64 * ((C * (C / 4) + 15) / 16); }

// w_dw [3][3][C], b_dw [C] or null, w_pw [Co][C] (TFLite OHWI with H = W = 1), bias [Co] or null, alpha [Co] or null.
void strip_pack_consts(int C, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst) {
    const int SF = strip_stage_floats(C);
    std::fill(dst, dst + strip_consts_floats(C), 0.f);
    for (int st = 0; st < C / 2; st++) {  // input channels (2st, 2st+1)
        float* rec = dst + (size_t)st * SF;
        for (int t = 0; t < 9; t++)
            for (int e = 0; e < 2; e++) rec[2 * t + e] = w_dw[t * C + 2 * st + e];
        for (int half = 0; half < 2; half++)
            for (int o = 0; o < C; o++) rec[32 + half * C + o] = w_pw[(size_t)o * C + 2 * st + half];
    }
    float* pb = dst + (size_t)(C / 2) * SF;
    float* ps = pb + 32;
    for (int c = 0; c < C; c++) {
        // PW(dw + b_dw) + b_pw = PW(dw) + (W b_dw + b_pw): the depthwise bias is folded into the pointwise bias
        double acc = bias ? bias[c] : 0.0;
        if (b_dw)
            for (int k = 0; k < C; k++) acc += (double)w_pw[(size_t)c * C + k] * b_dw[k];
        pb[c] = (float)acc;
        ps[c] = act == ACT_PRELU ? alpha[c] : (act == ACT_NONE ? 1.f : 0.f);
    }
    // A operands of v_mfma_f32_4x4x1_16b_f32 (SK::OFF_A): register n / 16, lanes 4 (n % 16) + i = W[4t + i][k], n = k CQ + t
    float* pa = ps + 32;
    const int CQ = C / 4;
    for (int k = 0; k < C; k++)
        for (int t = 0; t < CQ; t++)
            for (int i = 0; i < 4; i++) {
                const int n = k * CQ + t;
                pa[(n >> 4) * 64 + 4 * (n & 15) + i] = w_pw[(size_t)(4 * t + i) * C + k];
            }
}

const char* strip_kernel_label(const BlockArgs& a, char* buf, size_t cap) {
    snprintf(buf, cap, "strip_kernel<%d,%d>", a.C / 4, a.ep.act == ACT_RELU ? 1 : 0);
    return buf;
}

int launch_strip(const BlockArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const bool relu = a.ep.act == ACT_RELU;
#define MI_STRIP_CASE(Q) \
    if (a.C == 4 * Q) return relu ? launch_strip_inst<Q, true>(a, s) : launch_strip_inst<Q, false>(a, s);
    MI_STRIP_CASE(4) MI_STRIP_CASE(6) MI_STRIP_CASE(8)
#undef MI_STRIP_CASE
    return (int)hipErrorInvalidValue;
}

// Stride-2 block that may end a row pipeline: DW3x3 s2 (SAME on an even size) -> PW C -> Co in {C, 2C} -> + 2x2 max-pool of
// its input (zero channel-padded) -> act; constants packed by strip_pack_consts_s2().
static bool strip_tail_supports(const BlockArgs& t, const BlockArgs& first) {
    if (!t.w_strip || !t.has_dw || t.sh != 2 || t.sw != 2 || t.pt != 0 || t.pl != 0) return false;
    if (t.C != first.C || t.H != first.H || t.W != first.W || (t.H & 1) || (t.W & 1) || t.Ho * 2 != t.H || t.Wo * 2 != t.W) return false;
    if (t.Co != t.C && t.Co != 2 * t.C) return false;
    if (t.ep.res_mode != RES_NONE && (t.ep.res_mode != RES_MAXPOOL || t.ep.res_C != t.C || t.ep.res_H != t.H || t.ep.res_W != t.W)) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return aligned16(t.out) && !(t.out_fs & 3);
}

// Row-pipelined chain of n (2..4) blocks: blocks[k+1] consumes blocks[k]'s output, which never reaches HBM.  All but the
// last are strip-eligible stride-1 blocks; the last may be a stride-2 block (strip_tail_supports).
bool strip_pipe_supports(const BlockArgs* blocks, int n) {
    static const bool off = getenv("MI_NO_PIPE") != nullptr;  // tuning aid
    if (off || n < 2 || n > kMaxPipe) return false;
    const bool tail = blocks[n - 1].sh == 2;
    for (int k = 0; k < n - (tail ? 1 : 0); k++) {
        BlockArgs t = blocks[k];
        if (k > 0) { t.in = blocks[0].in; t.in_fs = blocks[0].in_fs; if (t.ep.res_mode != RES_NONE) { t.ep.res = t.in; t.ep.res_fs = t.in_fs; } }
        if (k < n - 1) { t.out = blocks[0].in == blocks[n - 1].out ? t.out : const_cast<float*>(blocks[0].in); t.out_fs = blocks[0].in_fs; }
        if (!strip_kernel_supports(t)) return false;
        if (t.C != blocks[0].C || t.H != blocks[0].H || t.W != blocks[0].W) return false;
        if ((t.ep.act == ACT_RELU) != (blocks[0].ep.act == ACT_RELU)) return false;
    }
    if (tail) {
        if (!strip_tail_supports(blocks[n - 1], blocks[0])) return false;
        if ((blocks[n - 1].ep.act == ACT_RELU) != (blocks[0].ep.act == ACT_RELU)) return false;
        if (blocks[0].ep.act != ACT_RELU) return false;  // only the ReLU tails are instantiated
        if (blocks[0].C != 24) return false;
    }
    return blocks[0].W <= 128 && blocks[0].C <= 24;
}

bool strip_pipe_shape_ok(int C, int W) { return getenv("MI_NO_PIPE") == nullptr && (C == 16 || C == 24) && W <= 128; }
bool strip_tail_shape_ok(int C, int Co, int H, int W) { return C == 24 && (Co == C || Co == 2 * C) && !(H & 1) && !(W & 1) && W <= 128; }

int strip_consts_s2_floats(int C, int Co) { return C / 2 * ((32 + 2 * Co + 15) / 16 * 16) + 128 + (Co / C) * ((C * (C / 4) + 15) / 16) * 64; }

// constants of a stride-2 tail block: w_dw [3][3][C], b_dw [C] or null, w_pw [Co][C], bias [Co] or null, alpha [Co] or null
void strip_pack_consts_s2(int C, int Co, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst) {
    const int SF = (32 + 2 * Co + 15) / 16 * 16;
    std::fill(dst, dst + strip_consts_s2_floats(C, Co), 0.f);
    for (int st = 0; st < C / 2; st++) {
        float* rec = dst + (size_t)st * SF;
        for (int t = 0; t < 9; t++)
            for (int e = 0; e < 2; e++) rec[2 * t + e] = w_dw[t * C + 2 * st + e];
        for (int half = 0; half < 2; half++)
            for (int o = 0; o < Co; o++) rec[32 + half * Co + o] = w_pw[(size_t)o * C + 2 * st + half];
    }
    float* pb = dst + (size_t)(C / 2) * SF;
    float* ps = pb + 64;
    for (int c = 0; c < Co; c++) {
        double acc = bias ? bias[c] : 0.0;
        if (b_dw)
            for (int k = 0; k < C; k++) acc += (double)w_pw[(size_t)c * C + k] * b_dw[k];
        pb[c] = (float)acc;
        ps[c] = act == ACT_PRELU ? alpha[c] : (act == ACT_NONE ? 1.f : 0.f);
    }
    // A operands of v_mfma_f32_4x4x1_16b_f32 for the tail waves of strip_pipe2m_kernel, one C x C slice per C output channels (hf):
    // register n / 16, lanes 4 (n % 16) + i = W[hf C + 4t + i][k], n = k CQ + t
    const int CQ = C / 4, NA = (C * CQ + 15) / 16;
    float* pa = pb + 128;
    for (int hf = 0; hf < Co / C; hf++)
        for (int k = 0; k < C; k++)
            for (int t = 0; t < CQ; t++)
                for (int i = 0; i < 4; i++) {
                    const int n = k * CQ + t;
                    pa[(size_t)(hf * NA + (n >> 4)) * 64 + 4 * (n & 15) + i] = w_pw[(size_t)(hf * C + 4 * t + i) * C + k];
                }
}

// rows a pipeline step handles: two (strip_pipe2_kernel) whenever the height is even
int strip_pipe_rows_per_step(int H, int hint) {
    static const int rows_forced = getenv("MI_PIPE_ROWS") ? atoi(getenv("MI_PIPE_ROWS")) : 0;  // tuning aid
    static const bool no_mfma = getenv("MI_PIPE_NO_MFMA") != nullptr;                           // tuning aid: packed-FMA pointwise convs
    if (rows_forced == 1 || hint == 1) return 1;
    if (rows_forced == 4 || hint == 4) return 4;            // one row per step, MFMA pointwise convs (any height)
    if (H & 1) return no_mfma ? 1 : 4;
    return (no_mfma || rows_forced == 2 || hint == 2) ? 2 : 3;
}

const char* strip_pipe_label(const BlockArgs* blocks, int n, char* buf, size_t cap) {
    const int nh2 = blocks[n - 1].sh == 2 ? blocks[n - 1].Co / blocks[n - 1].C : 0;
    const int rps = strip_pipe_rows_per_step(blocks[0].H, blocks[0].pipe_rows);
    snprintf(buf, cap, "strip_pipe%s_kernel<%d,%d,%d,%d>", rps == 4 ? "1m" : (rps == 3 ? "2m" : (rps == 2 ? "2" : "")), blocks[0].C / 4, n, blocks[0].ep.act == ACT_RELU ? 1 : 0, nh2);
    return buf;
}

int launch_strip_pipe(const BlockArgs* blocks, int n, void* stream) {
    const BlockArgs& a = blocks[0];
    const int nh2 = blocks[n - 1].sh == 2 ? blocks[n - 1].Co / blocks[n - 1].C : 0;
    PipeArgs pa{};
    pa.in = a.in; pa.out = blocks[n - 1].out; pa.in_fs = a.in_fs; pa.out_fs = blocks[n - 1].out_fs;
    pa.B = a.B; pa.H = a.H; pa.W = a.W;
    pa.strips = (a.W + 63) / 64;
    for (int k = 0; k < n; k++) {
        pa.consts[k] = blocks[k].w_strip;
        pa.has_res[k] = blocks[k].ep.res_mode != RES_NONE;
        pa.hi[k] = blocks[k].ep.act == ACT_RELU6 ? 6.f : INFINITY;
    }
    // bands: about one resident set of workgroups over the chip, but not so short that the pipeline fill steps dominate
    static const int forced = getenv("MI_PIPE_BAND") ? atoi(getenv("MI_PIPE_BAND")) : 0;  // tuning aid
    static const int per_cu_forced = getenv("MI_PIPE_PERCU") ? atoi(getenv("MI_PIPE_PERCU")) : 0;  // tuning aid
    const int per_cu = per_cu_forced > 0 ? per_cu_forced : (n <= 2 ? 2 : 1);
    const long wg_units = pa.strips == 2 ? 1 : 2;
    long bands = std::max<long>(1, (256L * per_cu * wg_units + a.B / 2) / std::max(1, a.B));
    int rows = (int)((a.H + bands - 1) / bands);
    rows = std::max(rows, std::min(a.H, 8 * n));
    if (a.pipe_band > 0) rows = std::min(a.pipe_band, a.H);   // option "pipe_band": with another batch in flight long bands win (fewer fill steps; the idle CUs are the other batch's)
    if (forced > 0) rows = std::min(forced, a.H);
    pa.rows_per_step = strip_pipe_rows_per_step(a.H, a.pipe_rows);
    if (nh2 || pa.rows_per_step == 2 || pa.rows_per_step == 3) rows = std::min(a.H, (rows + 1) & ~1);  // bands that start on even rows
    pa.band_rows = rows;
    pa.bands = (a.H + rows - 1) / rows;
    pa.units = a.B * pa.bands;
    static const int prio_forced = getenv("MI_PIPE_PRIO") ? atoi(getenv("MI_PIPE_PRIO")) : -1;  // tuning aid
    pa.prio = prio_forced >= 0 ? prio_forced : 1;
#ifdef MI_PIPE_STAMPS
    pa.stamps = g_strip_stamps;
#endif
    hipStream_t s = (hipStream_t)stream;
    const bool relu = a.ep.act == ACT_RELU;
#define MI_PIPE_CASE(Q, KBV) \
    if (a.C == 4 * Q && n == KBV && nh2 == 0) return relu ? launch_pipe_inst<Q, KBV, true, 0>(pa, s) : launch_pipe_inst<Q, KBV, false, 0>(pa, s);
#ifdef MI_DEV_ONE  // development builds (tools/isa_report.py, quick harness turns): only the four-stage 24-channel pipelines
    MI_PIPE_CASE(6, 4)
#else
    MI_PIPE_CASE(4, 2) MI_PIPE_CASE(4, 3) MI_PIPE_CASE(4, 4) MI_PIPE_CASE(6, 2) MI_PIPE_CASE(6, 3) MI_PIPE_CASE(6, 4)
#endif
#undef MI_PIPE_CASE
#define MI_TAIL_CASE(KBV, NH) \
    if (a.C == 24 && n == KBV && nh2 == NH && relu) return launch_pipe_inst<6, KBV, true, NH>(pa, s);
#ifdef MI_DEV_ONE
    MI_TAIL_CASE(4, 1) MI_TAIL_CASE(4, 2)
#else
    MI_TAIL_CASE(2, 1) MI_TAIL_CASE(3, 1) MI_TAIL_CASE(4, 1) MI_TAIL_CASE(2, 2) MI_TAIL_CASE(3, 2) MI_TAIL_CASE(4, 2)
#endif
#undef MI_TAIL_CASE
    return (int)hipErrorInvalidValue;
}

}  // namespace mi
