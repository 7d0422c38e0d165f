// mdblock_kernels.hip — full_range's "double BlazeBlock" on the wide layers (96x96x32 and 48x48x48), both pointwise convs on the
// matrix cores with everything else in the matrix operand layout (the mstrip_kernels.hip scheme, two stages deep):
//
//   a = act1( W1 . (DW3x3(x) + b_dw1) + b1 )         depthwise, pointwise C -> Cm
//   y = act2( W2 . (DW3x3(a) + b_dw2) + b2 + x )     depthwise, pointwise Cm -> Co (Co >= C: the skip is x zero-padded to Co channels)
//
// (face_detection_full_range.tflite: DEPTHWISE_CONV_2D, CONV_2D + fused RELU, DEPTHWISE_CONV_2D, CONV_2D, ADD, RELU behind
// `interpreter.invoke()`, /root/reference/src/face_detection_lite/face_detection.rs:235)
//
// dblock_kernels.hip runs these with 32-pixel x 32-channel MFMA tiles out of LDS tensors: with Cm = 8 .. 16 three quarters of every
// stage-1 tile are padding, each channel chunk of each pixel costs 18 sixteen-byte LDS reads (taps + window), and its eight waves
// share the pixel groups of a band unevenly (in-kernel stamps, round 3: a wave waits at the stage barriers for a third of a band):
// 0.16 ms per 96x96 block at 128 frames, 1.9 TB/s, for 0.3 GB of traffic.  Here:
//   * a workgroup = NWV waves side by side on one band of rows of one frame, each wave owning WT 16-pixel tiles of the row, all walking
//     down the band in step: per step one row of x arrives (LDS-DMA, every wave its own part), stage 1 finishes row r-1 of `a`,
//     stage 2 finishes row r-2 of `y`.  Two workgroup barriers per step (x row complete, a row complete), equal work per wave.
//   * lane (kq = lane / 16, p = lane % 16) owns channel 4 ks + kq of pixel p of a tile for every k-step: the depthwise 3x3 runs in
//     that layout with its taps in registers and vertical reuse in registers (an input row is read from LDS once: three 4-byte reads
//     per k-step and tile), and its result IS the B operand of v_mfma_f32_16x16x4_f32.  M = 16: Cm = 8 .. 16 is one tile.
//   * stage 1's result tile (4 consecutive channels of a pixel per lane) is written to a small LDS row image of `a` (the only
//     transposition), which stage 2 reads back in the operand layout, neighbours from the adjacent waves included.
//   * the skip of an output row is read from the x row image while that row is current and waits in the accumulator registers for
//     the two steps until its products arrive.
// Exact f32; results match dblock_kernels.hip to reassociation of the sums and the folded depthwise biases.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "kernels.hpp"
#include "launch.hpp"
#include "mrow.hpp"

namespace mi {

namespace {

struct MdbArgs {
    const float* in;
    float* out;
    const float* consts;   // mdblock_pack_consts()
    long in_fs, out_fs;
    int B, H;
    int bands, band_rows;
    float hi1, hi2;        // upper clamps of the two activations (6 for ReLU6, +inf otherwise)
    // STEM (round 6): `in` does not exist — x = act0(conv3x3 stride 2 (img) + b0) is computed row by row from the picture itself
    const float* sin;      // [B][2 H][2 W][3] f32
    const float* sconsts;  // mdblock_pack_stem(): A0 [7][64] | bias0 [16] | slope0 [16]
    long sin_fs;
    float hi0;
};

// CK1 = C / 4 k-steps of stage 1, CK2 = Cm / 4 of stage 2, MT2 = Co / 16 output tiles, WT pixel tiles per wave, NWV waves per workgroup
// DPX: pixels per LDS-DMA instruction; NF: frames per workgroup (NF x NWV waves share the constants); PW1: stage 1 is pointwise only
// (the bottleneck pair of the iris network: a = act1(W1 . x + b1), mbneck_kernel below)
// PAIR: two plain BlazeBlocks in a row instead of the double block — a = act1(.. + x), y = act2(.. + a): each stage adds its own input
// STEM: the network's first convolution (3 x 3, stride 2, RGB -> C = 16, SAME: one zero column / row behind the picture) runs inside this launch: a wave
// computes ITS 16 WT pixels of x row r + 1 while it works on row r — the 27 values of a pixel's window are the contraction (7 k-steps of
// v_mfma_f32_16x16x4_f32, B operands straight from global memory: lane (kq, p) reads value 4 ks + kq of pixel p's window), the result tile (4 consecutive
// channels of a pixel per lane) goes to the x row image the DMA would have filled.  The 96 x 96 x 16 tensor is never written nor read.
template <int CK1_, int CK2_, int MT2_, int WT_, int NWV_, bool TAPL2_ = false, int DPX_ = 4, int NF_ = 1, bool PW1_ = false, bool PAIR_ = false, bool STEM_ = false>
struct MD {
    static constexpr bool PAIR = PAIR_, STEM = STEM_;
    static constexpr int SK = 7, STEM_F = STEM_ ? SK * 64 + 32 : 0;   // stem constants behind the row images
    static constexpr int CK1 = CK1_, CK2 = CK2_, MT2 = MT2_, WT = WT_, NWV = NWV_, NF = NF_;
    static constexpr bool TAPL2 = TAPL2_;   // stage 2's taps stay in LDS (the registers do not hold 9 x (CK1 + CK2) taps beside the accumulators)
    static constexpr bool PW1 = PW1_;
    static constexpr int C = 4 * CK1, Cm = 4 * CK2, Co = 16 * MT2, W = 16 * WT * NWV, SW = 16 * WT;
    static constexpr int MT1 = (Cm + 15) / 16;
    static constexpr int QP = CK1 + 1, PS = 4 * QP;                             // x row image: float4 slots / floats per pixel
    static constexpr int QPA = ((CK2 + 1) & 1) ? CK2 + 1 : CK2 + 2, PSA = 4 * QPA;  // a row image (odd slot count: the 16 pixels of a tile on distinct banks)
    static_assert(QP % 2 == 1, "x pixel stride: odd number of float4 slots");
    static constexpr int XIMG_F = (W + 2) * PS, AIMG_F = (W + 2) * PSA;
    static constexpr int DPX = DPX_, NLD = SW / DPX, ACTIVE = DPX * QP;         // LDS-DMA: pixels per instruction, instructions per row and wave, lanes in use
    static_assert(ACTIVE > 32 && ACTIVE <= 64 && (NLD == 4 || NLD == 8), "DMA shape");
    static_assert((NLD - 1) * DPX * C * 4 < 4096, "immediate offsets of the DMA instructions");
    // constants blob (floats): A1 [CK1][MT1][64] | A2 [CK2][MT2][64] | taps1 [CK1][4][12] | taps2 [CK2][4][12] | bias1 [16 MT1] | slope1 [16 MT1] | bias2 [Co] | slope2 [Co]
    static constexpr int OFF_A1 = 0, A1_F = CK1 * MT1 * 64, OFF_A2 = A1_F, A2_F = CK2 * MT2 * 64, OFF_T1 = OFF_A2 + A2_F, T1_F = PW1 ? 0 : CK1 * 48, OFF_T2 = OFF_T1 + T1_F, T2_F = CK2 * 48;
    static constexpr int OFF_B1 = OFF_T2 + T2_F, OFF_S1 = OFF_B1 + 16 * MT1, OFF_B2 = OFF_S1 + 16 * MT1, OFF_S2 = OFF_B2 + Co, TOTAL = OFF_S2 + Co;
    static constexpr int LDS_F = TOTAL + NF * (2 * XIMG_F + 2 * AIMG_F) + STEM_F;
    static_assert(!STEM_ || (CK1_ == 4 && NF_ == 1), "stem: 16 output channels");
};

template <class K, bool RELU>
__global__ __launch_bounds__(K::NWV * 64, 2) void mdblock_kernel(MdbArgs a) {
    constexpr int CK1 = K::CK1, CK2 = K::CK2, MT1 = K::MT1, MT2 = K::MT2, WT = K::WT, NWV = K::NWV, C = K::C, Co = K::Co, W = K::W;
    constexpr int PS = K::PS, PSA = K::PSA, QP = K::QP, QPA = K::QPA, XIMG_F = K::XIMG_F, AIMG_F = K::AIMG_F, NLD = K::NLD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane >> 4, p = lane & 15;
    float* wgc = lds;                       // the constants blob
    float* ximg = lds + K::TOTAL;           // [2][XIMG_F]: row images of x, pixel 0 and W + 1 = zero border
    float* aimg = ximg + 2 * XIMG_F;        // [2][AIMG_F]: row images of a
    // workgroup = (band, frame): the frames of one band are neighbours in the grid
    const int band = blockIdx.x / a.B, b = blockIdx.x - band * a.B;
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.H);
    const int x0 = wave * K::SW;            // this wave's pixels [x0, x0 + SW)
    const float* in = K::STEM ? nullptr : a.in + (long)b * a.in_fs;
    // ---- STEM: this wave's 16 WT pixels of an x row from the picture
    float* wst = aimg + 2 * AIMG_F;         // [A0 SK x 64][bias0 16][slope0 16]
    const float* sin = K::STEM ? a.sin + (long)b * a.sin_fs : nullptr;
    constexpr int SKN = K::STEM ? K::SK : 1, SWT = K::STEM ? WT : 1;
    float sb[SWT][SKN];                     // B operands of the row in flight: issued at the top of a step, consumed at its end
    // Per lane and k-step, fixed for the whole launch: the byte offset of value 4 ks + kq of pixel x0 + p's window from the start of picture row 2 r.  The loads
    // are buffer loads — resource = this frame's picture, scalar offset = the row, vector offset = that register [+ 384 for the second tile]: no address
    // arithmetic per row, and the zero ROW behind the picture (ky = 2 of x row H - 1) is what a buffer load returns behind the end of its resource.  The zero
    // COLUMN (kx = 2 of the picture's last pixel: the last lane of the last wave's last tile) is the next row's first pixel in memory: a bit mask over the
    // k-steps replaces those values at the MFMA.  (The first version computed (ky, kx, channel) and both conditions per row: 250 VALU operations a row,
    // 0.284 ms against 0.277 for the two launches.)
    int voff[SKN];
    unsigned zcm = 0;
    if constexpr (K::STEM) {
#pragma unroll
        for (int ks = 0; ks < SKN; ks++) {
            const int k = min(4 * ks + kq, 26);           // (value 27 does not exist: its A operand is zero)
            const int ky = k >= 18 ? 2 : (k >= 9 ? 1 : 0), kr = k - 9 * ky;
            voff[ks] = (ky * (2 * W * 3) + kr + 6 * (x0 + p)) * 4;
            zcm |= (kr >= 6 && x0 + 16 * (WT - 1) + p == W - 1 ? 1u : 0u) << ks;
        }
    }
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K::STEM ? sin : a.consts), 0, K::STEM ? 2 * a.H * 2 * W * 3 * 4 : 0, 0x00020000);
    auto stem_load = [&](int rr) {
        const int rc = min(max(rr, 0), a.H - 1);   // (rows outside the image: any row — the result is replaced by zeros)
        const int soff = 2 * rc * (2 * W * 3) * 4;
#pragma unroll
        for (int ks = 0; ks < SKN; ks++)
#pragma unroll
            for (int nt = 0; nt < SWT; nt++) sb[nt][ks] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srsrc, voff[ks] + 384 * nt, soff, 0));
    };
    auto stem_emit = [&](int rr, int slot) {
        float* dstp = ximg + slot * XIMG_F + (1 + x0 + p) * PS + 4 * kq;
        const bool inside = rr >= 0 && rr < a.H;   // rows outside the image are the depthwise convolution's zero padding
        const float4 bs = *reinterpret_cast<const float4*>(wst + K::SK * 64 + 4 * kq), sl = *reinterpret_cast<const float4*>(wst + K::SK * 64 + 16 + 4 * kq);
        df32x4 D0[SWT];
#pragma unroll
        for (int nt = 0; nt < SWT; nt++) D0[nt] = df32x4{bs.x, bs.y, bs.z, bs.w};
#pragma unroll
        for (int ks = 0; ks < SKN; ks++) {
            const float a0 = wst[ks * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < SWT; nt++) {
                const bool z = nt == SWT - 1 && ((zcm >> ks) & 1u) != 0u;
                D0[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, z ? 0.f : sb[nt][ks], D0[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < SWT; nt++) {
            const df32x4 v = D0[nt];
            float4 o = make_float4(fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), a.hi0), fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), a.hi0),
                                   fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), a.hi0), fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), a.hi0));
            if (!inside) o = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(dstp + 16 * nt * PS) = o;
        }
    };

    // ---- LDS-DMA of this wave's part of one input row: instruction k brings in pixels [x0 + 4 k, x0 + 4 k + 4): lane -> (pixel
    // lane / QP, quad min(lane % QP, CK1 - 1)), lanes >= 4 QP idle; the immediate offset moves source and destination alike, M0 makes
    // up the difference between the image's pixel stride and the tensor's
    const int goff = ((lane / QP) * C + 4 * min(lane % QP, CK1 - 1)) * 4;
    const unsigned lds_x = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)ximg);
    auto issue_row = [&](int r, int slot) {
        if constexpr (K::STEM) return;
        const char* src = reinterpret_cast<const char*>(in + ((long)min(max(r, 0), a.H - 1) * W + x0) * C);
        const unsigned dstb = lds_x + (unsigned)((slot * XIMG_F + (1 + x0) * PS) * 4);
        unsigned long long saved;
#define MI_MDB_DMA(k) "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 offset:" #k "*%6\n\t"
        if constexpr (NLD == 8) {
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, %4\n\t"
                         "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                         MI_MDB_DMA(1) MI_MDB_DMA(2) MI_MDB_DMA(3) MI_MDB_DMA(4) MI_MDB_DMA(5) MI_MDB_DMA(6) MI_MDB_DMA(7)
                         "s_mov_b64 exec, %0"
                         : "=&s"(saved)
                         : "s"(src), "v"(goff), "s"(dstb), "n"((1u << (K::ACTIVE - 32)) - 1), "n"(K::DPX * (PS - C) * 4), "n"(K::DPX * C * 4)
                         : "memory", "scc", "m0");
        } else {
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, %4\n\t"
                         "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                         MI_MDB_DMA(1) MI_MDB_DMA(2) MI_MDB_DMA(3)
                         "s_mov_b64 exec, %0"
                         : "=&s"(saved)
                         : "s"(src), "v"(goff), "s"(dstb), "n"((1u << (K::ACTIVE - 32)) - 1), "n"(K::DPX * (PS - C) * 4), "n"(K::DPX * C * 4)
                         : "memory", "scc", "m0");
        }
#undef MI_MDB_DMA
    };
    // rows outside the image are zero padding: the landed (clamped) row is cleared before anybody reads it
    auto fix_row = [&](int r, int slot) {
        if (K::STEM || (r >= 0 && r < a.H)) return;  // wave-uniform
        float zz = 0.f;
        asm volatile("" : "+v"(zz));
        const float4 z = make_float4(zz, zz, zz, zz);
        float* part = ximg + slot * XIMG_F + (1 + x0) * PS;
        constexpr int N4 = K::SW * QP;
#pragma unroll
        for (int k = 0; k < (N4 + 63) / 64; k++)
            if (64 * (k + 1) <= N4 || lane < N4 - 64 * k) *reinterpret_cast<float4*>(part + 4 * (lane + 64 * k)) = z;
    };
    // border pixel columns of the four images are never written afterwards: cleared once
    if (wave == 0) {
        if (lane < 2 * QP) {
            const int col = lane / QP, qd = lane - col * QP;
#pragma unroll
            for (int s = 0; s < 2; s++) *reinterpret_cast<float4*>(ximg + s * XIMG_F + col * (W + 1) * PS + 4 * qd) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (lane < 2 * QPA) {
            const int col = lane / QPA, qd = lane - col * QPA;
#pragma unroll
            for (int s = 0; s < 2; s++) *reinterpret_cast<float4*>(aimg + s * AIMG_F + col * (W + 1) * PSA + 4 * qd) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    // the first row is on its way while the constants are fetched; the second goes out behind them, and only it may still be in
    // flight when the row loop starts (vector-memory operations retire in issue order; the loop's counted waits assume that nothing
    // else is outstanding)
    issue_row(y0 - 2, 0);
    if constexpr (K::STEM) stem_load(y0 - 2);
    for (int i = threadIdx.x; i < K::TOTAL / 4; i += NWV * 64) reinterpret_cast<float4*>(wgc)[i] = reinterpret_cast<const float4*>(a.consts)[i];
    if constexpr (K::STEM)
        for (int i = threadIdx.x; i < K::STEM_F / 4; i += NWV * 64) reinterpret_cast<float4*>(wst)[i] = reinterpret_cast<const float4*>(a.sconsts)[i];
    issue_row(y0 - 1, 1);
    if constexpr (K::STEM) __syncthreads();   // (the constants: also drains the loads above)
    else { dwait_vm<NLD>(); dwg_barrier(); }
    asm volatile("" ::: "memory");
    if constexpr (K::STEM) stem_emit(y0 - 2, 0);   // x row y0 - 2: complete for everybody at step 0's barrier
    // depthwise taps of this lane's channels: tap[ks][t] = w_dw[t][4 ks + kq], resident in registers for the whole kernel
    dv2f tap1[CK1][5], tap2[K::TAPL2 ? 1 : CK2][5];
    dload_taps<CK1>(reinterpret_cast<const float4*>(wgc + K::OFF_T1) + kq * 3, tap1);
    if constexpr (!K::TAPL2) dload_taps<CK2>(reinterpret_cast<const float4*>(wgc + K::OFF_T2) + kq * 3, tap2);
    // ---- per-lane addresses
    typedef __attribute__((address_space(3))) float lfloat;
    const unsigned x_lds = (unsigned)(uintptr_t)(lfloat*)(ximg + (x0 + p) * PS + kq);      // B layout: left neighbour of pixel x0 + p, channel kq
    const unsigned a_lds = (unsigned)(uintptr_t)(lfloat*)(aimg + (x0 + p) * PSA + kq);
    const unsigned a1_lds = (unsigned)(uintptr_t)(lfloat*)(wgc + K::OFF_A1 + lane);          // A operands [ks][mt][lane]
    const unsigned a2_lds = (unsigned)(uintptr_t)(lfloat*)(wgc + K::OFF_A2 + lane);
    const float* sme = ximg + (1 + x0 + p) * PS + 4 * kq;     // D layout: centre pixel, channels 4 kq .. 4 kq + 3 of a 16-channel tile
    float* awr = aimg + (1 + x0 + p) * PSA + 4 * kq;
    const unsigned ooff = (unsigned)((x0 + p) * Co + 4 * kq) * 4u;   // bytes: + (16 nt * Co + 16 mt) * 4 for tile (mt, nt)

    RowAcc<WT> p1A[CK1], p1B[CK1], p2A[CK2], p2B[CK2];   // partial depthwise rows: roles alternate from step to step
#pragma unroll
    for (int ks = 0; ks < CK1; ks++) { p1A[ks].clear(); p1B[ks].clear(); }
#pragma unroll
    for (int ks = 0; ks < CK2; ks++) { p2A[ks].clear(); p2B[ks].clear(); }
    df32x4 D1[MT1][WT], PA[MT2][WT], PB[MT2][WT];   // PA / PB: accumulators of the output rows of even / odd steps, waiting with bias + skip inside
    // D1 <- bias1 (+ with PAIR the first block's skip: the centre pixels of the x row in image `slot`, for the a row of the same index)
    auto init_D1 = [&](int slot) {
#pragma unroll
        for (int mt = 0; mt < MT1; mt++) {
            const float4 bs = *reinterpret_cast<const float4*>(wgc + K::OFF_B1 + 16 * mt + 4 * kq);
#pragma unroll
            for (int nt = 0; nt < WT; nt++) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (K::PAIR && 16 * mt < C) x = *reinterpret_cast<const float4*>(sme + slot * XIMG_F + 16 * nt * PS + 16 * mt);
                D1[mt][nt] = df32x4{bs.x + x.x, bs.y + x.y, bs.z + x.z, bs.w + x.w};
            }
        }
    };
    // P <- bias2 + skip for the output row whose centre input row sits in x image `slot`
    auto init_P = [&](df32x4 (&P)[MT2][WT], int slot) {
        float4 bs[MT2], x[MT2][WT];
#pragma unroll
        for (int mt = 0; mt < MT2; mt++) bs[mt] = *reinterpret_cast<const float4*>(wgc + K::OFF_B2 + 16 * mt + 4 * kq);
#pragma unroll
        for (int mt = 0; mt < MT2; mt++)
#pragma unroll
            for (int nt = 0; nt < WT; nt++) {
                x[mt][nt] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (K::PAIR) {  // the second block's skip: the a row in image `slot` (same pixels, D layout)
                    if (16 * mt < K::Cm) x[mt][nt] = *reinterpret_cast<const float4*>(awr + slot * AIMG_F + 16 * nt * PSA + 16 * mt);
                } else if (16 * mt < C) {
                    x[mt][nt] = *reinterpret_cast<const float4*>(sme + slot * XIMG_F + 16 * nt * PS + 16 * mt);  // channels >= C: the zero channel-pad of the skip
                }
            }
#pragma unroll
        for (int mt = 0; mt < MT2; mt++)
#pragma unroll
            for (int nt = 0; nt < WT; nt++) P[mt][nt] = df32x4{x[mt][nt].x + bs[mt].x, x[mt][nt].y + bs[mt].y, x[mt][nt].z + bs[mt].z, x[mt][nt].w + bs[mt].w};
    };
    auto act = [&](df32x4 v, const float4& sl, float hi) {
        if (RELU) return df32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
        // act(v) = min(max(v,0) + slope*min(v,0), hi): ReLU (slope 0), PReLU (alpha), none (1), ReLU6 (hi = 6)
        return df32x4{fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), hi), fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), hi),
                      fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), hi), fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), hi)};
    };
    // a row q (image row; zeros outside the image: the next depthwise conv's padding) -> a image `slot`
    auto store_a = [&](int q, int slot) {
        const bool inside = q >= 0 && q < a.H;
#pragma unroll
        for (int mt = 0; mt < MT1; mt++) {
            float4 sl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!RELU) sl = *reinterpret_cast<const float4*>(wgc + K::OFF_S1 + 16 * mt + 4 * kq);
            if (16 * mt + 4 * kq < K::Cm) {
#pragma unroll
                for (int nt = 0; nt < WT; nt++) {
                    df32x4 v = act(D1[mt][nt], sl, a.hi1);
                    if (!inside) v = df32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<float4*>(awr + slot * AIMG_F + 16 * nt * PSA + 16 * mt) = make_float4(v.x, v.y, v.z, v.w);
                }
            }
        }
    };
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) df32x4 gf32x4;
    auto epilogue = [&](df32x4 (&P)[MT2][WT], int y) {
        gchar* dst = (gchar*)(a.out + (long)b * a.out_fs + (long)y * W * Co);
        asm volatile("" : "+s"(dst));
#pragma unroll
        for (int mt = 0; mt < MT2; mt++) {
            float4 sl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!RELU) sl = *reinterpret_cast<const float4*>(wgc + K::OFF_S2 + 16 * mt + 4 * kq);
#pragma unroll
            for (int nt = 0; nt < WT; nt++) *(gf32x4*)(dst + ooff + (unsigned)((16 * nt * Co + 16 * mt) * 4)) = act(P[mt][nt], sl, a.hi2);
        }
    };

    // vmcnt bookkeeping: vector-memory operations retire in issue order.  A step's operations: the stores of its output row (NST, from
    // the fifth step on), then the DMA of the row two steps ahead (NLD, while the band has one).  "Row r has landed" = at most what was
    // issued behind its DMA is outstanding: the previous step's stores and DMA.
    constexpr int NST = MT2 * WT;
    const int TL = (y1 - y0) + 3;   // last step; step t handles x row y0 - 2 + t, a row y0 - 3 + t (from t = 2), y row y0 - 4 + t (from t = 4)
    auto wait_row = [&](int t) {
        if constexpr (K::STEM) return;   // (no DMA: the compiler counts the loads and stores it sees)
        const bool st = t - 1 >= 4, dm = t == 0 || t + 1 <= TL;  // step 0: the prologue's second row is behind it
        if (st && dm) dwait_vm<NST + NLD>();
        else if (dm) dwait_vm<NLD>();
        else if (st) dwait_vm<NST>();
        else dwait_vm<0>();
    };
    // e1 / e2: stage 1 / stage 2 finish a row in this step
    auto step = [&](auto e1c, auto e2c, int t, RowAcc<WT> (&q1PN)[CK1], RowAcc<WT> (&q1C)[CK1], RowAcc<WT> (&q2PN)[CK2], RowAcc<WT> (&q2C)[CK2], df32x4 (&P)[MT2][WT]) {
        constexpr bool E1 = decltype(e1c)::value, E2 = decltype(e2c)::value;
        const int r = y0 - 2 + t, slot = t & 1;
        if constexpr (K::STEM) stem_load(r + 1);   // the picture's values for x row r + 1: on their way during this step
        wait_row(t);
        fix_row(r, slot);
        dwg_barrier();   // x row r is complete (every wave's part has landed and is fixed / with STEM: was written in the previous step)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (E1 && !K::PAIR) init_D1(slot);
        mdb_row<CK1, MT1, WT, PS, E1>(x_lds + (unsigned)(slot * XIMG_F * 4), a1_lds, tap1, q1PN, q1C, D1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (K::PAIR) {
            // the next a row (r) starts from bias + ITS skip, the centre pixels of x row r, and waits in D1 for one step
            if constexpr (E1) store_a(r - 1, slot);
            init_D1(slot);
            if constexpr (E1) {
                dwg_barrier();   // a row r - 1 is complete
                __builtin_amdgcn_sched_barrier(0);
                // output row r - 2 starts from bias + the a row r - 2 (finished in the previous step, the other a image)
                if constexpr (E2) init_P(P, slot ^ 1);
            }
        } else if constexpr (E1) {
            store_a(r - 1, slot);
            dwg_barrier();   // a row r - 1 is complete
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (E1) {
            mdb_row<CK2, MT2, WT, PSA, E2, K::TAPL2>(a_lds + (unsigned)(slot * AIMG_F * 4), a2_lds, tap2, q2PN, q2C, P, reinterpret_cast<const float4*>(wgc + K::OFF_T2) + kq * 3);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (E2) epilogue(P, r - 2);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            dwg_barrier();   // priming steps: the neighbours have read this row's halo pixels before this wave's next DMA lands on them
        }
        if constexpr (!K::PAIR) init_P(P, slot);   // output row r starts from bias + its skip, the centre pixels of x row r
        __builtin_amdgcn_sched_barrier(0);
        dwave_sync();      // every read of x image `slot` by this wave is issued before the DMA below overwrites this wave's part of it
        __builtin_amdgcn_s_waitcnt(0xC07F);  // ... and has returned (the other waves read row r's halo pixels before the a-row barrier)
        if constexpr (K::STEM) {
            // x row r + 1 -> the other image: its last readers (row r - 1: stage 1 and the skips) passed the previous step's a-row barrier
            stem_emit(r + 1, slot ^ 1);   // (also in the last step: a row nobody reads)
        } else if (t + 2 <= TL) issue_row(r + 2, slot);
    };
    const std::false_type F{};
    const std::true_type T{};
    step(F, F, 0, p1A, p1B, p2A, p2B, PA);
    step(F, F, 1, p1B, p1A, p2B, p2A, PB);
    step(T, F, 2, p1A, p1B, p2A, p2B, PA);
    step(T, F, 3, p1B, p1A, p2B, p2A, PB);
    for (int t = 4; t <= TL; t += 2) {
        step(T, T, t, p1A, p1B, p2A, p2B, PA);
        if (t + 1 > TL) break;
        step(T, T, t + 1, p1B, p1A, p2B, p2A, PB);
    }
}

// ---- the bottleneck pair  a = act1(W1 . x + b1);  y = act2(W2 . (DW3x3(a) + b_dw) + b2 + x)  (iris_landmark.tflite: 64 -> 32 -> 64 on 32 x 32;
// CONV_2D 1x1 + PRELU, DEPTHWISE_CONV_2D, CONV_2D 1x1, ADD, PRELU behind `interpreter.invoke()`, /root/reference/src/face_detection_lite/
// iris_landmark.rs:203) in the same scheme: stage 1 has no depthwise part, so row r of `a` is finished in the step that
// brings row r of x (its B operand is the row image's centre pixels as they are), stage 2 finishes output row r - 1, whose bias + skip
// waited in the other accumulator set for one step.  A workgroup = NF frames x NWV waves on one band; x is read only by the wave that
// fetched it, so the step has ONE workgroup barrier (the `a` row is complete).
template <int CK, int MT, int WT, int PSV>
__device__ __forceinline__ void mpw_row(const unsigned src, const unsigned aop, df32x4 (&D)[MT][WT]) {
    float xs[2][WT], av[2][MT];
    auto load_ks = [&](auto ksc, float (&x)[WT], float (&aw)[MT]) {
        constexpr int ks = decltype(ksc)::value;
        const unsigned xa = src, aa = aop;
#pragma unroll
        for (int nt = 0; nt < WT; nt++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[nt]) : "v"(xa), "n"(4 * ks * 4 + nt * 16 * PSV * 4));
#pragma unroll
        for (int mt = 0; mt < MT; mt++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(aw[mt]) : "v"(aa), "n"((ks * MT + mt) * 64 * 4));
    };
    auto kstep = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        if constexpr (ks + 1 < CK) load_ks(std::integral_constant<int, ks + 1>{}, xs[(ks + 1) & 1], av[(ks + 1) & 1]);
        float (&x)[WT] = xs[ks & 1];
        float (&aw)[MT] = av[ks & 1];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < WT; nt++) D[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt], x[nt], D[mt][nt], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    load_ks(std::integral_constant<int, 0>{}, xs[0], av[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    dfor_each(kstep, std::make_integer_sequence<int, CK>{});
}

template <class K, bool RELU>
__global__ __launch_bounds__(K::NF * K::NWV * 64, 2) void mbneck_kernel(MdbArgs a) {
    static_assert(K::PW1 && !K::TAPL2, "pointwise first stage, stage-2 taps in registers");
    constexpr int CK1 = K::CK1, CK2 = K::CK2, MT1 = K::MT1, MT2 = K::MT2, WT = K::WT, NWV = K::NWV, NF = K::NF, C = K::C, Co = K::Co, W = K::W;
    constexpr int PS = K::PS, PSA = K::PSA, QP = K::QP, QPA = K::QPA, XIMG_F = K::XIMG_F, AIMG_F = K::AIMG_F, NLD = K::NLD;
    static_assert(NLD == 8, "eight DMA instructions per row and wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = wave / NWV, strip = wave - fr * NWV;
    const int kq = lane >> 4, p = lane & 15;
    float* wgc = lds;
    float* ximg = lds + K::TOTAL + fr * (2 * XIMG_F + 2 * AIMG_F);   // this frame's [2][XIMG_F] row images of x ...
    float* aimg = ximg + 2 * XIMG_F;                                  // ... and [2][AIMG_F] of a
    // workgroup = (band, group of NF frames); the last group of an odd batch repeats the last frame (same values stored twice)
    const int groups = (a.B + NF - 1) / NF;
    const int band = blockIdx.x / groups, b = min((blockIdx.x - band * groups) * NF + fr, a.B - 1);
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, a.H);
    const int x0 = strip * K::SW;
    const float* in = a.in + (long)b * a.in_fs;

    const int goff = ((lane / QP) * C + 4 * min(lane % QP, CK1 - 1)) * 4;
    const unsigned lds_x = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)ximg);
    auto issue_row = [&](int r, int slot) {
        const char* src = reinterpret_cast<const char*>(in + ((long)min(max(r, 0), a.H - 1) * W + x0) * C);
        const unsigned dstb = lds_x + (unsigned)((slot * XIMG_F + (1 + x0) * PS) * 4);
        unsigned long long saved;
#define MI_MDB_DMA(k) "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 offset:" #k "*%6\n\t"
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, %4\n\t"
                     "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
                     MI_MDB_DMA(1) MI_MDB_DMA(2) MI_MDB_DMA(3) MI_MDB_DMA(4) MI_MDB_DMA(5) MI_MDB_DMA(6) MI_MDB_DMA(7)
                     "s_mov_b64 exec, %0"
                     : "=&s"(saved)
                     : "s"(src), "v"(goff), "s"(dstb), "n"((1u << (K::ACTIVE - 32)) - 1), "n"(K::DPX * (PS - C) * 4), "n"(K::DPX * C * 4)
                     : "memory", "scc", "m0");
#undef MI_MDB_DMA
    };
    auto fix_row = [&](int r, int slot) {
        if (r >= 0 && r < a.H) return;  // wave-uniform
        float zz = 0.f;
        asm volatile("" : "+v"(zz));
        const float4 z = make_float4(zz, zz, zz, zz);
        float* part = ximg + slot * XIMG_F + (1 + x0) * PS;
        constexpr int N4 = K::SW * QP;
#pragma unroll
        for (int k = 0; k < (N4 + 63) / 64; k++)
            if (64 * (k + 1) <= N4 || lane < N4 - 64 * k) *reinterpret_cast<float4*>(part + 4 * (lane + 64 * k)) = z;
    };
    // border pixel columns of the a images are never written afterwards: cleared once (x's border columns are never read: stage 1 is pointwise)
    if (strip == 0 && lane < 2 * QPA) {
        const int col = lane / QPA, qd = lane - col * QPA;
#pragma unroll
        for (int s = 0; s < 2; s++) *reinterpret_cast<float4*>(aimg + s * AIMG_F + col * (W + 1) * PSA + 4 * qd) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    issue_row(y0 - 1, 0);
    for (int i = threadIdx.x; i < K::TOTAL / 4; i += NF * NWV * 64) reinterpret_cast<float4*>(wgc)[i] = reinterpret_cast<const float4*>(a.consts)[i];
    issue_row(y0, 1);
    dwait_vm<NLD>();
    dwg_barrier();
    asm volatile("" ::: "memory");
    dv2f tap2[CK2][5];
    dload_taps<CK2>(reinterpret_cast<const float4*>(wgc + K::OFF_T2) + kq * 3, tap2);
    typedef __attribute__((address_space(3))) float lfloat;
    const unsigned xc_lds = (unsigned)(uintptr_t)(lfloat*)(ximg + (1 + x0 + p) * PS + kq);   // B layout: pixel x0 + p itself, channel kq
    const unsigned a_lds = (unsigned)(uintptr_t)(lfloat*)(aimg + (x0 + p) * PSA + kq);         // ... its left neighbour in the a image
    const unsigned a1_lds = (unsigned)(uintptr_t)(lfloat*)(wgc + K::OFF_A1 + lane);
    const unsigned a2_lds = (unsigned)(uintptr_t)(lfloat*)(wgc + K::OFF_A2 + lane);
    const float* sme = ximg + (1 + x0 + p) * PS + 4 * kq;
    float* awr = aimg + (1 + x0 + p) * PSA + 4 * kq;
    const unsigned ooff = (unsigned)((x0 + p) * Co + 4 * kq) * 4u;

    RowAcc<WT> p2A[CK2], p2B[CK2];
#pragma unroll
    for (int ks = 0; ks < CK2; ks++) { p2A[ks].clear(); p2B[ks].clear(); }
    df32x4 D1[MT1][WT], PA[MT2][WT], PB[MT2][WT];
    auto init_D1 = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT1; mt++) {
            const float4 bs = *reinterpret_cast<const float4*>(wgc + K::OFF_B1 + 16 * mt + 4 * kq);
#pragma unroll
            for (int nt = 0; nt < WT; nt++) D1[mt][nt] = df32x4{bs.x, bs.y, bs.z, bs.w};
        }
    };
    auto init_P = [&](df32x4 (&P)[MT2][WT], int slot) {
        float4 bs[MT2], x[MT2][WT];
#pragma unroll
        for (int mt = 0; mt < MT2; mt++) bs[mt] = *reinterpret_cast<const float4*>(wgc + K::OFF_B2 + 16 * mt + 4 * kq);
#pragma unroll
        for (int mt = 0; mt < MT2; mt++)
#pragma unroll
            for (int nt = 0; nt < WT; nt++) {
                x[mt][nt] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (16 * mt < C) x[mt][nt] = *reinterpret_cast<const float4*>(sme + slot * XIMG_F + 16 * nt * PS + 16 * mt);
            }
#pragma unroll
        for (int mt = 0; mt < MT2; mt++)
#pragma unroll
            for (int nt = 0; nt < WT; nt++) P[mt][nt] = df32x4{x[mt][nt].x + bs[mt].x, x[mt][nt].y + bs[mt].y, x[mt][nt].z + bs[mt].z, x[mt][nt].w + bs[mt].w};
    };
    auto act = [&](df32x4 v, const float4& sl, float hi) {
        if (RELU) return df32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
        return df32x4{fminf(fmaxf(v.x, 0.f) + sl.x * fminf(v.x, 0.f), hi), fminf(fmaxf(v.y, 0.f) + sl.y * fminf(v.y, 0.f), hi),
                      fminf(fmaxf(v.z, 0.f) + sl.z * fminf(v.z, 0.f), hi), fminf(fmaxf(v.w, 0.f) + sl.w * fminf(v.w, 0.f), hi)};
    };
    auto store_a = [&](int q, int slot) {
        const bool inside = q >= 0 && q < a.H;
#pragma unroll
        for (int mt = 0; mt < MT1; mt++) {
            float4 sl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!RELU) sl = *reinterpret_cast<const float4*>(wgc + K::OFF_S1 + 16 * mt + 4 * kq);
            if (16 * mt + 4 * kq < K::Cm) {
#pragma unroll
                for (int nt = 0; nt < WT; nt++) {
                    df32x4 v = act(D1[mt][nt], sl, a.hi1);
                    if (!inside) v = df32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<float4*>(awr + slot * AIMG_F + 16 * nt * PSA + 16 * mt) = make_float4(v.x, v.y, v.z, v.w);
                }
            }
        }
    };
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) df32x4 gf32x4;
    auto epilogue = [&](df32x4 (&P)[MT2][WT], int y) {
        gchar* dst = (gchar*)(a.out + (long)b * a.out_fs + (long)y * W * Co);
        asm volatile("" : "+s"(dst));
#pragma unroll
        for (int mt = 0; mt < MT2; mt++) {
            float4 sl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!RELU) sl = *reinterpret_cast<const float4*>(wgc + K::OFF_S2 + 16 * mt + 4 * kq);
#pragma unroll
            for (int nt = 0; nt < WT; nt++) *(gf32x4*)(dst + ooff + (unsigned)((16 * nt * Co + 16 * mt) * 4)) = act(P[mt][nt], sl, a.hi2);
        }
    };
    // step t handles x row and a row y0 - 1 + t, y row y0 - 2 + t (from t = 2).  vmcnt: the stores of a step's output row (NST, from the
    // third step on), then the DMA of the row two steps ahead; "row r has landed" = at most the previous step's stores and DMA are outstanding.
    constexpr int NST = MT2 * WT;
    const int TL = (y1 - y0) + 1;
    auto wait_row = [&](int t) {
        const bool st = t - 1 >= 2, dm = t == 0 || t + 1 <= TL;
        if (st && dm) dwait_vm<NST + NLD>();
        else if (dm) dwait_vm<NLD>();
        else if (st) dwait_vm<NST>();
        else dwait_vm<0>();
    };
    auto step = [&](auto e2c, int t, RowAcc<WT> (&q2PN)[CK2], RowAcc<WT> (&q2C)[CK2], df32x4 (&Pc)[MT2][WT], df32x4 (&Pi)[MT2][WT]) {
        constexpr bool E2 = decltype(e2c)::value;
        const int r = y0 - 1 + t, slot = t & 1;
        wait_row(t);
        fix_row(r, slot);
        dwave_sync();
        init_D1();
        mpw_row<CK1, MT1, WT, PS>(xc_lds + (unsigned)(slot * XIMG_F * 4), a1_lds, D1);
        __builtin_amdgcn_sched_barrier(0);
        store_a(r, slot);
        dwg_barrier();   // a row r is complete (the neighbours' pixels included)
        __builtin_amdgcn_sched_barrier(0);
        mdb_row<CK2, MT2, WT, PSA, E2>(a_lds + (unsigned)(slot * AIMG_F * 4), a2_lds, tap2, q2PN, q2C, Pc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (E2) epilogue(Pc, r - 1);
        __builtin_amdgcn_sched_barrier(0);
        init_P(Pi, slot);   // output row r starts from bias + its skip, the centre pixels of x row r
        __builtin_amdgcn_sched_barrier(0);
        dwave_sync();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (t + 2 <= TL) issue_row(r + 2, slot);
    };
    const std::false_type F{};
    const std::true_type T{};
    step(F, 0, p2A, p2B, PB, PA);
    step(F, 1, p2B, p2A, PA, PB);
    for (int t = 2; t <= TL; t += 2) {
        step(T, t, p2A, p2B, PB, PA);
        if (t + 1 > TL) break;
        step(T, t + 1, p2B, p2A, PA, PB);
    }
}

using MB32 = MD<16, 8, 4, 1, 2, false, 2, 2, true>;   // 32 x 32: 64 -> 32 -> 64 (iris), two frames per workgroup

template <class K>
void mbn_pack(const float* w_pw1, const float* b1, const float* alpha1, int act1, const float* w_dw2, const float* b_dw2, const float* w_pw2, const float* b2,
              const float* alpha2, int act2, float* dst) {
    constexpr int C = K::C, Cm = K::Cm, Co = K::Co;
    std::fill(dst, dst + K::TOTAL, 0.f);
    for (int ks = 0; ks < K::CK1; ks++)
        for (int mt = 0; mt < K::MT1; mt++)
            for (int l = 0; l < 64; l++) {
                const int o = 16 * mt + (l & 15);
                if (o < Cm) dst[K::OFF_A1 + (ks * K::MT1 + mt) * 64 + l] = w_pw1[(size_t)o * C + 4 * ks + (l >> 4)];
            }
    for (int ks = 0; ks < K::CK2; ks++)
        for (int mt = 0; mt < K::MT2; mt++)
            for (int l = 0; l < 64; l++) dst[K::OFF_A2 + (ks * K::MT2 + mt) * 64 + l] = w_pw2[(size_t)(16 * mt + (l & 15)) * Cm + 4 * ks + (l >> 4)];
    for (int ks = 0; ks < K::CK2; ks++)
        for (int kq = 0; kq < 4; kq++)
            for (int t = 0; t < 9; t++) dst[K::OFF_T2 + (ks * 4 + kq) * 12 + t] = w_dw2[t * Cm + 4 * ks + kq];
    for (int c = 0; c < Cm; c++) {
        dst[K::OFF_B1 + c] = b1 ? b1[c] : 0.f;
        dst[K::OFF_S1 + c] = act1 == ACT_PRELU ? alpha1[c] : (act1 == ACT_NONE ? 1.f : 0.f);
    }
    for (int c = 0; c < Co; c++) {
        double acc = b2 ? b2[c] : 0.0;
        if (b_dw2)
            for (int k = 0; k < Cm; k++) acc += (double)w_pw2[(size_t)c * Cm + k] * b_dw2[k];
        dst[K::OFF_B2 + c] = (float)acc;
        dst[K::OFF_S2 + c] = act2 == ACT_PRELU ? alpha2[c] : (act2 == ACT_NONE ? 1.f : 0.f);
    }
}

using MD96 = MD<8, 2, 2, 2, 3>;    // 96 x 96: 32 -> 8 -> 32
using MD48a = MD<12, 3, 3, 1, 3>;  // 48 x 48: 48 -> 12 -> 48
using MD48b = MD<12, 4, 4, 1, 3>;  // 48 x 48: 48 -> 16 -> 64
using MD48c = MD<12, 6, 3, 1, 3, true>;  // 48 x 48: 48 -> 24 -> 48
using MP96 = MD<4, 4, 1, 2, 3, false, 8, 1, false, true>;  // 96 x 96: two BlazeBlocks 16 -> 16 -> 16 (face mesh)
using MP48 = MD<8, 8, 2, 1, 3, true, 4, 1, false, true>;   // 48 x 48: two BlazeBlocks 32 -> 32 -> 32 (face mesh; round 6: the tensor between them is neither written nor read)
using MP96S = MD<4, 4, 1, 2, 3, true, 8, 1, false, true, true>;  // ... with the network's first convolution (192 x 192 x 3 -> 96 x 96 x 16) in front, in the same launch

template <class K>
void mdb_pack(const float* w_dw1, const float* b_dw1, const float* w_pw1, const float* b1, const float* alpha1, int act1, const float* w_dw2, const float* b_dw2,
              const float* w_pw2, const float* b2, const float* alpha2, int act2, float* dst) {
    constexpr int C = K::C, Cm = K::Cm, Co = K::Co;
    std::fill(dst, dst + K::TOTAL, 0.f);
    // A operand of v_mfma_f32_16x16x4_f32 for (k-step ks, output tile mt): lane l holds W[16 mt + l % 16][4 ks + l / 16]
    for (int ks = 0; ks < K::CK1; ks++)
        for (int mt = 0; mt < K::MT1; mt++)
            for (int l = 0; l < 64; l++) {
                const int o = 16 * mt + (l & 15);
                if (o < Cm) dst[K::OFF_A1 + (ks * K::MT1 + mt) * 64 + l] = w_pw1[(size_t)o * C + 4 * ks + (l >> 4)];
            }
    for (int ks = 0; ks < K::CK2; ks++)
        for (int mt = 0; mt < K::MT2; mt++)
            for (int l = 0; l < 64; l++) dst[K::OFF_A2 + (ks * K::MT2 + mt) * 64 + l] = w_pw2[(size_t)(16 * mt + (l & 15)) * Cm + 4 * ks + (l >> 4)];
    // taps [ks][kq][12]: the nine taps of channel 4 ks + kq (three float4 loads per k-step)
    for (int ks = 0; ks < K::CK1; ks++)
        for (int kq = 0; kq < 4; kq++)
            for (int t = 0; t < 9; t++) dst[K::OFF_T1 + (ks * 4 + kq) * 12 + t] = w_dw1[t * C + 4 * ks + kq];
    for (int ks = 0; ks < K::CK2; ks++)
        for (int kq = 0; kq < 4; kq++)
            for (int t = 0; t < 9; t++) dst[K::OFF_T2 + (ks * 4 + kq) * 12 + t] = w_dw2[t * Cm + 4 * ks + kq];
    // PW(dw + b_dw) + b_pw = PW(dw) + (W b_dw + b_pw): the depthwise biases are folded into the pointwise biases
    for (int c = 0; c < Cm; c++) {
        double acc = b1 ? b1[c] : 0.0;
        if (b_dw1)
            for (int k = 0; k < C; k++) acc += (double)w_pw1[(size_t)c * C + k] * b_dw1[k];
        dst[K::OFF_B1 + c] = (float)acc;
        dst[K::OFF_S1 + c] = act1 == ACT_PRELU ? alpha1[c] : (act1 == ACT_NONE ? 1.f : 0.f);
    }
    for (int c = 0; c < Co; c++) {
        double acc = b2 ? b2[c] : 0.0;
        if (b_dw2)
            for (int k = 0; k < Cm; k++) acc += (double)w_pw2[(size_t)c * Cm + k] * b_dw2[k];
        dst[K::OFF_B2 + c] = (float)acc;
        dst[K::OFF_S2 + c] = act2 == ACT_PRELU ? alpha2[c] : (act2 == ACT_NONE ? 1.f : 0.f);
    }
}

template <class K>
int mdb_launch(const DblockArgs& a, hipStream_t s) {
    MdbArgs ma;
    ma.in = a.in; ma.out = a.out; ma.consts = a.mconsts; ma.in_fs = a.in_fs; ma.out_fs = a.out_fs;
    ma.B = a.B; ma.H = a.H;
    ma.hi1 = a.hi1; ma.hi2 = a.hi2;
    ma.sin = a.stem_in; ma.sconsts = a.stem_consts; ma.sin_fs = a.stem_in_fs; ma.hi0 = a.stem_hi;
    // bands: enough workgroups for every CU to hold as many as its LDS takes; a band costs four priming steps
    static const int forced = getenv("MI_MDB_BAND") ? atoi(getenv("MI_MDB_BAND")) : 0;  // tuning aid
    // (the pair form on 16 channels needs ~160 registers: three waves per SIMD)
    const int per_cu = std::max(1, std::min((K::PAIR ? 12 : 8) / K::NWV, (int)((160 * 1024) / (K::LDS_F * 4))));
    long bands = std::max<long>(1, ((long)per_cu * device_cu_count() + a.B / 2) / std::max(1, a.B));
    int rows = (int)((a.H + bands - 1) / bands);
    rows = std::max(rows, std::min(a.H, 8));
    if (forced > 0) rows = std::min(forced, a.H);
    if (a.band_rows > 0) rows = std::min(a.band_rows, a.H);
    ma.band_rows = rows;
    ma.bands = (a.H + rows - 1) / rows;
    const bool relu = a.act1 == ACT_RELU && a.act2 == ACT_RELU;
    const dim3 grid((unsigned)((long)a.B * ma.bands));
    const size_t lds_bytes = (size_t)K::LDS_F * 4;
    if (relu) {
        auto kern = mdblock_kernel<K, true>;
        if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
        return (int)launch_kernel(kern, grid, dim3(K::NWV * 64), lds_bytes, s, ma);
    }
    auto kern = mdblock_kernel<K, false>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(kern, grid, dim3(K::NWV * 64), lds_bytes, s, ma);
}

// 0: none; 1: 96 wide 32 -> 8 -> 32; 2: 48 wide 48 -> 12 -> 48; 3: 48 wide 48 -> 16 -> 64; 4: 48 wide 48 -> 24 -> 48;
// 5: two plain BlazeBlocks (pair), 96 wide 16 -> 16 -> 16
int mdb_shape(int W, int C, int Cm, int Co, bool pair = false) {
    static const bool off = getenv("MI_NO_MDBLOCK") != nullptr;  // tuning aid: the LDS-tensor double-block kernel instead
    if (off) return 0;
    if (pair) return (W == 96 && C == 16 && Cm == 16 && Co == 16) ? 5 : ((W == 48 && C == 32 && Cm == 32 && Co == 32) ? 7 : 0);   // (6: shape 5 behind the first convolution, launch_mdblock)
    if (W == 96 && C == 32 && Cm == 8 && Co == 32) return 1;
    if (W == 48 && C == 48 && Cm == 12 && Co == 48) return 2;
    if (W == 48 && C == 48 && Cm == 16 && Co == 64) return 3;
    if (W == 48 && C == 48 && Cm == 24 && Co == 48) return 4;
    return 0;
}

}  // namespace

bool mdblock_shape_ok(int W, int C, int Cm, int Co, bool pair) { return mdb_shape(W, C, Cm, Co, pair) != 0; }

int mdblock_consts_floats(int W, int C, int Cm, int Co, bool pair) {
    switch (mdb_shape(W, C, Cm, Co, pair)) {
        case 5: return MP96::TOTAL;
        case 7: return MP48::TOTAL;
        case 1: return MD96::TOTAL;
        case 2: return MD48a::TOTAL;
        case 3: return MD48b::TOTAL;
        case 4: return MD48c::TOTAL;
    }
    return 0;
}

// w_dw [3][3][C], b_dw [C] or null, w_pw [Co][C] (TFLite OHWI with H = W = 1), bias [Co] or null, alpha [Co] or null — per stage
void mdblock_pack_consts(int W, int C, int Cm, int Co, const float* w_dw1, const float* b_dw1, const float* w_pw1, const float* b1, const float* alpha1, int act1,
                         const float* w_dw2, const float* b_dw2, const float* w_pw2, const float* b2, const float* alpha2, int act2, float* dst, bool pair) {
    switch (mdb_shape(W, C, Cm, Co, pair)) {
        case 5: mdb_pack<MP96>(w_dw1, b_dw1, w_pw1, b1, alpha1, act1, w_dw2, b_dw2, w_pw2, b2, alpha2, act2, dst); break;
        case 7: mdb_pack<MP48>(w_dw1, b_dw1, w_pw1, b1, alpha1, act1, w_dw2, b_dw2, w_pw2, b2, alpha2, act2, dst); break;
        case 1: mdb_pack<MD96>(w_dw1, b_dw1, w_pw1, b1, alpha1, act1, w_dw2, b_dw2, w_pw2, b2, alpha2, act2, dst); break;
        case 2: mdb_pack<MD48a>(w_dw1, b_dw1, w_pw1, b1, alpha1, act1, w_dw2, b_dw2, w_pw2, b2, alpha2, act2, dst); break;
        case 3: mdb_pack<MD48b>(w_dw1, b_dw1, w_pw1, b1, alpha1, act1, w_dw2, b_dw2, w_pw2, b2, alpha2, act2, dst); break;
        case 4: mdb_pack<MD48c>(w_dw1, b_dw1, w_pw1, b1, alpha1, act1, w_dw2, b_dw2, w_pw2, b2, alpha2, act2, dst); break;
    }
}

// the double block proper (no skip on the first half, the second half's skip is x) or a pair of plain BlazeBlocks (both skips), constants
// packed by mdblock_pack_consts()
bool mdblock_kernel_supports(const DblockArgs& a) {
    if (a.skip1 != a.skip2_from_a) return false;
    if (!a.mconsts || !mdb_shape(a.W, a.C, a.Cm, a.Co, a.skip1 != 0) || a.H < 2 || a.B < 1) return false;
    // the waves of a band walk it row by row: below about one workgroup per CU the launch is latency-bound and the wider kernel finishes sooner
    static const int min_b = getenv("MI_MDB_MIN_B") ? atoi(getenv("MI_MDB_MIN_B")) : 32;
    if (a.B < min_b) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (a.stem_in)   // the pair behind the first convolution, which runs inside the launch: `in` is not read
        return mdb_shape(a.W, a.C, a.Cm, a.Co, true) == 5 && a.H == 96 && a.stem_consts && aligned16(a.stem_in) && aligned16(a.stem_consts) && !(a.stem_in_fs & 3) &&
               aligned16(a.out) && aligned16(a.mconsts) && !(a.out_fs & 3);
    return aligned16(a.in) && aligned16(a.out) && aligned16(a.mconsts) && !(a.in_fs & 3) && !(a.out_fs & 3);
}

// the face mesh's first convolution as the pair kernel's stage 0 (MD::STEM): 3 x 3, stride 2, SAME, 192 x 192 x 3 -> 96 x 96 x 16
bool mdblock_stem_shape_ok(int H, int W, int C, int KH, int KW, int sh, int sw, int Ho, int Wo, int Co) {
    return H == 192 && W == 192 && C == 3 && KH == 3 && KW == 3 && sh == 2 && sw == 2 && Ho == 96 && Wo == 96 && Co == 16;
}
int mdblock_stem_consts_floats() { return MP96S::STEM_F; }
// w [16][3][3][3] (TFLite OHWI), bias [16] or null, alpha [16] or null
void mdblock_pack_stem(const float* w, const float* bias, const float* alpha, int act, float* dst) {
    std::fill(dst, dst + MP96S::STEM_F, 0.f);
    // A operand of k-step ks: lane l holds w[l % 16][4 ks + l / 16] — the window's 27 values in (ky, kx, channel) order, value 27 = 0
    for (int ks = 0; ks < MP96S::SK; ks++)
        for (int l = 0; l < 64; l++) {
            const int k = 4 * ks + (l >> 4);
            if (k < 27) dst[ks * 64 + l] = w[(size_t)(l & 15) * 27 + k];
        }
    for (int c = 0; c < 16; c++) {
        dst[MP96S::SK * 64 + c] = bias ? bias[c] : 0.f;
        dst[MP96S::SK * 64 + 16 + c] = act == ACT_PRELU ? alpha[c] : (act == ACT_NONE ? 1.f : 0.f);
    }
}

int launch_mdblock(const DblockArgs& a, void* stream) {
    if (!mdblock_kernel_supports(a)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    if (a.stem_in) return mdb_launch<MP96S>(a, s);
    switch (mdb_shape(a.W, a.C, a.Cm, a.Co, a.skip1 != 0)) {
        case 5: return mdb_launch<MP96>(a, s);
        case 7: return mdb_launch<MP48>(a, s);
        case 1: return mdb_launch<MD96>(a, s);
        case 2: return mdb_launch<MD48a>(a, s);
        case 3: return mdb_launch<MD48b>(a, s);
        case 4: return mdb_launch<MD48c>(a, s);
    }
    return (int)hipErrorInvalidValue;
}

// ---- bottleneck pairs (mbneck_kernel): 32-pixel-wide frames, 64 -> 32 -> 64 channels, one pair per launch
bool mbneck_shape_ok(int W, int C, int Cm) {
    static const bool off = getenv("MI_NO_MBNECK") != nullptr;  // tuning aid: bneck_kernels.hip instead
    return !off && W == 32 && C == 64 && Cm == 32;
}

int mbneck_consts_floats(int W, int C, int Cm) { return mbneck_shape_ok(W, C, Cm) ? MB32::TOTAL : 0; }

void mbneck_pack_consts(int W, int C, int Cm, const float* w_pw1, const float* b1, const float* alpha1, int act1, const float* w_dw2, const float* b_dw2,
                        const float* w_pw2, const float* b2, const float* alpha2, int act2, float* dst) {
    if (mbneck_shape_ok(W, C, Cm)) mbn_pack<MB32>(w_pw1, b1, alpha1, act1, w_dw2, b_dw2, w_pw2, b2, alpha2, act2, dst);
}

bool mbneck_kernel_supports(const BneckArgs& a) {
    if (a.nblocks != 1 || !a.blocks[0].mconsts || !mbneck_shape_ok(a.W, a.C, a.Cm) || a.H < 2) return false;
    static const int min_b = getenv("MI_MDB_MIN_B") ? atoi(getenv("MI_MDB_MIN_B")) : 32;
    if (a.B < min_b) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return aligned16(a.in) && aligned16(a.out) && aligned16(a.blocks[0].mconsts) && !(a.in_fs & 3) && !(a.out_fs & 3);
}

int launch_mbneck(const BneckArgs& a, void* stream) {
    if (!mbneck_kernel_supports(a)) return (int)hipErrorInvalidValue;
    using K = MB32;
    MdbArgs ma;
    ma.in = a.in; ma.out = a.out; ma.consts = a.blocks[0].mconsts; ma.in_fs = a.in_fs; ma.out_fs = a.out_fs;
    ma.B = a.B; ma.H = a.H;
    ma.hi1 = a.blocks[0].hi1; ma.hi2 = a.blocks[0].hi2;
    static const int forced = getenv("MI_MBN_BAND") ? atoi(getenv("MI_MBN_BAND")) : 0;  // tuning aid
    const int groups = (a.B + K::NF - 1) / K::NF;
    const int per_cu = std::max(1, std::min(8 / (K::NF * K::NWV), (int)((160 * 1024) / (K::LDS_F * 4))));
    long bands = std::max<long>(1, ((long)per_cu * device_cu_count() + groups / 2) / std::max(1, groups));
    int rows = (int)((a.H + bands - 1) / bands);
    rows = std::max(rows, std::min(a.H, 8));
    if (forced > 0) rows = std::min(forced, a.H);
    ma.band_rows = rows;
    ma.bands = (a.H + rows - 1) / rows;
    const bool relu = a.blocks[0].act1 == ACT_RELU && a.blocks[0].act2 == ACT_RELU;
    const dim3 grid((unsigned)((long)groups * ma.bands));
    const size_t lds_bytes = (size_t)K::LDS_F * 4;
    hipStream_t s = (hipStream_t)stream;
    if (relu) {
        auto kern = mbneck_kernel<K, true>;
        if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
        return (int)launch_kernel(kern, grid, dim3(K::NF * K::NWV * 64), lds_bytes, s, ma);
    }
    auto kern = mbneck_kernel<K, false>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return (int)e;
    return (int)launch_kernel(kern, grid, dim3(K::NF * K::NWV * 64), lds_bytes, s, ma);
}

}  // namespace mi
